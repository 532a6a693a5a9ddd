// MPV.py:351-454 planar convention with PER-PLANE affine texel transforms and quad extents (VL3D_COORD_AFFINE_PLANES): the exact
// atlas-cell sampling of the reference (MPV.py:75-81, 394-439) on the dense stack -- shipped activations.  Homography records are
// 16 floats per plane: 3x3 matrix with the plane's texel transform folded in, then the coverage box.
#define VL3D_HS 16
#define VL3D_HN 13
#define VL3D_CONV_FN conv_affine_planes_hardcut_post
#define VL3D_CONV_COORD VL3D_COORD_AFFINE
#define VL3D_CONV_BORDER VL3D_BORDER_HARDCUT
#define VL3D_CONV_ORDER VL3D_ACT_POST
#define VL3D_CONV_ACTS 0
#include "vl3d_render_conv.inc"
