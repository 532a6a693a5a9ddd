// MPV.py:351-454 planar convention, the other activation pairs of the product table
#define VL3D_CONV_FN conv_affine_hardcut_post_other
#define VL3D_CONV_COORD VL3D_COORD_AFFINE
#define VL3D_CONV_BORDER VL3D_BORDER_HARDCUT
#define VL3D_CONV_ORDER VL3D_ACT_POST
#define VL3D_CONV_ACTS 2
#include "vl3d_render_conv.inc"
