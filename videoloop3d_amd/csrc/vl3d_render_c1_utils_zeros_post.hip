// cross-check convention (sample-then-activate with zeros padding): keeps the template axes independent in the tests
#define VL3D_CONV_FN conv_utils_zeros_post
#define VL3D_CONV_COORD VL3D_COORD_UTILS_MPI
#define VL3D_CONV_BORDER VL3D_BORDER_ZEROS
#define VL3D_CONV_ORDER VL3D_ACT_POST
#define VL3D_CONV_ACTS 0
#include "vl3d_render_conv.inc"
