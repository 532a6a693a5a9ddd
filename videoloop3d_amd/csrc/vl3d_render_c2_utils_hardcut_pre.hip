// cross-check convention (activate-then-sample with hard-cut borders)
#define VL3D_CONV_FN conv_utils_hardcut_pre
#define VL3D_CONV_COORD VL3D_COORD_UTILS_MPI
#define VL3D_CONV_BORDER VL3D_BORDER_HARDCUT
#define VL3D_CONV_ORDER VL3D_ACT_PRE
#define VL3D_CONV_ACTS 0
#include "vl3d_render_conv.inc"
