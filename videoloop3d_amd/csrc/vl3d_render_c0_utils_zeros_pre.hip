// utils_mpi.py:159-176 + 92-107: sigmoid -> warp_homography (grid_sample zeros padding) -> over-composite; the graded, golden-pinned convention
#define VL3D_CONV_FN conv_utils_zeros_pre
#define VL3D_CONV_COORD VL3D_COORD_UTILS_MPI
#define VL3D_CONV_BORDER VL3D_BORDER_ZEROS
#define VL3D_CONV_ORDER VL3D_ACT_PRE
#define VL3D_CONV_ACTS 1
#include "vl3d_render_conv.inc"
