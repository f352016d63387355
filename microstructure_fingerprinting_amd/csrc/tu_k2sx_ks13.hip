// tu_k2sx_ks13.hip -- [N, N, 1] screening kernel, 129..207 measurements (one padded row must stay free)
#include "k2sx_launch.h"
MFX_K2SX_TU(13, mfx_launch_k2sx_ks13)
