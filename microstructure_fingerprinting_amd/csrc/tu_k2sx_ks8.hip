// tu_k2sx_ks8.hip -- [N, N, 1] screening kernel, 64..127 measurements
#include "k2sx_launch.h"
MFX_K2SX_TU(8, mfx_launch_k2sx_ks8)
