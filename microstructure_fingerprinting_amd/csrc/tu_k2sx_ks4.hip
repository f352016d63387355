// tu_k2sx_ks4.hip -- [N, N, 1] screening kernel, up to 63 measurements
#include "k2sx_launch.h"
MFX_K2SX_TU(4, mfx_launch_k2sx_ks4)
