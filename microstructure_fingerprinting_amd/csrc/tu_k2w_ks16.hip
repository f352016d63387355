// tu_k2w_ks16.hip -- wide screening kernel, M <= 256: two row tiles per wave
#include "k2w_launch.h"
MFX_K2W_TU(16, 2, 2, mfx_launch_k2w_ks16)
