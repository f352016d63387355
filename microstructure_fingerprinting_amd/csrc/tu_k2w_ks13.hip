// tu_k2w_ks13.hip -- wide screening kernel, M <= 208: two row tiles per wave
#include "k2w_launch.h"
MFX_K2W_TU(13, 2, 2, mfx_launch_k2w_ks13)
