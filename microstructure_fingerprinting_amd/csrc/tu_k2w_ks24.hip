// tu_k2w_ks24.hip -- wide screening kernel, M <= 384: one row tile per wave, one chunk image
#include "k2w_launch.h"
MFX_K2W_TU(24, 1, 1, mfx_launch_k2w_ks24)
