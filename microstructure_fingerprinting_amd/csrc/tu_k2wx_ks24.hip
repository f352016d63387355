// tu_k2wx_ks24.hip -- wide screening kernel, [N, N, 1] form, 201..383 measurements
#include "k2wx_launch.h"
MFX_K2WX_TU(24, 1, 1, mfx_launch_k2wx_ks24)
