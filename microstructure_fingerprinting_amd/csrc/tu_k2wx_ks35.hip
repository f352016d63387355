// tu_k2wx_ks35.hip -- wide screening kernel, [N, N, 1] form, 384..559 measurements
#include "k2wx_launch.h"
MFX_K2WX_TU(35, 1, 1, mfx_launch_k2wx_ks35)
