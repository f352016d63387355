// tu_k2s_ks8.hip -- screening kernel instantiations for KS = 8 (k-steps of 16 measurements)
#include "k2s_launch.h"
MFX_K2S_TU(8, mfx_launch_k2s_ks8)
