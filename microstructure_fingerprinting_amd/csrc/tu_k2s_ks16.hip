// tu_k2s_ks16.hip -- screening kernel instantiations for KS = 16 (k-steps of 16 measurements)
#include "k2s_launch.h"
MFX_K2S_TU(16, mfx_launch_k2s_ks16)
