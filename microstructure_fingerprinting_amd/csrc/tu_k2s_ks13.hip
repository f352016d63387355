// tu_k2s_ks13.hip -- screening kernel instantiations for KS = 13 (k-steps of 16 measurements)
#include "k2s_launch.h"
MFX_K2S_TU(13, mfx_launch_k2s_ks13)
