// tu_k2s_ks4.hip -- screening kernel instantiations for KS = 4 (k-steps of 16 measurements)
#include "k2s_launch.h"
MFX_K2S_TU(4, mfx_launch_k2s_ks4)
