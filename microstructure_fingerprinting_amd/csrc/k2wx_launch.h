// k2wx_launch.h -- launcher of the wide screening kernel in its [N, N, 1] form (fit_k2w.hip, XC = true): kernel launch only;
// the pipeline around it (exact stage in list mode, hand-backs) is tu_k2x.hip's.
#pragma once
#include "mfx_host.h"
#include "fit_k2w.hip"

template <int KS, int TL, bool BR, int NB>
static int launch_k2wx_t(const FitK2Args& a, int nvox, hipStream_t st) {
  const size_t lds = mfx_k2wx_lds_bytes(KS, a.T.N, BR, NB, TL);
  if (lds > 160 * 1024) return mfx_fail(MFX_ERR_UNSUPPORTED, "wide screening kernel ([N, N, 1] form) needs %zu B of LDS: N=%d too large", lds, a.T.N);
  auto kern = mfx_fit_k2w_kernel<KS, TL, BR, NB, true>;
  HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(nvox), dim3(256), lds, st, a);
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

#define MFX_K2WX_TU(KS_, TL_, NB_, NAME_)                                                   \
  int NAME_(const FitK2Args& a, int nvox, hipStream_t st, bool br) {                        \
    return br ? launch_k2wx_t<KS_, TL_, true, NB_>(a, nvox, st) : launch_k2wx_t<KS_, TL_, false, NB_>(a, nvox, st); \
  }
