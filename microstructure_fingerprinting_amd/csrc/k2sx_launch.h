// k2sx_launch.h -- launcher of the screening kernel in its [N, N, 1] form (fit_k2s.hip, XC = true), shared by the
// tu_k2sx_*.hip translation units: kernel launch only; the pipeline around it (exact stage in list mode, hand-backs)
// is tu_k2x.hip's.
#pragma once
#include "mfx_host.h"
#include "fit_k2s.hip"

template <int KS, bool BR, int NB>
static int launch_k2sx_t(const FitK2Args& a, int nvox, hipStream_t st) {
  const size_t lds = mfx_k2sx_lds_bytes(KS, a.T.N, BR, NB);
  auto kern = mfx_fit_k2s_kernel<KS, BR, NB, true>;
  HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(nvox), dim3(512), lds, st, a);
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

// as many chunk images as fit (3, else 2)
#define MFX_K2SX_TU(KS_, NAME_)                                                                                        \
  int NAME_(const FitK2Args& a, int nvox, hipStream_t st, bool br) {                                                  \
    if (mfx_k2sx_lds_bytes(KS_, a.T.N, br, 3) <= 160 * 1024)                                                          \
      return br ? launch_k2sx_t<KS_, true, 3>(a, nvox, st) : launch_k2sx_t<KS_, false, 3>(a, nvox, st);               \
    if (mfx_k2sx_lds_bytes(KS_, a.T.N, br, 2) <= 160 * 1024)                                                          \
      return br ? launch_k2sx_t<KS_, true, 2>(a, nvox, st) : launch_k2sx_t<KS_, false, 2>(a, nvox, st);               \
    return mfx_fail(MFX_ERR_UNSUPPORTED, "screening kernel ([N, N, 1] form): N = %d does not fit the LDS", a.T.N);    \
  }
