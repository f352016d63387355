// tu_k2sx_ks13.hip -- the screening kernel in its [N, N, 1] form (fit_k2s.hip, XC = true) for protocols of 129..207
// measurements: kernel launch only; the pipeline around it (exact stage in list mode, hand-backs) is tu_k2x.hip's.
#include "mfx_host.h"
#include "fit_k2s.hip"

size_t mfx_k2sx_lds_bytes(int KS, int N, bool bracket, int NB) {
  const size_t MP = (size_t)KS * 16, NP = ((size_t)N + 31) / 32 * 32;
  return mfx_k2s_lds_bytes(KS, N, bracket, NB) + 4 * MP + 8 * NP;
}

template <bool BR, int NB>
static int launch_k2sx13(const FitK2Args& a, int nvox, hipStream_t st) {
  const size_t lds = mfx_k2sx_lds_bytes(13, a.T.N, BR, NB);
  auto kern = mfx_fit_k2s_kernel<13, BR, NB, true>;
  HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(nvox), dim3(512), lds, st, a);
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

// NB = 0: as many chunk images as fit
int mfx_launch_k2sx_ks13(const FitK2Args& a, int nvox, hipStream_t st, bool br) {
  if (mfx_k2sx_lds_bytes(13, a.T.N, br, 3) <= 160 * 1024) return br ? launch_k2sx13<true, 3>(a, nvox, st) : launch_k2sx13<false, 3>(a, nvox, st);
  if (mfx_k2sx_lds_bytes(13, a.T.N, br, 2) <= 160 * 1024) return br ? launch_k2sx13<true, 2>(a, nvox, st) : launch_k2sx13<false, 2>(a, nvox, st);
  return mfx_fail(MFX_ERR_UNSUPPORTED, "screening kernel ([N, N, 1] form): N = %d does not fit the LDS", a.T.N);
}
