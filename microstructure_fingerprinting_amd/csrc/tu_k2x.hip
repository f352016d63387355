// tu_k2x.hip -- translation unit of the two-fascicle + CSF/EAR kernel (fit_k2x.hip): instantiations and launcher.
#include "mfx_host.h"

#include <algorithm>

static size_t k2x_lds_bytes(int ksteps, bool bracket, int NP, int nw, int nbuf, int ntup, int M, int NX) {
  const size_t MP = (size_t)ksteps * 4;
  size_t dbl = (size_t)nbuf * MP * 16 + MP + 2 * MP + (bracket ? 2 * MP + 2 * MP : 0) + 4 * (size_t)NP + (size_t)nw * 16 * MFX_XS +
               2 * 16 * MFX_XS + MFX_XS + MFX_XS * MFX_XS + 32 + MFX_XS +        // ... s_Qx
               (4 * MFX_XS + 2) + (size_t)nw * 16 + 2 * 16 + 2;                    // s_tc, s_rowf, s_colf, s_thr
  dbl += (size_t)M * NX;                                                           // s_xx
  return dbl * 8 + sizeof(CandX) * MFX_XMAXC + sizeof(FamX) * MFX_XFAM + sizeof(ProjC) * ((size_t)nw * 16 + 2 * 16) * ntup +
         sizeof(int) * (2 * MP + (bracket ? 2 * MP : 0) + 4);
}

template <int KSTEPS, bool BRACKET, int NW = 8, int NBUF = 2>
static int launch_k2x_t(FitK2XArgs a, int nvox, hipStream_t st) {
  MfxThread& T = mfx_thread();
  const int ntup = (a.X.has_csf && a.X.E > 0) ? a.X.E : a.X.NX;   // extra tuples per atom pair (fit_k2x.hip)
  if (a.X.NX > 15) return mfx_fail(MFX_ERR_UNSUPPORTED, "two-fascicle classes support at most 15 CSF+EAR columns (got %d)", a.X.NX);
  if (16 * (ntup + 1) > NW * 64) return mfx_fail(MFX_ERR_UNSUPPORTED, "K=2+extras kernel: %d extra tuples exceed this protocol length's limit of %d", ntup, NW * 4 - 1);
  size_t lds = k2x_lds_bytes(KSTEPS, BRACKET, a.T.ldn, NW, NBUF, ntup, a.P.M, a.X.NX);
  a.xx_in_lds = 1;
  if (lds > 160 * 1024) { lds = k2x_lds_bytes(KSTEPS, BRACKET, a.T.ldn, NW, NBUF, ntup, 0, 0); a.xx_in_lds = 0; }   // extras stay in global memory
  if (lds > 160 * 1024) return mfx_fail(MFX_ERR_UNSUPPORTED, "K=2+extras kernel needs %zu B of LDS: N=%d too large", lds, a.T.N);
  auto kern = mfx_fit_k2x_kernel<KSTEPS, BRACKET, NW, NBUF>;
  HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // per-workgroup scratch slab: launch in chunks so the slab stays modest
  const int chunk = 2048;
  const size_t slab = (size_t)2 * a.T.ldn * (MFX_XS + 2 * (ntup + 1));   // doubles per workgroup: inner products + filter base values
  StreamMem ws(st), cnt(st);
  HIPCHK(ws.alloc(sizeof(double) * slab * std::min(chunk, nvox)));
  HIPCHK(cnt.alloc(4 * sizeof(int)));
  HIPCHK(hipMemsetAsync(cnt.p, 0, 4 * sizeof(int), st));
  a.ws = ws.as<double>();
  a.maxc = T.k2x_maxc;
  a.stamps = T.stamps;
  a.ovf_count = cnt.as<int>();
  if (int rc = mfx_prof_begin(st)) return rc;
  for (int base = 0; base < nvox; base += chunk) {
    a.vox_base = base;
    hipLaunchKernelGGL(kern, dim3(std::min(chunk, nvox - base)), dim3(NW * 64), lds, st, a);
  }
  HIPCHK(hipGetLastError());
  if (int rc = mfx_prof_end(st)) return rc;
  return mfx_fb_accumulate(cnt.as<int>(), 4, st);
}

int mfx_launch_k2x(const FitK2XArgs& a, int nvox, hipStream_t st) {
  const int M = a.P.M;
  const bool br = a.P.any_bracket != 0;
  if (M <= 64) return br ? launch_k2x_t<16, true>(a, nvox, st) : launch_k2x_t<16, false>(a, nvox, st);
  if (M <= 200) return br ? launch_k2x_t<50, true>(a, nvox, st) : launch_k2x_t<50, false>(a, nvox, st);
  if (M <= 400) return br ? launch_k2x_t<100, true, 4, 1>(a, nvox, st) : launch_k2x_t<100, false, 4, 1>(a, nvox, st);
  if (M <= 560) return br ? launch_k2x_t<140, true, 4, 1>(a, nvox, st) : launch_k2x_t<140, false, 4, 1>(a, nvox, st);
  return mfx_fail(MFX_ERR_UNSUPPORTED, "K=2 fused kernels support M <= 560 (got %d)", M);
}
