// tu_k2.hip -- translation unit of the FP64 two-fascicle kernel (fit_k2.hip): its instantiations and launcher.
#include "mfx_host.h"

#include <cstdlib>

static size_t k2_lds_bytes(int ksteps, bool bracket, int NP, int tiles, int nbuf) {
  const size_t MP = (size_t)ksteps * 4;
  const size_t MPS = (MP + 15) / 16 * 16;
  size_t dbl = (size_t)nbuf * tiles * MPS * 16 + MP + 2 * MP + (bracket ? 2 * MP + 2 * MP : 0) + 6 * (size_t)NP + 32;
  size_t bytes = dbl * 8 + sizeof(Cand) * MFX_MAXC + sizeof(int) * (2 * MP + (bracket ? 2 * MP : 0) + 4);
  return bytes;
}

template <int KSTEPS, bool BRACKET, bool PIPE = true, int NW = 8, int TILES = 2, int NBUF = 2>
static int launch_k2_t(const FitK2Args& a, int nvox, hipStream_t st, bool rec) {
  MfxThread& T = mfx_thread();
  if (T.k2_pipe < 0) { const char* e = getenv("MFX_K2_PIPE"); T.k2_pipe = (e && e[0] == '0') ? 0 : 1; }
  if constexpr (PIPE && !BRACKET) { if (!T.k2_pipe) return launch_k2_t<KSTEPS, BRACKET, false, NW, TILES, NBUF>(a, nvox, st, rec); }
  const size_t lds = k2_lds_bytes(KSTEPS, BRACKET, a.T.ldn, TILES, NBUF);
  if (lds > 160 * 1024) return mfx_fail(MFX_ERR_UNSUPPORTED, "K=2 kernel needs %zu B of LDS (> 160 KiB): N=%d too large", lds, a.T.N);
  auto kern = mfx_fit_k2_kernel<KSTEPS, BRACKET, PIPE && !BRACKET, NW, TILES, NBUF>;
  HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (rec) { if (int rc = mfx_prof_begin(st)) return rc; }
  FitK2Args aa = a;
  aa.stamps = T.stamps;
  aa.maxc = T.k2_maxc;
  hipLaunchKernelGGL(kern, dim3(nvox), dim3(NW * 64), lds, st, aa);
  HIPCHK(hipGetLastError());
  if (rec) { if (int rc = mfx_prof_end(st)) return rc; }
  return MFX_OK;
}

int mfx_launch_k2_f64(const FitK2Args& a, int nvox, hipStream_t st, bool rec) {
  const int M = a.P.M;
  const bool br = a.P.any_bracket != 0;
  if (M <= 64) return br ? launch_k2_t<16, true>(a, nvox, st, rec) : launch_k2_t<16, false>(a, nvox, st, rec);
  if (M <= 200) {
    if (k2_lds_bytes(50, br, a.T.ldn, 2, 2) <= 160 * 1024) return br ? launch_k2_t<50, true>(a, nvox, st, rec) : launch_k2_t<50, false>(a, nvox, st, rec);
    // large dictionaries (N > 960): the single-tile single-buffer form, one wave per SIMD
    return br ? launch_k2_t<50, true, false, 4, 1, 1>(a, nvox, st, rec) : launch_k2_t<50, false, false, 4, 1, 1>(a, nvox, st, rec);
  }
  // long protocols: one wave per SIMD (512 registers hold the A operand), single-tile single-buffer chunks
  if (M <= 400) return br ? launch_k2_t<100, true, false, 4, 1, 1>(a, nvox, st, rec) : launch_k2_t<100, false, false, 4, 1, 1>(a, nvox, st, rec);
  if (M <= 560) return br ? launch_k2_t<140, true, false, 4, 1, 1>(a, nvox, st, rec) : launch_k2_t<140, false, false, 4, 1, 1>(a, nvox, st, rec);
  return mfx_fail(MFX_ERR_UNSUPPORTED, "K=2 fused kernel supports M <= 560 (got %d)", M);
}

// whether mfx_launch_k2_f64 can serve this plan: the screening kernel hands voxels back to it, so it runs only then
bool mfx_k2_f64_fits(const FitK2Args& a) {
  const int M = a.P.M;
  const bool br = a.P.any_bracket != 0;
  if (M > 560) return false;
  const size_t lds = M <= 64 ? k2_lds_bytes(16, br, a.T.ldn, 2, 2) : M <= 200 ? k2_lds_bytes(50, br, a.T.ldn, 1, 1)
                   : M <= 400 ? k2_lds_bytes(100, br, a.T.ldn, 1, 1) : k2_lds_bytes(140, br, a.T.ldn, 1, 1);
  return lds <= 160 * 1024;
}
