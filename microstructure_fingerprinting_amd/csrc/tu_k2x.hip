// tu_k2x.hip -- translation unit of the two-fascicle + CSF/EAR kernel (fit_k2x.hip): instantiations and launcher.
#include "mfx_host.h"

#include <algorithm>
#include <cstdlib>

#define MFX_XLCAP 256   // short-list entries per voxel handed from the screening kernel to the exact stage (list mode); longer lists
                         // are cheaper on the plain kernel (with 1 024: 253 k instead of 517 k voxels/s at 200 measurements)

static size_t k2x_lds_bytes(int ksteps, bool bracket, int NP, int nw, int nbuf, int ntup, int M, int NX) {
  const size_t MP = (size_t)ksteps * 4;
  size_t dbl = (size_t)nbuf * MP * 16 + MP + 2 * MP + (bracket ? 2 * MP + 2 * MP : 0) + 4 * (size_t)NP + (size_t)nw * 16 * MFX_XS +
               2 * 16 * MFX_XS + MFX_XS + MFX_XS * MFX_XS + 24 + MFX_XS +        // ... s_Qx
               (4 * MFX_XS + 2) + (size_t)nw * 16 + 2 * 16 + 2;                    // s_tc, s_rowf, s_colf, s_thr
  dbl += (size_t)M * NX;                                                           // s_xx
  return dbl * 8 + sizeof(CandX) * MFX_XMAXC + sizeof(FamX) * MFX_XFAM + sizeof(ProjC) * ((size_t)nw * 16 + 2 * 16) * ntup +
         sizeof(int) * (2 * MP + (bracket ? 2 * MP : 0) + 4 + 2 * nw) + sizeof(QItem) * (size_t)nw * MFX_XQ;
}

// LDS plan of a launch: the extra columns in LDS when the 160 KB allow it, else read from global memory
static int k2x_lds_plan(FitK2XArgs& a, int ksteps, bool bracket, int nw, int nbuf, int ntup, size_t* lds) {
  const size_t cap = 160 * 1024;
  a.xx_in_lds = 1;
  *lds = k2x_lds_bytes(ksteps, bracket, a.T.ldn, nw, nbuf, ntup, a.P.M, a.X.NX);
  if (*lds <= cap) return MFX_OK;
  a.xx_in_lds = 0;
  *lds = k2x_lds_bytes(ksteps, bracket, a.T.ldn, nw, nbuf, ntup, 0, 0);
  if (*lds <= cap) return MFX_OK;
  return mfx_fail(MFX_ERR_UNSUPPORTED, "K=2+extras kernel needs %zu B of LDS: N=%d too large", *lds, a.T.N);
}
static bool k2x_fits(const FitK2XArgs& a, int ksteps, bool bracket, int nw, int nbuf) {
  const int ntup = (a.X.has_csf && a.X.E > 0) ? a.X.E : a.X.NX;
  return k2x_lds_bytes(ksteps, bracket, a.T.ldn, nw, nbuf, ntup, 0, 0) <= 160 * 1024;
}

template <int KSTEPS, bool BRACKET, int NW = 8, int NBUF = 2>
static int launch_k2x_t(FitK2XArgs a, int nvox, hipStream_t st) {
  MfxThread& T = mfx_thread();
  const int ntup = (a.X.has_csf && a.X.E > 0) ? a.X.E : a.X.NX;   // extra tuples per atom pair (fit_k2x.hip)
  if (a.X.NX > 15) return mfx_fail(MFX_ERR_UNSUPPORTED, "two-fascicle classes support at most 15 CSF+EAR columns (got %d)", a.X.NX);
  if (16 * (ntup + 1) > NW * 64) return mfx_fail(MFX_ERR_UNSUPPORTED, "K=2+extras kernel: %d extra tuples exceed this protocol length's limit of %d", ntup, NW * 4 - 1);
  size_t lds = 0;
  if (int rc = k2x_lds_plan(a, KSTEPS, BRACKET, NW, NBUF, ntup, &lds)) return rc;
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

// ---- [N, N, 1] (two fascicles + CSF) with up to 200 measurements: screening pipeline.  Per chunk of voxels:
//   1. the screening kernel in its XC form (fit_k2s.hip; split-FP16 MFMA, relaxed bound with x unconstrained) writes
//      every voxel's short list of atom pairs (or hands the voxel back: ring overflow, an atom inside span(x));
//   2. this file's kernel in LIST mode (statistics, families, exact stage - no pair scan) decides the voxel from the
//      list; a listed pair that beats its own bound by more than the margin hands the voxel back as well;
//   3. after the last chunk the plain kernel redoes the handed-back voxels from the device-side list (no host read).
// MFX_K2X_SCREEN=0 keeps every voxel on the plain kernel.
template <int KSTEPS, bool BRACKET, int NW = 8, int NBUF = 2>
static int launch_k2sx_pipeline(FitK2XArgs a, int nvox, hipStream_t st) {
  MfxThread& T = mfx_thread();
  const int ntup = 1;
  size_t lds = 0;
  if (int rc = k2x_lds_plan(a, KSTEPS, BRACKET, NW, NBUF, ntup, &lds)) return rc;
  auto kern_list = mfx_fit_k2x_kernel<KSTEPS, BRACKET, NW, NBUF, true>;
  auto kern_full = mfx_fit_k2x_kernel<KSTEPS, BRACKET, NW, NBUF, false>;
  HIPCHK(hipFuncSetAttribute((const void*)kern_list, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  HIPCHK(hipFuncSetAttribute((const void*)kern_full, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // the screening and list-mode launches take up to 32 768 voxels at a time (4 KB of list per voxel; the list-mode
  // kernel uses no slab), the plain kernel its usual 2 048 (256 KB of slab per workgroup)
  const int chunk = 2048, big = 32768, cap = MFX_XLCAP;
  const int nc = std::min(chunk, nvox), nb = std::min(big, nvox);
  const size_t slab = (size_t)2 * a.T.ldn * (MFX_XS + 2 * (ntup + 1));
  StreamMem ws(st), cnt(st), fbm(st), xlc(st), xln(st), xlm(st), aud(st);
  HIPCHK(aud.alloc(4 * sizeof(int)));      // population audit of the screening kernel: beyond DC/4, largest error (1e-11), pairs
  HIPCHK(hipMemsetAsync(aud.p, 0, 4 * sizeof(int), st));
  HIPCHK(ws.alloc(sizeof(double) * slab * nc));
  HIPCHK(cnt.alloc(4 * sizeof(int)));
  HIPCHK(fbm.alloc(sizeof(int) * ((size_t)nvox + 4)));   // [0] voxels handed back, [1] of them by the bound check, [4..] their list
  HIPCHK(xlc.alloc(sizeof(Cand) * (size_t)nb * cap));
  HIPCHK(xln.alloc(sizeof(int) * nb));
  HIPCHK(xlm.alloc(sizeof(double) * nb));
  HIPCHK(hipMemsetAsync(cnt.p, 0, 4 * sizeof(int), st));
  HIPCHK(hipMemsetAsync(fbm.p, 0, 4 * sizeof(int), st));
  int* fb = fbm.as<int>();
  a.ws = ws.as<double>();
  a.maxc = T.k2x_maxc;
  a.stamps = T.stamps;
  a.ovf_count = cnt.as<int>();
  a.xl_cand = xlc.as<Cand>(); a.xl_cnt = xln.as<int>(); a.xl_mrg = xlm.as<double>(); a.xl_cap = cap;
  a.fb_count = fb; a.fb_list = fb + 4; a.list_count = nullptr;
  FitK2Args s{};
  s.T = a.T; s.P = a.P; s.Y = a.Y; s.peaks = a.peaks; s.peaks_ld = a.peaks_ld; s.vox_list = a.vox_list; s.list_count = nullptr;
  s.params = a.params; s.num_params = a.num_params; s.maxfasc = a.maxfasc; s.csf_on = a.csf_on; s.ear_on = a.ear_on;
  s.stamps = nullptr; s.fb_count = fb; s.fb_list = fb + 4; s.maxc = 0; s.scap = T.k2s_cap ? T.k2s_cap : MFX_S_CAP;
  s.xc = a.X.x; s.xl_cand = xlc.as<Cand>(); s.xl_cnt = xln.as<int>(); s.xl_mrg = xlm.as<double>(); s.xl_cap = cap;
  s.audit = aud.as<int>();
  if (int rc = mfx_prof_begin(st)) return rc;
  for (int base = 0; base < nvox; base += big) {
    const int n = std::min(big, nvox - base);
    s.vox_base = base;
    const int M = a.P.M;
    if (int rc = (M < 128 ? mfx_launch_k2sx_ks8 : (M <= 200 ? mfx_launch_k2sx_ks13 : (M < 384 ? mfx_launch_k2wx_ks24 : mfx_launch_k2wx_ks35)))(s, n, st, BRACKET)) return rc;
    a.vox_base = base;
    hipLaunchKernelGGL(kern_list, dim3(n), dim3(NW * 64), lds, st, a);
  }
  // the handed-back voxels, plain kernel over the device-side list (blocks beyond its length exit at once)
  FitK2XArgs b = a;
  b.vox_list = fb + 4; b.list_count = fb; b.xl_cand = nullptr; b.xl_cnt = nullptr;
  for (int base = 0; base < nvox; base += chunk) {
    b.vox_base = base;
    hipLaunchKernelGGL(kern_full, dim3(std::min(chunk, nvox - base)), dim3(NW * 64), lds, st, b);
  }
  HIPCHK(hipGetLastError());
  if (int rc = mfx_prof_end(st)) return rc;
  if (int rc = mfx_fb_accumulate(fb, 2, st, 4)) return rc;   // counters [4], [5]: voxels handed to the plain kernel, of them by the bound check
  if (int rc = mfx_fb_accumulate_audit(aud.as<int>(), st)) return rc;
  return mfx_fb_accumulate(cnt.as<int>(), 4, st);
}

int mfx_launch_k2x(const FitK2XArgs& a, int nvox, hipStream_t st) {
  const int M = a.P.M;
  const bool br = a.P.any_bracket != 0;
  {
    MfxThread& T = mfx_thread();
    if (T.k2x_screen < 0) { const char* e = getenv("MFX_K2X_SCREEN"); T.k2x_screen = (e && e[0] == '0') ? 0 : 1; }
    // (the screening kernel needs one free padded row among its 16 KS: M < 128 -> 8 k-steps, M <= 200 -> 13.  Below 64
    // measurements the pipeline does not pay: 254 k against 262 k voxels/s at 45 rows, 27 % of the voxels handed back.)
    const int ksx = M < 128 ? 8 : 13;
    if (T.k2x_screen && T.k2x_maxc == MFX_XMAXC && a.X.has_csf && a.X.E == 0 && a.X.NX == 1 && M >= 64 && M <= 200 &&
        mfx_k2sx_lds_bytes(ksx, a.T.N, br, 2) <= 160 * 1024) {
      if (M == 64) return br ? launch_k2sx_pipeline<16, true>(a, nvox, st) : launch_k2sx_pipeline<16, false>(a, nvox, st);
      return br ? launch_k2sx_pipeline<50, true>(a, nvox, st) : launch_k2sx_pipeline<50, false>(a, nvox, st);
    }
    // longer protocols: the wide screening kernel in its [N, N, 1] form (24 / 35 k-steps; one padded row must stay free)
    if (T.k2x_screen && T.k2x_maxc == MFX_XMAXC && a.X.has_csf && a.X.E == 0 && a.X.NX == 1 && M > 200 && M < 560 &&
        mfx_k2wx_lds_bytes(M < 384 ? 24 : 35, a.T.N, br, 1, 1) <= 160 * 1024) {
      if (M <= 400) return br ? launch_k2sx_pipeline<100, true, 4, 1>(a, nvox, st) : launch_k2sx_pipeline<100, false, 4, 1>(a, nvox, st);
      return br ? launch_k2sx_pipeline<140, true, 4, 1>(a, nvox, st) : launch_k2sx_pipeline<140, false, 4, 1>(a, nvox, st);
    }
  }
  // (a variant whose LDS does not fit - many atoms, many extra columns - hands over to the next: the four-wave variants
  // hold one chunk buffer instead of two; padded measurement rows are zero rows)
  if (M <= 64 && k2x_fits(a, 16, br, 8, 2)) return br ? launch_k2x_t<16, true>(a, nvox, st) : launch_k2x_t<16, false>(a, nvox, st);
  if (M <= 200 && k2x_fits(a, 50, br, 8, 2)) return br ? launch_k2x_t<50, true>(a, nvox, st) : launch_k2x_t<50, false>(a, nvox, st);
  if (M <= 400 && k2x_fits(a, 100, br, 4, 1)) return br ? launch_k2x_t<100, true, 4, 1>(a, nvox, st) : launch_k2x_t<100, false, 4, 1>(a, nvox, st);
  if (M <= 560) return br ? launch_k2x_t<140, true, 4, 1>(a, nvox, st) : launch_k2x_t<140, false, 4, 1>(a, nvox, st);
  return mfx_fail(MFX_ERR_UNSUPPORTED, "K=2 fused kernels support M <= 560 (got %d)", M);
}
