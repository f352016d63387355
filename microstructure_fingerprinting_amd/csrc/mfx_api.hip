// mfx_api.hip -- host side of the C ABI declared in include/mfx.h.
// Owns device tables/plans, bins voxels by compartment class, launches the HIP kernels.
// There is deliberately no CPU compute path in this file: without a usable gfx950 device
// every compute entry point returns MFX_ERR_NO_DEVICE.
#include "../../include/mfx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "fit_k2.hip"
#include "fit_k2s.hip"
#include "fit_small.hip"
#include "fit_k2x.hip"
#include "rotate.hip"
#include "solve_generic.hip"
#include "mc_average.hip"
#include "mfx_device.h"

// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static thread_local hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
static thread_local int g_ev_launches = 0;
static thread_local bool g_ev_valid = false;
static bool g_profiling = false;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(x)                                                                              \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) return fail(MFX_ERR_HIP, "%s failed: %s", #x, hipGetErrorString(e_)); \
  } while (0)

extern "C" const char* mfx_last_error(void) { return g_err.c_str(); }
extern "C" int mfx_abi_version(void) { return 1; }
extern "C" int mfx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" void mfx_set_profiling(int enabled) { g_profiling = enabled != 0; }
extern "C" double mfx_last_kernel_ms(void) {
  if (!g_ev_valid || !g_ev0 || !g_ev1 || g_ev_launches == 0) return -1.0;
  if (hipEventSynchronize(g_ev1) != hipSuccess) return -1.0;
  float ms = 0;
  if (hipEventElapsedTime(&ms, g_ev0, g_ev1) != hipSuccess) return -1.0;
  return (double)ms / g_ev_launches;
}

// ---------------------------------------------------------------------------------------------
struct mfx_tables {
  int device = 0;
  TablesDev d{};
  std::vector<double> h_x, h_G, h_Y;   // h_Y [P x N]: kept for the per-plan virtual shells of bracketed rows
  double scr_scale = 1.0;              // power of two baked into the FP32 screening tables (see mfx_tables_create)
  std::vector<int> h_off;
  void* dx = nullptr;
  void* doff = nullptr;
  void* dtab = nullptr;
  void* dtab32 = nullptr;
  void* dG = nullptr;
};

struct mfx_plan {
  const mfx_tables* t = nullptr;
  PlanDev d{};
  void* dg = nullptr;
  void* dslo = nullptr;
  void* dshi = nullptr;
  void* dtG = nullptr;
  void* ddG = nullptr;
  // screening view (only allocated when the protocol has G-bracketed rows)
  void* dtab32s = nullptr;
  void* dxs = nullptr;
  void* doffs = nullptr;
  void* dsscr = nullptr;
};

static int require_device(int device) {
  int n = mfx_device_count();
  if (n <= 0) return fail(MFX_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= n) return fail(MFX_ERR_ARG, "device %d out of range (have %d)", device, n);
  HIPCHK(hipSetDevice(device));
  return MFX_OK;
}

extern "C" int mfx_tables_create(const double* knots_x, const int32_t* shell_off, const double* knots_Y,
                                 const double* G_un, int S, int N, int device, mfx_tables** out) {
  if (!knots_x || !shell_off || !knots_Y || !G_un || !out || S < 1 || N < 1)
    return fail(MFX_ERR_ARG, "mfx_tables_create: null or empty argument");
  if (int rc = require_device(device)) return rc;
  const int P = shell_off[S];
  for (int s = 0; s < S; ++s)
    if (shell_off[s + 1] - shell_off[s] < 2) return fail(MFX_ERR_ARG, "shell %d has fewer than 2 knots", s);
  const int ldn = (N + 15) / 16 * 16;
  std::vector<double2> tab((size_t)(P + 1) * ldn, double2{0.0, 0.0});
  for (int s = 0; s < S; ++s) {
    for (int j = shell_off[s]; j < shell_off[s + 1]; ++j) {
      const bool last = (j == shell_off[s + 1] - 1);
      for (int n = 0; n < N; ++n) {
        double sl = 0.0;
        if (!last) {
          // interp1d._call_linear: slope = (y_hi - y_lo) / (x_hi - x_lo)
          sl = (knots_Y[(size_t)(j + 1) * N + n] - knots_Y[(size_t)j * N + n]) / (knots_x[j + 1] - knots_x[j]);
        }
        tab[(size_t)j * ldn + n] = double2{knots_Y[(size_t)j * N + n], sl};
      }
    }
  }
  mfx_tables* t = new mfx_tables();
  t->device = device;
  t->h_x.assign(knots_x, knots_x + P);
  t->h_G.assign(G_un, G_un + S);
  t->h_Y.assign(knots_Y, knots_Y + (size_t)P * N);
  t->h_off.assign(shell_off, shell_off + S + 1);
  HIPCHK(hipMalloc(&t->dx, sizeof(double) * P));
  HIPCHK(hipMalloc(&t->doff, sizeof(int) * (S + 1)));
  HIPCHK(hipMalloc(&t->dtab, sizeof(double2) * tab.size()));
  HIPCHK(hipMalloc(&t->dG, sizeof(double) * S));
  HIPCHK(hipMemcpy(t->dx, knots_x, sizeof(double) * P, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(t->doff, shell_off, sizeof(int) * (S + 1), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(t->dtab, tab.data(), sizeof(double2) * tab.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(t->dG, G_un, sizeof(double) * S, hipMemcpyHostToDevice));
  {
    // FP32 screening copy, scaled by a power of two so that the largest table value lies in (64, 128]: the screening
    // kernel feeds D2 to the FP16 MFMA WITHOUT normalising it (the column norm is applied to the accumulator), so the
    // raw values must sit comfortably inside the FP16 range whatever units the dictionary uses.  Ranking statistics
    // (1/|d|, d.y/|d|) are invariant under this scale.
    double vmax = 0.0;
    for (size_t q = 0; q < (size_t)P * N; ++q) vmax = std::max(vmax, std::fabs(knots_Y[q]));
    t->scr_scale = (vmax > 0.0 && std::isfinite(vmax)) ? std::exp2(7.0 - std::ceil(std::log2(vmax))) : 1.0;
    std::vector<float2> tab32(tab.size());
    for (size_t q = 0; q < tab.size(); ++q) tab32[q] = float2{(float)(tab[q].x * t->scr_scale), (float)(tab[q].y * t->scr_scale)};
    HIPCHK(hipMalloc(&t->dtab32, sizeof(float2) * tab32.size()));
    HIPCHK(hipMemcpy(t->dtab32, tab32.data(), sizeof(float2) * tab32.size(), hipMemcpyHostToDevice));
  }
  t->d.S = S;
  t->d.N = N;
  t->d.ldn = ldn;
  t->d.P = P;
  t->d.x = (const double*)t->dx;
  t->d.off = (const int*)t->doff;
  t->d.tab = (const double2*)t->dtab;
  t->d.tab32 = (const float2*)t->dtab32;
  t->d.G_un = (const double*)t->dG;
  *out = t;
  return MFX_OK;
}

extern "C" void mfx_tables_destroy(mfx_tables* t) {
  if (!t) return;
  (void)hipSetDevice(t->device);
  (void)hipFree(t->dx);
  (void)hipFree(t->doff);
  (void)hipFree(t->dtab);
  (void)hipFree(t->dtab32);
  (void)hipFree(t->dG);
  delete t;
}
extern "C" int mfx_tables_num_atoms(const mfx_tables* t) { return t ? t->d.N : 0; }

static int plan_upload(const mfx_tables* t, int M, const std::vector<double>& g, const std::vector<int>& slo,
                       const std::vector<int>& shi, const std::vector<double>& tG, const std::vector<double>& dG,
                       mfx_plan** out) {
  mfx_plan* p = new mfx_plan();
  p->t = t;
  HIPCHK(hipSetDevice(t->device));
  HIPCHK(hipMalloc(&p->dg, sizeof(double) * 3 * M));
  HIPCHK(hipMalloc(&p->dslo, sizeof(int) * M));
  HIPCHK(hipMalloc(&p->dshi, sizeof(int) * M));
  HIPCHK(hipMalloc(&p->dtG, sizeof(double) * M));
  HIPCHK(hipMalloc(&p->ddG, sizeof(double) * M));
  HIPCHK(hipMemcpy(p->dg, g.data(), sizeof(double) * 3 * M, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(p->dslo, slo.data(), sizeof(int) * M, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(p->dshi, shi.data(), sizeof(int) * M, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(p->dtG, tG.data(), sizeof(double) * M, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(p->ddG, dG.data(), sizeof(double) * M, hipMemcpyHostToDevice));
  p->d.M = M;
  p->d.g = (const double*)p->dg;
  p->d.s_lo = (const int*)p->dslo;
  p->d.s_hi = (const int*)p->dshi;
  p->d.tG = (const double*)p->dtG;
  p->d.dG = (const double*)p->ddG;
  p->d.any_bracket = 0;
  for (int m = 0; m < M; ++m) p->d.any_bracket |= (shi[m] >= 0);
  // ---- screening view
  p->d.tab32s = t->d.tab32; p->d.xs = nullptr; p->d.offs = nullptr; p->d.s_scr = nullptr;
  if (p->d.any_bracket) {
    const int S = t->d.S, N = t->d.N, ldn = t->d.ldn, P = t->d.P;
    const double* X = t->h_x.data();
    const double* Y = t->h_Y.data();
    const int* off = t->h_off.data();
    // SciPy interp1d._call_linear on shell s at u, atom n (same clipping as mfx_shell_locate)
    auto interp = [&](int s, double u, int n) {
      const int o = off[s], Ps = off[s + 1] - o;
      int j = (int)(std::lower_bound(X + o, X + o + Ps, u) - (X + o));
      j = j < 1 ? 1 : (j > Ps - 1 ? Ps - 1 : j);
      const int r = o + j - 1;
      const double sl = (Y[(size_t)(r + 1) * N + n] - Y[(size_t)r * N + n]) / (X[r + 1] - X[r]);
      return sl * (u - X[r]) + Y[(size_t)r * N + n];
    };
    struct VS { int lo, hi; double w; };
    std::vector<VS> vs;
    std::vector<int> sscr(M);
    std::vector<double> xs(t->h_x);      // knot value per table row
    xs.push_back(0.0);                   // row P (zero row) has no knot
    std::vector<int> offs;               // [2 x n_shells]: first row, knot count
    for (int s = 0; s < S; ++s) { offs.push_back(off[s]); offs.push_back(off[s + 1] - off[s]); }
    std::vector<float2> tabs((size_t)(P + 1) * ldn, float2{0.0f, 0.0f});
    for (int r = 0; r < P; ++r)
      for (int n = 0; n < N; ++n) {
        const int s = (int)(std::upper_bound(off, off + S + 1, r) - off) - 1;
        const bool last = (r == off[s + 1] - 1);
        const double sl = last ? 0.0 : (Y[(size_t)(r + 1) * N + n] - Y[(size_t)r * N + n]) / (X[r + 1] - X[r]);
        tabs[(size_t)r * ldn + n] = float2{(float)(Y[(size_t)r * N + n] * t->scr_scale), (float)(sl * t->scr_scale)};
      }
    for (int m = 0; m < M; ++m) {
      if (shi[m] < 0) { sscr[m] = slo[m]; continue; }
      const double w = tG[m] / dG[m];
      int v = -1;
      for (size_t q = 0; q < vs.size(); ++q)
        if (vs[q].lo == slo[m] && vs[q].hi == shi[m] && vs[q].w == w) { v = (int)q; break; }
      if (v < 0) {
        v = (int)vs.size();
        vs.push_back(VS{slo[m], shi[m], w});
        std::vector<double> xv(X + off[slo[m]], X + off[slo[m] + 1]);
        xv.insert(xv.end(), X + off[shi[m]], X + off[shi[m] + 1]);
        std::sort(xv.begin(), xv.end());
        xv.erase(std::unique(xv.begin(), xv.end()), xv.end());
        const int K = (int)xv.size();
        std::vector<double> yv((size_t)K * N);
        for (int k = 0; k < K; ++k)
          for (int n = 0; n < N; ++n) {
            const double v0 = interp(slo[m], xv[k], n), v1 = interp(shi[m], xv[k], n);
            yv[(size_t)k * N + n] = v0 + (v1 - v0) * w;
          }
        const size_t base = tabs.size() / ldn;   // first row of this virtual shell
        tabs.resize(tabs.size() + (size_t)K * ldn, float2{0.0f, 0.0f});
        for (int k = 0; k < K; ++k)
          for (int n = 0; n < N; ++n) {
            const double sl = (k == K - 1) ? 0.0 : (yv[(size_t)(k + 1) * N + n] - yv[(size_t)k * N + n]) / (xv[k + 1] - xv[k]);
            tabs[(base + k) * ldn + n] = float2{(float)(yv[(size_t)k * N + n] * t->scr_scale), (float)(sl * t->scr_scale)};
          }
        xs.insert(xs.end(), xv.begin(), xv.end());   // xs.size() == base before: knot index == table row
        offs.push_back((int)base);
        offs.push_back(K);
      }
      sscr[m] = t->d.S + v;
    }
    HIPCHK(hipMalloc(&p->dtab32s, sizeof(float2) * tabs.size()));
    HIPCHK(hipMalloc(&p->dxs, sizeof(double) * xs.size()));
    HIPCHK(hipMalloc(&p->doffs, sizeof(int) * offs.size()));
    HIPCHK(hipMalloc(&p->dsscr, sizeof(int) * M));
    HIPCHK(hipMemcpy(p->dtab32s, tabs.data(), sizeof(float2) * tabs.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->dxs, xs.data(), sizeof(double) * xs.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->doffs, offs.data(), sizeof(int) * offs.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->dsscr, sscr.data(), sizeof(int) * M, hipMemcpyHostToDevice));
    p->d.tab32s = (const float2*)p->dtab32s; p->d.xs = (const double*)p->dxs;
    p->d.offs = (const int*)p->doffs; p->d.s_scr = (const int*)p->dsscr;
  }
  *out = p;
  return MFX_OK;
}

extern "C" int mfx_plan_create_multishell(const mfx_tables* t, const double* scheme, int M, mfx_plan** out) {
  if (!t || !scheme || !out || M < 1) return fail(MFX_ERR_ARG, "mfx_plan_create_multishell: bad argument");
  const int S = t->d.S;
  std::vector<double> g(3 * (size_t)M), tG(M, 0.0), dG(M, 1.0);
  std::vector<int> slo(M, 0), shi(M, -1);
  for (int m = 0; m < M; ++m) {
    const double* r = scheme + 7 * (size_t)m;
    g[3 * m] = r[0]; g[3 * m + 1] = r[1]; g[3 * m + 2] = r[2];
    const double G = r[3];
    int sx = -1;
    for (int s = 0; s < S; ++s)
      if (t->h_G[s] == G) { sx = s; break; }  // exact float equality, mf_utils.py:1822
    if (sx >= 0) { slo[m] = sx; continue; }
    int ih = 0;  // np.argmax(Gms_un > Gnew), mf_utils.py:1829
    for (int s = 0; s < S; ++s)
      if (t->h_G[s] > G) { ih = s; break; }
    if (ih == 0)
      return fail(MFX_ERR_G_RANGE,
                  "Gradient intensity %g is not in the [%g, %g] range spanned by the multi-shell sampling. "
                  "Extrapolation not supported.", G, t->h_G[0], t->h_G[S - 1]);
    slo[m] = ih - 1;
    shi[m] = ih;
    tG[m] = G - t->h_G[ih - 1];
    dG[m] = t->h_G[ih] - t->h_G[ih - 1];
  }
  return plan_upload(t, M, g, slo, shi, tG, dG, out);
}

extern "C" int mfx_plan_create_explicit(const mfx_tables* t, const double* gdirs, const int32_t* shell_of_row, int M,
                                        mfx_plan** out) {
  if (!t || !gdirs || !shell_of_row || !out || M < 1) return fail(MFX_ERR_ARG, "mfx_plan_create_explicit: bad argument");
  std::vector<double> g(gdirs, gdirs + 3 * (size_t)M), tG(M, 0.0), dG(M, 1.0);
  std::vector<int> slo(M, 0), shi(M, -1);
  for (int m = 0; m < M; ++m) {
    if (shell_of_row[m] < 0 || shell_of_row[m] >= t->d.S) return fail(MFX_ERR_ARG, "row %d: shell %d out of range", m, shell_of_row[m]);
    slo[m] = shell_of_row[m];
  }
  return plan_upload(t, M, g, slo, shi, tG, dG, out);
}

extern "C" void mfx_plan_destroy(mfx_plan* p) {
  if (!p) return;
  (void)hipSetDevice(p->t->device);
  (void)hipFree(p->dg);
  (void)hipFree(p->dslo);
  (void)hipFree(p->dshi);
  (void)hipFree(p->dtG);
  (void)hipFree(p->ddG);
  (void)hipFree(p->dtab32s); (void)hipFree(p->dxs); (void)hipFree(p->doffs); (void)hipFree(p->dsscr);
  delete p;
}

// ---------------------------------------------------------------------------------------------
// kernel dispatch
static size_t k2_lds_bytes(int ksteps, bool bracket, int NP, int tiles, int nbuf) {
  const size_t MP = (size_t)ksteps * 4;
  const size_t MPS = (MP + 15) / 16 * 16;
  size_t dbl = (size_t)nbuf * tiles * MPS * 16 + MP + 2 * MP + (bracket ? 2 * MP + 2 * MP : 0) + 6 * (size_t)NP + 32;
  size_t bytes = dbl * 8 + sizeof(Cand) * MFX_MAXC + sizeof(int) * (2 * MP + (bracket ? 2 * MP : 0) + 4);
  return bytes;
}

static unsigned long long* g_stamps = nullptr;
extern "C" void mfx_debug_set_stamps(void* dev_ptr) { g_stamps = (unsigned long long*)dev_ptr; }
static int g_k2_pipe = -1;  // MFX_K2_PIPE=0 selects the un-pipelined chunk loop (A/B measurements)
static int g_k2_maxc = MFX_MAXC;
extern "C" void mfx_debug_set_k2_maxc(int maxc) { g_k2_maxc = (maxc < 0 || maxc > MFX_MAXC) ? MFX_MAXC : maxc; }
static int g_k2s_nb = 0;    // 0: as many chunk images as fit; 2: force the two-image schedule
extern "C" void mfx_debug_set_k2s_images(int nb) { g_k2s_nb = (nb == 2) ? 2 : 0; }
static int g_k2s_cap = 0;   // 0: MFX_S_CAP
extern "C" void mfx_debug_set_k2s_cap(int cap) {
  int c = 4;
  while (2 * c <= cap) c *= 2;   // a power of two
  g_k2s_cap = (cap <= 0 || c >= MFX_S_CAP) ? 0 : c;
}

template <int KSTEPS, bool BRACKET, bool PIPE = true, int NW = 8, int TILES = 2, int NBUF = 2>
static int launch_k2_t(const FitK2Args& a, int nvox, hipStream_t st) {
  if (g_k2_pipe < 0) { const char* e = getenv("MFX_K2_PIPE"); g_k2_pipe = (e && e[0] == '0') ? 0 : 1; }
  if constexpr (PIPE && !BRACKET) { if (!g_k2_pipe) return launch_k2_t<KSTEPS, BRACKET, false, NW, TILES, NBUF>(a, nvox, st); }
  const size_t lds = k2_lds_bytes(KSTEPS, BRACKET, a.T.ldn, TILES, NBUF);
  if (lds > 160 * 1024) return fail(MFX_ERR_UNSUPPORTED, "K=2 kernel needs %zu B of LDS (> 160 KiB): N=%d too large", lds, a.T.N);
  auto kern = mfx_fit_k2_kernel<KSTEPS, BRACKET, PIPE && !BRACKET, NW, TILES, NBUF>;
  HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (g_profiling) {
    if (!g_ev0) { HIPCHK(hipEventCreate(&g_ev0)); HIPCHK(hipEventCreate(&g_ev1)); }
    HIPCHK(hipEventRecord(g_ev0, st));
  }
  FitK2Args aa = a;
  aa.stamps = g_stamps;
  aa.maxc = g_k2_maxc;
  hipLaunchKernelGGL(kern, dim3(nvox), dim3(NW * 64), lds, st, aa);
  HIPCHK(hipGetLastError());
  if (g_profiling) {
    HIPCHK(hipEventRecord(g_ev1, st));
    g_ev_launches = 1;
    g_ev_valid = true;
  }
  return MFX_OK;
}

static int launch_k2_f64(const FitK2Args& a, int nvox, hipStream_t st) {
  const int M = a.P.M;
  const bool br = a.P.any_bracket != 0;
  if (M <= 64) return br ? launch_k2_t<16, true>(a, nvox, st) : launch_k2_t<16, false>(a, nvox, st);
  if (M <= 200) {
    if (k2_lds_bytes(50, br, a.T.ldn, 2, 2) <= 160 * 1024) return br ? launch_k2_t<50, true>(a, nvox, st) : launch_k2_t<50, false>(a, nvox, st);
    // large dictionaries (N > 960): the single-tile single-buffer form, one wave per SIMD
    return br ? launch_k2_t<50, true, false, 4, 1, 1>(a, nvox, st) : launch_k2_t<50, false, false, 4, 1, 1>(a, nvox, st);
  }
  // long protocols: one wave per SIMD (512 registers hold the A operand), single-tile single-buffer chunks
  if (M <= 400) return br ? launch_k2_t<100, true, false, 4, 1, 1>(a, nvox, st) : launch_k2_t<100, false, false, 4, 1, 1>(a, nvox, st);
  if (M <= 560) return br ? launch_k2_t<140, true, false, 4, 1, 1>(a, nvox, st) : launch_k2_t<140, false, false, 4, 1, 1>(a, nvox, st);
  return fail(MFX_ERR_UNSUPPORTED, "K=2 fused kernel supports M <= 560 (got %d)", M);
}

// ---- split-FP16 screening kernel (fit_k2s.hip) for exact-G protocols, FP64 kernel for what it hands back
static size_t k2s_lds_bytes(int KS, int N, bool bracket, int NB) {
  const size_t MP = (size_t)KS * 16, NP = ((size_t)N + 31) / 32 * 32;
  return (size_t)2 * NB * KS * 512 * 2 + 8 * (MP + 2 * MP + 32) + sizeof(Cand) * MFX_S_CAP + 16 + 4 * (2 * MP + 4) + 4 * (2 * MP) + 4 * (4 * NP) + 4 * MP + 4 * 8 * 64 +
         (bracket ? 48 * MP : 0) + (KS < 8 ? 4 * MFX_S_CAP : 0);
}
static int g_k2_screen = -1;   // MFX_K2_SCREEN=0 disables the screening kernel (A/B measurements)
static thread_local int g_last_fallback = 0;
extern "C" int mfx_debug_last_fallback_count(void) { return g_last_fallback; }
extern "C" void mfx_debug_set_k2_screen(int enabled) { g_k2_screen = enabled ? 1 : 0; }

template <int KS, bool BR, int NB>
static int launch_k2s_t(const FitK2Args& a, int nvox, hipStream_t st) {
  const size_t lds = k2s_lds_bytes(KS, a.T.N, BR, NB);
  auto kern = mfx_fit_k2s_kernel<KS, BR, NB>;
  HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  struct StreamMem {   // stream-ordered allocation released on every exit path
    void* p = nullptr;
    hipStream_t s;
    explicit StreamMem(hipStream_t s_) : s(s_) {}
    ~StreamMem() { if (p) (void)hipFreeAsync(p, s); }
  } fbm(st);
  HIPCHK(hipMallocAsync(&fbm.p, sizeof(int) * ((size_t)nvox + 1), st));
  int* fb = (int*)fbm.p;   // [0] count, [1..] voxel list
  HIPCHK(hipMemsetAsync(fb, 0, sizeof(int), st));
  if (g_profiling) {
    if (!g_ev0) { HIPCHK(hipEventCreate(&g_ev0)); HIPCHK(hipEventCreate(&g_ev1)); }
    HIPCHK(hipEventRecord(g_ev0, st));
  }
  FitK2Args aa = a;
  aa.stamps = g_stamps;
  aa.fb_count = fb;
  aa.fb_list = fb + 1;
  aa.scap = g_k2s_cap ? g_k2s_cap : MFX_S_CAP;
  hipLaunchKernelGGL(kern, dim3(nvox), dim3(512), lds, st, aa);
  HIPCHK(hipGetLastError());
  if (g_profiling) HIPCHK(hipEventRecord(g_ev1, st));
  int nfb = 0;
  HIPCHK(hipMemcpyAsync(&nfb, fb, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  g_last_fallback = nfb;
  int rc = MFX_OK;
  if (nfb > 0) {           // voxels whose short list overflowed: redo them with the FP64 kernel
    const bool prof = g_profiling;
    g_profiling = false;   // keep the event pair of the screening kernel
    FitK2Args ab = a;
    ab.vox_list = fb + 1;
    rc = launch_k2_f64(ab, nfb, st);
    g_profiling = prof;
  }
  if (g_profiling) { g_ev_launches = 1; g_ev_valid = true; }
  return rc;
}

// whether launch_k2_f64 can serve this plan: the screening kernel hands voxels back to it, so it runs only then
static bool k2_f64_fits(const FitK2Args& a) {
  const int M = a.P.M;
  const bool br = a.P.any_bracket != 0;
  if (M > 560) return false;
  const size_t lds = M <= 64 ? k2_lds_bytes(16, br, a.T.ldn, 2, 2) : M <= 200 ? k2_lds_bytes(50, br, a.T.ldn, 1, 1)
                   : M <= 400 ? k2_lds_bytes(100, br, a.T.ldn, 1, 1) : k2_lds_bytes(140, br, a.T.ldn, 1, 1);
  return lds <= 160 * 1024;
}

static int launch_k2(const FitK2Args& a, int nvox, hipStream_t st) {
  if (g_k2_screen < 0) { const char* e = getenv("MFX_K2_SCREEN"); g_k2_screen = (e && e[0] == '0') ? 0 : 1; }
  const int M = a.P.M;
  const int KSm = M <= 64 ? 4 : (M <= 128 ? 8 : (M <= 208 ? 13 : 16));   // k-steps of 16 measurements
  if (g_k2_screen && M <= 256 && k2_f64_fits(a)) {
    const bool br = a.P.any_bracket != 0;
    // three chunk images (one barrier per chunk) where they fit into the 160 KB of LDS, else two
    const int NB = (g_k2s_nb != 2 && k2s_lds_bytes(KSm, a.T.N, br, 3) <= 160 * 1024) ? 3 : (k2s_lds_bytes(KSm, a.T.N, br, 2) <= 160 * 1024 ? 2 : 0);
#define MFX_K2S_CASE(KS_, BR_, NB_) if (KSm == KS_ && br == BR_ && NB == NB_) return launch_k2s_t<KS_, BR_, NB_>(a, nvox, st)
    MFX_K2S_CASE(4, false, 3);  MFX_K2S_CASE(4, true, 3);
    MFX_K2S_CASE(8, false, 3);  MFX_K2S_CASE(8, true, 3);
    MFX_K2S_CASE(13, false, 3); MFX_K2S_CASE(13, true, 3);
    MFX_K2S_CASE(4, false, 2);  MFX_K2S_CASE(4, true, 2);
    MFX_K2S_CASE(8, false, 2);  MFX_K2S_CASE(8, true, 2);
    MFX_K2S_CASE(13, false, 2); MFX_K2S_CASE(13, true, 2);
    MFX_K2S_CASE(16, false, 2); MFX_K2S_CASE(16, true, 2);
    MFX_K2S_CASE(16, false, 3); MFX_K2S_CASE(16, true, 3);
#undef MFX_K2S_CASE
  }
  return launch_k2_f64(a, nvox, st);
}

// ---- extra (voxel-independent) columns of one voxel class: [csf] + [ear_0..ear_{E-1}]
struct ExtrasHost {
  ExtrasDev d{};
  void* dx = nullptr;
  void* dG = nullptr;
  int build(int M, int has_csf, int E, const double* sig_csf, const double* sig_ear, bool src_on_device) {
    const int NX = has_csf + E;
    d.NX = NX; d.has_csf = has_csf; d.E = E; d.x = nullptr; d.Gxx = nullptr;
    if (NX == 0) return MFX_OK;
    if (NX > MFX_NXMAX) return fail(MFX_ERR_UNSUPPORTED, "at most %d CSF+EAR columns are supported (got %d)", MFX_NXMAX, NX);
    std::vector<double> hc(has_csf ? M : 0), he((size_t)M * E);
    if (has_csf) {
      if (!sig_csf) return fail(MFX_ERR_ARG, "sig_csf missing");
      if (src_on_device) HIPCHK(hipMemcpy(hc.data(), sig_csf, sizeof(double) * M, hipMemcpyDeviceToHost));
      else std::memcpy(hc.data(), sig_csf, sizeof(double) * M);
    }
    if (E) {
      if (!sig_ear) return fail(MFX_ERR_ARG, "sig_ear missing");
      if (src_on_device) HIPCHK(hipMemcpy(he.data(), sig_ear, sizeof(double) * M * E, hipMemcpyDeviceToHost));
      else std::memcpy(he.data(), sig_ear, sizeof(double) * (size_t)M * E);
    }
    std::vector<double> x((size_t)M * NX), G((size_t)NX * NX, 0.0);
    for (int m = 0; m < M; ++m) {
      if (has_csf) x[(size_t)m * NX] = hc[m];
      for (int e = 0; e < E; ++e) x[(size_t)m * NX + has_csf + e] = he[(size_t)m * E + e];
    }
    for (int p = 0; p < NX; ++p)
      for (int q = 0; q < NX; ++q) {
        double acc = 0.0;  // sequential over rows, as the reference's Gram loops (mf_utils.py:311-319, 517-531)
        for (int m = 0; m < M; ++m) acc += x[(size_t)m * NX + p] * x[(size_t)m * NX + q];
        G[(size_t)p * NX + q] = acc;
      }
    HIPCHK(hipMalloc(&dx, sizeof(double) * x.size()));
    HIPCHK(hipMalloc(&dG, sizeof(double) * G.size()));
    HIPCHK(hipMemcpy(dx, x.data(), sizeof(double) * x.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dG, G.data(), sizeof(double) * G.size(), hipMemcpyHostToDevice));
    d.x = (const double*)dx;
    d.Gxx = (const double*)dG;
    return MFX_OK;
  }
  ~ExtrasHost() { (void)hipFree(dx); (void)hipFree(dG); }
};

static int launch_small(const FitSmallArgs& a, int nvox, hipStream_t st) {
  const int M = a.P.M;
  const size_t lds = sizeof(double) * (3 * (size_t)M + MFX_NXMAX + 8 + MFX_SWG + 3 * MFX_SWG) + sizeof(long) * MFX_SWG +
                     sizeof(int) * 2 * (size_t)M;
  if (lds > 160 * 1024) return fail(MFX_ERR_UNSUPPORTED, "too many measurements (%d)", M);
  if (a.P.any_bracket) {
    HIPCHK(hipFuncSetAttribute((const void*)mfx_fit_small_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mfx_fit_small_kernel<true>, dim3(nvox), dim3(MFX_SWG), lds, st, a);
  } else {
    HIPCHK(hipFuncSetAttribute((const void*)mfx_fit_small_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mfx_fit_small_kernel<false>, dim3(nvox), dim3(MFX_SWG), lds, st, a);
  }
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

static size_t k2x_lds_bytes(int ksteps, bool bracket, int NP, int nw, int nbuf, int NX) {
  const size_t MP = (size_t)ksteps * 4;
  size_t dbl = (size_t)nbuf * MP * 16 + MP + 2 * MP + (bracket ? 2 * MP + 2 * MP : 0) + 4 * (size_t)NP + (size_t)nw * 16 * MFX_XS +
               MFX_XS + (size_t)(nw + 2) * 16 * (NX + 1) +
               2 * 16 * MFX_XS + MFX_XS + MFX_XS * MFX_XS + 32;
  return dbl * 8 + sizeof(CandX) * MFX_XMAXC + sizeof(int) * (2 * MP + (bracket ? 2 * MP : 0) + 4);
}

template <int KSTEPS, bool BRACKET, int NW = 8, int NBUF = 2>
static int launch_k2x_t(FitK2XArgs a, int nvox, hipStream_t st) {
  const size_t lds = k2x_lds_bytes(KSTEPS, BRACKET, a.T.ldn, NW, NBUF, a.X.NX);
  if (lds > 160 * 1024) return fail(MFX_ERR_UNSUPPORTED, "K=2+extras kernel needs %zu B of LDS: N=%d too large", lds, a.T.N);
  auto kern = mfx_fit_k2x_kernel<KSTEPS, BRACKET, NW, NBUF>;
  HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // per-workgroup scratch slab: launch in chunks so the slab stays modest
  const int chunk = 2048;
  const size_t slab = (size_t)2 * a.T.ldn * MFX_XS;
  double* ws = nullptr;
  HIPCHK(hipMallocAsync((void**)&ws, sizeof(double) * slab * std::min(chunk, nvox), st));
  a.ws = ws;
  for (int base = 0; base < nvox; base += chunk) {
    a.vox_base = base;
    hipLaunchKernelGGL(kern, dim3(std::min(chunk, nvox - base)), dim3(NW * 64), lds, st, a);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipFreeAsync(ws, st));
  return MFX_OK;
}

static int launch_k2x(const FitK2XArgs& a, int nvox, hipStream_t st) {
  const int M = a.P.M;
  const bool br = a.P.any_bracket != 0;
  if (M <= 64) return br ? launch_k2x_t<16, true>(a, nvox, st) : launch_k2x_t<16, false>(a, nvox, st);
  if (M <= 200) return br ? launch_k2x_t<50, true>(a, nvox, st) : launch_k2x_t<50, false>(a, nvox, st);
  if (M <= 400) return br ? launch_k2x_t<100, true, 4, 1>(a, nvox, st) : launch_k2x_t<100, false, 4, 1>(a, nvox, st);
  if (M <= 560) return br ? launch_k2x_t<140, true, 4, 1>(a, nvox, st) : launch_k2x_t<140, false, 4, 1>(a, nvox, st);
  return fail(MFX_ERR_UNSUPPORTED, "K=2 fused kernels support M <= 560 (got %d)", M);
}

// one homogeneous voxel class (every voxel: K fascicles, has_csf, has_ear); device pointers
static int fit_class_dev(const mfx_plan* p, const double* d_Y, const double* d_peaks, int peaks_ld, const int* d_list,
                         int nvox, int K, int has_csf, int has_ear, const ExtrasHost& X, int maxfasc, int csf_on,
                         int ear_on, double* d_params, hipStream_t st) {
  const int num_params = 1 + 2 * maxfasc + csf_on + 2 * ear_on + 2;
  if (nvox == 0) return MFX_OK;
  if (K + has_csf + has_ear == 0) return MFX_OK;  // mf.py:387-388: rows stay zero
  if (K <= 1) {
    FitSmallArgs a{};
    a.T = p->t->d; a.P = p->d; a.X = X.d;
    a.Y = d_Y; a.peaks = d_peaks; a.peaks_ld = peaks_ld; a.vox_list = d_list;
    a.params = d_params; a.num_params = num_params; a.maxfasc = maxfasc; a.csf_on = csf_on; a.ear_on = ear_on;
    a.K = K;
    return launch_small(a, nvox, st);
  }
  if (K == 2 && !has_csf && !has_ear) {
    FitK2Args a{};
    a.T = p->t->d; a.P = p->d;
    a.Y = d_Y; a.peaks = d_peaks; a.peaks_ld = peaks_ld; a.vox_list = d_list;
    a.params = d_params; a.num_params = num_params; a.maxfasc = maxfasc;
    return launch_k2(a, nvox, st);
  }
  if (K == 2) {
    FitK2XArgs a{};
    a.T = p->t->d; a.P = p->d; a.X = X.d;
    a.Y = d_Y; a.peaks = d_peaks; a.peaks_ld = peaks_ld; a.vox_list = d_list;
    a.params = d_params; a.num_params = num_params; a.maxfasc = maxfasc; a.csf_on = csf_on; a.ear_on = ear_on;
    return launch_k2x(a, nvox, st);
  }
  return fail(MFX_ERR_UNSUPPORTED, "voxel class (K=%d, csf=%d, ear=%d) not implemented", K, has_csf, has_ear);
}

extern "C" int mfx_fit_batch_dev(const mfx_plan* p, const double* d_Y, const double* d_peaks, int maxfasc, int csf_on,
                                 int ear_on, const double* d_sig_csf, const double* d_sig_ear, int E, int64_t V,
                                 double* d_params_out, void* stream) {
  if (!p || !d_Y || !d_params_out || V < 0) return fail(MFX_ERR_ARG, "mfx_fit_batch_dev: bad argument");
  if (maxfasc < 0 || maxfasc > 2) return fail(MFX_ERR_ARG, "maxfasc must be 0..2 (MFModel.MAX_FASC, mf.py:467)");
  if (V == 0) return MFX_OK;
  if (V > 0x7fffffff) return fail(MFX_ERR_ARG, "V too large for one launch");
  if (int rc = require_device(p->t->device)) return rc;
  ExtrasHost X;
  if (int rc = X.build(p->d.M, csf_on ? 1 : 0, ear_on ? E : 0, d_sig_csf, d_sig_ear, true)) return rc;
  return fit_class_dev(p, d_Y, d_peaks, 3 * maxfasc, nullptr, (int)V, maxfasc, csf_on ? 1 : 0, ear_on ? 1 : 0, X, maxfasc,
                       csf_on ? 1 : 0, ear_on ? 1 : 0, d_params_out, (hipStream_t)stream);
}

extern "C" int mfx_fit_batch(const mfx_plan* p, const double* Y, const int32_t* K, const uint8_t* csf,
                             const uint8_t* ear, const double* peaks, int maxfasc, int csf_on, int ear_on,
                             const double* sig_csf, const double* sig_ear, int E, int64_t V, double* params_out) {
  if (!p || !Y || !K || !params_out || V < 0) return fail(MFX_ERR_ARG, "mfx_fit_batch: bad argument");
  if (maxfasc < 0 || maxfasc > 2) return fail(MFX_ERR_ARG, "maxfasc must be 0..2 (MFModel.MAX_FASC, mf.py:467)");
  if (V > 0x7fffffff) return fail(MFX_ERR_ARG, "V too large");
  if (int rc = require_device(p->t->device)) return rc;
  csf_on = csf_on ? 1 : 0;
  ear_on = ear_on ? 1 : 0;
  const int M = p->d.M;
  const int num_params = 1 + 2 * maxfasc + csf_on + 2 * ear_on + 2;
  std::memset(params_out, 0, sizeof(double) * (size_t)V * num_params);
  if (V == 0) return MFX_OK;
  // bin voxels by class (K, csf, ear); direction check once per batch (the reference checks per
  // voxel inside interp_PGSE_from_multishell, mf_utils.py:1798-1802)
  std::vector<int> cls[12];
  for (int64_t v = 0; v < V; ++v) {
    const int k = K[v];
    if (k < 0 || k > maxfasc) return fail(MFX_ERR_ARG, "voxel %lld: numfasc %d outside 0..%d", (long long)v, k, maxfasc);
    const int c = (csf && csf[v]) ? 1 : 0, e = (ear && ear[v]) ? 1 : 0;
    if ((c && !csf_on) || (e && !ear_on)) return fail(MFX_ERR_ARG, "voxel %lld has a CSF/EAR flag but csf_on/ear_on is 0", (long long)v);
    for (int f = 0; f < k; ++f) {
      const double* d = peaks + (size_t)v * 3 * maxfasc + 3 * f;
      const double nrm = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      if (!(std::fabs(1 - nrm) <= 1e-3))
        return fail(MFX_ERR_DIR_NORM, "Orientation vector of the new signal must have unit norm. Detected %g.", nrm);
    }
    cls[k * 4 + c * 2 + e].push_back((int)v);
  }
  double *dY = nullptr, *dpk = nullptr, *dpar = nullptr;
  int* dlist = nullptr;
  HIPCHK(hipMalloc(&dY, sizeof(double) * (size_t)V * M));
  HIPCHK(hipMalloc(&dpk, sizeof(double) * (size_t)V * 3 * std::max(maxfasc, 1)));
  HIPCHK(hipMalloc(&dpar, sizeof(double) * (size_t)V * num_params));
  HIPCHK(hipMalloc(&dlist, sizeof(int) * (size_t)V));
  HIPCHK(hipMemcpy(dY, Y, sizeof(double) * (size_t)V * M, hipMemcpyHostToDevice));
  if (maxfasc > 0) HIPCHK(hipMemcpy(dpk, peaks, sizeof(double) * (size_t)V * 3 * maxfasc, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dpar, 0, sizeof(double) * (size_t)V * num_params));
  int rc = MFX_OK;
  size_t off = 0;
  std::vector<ExtrasHost> xs(4);
  for (int ce = 0; ce < 4 && rc == MFX_OK; ++ce) {
    if (((ce >> 1) && !csf_on) || ((ce & 1) && !ear_on)) continue;  // class cannot occur
    rc = xs[ce].build(M, ce >> 1, (ce & 1) ? E : 0, sig_csf, sig_ear, false);
  }
  for (int c = 0; c < 12 && rc == MFX_OK; ++c) {
    if (cls[c].empty()) continue;
    HIPCHK(hipMemcpy(dlist + off, cls[c].data(), sizeof(int) * cls[c].size(), hipMemcpyHostToDevice));
    rc = fit_class_dev(p, dY, dpk, 3 * maxfasc, dlist + off, (int)cls[c].size(), c >> 2, (c >> 1) & 1, c & 1, xs[c & 3],
                       maxfasc, csf_on, ear_on, dpar, nullptr);
    off += cls[c].size();
  }
  if (rc == MFX_OK) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) rc = fail(MFX_ERR_HIP, "kernel execution failed: %s", hipGetErrorString(e));
  }
  if (rc == MFX_OK) {
    hipError_t e = hipMemcpy(params_out, dpar, sizeof(double) * (size_t)V * num_params, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(MFX_ERR_HIP, "copy back failed: %s", hipGetErrorString(e));
  }
  (void)hipFree(dY); (void)hipFree(dpk); (void)hipFree(dpar); (void)hipFree(dlist);
  return rc;
}

// ---------------------------------------------------------------------------------------------
// rotation
extern "C" int mfx_rotate_dev(const mfx_plan* p, const double* d_dirs, int64_t B, int normalise_dirs, double* d_out,
                              void* stream) {
  if (!p || !d_dirs || !d_out || B < 0) return fail(MFX_ERR_ARG, "mfx_rotate_dev: bad argument");
  if (B == 0) return MFX_OK;
  if (int rc = require_device(p->t->device)) return rc;
  const int M = p->d.M;
  for (int64_t b0 = 0; b0 < B; b0 += 32768) {  // gridDim.y limit
    const int nb = (int)std::min<int64_t>(32768, B - b0);
    dim3 grid((M + MFX_ROT_ROWS - 1) / MFX_ROT_ROWS, nb);
    hipLaunchKernelGGL(mfx_rotate_kernel, grid, dim3(MFX_ROT_WG), 0, (hipStream_t)stream, p->t->d, p->d,
                       d_dirs + 3 * b0, normalise_dirs, d_out + (size_t)b0 * M * p->t->d.N);
  }
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

extern "C" int mfx_rotate(const mfx_plan* p, const double* dirs, int64_t B, int normalise_dirs, double* out) {
  if (!p || !dirs || !out || B < 0) return fail(MFX_ERR_ARG, "mfx_rotate: bad argument");
  if (B == 0) return MFX_OK;
  if (int rc = require_device(p->t->device)) return rc;
  const size_t n_out = (size_t)B * p->d.M * p->t->d.N;
  double *dd = nullptr, *dout = nullptr;
  HIPCHK(hipMalloc(&dd, sizeof(double) * 3 * B));
  HIPCHK(hipMalloc(&dout, sizeof(double) * n_out));
  HIPCHK(hipMemcpy(dd, dirs, sizeof(double) * 3 * B, hipMemcpyHostToDevice));
  int rc = mfx_rotate_dev(p, dd, B, normalise_dirs, dout, nullptr);
  if (rc == MFX_OK) {
    hipError_t e = hipMemcpy(out, dout, sizeof(double) * n_out, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(MFX_ERR_HIP, "mfx_rotate: %s", hipGetErrorString(e));
  }
  (void)hipFree(dd); (void)hipFree(dout);
  return rc;
}

extern "C" int mfx_rotate_cols_dev(const mfx_plan* p, const double* d_dirs, const int32_t* d_cols, int64_t B,
                                   int normalise_dirs, double* d_out, void* stream) {
  if (!p || !d_dirs || !d_cols || !d_out || B < 0) return fail(MFX_ERR_ARG, "mfx_rotate_cols_dev: bad argument");
  if (B == 0) return MFX_OK;
  if (int rc = require_device(p->t->device)) return rc;
  const int64_t total = B * p->d.M;
  const int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffff) return fail(MFX_ERR_ARG, "batch too large");
  hipLaunchKernelGGL(mfx_rotate_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p->t->d, p->d,
                     d_dirs, d_cols, B, normalise_dirs, d_out);
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

extern "C" int mfx_rotate_cols(const mfx_plan* p, const double* dirs, const int32_t* cols, int64_t B,
                               int normalise_dirs, double* out) {
  if (!p || !dirs || !cols || !out || B < 0) return fail(MFX_ERR_ARG, "mfx_rotate_cols: bad argument");
  if (B == 0) return MFX_OK;
  if (int rc = require_device(p->t->device)) return rc;
  for (int64_t b = 0; b < B; ++b)
    if (cols[b] < 0 || cols[b] >= p->t->d.N) return fail(MFX_ERR_ARG, "atom index %d out of range", cols[b]);
  double *dd = nullptr, *dout = nullptr;
  int* dc = nullptr;
  HIPCHK(hipMalloc(&dd, sizeof(double) * 3 * B));
  HIPCHK(hipMalloc(&dc, sizeof(int) * B));
  HIPCHK(hipMalloc(&dout, sizeof(double) * B * p->d.M));
  HIPCHK(hipMemcpy(dd, dirs, sizeof(double) * 3 * B, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dc, cols, sizeof(int) * B, hipMemcpyHostToDevice));
  int rc = mfx_rotate_cols_dev(p, dd, dc, B, normalise_dirs, dout, nullptr);
  if (rc == MFX_OK) {
    hipError_t e = hipMemcpy(out, dout, sizeof(double) * B * p->d.M, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(MFX_ERR_HIP, "mfx_rotate_cols: %s", hipGetErrorString(e));
  }
  (void)hipFree(dd); (void)hipFree(dc); (void)hipFree(dout);
  return rc;
}

// ---------------------------------------------------------------------------------------------
// explicit-dictionary solver
extern "C" int mfx_solve_exhaustive(const double* A, int64_t lda, int M, const int64_t* dicsizes, int Kp, const double* y,
                                    double* w, int64_t* sub, int64_t* tot, double* min_obj, double* y_rec) {
  if (!A || !dicsizes || !y || !w || !sub || !tot || !min_obj || !y_rec || M < 1 || Kp < 1)
    return fail(MFX_ERR_ARG, "mfx_solve_exhaustive: bad argument");
  if (Kp > MFX_GK) return fail(MFX_ERR_UNSUPPORTED, "at most %d sub-dictionaries are supported (got %d)", MFX_GK, Kp);
  if (int rc = require_device(0)) return rc;
  SolveArgs a{};
  long Ntot = 0, ntup = 1;
  for (int k = 0; k < Kp; ++k) {
    if (dicsizes[k] < 1) return fail(MFX_ERR_ARG, "All entries of dicsizes should be > 0");
    a.sizes[k] = dicsizes[k];
    a.start[k] = Ntot;
    Ntot += dicsizes[k];
    if (ntup > (1L << 40) / dicsizes[k]) return fail(MFX_ERR_UNSUPPORTED, "too many index tuples for the explicit solver");
    ntup *= dicsizes[k];
  }
  if (lda < Ntot) return fail(MFX_ERR_ARG, "lda (%lld) < number of columns (%ld)", (long long)lda, Ntot);
  if (Ntot > 30000) return fail(MFX_ERR_UNSUPPORTED, "explicit solver supports up to 30000 columns (got %ld)", Ntot);
  a.M = M; a.Kp = Kp; a.Ntot = (int)Ntot; a.lda = Ntot; a.ntuples = ntup;
  a.nblocks = (int)std::min<long>(8192, (ntup + 255) / 256);
  std::vector<double> Ac((size_t)M * Ntot);
  for (int k = 0; k < M; ++k) std::memcpy(&Ac[(size_t)k * Ntot], A + (size_t)k * lda, sizeof(double) * Ntot);
  double *dA = nullptr, *dy = nullptr, *dG = nullptr, *dAty = nullptr, *dysq = nullptr, *dbs = nullptr, *dw = nullptr,
         *dobj = nullptr, *dyrec = nullptr;
  long *dbt = nullptr, *dsub = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(dA); (void)hipFree(dy); (void)hipFree(dG); (void)hipFree(dAty); (void)hipFree(dysq); (void)hipFree(dbs);
    (void)hipFree(dw); (void)hipFree(dobj); (void)hipFree(dyrec); (void)hipFree(dbt); (void)hipFree(dsub);
  };
#define SCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(MFX_ERR_HIP, "%s failed: %s", #x, hipGetErrorString(e_)); } } while (0)
  SCHK(hipMalloc(&dA, sizeof(double) * Ac.size()));
  SCHK(hipMalloc(&dy, sizeof(double) * M));
  SCHK(hipMalloc(&dG, sizeof(double) * (size_t)Ntot * Ntot));
  SCHK(hipMalloc(&dAty, sizeof(double) * Ntot));
  SCHK(hipMalloc(&dysq, sizeof(double) * 2));
  SCHK(hipMalloc(&dbs, sizeof(double) * a.nblocks));
  SCHK(hipMalloc(&dbt, sizeof(long) * a.nblocks));
  SCHK(hipMalloc(&dw, sizeof(double) * MFX_GK));
  SCHK(hipMalloc(&dsub, sizeof(long) * MFX_GK));
  SCHK(hipMalloc(&dobj, sizeof(double)));
  SCHK(hipMalloc(&dyrec, sizeof(double) * M));
  SCHK(hipMemcpy(dA, Ac.data(), sizeof(double) * Ac.size(), hipMemcpyHostToDevice));
  SCHK(hipMemcpy(dy, y, sizeof(double) * M, hipMemcpyHostToDevice));
  a.A = dA; a.y = dy; a.G = dG; a.Aty = dAty; a.ysq = dysq; a.blk_score = dbs; a.blk_tuple = dbt;
  a.w = dw; a.sub = dsub; a.minobj = dobj; a.yrec = dyrec;
  const long work = Ntot * Ntot + Ntot + 2;
  hipLaunchKernelGGL(mfx_gram_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, nullptr, a);
  hipLaunchKernelGGL(mfx_tuple_scan, dim3(a.nblocks), dim3(256), 0, nullptr, a);
  hipLaunchKernelGGL(mfx_tuple_finalize, dim3(1), dim3(256), 0, nullptr, a);
  SCHK(hipGetLastError());
  SCHK(hipDeviceSynchronize());
  std::vector<long> hsub(MFX_GK);
  SCHK(hipMemcpy(w, dw, sizeof(double) * Kp, hipMemcpyDeviceToHost));
  SCHK(hipMemcpy(hsub.data(), dsub, sizeof(long) * Kp, hipMemcpyDeviceToHost));
  SCHK(hipMemcpy(min_obj, dobj, sizeof(double), hipMemcpyDeviceToHost));
  SCHK(hipMemcpy(y_rec, dyrec, sizeof(double) * M, hipMemcpyDeviceToHost));
#undef SCHK
  for (int k = 0; k < Kp; ++k) { sub[k] = hsub[k]; tot[k] = a.start[k] + hsub[k]; }
  cleanup();
  return MFX_OK;
}

// ---------------------------------------------------------------------------------------------
// Monte-Carlo signal synthesis from spin phases (mf_utils.py:2758-2810)
namespace {
struct DevMem {   // frees on scope exit
  void* p = nullptr;
  ~DevMem() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  template <class T> T* as() const { return (T*)p; }
};
}  // namespace

extern "C" int mfx_monte_carlo_average_dev(const double* d_phases, int64_t n_entries, int64_t spin_stride,
                                           int64_t dim_stride, int dim, const int64_t* delta_mapping,
                                           const double* gscaling, double Dscaling, int64_t num_spins, int64_t n_seq,
                                           double* signal, void* stream) {
  if (n_seq < 0 || dim < 1 || dim > 3 || num_spins < 1 || n_entries < 0)
    return fail(MFX_ERR_ARG, "mfx_monte_carlo_average: need n_seq >= 0, 1 <= dim <= 3, num_spins >= 1");
  if (n_seq == 0) return MFX_OK;
  if (!d_phases || !delta_mapping || !gscaling || !signal) return fail(MFX_ERR_ARG, "mfx_monte_carlo_average: null argument");
  if (n_seq > (1 << 24)) return fail(MFX_ERR_UNSUPPORTED, "mfx_monte_carlo_average: more than 2^24 sequences");
  const int64_t n_ref = n_entries / num_spins;
  for (int64_t i = 0; i < n_seq; ++i)
    if (delta_mapping[i] < 0 || delta_mapping[i] >= n_ref)
      return fail(MFX_ERR_ARG, "delta_mapping[%lld] = %lld outside the %lld simulated acquisitions of the phase table",
                  (long long)i, (long long)delta_mapping[i], (long long)n_ref);
  hipStream_t st = (hipStream_t)stream;
  // group the sequences by simulated acquisition (stable), cut the groups into tiles of MFX_MC_TS
  std::vector<int> order((size_t)n_seq);
  for (int64_t i = 0; i < n_seq; ++i) order[(size_t)i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return delta_mapping[x] < delta_mapping[y]; });
  std::vector<int> t_first, t_cnt;
  std::vector<long> t_start;
  std::vector<double> gs((size_t)n_seq * 3, 0.0);
  for (int64_t q = 0; q < n_seq; ++q) {
    const int i = order[(size_t)q];
    for (int d = 0; d < dim; ++d) gs[(size_t)q * 3 + d] = gscaling[(size_t)i * dim + d];
    const int64_t ref = delta_mapping[i];
    if (q == 0 || delta_mapping[order[(size_t)q - 1]] != ref || t_cnt.back() == MFX_MC_TS) {
      t_first.push_back((int)q); t_cnt.push_back(0); t_start.push_back((long)(ref * num_spins));
    }
    ++t_cnt.back();
  }
  const int64_t nchunks64 = (num_spins + MFX_MC_CHUNK - 1) / MFX_MC_CHUNK;
  const int64_t nblocks = nchunks64 * (int64_t)t_first.size();
  if (nchunks64 > (1 << 30) || nblocks >= (1LL << 31)) return fail(MFX_ERR_UNSUPPORTED, "mfx_monte_carlo_average: launch too large");
  McArgs a{};
  a.ph = d_phases; a.spin_stride = spin_stride; a.dim_stride = dim_stride; a.dim = dim; a.Ds = Dscaling;
  a.num_spins = num_spins; a.n_tiles = (int)t_first.size(); a.nchunks = (int)nchunks64; a.n_seq = (int)n_seq;
  DevMem dfirst, dcnt, dstart, dgs, dord, dpart, dsig;
  HIPCHK(dfirst.alloc(sizeof(int) * t_first.size()));
  HIPCHK(dcnt.alloc(sizeof(int) * t_cnt.size()));
  HIPCHK(dstart.alloc(sizeof(long) * t_start.size()));
  HIPCHK(dgs.alloc(sizeof(double) * gs.size()));
  HIPCHK(dord.alloc(sizeof(int) * order.size()));
  HIPCHK(dpart.alloc(sizeof(double) * (size_t)n_seq * a.nchunks));
  HIPCHK(dsig.alloc(sizeof(double) * (size_t)n_seq));
  HIPCHK(hipMemcpyAsync(dfirst.p, t_first.data(), sizeof(int) * t_first.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dcnt.p, t_cnt.data(), sizeof(int) * t_cnt.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dstart.p, t_start.data(), sizeof(long) * t_start.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dgs.p, gs.data(), sizeof(double) * gs.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dord.p, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice, st));
  a.tile_first = dfirst.as<int>(); a.tile_cnt = dcnt.as<int>(); a.tile_start = dstart.as<long>();
  a.gs = dgs.as<double>(); a.order = dord.as<int>(); a.partial = dpart.as<double>(); a.signal = dsig.as<double>();
  if (g_profiling) {
    if (!g_ev0) { HIPCHK(hipEventCreate(&g_ev0)); HIPCHK(hipEventCreate(&g_ev1)); }
    HIPCHK(hipEventRecord(g_ev0, st));
  }
  hipLaunchKernelGGL(mfx_mc_partial_kernel, dim3((unsigned)nblocks), dim3(MFX_MC_THREADS), 0, st, a);
  HIPCHK(hipGetLastError());
  if (g_profiling) {
    HIPCHK(hipEventRecord(g_ev1, st));
    g_ev_launches = 1;
    g_ev_valid = true;
  }
  hipLaunchKernelGGL(mfx_mc_finalize_kernel, dim3((unsigned)n_seq), dim3(64), 0, st, a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(signal, dsig.p, sizeof(double) * (size_t)n_seq, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));      // temporaries are released on return
  return MFX_OK;
}

extern "C" int mfx_monte_carlo_average(const double* sim_phases, int64_t n_entries, int dim, const int64_t* delta_mapping,
                                       const double* gscaling, double Dscaling, int64_t num_spins, int64_t n_seq,
                                       double* signal, int device) {
  if (n_seq < 0 || dim < 1 || dim > 3 || num_spins < 1 || n_entries < 0)
    return fail(MFX_ERR_ARG, "mfx_monte_carlo_average: need n_seq >= 0, 1 <= dim <= 3, num_spins >= 1");
  if (n_seq == 0) return MFX_OK;
  if (!sim_phases) return fail(MFX_ERR_ARG, "mfx_monte_carlo_average: null argument");
  if (int rc = require_device(device)) return rc;
  DevMem dph;
  HIPCHK(dph.alloc(sizeof(double) * (size_t)n_entries * dim));
  HIPCHK(hipMemcpy(dph.p, sim_phases, sizeof(double) * (size_t)n_entries * dim, hipMemcpyHostToDevice));
  return mfx_monte_carlo_average_dev(dph.as<double>(), n_entries, dim, 1, dim, delta_mapping, gscaling, Dscaling, num_spins,
                                     n_seq, signal, nullptr);
}
