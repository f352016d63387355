// mfx_api.hip -- host side of the C ABI declared in include/mfx.h.
// Owns device tables/plans, bins voxels by compartment class, launches the HIP kernels.
// There is deliberately no CPU compute path in this file: without a usable gfx950 device
// every compute entry point returns MFX_ERR_NO_DEVICE.
#include "mfx_host.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "fit_small.hip"
#include "extras.hip"
#include "rotate.hip"
#include "solve_generic.hip"
#include "solve_k3.hip"
#include "fit_k3.hip"
#include "mc_average.hip"
#include "cleanup.hip"
#include "mfx_device.h"

// ---------------------------------------------------------------------------------------------
// per-thread state (mfx_host.h): error string, timing events, diagnostic switches
MfxThread& mfx_thread() {
  static thread_local MfxThread t;
  return t;
}

int mfx_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  mfx_thread().err = buf;
  return code;
}
#define fail mfx_fail

namespace {
struct HostTrace {   // MFX_HOST_TRACE=1: stage times of mfx_fit_batch on stderr (developer diagnostics)
  bool on;
  std::chrono::steady_clock::time_point t0;
  HostTrace() : on(std::getenv("MFX_HOST_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void mark(const char* what) const {
    if (on) std::fprintf(stderr, "[mfx_fit_batch] %8.3f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
  }
};
}  // namespace

// ---- scratch arenas (mfx_host.h)
namespace {
int poison_byte() {
  static const int b = [] { const char* e = std::getenv("MFX_POISON"); return e ? (std::atoi(e) & 0xff) : -1; }();
  return b;
}
constexpr size_t ARENA_ALIGN = 256;
constexpr size_t ARENA_MAX = 8;   // arenas a thread keeps (least recently used one goes)
void arena_drop(MfxThread::Arena& A) {   // (the owner has made sure nothing on the stream still uses the blocks)
  for (void* b : A.blocks) (void)hipFree(b);
  A.blocks.clear(); A.sizes.clear(); A.cur = 0; A.off = 0;
}
}  // namespace
hipError_t mfx_scratch_alloc(void** p, size_t bytes, hipStream_t s) {
  MfxThread& T = mfx_thread();
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  MfxThread::Arena* A = nullptr;
  for (auto& a : T.arenas) if (a.device == dev && a.stream == s) { A = &a; break; }
  if (!A) {
    if (T.arenas.size() >= ARENA_MAX) {   // retire the least recently used idle arena (its stream may be gone: errors ignored)
      size_t lru = T.arenas.size();
      for (size_t q = 0; q < T.arenas.size(); ++q)
        if (T.arenas[q].live == 0 && (lru == T.arenas.size() || T.arenas[q].used < T.arenas[lru].used)) lru = q;
      if (lru == T.arenas.size()) return hipErrorOutOfMemory;
      int cur = 0;
      (void)hipGetDevice(&cur);
      (void)hipSetDevice(T.arenas[lru].device);
      (void)hipDeviceSynchronize();
      arena_drop(T.arenas[lru]);
      (void)hipSetDevice(cur);
      T.arenas.erase(T.arenas.begin() + (long)lru);
    }
    T.arenas.emplace_back();
    A = &T.arenas.back();
    A->device = dev; A->stream = s;
  }
  A->used = ++T.arena_clock;
  bytes = ((bytes ? bytes : 8) + ARENA_ALIGN - 1) / ARENA_ALIGN * ARENA_ALIGN;
  if (A->live == 0) { A->cur = 0; A->off = 0; }   // a new call: the same requests land in the same places, nothing grows
  while (A->cur < A->blocks.size() && A->off + bytes > A->sizes[A->cur]) { ++A->cur; A->off = 0; }
  if (A->cur == A->blocks.size()) {   // a new block (hipMalloc synchronises the device; rare: sizes settle after a call or two)
    const size_t last = A->sizes.empty() ? 0 : A->sizes.back();
    const size_t want = std::max(bytes + bytes / 4, std::min<size_t>(2 * last, (size_t)1 << 30));
    void* nb = nullptr;
    if ((e = hipMalloc(&nb, want)) != hipSuccess) return e;
    A->blocks.push_back(nb); A->sizes.push_back(want);
    A->off = 0;
  }
  *p = (char*)A->blocks[A->cur] + A->off;
  A->off += bytes;
  ++A->live;
  if (poison_byte() >= 0) return hipMemsetAsync(*p, poison_byte(), bytes, s);
  return hipSuccess;
}
void mfx_scratch_free(void* p, hipStream_t s) {
  (void)p;
  MfxThread& T = mfx_thread();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  for (auto& a : T.arenas)
    if (a.stream == s && a.device == dev && a.live > 0) {
      bool mine = false;
      for (size_t q = 0; q < a.blocks.size() && !mine; ++q) mine = (char*)p >= (char*)a.blocks[q] && (char*)p < (char*)a.blocks[q] + a.sizes[q];
      if (mine) { --a.live; return; }
    }
}

int mfx_prof_begin(hipStream_t st) {
  MfxThread& T = mfx_thread();
  if (!T.profiling) return MFX_OK;
  if (!T.ev0) { HIPCHK(hipEventCreate(&T.ev0)); HIPCHK(hipEventCreate(&T.ev1)); }
  T.ev_valid = false;
  HIPCHK(hipEventRecord(T.ev0, st));
  return MFX_OK;
}
int mfx_prof_end(hipStream_t st) {
  MfxThread& T = mfx_thread();
  if (!T.profiling) return MFX_OK;
  HIPCHK(hipEventRecord(T.ev1, st));
  T.ev_launches = 1;
  T.ev_valid = true;
  return MFX_OK;
}
__global__ void mfx_fb_add_kernel(const int* __restrict__ cnt, int n, int* __restrict__ tot) {   // tot: already offset
  if (threadIdx.x < n) atomicAdd(&tot[threadIdx.x], cnt[threadIdx.x]);
}
// audit counters of a screening launch (k2s_shared.h): [0] pairs beyond DC/4 and [2] audited pairs add up, [1] is a maximum
__global__ void mfx_fb_audit_kernel(const int* __restrict__ cnt, int* __restrict__ tot) {
  if (threadIdx.x == 0) { atomicAdd(&tot[0], cnt[0]); atomicMax(&tot[1], cnt[1]); atomicAdd(&tot[2], cnt[2]); }
}
static int fb_setup() {
  MfxThread& T = mfx_thread();
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (T.fb_dev && T.fb_device == dev) return MFX_OK;
  if (T.fb_dev) { (void)hipFree(T.fb_dev); T.fb_dev = nullptr; }
  HIPCHK(hipMalloc((void**)&T.fb_dev, MFX_NCOUNTERS * sizeof(int)));
  T.fb_device = dev;
  if (!T.fb_host) {
    HIPCHK(hipHostMalloc((void**)&T.fb_host, MFX_NCOUNTERS * sizeof(int), hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&T.fb_event, hipEventDisableTiming));
  }
  return MFX_OK;
}
int mfx_fb_begin(hipStream_t st) {
  if (int rc = fb_setup()) return rc;
  HIPCHK(hipMemsetAsync(mfx_thread().fb_dev, 0, MFX_NCOUNTERS * sizeof(int), st));
  return MFX_OK;
}
int mfx_fb_accumulate(const int* d_counters, int n, hipStream_t st, int offset) {
  MfxThread& T = mfx_thread();
  if (!T.fb_dev) return MFX_OK;   // (a launcher used outside mfx_fit_batch*: nothing to report to)
  if (offset < 0 || offset + n > 8) return fail(MFX_ERR_ARG, "counter range");   // ([8..10]: mfx_fb_accumulate_audit)
  hipLaunchKernelGGL(mfx_fb_add_kernel, dim3(1), dim3(64), 0, st, d_counters, n, T.fb_dev + offset);
  HIPCHK(hipGetLastError());
  return MFX_OK;
}
int mfx_fb_accumulate_audit(const int* d_audit, hipStream_t st) {
  MfxThread& T = mfx_thread();
  if (!T.fb_dev) return MFX_OK;
  hipLaunchKernelGGL(mfx_fb_audit_kernel, dim3(1), dim3(64), 0, st, d_audit, T.fb_dev + 8);
  HIPCHK(hipGetLastError());
  return MFX_OK;
}
int mfx_fb_end(hipStream_t st) {
  MfxThread& T = mfx_thread();
  if (!T.fb_dev) return MFX_OK;
  HIPCHK(hipMemcpyAsync(T.fb_host, T.fb_dev, MFX_NCOUNTERS * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(T.fb_event, st));
  T.fb_pending = true;
  return MFX_OK;
}
static int fb_read(int which) {
  MfxThread& T = mfx_thread();
  if (!T.fb_host) return 0;
  if (T.fb_pending) { if (hipEventSynchronize(T.fb_event) != hipSuccess) return -1; T.fb_pending = false; }
  return T.fb_host[which];
}

extern "C" const char* mfx_last_error(void) { return mfx_thread().err.c_str(); }
extern "C" int mfx_abi_version(void) { return 3; }
extern "C" int mfx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" void mfx_set_profiling(int enabled) { mfx_thread().profiling = enabled != 0; }
extern "C" double mfx_last_kernel_ms(void) {
  MfxThread& T = mfx_thread();
  if (!T.ev_valid || !T.ev0 || !T.ev1 || T.ev_launches == 0) return -1.0;
  if (hipEventSynchronize(T.ev1) != hipSuccess) return -1.0;
  float ms = 0;
  if (hipEventElapsedTime(&ms, T.ev0, T.ev1) != hipSuccess) return -1.0;
  return (double)ms / T.ev_launches;
}
extern "C" int mfx_debug_last_fallback_count(void) { return fb_read(0); }
extern "C" int mfx_debug_last_guard_count(void) { return fb_read(1); }
extern "C" int mfx_debug_last_counter(int which) { return (which >= 0 && which < MFX_NCOUNTERS) ? fb_read(which) : -1; }
extern "C" void mfx_debug_set_k2_screen(int enabled) { mfx_thread().k2_screen = enabled ? 1 : 0; }
extern "C" void mfx_debug_set_k2_wide(int mode) { mfx_thread().k2_wide = mode == 1 ? 1 : (mode == 0 ? 0 : 2); }
extern "C" void mfx_debug_set_k2x_screen(int on) { mfx_thread().k2x_screen = on ? 1 : 0; }
extern "C" void mfx_debug_set_stamps(void* dev_ptr) { mfx_thread().stamps = (unsigned long long*)dev_ptr; }
extern "C" void mfx_debug_set_k2_maxc(int maxc) { mfx_thread().k2_maxc = (maxc < 0 || maxc > MFX_MAXC) ? MFX_MAXC : maxc; }
extern "C" void mfx_debug_set_k2x_maxc(int maxc) { mfx_thread().k2x_maxc = (maxc < 0 || maxc > MFX_XMAXC) ? MFX_XMAXC : maxc; }
extern "C" void mfx_debug_set_k3_cap(int cap) { mfx_thread().k3_cap = cap > 0 ? cap : 0; }
extern "C" void mfx_debug_set_force_generic(int enabled) { mfx_thread().force_generic = enabled ? 1 : 0; }
extern "C" void mfx_debug_set_k3_screen(int enabled) { mfx_thread().k3_screen = enabled ? 1 : 0; }
extern "C" void mfx_debug_set_k2s_images(int nb) { mfx_thread().k2s_nb = (nb == 2) ? 2 : 0; }
extern "C" void mfx_debug_set_k2s_cap(int cap) {
  int c = 4;
  while (2 * c <= cap) c *= 2;   // a power of two
  mfx_thread().k2s_cap = (cap <= 0 || c >= MFX_S_CAP) ? 0 : c;
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
  ~mfx_tables() {   // also runs when mfx_tables_create gives up half way
    (void)hipFree(dx); (void)hipFree(doff); (void)hipFree(dtab); (void)hipFree(dtab32); (void)hipFree(dG);
  }
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
  void* dstatus = nullptr;   // int[4]: [0] flag word the fit kernels OR into (MFX_ST_*), read by mfx_plan_status
  ~mfx_plan() {
    (void)hipFree(dg); (void)hipFree(dslo); (void)hipFree(dshi); (void)hipFree(dtG); (void)hipFree(ddG);
    (void)hipFree(dtab32s); (void)hipFree(dxs); (void)hipFree(doffs); (void)hipFree(dsscr); (void)hipFree(dstatus);
  }
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
  std::unique_ptr<mfx_tables> t(new mfx_tables());
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
  *out = t.release();
  return MFX_OK;
}

extern "C" void mfx_tables_destroy(mfx_tables* t) {
  if (!t) return;
  (void)hipSetDevice(t->device);
  delete t;
}
extern "C" int mfx_tables_num_atoms(const mfx_tables* t) { return t ? t->d.N : 0; }

static int plan_upload(const mfx_tables* t, int M, const std::vector<double>& g, const std::vector<int>& slo,
                       const std::vector<int>& shi, const std::vector<double>& tG, const std::vector<double>& dG,
                       int normalise, mfx_plan** out) {
  std::unique_ptr<mfx_plan> p(new mfx_plan());
  p->t = t;
  HIPCHK(hipSetDevice(t->device));
  HIPCHK(hipMalloc(&p->dstatus, 4 * sizeof(int)));
  HIPCHK(hipMemset(p->dstatus, 0, 4 * sizeof(int)));
  p->d.status = (int*)p->dstatus;
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
  p->d.normalise = normalise;
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
  *out = p.release();
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
  return plan_upload(t, M, g, slo, shi, tG, dG, 0, out);
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
  // rotate_atom semantics: the fit kernels divide every fascicle direction by its norm first (mf_utils.py:1262-1270)
  return plan_upload(t, M, g, slo, shi, tG, dG, 1, out);
}

extern "C" void mfx_plan_destroy(mfx_plan* p) {
  if (!p) return;
  (void)hipSetDevice(p->t->device);
  delete p;
}

// Deferred error channel of the asynchronous "_dev" entry points: waits for `stream`, then reports (and clears) what
// the fit kernels flagged since the last call.  MFX_ERR_DIR_NORM: a voxel's fascicle direction failed the reference's
// unit-norm check (mf_utils.py:1798-1802; the reference raises ValueError from inside the voxel loop).
extern "C" int mfx_plan_status(const mfx_plan* p, void* stream) {
  if (!p) return fail(MFX_ERR_ARG, "mfx_plan_status: null plan");
  if (int rc = require_device(p->t->device)) return rc;
  int st[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(st, p->dstatus, sizeof(st), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  if (st[0] == 0) return MFX_OK;
  HIPCHK(hipMemsetAsync(p->dstatus, 0, sizeof(st), (hipStream_t)stream));
  if (st[0] & MFX_ST_DIR_NORM)
    return fail(MFX_ERR_DIR_NORM, "Orientation vector of the new signal must have unit norm (voxel %d of the batch).", st[1]);
  return fail(MFX_ERR_HIP, "fit kernels reported status 0x%x", st[0]);
}

// ---------------------------------------------------------------------------------------------
// kernel dispatch (the launchers of the K=2 kernel families live in their own translation units: mfx_host.h)
size_t mfx_k2s_lds_bytes(int KS, int N, bool bracket, int NB) {
  const size_t MP = (size_t)KS * 16, NP = ((size_t)N + 31) / 32 * 32;
  return (size_t)2 * NB * KS * 512 * 2 + 8 * (MP + 2 * MP + 32) + sizeof(Cand) * MFX_S_CAP + 16 + 4 * (2 * MP + 4) + 4 * (2 * MP) + 4 * (4 * NP) + 4 * MP + 4 * 8 * 64 +
         (bracket ? 48 * MP : 0) + (KS < 8 ? 4 * MFX_S_CAP : 0);
}

size_t mfx_k2sx_lds_bytes(int KS, int N, bool bracket, int NB) {   // + x^ [MP] and u [2][NP] (FP32)
  const size_t MP = (size_t)KS * 16, NP = ((size_t)N + 31) / 32 * 32;
  return mfx_k2s_lds_bytes(KS, N, bracket, NB) + 4 * MP + 8 * NP;
}

size_t mfx_k2w_lds_bytes(int KS, int N, bool bracket, int NB, int TL) {
  const size_t MP = (size_t)KS * 16, NP = ((size_t)N + 31) / 32 * 32;
  return (size_t)2 * NB * KS * 512 * 2 + 8 * (MP + 2 * MP + 32) + sizeof(Cand) * MFX_S_CAP + 16 + 4 * (2 * MP + 4) + 4 * (2 * MP) + 4 * (4 * NP) + 4 * MP +
         4 * 4 * TL * 64 + (bracket ? 48 * MP : 0) + (KS < 8 ? 4 * MFX_S_CAP : 0);
}

size_t mfx_k2wx_lds_bytes(int KS, int N, bool bracket, int NB, int TL) {   // + x [MP] and u [2][NP] (FP32)
  const size_t MP = (size_t)KS * 16, NP = ((size_t)N + 31) / 32 * 32;
  return mfx_k2w_lds_bytes(KS, N, bracket, NB, TL) + 4 * MP + 8 * NP;
}

// Which kernel serves a two-fascicle class:
//   M <= 256: the two-waves-per-SIMD screening kernel (fit_k2s.hip), or - MFX_K2_WIDE=1 / mfx_debug_set_k2_wide(1) - the
//             wide one (fit_k2w.hip, KS = 13 / 16);
//   256 < M <= 560: the wide screening kernel (KS = 24 / 35; the long A tile needs its one-wave-per-SIMD register file);
//   otherwise, with MFX_K2_SCREEN=0, or when the images do not fit the LDS: the FP64 kernel (fit_k2.hip).
static int launch_k2(const FitK2Args& a, int nvox, hipStream_t st) {
  MfxThread& T = mfx_thread();
  if (T.k2_screen < 0) { const char* e = getenv("MFX_K2_SCREEN"); T.k2_screen = (e && e[0] == '0') ? 0 : 1; }
  if (T.k2_wide < 0) { const char* e = getenv("MFX_K2_WIDE"); T.k2_wide = e ? (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : 2)) : 2; }   // 2: automatic
  const int M = a.P.M;
  const bool br = a.P.any_bracket != 0;
  if (T.k2_screen && mfx_k2_f64_fits(a)) {
    const bool wide_ok = T.k2_wide != 0;
    if (M > 256 && M <= 560 && wide_ok) {
      if (M <= 384 && mfx_k2w_lds_bytes(24, a.T.N, br, 1, 1) <= 160 * 1024) return mfx_launch_k2w_ks24(a, nvox, st, br);
      if (mfx_k2w_lds_bytes(35, a.T.N, br, 1, 1) <= 160 * 1024) return mfx_launch_k2w_ks35(a, nvox, st, br);
    }
    if (M <= 256) {
      if (T.k2_wide == 1 && M > 128) {
        if (M <= 208 && mfx_k2w_lds_bytes(13, a.T.N, br, 2, 2) <= 160 * 1024) return mfx_launch_k2w_ks13(a, nvox, st, br);
        if (mfx_k2w_lds_bytes(16, a.T.N, br, 2, 2) <= 160 * 1024) return mfx_launch_k2w_ks16(a, nvox, st, br);
      }
      const int KSm = M <= 64 ? 4 : (M <= 128 ? 8 : (M <= 208 ? 13 : 16));   // k-steps of 16 measurements
      // three chunk images (one barrier per chunk) where they fit into the 160 KB of LDS, else two
      const int NB = (T.k2s_nb != 2 && mfx_k2s_lds_bytes(KSm, a.T.N, br, 3) <= 160 * 1024) ? 3 : (mfx_k2s_lds_bytes(KSm, a.T.N, br, 2) <= 160 * 1024 ? 2 : 0);
      if (NB) {
        if (KSm == 4) return mfx_launch_k2s_ks4(a, nvox, st, br, NB);
        if (KSm == 8) return mfx_launch_k2s_ks8(a, nvox, st, br, NB);
        if (KSm == 13) return mfx_launch_k2s_ks13(a, nvox, st, br, NB);
        return mfx_launch_k2s_ks16(a, nvox, st, br, NB);
      }
    }
  }
  if (int rc = mfx_launch_k2_f64(a, nvox, st, true)) return rc;
  return MFX_OK;
}

// ---- extra (voxel-independent) columns of one voxel class: [csf] + [ear_0..ear_{E-1}], built on the device in
// stream order (mfx_extras_kernel) from DEVICE copies of sig_csf / sig_ear; released in stream order too
struct ExtrasHost {
  ExtrasDev d{};
  void* dx = nullptr;
  void* dG = nullptr;
  hipStream_t st = nullptr;
  ExtrasHost() = default;
  ExtrasHost(const ExtrasHost&) = delete;
  ExtrasHost& operator=(const ExtrasHost&) = delete;
  int build(int M, int has_csf, int E, const double* d_sig_csf, const double* d_sig_ear, hipStream_t stream) {
    const int NX = has_csf + E;
    st = stream;
    d.NX = NX; d.has_csf = has_csf; d.E = E; d.x = nullptr; d.Gxx = nullptr;
    if (NX == 0) return MFX_OK;
    if (has_csf && !d_sig_csf) return fail(MFX_ERR_ARG, "sig_csf missing");
    if (E && !d_sig_ear) return fail(MFX_ERR_ARG, "sig_ear missing");
    HIPCHK(mfx_scratch_alloc(&dx, sizeof(double) * (size_t)M * NX, st));
    HIPCHK(mfx_scratch_alloc(&dG, sizeof(double) * (size_t)NX * NX, st));
    hipLaunchKernelGGL(mfx_extras_kernel, dim3(1), dim3(256), 0, st, d_sig_csf, d_sig_ear, M, has_csf, E, (double*)dx, (double*)dG);
    HIPCHK(hipGetLastError());
    d.x = (const double*)dx;
    d.Gxx = (const double*)dG;
    return MFX_OK;
  }
  ~ExtrasHost() {
    if (dx) mfx_scratch_free(dx, st);
    if (dG) mfx_scratch_free(dG, st);
  }
};

static int launch_small(const FitSmallArgs& a, int nvox, hipStream_t st) {
  const int M = a.P.M;
  const size_t lds = sizeof(double) * (3 * (size_t)M + MFX_NXMAX + 8 + MFX_SWG + 3 * MFX_SWG + (size_t)a.T.N + MFX_SATOMS * ((size_t)M + 32) + (size_t)M * a.X.NX) + sizeof(long) * MFX_SWG +
                     sizeof(int) * (2 * (size_t)M + MFX_SLIST + 1);
  if (a.X.NX > MFX_NXMAX) return fail(MFX_ERR_UNSUPPORTED, "one-fascicle kernel: at most %d CSF+EAR columns (got %d)", MFX_NXMAX, a.X.NX);
  if (lds > 160 * 1024) return fail(MFX_ERR_UNSUPPORTED, "too many measurements (%d) or atoms (%d) for the one-fascicle kernel", M, a.T.N);
  if (int rc = mfx_prof_begin(st)) return rc;
  if (a.P.any_bracket) {
    HIPCHK(hipFuncSetAttribute((const void*)mfx_fit_small_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mfx_fit_small_kernel<true>, dim3(nvox), dim3(MFX_SWG), lds, st, a);
  } else {
    HIPCHK(hipFuncSetAttribute((const void*)mfx_fit_small_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mfx_fit_small_kernel<false>, dim3(nvox), dim3(MFX_SWG), lds, st, a);
  }
  HIPCHK(hipGetLastError());
  return mfx_prof_end(st);
}

// one homogeneous voxel class (every voxel: K fascicles, has_csf, has_ear); device pointers
static int fit_class_generic(const mfx_plan* p, const double* d_Y, const double* d_peaks, int peaks_ld, const int* h_list,
                             int nvox, int K, const ExtrasHost& X, int maxfasc, int csf_on, int ear_on, double* d_params,
                             hipStream_t st);
static bool k3b_applies(int K, int NX, int N, long ntuples);
static int fit_k3_batched(const mfx_plan* p, const double* d_Y, const double* d_peaks, int peaks_ld, const int* h_list, int nvox,
                          int maxfasc, int csf_on, int ear_on, double* d_params, hipStream_t st);

// h_list: host copy of d_list (null with d_list == null: the identity)
static int fit_class_fused(const mfx_plan* p, const double* d_Y, const double* d_peaks, int peaks_ld, const int* d_list,
                           const int* h_list, int nvox, int K, int has_csf, int has_ear, const ExtrasHost& X, int maxfasc,
                           int csf_on, int ear_on, double* d_params, hipStream_t st);
static int fit_class_dev(const mfx_plan* p, const double* d_Y, const double* d_peaks, int peaks_ld, const int* d_list,
                         const int* h_list, int nvox, int K, int has_csf, int has_ear, const ExtrasHost& X, int maxfasc,
                         int csf_on, int ear_on, double* d_params, hipStream_t st) {
  if (nvox == 0) return MFX_OK;
  if (K + has_csf + has_ear == 0) return MFX_OK;  // mf.py:387-388: rows stay zero
  MfxThread& T = mfx_thread();
  int rc = T.force_generic ? MFX_ERR_UNSUPPORTED
                           : fit_class_fused(p, d_Y, d_peaks, peaks_ld, d_list, h_list, nvox, K, has_csf, has_ear, X, maxfasc, csf_on, ear_on, d_params, st);
  // a shape outside the fused kernels' limits (dictionary or protocol too large for the LDS, more CSF+EAR columns than
  // their register arrays hold; they refuse before anything is enqueued): voxel by voxel through the explicit-dictionary
  // solver, which has no such limits - slower by orders of magnitude, same results
  if (rc == MFX_ERR_UNSUPPORTED && K <= 2) {
    if (!h_list && d_list) return fail(MFX_ERR_ARG, "internal: class list without its host copy");
    rc = fit_class_generic(p, d_Y, d_peaks, peaks_ld, h_list, nvox, K, X, maxfasc, csf_on, ear_on, d_params, st);
  }
  return rc;
}

static int fit_class_fused(const mfx_plan* p, const double* d_Y, const double* d_peaks, int peaks_ld, const int* d_list,
                           const int* h_list, int nvox, int K, int has_csf, int has_ear, const ExtrasHost& X, int maxfasc,
                           int csf_on, int ear_on, double* d_params, hipStream_t st) {
  const int num_params = 1 + 2 * maxfasc + csf_on + 2 * ear_on + 2;
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
    return mfx_launch_k2x(a, nvox, st);
  }
  if (K == 3) {
    MfxThread& T = mfx_thread();
    if (T.k3_batch < 0) { const char* e = getenv("MFX_K3_BATCH"); T.k3_batch = (e && e[0] == '0') ? 0 : 1; }
    if (T.k3_batch && T.k3_screen && k3b_applies(K, X.d.NX, p->t->d.N, (long)p->t->d.N * p->t->d.N * p->t->d.N))
      return fit_k3_batched(p, d_Y, d_peaks, peaks_ld, h_list, nvox, maxfasc, csf_on, ear_on, d_params, st);
    return fit_class_generic(p, d_Y, d_peaks, peaks_ld, h_list, nvox, K, X, maxfasc, csf_on, ear_on, d_params, st);
  }
  return fail(MFX_ERR_UNSUPPORTED, "voxel class (K=%d, csf=%d, ear=%d) not implemented", K, has_csf, has_ear);
}

// Launch sequence of the explicit-dictionary solver on device buffers.  Three sub-dictionaries with many triples go
// through solve_k3.hip (Gram on FP64 MFMA, relaxed-bound screen, candidate list); k3 = its extra buffers (thr, ncand,
// nblocks_dev), or null for the plain one-thread-per-tuple scan.
struct K3Bufs { unsigned long long* thr; int* ncand; int* nblocks_dev; double2* st3; };
static size_t k3_buf_bytes(long N3) { return 64 + sizeof(double2) * (size_t)N3; }
static K3Bufs k3_bufs(char* p) { return K3Bufs{(unsigned long long*)p, (int*)(p + 16), (int*)(p + 32), (double2*)(p + 64)}; }
static bool k3_applies(const SolveArgs& a) { return a.Kp == 3 && a.ntuples >= (1L << 18) && a.sizes[0] >= 16 && a.sizes[1] >= 16 && a.sizes[2] >= 16; }
static int launch_solver(SolveArgs a, const K3Bufs* k3, hipStream_t st) {
  if (k3) {
    K3Args k{};
    a.nblocks_dev = k3->nblocks_dev; a.scan_enable = k3->ncand + 1; a.gram_ranking_only = 1;
    k.s = a; k.thr = k3->thr; k.ncand = k3->ncand; k.st3 = k3->st3; k.cand_score = a.blk_score; k.cand_tuple = a.blk_tuple;
    HIPCHK(hipMemsetAsync(k3->thr, 0, sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(k3->ncand, 0, 2 * sizeof(int), st));
    const int nt = (a.Ntot + 63) / 64;
    hipLaunchKernelGGL(mfx_k3_gram_kernel, dim3(nt, nt), dim3(256), 0, st, a);
    hipLaunchKernelGGL(mfx_k3_aty_kernel, dim3((a.Ntot + 2 + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(mfx_k3_st3_kernel, dim3((unsigned)((a.sizes[2] + 255) / 256)), dim3(256), 0, st, k);
    hipLaunchKernelGGL(mfx_k3_pairs_kernel, dim3(2048), dim3(256), 0, st, k);
    hipLaunchKernelGGL(mfx_k3_screen_kernel, dim3((unsigned)((a.sizes[1] + 31) / 32), (unsigned)((a.sizes[0] + 31) / 32)), dim3(256), 0, st, k);
    hipLaunchKernelGGL(mfx_k3_publish_kernel, dim3(1), dim3(64), 0, st, k, k3->nblocks_dev);
    hipLaunchKernelGGL(mfx_tuple_scan, dim3(a.nblocks), dim3(256), 0, st, a);   // exits at once unless the list overflowed
    hipLaunchKernelGGL(mfx_tuple_finalize, dim3(1), dim3(256), 0, st, a);
  } else {
    const long work = (long)a.Ntot * a.Ntot + a.Ntot + 2;
    hipLaunchKernelGGL(mfx_gram_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(mfx_tuple_scan, dim3(a.nblocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(mfx_tuple_finalize, dim3(1), dim3(256), 0, st, a);
  }
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

// internal streams of the calling thread (forked from and joined to the caller's stream by events): two voxels in flight
// on the voxel-by-voxel path, the gated fallback sequences of the batched three-fascicle path
static int lanes_setup(int device) {
  MfxThread& T = mfx_thread();
  if (T.lane_device != device) {
    for (int l = 0; l < 4; ++l) { if (T.s_lane[l]) (void)hipStreamDestroy(T.s_lane[l]); T.s_lane[l] = nullptr; }
    for (int l = 0; l < 5; ++l) { if (T.ev_lane[l]) (void)hipEventDestroy(T.ev_lane[l]); T.ev_lane[l] = nullptr; }
    for (int l = 0; l < 4; ++l) HIPCHK(hipStreamCreateWithFlags(&T.s_lane[l], hipStreamNonBlocking));
    for (int l = 0; l < 5; ++l) HIPCHK(hipEventCreateWithFlags(&T.ev_lane[l], hipEventDisableTiming));
    T.lane_device = device;
  }
  return MFX_OK;
}

// Three fascicles without extra columns (BASELINE config 5) in batches of voxels: fit_k3.hip.  Everything is enqueued on `st`.
#define MFX_K3B_BATCH 32
static bool k3b_applies(int K, int NX, int N, long ntuples) { return K == 3 && NX == 0 && N >= 32 && ntuples >= (1L << 18); }
static int fit_k3_batched(const mfx_plan* p, const double* d_Y, const double* d_peaks, int peaks_ld, const int* h_list, int nvox,
                          int maxfasc, int csf_on, int ear_on, double* d_params, hipStream_t st) {
  const int M = p->d.M, N = p->t->d.N, LD = 3 * N;
  const int num_params = 1 + 2 * maxfasc + csf_on + 2 * ear_on + 2;
  const int BT = std::min(nvox, MFX_K3B_BATCH);
  const size_t nn = (size_t)N * N;
  StreamMem dA(st), dG(st), dcol(st), dst3(st), dsm(st), dcs(st), dct(st), dpart(st), dout(st), dvox(st);
  HIPCHK(dA.alloc(sizeof(double) * (size_t)BT * M * LD));
  HIPCHK(dG.alloc(sizeof(double) * (size_t)BT * 3 * nn));
  HIPCHK(dcol.alloc(sizeof(double) * ((size_t)BT * LD * 2 + (size_t)BT * 2)));
  HIPCHK(dst3.alloc(sizeof(double2) * (size_t)BT * N));
  const int nblk3 = (N + MFX_K3M_KB - 1) / MFX_K3M_KB;
  StreamMem ditm(st);
  HIPCHK(ditm.alloc(sizeof(K3Item) * (size_t)BT * 2 * nblk3 * N * MFX_K3M_KB));
  HIPCHK(dsm.alloc(sizeof(unsigned long long) * (size_t)BT * 8 + sizeof(int) * (size_t)BT * 2));
  const bool k3dbg = getenv("MFX_K3_DEBUG") != nullptr;
  HIPCHK(dcs.alloc(sizeof(double) * (size_t)BT * MFX_K3B_CAP));
  HIPCHK(dct.alloc(sizeof(long) * (size_t)BT * MFX_K3B_CAP));
  HIPCHK(dpart.alloc(sizeof(double) * (size_t)BT * MFX_K3B_FW * 8));
  HIPCHK(dout.alloc(sizeof(double) * ((size_t)BT * 8 + (size_t)BT * 8 + BT + (size_t)BT * M)));
  HIPCHK(dvox.alloc(sizeof(int) * (size_t)nvox));
  {   // voxel of every slot of the class list (the identity without a list)
    std::vector<int> hv(nvox);
    for (int q = 0; q < nvox; ++q) hv[q] = h_list ? h_list[q] : q;
    HIPCHK(hipMemcpyAsync(dvox.p, hv.data(), sizeof(int) * (size_t)nvox, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));   // (hv leaves scope; a pageable copy has returned by then anyway)
  }
  K3BArgs k{};
  k.M = M; k.N = N; k.LD = LD;
  k.cap = (mfx_thread().k3_cap > 0 && mfx_thread().k3_cap < MFX_K3B_CAP) ? mfx_thread().k3_cap : MFX_K3B_CAP;
  k.A = dA.as<double>(); k.Y = d_Y; k.G = dG.as<double>();
  k.nrm2 = dcol.as<double>(); k.aty = k.nrm2 + (size_t)BT * LD; k.ysq = k.aty + (size_t)BT * LD;
  k.st3 = dst3.as<double2>();
  k.items = ditm.as<K3Item>();
  k.thr = dsm.as<unsigned long long>(); k.seed = k.thr + BT; k.ncand = (int*)(k.seed + 7 * (size_t)BT);
  k.dbg = k3dbg ? k.seed + 3 * (size_t)BT : nullptr;
  k.cand_score = dcs.as<double>(); k.cand_tuple = dct.as<long>();
  k.part = dpart.as<double>();
  k.w = dout.as<double>(); k.sub = (long*)(k.w + (size_t)BT * 8); k.minobj = (double*)(k.sub + (size_t)BT * 8); k.yrec = k.minobj + BT;
  const size_t lds = (size_t)2 * MFX_K3M_KB * ((MFX_K3M_TI + MFX_K3M_TJ) * 64 + 2 * MFX_K3M_PAD) * 16 + (size_t)(MFX_K3M_TI + MFX_K3M_TJ) * 32 * 16 +
                     2 * MFX_K3M_KB * 16 + 2 * MFX_K3M_Q * 4 + 16 + 32;
  HIPCHK(hipFuncSetAttribute((const void*)mfx_k3b_screen_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  PackArgs pk{};
  pk.M = M; pk.K = 3; pk.has_csf = 0; pk.E = 0; pk.maxfasc = maxfasc; pk.csf_on = csf_on; pk.ear_on = ear_on; pk.num_params = num_params;
  if (k3dbg) HIPCHK(hipMemsetAsync(k.dbg, 0, sizeof(unsigned long long) * 4 * BT, st));
  // fallback of a candidate-list overflow (a voxel with more than MFX_K3B_CAP triples within 1e-9 |y|^2 of its optimum): the
  // voxel-by-voxel path of solve_k3.hip (itself backed by the full scan), enqueued for EVERY slot of a batch and gated on the
  // device by the slot's overflow flag - no host read; an idle launch sequence costs ~25 us per voxel
  // (the sequences of a batch's voxels are independent: they go to FL internal streams side by side, each with its own
  // buffers, between a fork behind the batch's last kernel and a join in front of the next batch's first)
  constexpr int FL = 4;
  if (int rc = lanes_setup(p->t->device)) return rc;
  MfxThread& TT = mfx_thread();
  const int nfl = std::min(FL, BT);
  struct FSet {
    StreamMem fG, fcol, fbs, fbt, fout, fk3;
    explicit FSet(hipStream_t s) : fG(s), fcol(s), fbs(s), fbt(s), fout(s), fk3(s) {}
  };
  FSet fs0(st), fs1(st), fs2(st), fs3(st);   // allocated and released in the order of the caller's stream
  FSet* fsets[FL] = {&fs0, &fs1, &fs2, &fs3};
  SolveArgs fav[FL];
  K3Bufs fkbv[FL];
  for (int l = 0; l < nfl; ++l) {
    SolveArgs fa{};
    for (int q = 0; q < 3; ++q) { fa.sizes[q] = N; fa.start[q] = (long)q * N; }
    fa.M = M; fa.Kp = 3; fa.Ntot = LD; fa.lda = LD; fa.ntuples = (long)N * N * N;
    fa.nblocks = (int)std::min<long>(16384, (fa.ntuples + 255) / 256);
    FSet& S = *fsets[l];
    HIPCHK(S.fG.alloc(sizeof(double) * (size_t)LD * LD));
    HIPCHK(S.fcol.alloc(sizeof(double) * ((size_t)LD + 2)));
    HIPCHK(S.fbs.alloc(sizeof(double) * (size_t)MFX_K3_CAP));
    HIPCHK(S.fbt.alloc(sizeof(long) * (size_t)MFX_K3_CAP));
    HIPCHK(S.fout.alloc(sizeof(double) * (MFX_GK + MFX_GK + 1 + (size_t)M)));
    HIPCHK(S.fk3.alloc(k3_buf_bytes(N)));
    fa.G = S.fG.as<double>(); fa.Aty = S.fcol.as<double>(); fa.ysq = fa.Aty + LD;
    fa.blk_score = S.fbs.as<double>(); fa.blk_tuple = S.fbt.as<long>();
    fa.w = S.fout.as<double>(); fa.sub = (long*)(fa.w + MFX_GK); fa.minobj = (double*)(fa.sub + MFX_GK); fa.yrec = fa.minobj + 1;
    fav[l] = fa;
    fkbv[l] = k3_bufs(S.fk3.as<char>());
  }
  if (int rc = mfx_prof_begin(st)) return rc;
  for (int q0 = 0; q0 < nvox; q0 += BT) {
    const int B = std::min(BT, nvox - q0);
    k.B = B; k.vox = dvox.as<int>() + q0;
    hipLaunchKernelGGL(mfx_rotate_voxels_kernel, dim3((M + MFX_ROT_ROWS - 1) / MFX_ROT_ROWS, 3, B), dim3(MFX_ROT_WG), 0, st, p->t->d, p->d,
                       d_peaks, peaks_ld, k.vox, dA.as<double>());
    hipLaunchKernelGGL(mfx_k3b_stats_kernel, dim3((LD + 255) / 256, B), dim3(256), 0, st, k);
    hipLaunchKernelGGL(mfx_k3b_gram_kernel, dim3((N + 63) / 64, (N + 127) / 128, 3 * B), dim3(256), 0, st, k);
    hipLaunchKernelGGL(mfx_k3b_items_kernel, dim3((unsigned)(((size_t)N * nblk3 * MFX_K3M_KB + 255) / 256), 2, B), dim3(256), 0, st, k);
    hipLaunchKernelGGL(mfx_k3b_pairs_kernel, dim3((N + 7) / 8, 3, B), dim3(256), 0, st, k);
    hipLaunchKernelGGL(mfx_k3b_greedy_kernel, dim3(3, B), dim3(256), 0, st, k);
    hipLaunchKernelGGL(mfx_k3b_screen_kernel, dim3((N + MFX_K3M_TJ * 32 - 1) / (MFX_K3M_TJ * 32), (N + MFX_K3M_TI * 32 - 1) / (MFX_K3M_TI * 32), B),
                       dim3(MFX_K3M_TI * 64), lds, st, k);
    hipLaunchKernelGGL(mfx_k3b_finalize_kernel, dim3(MFX_K3B_FW, B), dim3(256), 0, st, k);
    hipLaunchKernelGGL(mfx_k3b_finish_kernel, dim3(B), dim3(256), 0, st, k, pk, d_params, num_params);
    HIPCHK(hipGetLastError());
    // gated fallback (see above): fork, the slots dealt to the lanes, join
    HIPCHK(hipEventRecord(TT.ev_lane[4], st));
    const int nl = std::min(nfl, B);
    int rc_fb = MFX_OK;
    for (int l = 0; l < nl; ++l) HIPCHK(hipStreamWaitEvent(TT.s_lane[l], TT.ev_lane[4], 0));
    for (int b = 0; b < B && rc_fb == MFX_OK; ++b) {
      const int l = b % nl;
      const long v = h_list ? h_list[q0 + b] : q0 + b;
      SolveArgs& fa = fav[l];
      fa.A = dA.as<double>() + (size_t)b * M * LD;
      fa.y = d_Y + (size_t)v * M;
      fa.run_if = k.ncand + 2 * b + 1;
      rc_fb = launch_solver(fa, &fkbv[l], TT.s_lane[l]);
      if (rc_fb != MFX_OK) break;
      PackArgs pa = pk;
      pa.run_if = fa.run_if;
      pa.w = fa.w; pa.sub = fa.sub; pa.minobj = fa.minobj; pa.yrec = fa.yrec; pa.y = fa.y;
      pa.out = d_params + (size_t)v * num_params;
      hipLaunchKernelGGL(mfx_pack_params_kernel, dim3(1), dim3(64), 0, TT.s_lane[l], pa);
    }
    for (int l = 0; l < nl; ++l) {   // (also on the error path: the buffers are released in the order of `st`)
      const hipError_t e1 = hipEventRecord(TT.ev_lane[l], TT.s_lane[l]);
      const hipError_t e2 = hipStreamWaitEvent(st, TT.ev_lane[l], 0);
      if ((e1 != hipSuccess || e2 != hipSuccess) && rc_fb == MFX_OK) rc_fb = fail(MFX_ERR_HIP, "stream join failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    }
    if (rc_fb != MFX_OK) return rc_fb;
    HIPCHK(hipGetLastError());
    if (k3dbg) {   // developer diagnostics: synchronises
      std::vector<unsigned long long> h(4 * (size_t)B + 0), ht(B);
      std::vector<int> hn(2 * (size_t)B);
      HIPCHK(hipStreamSynchronize(st));
      HIPCHK(hipMemcpy(h.data(), k.dbg, sizeof(unsigned long long) * 4 * B, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(ht.data(), k.thr, sizeof(unsigned long long) * B, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(hn.data(), k.ncand, sizeof(int) * 2 * B, hipMemcpyDeviceToHost));
      for (int b = 0; b < B; ++b) {
        double t0, t1;
        memcpy(&t0, &h[4 * b + 3], 8); memcpy(&t1, &ht[b], 8);
        fprintf(stderr, "[k3] slot %d: scored %llu (on the spot %llu, always-flag hits %llu), listed %d, threshold at screen start %.9g -> final %.9g (%.3e below)\n", b, h[4 * b], h[4 * b + 1], h[4 * b + 2],
                hn[2 * b], t0, t1, (t1 - t0) / t1);
      }
      HIPCHK(hipMemsetAsync(k.dbg, 0, sizeof(unsigned long long) * 4 * B, st));
    }
  }
  return mfx_prof_end(st);
}

// Three fascicles (BASELINE config 5; opt-in, MFModel.fit itself stops at two): no fused kernel yet - voxel after voxel
// the rotated dictionaries are materialised into one row-major [M x (K N + extras)] matrix on the device and go through
// the explicit-dictionary solver (solve_generic.hip: Gram, one thread per index tuple, exact finalize in the
// reference's _3 / _4up arithmetic and scan order), then the params row is packed.  Everything is enqueued on `st`.
static int fit_class_generic(const mfx_plan* p, const double* d_Y, const double* d_peaks, int peaks_ld, const int* h_list,
                             int nvox, int K, const ExtrasHost& X, int maxfasc, int csf_on, int ear_on, double* d_params,
                             hipStream_t st) {
  const int M = p->d.M, N = p->t->d.N, NX = X.d.NX, E = X.d.E, has_csf = X.d.has_csf;
  const int Kp = K + has_csf + (E > 0);
  const int num_params = 1 + 2 * maxfasc + csf_on + 2 * ear_on + 2;
  if (Kp > MFX_GK) return fail(MFX_ERR_UNSUPPORTED, "at most %d sub-dictionaries", MFX_GK);
  SolveArgs a{};
  long Ntot = 0, ntup = 1;
  for (int k = 0; k < Kp; ++k) {
    const long sz = k < K ? N : ((has_csf && k == K) ? 1 : E);
    a.sizes[k] = sz; a.start[k] = Ntot; Ntot += sz;
    if (ntup > (1L << 42) / sz) return fail(MFX_ERR_UNSUPPORTED, "too many index tuples per voxel for the generic three-fascicle path");
    ntup *= sz;
  }
  a.M = M; a.Kp = Kp; a.Ntot = (int)Ntot; a.lda = Ntot; a.ntuples = ntup;
  a.nblocks = (int)std::min<long>(16384, (ntup + 255) / 256);
  const bool k3 = mfx_thread().k3_screen && k3_applies(a);
  const size_t nlist = k3 ? (size_t)MFX_K3_CAP : (size_t)a.nblocks;
  // two voxels in flight on two internal streams, each with its own set of buffers (at config 5: 162 MB of Gram, 11 MB
  // of dictionary, 16 MB of candidate list per set)
  constexpr int LANES = 2;   // (3: no further gain, 4: slower - measured at config 5)
  MfxThread& T = mfx_thread();
  if (int rc = lanes_setup(p->t->device)) return rc;
  const int nl = std::min(nvox, LANES);
  struct Set {
    StreamMem dA, dG, dAty, dysq, dbs, dbt, dw, dsub, dobj, dyrec, dk3;
    explicit Set(hipStream_t s) : dA(s), dG(s), dAty(s), dysq(s), dbs(s), dbt(s), dw(s), dsub(s), dobj(s), dyrec(s), dk3(s) {}
  };
  Set set0(st), set1(st);   // allocated and released in the order of the caller's stream: before the fork, after the join
  Set* sets[LANES] = {&set0, &set1};
  SolveArgs av[LANES];
  K3Bufs kbv[LANES];
  for (int l = 0; l < nl; ++l) {
    Set& S = *sets[l];
    HIPCHK(S.dk3.alloc(k3_buf_bytes(N)));
    kbv[l] = k3_bufs(S.dk3.as<char>());
    HIPCHK(S.dA.alloc(sizeof(double) * (size_t)M * Ntot));
    HIPCHK(S.dG.alloc(sizeof(double) * (size_t)Ntot * Ntot));
    HIPCHK(S.dAty.alloc(sizeof(double) * Ntot));
    HIPCHK(S.dysq.alloc(sizeof(double) * 2));
    HIPCHK(S.dbs.alloc(sizeof(double) * nlist));
    HIPCHK(S.dbt.alloc(sizeof(long) * nlist));
    HIPCHK(S.dw.alloc(sizeof(double) * MFX_GK));
    HIPCHK(S.dsub.alloc(sizeof(long) * MFX_GK));
    HIPCHK(S.dobj.alloc(sizeof(double)));
    HIPCHK(S.dyrec.alloc(sizeof(double) * M));
    av[l] = a;
    av[l].A = S.dA.as<double>(); av[l].G = S.dG.as<double>(); av[l].Aty = S.dAty.as<double>(); av[l].ysq = S.dysq.as<double>();
    av[l].blk_score = S.dbs.as<double>(); av[l].blk_tuple = S.dbt.as<long>(); av[l].w = S.dw.as<double>(); av[l].sub = S.dsub.as<long>();
    av[l].minobj = S.dobj.as<double>(); av[l].yrec = S.dyrec.as<double>();
    // the voxel-independent extra columns sit behind the fascicle blocks (sub-dictionary order of mf.py:391-408)
    if (NX > 0)
      HIPCHK(hipMemcpy2DAsync(S.dA.as<double>() + (size_t)K * N, sizeof(double) * Ntot, X.d.x, sizeof(double) * NX, sizeof(double) * NX, M,
                              hipMemcpyDeviceToDevice, st));
  }
  if (int rc = mfx_prof_begin(st)) return rc;
  hipStream_t ls[LANES] = {st, st};
  if (nl > 1) {   // fork
    HIPCHK(hipEventRecord(T.ev_lane[4], st));
    for (int l = 0; l < nl; ++l) { ls[l] = T.s_lane[l]; HIPCHK(hipStreamWaitEvent(ls[l], T.ev_lane[4], 0)); }
  }
  int rc_loop = MFX_OK;
  for (int q = 0; q < nvox && rc_loop == MFX_OK; ++q) {
    const int l = q % nl;
    const long v = h_list ? h_list[q] : q;
    dim3 grid((M + MFX_ROT_ROWS - 1) / MFX_ROT_ROWS, std::max(K, 1));
    if (K > 0)
      hipLaunchKernelGGL(mfx_rotate_kernel, grid, dim3(MFX_ROT_WG), 0, ls[l], p->t->d, p->d, d_peaks + (size_t)v * peaks_ld, 0,
                         sets[l]->dA.as<double>(), (long)N, Ntot);   // (an explicit plan normalises the direction inside mfx_row_desc)
    av[l].y = d_Y + (size_t)v * M;
    rc_loop = launch_solver(av[l], k3 ? &kbv[l] : nullptr, ls[l]);
    if (rc_loop != MFX_OK) break;
    PackArgs pa{};
    pa.w = av[l].w; pa.sub = av[l].sub; pa.minobj = av[l].minobj; pa.yrec = av[l].yrec; pa.y = av[l].y;
    pa.M = M; pa.K = K; pa.has_csf = has_csf; pa.E = E; pa.maxfasc = maxfasc; pa.csf_on = csf_on; pa.ear_on = ear_on;
    pa.num_params = num_params; pa.out = d_params + (size_t)v * num_params;
    hipLaunchKernelGGL(mfx_pack_params_kernel, dim3(1), dim3(64), 0, ls[l], pa);
  }
  if (nl > 1) {   // join (also on the error path: the buffers are released in the order of `st`)
    for (int l = 0; l < nl; ++l) {
      const hipError_t e1 = hipEventRecord(T.ev_lane[l], ls[l]);
      const hipError_t e2 = hipStreamWaitEvent(st, T.ev_lane[l], 0);
      if ((e1 != hipSuccess || e2 != hipSuccess) && rc_loop == MFX_OK) rc_loop = fail(MFX_ERR_HIP, "stream join failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    }
  }
  if (rc_loop != MFX_OK) return rc_loop;
  HIPCHK(hipGetLastError());
  return mfx_prof_end(st);
}

extern "C" int mfx_fit_batch_dev(const mfx_plan* p, const double* d_Y, const double* d_peaks, int maxfasc, int csf_on,
                                 int ear_on, const double* d_sig_csf, const double* d_sig_ear, int E, int64_t V,
                                 double* d_params_out, void* stream) {
  if (!p || !d_Y || !d_params_out || V < 0) return fail(MFX_ERR_ARG, "mfx_fit_batch_dev: bad argument");
  if (maxfasc < 0 || maxfasc > 3) return fail(MFX_ERR_ARG, "maxfasc must be 0..3 (MFModel.fit itself allows 2: MAX_FASC, mf.py:467)");
  if (maxfasc > 0 && !d_peaks) return fail(MFX_ERR_ARG, "mfx_fit_batch_dev: d_peaks is null but maxfasc = %d", maxfasc);
  if (V == 0) return MFX_OK;
  if (V > 0x7fffffff) return fail(MFX_ERR_ARG, "V too large for one launch");
  if (int rc = require_device(p->t->device)) return rc;
  hipStream_t st = (hipStream_t)stream;
  HostTrace tr;
  if (int rc = mfx_fb_begin(st)) return rc;
  tr.mark("dev: fb_begin");
  ExtrasHost X;
  if (int rc = X.build(p->d.M, csf_on ? 1 : 0, ear_on ? E : 0, d_sig_csf, d_sig_ear, st)) return rc;
  tr.mark("dev: extras");
  if (int rc = fit_class_dev(p, d_Y, d_peaks, 3 * maxfasc, nullptr, nullptr, (int)V, maxfasc, csf_on ? 1 : 0, ear_on ? 1 : 0, X, maxfasc,
                             csf_on ? 1 : 0, ear_on ? 1 : 0, d_params_out, st)) return rc;
  tr.mark("dev: class launched (incl. scratch frees)");
  const int rc_end = mfx_fb_end(st);
  tr.mark("dev: fb_end");
  return rc_end;
}

// ---- host-buffer voxel loop: pinned staging + chunked H2D on a copy stream overlapped with the kernels on a compute
// stream (reference loop: mf.py:976-1032).  `rows` (optional) fuses the reference's mask gather `data[mask > 0]`
// (mf.py:644, 1020-1022) into the staging copy: voxel v's signal is the M doubles at Y + rows[v] * M.
static void pipe_release(MfxThread& T) {
  for (int q = 0; q < 2; ++q) { if (T.stage[q]) (void)hipHostFree(T.stage[q]); T.stage[q] = nullptr; }
  T.stage_bytes = 0;
  for (int q = 0; q < 6; ++q) { if (T.pool[q]) (void)hipFree(T.pool[q]); T.pool[q] = nullptr; T.pool_bytes[q] = 0; }
}

static int pipe_setup(int device, size_t want_bytes) {
  MfxThread& T = mfx_thread();
  if (T.pipe_device != device) {
    pipe_release(T);
    if (T.s_copy) (void)hipStreamDestroy(T.s_copy);
    if (T.s_comp) (void)hipStreamDestroy(T.s_comp);
    T.s_copy = T.s_comp = nullptr;
    for (int q = 0; q < 2; ++q) { if (T.ev_h2d[q]) (void)hipEventDestroy(T.ev_h2d[q]); T.ev_h2d[q] = nullptr; }
    HIPCHK(hipStreamCreateWithFlags(&T.s_copy, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&T.s_comp, hipStreamNonBlocking));
    for (int q = 0; q < 2; ++q) HIPCHK(hipEventCreateWithFlags(&T.ev_h2d[q], hipEventDisableTiming));
    T.pipe_device = device;
  }
  if (T.stage_bytes < want_bytes) {
    for (int q = 0; q < 2; ++q) { if (T.stage[q]) (void)hipHostFree(T.stage[q]); T.stage[q] = nullptr; }
    T.stage_bytes = 0;
    for (int q = 0; q < 2; ++q) HIPCHK(hipHostMalloc(&T.stage[q], want_bytes, hipHostMallocDefault));
    T.stage_bytes = want_bytes;
  }
  return MFX_OK;
}

// grow-only device buffer `slot` of the calling thread's pipeline (valid after pipe_setup on the same device)
static int pool_get(int slot, size_t bytes, void** out) {
  MfxThread& T = mfx_thread();
  if (T.pool_bytes[slot] < bytes) {
    if (T.pool[slot]) HIPCHK(hipFree(T.pool[slot]));
    T.pool[slot] = nullptr;
    T.pool_bytes[slot] = 0;
    const size_t want = bytes + bytes / 8;   // head room: slabs of a volume differ a little in size
    HIPCHK(hipMalloc(&T.pool[slot], want));
    T.pool_bytes[slot] = want;
  }
  *out = T.pool[slot];
  return MFX_OK;
}

extern "C" int mfx_thread_release(void) {
  MfxThread& T = mfx_thread();
  int cur = 0;
  HIPCHK(hipGetDevice(&cur));
  if (T.pipe_device >= 0) {
    HIPCHK(hipSetDevice(T.pipe_device));
    if (T.s_copy) (void)hipStreamSynchronize(T.s_copy);
    if (T.s_comp) (void)hipStreamSynchronize(T.s_comp);
    pipe_release(T);
  }
  for (auto& a : T.arenas)   // idle scratch arenas of this thread (any device): wait for their work, then free the blocks
    if (a.live == 0) {
      (void)hipSetDevice(a.device);
      (void)hipDeviceSynchronize();
      arena_drop(a);
    }
  T.arenas.erase(std::remove_if(T.arenas.begin(), T.arenas.end(), [](const MfxThread::Arena& a) { return a.live == 0; }), T.arenas.end());
  HIPCHK(hipSetDevice(cur));
  return MFX_OK;
}

namespace {
struct PoolPtr {     // borrowed pointer into the thread's buffer pool
  void* p = nullptr;
  template <class U> U* as() const { return (U*)p; }
};
}  // namespace

// ---- a volume kept the way a NIfTI file stores it: one 3-D image per measurement, [M][nvox] scalars of the file's
// data type.  The volume goes to the device as it is (slices through the pinned staging buffers) and the reference's
// `get_fdata()[mask > 0]` (mf.py:623-657: conversion to float64, scaling, ROI gather) happens there
template <class S>
__global__ __launch_bounds__(256) void mfx_volume_gather_kernel(const S* __restrict__ vol, long long nvox,
                                                                const long long* __restrict__ vox, int V, int M,
                                                                double slope, double inter, int scaled,
                                                                double* __restrict__ Y) {
  __shared__ double s[64][65];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int v0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
  const long long src = (v0 + lane < V) ? vox[v0 + lane] : -1;
  for (int r = w; r < 64; r += 4) {          // lane = voxel: one image per row of the tile
    double x = 0.0;
    if (src >= 0 && m0 + r < M) {
      x = (double)vol[(size_t)(m0 + r) * (size_t)nvox + (size_t)src];
      if (scaled) x = __dadd_rn(__dmul_rn(x, slope), inter);      // two roundings, as NumPy's data * slope + inter
    }
    s[r][lane] = x;
  }
  __syncthreads();
  for (int r = w; r < 64; r += 4)            // lane = measurement: one voxel's signal per row
    if (v0 + r < V && m0 + lane < M) Y[(size_t)(v0 + r) * M + m0 + lane] = s[lane][r];
}

static size_t volume_elem_bytes(int code) {   // NIfTI-1 datatype codes
  switch (code) {
    case 2: case 256: return 1;
    case 4: case 512: return 2;
    case 8: case 768: case 16: return 4;
    case 64: return 8;
    default: return 0;
  }
}

namespace {
struct HostSource {    // where the host-buffer voxel loop finds the signals
  const double* Y = nullptr;          // row-major [.. x M] doubles; voxel v: row rows[v] (rows == NULL: v)
  const int64_t* rows = nullptr;
  const void* vol = nullptr;          // or a measurement-major volume [M][nvox] of NIfTI type `dtype`; voxel v: element vox[v] of every image
  int dtype = 0;
  double slope = 0.0, inter = 0.0;
  int64_t nvox = 0;
  const int64_t* vox = nullptr;
};
}  // namespace

static int volume_gather_launch(const HostSource& src, const void* d_vol, const long long* d_vox, int V, int M, double* d_Y,
                                hipStream_t st) {
  const int scaled = (src.slope != 0.0 && !(src.slope == 1.0 && src.inter == 0.0) && std::isfinite(src.slope)) ? 1 : 0;
  const dim3 grid((unsigned)((V + 63) / 64), (unsigned)((M + 63) / 64));
#define MFX_VG(TYPE) hipLaunchKernelGGL(mfx_volume_gather_kernel<TYPE>, grid, dim3(256), 0, st, (const TYPE*)d_vol, (long long)src.nvox, d_vox, V, M, src.slope, src.inter, scaled, d_Y)
  switch (src.dtype) {
    case 2: MFX_VG(uint8_t); break;
    case 256: MFX_VG(int8_t); break;
    case 4: MFX_VG(int16_t); break;
    case 512: MFX_VG(uint16_t); break;
    case 8: MFX_VG(int32_t); break;
    case 768: MFX_VG(uint32_t); break;
    case 16: MFX_VG(float); break;
    case 64: MFX_VG(double); break;
    default: return fail(MFX_ERR_ARG, "unsupported volume data type %d", src.dtype);
  }
#undef MFX_VG
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

static int fit_batch_host(const mfx_plan* p, const HostSource& src, const int32_t* K,
                          const uint8_t* csf, const uint8_t* ear, const double* peaks, int maxfasc, int csf_on,
                          int ear_on, const double* sig_csf, const double* sig_ear, int E, int64_t V,
                          double* params_out) {
  const double* Y = src.Y;
  const int64_t* rows = src.rows;
  if (!p || (!Y && !src.vol) || !K || !params_out || V < 0) return fail(MFX_ERR_ARG, "mfx_fit_batch: bad argument");
  if (maxfasc < 0 || maxfasc > 3) return fail(MFX_ERR_ARG, "maxfasc must be 0..3 (MFModel.fit itself allows 2: MAX_FASC, mf.py:467)");
  if (maxfasc > 0 && !peaks) return fail(MFX_ERR_ARG, "mfx_fit_batch: peaks is null but maxfasc = %d", maxfasc);
  if (V > 0x7fffffff) return fail(MFX_ERR_ARG, "V too large");
  if (int rc = require_device(p->t->device)) return rc;
  csf_on = csf_on ? 1 : 0;
  ear_on = ear_on ? 1 : 0;
  const int M = p->d.M;
  const int num_params = 1 + 2 * maxfasc + csf_on + 2 * ear_on + 2;
  if (V == 0) return MFX_OK;
  // chunks of up to ~128 MB of signal (MFX_CHUNK_MB; per chunk every voxel class is one launch sequence: with 32 MB chunks
  // a mixed ROI spent 12 % more time in launch tails), the first ones smaller (1/8, 1/4, 1/2 of that) so that the first kernel starts
  // after a short staging copy; per chunk the voxels are binned by class (K, csf, ear); the direction check runs
  // once per batch on the host (the reference checks per voxel inside interp_PGSE_from_multishell, mf_utils.py:1798-1802)
  HostTrace tr;
  static const int64_t chunk_mb = [] { const char* e = std::getenv("MFX_CHUNK_MB"); const int v = e ? std::atoi(e) : 0; return (int64_t)(v > 0 ? v : 128); }();
  const int64_t CH = std::max<int64_t>(1024, std::min<int64_t>(V, (chunk_mb << 20) / ((int64_t)M * 8)));
  std::vector<int64_t> cstart;                   // first voxel of every chunk, then V
  for (int64_t v0 = 0, n = src.vol ? V : std::max<int64_t>(1024, CH / 8); v0 < V; n = std::min<int64_t>(CH, 2 * n)) {   // (a volume is on the device before the first kernel: one chunk)
    cstart.push_back(v0);
    v0 += std::min<int64_t>(n, V - v0);
  }
  cstart.push_back(V);
  const int64_t nch = (int64_t)cstart.size() - 1;
  std::vector<int> list((size_t)V);              // voxel lists, chunk-major then class-major
  std::vector<int> cnt((size_t)nch * 16, 0);
  {
    std::vector<uint8_t> cls((size_t)V);
    for (int64_t c = 0; c < nch; ++c)
      for (int64_t v = cstart[c]; v < cstart[c + 1]; ++v) {
        const int k = K[v];
        if (k < 0 || k > maxfasc) return fail(MFX_ERR_ARG, "voxel %lld: numfasc %d outside 0..%d", (long long)v, k, maxfasc);
        const int cf = (csf && csf[v]) ? 1 : 0, e = (ear && ear[v]) ? 1 : 0;
        if ((cf && !csf_on) || (e && !ear_on)) return fail(MFX_ERR_ARG, "voxel %lld has a CSF/EAR flag but csf_on/ear_on is 0", (long long)v);
        for (int f = 0; f < k && !p->d.normalise; ++f) {   // (explicit rotate_atom plans normalise the direction themselves)
          const double* d = peaks + (size_t)v * 3 * maxfasc + 3 * f;
          const double nrm = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
          if (!(std::fabs(1 - nrm) <= 1e-3))
            return fail(MFX_ERR_DIR_NORM, "Orientation vector of the new signal must have unit norm. Detected %g.", nrm);
        }
        cls[(size_t)v] = (uint8_t)(k * 4 + cf * 2 + e);
        ++cnt[(size_t)c * 16 + cls[(size_t)v]];
      }
    std::vector<size_t> pos((size_t)nch * 16);
    size_t acc = 0;
    for (size_t q = 0; q < pos.size(); ++q) { pos[q] = acc; acc += (size_t)cnt[q]; }
    for (int64_t c = 0; c < nch; ++c)
      for (int64_t v = cstart[c]; v < cstart[c + 1]; ++v) list[pos[(size_t)c * 16 + cls[(size_t)v]]++] = (int)v;
  }
  tr.mark("voxels binned");
  const size_t vol_bytes = src.vol ? volume_elem_bytes(src.dtype) * (size_t)src.nvox * (size_t)M : 0;
  if (int rc = pipe_setup(p->t->device, src.vol ? std::min<size_t>(vol_bytes, (size_t)64 << 20) : (size_t)CH * M * sizeof(double))) return rc;
  MfxThread& T = mfx_thread();
  PoolPtr dY, dpk, dpar, dlist;
  DevMem dsc, dse;
  if (int rc = pool_get(0, sizeof(double) * (size_t)V * M, &dY.p)) return rc;
  if (int rc = pool_get(1, sizeof(double) * (size_t)V * 3 * std::max(maxfasc, 1), &dpk.p)) return rc;
  if (int rc = pool_get(2, sizeof(double) * (size_t)V * num_params, &dpar.p)) return rc;
  if (int rc = pool_get(3, sizeof(int) * (size_t)V, &dlist.p)) return rc;
  PoolPtr dvol, dvox;
  if (src.vol) {
    if (int rc = pool_get(4, vol_bytes, &dvol.p)) return rc;
    if (int rc = pool_get(5, sizeof(int64_t) * (size_t)V, &dvox.p)) return rc;
  }
  tr.mark("buffers ready");
  // everything below is stream-ordered; any failure drains both streams before returning
  auto run = [&]() -> int {
    if (maxfasc > 0) HIPCHK(hipMemcpyAsync(dpk.p, peaks, sizeof(double) * (size_t)V * 3 * maxfasc, hipMemcpyHostToDevice, T.s_comp));
    HIPCHK(hipMemcpyAsync(dlist.p, list.data(), sizeof(int) * (size_t)V, hipMemcpyHostToDevice, T.s_comp));
    HIPCHK(hipMemsetAsync(dpar.p, 0, sizeof(double) * (size_t)V * num_params, T.s_comp));
    if (csf_on) {
      if (!sig_csf) return fail(MFX_ERR_ARG, "sig_csf missing");
      HIPCHK(dsc.alloc(sizeof(double) * M));
      HIPCHK(hipMemcpyAsync(dsc.p, sig_csf, sizeof(double) * M, hipMemcpyHostToDevice, T.s_comp));
    }
    if (ear_on) {
      if (!sig_ear || E < 1) return fail(MFX_ERR_ARG, "sig_ear missing");
      HIPCHK(dse.alloc(sizeof(double) * (size_t)M * E));
      HIPCHK(hipMemcpyAsync(dse.p, sig_ear, sizeof(double) * (size_t)M * E, hipMemcpyHostToDevice, T.s_comp));
    }
    if (int rc = mfx_fb_begin(T.s_comp)) return rc;
    ExtrasHost xs[4];
    for (int ce = 0; ce < 4; ++ce) {
      if (((ce >> 1) && !csf_on) || ((ce & 1) && !ear_on)) continue;  // class cannot occur
      if (int rc = xs[ce].build(M, ce >> 1, (ce & 1) ? E : 0, dsc.as<double>(), dse.as<double>(), T.s_comp)) return rc;
    }
    size_t off = 0;
    if (src.vol) {   // the whole volume, slice by slice through the two staging buffers; then one gather on the device
      HIPCHK(hipMemcpyAsync(dvox.p, src.vox, sizeof(int64_t) * (size_t)V, hipMemcpyHostToDevice, T.s_comp));
      const size_t slice = T.stage_bytes;
      int64_t q = 0;
      for (size_t o = 0; o < vol_bytes; o += slice, ++q) {
        const size_t nb = std::min(slice, vol_bytes - o);
        if (q >= 2) HIPCHK(hipEventSynchronize(T.ev_h2d[q & 1]));
        std::memcpy(T.stage[q & 1], (const char*)src.vol + o, nb);
        HIPCHK(hipMemcpyAsync((char*)dvol.p + o, T.stage[q & 1], nb, hipMemcpyHostToDevice, T.s_copy));
        HIPCHK(hipEventRecord(T.ev_h2d[q & 1], T.s_copy));
      }
      HIPCHK(hipStreamWaitEvent(T.s_comp, T.ev_h2d[(q + 1) & 1], 0));   // the last copy: everything before it on s_copy is done too
      tr.mark("  volume staged");
      if (int rc = volume_gather_launch(src, dvol.p, dvox.as<long long>(), (int)V, M, dY.as<double>(), T.s_comp)) return rc;
    }
    for (int64_t c = 0; c < nch; ++c) {
      const int64_t v0 = cstart[c], nv = cstart[c + 1] - v0;
      if (!src.vol) {
        double* stg = (double*)T.stage[c & 1];
        if (c >= 2) HIPCHK(hipEventSynchronize(T.ev_h2d[c & 1]));   // the copy that last used this staging buffer is done
        tr.mark("  staging buffer free");
        if (rows) {
          for (int64_t v = 0; v < nv; ++v) std::memcpy(stg + (size_t)v * M, Y + (size_t)rows[v0 + v] * M, sizeof(double) * M);
        } else {
          std::memcpy(stg, Y + (size_t)v0 * M, sizeof(double) * (size_t)nv * M);
        }
        tr.mark("  chunk gathered");
        HIPCHK(hipMemcpyAsync(dY.as<double>() + (size_t)v0 * M, stg, sizeof(double) * (size_t)nv * M, hipMemcpyHostToDevice, T.s_copy));
        HIPCHK(hipEventRecord(T.ev_h2d[c & 1], T.s_copy));
        HIPCHK(hipStreamWaitEvent(T.s_comp, T.ev_h2d[c & 1], 0));
        tr.mark("  copy queued");
      }
      for (int q = 0; q < 16; ++q) {
        const int n = cnt[(size_t)c * 16 + q];
        if (!n) continue;
        if (int rc = fit_class_dev(p, dY.as<double>(), dpk.as<double>(), 3 * maxfasc, dlist.as<int>() + off, list.data() + off, n, q >> 2,
                                   (q >> 1) & 1, q & 1, xs[q & 3], maxfasc, csf_on, ear_on, dpar.as<double>(), T.s_comp)) return rc;
        off += (size_t)n;
      }
      tr.mark("  kernels queued");
    }
    if (int rc = mfx_fb_end(T.s_comp)) return rc;
    tr.mark("all chunks queued");
    HIPCHK(hipMemcpyAsync(params_out, dpar.p, sizeof(double) * (size_t)V * num_params, hipMemcpyDeviceToHost, T.s_comp));
    return MFX_OK;
  };
  int rc = run();
  const hipError_t e1 = hipStreamSynchronize(T.s_copy), e2 = hipStreamSynchronize(T.s_comp);
  tr.mark("done");
  if (rc == MFX_OK && (e1 != hipSuccess || e2 != hipSuccess))
    rc = fail(MFX_ERR_HIP, "kernel execution failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  return rc;
}

extern "C" int mfx_fit_batch_rows(const mfx_plan* p, const double* Y, const int64_t* rows, const int32_t* K,
                                  const uint8_t* csf, const uint8_t* ear, const double* peaks, int maxfasc, int csf_on,
                                  int ear_on, const double* sig_csf, const double* sig_ear, int E, int64_t V,
                                  double* params_out) {
  if (!Y) return fail(MFX_ERR_ARG, "mfx_fit_batch: bad argument");
  HostSource src;
  src.Y = Y;
  src.rows = rows;
  return fit_batch_host(p, src, K, csf, ear, peaks, maxfasc, csf_on, ear_on, sig_csf, sig_ear, E, V, params_out);
}

extern "C" int mfx_fit_batch_volume(const mfx_plan* p, const void* vol, int vol_dtype, double scl_slope, double scl_inter,
                                    int64_t nvox, const int64_t* vox, const int32_t* K, const uint8_t* csf,
                                    const uint8_t* ear, const double* peaks, int maxfasc, int csf_on, int ear_on,
                                    const double* sig_csf, const double* sig_ear, int E, int64_t V, double* params_out) {
  if (!vol || !vox || nvox <= 0) return fail(MFX_ERR_ARG, "mfx_fit_batch_volume: bad argument");
  if (!volume_elem_bytes(vol_dtype)) return fail(MFX_ERR_ARG, "mfx_fit_batch_volume: unsupported NIfTI data type %d", vol_dtype);
  for (int64_t v = 0; v < V; ++v)
    if (vox[v] < 0 || vox[v] >= nvox) return fail(MFX_ERR_ARG, "mfx_fit_batch_volume: voxel %lld: index %lld outside the volume", (long long)v, (long long)vox[v]);
  HostSource src;
  src.vol = vol;
  src.dtype = vol_dtype;
  src.slope = scl_slope;
  src.inter = scl_inter;
  src.nvox = nvox;
  src.vox = vox;
  return fit_batch_host(p, src, K, csf, ear, peaks, maxfasc, csf_on, ear_on, sig_csf, sig_ear, E, V, params_out);
}

// The same gather for any per-voxel quantity that arrives as a file-order volume (fascicle directions, tensors: what
// MFModel.fit reads beside the data, mf.py:693-800): rows_out [V x ncomp] float64 on the host.
extern "C" int mfx_volume_rows(const void* vol, int vol_dtype, double scl_slope, double scl_inter, int64_t nvox, int ncomp,
                               const int64_t* vox, int64_t V, double* rows_out, int device) {
  if (!vol || !vox || !rows_out || nvox <= 0 || ncomp < 1 || V < 0) return fail(MFX_ERR_ARG, "mfx_volume_rows: bad argument");
  const size_t es = volume_elem_bytes(vol_dtype);
  if (!es) return fail(MFX_ERR_ARG, "mfx_volume_rows: unsupported NIfTI data type %d", vol_dtype);
  if (V == 0) return MFX_OK;
  if (V > 0x7fffffff) return fail(MFX_ERR_ARG, "V too large");
  for (int64_t v = 0; v < V; ++v)
    if (vox[v] < 0 || vox[v] >= nvox) return fail(MFX_ERR_ARG, "mfx_volume_rows: voxel %lld: index %lld outside the volume", (long long)v, (long long)vox[v]);
  if (int rc = require_device(device)) return rc;
  const size_t vol_bytes = es * (size_t)nvox * (size_t)ncomp;
  if (int rc = pipe_setup(device, std::min<size_t>(vol_bytes, (size_t)64 << 20))) return rc;
  MfxThread& T = mfx_thread();
  PoolPtr dvol, dvox;
  if (int rc = pool_get(4, vol_bytes, &dvol.p)) return rc;
  if (int rc = pool_get(5, sizeof(int64_t) * (size_t)V, &dvox.p)) return rc;
  DevMem drows;
  HIPCHK(drows.alloc(sizeof(double) * (size_t)V * ncomp));
  HostSource src;
  src.vol = vol; src.dtype = vol_dtype; src.slope = scl_slope; src.inter = scl_inter; src.nvox = nvox; src.vox = vox;
  auto run = [&]() -> int {
    HIPCHK(hipMemcpyAsync(dvox.p, vox, sizeof(int64_t) * (size_t)V, hipMemcpyHostToDevice, T.s_comp));
    const size_t slice = T.stage_bytes;
    int64_t q = 0;
    for (size_t o = 0; o < vol_bytes; o += slice, ++q) {
      const size_t nb = std::min(slice, vol_bytes - o);
      if (q >= 2) HIPCHK(hipEventSynchronize(T.ev_h2d[q & 1]));
      std::memcpy(T.stage[q & 1], (const char*)vol + o, nb);
      HIPCHK(hipMemcpyAsync((char*)dvol.p + o, T.stage[q & 1], nb, hipMemcpyHostToDevice, T.s_copy));
      HIPCHK(hipEventRecord(T.ev_h2d[q & 1], T.s_copy));
    }
    HIPCHK(hipStreamWaitEvent(T.s_comp, T.ev_h2d[(q + 1) & 1], 0));
    if (int rc = volume_gather_launch(src, dvol.p, dvox.as<long long>(), (int)V, ncomp, drows.as<double>(), T.s_comp)) return rc;
    HIPCHK(hipMemcpyAsync(rows_out, drows.p, sizeof(double) * (size_t)V * ncomp, hipMemcpyDeviceToHost, T.s_comp));
    return MFX_OK;
  };
  int rc = run();
  const hipError_t e1 = hipStreamSynchronize(T.s_copy), e2 = hipStreamSynchronize(T.s_comp);
  if (rc == MFX_OK && (e1 != hipSuccess || e2 != hipSuccess))
    rc = fail(MFX_ERR_HIP, "mfx_volume_rows failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  return rc;
}

extern "C" int mfx_fit_batch(const mfx_plan* p, const double* Y, const int32_t* K, const uint8_t* csf,
                             const uint8_t* ear, const double* peaks, int maxfasc, int csf_on, int ear_on,
                             const double* sig_csf, const double* sig_ear, int E, int64_t V, double* params_out) {
  return mfx_fit_batch_rows(p, Y, nullptr, K, csf, ear, peaks, maxfasc, csf_on, ear_on, sig_csf, sig_ear, E, V, params_out);
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
                       d_dirs + 3 * b0, normalise_dirs, d_out + (size_t)b0 * M * p->t->d.N, (long)M * p->t->d.N, (long)p->t->d.N);
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
// cleanup_2fascicles: the per-voxel selection (ref mf.py:170-335)
extern "C" int mfx_cleanup_2fascicles_dev(const double* d_f1, const double* d_f2, const double* d_p1, const double* d_p2, int64_t n,
                                          double cos_min, double ratio, double w_keep, double w_small, double* d_peaks_out,
                                          double* d_count_out, void* stream) {
  if (n < 0 || (n > 0 && (!d_f1 || !d_f2 || !d_p1 || !d_p2 || !d_peaks_out || !d_count_out)))
    return fail(MFX_ERR_ARG, "mfx_cleanup_2fascicles_dev: bad argument");
  if (n == 0) return MFX_OK;
  const int64_t blocks = (n + 255) / 256;
  if (blocks > 0x7fffffff) return fail(MFX_ERR_ARG, "too many voxels for one launch");
  CleanupArgs a{d_f1, d_f2, d_p1, d_p2, (long)n, cos_min, ratio, w_keep, w_small, d_peaks_out, d_count_out};
  hipLaunchKernelGGL(mfx_cleanup_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

extern "C" int mfx_cleanup_2fascicles(const double* f1, const double* f2, const double* p1, const double* p2, int64_t n,
                                      double cos_min, double ratio, double w_keep, double w_small, double* peaks_out,
                                      double* count_out, int device) {
  if (n < 0 || (n > 0 && (!f1 || !f2 || !p1 || !p2 || !peaks_out || !count_out)))
    return fail(MFX_ERR_ARG, "mfx_cleanup_2fascicles: bad argument");
  if (n == 0) return MFX_OK;
  if (int rc = require_device(device)) return rc;
  DevMem din, dout;
  HIPCHK(din.alloc(sizeof(double) * 8 * (size_t)n));
  HIPCHK(dout.alloc(sizeof(double) * 7 * (size_t)n));
  double* b = din.as<double>();
  HIPCHK(hipMemcpy(b, f1, sizeof(double) * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b + n, f2, sizeof(double) * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b + 2 * n, p1, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b + 5 * n, p2, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
  if (int rc = mfx_cleanup_2fascicles_dev(b, b + n, b + 2 * n, b + 5 * n, n, cos_min, ratio, w_keep, w_small, dout.as<double>(),
                                          dout.as<double>() + 6 * n, nullptr)) return rc;
  HIPCHK(hipMemcpy(peaks_out, dout.p, sizeof(double) * 6 * n, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(count_out, dout.as<double>() + 6 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
  return MFX_OK;
}

// ---------------------------------------------------------------------------------------------
// explicit-dictionary solver
extern "C" int mfx_solve_exhaustive(const double* A, int64_t lda, int M, const int64_t* dicsizes, int Kp, const double* y,
                                    double* w, int64_t* sub, int64_t* tot, double* min_obj, double* y_rec) {
  if (!A || !dicsizes || !y || !w || !sub || !tot || !min_obj || !y_rec || M < 1 || Kp < 1)
    return fail(MFX_ERR_ARG, "mfx_solve_exhaustive: bad argument");
  if (Kp > MFX_GK) return fail(MFX_ERR_UNSUPPORTED, "at most %d sub-dictionaries are supported (got %d)", MFX_GK, Kp);
  {   // runs on the calling thread's current device
    int dev = 0;
    if (mfx_device_count() <= 0) return fail(MFX_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    HIPCHK(hipGetDevice(&dev));
    if (int rc = require_device(dev)) return rc;
  }
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
  const bool k3 = mfx_thread().k3_screen && k3_applies(a);
  const size_t nlist = k3 ? (size_t)MFX_K3_CAP : (size_t)a.nblocks;
  std::vector<double> Ac((size_t)M * Ntot);
  for (int k = 0; k < M; ++k) std::memcpy(&Ac[(size_t)k * Ntot], A + (size_t)k * lda, sizeof(double) * Ntot);
  double *dA = nullptr, *dy = nullptr, *dG = nullptr, *dAty = nullptr, *dysq = nullptr, *dbs = nullptr, *dw = nullptr,
         *dobj = nullptr, *dyrec = nullptr;
  long *dbt = nullptr, *dsub = nullptr;
  char* dk3 = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(dA); (void)hipFree(dy); (void)hipFree(dG); (void)hipFree(dAty); (void)hipFree(dysq); (void)hipFree(dbs);
    (void)hipFree(dw); (void)hipFree(dobj); (void)hipFree(dyrec); (void)hipFree(dbt); (void)hipFree(dsub); (void)hipFree(dk3);
  };
#define SCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(MFX_ERR_HIP, "%s failed: %s", #x, hipGetErrorString(e_)); } } while (0)
  SCHK(hipMalloc(&dA, sizeof(double) * Ac.size()));
  SCHK(hipMalloc(&dy, sizeof(double) * M));
  SCHK(hipMalloc(&dG, sizeof(double) * (size_t)Ntot * Ntot));
  SCHK(hipMalloc(&dAty, sizeof(double) * Ntot));
  SCHK(hipMalloc(&dysq, sizeof(double) * 2));
  SCHK(hipMalloc(&dbs, sizeof(double) * nlist));
  SCHK(hipMalloc(&dbt, sizeof(long) * nlist));
  SCHK(hipMalloc(&dw, sizeof(double) * MFX_GK));
  SCHK(hipMalloc(&dsub, sizeof(long) * MFX_GK));
  SCHK(hipMalloc(&dobj, sizeof(double)));
  SCHK(hipMalloc(&dyrec, sizeof(double) * M));
  SCHK(hipMalloc(&dk3, k3_buf_bytes(dicsizes[Kp - 1])));
  SCHK(hipMemcpy(dA, Ac.data(), sizeof(double) * Ac.size(), hipMemcpyHostToDevice));
  SCHK(hipMemcpy(dy, y, sizeof(double) * M, hipMemcpyHostToDevice));
  a.A = dA; a.y = dy; a.G = dG; a.Aty = dAty; a.ysq = dysq; a.blk_score = dbs; a.blk_tuple = dbt;
  a.w = dw; a.sub = dsub; a.minobj = dobj; a.yrec = dyrec;
  {
    K3Bufs kb = k3_bufs(dk3);
    if (int rc = launch_solver(a, k3 ? &kb : nullptr, nullptr)) { cleanup(); return rc; }
  }
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
  if (int rc = mfx_prof_begin(st)) return rc;
  hipLaunchKernelGGL(mfx_mc_partial_kernel, dim3((unsigned)nblocks), dim3(MFX_MC_THREADS), 0, st, a);
  HIPCHK(hipGetLastError());
  if (int rc = mfx_prof_end(st)) return rc;
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
