// mfx_host.h -- host-side plumbing shared by the translation units of libmfx.so.
//
// The library is built from several translation units so that the kernel families compile in parallel
// (mfx_api.hip: C ABI, tables/plans, small kernels; tu_k2.hip: FP64 two-fascicle kernel; tu_k2s_*.hip: the
// screening kernel's instantiations; tu_k2x.hip: two fascicles + CSF/EAR).  Everything mutable that is not owned
// by a handle lives in ONE thread-local record: one host thread drives one GPU (mf.py:_fit_sharded), so the error
// string, the timing events and the diagnostic switches of a thread never meet those of another.
#pragma once
#include "../../include/mfx.h"

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "fit_k2.hip"
#include "fit_k2x.hip"

#define MFX_NCOUNTERS 12   // per-call counters (mfx_debug_last_counter): [0..7] hand-backs of the kernel families, [8..10] screening audit

struct MfxThread {
  std::string err;
  // timing hook (mfx_set_profiling / mfx_last_kernel_ms)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int ev_launches = 0;
  bool ev_valid = false;
  bool profiling = false;
  // diagnostics (mfx_debug_*)
  unsigned long long* stamps = nullptr;
  int k2_pipe = -1;        // MFX_K2_PIPE=0 selects the un-pipelined chunk loop of the FP64 kernel
  int k2_maxc = MFX_MAXC;  // FP64 kernel: short-list size beyond which its exhaustive exact pass runs
  int k2x_maxc = MFX_XMAXC;
  int k2s_nb = 0;          // 0: as many chunk images as fit; 2: force the two-image schedule
  int k2s_cap = 0;         // 0: MFX_S_CAP
  int force_generic = 0;   // 1: every voxel class through the explicit-dictionary solver (tests of the out-of-limits fallback)
  int k3_screen = 1;       // 0: three sub-dictionaries through the plain one-thread-per-tuple scan (no batched path, no relaxed-bound screen: the referee of the full-size test)
  int k3_cap = 0;          // > 0: candidate-list entries of the batched three-fascicle path (tests force the overflow fallback)
  int k3_batch = -1;       // MFX_K3_BATCH=0: three-fascicle voxels one by one through solve_k3.hip (no batched fit_k3.hip path)
  int k2_screen = -1;      // MFX_K2_SCREEN=0 disables the screening kernels
  int k2x_screen = -1;     // MFX_K2X_SCREEN=0: the [N, N, 1] class stays on the FP64 kernel (no screening pipeline)
  int k2_wide = -1;        // MFX_K2_WIDE: 1 forces the wide screening kernel (fit_k2w.hip) wherever it applies, 0 never uses it
  // hand-back counters of the last mfx_fit_batch* call: summed on the device over its launches, copied to pinned memory
  // behind the kernels and read only when somebody asks (mfx_debug_last_*_count): the _dev entry points never synchronise
  int* fb_dev = nullptr;           // device [MFX_NCOUNTERS]: [0] voxels handed back to an exact kernel, [1] of them by the screening-error guard
  int* fb_host = nullptr;          // pinned [MFX_NCOUNTERS]
  int fb_device = -1;              // device fb_dev lives on
  hipEvent_t fb_event = nullptr;
  bool fb_pending = false;
  // host pipeline of mfx_fit_batch (pinned staging, copy/compute streams), created on first use per device
  void* stage[2] = {nullptr, nullptr};
  size_t stage_bytes = 0;
  hipStream_t s_copy = nullptr, s_comp = nullptr;
  hipEvent_t ev_h2d[2] = {nullptr, nullptr};
  int pipe_device = -1;
  // device buffers of mfx_fit_batch (signals, directions, parameters, voxel lists): kept between calls and only ever
  // grown, so that a volume fitted slab by slab does not pay an allocation and a (synchronising) release per call;
  // mfx_thread_release() returns them
  void* pool[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [4], [5]: a volume in its file's layout and the ROI's indices into it
  size_t pool_bytes[6] = {0, 0, 0, 0, 0, 0};
  // two internal streams of the voxel-by-voxel three-fascicle path (two voxels in flight: the launch gaps and the
  // tail of one voxel's kernels are covered by the other's), forked from and joined to the caller's stream by events
  // scratch arenas (mfx_scratch_alloc): one per (device, stream) this thread has launched on
  struct Arena {
    int device = -1;
    hipStream_t stream = nullptr;
    std::vector<void*> blocks;
    std::vector<size_t> sizes;
    size_t cur = 0, off = 0;   // block in use, bytes handed out of it
    int live = 0;              // allocations not yet released
    unsigned long long used = 0;   // (LRU stamp)
  };
  std::vector<Arena> arenas;
  unsigned long long arena_clock = 0;
  hipStream_t s_lane[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_lane[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // [lane] join, [4] fork
  int lane_device = -1;
};

MfxThread& mfx_thread();
int mfx_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) return mfx_fail(MFX_ERR_HIP, "%s failed: %s", #x, hipGetErrorString(e_)); \
  } while (0)

// device allocation released on every exit path
struct DevMem {
  void* p = nullptr;
  DevMem() = default;
  DevMem(const DevMem&) = delete;
  DevMem& operator=(const DevMem&) = delete;
  ~DevMem() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  void* release() { void* q = p; p = nullptr; return q; }
  template <class T> T* as() const { return (T*)p; }
};
// Scratch memory of a call (per-workgroup slabs, short lists, extra columns, Gram / candidate buffers) comes from an ARENA
// per (host thread, stream): a few device blocks (a new one whenever a request does not fit: hipMalloc, once), handed out
// by bumping an offset through them in order, and taken back when the call's last StreamMem dies.  Calls on one stream execute in order, so the next call may reuse the
// addresses while the previous one is still running; different streams and different host threads have their own arenas.
// No HIP allocator call in the steady state: on the MI355X boxes of round 2 memory of HIP's DEFAULT stream-ordered pool
// read back as zeros after being re-mapped (tools/micro/mempool_remap.hip), and hipFreeAsync - also into a
// pool that keeps its memory - blocked the host for a kernel's duration from the second call in flight on.
// MFX_POISON=<byte>: every scratch allocation is filled with that byte before use (developer check for reads of
// uninitialised scratch memory).
hipError_t mfx_scratch_alloc(void** p, size_t bytes, hipStream_t s);
void mfx_scratch_free(void* p, hipStream_t s);
// scratch allocation released on every exit path
struct StreamMem {
  void* p = nullptr;
  hipStream_t s;
  explicit StreamMem(hipStream_t s_) : s(s_) {}
  StreamMem(const StreamMem&) = delete;
  StreamMem& operator=(const StreamMem&) = delete;
  ~StreamMem() { if (p) mfx_scratch_free(p, s); }
  hipError_t alloc(size_t bytes) { return mfx_scratch_alloc(&p, bytes, s); }
  template <class T> T* as() const { return (T*)p; }
};

// event pair around the dominant kernel of a class launch (bench.py reads it through mfx_last_kernel_ms)
int mfx_prof_begin(hipStream_t st);
int mfx_prof_end(hipStream_t st);
// hand-back counters: zero the call's totals / add one launch's device counters [n <= 4 ints] / queue their copy to
// pinned memory, all in stream order behind the work on `st` (no host synchronisation)
int mfx_fb_begin(hipStream_t st);
int mfx_fb_accumulate(const int* d_counters, int n, hipStream_t st, int offset = 0);
int mfx_fb_accumulate_audit(const int* d_audit, hipStream_t st);   // [0] beyond DC/4 (+), [1] largest error (max), [2] audited pairs (+) -> counters 8..10
int mfx_fb_end(hipStream_t st);

// ---- kernel launchers, one translation unit each
// FP64 two-fascicle kernel (tu_k2.hip).  With a.list_count set, a.vox_list is a device-side list whose length only the
// device knows: the launch covers nvox blocks and those beyond *a.list_count exit at once.
int mfx_launch_k2_f64(const FitK2Args& a, int nvox, hipStream_t st, bool record_events);
bool mfx_k2_f64_fits(const FitK2Args& a);
// screening kernel (tu_k2s_*.hip): KS k-steps of 16 measurements, bracketed protocol or not, NB chunk images
size_t mfx_k2s_lds_bytes(int KS, int N, bool bracket, int NB);
int mfx_launch_k2s_ks4(const FitK2Args& a, int nvox, hipStream_t st, bool br, int NB);
int mfx_launch_k2s_ks8(const FitK2Args& a, int nvox, hipStream_t st, bool br, int NB);
int mfx_launch_k2s_ks13(const FitK2Args& a, int nvox, hipStream_t st, bool br, int NB);
int mfx_launch_k2s_ks16(const FitK2Args& a, int nvox, hipStream_t st, bool br, int NB);
// wide screening kernel (tu_k2w_*.hip): one wave per SIMD, TL row tiles per wave (fit_k2w.hip)
size_t mfx_k2sx_lds_bytes(int KS, int N, bool bracket, int NB);
int mfx_launch_k2sx_ks8(const FitK2Args& a, int nvox, hipStream_t st, bool br);
int mfx_launch_k2sx_ks13(const FitK2Args& a, int nvox, hipStream_t st, bool br);
size_t mfx_k2w_lds_bytes(int KS, int N, bool bracket, int NB, int TL);
size_t mfx_k2wx_lds_bytes(int KS, int N, bool bracket, int NB, int TL);
int mfx_launch_k2wx_ks24(const FitK2Args& a, int nvox, hipStream_t st, bool br);
int mfx_launch_k2wx_ks35(const FitK2Args& a, int nvox, hipStream_t st, bool br);
int mfx_launch_k2w_ks13(const FitK2Args& a, int nvox, hipStream_t st, bool br);
int mfx_launch_k2w_ks16(const FitK2Args& a, int nvox, hipStream_t st, bool br);
int mfx_launch_k2w_ks24(const FitK2Args& a, int nvox, hipStream_t st, bool br);
int mfx_launch_k2w_ks35(const FitK2Args& a, int nvox, hipStream_t st, bool br);
// two fascicles + CSF/EAR (tu_k2x.hip)
int mfx_launch_k2x(const FitK2XArgs& a, int nvox, hipStream_t st);
