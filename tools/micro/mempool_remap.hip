// mempool_remap.hip -- does scratch memory from HIP's DEFAULT stream-ordered pool stay coherent across calls?
//
// Background: repeated small mfx_fit_batch calls returned blocks of voxels fitted on stale data as
// long as the library's scratch came from the default pool, whose memory goes back to the driver at every stream
// synchronisation and is mapped again by the next call.  This is the pattern of those calls in isolation:
//   per "call": three stream-ordered allocations of changing sizes (so that addresses move between purposes);
//               a ONE-workgroup kernel fills buffer X with a call-specific pattern (one XCD writes, like mfx_extras_kernel);
//               a wide kernel (every CU of every XCD) fills the other buffers and checks X against the pattern;
//               stream-ordered frees; hipStreamSynchronize.
// Run with the default pool, with hipMalloc / hipFree per call and with a pool whose release threshold is unlimited;
// mismatches are counted per call.
//   hipcc -O2 --offload-arch=gfx950 tools/micro/mempool_remap.hip -o tools/micro/bin/mempool_remap && tools/micro/bin/mempool_remap
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void fill_one_wg(double* x, int n, double seed) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = seed + i;
}
__global__ void use_all(const double* x, int n, double seed, double* slab, size_t slab_per_wg, int* bad) {
  double* mine = slab + (size_t)blockIdx.x * slab_per_wg;
  for (size_t i = threadIdx.x; i < slab_per_wg; i += blockDim.x) mine[i] = seed + blockIdx.x;   // per-workgroup scratch, as the fit kernels
  __syncthreads();
  int wx = 0, ws = 0, wz = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { const double v = x[i]; wx += (v != seed + i); wz += (v != seed + i && v == 0.0); }
  for (size_t i = threadIdx.x; i < slab_per_wg; i += blockDim.x) { const double v = mine[i]; ws += (v != seed + blockIdx.x); wz += (v != seed + blockIdx.x && v == 0.0); }
  if (wx) atomicAdd(bad, wx);        // [0] wrong values of X (written by the previous kernel)
  if (ws) atomicAdd(bad + 1, ws);    // [1] wrong values of the workgroup's OWN slab, written a barrier earlier
  if (wz) atomicAdd(bad + 2, wz);    // [2] of both, the ones that read as zero
  if (wx || ws) atomicAdd(bad + 3, 1);   // [3] (threads of) workgroups affected
}

// mode 0: default pool, 1: `pool`, 2: hipMalloc / hipFree (synchronous)
static int run(int mode, hipMemPool_t pool, const char* what, int calls) {
  hipStream_t s;
  CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int* d_bad;
  CHK(hipMalloc(&d_bad, 4 * sizeof(int)));
  int bad_calls = 0;
  long tot[4] = {0, 0, 0, 0};
  for (int c = 0; c < calls; ++c) {
    const int n = 200 + 37 * (c % 5);                                   // "extra columns"
    const int wgs = 64 + 192 * (c % 3);                                 // "voxels" of the call
    const size_t slab = (size_t)(28000 + 4096 * (c % 4));               // doubles per workgroup
    double *x, *ws, *other;
    auto alloc = [&](void** p, size_t bytes) {
      if (mode == 1) CHK(hipMallocFromPoolAsync(p, bytes, pool, s)); else if (mode == 0) CHK(hipMallocAsync(p, bytes, s)); else CHK(hipMalloc(p, bytes));
    };
    alloc((void**)&x, sizeof(double) * n);
    alloc((void**)&other, sizeof(double) * 11 * (1 + c % 7));
    alloc((void**)&ws, sizeof(double) * slab * wgs);
    CHK(hipMemsetAsync(d_bad, 0, 4 * sizeof(int), s));
    const double seed = 1000.0 * (c + 1);
    hipLaunchKernelGGL(fill_one_wg, dim3(1), dim3(256), 0, s, x, n, seed);
    hipLaunchKernelGGL(use_all, dim3(wgs), dim3(512), 0, s, x, n, seed, ws, slab, d_bad);
    int h_bad[4] = {0, 0, 0, 0};
    CHK(hipMemcpyAsync(h_bad, d_bad, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    if (mode < 2) {
      CHK(hipFreeAsync(ws, s));
      CHK(hipFreeAsync(other, s));
      CHK(hipFreeAsync(x, s));
    }
    CHK(hipStreamSynchronize(s));
    if (mode == 2) { CHK(hipFree(ws)); CHK(hipFree(other)); CHK(hipFree(x)); }
    if (h_bad[0] || h_bad[1]) { ++bad_calls; for (int q = 0; q < 4; ++q) tot[q] += h_bad[q]; }
  }
  std::printf("%-46s %4d calls: %d calls with wrong data (wrong values: %ld of X, %ld of the workgroups' own slabs; %ld of them read as zero; %ld threads affected)\n",
              what, calls, bad_calls, tot[0], tot[1], tot[2], tot[3]);
  CHK(hipFree(d_bad));
  CHK(hipStreamDestroy(s));
  return bad_calls;
}

int main() {
  CHK(hipSetDevice(0));
  run(0, nullptr, "default pool (memory released at every sync)", 400);
  run(2, nullptr, "hipMalloc / hipFree in every call", 400);
  hipMemPoolProps props{};
  props.allocType = hipMemAllocationTypePinned;
  props.handleTypes = hipMemHandleTypeNone;
  props.location.type = hipMemLocationTypeDevice;
  props.location.id = 0;
  hipMemPool_t pool;
  CHK(hipMemPoolCreate(&pool, &props));
  uint64_t keep = UINT64_MAX;
  CHK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
  run(1, pool, "own pool, release threshold unlimited", 400);
  run(0, nullptr, "default pool again", 400);
  CHK(hipMemPoolDestroy(pool));
  return 0;
}
