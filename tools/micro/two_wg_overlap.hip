// Micro-benchmark: two INDEPENDENT 256-thread workgroups per CU (one wave of each per SIMD, no synchronisation between
// them) instead of the ping-pong of one 512-thread workgroup: every wave alternates 39 dependent
// v_mfma_f32_32x32x16_f16 and V plain v_fma_f32, a workgroup barrier (4 waves) per step.  Cycles per step and per wave;
// the SIMD executes two such steps (2 x 1248 cycles of matrix pipe) in that time if the waves interleave.
//   hipcc -O3 --offload-arch=gfx950 two_wg_overlap.hip -o two_wg_overlap && ./two_wg_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, int NACC>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ float lds[];
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int g = 0; g < 16; ++g) acc[n][g] = 0;
  float v[8];
  for (int q = 0; q < 8; ++q) v[q] = threadIdx.x * 0.5f + q;
  const float m = 1.0001f, c = 0.5f;
  lds[threadIdx.x] = 0.0f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
#pragma unroll
      for (int i = 0; i < 39; ++i) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < V * NACC; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(m), "v"(c));
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int n = 0; n < NACC; ++n) for (int g = 0; g < 16; ++g) s += acc[n][g];
  for (int q = 0; q < 8; ++q) s += v[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[(threadIdx.x * 7) & 255];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, int NACC>
static void run() {
  const int iters = 1000, grid = 512;   // 2 workgroups per CU, 70 KB of LDS each
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * grid * 256);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * grid);
  auto kern = k<V, NACC>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 70 * 1024, 0, out, cyc, iters);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 70 * 1024, 0, out, cyc, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long h[512];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < grid; ++i) s += (double)h[i];
  const double per = s / grid / iters;
  printf("tiles per wave %d, V=%3d per tile: %.0f cycles per step and wave = %.0f per 39 MFMAs and SIMD (pipe: 1248); wall clock %.2f ms\n",
         NACC, V, per, per / (2 * NACC), ms);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  run<0, 1>(); run<100, 1>(); run<200, 1>(); run<300, 1>(); run<400, 1>();
  run<150, 3>(); run<250, 3>();
  return 0;
}
