// Micro-benchmark: v_mfma_f32_32x32x16_f16 issue rate of ONE wave as a function of the number of independent
// accumulators it rotates through (1 = a dependent chain), with 1 or 2 waves per SIMD; cycles from s_memtime and the
// flop rate from the wall clock (hipEvents), so the two clocks can be compared.
//   hipcc -O3 --offload-arch=gfx950 mfma16_chain.hip -o mfma16_chain && ./mfma16_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void k(float* out, unsigned long long* cyc, int iters) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int g = 0; g < 16; ++g) acc[n][g] = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i % NACC], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int n = 0; n < NACC; ++n) for (int g = 0; g < 16; ++g) s += acc[n][g];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int WPS>
static void run() {
  const int iters = 4000, grid = 1024;   // 4 workgroups per CU in sequence
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * grid * 512);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * grid);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, WPS>), dim3(grid), dim3(256 * WPS), 0, 0, out, cyc, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NACC, WPS>), dim3(grid), dim3(256 * WPS), 0, 0, out, cyc, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < grid; ++i) s += (double)h[i];
  const double flops = (double)grid * 4 * WPS * iters * 12 * 32768.0;
  printf("accumulators=%d waves/SIMD=%d: %.1f s_memtime ticks per MFMA per wave, wall clock %.1f TFLOP/s\n", NACC, WPS, s / grid / iters / 12, flops / (ms * 1e-3) / 1e12);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  run<1, 1>(); run<2, 1>(); run<4, 1>();
  run<1, 2>(); run<2, 2>(); run<4, 2>();
  return 0;
}
