// Micro-benchmark: how many plain FP32 VALU instructions of the SAME wave hide behind v_mfma_f32_32x32x16_f16
// when one wave per SIMD runs (256-thread workgroup, 1 per CU)?  Region = 6 MFMAs with V v_fma_f32 (8 independent chains) sliced evenly between them.
//   hipcc -O3 --offload-arch=gfx950 mfma16_valu.hip -o mfma16_valu && ./mfma16_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void k(float* out, unsigned long long* cyc, int iters) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
  f32x16 acc0, acc1;
  for (int g = 0; g < 16; ++g) { acc0[g] = 0; acc1[g] = 0; }
  float v[8];
  for (int q = 0; q < 8; ++q) v[q] = threadIdx.x * 0.5f + q;
  const float m = 1.0001f, c = 0.5f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define VSLICE(i) _Pragma("unroll") for (int q = (i) * V / 6; q < ((i) + 1) * V / 6; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], m, c); __builtin_amdgcn_sched_barrier(0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0); VSLICE(0)
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0); VSLICE(1)
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0); VSLICE(2)
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0); VSLICE(3)
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0); VSLICE(4)
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0); VSLICE(5)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int g = 0; g < 16; ++g) s += acc0[g] + acc1[g];
  for (int q = 0; q < 8; ++q) s += v[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, int WPS>
static void run(const char* tag) {
  const int iters = 2000, grid = 256;
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * grid * 512);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * grid);
  hipLaunchKernelGGL((k<V, WPS>), dim3(grid), dim3(256 * WPS), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k<V, WPS>), dim3(grid), dim3(256 * WPS), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[256];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < grid; ++i) s += (double)h[i];
  printf("%s V=%2d waves/SIMD=%d: %.1f cycles per region (6 MFMA = 192 cycles of matrix pipe per wave)\n", tag, V, WPS, s / grid / iters);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  run<0, 1>("1w"); run<12, 1>("1w"); run<24, 1>("1w"); run<36, 1>("1w"); run<48, 1>("1w"); run<60, 1>("1w"); run<72, 1>("1w"); run<96, 1>("1w");
  run<0, 2>("2w"); run<24, 2>("2w"); run<48, 2>("2w"); run<72, 2>("2w");
  return 0;
}
