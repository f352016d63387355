// Micro-benchmark: does plain FP32 VALU work of ONE wave overlap with the v_mfma_f32_32x32x16_f16 chain of the OTHER
// wave on the same SIMD?  512-thread workgroups, one per CU (100 KB of LDS), waves 0-3 issue 39 dependent MFMAs per
// step, waves 4-7 issue V dependent-free v_fma_f32 per step (8 chains), a workgroup barrier ends every step.
//   hipcc -O3 --offload-arch=gfx950 pingpong_overlap.hip -o pingpong_overlap && ./pingpong_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, int MODE, bool AGPR = false, int OP = 0>   // OP: 0 v_fma_f32, 1 v_max_f32, 2 v_and_b32, 3 v_add_f32, 4 v_cvt_f16_f32, 5 plain v_fma_f32 (OP 0 compiles to v_pk_fma_f32 unless -fno-slp-vectorize); MODE 0: waves 0-3 MFMA, 4-7 VALU; 1: MFMA only (4-7 idle); 2: VALU only (0-3 idle); 3: roles swap every step
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
  f32x16 acc;
  for (int g = 0; g < 16; ++g) acc[g] = 0;
  float v[8];
  for (int q = 0; q < 8; ++q) v[q] = threadIdx.x * 0.5f + q;
  const float m = 1.0001f, c = 0.5f;
  lds[threadIdx.x] = 0.0f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const bool mf = MODE == 3 ? (((wave >> 2) ^ it) & 1) == 0 : wave < 4;
    if (MODE >= 4) {
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 39; ++i) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
          for (int q = i * V / 39; q < (i + 1) * V / 39; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(m), "v"(c));
        }
      } else if (MODE == 5) {
#pragma unroll
        for (int q = 0; q < 200; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(m), "v"(c));
      }
    } else if (mf) {
      if (MODE != 2) {
#pragma unroll
        for (int i = 0; i < 39; ++i) {
          if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));   // accumulator in AccVGPRs
          else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
      }
    } else {
      if (MODE != 1) {
#pragma unroll
        for (int q = 0; q < V; ++q) {
          float& x = v[q & 7];
          if constexpr (OP == 0) x = __builtin_fmaf(x, m, c);
          else if constexpr (OP == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(c));
          else if constexpr (OP == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(m));
          else if constexpr (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
          else if constexpr (OP == 4) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(x));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int g = 0; g < 16; ++g) s += acc[g];
  for (int q = 0; q < 8; ++q) s += v[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[(threadIdx.x * 7) & 511];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, int MODE, bool AGPR = false, int OP = 0>
static void run() {
  const int iters = 2000, grid = 256;
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * grid * 512);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * grid);
  auto kern = k<V, MODE, AGPR, OP>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 100 * 1024, 0, out, cyc, iters);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 100 * 1024, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  static unsigned long long h[256];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < grid; ++i) s += (double)h[i];
  const char* names[6] = {"MFMA waves 0-3 + VALU waves 4-7", "MFMA only", "VALU only", "roles swap every step", "V plain fma sliced into waves 0-3's own MFMAs", "same + 200 plain fma in waves 4-7"};
  printf("%s%sV=%3d %-32s: %.0f cycles per step (39 MFMA = 1248 cycles of matrix pipe)\n", AGPR ? "[AccVGPR accumulator] " : "", OP == 0 ? "" : OP == 1 ? "[v_max_f32] " : OP == 2 ? "[v_and_b32] " : OP == 3 ? "[v_add_f32] " : OP == 4 ? "[v_cvt_f16_f32] " : "[plain v_fma_f32] ", V, names[MODE], s / grid / iters);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  run<0, 1>();
  run<100, 2>(); run<200, 2>(); run<300, 2>();
  run<100, 0>(); run<200, 0>(); run<300, 0>(); run<400, 0>();
  run<100, 3>(); run<200, 3>(); run<300, 3>();
  run<200, 2, false, 1>(); run<200, 0, false, 1>(); run<200, 2, false, 2>(); run<200, 0, false, 2>();
  run<200, 2, false, 3>(); run<200, 0, false, 3>(); run<200, 2, false, 4>(); run<200, 0, false, 4>();
  run<200, 2, false, 5>(); run<100, 0, false, 5>(); run<200, 0, false, 5>(); run<300, 0, false, 5>(); run<400, 0, false, 5>(); run<300, 3, false, 5>();
  run<78, 4>(); run<156, 4>(); run<234, 4>(); run<312, 4>(); run<78, 5>(); run<156, 5>(); run<234, 5>();
  run<0, 1, true>(); run<100, 0, true>(); run<200, 0, true>(); run<300, 0, true>();
  return 0;
}
