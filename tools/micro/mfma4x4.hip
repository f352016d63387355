// Microbenchmark 2: v_mfma_f64_4x4x4 (4 blocks) rate vs 16x16x4, plus MFMA16 + VALU split across waves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k4(double* out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// even waves: MFMA 16x16x4 chain; odd waves: VALU fma chain (NV independent) -- tests cross-wave overlap
template <int NV>
__global__ __launch_bounds__(512) void ksplit(double* out, int iters_m, int iters_v, double a0, double b0) {
  const int wave = threadIdx.x >> 6;
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  double s = 0;
  if (wave & 1) {
    double v[NV];
    for (int i = 0; i < NV; ++i) v[i] = a * (i + 1);
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] = __builtin_fma(v[j], b, a);
    }
    for (int i = 0; i < NV; ++i) s += v[i];
  } else {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters_m; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 4; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}

int main() {
  double* out; CK(hipMalloc(&out, 8 * 4096 * 512));
  int iters = 20000;
  for (int blocks : {256, 512, 1024}) {
    float ms = timeit([&] { k4<8><<<blocks, 256>>>(out, iters, 1.0, 0.5); });
    double n = (double)blocks * 4 * iters * 8;
    printf("mfma 4x4x4(4blk) 8acc blocks=%d: %.3f ms  %.1f TF  cyc/mfma/simd@2.4=%.1f\n", blocks, ms, n * 512 / ms * 1e-9, ms * 1e-3 * 2.4e9 / (n / 1024));
  }
  // split: 1 block/CU of 512 threads: each SIMD has 1 MFMA wave + 1 VALU wave
  for (int iv : {0, 20000, 40000, 80000}) {
    float ms = timeit([&] { ksplit<16><<<256, 512>>>(out, iters, iv, 1.0, 0.5); });
    double nm = 256.0 * 4 * iters * 4, nv = 256.0 * 4 * iv * 16;
    printf("split 1 mfma-wave + 1 valu-wave per SIMD, valu iters=%d: %.3f ms  mfma %.1f TF + valu %.1f TF = %.1f TF\n", iv, ms, nm * 2048 / ms * 1e-9, nv * 128 / ms * 1e-9, (nm * 2048 + nv * 128) / ms * 1e-9);
  }
  // 2 blocks/CU: 2 mfma waves + 2 valu waves per SIMD
  for (int iv : {0, 40000, 80000}) {
    float ms = timeit([&] { ksplit<16><<<512, 512>>>(out, iters, iv, 1.0, 0.5); });
    double nm = 512.0 * 4 * iters * 4, nv = 512.0 * 4 * iv * 16;
    printf("split 2+2 waves per SIMD, valu iters=%d: %.3f ms  mfma %.1f TF + valu %.1f TF = %.1f TF\n", iv, ms, nm * 2048 / ms * 1e-9, nv * 128 / ms * 1e-9, (nm * 2048 + nv * 128) / ms * 1e-9);
  }
  return 0;
}
