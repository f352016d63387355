// Microbenchmark: FP64 MFMA vs FP64 VALU issue rates on gfx950, and whether they overlap.
// Build: hipcc -O3 --offload-arch=gfx950 f64_rates.hip -o f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: mfma only (NACC independent accumulators); MODE 1: valu fma only; MODE 2: both interleaved
template <int MODE, int NACC, int NV>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* cyc_out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double v[NV > 0 ? NV : 1];
  for (int i = 0; i < NV; ++i) v[i] = a0 * (i + 1) + threadIdx.x;
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (MODE == 0 || MODE == 2) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = __builtin_fma(v[j], b, a);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc_out[0] = t1 - t0;
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NV; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NACC, int NV>
int run(const char* name, int blocks, int threads, int iters) {
  double* out;
  CK(hipMalloc(&out, sizeof(double) * blocks * threads));
  unsigned long long* dcyc; CK(hipMalloc(&dcyc, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE, NACC, NV><<<blocks, threads>>>(out, dcyc, 10, 1.0, 0.5);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    k<MODE, NACC, NV><<<blocks, threads>>>(out, dcyc, iters, 1.0, 0.5);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double waves = (double)blocks * threads / 64;
  double nm = (MODE == 1) ? 0 : waves * iters * NACC;
  double nv = (MODE == 0) ? 0 : waves * iters * NACC * NV;
  double mf = nm * 2048.0, vf = nv * 128.0;
  // cycles per wave-instruction per SIMD assuming 2.4 GHz and even distribution
  double simds = 1024.0;
  double waves_per_simd = waves / simds;
  double cyc = best * 1e-3 * 2.4e9;  // per SIMD total cycles
  printf("%-40s blocks=%d thr=%d  %.3f ms  mfma %.1f TF  valu %.1f TF  total %.1f TF", name, blocks, threads, best,
         mf / best * 1e-9, vf / best * 1e-9, (mf + vf) / best * 1e-9);
  if (nm > 0) printf("  cyc/mfma/simd@2.4GHz=%.1f", cyc / (nm / simds));
  if (MODE == 1) printf("  cyc/fma/simd@2.4GHz=%.2f", cyc / (nv / simds));
  unsigned long long hc; CK(hipMemcpy(&hc, dcyc, 8, hipMemcpyDeviceToHost));
  printf("  (waves/simd=%.1f)  block0 cycles=%llu => clk~%.2f GHz, cyc/iter/wave=%.1f\n", waves_per_simd, hc, hc / (best * 1e-3) * 1e-9, (double)hc / iters);
  CK(hipFree(out));
  return 0;
}

int main() {
  int iters = 20000;
  run<0, 4, 0>("mfma only, 4 acc, 1 wave/simd", 256, 256, iters);
  run<0, 1, 0>("mfma only, 1 acc (dep chain), 1 w/simd", 256, 256, iters);
  run<0, 4, 0>("mfma only, 4 acc, 2 waves/simd", 512, 256, iters);
  run<0, 4, 0>("mfma only, 4 acc, 4 waves/simd", 1024, 256, iters);
  run<0, 4, 0>("mfma only, 4 acc, 8 waves/simd", 2048, 256, iters);
  run<0, 4, 0>("mfma only, 4 acc, 1 wave/simd 64thr x1024", 1024, 64, iters);
  run<0, 8, 0>("mfma only, 8 acc, 1 wave/simd", 256, 256, iters);
  run<1, 4, 4>("valu fma only, 16 indep, 1 wave/simd", 256, 256, iters);
  run<1, 4, 4>("valu fma only, 16 indep, 2 waves/simd", 512, 256, iters);
  run<1, 4, 4>("valu fma only, 16 indep, 4 waves/simd", 1024, 256, iters);
  run<2, 4, 2>("mfma + 2 fma each, 1 wave/simd", 256, 256, iters);
  run<2, 4, 4>("mfma + 4 fma each, 1 wave/simd", 256, 256, iters);
  run<2, 4, 8>("mfma + 8 fma each, 1 wave/simd", 256, 256, iters);
  run<2, 4, 16>("mfma + 16 fma each, 1 wave/simd", 256, 256, iters / 2);
  run<2, 4, 8>("mfma + 8 fma each, 2 waves/simd", 512, 256, iters);
  run<2, 4, 16>("mfma + 16 fma each, 2 waves/simd", 512, 256, iters / 2);
  return 0;
}
