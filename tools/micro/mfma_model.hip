// Microbenchmark 3: issue model of v_mfma_f64_16x16x4 with VGPR accumulators.
//  - chains per wave (1,2,4), waves per SIMD (1,2,3,4)
//  - own-wave VALU (f64 fma / b32 ops) interleaved between MFMAs: does it hide in the MFMA shadow?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NCH, int NV, int NI>
__global__ void k(double* out, int iters, double a0, double b0) {
  d4 acc[NCH];
  for (int i = 0; i < NCH; ++i) acc[i] = d4{0, 0, 0, 0};
  double v[NV > 0 ? NV : 1];
  int iv[NI > 0 ? NI : 1];
  for (int i = 0; i < NV; ++i) v[i] = a0 * (i + 1) + threadIdx.x;
  for (int i = 0; i < NI; ++i) iv[i] = threadIdx.x + i;
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] = __builtin_fma(v[j], b, a);
#pragma unroll
      for (int j = 0; j < NI; ++j) iv[j] = iv[j] * 3 + it;
    }
  }
  double s = 0;
  for (int i = 0; i < NCH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NV; ++i) s += v[i];
  for (int i = 0; i < NI; ++i) s += iv[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NCH, int NV, int NI>
void run(const char* name, int waves_per_simd, int iters) {
  // 256-thread blocks; waves_per_simd blocks per CU
  int blocks = 256 * waves_per_simd;
  double* out; hipMalloc(&out, 8 * (size_t)blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NCH, NV, NI><<<blocks, 256>>>(out, 10, 1.0, 0.5); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) { hipEventRecord(e0); k<NCH, NV, NI><<<blocks, 256>>>(out, iters, 1.0, 0.5); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  double nm = (double)blocks * 4 * iters * NCH;   // MFMAs
  double cyc_per_mfma_simd = best * 1e-3 * 2.38e9 / (nm / 1024);
  printf("%-52s w/simd=%d  %.3f ms  %.1f TF  cyc/MFMA/SIMD=%.1f  cyc/MFMA/wave=%.1f\n", name, waves_per_simd, best, nm * 2048 / best * 1e-9, cyc_per_mfma_simd, cyc_per_mfma_simd * waves_per_simd);
  hipFree(out);
}

int main() {
  int it = 20000;
  printf("--- chains per wave x waves per SIMD (no VALU)\n");
  for (int w : {1, 2, 3, 4}) { run<1, 0, 0>("1 chain", w, it); }
  for (int w : {1, 2, 3, 4}) { run<2, 0, 0>("2 chains", w, it); }
  for (int w : {1, 2, 3, 4}) { run<4, 0, 0>("4 chains", w, it); }
  printf("--- own-wave f64 FMA between MFMAs (2 chains)\n");
  for (int w : {1, 2, 3}) { run<2, 2, 0>("2 chains + 2 fma/MFMA", w, it); }
  for (int w : {1, 2, 3}) { run<2, 4, 0>("2 chains + 4 fma/MFMA", w, it); }
  for (int w : {1, 2, 3}) { run<2, 8, 0>("2 chains + 8 fma/MFMA", w, it); }
  for (int w : {1, 2, 3}) { run<2, 16, 0>("2 chains + 16 fma/MFMA", w, it / 2); }
  printf("--- own-wave int ops between MFMAs (2 chains)\n");
  for (int w : {1, 2, 3}) { run<2, 0, 4>("2 chains + 4 imul-add/MFMA", w, it); }
  for (int w : {1, 2, 3}) { run<2, 0, 8>("2 chains + 8 imul-add/MFMA", w, it); }
  for (int w : {1, 2, 3}) { run<2, 0, 16>("2 chains + 16 imul-add/MFMA", w, it / 2); }
  return 0;
}
