// Micro-kernel: what v_mfma_f32_32x32x16_f16 does with its 17 addends (16 exact FP16 x FP16 products + the FP32
// accumulator) - the ASSUMPTION under the screening margin MFX_S_DC of fit_k2s.hip.
//
// For every test case the host knows the exact value of  c + sum_k a_k b_k  (x86 long double: the families keep all
// addends within 60 bits of each other, so the 64-bit significand holds the sum exactly) and its correctly rounded FP32
// value R.  Per family it reports
//   * how many results are NOT bit-identical to R (a fused 17-term add with ONE rounding would give 0),
//   * the largest |result - exact| in units of u = 2^-24 times  max(|c|, |a_k b_k|, |exact|)  ("kappa_max": 0.5 for a
//     single correctly rounded result; a chain of FP32 additions could reach ~16)  and times  |c| + sum |a_k b_k|
//     ("kappa_sum", the quantity the margin uses),
//   * for the accumulation CHAIN of the screening kernel (39 dependent MFMAs: hi.hi, hi.lo, lo.hi of 13 k-steps on
//     split FP32 operands) the same against the exact FP64 dot product of the FP32 operands, in cosine units.
// Rows of A carry 32 different test cases per instruction; all columns of B carry the same b.  Build + run:
//   hipcc -O3 --offload-arch=gfx950 -I../../microstructure_fingerprinting_amd/csrc mfma_sum_model.hip -o bin/mfma_sum_model && bin/mfma_sum_model
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one wave per block of 32 cases: A [32 x 16] halfs (row = case), b [16] halfs (shared by the 32 columns), c [32]
__global__ __launch_bounds__(64) void one_mfma(const _Float16* __restrict__ A, const _Float16* __restrict__ B, const float* __restrict__ C,
                                               float* __restrict__ D) {
  const int lane = threadIdx.x, lr = lane & 31, lh = lane >> 5;
  const size_t blk = blockIdx.x;
  h8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(blk * 32 + lr) * 16 + 8 * lh + j];
    b[j] = B[blk * 16 + 8 * lh + j];
  }
  f32x16 acc;
#pragma unroll
  for (int g = 0; g < 16; ++g) acc[g] = C[blk * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh];   // row of accumulator register g
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (lr == 0) {   // column 0
#pragma unroll
    for (int g = 0; g < 16; ++g) D[blk * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh] = acc[g];
  }
}

static float h2f(_Float16 h) { return (float)h; }

struct Fam {
  const char* name;
  long n = 0, notexact = 0;
  double kmax = 0, ksum = 0;
};

int main() {
  const int blocks = 1 << 15;              // 32 cases each: ~1e6 cases per family
  const size_t ncase = (size_t)blocks * 32;
  std::mt19937_64 rng(987654321);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::normal_distribution<double> Nrm(0.0, 1.0);
  std::vector<_Float16> A(ncase * 16), B((size_t)blocks * 16);
  std::vector<float> C(ncase), D(ncase);
  _Float16 *dA, *dB;
  float *dC, *dD;
  (void)hipMalloc(&dA, A.size() * 2); (void)hipMalloc(&dB, B.size() * 2); (void)hipMalloc(&dC, C.size() * 4); (void)hipMalloc(&dD, D.size() * 4);
  const double u = std::ldexp(1.0, -24);
  auto run = [&](const char* name) {
    (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one_mfma, dim3(blocks), dim3(64), 0, 0, dA, dB, dC, dD);
    (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    Fam f; f.name = name;
    for (size_t q = 0; q < ncase; ++q) {
      long double ex = (long double)C[q];
      double mx = std::fabs((double)C[q]), sm = std::fabs((double)C[q]);
      for (int k = 0; k < 16; ++k) {
        const double p = (double)h2f(A[q * 16 + k]) * (double)h2f(B[(q / 32) * 16 + k]);   // exact: 11 x 11 bits
        ex += (long double)p;
        mx = std::fmax(mx, std::fabs(p));
        sm += std::fabs(p);
      }
      const float R = (float)ex;    // one rounding of the exact sum
      if (std::memcmp(&R, &D[q], 4) != 0 && !(R == 0.0f && D[q] == 0.0f)) ++f.notexact;
      const double err = std::fabs((double)((long double)D[q] - ex));
      mx = std::fmax(mx, std::fabs((double)ex));
      if (mx > 0) f.kmax = std::fmax(f.kmax, err / (u * mx));
      if (sm > 0) f.ksum = std::fmax(f.ksum, err / (u * sm));
      ++f.n;
    }
    printf("%-72s cases %8ld   not correctly rounded %8ld   kappa_max %.3f   kappa_sum %.4f\n", f.name, f.n, f.notexact, f.kmax, f.ksum);
    return f;
  };
  auto H = [](double x) { return (_Float16)x; };
  double worst_sum = 0, worst_max = 0;
  auto note = [&](const Fam& f) { worst_sum = std::fmax(worst_sum, f.ksum); worst_max = std::fmax(worst_max, f.kmax); };

  // 1. the accumulator swamps the products: c = 2^24, all products 1 (sequential FP32 adds would lose every one of them)
  for (size_t q = 0; q < A.size(); ++q) A[q] = H(1.0);
  for (auto& v : B) v = H(1.0);
  for (auto& v : C) v = 16777216.0f;
  note(run("c = 2^24, sixteen products of 1 (exact: 2^24 + 16)"));
  // 2. one product swamps the others: 4096*4096 = 2^24 first / last, fifteen 1s, c = 0 (exact 2^24 + 15: a rounding tie)
  for (size_t q = 0; q < ncase; ++q)
    for (int k = 0; k < 16; ++k) A[q * 16 + k] = H(k == (int)(q % 16) ? 4096.0 : 1.0);
  for (size_t b = 0; b < (size_t)blocks; ++b)
    for (int k = 0; k < 16; ++k) B[b * 16 + k] = H(1.0);
  // (b is shared by the 32 cases of a block: put the large factor in a only: 4096 * 1 = 2^12; c carries the 2^24)
  for (auto& v : C) v = 16777216.0f + 0.0f;
  note(run("c = 2^24, one product 2^12 at position q mod 16, fifteen products of 1"));
  // 3. cancellation: +P, -P, and small addends that a limited-width aligner would drop (P = 2^20, smalls ~ 2^-6)
  for (size_t q = 0; q < ncase; ++q)
    for (int k = 0; k < 16; ++k) A[q * 16 + k] = H(k == 0 ? 1024.0 : (k == 1 ? -1024.0 : std::ldexp(1.0 + (double)((q * 16 + k) % 7) / 8.0, -6)));
  for (size_t b = 0; b < (size_t)blocks; ++b)
    for (int k = 0; k < 16; ++k) B[b * 16 + k] = H(k < 2 ? 1024.0 : 1.0);
  for (auto& v : C) v = 0.0f;
  note(run("+2^20, -2^20 and fourteen addends ~2^-6 (26 binades below), c = 0"));
  // 3b. the same with the small addends 2^-12 .. 2^-18 (32 .. 38 binades below the cancelling pair)
  for (size_t q = 0; q < ncase; ++q)
    for (int k = 0; k < 16; ++k) A[q * 16 + k] = H(k == 0 ? 1024.0 : (k == 1 ? -1024.0 : std::ldexp(1.0 + (double)((q + k) % 5) / 8.0, -12 - (int)(q % 7))));
  note(run("+2^20, -2^20 and fourteen addends 2^-12 .. 2^-18, c = 0"));
  // 4. dictionary-like: positive values at table scale (hi x hi of the kernel's split), c = a running positive sum
  for (size_t q = 0; q < ncase; ++q)
    for (int k = 0; k < 16; ++k) A[q * 16 + k] = H(128.0 * (0.05 + 0.95 * U(rng)));
  for (auto& v : B) v = H(128.0 * (0.05 + 0.95 * U(rng)));
  for (auto& v : C) v = (float)(128.0 * 128.0 * 16.0 * 12.0 * U(rng));
  note(run("positive, table scale (<= 128), c = positive running sum (hi.hi steps)"));
  // 5. the same with small signed b (hi x lo steps: products 2^-11 of the accumulator, either sign)
  for (auto& v : B) v = H(128.0 * std::ldexp(Nrm(rng), -11));
  note(run("positive a, signed b ~ 2^-11 (hi.lo steps), c = positive running sum"));
  // 6. Gaussian, random signs, c Gaussian
  for (size_t q = 0; q < A.size(); ++q) A[q] = H(32.0 * Nrm(rng));
  for (auto& v : B) v = H(32.0 * Nrm(rng));
  for (auto& v : C) v = (float)(4000.0 * Nrm(rng));
  note(run("Gaussian a, b, c (cancellation)"));
  // 7. log-uniform magnitudes over 2^-14 .. 2^6 per factor (products over 40 binades), random signs, c = 0
  for (size_t q = 0; q < A.size(); ++q) A[q] = H((U(rng) < 0.5 ? -1.0 : 1.0) * std::exp2(-14.0 + 20.0 * U(rng)));
  for (auto& v : B) v = H(std::exp2(-14.0 + 20.0 * U(rng)));
  for (auto& v : C) v = 0.0f;
  note(run("log-uniform factors over 20 binades each, random signs, c = 0"));
  // 8. all mantissa bits set: 2047/1024 * 2^e, products with 22 significant bits, positive, c with 24 significant bits
  for (size_t q = 0; q < A.size(); ++q) A[q] = H(std::ldexp(2047.0 / 1024.0, (int)(q % 5)));
  for (auto& v : B) v = H(2047.0 / 1024.0);
  for (size_t q = 0; q < ncase; ++q) C[q] = std::ldexp((float)(16777215 - (int)(q % 1000)), -18 + (int)(q % 9));
  note(run("22-bit products (all mantissa bits set), 24-bit c"));
  // 9-12. truncation probes: full-mantissa factors (2047/1024 and neighbours) with the exponents of the 16 products spread
  // over W binades, positive: after alignment to the largest addend every smaller one has low bits to lose
  for (int W : {2, 4, 8, 12, 24}) {
    for (size_t q = 0; q < A.size(); ++q) A[q] = H(std::ldexp((2047.0 - (double)(rng() % 4)) / 1024.0, -(int)(rng() % (unsigned)W)));
    for (auto& v : B) v = H((2047.0 - (double)(rng() % 4)) / 1024.0);
    for (size_t q = 0; q < ncase; ++q) C[q] = (q % 3 == 0) ? 0.0f : std::ldexp((float)(16777215 - (int)(rng() % 64)), -22 - (int)(rng() % (unsigned)W));
    char nm[96];
    snprintf(nm, sizeof nm, "full-mantissa positive addends spread over %d binades", W);
    note(run(nm));
  }
  // 13. the same, random signs
  for (size_t q = 0; q < A.size(); ++q) A[q] = H((rng() & 1 ? -1.0 : 1.0) * std::ldexp((2047.0 - (double)(rng() % 4)) / 1024.0, -(int)(rng() % 8u)));
  note(run("full-mantissa addends over 8 binades, random signs"));
  printf("worst over the families: kappa_max %.3f  kappa_sum %.4f   (|result - exact| <= kappa_sum * 2^-24 * (|c| + sum|a_k b_k|))\n", worst_max, worst_sum);
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dD);
  return 0;
}
