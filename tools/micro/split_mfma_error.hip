// Micro-kernel: worst-case error of the split-FP16 product the screening kernel ranks with (fit_k2s.hip),
//     c~ = sum over k-steps of  hi.hi + hi.lo + lo.hi   on v_mfma_f32_32x32x16_f16, ONE FP32 accumulator,
// against the exact (FP64) dot product of the same FP32 operands, in cosine units (divided by |a| |b|): the quantity
// MFX_S_DC bounds.  The split is the kernel's own mfx_split16 (included, not copied).  Operand families are chosen to
// be adversarial for each error source:
//   lo.lo dropped + residual r of the split (deterministic, <= 2^-19 sum|a b| at worst: mantissas with all low bits set)
//   FP32 accumulation inside and between the MFMAs (39 instructions, 624 products at K = 208): all-positive operands
//   (no cancellation: sum|a_i b_i| = sum a_i b_i, every partial sum as large as possible), partial sums straddling
//   powers of two, operands spanning the table's whole dynamic range down to FP16-subnormal low halves.
// Prints per family: max |c~ - c|, mean (a bias that grows with K would reveal truncating accumulation), and the
// largest error relative to sum|a b|.   Build + run:  hipcc -O3 --offload-arch=gfx950 -I../../microstructure_fingerprinting_amd/csrc
//                                                       split_mfma_error.hip -o bin/split_mfma_error && bin/split_mfma_error
#include "fit_k2s.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

// one wave per tile: A [32 x K], B [32 x K] FP32 row-major (B holds the columns as rows) -> acc [32 x 32] FP32
template <int KS>
__global__ __launch_bounds__(64) void split_tile(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C) {
  constexpr int K = KS * 16;
  const int lane = threadIdx.x, lr = lane & 31, lh = lane >> 5;
  const float* a = A + ((size_t)blockIdx.x * 32 + lr) * K;
  const float* b = B + ((size_t)blockIdx.x * 32 + lr) * K;
  f32x16 acc;
#pragma unroll
  for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    h8 ah, al, bh, bl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      _Float16 x, y;
      mfx_split16(a[16 * ks + 8 * lh + j], x, y); ah[j] = x; al[j] = y;
      mfx_split16(b[16 * ks + 8 * lh + j], x, y); bh[j] = x; bl[j] = y;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);   // the kernel's order
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
  }
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int row = (g & 3) + 8 * (g >> 2) + 4 * lh;
    C[((size_t)blockIdx.x * 32 + row) * 32 + lr] = acc[g];
  }
}

struct Stat { double maxabs = 0, sum = 0, maxrel = 0; long n = 0; };

template <int KS>
static Stat run_family(const char* name, std::vector<float>& A, std::vector<float>& B, int tiles) {
  constexpr int K = KS * 16;
  float *dA, *dB, *dC;
  (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, (size_t)tiles * 1024 * 4);
  (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(split_tile<KS>, dim3(tiles), dim3(64), 0, 0, dA, dB, dC);
  std::vector<float> C((size_t)tiles * 1024);
  (void)hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  Stat s;
  for (int t = 0; t < tiles; ++t)
    for (int i = 0; i < 32; ++i) {
      const float* a = &A[((size_t)t * 32 + i) * K];
      double na = 0;
      for (int k = 0; k < K; ++k) na += (double)a[k] * a[k];
      for (int j = 0; j < 32; ++j) {
        const float* b = &B[((size_t)t * 32 + j) * K];
        double nb = 0, dot = 0, sab = 0;
        for (int k = 0; k < K; ++k) { nb += (double)b[k] * b[k]; dot += (double)a[k] * b[k]; sab += std::fabs((double)a[k] * b[k]); }
        const double nrm = std::sqrt(na) * std::sqrt(nb);
        if (!(nrm > 0)) continue;
        const double err = ((double)C[((size_t)t * 32 + i) * 32 + j] - dot) / nrm;
        s.maxabs = std::fmax(s.maxabs, std::fabs(err));
        s.maxrel = std::fmax(s.maxrel, std::fabs(err) * nrm / sab);
        s.sum += err;
        ++s.n;
      }
    }
  printf("K=%3d  %-58s pairs %8ld   max|c~-c| %.3e   mean %+.3e   max err/sum|ab| %.3e\n", K, name, s.n, s.maxabs, s.sum / s.n, s.maxrel);
  return s;
}

static float lowbits(float f, unsigned mask) {   // set the mantissa bits below the FP16 'hi' cut
  unsigned u;
  std::memcpy(&u, &f, 4);
  u |= mask;
  std::memcpy(&f, &u, 4);
  return f;
}

template <int KS>
static double sweep(int tiles) {
  constexpr int K = KS * 16;
  std::mt19937_64 rng(12345 + KS);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::normal_distribution<double> Nrm(0.0, 1.0);
  const size_t n = (size_t)tiles * 32 * K;
  std::vector<float> A(n), B(n);
  double worst = 0;
  auto fill = [&](auto gen) { for (size_t q = 0; q < n; ++q) { A[q] = gen((int)(q % K), 0); B[q] = gen((int)(q % K), 1); } };
  // 1. dictionary-like: smooth positive decays, table scale (values up to 128), near-collinear pairs (c ~ 0.9 .. 1)
  {
    std::vector<double> pa((size_t)tiles * 32 * 2), pb((size_t)tiles * 32 * 2);
    for (auto& v : pa) v = U(rng);
    for (auto& v : pb) v = U(rng);
    for (size_t q = 0; q < n; ++q) {
      const size_t v = q / K; const int k = (int)(q % K);
      A[q] = (float)(128.0 * (0.3 + 0.7 * pa[2 * v]) * std::exp(-3.0 * pa[2 * v + 1] * k / K));
      B[q] = (float)(128.0 * (0.3 + 0.7 * pb[2 * v]) * std::exp(-3.0 * pb[2 * v + 1] * k / K));
    }
    worst = std::fmax(worst, run_family<KS>("smooth positive decays (dictionary-like)", A, B, tiles).maxabs);
  }
  // 2. all-positive uniform in (64, 128]: every product and every partial sum as large as it can be
  fill([&](int, int) { return (float)(64.0 + 64.0 * U(rng)); });
  worst = std::fmax(worst, run_family<KS>("all-positive, uniform in (64,128]", A, B, tiles).maxabs);
  // 3. the same with all mantissa bits below the hi cut set (largest lo halves, largest lo.lo term dropped)
  fill([&](int, int) { return lowbits((float)(64.0 + 64.0 * U(rng)), 0x1fffu); });
  worst = std::fmax(worst, run_family<KS>("all-positive, low mantissa bits all ones (max lo.lo)", A, B, tiles).maxabs);
  // 4. low bits 0x1000 pattern: lo halves that are exact ties for the FP16 rounding of lo
  fill([&](int, int) { return lowbits((float)(64.0 + 64.0 * U(rng)), 0x1001u); });
  worst = std::fmax(worst, run_family<KS>("all-positive, lo halves at FP16 rounding ties", A, B, tiles).maxabs);
  // 5. whole dynamic range: log-uniform magnitudes 128 * 2^-[0, 24] (low halves go FP16-subnormal below ~0.06), positive
  fill([&](int, int) { return (float)(128.0 * std::exp2(-24.0 * U(rng))); });
  worst = std::fmax(worst, run_family<KS>("positive, log-uniform over 24 binades (subnormal lo halves)", A, B, tiles).maxabs);
  // 6. one dominant row + tiny rest (a b0-like measurement): accumulation around one big term
  fill([&](int k, int) { return (float)(k == 0 ? 128.0 : 0.01 * U(rng)); });
  worst = std::fmax(worst, run_family<KS>("one dominant measurement + small rest", A, B, tiles).maxabs);
  // 7. random signs, Gaussian (heavy cancellation: sum|ab| >> |sum ab|)
  fill([&](int, int) { return (float)(32.0 * Nrm(rng)); });
  worst = std::fmax(worst, run_family<KS>("Gaussian, random signs (cancellation)", A, B, tiles).maxabs);
  // 8. identical vectors (c = 1 exactly), positive
  fill([&](int, int) { return 0.0f; });
  for (size_t q = 0; q < n; ++q) A[q] = B[q] = (float)(128.0 * U(rng));
  worst = std::fmax(worst, run_family<KS>("a == b (c = 1), uniform positive", A, B, tiles).maxabs);
  // 9. constant vectors 2^e * (1 + 2^-11): partial sums run through exact powers of two
  fill([&](int, int) { return 64.0f * (1.0f + 0x1p-11f); });
  worst = std::fmax(worst, run_family<KS>("constant 64 (1 + 2^-11): sums straddle powers of two", A, B, tiles).maxabs);
  // 10. increasing ramps: small terms first, large last (and the reverse) - order sensitivity of the accumulation
  fill([&](int k, int) { return (float)(128.0 * (k + 1) / K); });
  worst = std::fmax(worst, run_family<KS>("increasing ramp", A, B, tiles).maxabs);
  fill([&](int k, int) { return (float)(128.0 * (K - k) / K); });
  worst = std::fmax(worst, run_family<KS>("decreasing ramp", A, B, tiles).maxabs);
  // 12. the [N, N, 1] form of the screening kernel: dictionary-like vectors whose LAST row carries -u1 / u2, u = 0.8 |d|
  // (the projected cross product d1.d2 - u1 u2 out of the same MFMAs: one product of the size of all others together,
  // with the opposite sign).  The error is relative to the norms of the AUGMENTED vectors here: x 1.64 for |d1||d2|.
  {
    std::vector<double> pa((size_t)tiles * 32 * 2), pb((size_t)tiles * 32 * 2);
    for (auto& v : pa) v = U(rng);
    for (auto& v : pb) v = U(rng);
    for (size_t v = 0; v < (size_t)tiles * 32; ++v) {
      double na = 0, nb = 0;
      for (int k = 0; k + 1 < K; ++k) {
        const double x = 128.0 * (0.3 + 0.7 * pa[2 * v]) * std::exp(-3.0 * pa[2 * v + 1] * k / K);
        const double y = 128.0 * (0.3 + 0.7 * pb[2 * v]) * std::exp(-3.0 * pb[2 * v + 1] * k / K);
        A[v * K + k] = (float)x; B[v * K + k] = (float)y;
        na += (double)(float)x * (float)x; nb += (double)(float)y * (float)y;
      }
      A[v * K + K - 1] = (float)(-0.8 * std::sqrt(na));
      B[v * K + K - 1] = (float)(0.8 * std::sqrt(nb));
    }
    worst = std::fmax(worst, run_family<KS>("dictionary-like + spare row (-u1, u2), u = 0.8 |d|", A, B, tiles).maxabs);
  }
  return worst;
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 1024;   // 1024 tiles x 1024 pairs ~ 1e6 pairs per family
  double w = 0;
  w = std::fmax(w, sweep<4>(tiles));
  w = std::fmax(w, sweep<8>(tiles));
  w = std::fmax(w, sweep<13>(tiles));
  w = std::fmax(w, sweep<16>(tiles));
  printf("worst |c~ - c| over all families and lengths: %.3e  (MFX_S_DC = %.1e: factor %.1f)\n", w, (double)MFX_S_DC, MFX_S_DC / w);
  return 0;
}
