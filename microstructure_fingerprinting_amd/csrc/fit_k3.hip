// fit_k3.hip -- three-fascicle voxels (BASELINE config 5: solve_exhaustive_posweights_3, mf_utils.py:470-607, on three
// rotated dictionaries of N atoms; 1500 atoms x 300 measurements = 3.4e9 index triples per voxel) in BATCHES of voxels.
//
// Same mathematics as solve_k3.hip (which stays the path of mfx_solve_exhaustive and of the classes with extra columns):
// the FP64 Gram only RANKS, every triple goes through a relaxed-bound test (third atom unconstrained: a two-atom problem in
// the orthogonal complement of d3, angle-sum test of fit_k2s.hip), what passes is scored from the Gram, what reaches the
// running threshold is listed, and the list is decided in the reference's arithmetic and scan order (Cramer test with its
// tolerance, explicit residual, first hit in i3 -> i1 -> i2 order).  What is new:
//   * the TEST RUNS ON THE MATRIX PIPE.  For a fixed third atom i3 the test value of the pair (i1, i2) is BILINEAR in
//     per-(atom, i3) constants:   b = pn1 pn2 - qn1 qn2 + u1 u2 - a12,   pn = (P + D)|d'|, qn = (1 - D) Q |d'|, u = d.d3/|d3|
//     (solve_k3.hip computes exactly this with 11 vector instructions per triple).  One v_mfma_f32_32x32x16_f16 evaluates
//     it for a 32 x 32 tile of pairs: the accumulator input is -a12 (+ margin), the 16 k-slots carry the split-FP16 halves
//     of the three products (hi.hi + hi.lo + lo.hi each: 9 slots) and two "always pass" slots; 1024 triples per 32 pipe
//     cycles against ~45 vector-issue cycles per 64 triples before.  The errors of the split (2^-19 per product), of the
//     FP32 accumulator input and of the pipe's summation (kappa 2^-24 per addend, tools/micro/mfma_sum_model.hip) are
//     proportional to |d1||d2| and are covered by a margin added to the accumulator input (MFX_K3M_C); a test on host data
//      shows the pass fraction does not change with it.  A passing triple is only a candidate for scoring:
//     nothing depends on the test's value;
//   * a workgroup owns 256 x 128 pairs (8 waves x 4 accumulator-input tiles in registers) and walks i3 in blocks of 4
//     whose operands all threads build in LDS (double-buffered: one barrier per block) from the CURRENT threshold;
//   * the threshold starts from a greedy triple (best pair of every two dictionaries + its best third atom): the test
//     lets 6e-3 .. 0.2 of all triples through at the best two-atom score, 3e-5 .. 1e-3 at the greedy one, 2e-7 at the end;
//   * everything is batched over voxels (grid z / y): rotation, Gram cross blocks (only the three N x N blocks the triples
//     need, not the 3N x 3N matrix), statistics, seed, screen, finalize, packing - no per-voxel launch chain.
// A candidate list that overflows (more than MFX_K3B_CAP triples within 1e-9 |y|^2 of the optimum: a one-atom signal
// fitted with three fascicles ties N^2 triples) flags the voxel; the launcher then runs solve_k3.hip's path for it, gated
// on the device by that flag (no host read).
#pragma once
#include "solve_k3.hip"

typedef _Float16 k3_h8 __attribute__((ext_vector_type(8)));
// f = hi + lo + r, hi = f with its mantissa cut to 10 bits (exact in FP16), lo = fp16(f - hi), |r| <= 2^-21 |f| (fit_k2s.hip)
__device__ __forceinline__ void k3_split16(float f, _Float16& hi, _Float16& lo) {
  asm("" : "+v"(f));
  const float h = __uint_as_float(__float_as_uint(f) & 0xffffe000u);
  hi = (_Float16)h;
  lo = (_Float16)(f - h);
}
typedef float k3_f16v __attribute__((ext_vector_type(16)));

#define MFX_K3B_CAP (1 << 22)      // candidate list entries per voxel
#define MFX_K3M_KB 4               // i3 values per operand block
#ifndef MFX_K3M_TI
#define MFX_K3M_TI 8               // i1 tiles of 32 per workgroup (one per wave)
#endif
#ifndef MFX_K3M_TJ
#define MFX_K3M_TJ 4               // i2 tiles of 32 per workgroup (each wave multiplies all of them)
#endif
#ifndef MFX_K3M_Q
#define MFX_K3M_Q 1024             // queue entries per block parity
#endif
#define MFX_K3M_PAD 4              // 16-byte entries of padding behind every operand slot
#define MFX_K3M_D 8e-6f            // margin in units of |d1'||d2'|: the FP32 constants (4e-6, as solve_k3.hip) + the split of P and Q
#define MFX_K3M_C 1.0e-6           // margin in units of |d1||d2|: FP32 rounding of a12 and u (1.8e-7), summation inside the pipe
                                   // (5.1 x 2^-24 x ~2.2 |d1||d2| = 6.7e-7), three-half split of u1 u2 (< 1e-8)
#define MFX_K3M_BIG 60000.0f       // "always pass" factor (FP16 range)

// What the screen needs of an atom d (first or second dictionary) beside a third atom d3, whatever the threshold: d' = d
// minus its projection on d3, y' likewise; z = d'.y'/|d'|, n = |d'| (0: d is (nearly) inside span(d3): every partner
// passes), u = d.d3/|d3| in three FP16 halves, the margin slot sqrt(D) |d'|.  Built once per voxel (mfx_k3b_items_kernel);
// the screen's workgroups - 24 of them share an atom of the first dictionary, 12 one of the second - only add what
// depends on the running threshold.
struct K3Item {
  float z, n;
  _Float16 uh, um, ul, mg;
};
static_assert(sizeof(K3Item) == 16, "K3Item is one 16-byte load");

struct K3BArgs {
  int B, M, N, LD;                 // voxels in the batch, measurements, atoms per dictionary, 3 N
  int cap;                         // candidate list entries in use (<= MFX_K3B_CAP; tests lower it to force the fallback)
  const double* A;                 // [B][M][LD] rotated dictionaries, row-major
  const double* Y;                 // signals of all voxels [V][M]
  const int* vox;                  // [B] voxel of slot b
  double* G;                       // [B][3][N][N]: G12[i1][i2], G13[i1][i3], G23[i2][i3]
  double* nrm2;                    // [B][LD] |d|^2
  double* aty;                     // [B][LD] d.y
  double* ysq;                     // [B][2] sequential, pairwise
  double2* st3;                    // [B][N] 1/|d3|, y.d3/|d3|
  struct K3Item* items;            // [B][2][ceil(N / KB)][N][KB]: the threshold-independent part of every (atom, third atom) item
  unsigned long long* thr;         // [B] bits of the best score so far
  unsigned long long* seed;        // [B][3] best pair of each dictionary pair: (float score bits << 32) | (p N + q)
  int* ncand;                      // [B][2] candidates appended, overflow flag
  unsigned long long* dbg;         // diagnostics (MFX_K3_DEBUG=1), or null: [B][4] triples scored, of them on the spot, pushed, threshold at screen start
  double* cand_score;              // [B][MFX_K3B_CAP]
  long* cand_tuple;                // [B][MFX_K3B_CAP]
  // finalize
  double* part;                    // [B][FW][8]: res, key, tuple, w0, w1, w2
  double* w; long* sub; double* minobj; double* yrec;   // [B][8], [B][8], [B], [B][M]
};
#define MFX_K3B_FW 8

// ---- column statistics: |d|^2, d.y (sequential over the rows), |y|^2, and 1/|d3|, y.d3/|d3| for the third dictionary
__global__ __launch_bounds__(256) void mfx_k3b_stats_kernel(K3BArgs k) {
  const int b = blockIdx.y, col = blockIdx.x * 256 + threadIdx.x;
  const double* __restrict__ A = k.A + (size_t)b * k.M * k.LD;
  const double* __restrict__ y = k.Y + (size_t)k.vox[b] * k.M;
  if (col < k.LD) {
    double s2 = 0.0, sy = 0.0;
    for (int m = 0; m < k.M; ++m) { const double d = A[(size_t)m * k.LD + col]; s2 += d * d; sy += y[m] * d; }
    k.nrm2[(size_t)b * k.LD + col] = s2;
    k.aty[(size_t)b * k.LD + col] = sy;
    if (col >= 2 * k.N) {
      const double in3 = 1.0 / sqrt(s2);
      k.st3[(size_t)b * k.N + col - 2 * k.N] = double2{in3, sy * in3};
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int m = 0; m < k.M; ++m) s += y[m] * y[m];
    k.ysq[2 * b] = s;
    k.ysq[2 * b + 1] = mfx_np_sumsq(y, k.M);
    k.thr[b] = 0ull;
    k.seed[3 * b] = k.seed[3 * b + 1] = k.seed[3 * b + 2] = 0ull;
    k.ncand[2 * b] = k.ncand[2 * b + 1] = 0;
  }
}

// ---- the three cross blocks of the Gram on FP64 MFMA (ranking only): grid (N/64, N/128, 3 B), one wave = 32 x 64 outputs
// (two row fragments x four column fragments per k-step: 6 operand loads per 8 MFMAs; with 16 x 64 per wave it was 5 per 4
// and the kernel ran at the rate of its loads)
__global__ __launch_bounds__(256) void mfx_k3b_gram_kernel(K3BArgs k) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane >> 4, lc = lane & 15;
  // Workgroups go to the 8 XCDs round-robin by their linear index, and each XCD has its own L2: the tiles of ONE matrix
  // (they share its operand columns) are dealt to ONE XCD - linear index l -> matrix 8 (l / (8 T)) + l % 8, tile (l / 8) % T
  // - instead of every XCD streaming every matrix (the last, incomplete group of matrices keeps the plain order)
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const unsigned T = gridDim.x * gridDim.y, l = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned grp = l / (8 * T);
    if (8 * (grp + 1) <= gridDim.z) {
      const unsigned tl = (l / 8) % T;
      bz = (int)(8 * grp + l % 8); bx = (int)(tl % gridDim.x); by = (int)(tl / gridDim.x);
    }
  }
  const int b = bz / 3, which = bz % 3;
  const int N = k.N, M = k.M, LD = k.LD;
  const int cp = (which == 2) ? N : 0, cq = (which == 0) ? N : 2 * N;   // column offsets of the row / column dictionary
  const int p0 = by * 128 + wave * 32, q0 = bx * 64;
  if (p0 >= N) return;
  const double* __restrict__ A = k.A + (size_t)b * M * LD;
  k3_d4 acc[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[s][t] = k3_d4{0, 0, 0, 0};
  int pi[2], qj[4];
#pragma unroll
  for (int s = 0; s < 2; ++s) pi[s] = cp + min(p0 + 16 * s + lc, N - 1);
#pragma unroll
  for (int t = 0; t < 4; ++t) qj[t] = cq + min(q0 + 16 * t + lc, N - 1);
  // Operands of the NEXT group of U k-steps are requested before the current group is multiplied: a k-step is 8 MFMAs =
  // 512 cycles of the pipe, a load from L2 / HBM takes 1 500-5 000 - with one k-step in flight the kernel ran at the
  // rate of its load latency (31 TFLOP/s whatever the tile shape or the placement of the tiles on the XCDs)
  constexpr int U = 4;
  double av[U][2], bv[U][4];
  auto load_group = [&](int kbase, double (&a)[U][2], double (&bb)[U][4]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kr = kbase + 4 * u + lg;
      const bool ok = kr < M;
      const size_t ro = (size_t)min(kr, M - 1) * LD;
#pragma unroll
      for (int s = 0; s < 2; ++s) { const double v = A[ro + pi[s]]; a[u][s] = ok ? v : 0.0; }
#pragma unroll
      for (int t = 0; t < 4; ++t) { const double v = A[ro + qj[t]]; bb[u][t] = ok ? v : 0.0; }
    }
  };
  load_group(0, av, bv);
  for (int k0 = 0; k0 < M; k0 += 4 * U) {
    double avn[U][2], bvn[U][4];
    load_group(k0 + 4 * U, avn, bvn);      // (beyond M: clamped addresses, zeros)
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[s][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][s], bv[u][t], acc[s][t], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int s = 0; s < 2; ++s) av[u][s] = avn[u][s];
#pragma unroll
      for (int t = 0; t < 4; ++t) bv[u][t] = bvn[u][t];
    }
  }
  double* __restrict__ G = k.G + ((size_t)b * 3 + which) * N * N;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = p0 + 16 * s + lg + 4 * r, q = q0 + 16 * t + lc;
        if (p < N && q < N) G[(size_t)p * N + q] = acc[s][t][r];
      }
}

// ---- threshold seed, step 1: the best pair of each of the three dictionary pairs (FP32 ranking of score2), grid (x, B)
__global__ __launch_bounds__(256) void mfx_k3b_pairs_kernel(K3BArgs k) {   // grid (ceil(N / 8), 3, B): 8 rows of one cross block
  const int b = blockIdx.z, which = blockIdx.y, N = k.N;
  const double* __restrict__ n2 = k.nrm2 + (size_t)b * k.LD;
  const double* __restrict__ ay = k.aty + (size_t)b * k.LD;
  const long nn = (long)N * N;
  const int p = blockIdx.x * 8 + (threadIdx.x >> 5);
  unsigned long long best = 0ull;
  if (p < N) {
    const int cp = (which == 2) ? N + p : p, cq0 = (which == 0) ? N : 2 * N;
    const double a11 = n2[cp], y1 = ay[cp];
    const double* __restrict__ Gp = k.G + ((size_t)b * 3 + which) * nn + (size_t)p * N;
    for (int q = threadIdx.x & 31; q < N; q += 32) {
      const double s = score2(a11, Gp[q], n2[cq0 + q], y1, ay[cq0 + q]);
      const unsigned long long key = ((unsigned long long)__float_as_uint(fmaxf((float)s, 0.0f)) << 32) | (unsigned long long)((long)p * N + q);
      if (key > best) best = key;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long u = __shfl_xor(best, o); best = u > best ? u : best; }
  if ((threadIdx.x & 63) == 0 && (best >> 32)) atomicMax(&k.seed[3 * b + which], best);
}
// ---- step 2: the best third atom for each of those pairs; the best of the three triples (and pairs) starts the threshold
__global__ __launch_bounds__(256) void mfx_k3b_greedy_kernel(K3BArgs k) {
  const int b = blockIdx.y, which = blockIdx.x, N = k.N;
  const unsigned long long sd = k.seed[3 * b + which];
  if (!(sd >> 32)) return;
  const long nn = (long)N * N, e = (long)(sd & 0xffffffffull);
  const int p = (int)(e / N), q = (int)(e - (long)p * N);
  const double* __restrict__ n2 = k.nrm2 + (size_t)b * k.LD;
  const double* __restrict__ ay = k.aty + (size_t)b * k.LD;
  const double* __restrict__ G12 = k.G + (size_t)b * 3 * nn, *G13 = G12 + nn, *G23 = G13 + nn;
  double best = 0.0;
  for (int t = threadIdx.x; t < N; t += 256) {
    int i1, i2, i3;
    if (which == 0) { i1 = p; i2 = q; i3 = t; } else if (which == 1) { i1 = p; i3 = q; i2 = t; } else { i2 = p; i3 = q; i1 = t; }
    best = fmax(best, score3(n2[i1], G12[(size_t)i1 * N + i2], G13[(size_t)i1 * N + i3], n2[N + i2], G23[(size_t)i2 * N + i3], n2[2 * N + i3],
                             ay[i1], ay[N + i2], ay[2 * N + i3]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = fmax(best, __shfl_xor(best, o));
  if ((threadIdx.x & 63) == 0) k3_raise(k.thr + b, best);
}

// ---- the items of the screen (see K3Item): grid (ceil(N * nblk * KB / 256), 2, B), one thread per (atom, third atom)
__global__ __launch_bounds__(256) void mfx_k3b_items_kernel(K3BArgs k) {
  constexpr int KB = MFX_K3M_KB;
  const int b = blockIdx.z, side = blockIdx.y, N = k.N;
  const int nblk = (N + KB - 1) / KB;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;       // a * (nblk KB) + k3: the Gram row is read along k3
  const int a = (int)(e / (nblk * KB)), k3 = (int)(e - (long)a * (nblk * KB));
  if (a >= N || k3 >= N) return;
  const long nn = (long)N * N;
  const double* __restrict__ G3 = k.G + ((size_t)b * 3 + 1 + side) * nn;     // G13 / G23
  const double2 s3 = k.st3[(size_t)b * N + k3];
  const double aa = k.nrm2[(size_t)b * k.LD + side * N + a], ya = k.aty[(size_t)b * k.LD + side * N + a];
  const double u = G3[(size_t)a * N + k3] * s3.x;
  const double np2 = aa - u * u;                            // |d'|^2
  const double zn = ya - u * s3.y;                          // d'.y'
  K3Item it;
  it.z = 0.0f; it.n = 0.0f; it.mg = (_Float16)0.0f;
  if (np2 > 1e-10 * aa) {
    const float n2f = (float)np2, rs = __builtin_amdgcn_rsqf(n2f);
    it.n = n2f * rs;
    it.z = (float)zn * rs;
    it.mg = (_Float16)(__builtin_amdgcn_sqrtf(MFX_K3M_D) * it.n * 1.002f);
  }
  {   // u in THREE halves (u = uh + um + ul to 2^-32: the products uh uh, uh um, um uh, um um, uh ul, ul uh leave ~1e-9 |u1 u2|);
      // u1 u2 nearly cancels a12, so its error is what the accumulator margin is made of
    float f = (float)u;
    asm("" : "+v"(f));
    const float h1 = __uint_as_float(__float_as_uint(f) & 0xffffe000u);
    const float r1 = f - h1;
    const float h2 = __uint_as_float(__float_as_uint(r1) & 0xffffe000u);
    it.uh = (_Float16)h1; it.um = (_Float16)h2; it.ul = (_Float16)(r1 - h2);
  }
  k.items[((((size_t)b * 2 + side) * nblk + (k3 / KB)) * N + a) * KB + (k3 % KB)] = it;
}

// ---- the triple screen on the matrix pipe: grid (ceil(N / 128), ceil(N / 256), B), 512 threads
// (two waves per SIMD - two workgroups per CU - is what the LDS allows; saying so keeps the accumulators of the matrix
// instructions in the vector registers the comparisons read, instead of copying 16 registers in and out per instruction)
__global__ __launch_bounds__(MFX_K3M_TI * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void mfx_k3b_screen_kernel(K3BArgs k) {
  constexpr int KB = MFX_K3M_KB, TI = MFX_K3M_TI, TJ = MFX_K3M_TJ, WGS = TI * 64;
  extern __shared__ double k3m_smem[];
  // (operand slots padded by 64 bytes: the four lanes that build one atom's items for the block's four third atoms store
  // 16 bytes each a whole slot apart - without the padding into the same banks: 54 % of the LDS cycles were conflicts)
  constexpr int SA = TI * 64 + MFX_K3M_PAD, SB = TJ * 64 + MFX_K3M_PAD;
  k3_h8* sA = (k3_h8*)k3m_smem;                          // [2][KB][SA]  operands of the i1 side, fragment order: [TI][64] + padding
  k3_h8* sB = sA + 2 * KB * SA;                            // [2][KB][SB]
  double* s_aa = (double*)(sB + 2 * KB * SB);              // [(TI + TJ) * 32] |d|^2 of the workgroup's atoms (0: beyond the dictionary)
  double* s_ay = s_aa + (TI + TJ) * 32;                    // [(TI + TJ) * 32] d.y
  float* s_b3 = (float*)(s_ay + (TI + TJ) * 32);           // [2][KB][4] per third atom: T - z3^2, z3 = y.d3/|d3|, -, -
  unsigned* s_q = (unsigned*)(s_b3 + 2 * KB * 4);          // [2][MFX_K3M_Q] passing triples: (i local << 9) | (j local << 2) | kk
  int* s_qn = (int*)(s_q + 2 * MFX_K3M_Q);                 // [2]
  double* s_T = (double*)(s_qn + 2);                       // [3] the threshold of block b in slot b mod 3
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, N = k.N;
  const long nn = (long)N * N;
  const int i0 = blockIdx.y * (TI * 32), j0 = blockIdx.x * (TJ * 32);
  const double* __restrict__ G12 = k.G + (size_t)b * 3 * nn, *G13 = G12 + nn, *G23 = G13 + nn;
  const double* __restrict__ n2 = k.nrm2 + (size_t)b * k.LD;
  const double* __restrict__ ay = k.aty + (size_t)b * k.LD;
  const double2* __restrict__ st3 = k.st3 + (size_t)b * N;
  unsigned long long* thrp = k.thr + b;
  const double y_sq = k.ysq[2 * b];
  const double eps_abs = 1e-9 * y_sq;
  for (int q = tid; q < (TI + TJ) * 32; q += WGS) {
    const bool side = q >= TI * 32;
    const int a = side ? j0 + q - TI * 32 : i0 + q;
    const bool ok = a < N;
    s_aa[q] = ok ? n2[side ? N + a : a] : 0.0;
    s_ay[q] = ok ? ay[side ? N + a : a] : 0.0;
  }
  if (tid < 2) s_qn[tid] = 0;
  if (tid == 0) s_T[0] = __longlong_as_double((long long)*(volatile unsigned long long*)thrp) - eps_abs;
  if (k.dbg && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) k.dbg[4 * b + 3] = *thrp;
  __syncthreads();
  // accumulator inputs of this wave's TJ tile pairs: -a12 + margin |d1||d2|, rounded up; -1e30 where there is no pair.
  // Tiles with a hit get a second test (below): the relaxed score in FP32 without the
  // matrix-pipe margin, and the sign of the third atom's weight in the relaxed optimum - the relaxation lets it go negative,
  // and a triple whose relaxed optimum gives atom 3 a CLEARLY negative weight has its NNLS optimum on a face: {1, 3}, {2, 3},
  // {3} stay below the threshold unless an "always pass" slot fired (accumulator ~1e9: kept), the pair {1, 2}'s own score
  // reaches it only for pairs marked here (accumulator input 3e9: every third atom kept).  In a voxel with a flat optimum
  // (an inactive atom) the first test alone lets 1e7 .. 3e8 triples through at N = 1500.
  k3_f16v C[TJ];
  {
    const double T0 = __longlong_as_double((long long)*(volatile unsigned long long*)thrp) - eps_abs;   // (the threshold only rises)
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
      const int jl = t * 32 + lr, j = j0 + jl;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int il = wave * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh, i = i0 + il;
        float c = -1e30f;
        if (i < N && j < N) {
          const double a11 = s_aa[il], a22 = s_aa[TI * 32 + jl], a12 = G12[(size_t)i * N + j], y1 = s_ay[il], y2 = s_ay[TI * 32 + jl];
          const double v = fma(MFX_K3M_C, sqrt(a11 * a22), -a12);
          c = (float)(v + fabs(v) * 1.3e-7);
          if (score2(a11, a12, a22, y1, y2) >= T0) c = 3e9f;
        }
        C[t][g] = c;
      }
    }
  }
  // {z', |d'| (0: the item passes with every partner at this block's threshold, or lies inside span(d3)), u} of the item
  // (atom a of dictionary `side`, third atom k3), for the rare paths below: re-read from the item array (the block's
  // threshold constant Tp comes from s_b3) instead of being kept in LDS for every item of every block - two LDS stores and
  // ~10 vector instructions per built item, 37 KB of LDS
  const int nblk_items = (N + KB - 1) / KB;
  auto item_stat = [&](int side, int a, int k3, float Tp, float& zit, float& nit, float& uu) {
    const K3Item it = k.items[((((size_t)b * 2 + side) * nblk_items + (k3 / KB)) * N + min(a, N - 1)) * KB + (k3 % KB)];
    const bool has = it.n > 0.0f;
    const bool always = !(Tp > 0.0f) || (it.z > 0.0f && it.z * it.z >= Tp * (1.0f - 3e-6f));
    zit = has ? it.z : -1e30f;
    nit = (has && !always) ? it.n : 0.0f;
    uu = ((float)it.uh + (float)it.um) + (float)it.ul;
  };
  // score a passing triple from the Gram; list it when it reaches the running threshold (solve_k3.hip's score_triple)
  double best = 0.0;
  // (the Gram entries with the third atom come from the block's LDS copies: three random reads of an 18 MB matrix per
  // scored triple were what a voxel with a loose bound - an inactive atom at the optimum: millions of passing triples - paid)
  auto score_triple = [&](int il, int jl, int par, int kk, int k3) {
#ifdef MFX_K3M_EXP_NOSCORE   // timing experiment (wrong results)
    return;
#endif
    const int i = i0 + il, j = j0 + jl;
    if (i >= N || j >= N || k3 >= N) return;
    if (k.dbg) atomicAdd(&k.dbg[4 * b], 1ull);
    const double a11 = s_aa[il], a12 = G12[(size_t)i * N + j], a13 = G13[(size_t)i * N + k3], a22 = s_aa[TI * 32 + jl], a23 = G23[(size_t)j * N + k3],
                 a33 = n2[2 * N + k3], y1 = s_ay[il], y2 = s_ay[TI * 32 + jl], y3 = ay[2 * N + k3];
    double s;
    // The relaxed bound lets the third atom's weight go negative: in a voxel whose optimum has an inactive atom millions of
    // triples pass on that alone.  If the unconstrained three-atom optimum gives atom 3 a CLEARLY negative weight (Cramer
    // numerator D3 < 0), the NNLS optimum of the triple lies on a face: without atom 3 - then it is the pair (1, 2)'s - or on
    // {1, 3}, {2, 3}, {3}, which stay below the threshold unless the item's "always pass" flag is set (the flag IS that
    // test with the partner unconstrained).  Ten FP64 operations instead of score3's hundred and fifty.
    const double c13 = a12 * a23 - a13 * a22, c23 = a12 * a13 - a11 * a23, c33 = a11 * a22 - a12 * a12;
    const double D3 = y1 * c13 + y2 * c23 + y3 * c33;
    bool alw;   // an "always pass" item
    {
      const float Tpb = s_b3[(par * KB + kk) * 4];
      float z1_, n1_, u1_, z2_, n2_, u2_;
      item_stat(0, i, k3, Tpb, z1_, n1_, u1_);
      item_stat(1, j, k3, Tpb, z2_, n2_, u2_);
      alw = !(n1_ > 0.0f) || !(n2_ > 0.0f);
    }
    if (!alw && D3 < -1e-10 * (fabs(y1 * c13) + fabs(y2 * c23) + fabs(y3 * c33)))
      s = score2(a11, a12, a22, y1, y2);
    else
      s = score3(a11, a12, a13, a22, a23, a33, y1, y2, y3);
    const double Tn = __longlong_as_double((long long)*(volatile unsigned long long*)thrp) - eps_abs;
    if (s >= Tn && s > 0.0) {
      const int slot = atomicAdd(&k.ncand[2 * b], 1);
      if (slot < k.cap) {
        k.cand_score[(size_t)b * MFX_K3B_CAP + slot] = s;
        k.cand_tuple[(size_t)b * MFX_K3B_CAP + slot] = ((long)i * N + j) * N + k3;   // itertools order: last index fastest
      } else {
        k.ncand[2 * b + 1] = 1;   // overflow: solve_k3.hip's path redoes the voxel
      }
      if (s > best) { best = s; k3_raise(thrp, s); }
    }
  };
  auto drain = [&](int par, int k0) {   // every thread scores its share of the queue of block parity `par` (first i3: k0)
    const int nq = min(s_qn[par], MFX_K3M_Q);
    for (int q = tid; q < nq; q += WGS) {
      const unsigned e = s_q[par * MFX_K3M_Q + q];
      score_triple((int)(e >> 9), (int)((e >> 2) & 127u), par, (int)(e & 3u), k0 + (int)(e & 3u));
    }
  };
  const int nblk = (N + KB - 1) / KB;
  // a thread's items of a block: (TI + TJ) x 32 atoms x KB third atoms over WGS threads; WGS is a multiple of KB, so they
  // all belong to one third atom (k0 + kk).  They are loaded a block ahead: the loads fly behind the multiplications
  constexpr int NIT = (TI + TJ) * 32 * KB / WGS;
  static_assert(NIT * WGS == (TI + TJ) * 32 * KB && WGS % KB == 0, "items per thread");
  const int kk = tid & (KB - 1);
  uint4 nxt[NIT];    // (K3Item as four dwords: unpacked where it is used)
  double z3n = 0.0;
  auto load_items = [&](int blk_) {   // (unconditional loads from clamped addresses: nothing here waits for them)
    const int bq = min(blk_, nblk - 1);
    const uint4* __restrict__ itm = (const uint4*)k.items + ((size_t)b * 2 * nblk + bq) * N * KB;
    z3n = st3[min(bq * KB + kk, N - 1)].y;
#pragma unroll
    for (int r = 0; r < NIT; ++r) {
      const int al = (tid + r * WGS) / KB;
      const bool side = al >= TI * 32;
      const int a = min(side ? j0 + al - TI * 32 : i0 + al, N - 1);
      nxt[r] = itm[((size_t)(side ? nblk : 0) * N + a) * KB + kk];
    }
  };
  load_items(0);
  unsigned long long thr_ahead = __hip_atomic_load(thrp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (thread 0's)
  for (int blk = 0; blk <= nblk; ++blk) {
    const int buf = blk & 1, k0 = blk * KB;
    uint4 cur[NIT];
#pragma unroll
    for (int r = 0; r < NIT; ++r) cur[r] = nxt[r];
    const double z3 = z3n;
    load_items(blk + 1);   // (outside the condition below: a value defined on one side of a branch is copied, and waited for, at its end)
#ifdef MFX_K3M_EXP_NOBUILD   // timing experiment (wrong results): operands of the first two blocks only
    if (blk < 2) {
#else
    if (blk < nblk) {
#endif
      // ---- operands of the block from ONE recent threshold (read by thread 0 a block ahead, behind a barrier):
      // (TI + TJ) x 32 atoms x KB third atoms, 3 items per thread
      // (the value published for block blk + 1 was requested from memory while block blk - 1 was multiplied: waiting for
      // the round trip at every block - four waves behind a barrier behind one load - was half of this kernel's time)
      const double T = s_T[blk % 3];
      if (tid == 0) {
        s_T[(blk + 1) % 3] = __longlong_as_double((long long)thr_ahead) - eps_abs;
        thr_ahead = __hip_atomic_load(thrp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const int k3 = k0 + kk;
      float Tp = 1e30f, rth = 0.0f, z3f = 0.0f;
      if (k3 < N) {
        Tp = (float)(T - z3 * z3) * (1.0f - 2e-7f);         // what the two projected atoms must reach
        rth = __builtin_amdgcn_rsqf(fmaxf(Tp, 1e-30f)) * (1.0f + 1e-6f);
        z3f = (float)z3;
      }
      // Straight-line code, selects instead of branches (the branchy form spent a third of its vector instructions
      // re-initialising defaults on every path); which operand an item belongs to is known at compile time: a thread's
      // r-th item is atom (tid + r WGS) / KB of the workgroup, the i1 side first.
      mfx_static_for<0, NIT>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        constexpr bool side = (r * WGS) / KB >= TI * 32;
        static_assert((TI * 32 * KB) % WGS == 0, "an item index r belongs to one side");
        const int al = (tid + r * WGS) / KB;                 // al: atom within the workgroup
        const bool valid = (k3 < N) && ((side ? j0 + al - TI * 32 : i0 + al) < N);   // beyond the dictionary: never passes
        const float z = __uint_as_float(cur[r].x), nrm = __uint_as_float(cur[r].y);
        const bool has = valid && nrm > 0.0f;                // (valid, nrm == 0: (nearly) inside span(d3), every partner passes)
        const bool always = !(Tp > 0.0f) || (z > 0.0f && z * z >= Tp * (1.0f - 3e-6f));
        // P = cos theta over the whole range of z: an atom whose projection on the signal is NEGATIVE can still carry a
        // positive weight beside a partner at an obtuse angle (after the projection on d3's complement that is common),
        // and S(c) = T at c = cos(theta1 + theta2) holds for either sign (cos^2 a + cos^2 b - 2 cos(a+b) cos a cos b =
        // sin^2(a+b)); clamping z at 0 (solve_k3.hip) keeps the test valid but lets every sufficiently obtuse pair of
        // such an atom through: 1e7 .. 3e8 triples per voxel in voxels with a flat optimum.  The margin D |d1'||d2'| has
        // its own k-slot (the folding of fit_k2s.hip assumes P >= 0).
        // (valid for either sign of z only if both atoms of a test see the SAME threshold - for z > 0 a lower threshold
        // lets more pairs through, for z < 0, where theta decreases with T, fewer: all items of a block are built from
        // one value, s_T[block mod 3], published a block ahead)
        const float Pc = __builtin_amdgcn_fmed3f(z * rth, -1.0f, 1.0f);
        const float Qc = __builtin_amdgcn_sqrtf(fmaxf(0.0f, fmaf(-Pc, Pc, 1.0f)));
        const float P = has ? Pc * nrm : 0.0f, Qv = has ? Qc * nrm : 0.0f;
        const bool alw = valid && !(has && !always);
        const unsigned uw = valid ? cur[r].z : 0u;           // halves (uh, um) of u
        const unsigned lw = valid ? cur[r].w : 0u;           // halves (ul, margin slot); the margin is 0 unless nrm > 0
        if (al == 0) { float* b3 = s_b3 + (buf * KB + kk) * 4; b3[0] = Tp; b3[1] = z3f; }   // (item (atom 0, kk) of the i1 side; beyond the dictionary: 1e30)
        _Float16 ph, pl, qh, ql;
        k3_split16(P, ph, pl);
        k3_split16(Qv, qh, ql);
        const unsigned php = __builtin_bit_cast(unsigned short, ph), plp = __builtin_bit_cast(unsigned short, pl);
        const unsigned qhp = __builtin_bit_cast(unsigned short, qh), qlp = __builtin_bit_cast(unsigned short, ql);
        constexpr unsigned BIGH = 0x7b53u;                   // 60000 in FP16 (MFX_K3M_BIG)
        static_assert(MFX_K3M_BIG == 60000.0f, "BIGH is MFX_K3M_BIG in FP16");
        uint4 lo8, hi8;
        if constexpr (!side) {   // A operand (rows): P P P' -Q -Q -Q' u u | u' u' u u'' alw BIG margin 0
          const unsigned nqh = qhp ^ 0x8000u, nql = qlp ^ 0x8000u;
          lo8 = uint4{php | (php << 16), plp | (nqh << 16), nqh | (nql << 16), (uw & 0xffffu) | (uw << 16)};
          hi8 = uint4{(uw >> 16) | (uw & 0xffff0000u), (uw & 0xffffu) | (lw << 16), alw ? (BIGH | (BIGH << 16)) : (BIGH << 16), lw >> 16};
          uint4* dst = (uint4*)sA + (size_t)(buf * KB + kk) * SA + (al >> 5) * 64 + (al & 31);
          dst[0] = lo8; dst[32] = hi8;
        } else {                 // B operand (columns): P P' P Q Q' Q u u' | u u' u'' u BIG alw margin 0
          const int bl = al - TI * 32;
          lo8 = uint4{php | (plp << 16), php | (qhp << 16), qlp | (qhp << 16), uw};
          hi8 = uint4{uw, (lw & 0xffffu) | (uw << 16), alw ? (BIGH | (BIGH << 16)) : BIGH, lw >> 16};
          uint4* dst = (uint4*)sB + (size_t)(buf * KB + kk) * SB + (bl >> 5) * 64 + (bl & 31);
          dst[0] = lo8; dst[32] = hi8;
        }
      });
    }
    __syncthreads();   // the block's operands are in place; every push of the previous block is in its queue
    if (blk >= 1) {
      const int par = (blk - 1) & 1;
      if (s_qn[par] > 0) {   // workgroup-uniform: nobody pushes to this queue before the next barrier
        drain(par, (blk - 1) * KB);
        __syncthreads();
        if (tid == 0) s_qn[par] = 0;
      }
    }
#ifdef MFX_K3M_EXP_NOMFMA   // timing experiment (wrong results): operand build and barriers only
    if (blk > nblk) {
#else
    if (blk < nblk) {
#endif
      const int nk = min(KB, N - k0);
      // One multiplication ahead: the matrix instruction of the NEXT (third atom, column tile) is issued before the result
      // of the current one is looked at, so its 8 passes run behind these ~10 vector instructions instead of in front of them.
      auto mma = [&](const k3_h8& af, int kk, auto tc) -> k3_f16v {
        constexpr int t = decltype(tc)::value;
        const k3_h8 bf = sB[(size_t)(buf * KB + kk) * SB + t * 64 + lane];
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, C[t], 0, 0, 0);
      };
      auto post = [&](const k3_f16v& d, int kk, auto tc) {
        constexpr int t = decltype(tc)::value;
          // any test value >= 0 in the tile?  As signed integers the bit patterns of non-negative floats are the non-negative
          // ones: an integer maximum (three operands per instruction, no NaN handling) answers it.  (-0.0 counts as negative:
          // it would take an accumulator input of exactly -0, and that one carries a positive margin.)
          int mi = max(max(__float_as_int(d[0]), __float_as_int(d[1])), __float_as_int(d[2]));
#pragma unroll
          for (int g = 3; g < 15; g += 2) mi = max(max(mi, __float_as_int(d[g])), __float_as_int(d[g + 1]));
          mi = max(mi, __float_as_int(d[15]));
#ifdef MFX_K3M_EXP_NOHIT   // timing experiment (wrong results)
          if (mi == 0x7fffffff) {
#else
          if (__any(mi >= 0)) {
#endif
            // (rare path) the third atom's weight in the unconstrained optimum, per accumulator register
            // (rare path: the tile has a hit) second test, in registers, before anything is queued: (i) the relaxed two-atom
            // score itself in FP32 from the items' statistics (the matrix-pipe test carries a margin ~1e-6 |d1||d2|, i.e.
            // 1e-5 .. 1e-4 of the projected quantities: in a voxel with a flat optimum millions of triples sit inside it);
            // (ii) the third atom's weight in the unconstrained optimum (see the accumulator inputs above)
            const float* b3 = s_b3 + (buf * KB + kk) * 4;
            const float Tpf = b3[0], z3f = b3[1];
            float2 it2;
            float u2f;
            item_stat(1, j0 + t * 32 + lr, k0 + kk, Tpf, it2.x, it2.y, u2f);
#pragma unroll 1
            for (int g = 0; g < 16; ++g) {
              bool hit = d[g] >= 0.0f;
              if (!__any(hit)) continue;
              if (hit && d[g] < 1e8f) {
                const int ilr = wave * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                float2 it1;
                float u1f;
                item_stat(0, i0 + ilr, k0 + kk, Tpf, it1.x, it1.y, u1f);
                if (it1.y > 0.0f && it2.y > 0.0f) {
                  // everything multiplied through by N12 = |d1'||d2'| (no division): c' = a / N12, a = a12 - u1 u2;
                  // E_i = e_i N12, DEN = (1 - c'^2) N12^2, NUM = num N12
                  // (a12 itself comes from the Gram in memory: this path is rare, and a register copy of the tile per column
                  // tile is what kept the workgroup at 128 x 64 pairs)
                  const float a12f = (float)G12[(size_t)(i0 + ilr) * N + (j0 + t * 32 + lr)];
                  const float n12 = it1.y * it2.y, av = fmaf(-u1f, u2f, a12f);
                  const float E1 = fmaf(it1.x, n12, -av * it2.x), E2 = fmaf(it2.x, n12, -av * it1.x);
                  const float DEN = fmaf(n12, n12, -av * av), NUM = fmaf(it2.x, E2, it1.x * E1);
                  const float zt = 3e-6f * n12 * (fabsf(it1.x) + fabsf(it2.x));
                  // (i) two positive weights and S' >= T' (6e-6 of slack: FP32 evaluation); nearly collinear projected atoms pass
                  const bool illc = DEN < 1e-3f * n12 * n12;
                  hit = illc || (E1 > -zt && E2 > -zt && NUM * n12 >= Tpf * (1.0f - 6e-6f) * DEN);
                  // (ii) w3 |d3| DEN = z3 DEN - E1 |d2'| u1 - E2 |d1'| u2 >= 0 (clearly negative: dropped)
                  const float t1 = z3f * DEN, t2 = E1 * it2.y * u1f, t3 = E2 * it1.y * u2f;
                  hit = hit && (illc || (t1 - t2) - t3 >= -1e-5f * (fabsf(t1) + fabsf(t2) + fabsf(t3)));
                }
              }
              const unsigned long long mask = __ballot(hit);
              if (k.dbg && hit && d[g] >= 1e8f) atomicAdd(&k.dbg[4 * b + 2], 1ull);
              if (!mask) continue;
              int base = 0;
              if (lane == 0) base = atomicAdd(&s_qn[buf], __popcll(mask));      // one LDS atomic per register and wave
              base = __builtin_amdgcn_readfirstlane(base);
              if (hit) {
                const int il = wave * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh, jl = t * 32 + lr;
                const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
                if (slot < MFX_K3M_Q) s_q[buf * MFX_K3M_Q + slot] = ((unsigned)il << 9) | ((unsigned)jl << 2) | (unsigned)kk;
                else { if (k.dbg) atomicAdd(&k.dbg[4 * b + 1], 1ull); score_triple(il, jl, buf, kk, k0 + kk); }   // queue full: on the spot
              }
            }
          }
      };
      k3_h8 af = sA[(size_t)(buf * KB) * SA + wave * 64 + lane];
      k3_f16v dcur = mma(af, 0, std::integral_constant<int, 0>{});
      for (int kk = 0; kk < nk; ++kk) {
        mfx_static_for<0, TJ>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          k3_f16v dnext = dcur;
          if constexpr (t + 1 < TJ) {
            dnext = mma(af, kk, std::integral_constant<int, t + 1>{});
          } else {
            if (kk + 1 < nk) {
              af = sA[(size_t)(buf * KB + kk + 1) * SA + wave * 64 + lane];
              dnext = mma(af, kk + 1, std::integral_constant<int, 0>{});
            }
          }
          post(dcur, kk, tc);
          dcur = dnext;
        });
      }
    }
    // the next block's items and third-atom constant are first touched here, behind the multiplications: the wait for
    // their loads (issued before this block's operands were built) goes here instead of in front of the barrier
#pragma unroll
    for (int r = 0; r < NIT; ++r) asm volatile("" : "+v"(nxt[r].x), "+v"(nxt[r].y), "+v"(nxt[r].z), "+v"(nxt[r].w));
    asm volatile("" : "+v"(z3n));
  }
}

// ---- exact stage.  Step 1 (grid (MFX_K3B_FW, B)): every listed triple within 1e-9 |y|^2 of the best score in the
// reference's arithmetic (sequential Gram scalars mf_utils.py:503-535, Cramer test, explicit residual :562-593); the best
// (residual, scan-order key) of each workgroup's share
__global__ __launch_bounds__(256) void mfx_k3b_finalize_kernel(K3BArgs k) {
  __shared__ double s_res[256];
  __shared__ long s_key[256], s_tt[256];
  __shared__ double s_w[256][3];
  const int tid = threadIdx.x, b = blockIdx.y, N = k.N, M = k.M, LD = k.LD;
  const double* __restrict__ A = k.A + (size_t)b * M * LD;
  const double* __restrict__ y = k.Y + (size_t)k.vox[b] * M;
  const double y_sq = k.ysq[2 * b];
  const int nc = min(k.ncand[2 * b], k.cap);
  // (every raise of the threshold comes with a listed triple of that score - the seed's triples are found again by the
  // screen - so the threshold IS the list's maximum)
  const double thr = __longlong_as_double((long long)k.thr[b]) - 1e-9 * y_sq;
  double bres = INFINITY, bw[3] = {0.0, 0.0, 0.0};
  long bkey = -1, btt = -1;
  for (int c = blockIdx.x * 256 + tid; c < nc; c += MFX_K3B_FW * 256) {
    if (k.cand_score[(size_t)b * MFX_K3B_CAP + c] < thr) continue;
    const long t = k.cand_tuple[(size_t)b * MFX_K3B_CAP + c];
    const int i3 = (int)(t % N), i2 = (int)((t / N) % N), i1 = (int)(t / ((long)N * N));
    const int c1 = i1, c2 = N + i2, c3 = 2 * N + i3;
    double g11 = 0, g12 = 0, g13 = 0, g22 = 0, g23 = 0, g33 = 0, y1 = 0, y2 = 0, y3 = 0;
    for (int m = 0; m < M; ++m) {
      const double d1 = A[(size_t)m * LD + c1], d2 = A[(size_t)m * LD + c2], d3 = A[(size_t)m * LD + c3], ym = y[m];
      g11 += d1 * d1; g22 += d2 * d2; g33 += d3 * d3; g12 += d1 * d2; g13 += d1 * d3; g23 += d2 * d3;
      y1 += ym * d1; y2 += ym * d2; y3 += ym * d3;
    }
    auto explicit_res = [&](const double* ww) {
      double rr = 0.0;
      for (int m = 0; m < M; ++m) {
        const double t3 = (ww[0] * A[(size_t)m * LD + c1] + ww[1] * A[(size_t)m * LD + c2] + ww[2] * A[(size_t)m * LD + c3] - y[m]);
        rr += t3 * t3;
      }
      return rr;
    };
    double w[3], res;
    nnls3_cramer(y_sq, g11, g12, g13, g22, g23, g33, y1, y2, y3, explicit_res, w, res);
    const long key = ((long)i3 * N + i1) * N + i2;   // the reference scans i3 -> i1 -> i2 (mf_utils.py:540-547)
    if (res < bres || (res == bres && key < bkey)) { bres = res; bkey = key; btt = t; bw[0] = w[0]; bw[1] = w[1]; bw[2] = w[2]; }
  }
  s_res[tid] = bres; s_key[tid] = bkey; s_tt[tid] = btt;
  s_w[tid][0] = bw[0]; s_w[tid][1] = bw[1]; s_w[tid][2] = bw[2];
  __syncthreads();
  if (tid == 0) {
    int bi = -1;
    double br = INFINITY;
    long bk = -1;
    for (int i = 0; i < 256; ++i) {
      if (s_key[i] < 0) continue;
      if (s_res[i] < br || (s_res[i] == br && s_key[i] < bk)) { br = s_res[i]; bk = s_key[i]; bi = i; }
    }
    double* o = k.part + ((size_t)b * MFX_K3B_FW + blockIdx.x) * 8;
    o[0] = br;
    ((long*)o)[1] = bk;
    ((long*)o)[2] = bi >= 0 ? s_tt[bi] : -1;
    o[3] = bi >= 0 ? s_w[bi][0] : 0.0; o[4] = bi >= 0 ? s_w[bi][1] : 0.0; o[5] = bi >= 0 ? s_w[bi][2] : 0.0;
  }
}
// Step 2 (grid B): fold the shares into the reference's initial state (min_obj = |y|^2, w = 0, indices 0; strict '<'),
// the reconstruction y_recons = A[:, tot] @ w (mf_utils.py:606), and the voxel's parameter row (mf.py:420-450)
__global__ __launch_bounds__(256) void mfx_k3b_finish_kernel(K3BArgs k, PackArgs pk, double* params, int num_params) {
  __shared__ double s_w3[3];
  __shared__ int s_col[3];
  const int tid = threadIdx.x, b = blockIdx.x, N = k.N, M = k.M, LD = k.LD;
  const double* __restrict__ A = k.A + (size_t)b * M * LD;
  if (tid == 0) {
    double br = k.ysq[2 * b];
    long bk = -1, bt = -1;
    double w[3] = {0.0, 0.0, 0.0};
    for (int f = 0; f < MFX_K3B_FW; ++f) {
      const double* o = k.part + ((size_t)b * MFX_K3B_FW + f) * 8;
      const long key = ((const long*)o)[1];
      if (key < 0) continue;
      if (o[0] < br || (o[0] == br && bk >= 0 && key < bk)) { br = o[0]; bk = key; bt = ((const long*)o)[2]; w[0] = o[3]; w[1] = o[4]; w[2] = o[5]; }
    }
    const long t = bt < 0 ? 0 : bt;
    const int i3 = (int)(t % N), i2 = (int)((t / N) % N), i1 = (int)(t / ((long)N * N));
    k.w[8 * b] = w[0]; k.w[8 * b + 1] = w[1]; k.w[8 * b + 2] = w[2];
    k.sub[8 * b] = i1; k.sub[8 * b + 1] = i2; k.sub[8 * b + 2] = i3;
    k.minobj[b] = br;
    s_w3[0] = w[0]; s_w3[1] = w[1]; s_w3[2] = w[2];
    s_col[0] = i1; s_col[1] = N + i2; s_col[2] = 2 * N + i3;
  }
  __syncthreads();
  for (int m = tid; m < M; m += 256) {
    double t = 0.0;
    for (int p = 0; p < 3; ++p) t += A[(size_t)m * LD + s_col[p]] * s_w3[p];
    k.yrec[(size_t)b * M + m] = t;
  }
  __syncthreads();
  if (tid < 64) {
    pk.w = k.w + 8 * b; pk.sub = k.sub + 8 * b; pk.minobj = k.minobj + b; pk.yrec = k.yrec + (size_t)b * M;
    pk.y = k.Y + (size_t)k.vox[b] * M;
    pk.out = params + (size_t)k.vox[b] * num_params;
    mfx_pack_params_body(pk);
  }
}
