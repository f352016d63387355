// solve_k3.hip -- three sub-dictionaries (solve_exhaustive_posweights_3, mf_utils.py:470-607; BASELINE config 5:
// three fascicles, 1500 atoms x 300 measurements = 3.4e9 index triples per voxel) on top of the explicit-dictionary
// solver of solve_generic.hip: its one-thread-per-tuple scan is replaced by
//   1. mfx_k3_gram_kernel     the Gram G = A^T A on FP64 MFMA (v_mfma_f64_16x16x4_f64, 16 x 64 output per wave).  It only
//                             RANKS: the finalize stage re-sums the six Gram scalars of every candidate sequentially,
//                             as the reference does (mf_utils.py:503-535);
//   2. mfx_k3_pairs_kernel    threshold start: the best one- and two-atom supports (a triple is at least as good as the
//                             pairs inside it), one thread per Gram entry of the three cross blocks, global atomicMax;
//   3. mfx_k3_screen_kernel   every triple (i1, i2, i3) through a 5-instruction test before anybody scores it: with atom 3
//                             UNCONSTRAINED the problem is a two-atom problem in the orthogonal complement of d3 - projected
//                             atoms d' = d - (d.d3) d3 / |d3|^2, projected signal - whose score + (y.d3)^2/|d3|^2 bounds the
//                             triple's NNLS score from above; two positive weights reaching T need
//                             cos(d1', d2') <= cos(theta1 + theta2), theta_i = acos(min(1, z_i' / sqrt(T - q3))): the
//                             angle-sum test of fit_k2s.hip / fit_k2x.hip.  The cross term is updated per triple in FP64
//                             (a12' = a12 - u1 u2), the test runs in FP32 from two constants per (atom, i3).  A workgroup
//                             takes a 32 x 32 tile of (i1, i2) pairs and walks i3 in blocks of 64 whose constants it
//                             builds in LDS; a triple that passes is scored (feasible-support optimum from the Gram) and,
//                             when it reaches the running threshold, appended to a candidate list; the threshold T =
//                             best score so far - tie tolerance lives in global memory (atomicMax).
//   4. mfx_tuple_finalize     (solve_generic.hip) on the candidate list: exact _3 arithmetic, first hit in the
//                             reference's i3 -> i1 -> i2 order.
// A candidate list that overflows (massive ties: e.g. a single-fascicle signal fitted with three - every (i2, i3)
// ties) raises a device flag; the one-thread-per-tuple scan of solve_generic.hip then runs as before (it is launched
// unconditionally and exits at once when the flag is clear), so nothing is dropped silently.
#pragma once
#include "solve_generic.hip"

typedef double k3_d4 __attribute__((ext_vector_type(4)));

#define MFX_K3_CAP (1 << 20)   // candidate list entries
#define MFX_K3_KBL 4
#define MFX_K3_KB (1 << MFX_K3_KBL)   // i3 values per LDS block: 16 (17 KB of constants: four workgroups per CU; with two voxels in
                                      // flight 330 voxels/s at config 5 against 310 for 32 and 221 for 64)
#define MFX_K3_QCAP 2048       // queue of passing triples per block (of 16 384)
#define MFX_K3_D 4e-6f         // margin folded into the test constants (FP32 evaluation of the test and of its constants)

struct K3Args {
  SolveArgs s;                 // the generic solver's view of the problem (A, y, sizes, G, Aty, ysq, outputs)
  unsigned long long* thr;     // [1] bits of the best score so far (non-negative double)
  int* ncand;                  // [2]: candidates appended, overflow flag
  double2* st3;                // [N3] per atom of dictionary 3: 1 / |d3|, y.d3 / |d3|
  double* cand_score;          // [MFX_K3_CAP]  (aliases the generic solver's blk_score)
  long* cand_tuple;            // [MFX_K3_CAP]  (aliases blk_tuple)
};

// ---- 1. Gram on FP64 MFMA: one wave = 16 rows x 64 columns of G, 256-thread workgroups = 4 waves = 64 x 64
__global__ __launch_bounds__(256) void mfx_k3_gram_kernel(SolveArgs a) {
  if (a.run_if && !*a.run_if) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane >> 4, lc = lane & 15;
  const int N = a.Ntot, M = a.M;
  const int p0 = blockIdx.y * 64 + wave * 16, q0 = blockIdx.x * 64;
  if (p0 >= N) return;
  const double* __restrict__ A = a.A;
  k3_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const int pi = min(p0 + lc, N - 1);
  int qj[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) qj[t] = min(q0 + 16 * t + lc, N - 1);
  for (int k0 = 0; k0 < M; k0 += 4) {
    const int k = k0 + lg;
    const bool ok = k < M;
    const double av = ok ? A[(size_t)k * a.lda + pi] : 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double bv = ok ? A[(size_t)k * a.lda + qj[t]] : 0.0;
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = p0 + lg + 4 * r, q = q0 + 16 * t + lc;
      if (p < N && q < N) a.G[(size_t)p * N + q] = acc[t][r];
    }
}

// A^T y and |y|^2 (sequential sums, as the generic Gram kernel computes them)
__global__ void mfx_k3_aty_kernel(SolveArgs a) {
  if (a.run_if && !*a.run_if) return;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < a.Ntot) {
    double s = 0.0;
    for (int k = 0; k < a.M; ++k) s += a.y[k] * a.A[k * a.lda + idx];
    a.Aty[idx] = s;
  } else if (idx == a.Ntot) {
    double s = 0.0;
    for (int k = 0; k < a.M; ++k) s += a.y[k] * a.y[k];
    a.ysq[0] = s;
  } else if (idx == a.Ntot + 1) {
    a.ysq[1] = mfx_np_sumsq(a.y, a.M);
  }
}

__device__ __forceinline__ void k3_raise(unsigned long long* thr, double s) {
  if (s > 0.0) atomicMax(thr, (unsigned long long)__double_as_longlong(s));
}

// per atom of dictionary 3: 1/|d3| and y.d3/|d3| (used by every (atom, i3) item of the screen)
__global__ void mfx_k3_st3_kernel(K3Args k) {
  if (k.s.run_if && !*k.s.run_if) return;
  const SolveArgs& a = k.s;
  const int k3 = blockIdx.x * blockDim.x + threadIdx.x;
  if (k3 < a.sizes[2]) {
    const int c3 = (int)(a.sizes[0] + a.sizes[1]) + k3;
    const double in3 = 1.0 / sqrt(a.G[(size_t)c3 * a.Ntot + c3]);
    k.st3[k3] = double2{in3, a.Aty[c3] * in3};
  }
}

// ---- 2. threshold start: best support with at most two atoms
__global__ __launch_bounds__(256) void mfx_k3_pairs_kernel(K3Args k) {
  if (k.s.run_if && !*k.s.run_if) return;
  const SolveArgs& a = k.s;
  const long N1 = a.sizes[0], N2 = a.sizes[1], N3 = a.sizes[2];
  const long n12 = N1 * N2, n13 = N1 * N3, n23 = N2 * N3;
  const int N = a.Ntot;
  double best = 0.0;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n12 + n13 + n23; idx += (long)gridDim.x * 256) {
    int p, q;
    if (idx < n12) { p = (int)(idx / N2); q = (int)(N1 + idx % N2); }
    else if (idx < n12 + n13) { const long r = idx - n12; p = (int)(r / N3); q = (int)(N1 + N2 + r % N3); }
    else { const long r = idx - n12 - n13; p = (int)(N1 + r / N3); q = (int)(N1 + N2 + r % N3); }
    best = fmax(best, score2(a.G[(size_t)p * N + p], a.G[(size_t)p * N + q], a.G[(size_t)q * N + q], a.Aty[p], a.Aty[q]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = fmax(best, __shfl_xor(best, o));
  if ((threadIdx.x & 63) == 0) k3_raise(k.thr, best);
}

// ---- 3. the triple screen
struct K3C {     // test constants of one (atom of dictionary 1 or 2, i3)
  double u;      // d . d3 / |d3|
  float pn, qn;  // (P + D) |d'|, (1 - D) Q |d'|
};
typedef unsigned int k3_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ K3C k3_ldc(const K3C* p) {
  const k3_u32x4 r = *(const k3_u32x4*)__builtin_assume_aligned(p, 16);
  K3C c;
  c.u = __longlong_as_double((long long)(((unsigned long long)r[1] << 32) | r[0]));
  c.pn = __uint_as_float(r[2]);
  c.qn = __uint_as_float(r[3]);
  return c;
}

__global__ __launch_bounds__(256) void mfx_k3_screen_kernel(K3Args k) {
  if (k.s.run_if && !*k.s.run_if) return;
  const SolveArgs& a = k.s;
  // [side][atom][i3 in block], 65 KB; the atom stride is padded by one entry: a thread's column constants sit 1 040 bytes
  // apart (without it: 1 024 bytes, a 32-way bank conflict on every read - the screen was bound by the LDS, 2.8 of 4.3 ms)
  __shared__ K3C s_c[2][32][MFX_K3_KB + 1];
  __shared__ double s_aa[2][32], s_ay[2][32];
  __shared__ unsigned short s_q[MFX_K3_QCAP];   // passing triples of the current i3 block: (row << 11) | (column << 6) | i3
  __shared__ int s_qn;
  const int tid = threadIdx.x;
  const int N = a.Ntot;
  const int N1 = (int)a.sizes[0], N2 = (int)a.sizes[1], N3 = (int)a.sizes[2];
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const double* __restrict__ G = a.G;
  const double y_sq = a.ysq[0];
  const double eps_abs = 1e-9 * y_sq;
  // pair data: thread -> column j = tid & 31, rows i = (tid >> 5) + 8 r
  const int jl = tid & 31, il0 = tid >> 5;
  const int j = j0 + jl;
  double a12[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + il0 + 8 * r;
    a12[r] = (i < N1 && j < N2) ? G[(size_t)i * N + N1 + j] : 0.0;
  }
  if (tid < 64) {
    const int side = tid >> 5, al = tid & 31;
    const int col = side ? N1 + j0 + al : i0 + al;
    const bool ok = side ? (j0 + al < N2) : (i0 + al < N1);
    s_aa[side][al] = ok ? G[(size_t)col * N + col] : 0.0;
    s_ay[side][al] = ok ? a.Aty[col] : 0.0;
  }
  if (tid == 0) s_qn = 0;
  __syncthreads();
  double best = 0.0;
  for (int k0 = 0; k0 < N3; k0 += MFX_K3_KB) {
    // ---- constants of the block: 2 x 32 atoms x 64 i3 values, 16 items per thread (coalesced over i3)
    const double T = __longlong_as_double((long long)*(volatile unsigned long long*)k.thr) - eps_abs;
    for (int q = tid; q < 2 * 32 * MFX_K3_KB; q += 256) {
      const int kk = q & (MFX_K3_KB - 1), al = (q >> MFX_K3_KBL) & 31, side = q >> (MFX_K3_KBL + 5);
      const int k3 = k0 + kk;
      K3C c;
      c.u = 0.0; c.pn = 1e18f; c.qn = 0.0f;                     // beyond the dictionary: passes (and is skipped below)
      const int col = side ? N1 + j0 + al : i0 + al;
      const bool ok = (k3 < N3) && (side ? (j0 + al < N2) : (i0 + al < N1));
      if (ok) {
        const int c3 = N1 + N2 + k3;
        const double2 s3 = k.st3[k3];
        const double in3 = s3.x, z3 = s3.y;
        const double u = G[(size_t)col * N + c3] * in3;
        const double n2 = s_aa[side][al] - u * u;                 // |d'|^2
        const double zn = s_ay[side][al] - u * z3;                // d' . y'
        const float Tp = (float)(T - z3 * z3) * (1.0f - 2e-7f);   // what the two projected atoms must reach
        c.u = u;
        if (n2 > 1e-10 * s_aa[side][al]) {
          // |d'| and z = d'.y'/|d'| in FP32 from one reciprocal square root (1 ulp): relative errors <= 3e-7, inside
          // the margin D and the factors below (an FP64 sqrt and division here cost as much as the 64 tests they serve)
          const float n2f = (float)n2, rs = __builtin_amdgcn_rsqf(n2f);
          const float npf = n2f * rs, z = (float)zn * rs;
          const bool always = !(Tp > 0.0f) || (z > 0.0f && z * z >= Tp * (1.0f - 3e-6f));
          const float rth = __builtin_amdgcn_rsqf(fmaxf(Tp, 1e-30f)) * (1.0f + 1e-6f);
          const float P = fminf(1.0f, fmaxf(z, 0.0f) * rth);
          const float Q = __builtin_amdgcn_sqrtf(fmaxf(0.0f, fmaf(-P, P, 1.0f) - 1.2e-7f)) * (1.0f - 3e-7f);
          c.pn = always ? 1e18f : (P + MFX_K3_D) * npf * (1.0f + 6e-7f);
          c.qn = always ? 0.0f : Q * ((1.0f - MFX_K3_D) * npf) * (1.0f - 6e-7f);
        }
      }
      s_c[side][al][kk] = c;
    }
    __syncthreads();
    // ---- the tests: 4 pairs per thread x 64 i3.  What passes goes to a queue in LDS and is scored afterwards by
    // all threads side by side: scored on the spot, one or two lanes of a wave walk through score3 (~150 FP64
    // instructions, three dependent reads of G) while the other 62 idle, in most of the 64 iterations.
    const int nk = min(MFX_K3_KB, N3 - k0);
    auto score_triple = [&](int il, int jq, int kk) {
      const int i = i0 + il, jj = j0 + jq;
      if (i >= N1 || jj >= N2) return;
      const int c3 = N1 + N2 + k0 + kk;
      const double s = score3(s_aa[0][il], G[(size_t)i * N + N1 + jj], G[(size_t)i * N + c3], s_aa[1][jq], G[(size_t)(N1 + jj) * N + c3],
                              G[(size_t)c3 * N + c3], s_ay[0][il], s_ay[1][jq], a.Aty[c3]);
      const double Tn = __longlong_as_double((long long)*(volatile unsigned long long*)k.thr) - eps_abs;
      if (s >= Tn && s > 0.0) {
        const int slot = atomicAdd(&k.ncand[0], 1);
        if (slot < MFX_K3_CAP) {
          k.cand_score[slot] = s;
          k.cand_tuple[slot] = ((long)i * N2 + jj) * N3 + (k0 + kk);   // itertools order: last index fastest (mfx_decode)
        } else {
          k.ncand[1] = 1;   // overflow: the full scan takes over
        }
        if (s > best) { best = s; k3_raise(k.thr, s); }
      }
    };
    for (int kk = 0; kk < nk; ++kk) {
      const K3C c2 = k3_ldc(&s_c[1][jl][kk]);
      unsigned hit = 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const K3C c1 = k3_ldc(&s_c[0][il0 + 8 * r][kk]);
        const float b = fmaf(-c1.qn, c2.qn, fmaf(c1.pn, c2.pn, -(float)fma(-c1.u, c2.u, a12[r])));
        hit |= (b >= 0.0f) ? (1u << r) : 0u;
      }
      if (hit) {
        for (int r = 0; r < 4; ++r) {
          if (!((hit >> r) & 1u)) continue;
          const int slot = atomicAdd(&s_qn, 1);
          if (slot < MFX_K3_QCAP) s_q[slot] = (unsigned short)(((il0 + 8 * r) << 11) | (jl << 6) | kk);
          else score_triple(il0 + 8 * r, jl, kk);   // queue full (threshold still far from the optimum): on the spot
        }
      }
    }
    __syncthreads();
    const int nq = min(s_qn, MFX_K3_QCAP);
    for (int q = tid; q < nq; q += 256) {
      const unsigned e = s_q[q];
      score_triple((int)(e >> 11), (int)((e >> 6) & 31u), (int)(e & 63u));
    }
    __syncthreads();
    if (tid == 0) s_qn = 0;
  }
}

// candidate list -> the generic finalize's view: nblocks = number of candidates (device side)
__global__ void mfx_k3_publish_kernel(K3Args k, int* nblocks_dev) {
  if (k.s.run_if && !*k.s.run_if) return;
  if (threadIdx.x == 0) {
    const int n = k.ncand[0];
    nblocks_dev[0] = (k.ncand[1] || n > MFX_K3_CAP) ? -1 : n;   // -1: overflow, the full scan's per-block results are used
  }
}
