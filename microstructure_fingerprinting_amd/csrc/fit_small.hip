// fit_small.hip -- fused per-voxel kernel for voxels with at most ONE fascicle.
//
// Classes (sub-dictionary sizes, in the order mf.py:391-408 assembles them):
//   K=1: [N], [N,1], [N,E], [N,1,E]      K=0: [1], [E], [1,E]
// which the reference solves with solve_exhaustive_posweights_1 / _2 / _3 (mf_utils.py:225-607).
// There is no Gram matrix to speak of (N x (1+E) inner products per voxel), so this kernel is a direct,
// reference-order evaluation: one thread per atom accumulates its column sums sequentially over the M
// measurements (bit-identical to the reference's loops), evaluates every tuple with the reference's
// closed forms (including _3's explicit residual), and the workgroup takes the lexicographic
// (residual, scan-order) minimum = the reference's strict-'<' first hit.  HBM-latency/launch bound;
// per voxel it reads y and one direction and writes num_params doubles.
#pragma once
#include "mfx_device.h"
#include "nnls_small.h"

#include <type_traits>

#define MFX_SWG 256
#define MFX_NXMAX 16
#define MFX_SATOMS 16   // [N,1,E]: atoms whose columns the exact pass stages in LDS (more: one lane per tuple, table look-ups)
#define MFX_SLIST MFX_SATOMS

struct ExtrasDev {
  int NX;             // active extra columns of this voxel class (csf_i + ear_i * E), <= MFX_NXMAX
  int has_csf, E;     // column 0 = CSF if has_csf; then E EAR columns (E = 0 if the class has no EAR)
  const double* x;    // [M x NX] row-major, voxel independent (mf.py:918-925)
  const double* Gxx;  // [NX x NX] Gram of the extra columns, summed sequentially over the rows
};

struct FitSmallArgs {
  TablesDev T;
  PlanDev P;
  ExtrasDev X;
  const double* Y;
  const double* peaks;
  int peaks_ld;
  const int* vox_list;
  double* params;
  int num_params, maxfasc, csf_on, ear_on;
  int K;  // 0 or 1
};

// np.sum(y**2): NumPy's pairwise reduction (8 accumulators per <=128 block, halving above), used by
// solve_exhaustive_posweights_1 (mf_utils.py:248)
__device__ inline double mfx_np_sumsq_block(const double* a, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += a[i] * a[i];
    return r;
  }
  double r[8];
  int i;
  for (i = 0; i < 8; ++i) r[i] = a[i] * a[i];
  for (i = 8; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a[i + j] * a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i] * a[i];
  return res;
}
__device__ inline double mfx_np_sumsq(const double* a, int n) {
  // explicit-stack version of: n <= 128 ? block : sumsq(a, n2) + sumsq(a + n2, n - n2), n2 = (n/2) & ~7
  int off[12], len[12], state[12];
  double val[12];
  int sp = 0;
  off[0] = 0; len[0] = n; state[0] = 0;
  double ret = 0.0;
  while (sp >= 0) {
    if (len[sp] <= 128) { ret = mfx_np_sumsq_block(a + off[sp], len[sp]); --sp; continue; }
    int n2 = len[sp] / 2;
    n2 -= n2 % 8;
    if (state[sp] == 0) { state[sp] = 1; off[sp + 1] = off[sp]; len[sp + 1] = n2; state[sp + 1] = 0; ++sp; }
    else if (state[sp] == 1) { val[sp] = ret; state[sp] = 2; off[sp + 1] = off[sp] + n2; len[sp + 1] = len[sp] - n2; state[sp + 1] = 0; ++sp; }
    else { ret = val[sp] + ret; --sp; }
  }
  return ret;
}

template <bool BRACKET>
__global__ __launch_bounds__(MFX_SWG) void mfx_fit_small_kernel(FitSmallArgs a) {
  extern __shared__ double smem[];
  const int tid = threadIdx.x;
  const int M = a.P.M, N = a.T.N, ldn = a.T.ldn;
  const int NX = a.X.NX, E = a.X.E, has_csf = a.X.has_csf, K = a.K;
  const int Kp = K + has_csf + (E > 0);
  const double2* __restrict__ tab = a.T.tab;
  const int vox = a.vox_list ? a.vox_list[blockIdx.x] : blockIdx.x;
  double* s_y = smem;                 // [M]
  double* s_t0 = s_y + M;             // [M]
  double* s_t1 = s_t0 + M;            // [M]
  double* s_Yx = s_t1 + M;            // [NXMAX]
  double* s_misc = s_Yx + MFX_NXMAX;  // [8]: y_sq sequential, y_sq pairwise, winner
  double* s_res = s_misc + 8;         // [SWG]
  double* s_w = s_res + MFX_SWG;      // [SWG][3]
  long* s_key = (long*)(s_w + 3 * MFX_SWG);  // [SWG]
  double* s_amin = (double*)(s_key + MFX_SWG);   // [N] ([N,1,E] only) best ranking residual of each atom's tuples
  double* s_col = s_amin + N;                // [SATOMS][M] ([N,1,E]) staged rotated columns of the atoms of the exact pass
  double* s_sum = s_col + MFX_SATOMS * M;    // [SATOMS][32] their column sums
  double* s_xx = s_sum + MFX_SATOMS * 32;    // [M][NX] the extra columns (every row of every column sum reads them: broadcast
                                             // LDS reads; as uniform GLOBAL loads they were 12 vector-memory instructions per row)
  int* s_r0 = (int*)(s_xx + (size_t)M * NX);     // [M]
  int* s_r1 = s_r0 + M;                      // [M]
  int* s_list = s_r1 + M;                    // [SLIST] ([N,1,E]) tuples for the exact pass
  int* s_cnt = s_list + MFX_SLIST;           // [1]

  const double* __restrict__ yv = a.Y + (size_t)vox * M;
  for (int m = tid; m < M; m += MFX_SWG) s_y[m] = yv[m];
  for (int q = tid; q < M * NX; q += MFX_SWG) s_xx[q] = a.X.x[q];
  if (K == 1) {
    const double* pk = a.peaks + (size_t)vox * a.peaks_ld;
    if (tid == 0) mfx_check_dir(a.P, pk, vox);
    for (int m = tid; m < M; m += MFX_SWG) {
      const RowDesc rd = mfx_row_desc(a.T, a.P, m, pk[0], pk[1], pk[2]);
      s_r0[m] = rd.r0; s_t0[m] = rd.t0; s_r1[m] = rd.r1; s_t1[m] = rd.t1;
    }
  }
  __syncthreads();
  if (tid < NX) {  // Adoty of the extra columns, sequential over rows (mf_utils.py:320-325 / 532-535)
    double s = 0.0;
    for (int m = 0; m < M; ++m) s += s_y[m] * a.X.x[(size_t)m * NX + tid];
    s_Yx[tid] = s;
  }
  if (tid == 32) {
    double s = 0.0;
    for (int m = 0; m < M; ++m) s += s_y[m] * s_y[m];
    s_misc[0] = s;
  }
  if (tid == 64) s_misc[1] = mfx_np_sumsq(s_y, M);
  __syncthreads();
  const double y_sq = (Kp == 1) ? s_misc[1] : s_misc[0];  // _1 uses np.sum(y**2), _2/_3 the sequential loop

  auto elem = [&](int m, int n) -> double {
    if (BRACKET) {
      RowDesc rd;
      rd.r0 = s_r0[m]; rd.t0 = s_t0[m]; rd.r1 = s_r1[m]; rd.t1 = s_t1[m];
      return mfx_eval_br(tab, ldn, rd, a.P.tG[m], a.P.dG[m], n);
    }
    return mfx_eval(tab, ldn, s_r0[m], s_t0[m], n);
  };
  const double* xx = s_xx;
  const double* __restrict__ Gxx = a.X.Gxx;

  // thread-local best in the reference's scan order; key < 0 = the reference's initial state
  double bres = y_sq, bw[3] = {0.0, 0.0, 0.0};
  long bkey = -1;
  auto consider = [&](double res, long key, double w0, double w1, double w2) {
    if (res < bres || (res == bres && bkey >= 0 && key < bkey)) { bres = res; bkey = key; bw[0] = w0; bw[1] = w1; bw[2] = w2; }
  };

  // column sums of one rotated atom, sequential over the rows (the reference's loops).  The extra columns go in groups
  // of four with a compile-time count (a run-time column count costs a scalar branch per column and row: 4x the
  // instructions); the values come through uniform (scalar) loads, columns beyond NX are multiplied by zero.
  auto atom_stats_n = [&](auto ngc, int i, double& a11, double& Y1, double* a1x) {
    constexpr int NE = decltype(ngc)::value * 4;
    for (int m = 0; m < M; ++m) {
      const double d = elem(m, i);
      a11 += d * d;
      Y1 += s_y[m] * d;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const double xe = xx[(size_t)m * NX + (e < NX ? e : NX - 1)];
        a1x[e] += d * (e < NX ? xe : 0.0);
      }
    }
  };
  auto atom_stats = [&](int i, double& a11, double& Y1, double* a1x) {
    a11 = 0.0; Y1 = 0.0;
#pragma unroll
    for (int e = 0; e < MFX_NXMAX; ++e) a1x[e] = 0.0;
    if (NX == 0) atom_stats_n(std::integral_constant<int, 0>{}, i, a11, Y1, a1x);
    else if (NX <= 4) atom_stats_n(std::integral_constant<int, 1>{}, i, a11, Y1, a1x);
    else if (NX <= 8) atom_stats_n(std::integral_constant<int, 2>{}, i, a11, Y1, a1x);
    else if (NX <= 12) atom_stats_n(std::integral_constant<int, 3>{}, i, a11, Y1, a1x);
    else atom_stats_n(std::integral_constant<int, 4>{}, i, a11, Y1, a1x);
  };
  if (K == 1 && Kp == 3) {
    // [N,1,E]: i1 = atom, i2 = CSF (single), i3 = EAR atom, scan order i3 -> i1 -> i2.  The reference sums an explicit
    // residual over the M rows for every tuple whose Cramer solution is positive - N E M table look-ups per voxel, ten
    // times the rotation itself.  Two passes instead: every tuple is RANKED with the residual of the same weights from the
    // Gram scalars (|w' G w - 2 w' b + y'y| differs from the explicit sum by its evaluation rounding only, not by the
    // conditioning of the triple), then every tuple of every atom that comes within 1e-7 |y|^2 of the best ranking
    // residual is evaluated in the reference's arithmetic and scan order - the winner and everything tying with it.
    double rmin_t = y_sq;
    for (int i = tid; i < N; i += MFX_SWG) {
      double a11, Y1, a1x[MFX_NXMAX];
      atom_stats(i, a11, Y1, a1x);
      double amin = y_sq;
#pragma unroll
      for (int e = 0; e < MFX_NXMAX - 1; ++e)
        if (e < E) {
          const int ce = 1 + e;
          const double g22 = Gxx[0], g23 = Gxx[ce], g33 = Gxx[ce * NX + ce], y2 = s_Yx[0], y3 = s_Yx[ce];
          double w[3], r;
          auto gram_res = [&](const double* ww) {
            return y_sq + (ww[0] * (ww[0] * a11 + 2.0 * (ww[1] * a1x[0] + ww[2] * a1x[ce] - Y1)) +
                           ww[1] * (ww[1] * g22 + 2.0 * (ww[2] * g23 - y2)) + ww[2] * (ww[2] * g33 - 2.0 * y3));
          };
          nnls3_cramer(y_sq, a11, a1x[0], a1x[ce], g22, g23, g33, Y1, y2, y3, gram_res, w, r);
          amin = fmin(amin, r);
        }
      s_amin[i] = amin;
      rmin_t = fmin(rmin_t, amin);
    }
    s_res[tid] = rmin_t;
    __syncthreads();
    for (int o = MFX_SWG / 2; o > 0; o >>= 1) {
      if (tid < o) s_res[tid] = fmin(s_res[tid], s_res[tid + o]);
      __syncthreads();
    }
    const double cut = s_res[0] + 1e-7 * y_sq;
    if (tid == 0) s_cnt[0] = 0;
    __syncthreads();   // s_res is reused below
    // the atoms whose tuples go through the exact arithmetic
    for (int i = tid; i < N; i += MFX_SWG) {
      if (!(s_amin[i] <= cut)) continue;
      const int q0 = atomicAdd(&s_cnt[0], 1);
      if (q0 < MFX_SATOMS) s_list[q0] = i;
    }
    __syncthreads();
    const int na = s_cnt[0];
    if (na <= MFX_SATOMS) {
      // The usual case, a handful of atoms: the workgroup stages their rotated columns in LDS (the table look-ups of all
      // of them in flight together), 32 lanes per atom take one of the reference's sequential column sums each (|d|^2,
      // d.y, d.x_e: every sum still runs over the rows in order), then one lane per tuple solves the triple and sums its
      // explicit residual from the staged column.  (A lane walking 2 x M table rows per tuple on its own is a chain of
      // L2 round trips: three quarters of this class's time.)
      for (int q = tid; q < na * M; q += MFX_SWG) { const int c = q / M, m = q - c * M; s_col[q] = elem(m, s_list[c]); }
      __syncthreads();
      for (int q = tid; q < na * 32; q += MFX_SWG) {
        const int c = q >> 5, sl = q & 31;
        const double* d = s_col + (size_t)c * M;
        double acc = 0.0;
        if (sl == 0) { for (int m = 0; m < M; ++m) acc += d[m] * d[m]; }
        else if (sl == 1) { for (int m = 0; m < M; ++m) acc += s_y[m] * d[m]; }
        else if (sl - 2 < NX) { const int e = sl - 2; for (int m = 0; m < M; ++m) acc += d[m] * xx[(size_t)m * NX + e]; }
        s_sum[q] = acc;
      }
      __syncthreads();
      for (int q = tid; q < na * E; q += MFX_SWG) {
        const int c = q / E, e = q - c * E, ce = 1 + e, i = s_list[c];
        const double* d = s_col + (size_t)c * M;
        const double a11 = s_sum[c * 32], Y1 = s_sum[c * 32 + 1], a1c = s_sum[c * 32 + 2], a1e = s_sum[c * 32 + 2 + ce];
        double w[3], r;
        auto explicit_res = [&](const double* ww) {
          double rr = 0.0;
          for (int m = 0; m < M; ++m) {
            const double t = (ww[0] * d[m] + ww[1] * xx[(size_t)m * NX] + ww[2] * xx[(size_t)m * NX + ce] - s_y[m]);
            rr += t * t;
          }
          return rr;
        };
        nnls3_cramer(y_sq, a11, a1c, a1e, Gxx[0], Gxx[ce], Gxx[ce * NX + ce], Y1, s_Yx[0], s_Yx[ce], explicit_res, w, r);
        consider(r, (long)e * N + i, w[0], w[1], w[2]);
      }
    } else {   // (a voxel in which many atoms tie, e.g. no fascicle signal at all: every tuple of the qualifying atoms, one per lane)
      for (int q = tid; q < N * E; q += MFX_SWG) {
        const int i = q % N, e = q / N, ce = 1 + e;
        if (!(s_amin[i] <= cut)) continue;
        double a11 = 0.0, Y1 = 0.0, a1c = 0.0, a1e = 0.0;
        for (int m = 0; m < M; ++m) {
          const double d = elem(m, i);
          a11 += d * d;
          Y1 += s_y[m] * d;
          a1c += d * xx[(size_t)m * NX];
          a1e += d * xx[(size_t)m * NX + ce];
        }
        double w[3], r;
        auto explicit_res = [&](const double* ww) {
          double rr = 0.0;
          for (int m = 0; m < M; ++m) {
            const double t = (ww[0] * elem(m, i) + ww[1] * xx[(size_t)m * NX] + ww[2] * xx[(size_t)m * NX + ce] - s_y[m]);
            rr += t * t;
          }
          return rr;
        };
        nnls3_cramer(y_sq, a11, a1c, a1e, Gxx[0], Gxx[ce], Gxx[ce * NX + ce], Y1, s_Yx[0], s_Yx[ce], explicit_res, w, r);
        consider(r, (long)e * N + i, w[0], w[1], w[2]);
      }
    }
  } else if (K == 1) {
    for (int i = tid; i < N; i += MFX_SWG) {
      double a11, Y1, a1x[MFX_NXMAX];
      atom_stats(i, a11, Y1, a1x);
      if (Kp == 1) {
        double w, r;
        nnls1_exact(y_sq, a11, Y1, w, r);
        consider(r, i, w, 0.0, 0.0);
      } else {  // Kp == 2: [N,1] or [N,E]: i1 = atom (outer), i2 = extra column (inner)
#pragma unroll
        for (int e = 0; e < MFX_NXMAX; ++e)
          if (e < NX) {
            double w0, w1, r;
            nnls2_exact(y_sq, a11, a1x[e], Gxx[e * NX + e], Y1, s_Yx[e], w0, w1, r);
            consider(r, (long)i * NX + e, w0, w1, 0.0);
          }
      }
    }
  } else if (tid == 0) {  // K == 0: the dictionary is the extra columns only
    if (Kp == 1) {
      for (int e = 0; e < NX; ++e) {
        double w, r;
        nnls1_exact(y_sq, Gxx[e * NX + e], s_Yx[e], w, r);
        consider(r, e, w, 0.0, 0.0);
      }
    } else if (Kp == 2) {  // [1,E]
      for (int e = 0; e < E; ++e) {
        double w0, w1, r;
        nnls2_exact(y_sq, Gxx[0], Gxx[1 + e], Gxx[(1 + e) * NX + 1 + e], s_Yx[0], s_Yx[1 + e], w0, w1, r);
        consider(r, e, w0, w1, 0.0);
      }
    }
  }
  // workgroup minimum of (residual, scan key) with consider()'s rule, as a tree (a serial fold over the 256 thread results
  // by one thread was 4 us of every voxel); the owner of the winning pair then publishes its weights
  s_res[tid] = bres;
  s_key[tid] = bkey;
  __syncthreads();
  for (int o = MFX_SWG / 2; o > 0; o >>= 1) {
    if (tid < o) {
      const double r1 = s_res[tid], r2 = s_res[tid + o];
      const long k1 = s_key[tid], k2 = s_key[tid + o];
      if (r2 < r1 || (r2 == r1 && k1 >= 0 && k2 < k1)) { s_res[tid] = r2; s_key[tid] = k2; }
    }
    __syncthreads();
  }
  if (bres == s_res[0] && bkey == s_key[0]) {   // (keys of real tuples are unique; several threads in the initial state write the same zeros)
    s_misc[2] = bres; s_misc[3] = bw[0]; s_misc[4] = bw[1]; s_misc[5] = bw[2];
    ((long*)s_misc)[6] = bkey;
  }
  __syncthreads();
  const double res = s_misc[2];
  const double w0 = s_misc[3], w1 = s_misc[4], w2 = s_misc[5];
  const long key = ((long*)s_misc)[6];
  // decode the winning tuple: atom index, extra-column indices
  int ia = 0, ic1 = -1, ic2 = -1;  // rotated atom, first extra column, second extra column (indices into x)
  int id_ear = 0;
  if (key >= 0) {
    if (K == 1) {
      if (Kp == 1) ia = (int)key;
      else if (Kp == 2) { ia = (int)(key / NX); ic1 = (int)(key % NX); id_ear = has_csf ? 0 : ic1; }
      else { ia = (int)(key % N); ic1 = 0; ic2 = 1 + (int)(key / N); id_ear = (int)(key / N); }
    } else {
      if (Kp == 1) { ic1 = (int)key; id_ear = has_csf ? 0 : ic1; }
      else { ic1 = 0; ic2 = 1 + (int)key; id_ear = (int)key; }
    }
  } else {
    // reference's initial state: indices 0 in every sub-dictionary, w = 0
    if (K == 1) { ic1 = (Kp >= 2) ? 0 : -1; ic2 = (Kp == 3) ? 1 : -1; }
    else { ic1 = 0; ic2 = (Kp == 2) ? 1 : -1; }
  }
  // weights in sub-dictionary order [fascicle, csf, ear]
  double wv[3] = {w0, w1, w2};
  // y_rec = A[:, tot] @ w (w * column for one sub-dictionary) and R^2
  if (tid < 64) {
    const int lane = tid;
    double sy = 0.0, sr = 0.0;
    double* s_yrec = s_res;  // reuse
    for (int m = lane; m < M; m += 64) {
      double yr;
      if (K == 1) {
        yr = wv[0] * elem(m, ia);
        if (Kp == 1) { /* w * column */ }
        else { yr = elem(m, ia) * wv[0] + xx[(size_t)m * NX + ic1] * wv[1]; if (Kp == 3) yr += xx[(size_t)m * NX + ic2] * wv[2]; }
      } else {
        yr = wv[0] * xx[(size_t)m * NX + ic1];
        if (Kp == 2) yr = xx[(size_t)m * NX + ic1] * wv[0] + xx[(size_t)m * NX + ic2] * wv[1];
      }
      s_yrec[m] = yr;
      sy += s_y[m];
      sr += yr;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sy += __shfl_xor(sy, o); sr += __shfl_xor(sr, o); }
    sy /= M; sr /= M;
    double cyy = 0.0, crr = 0.0, cyr = 0.0;
    for (int m = lane; m < M; m += 64) {
      const double da = s_y[m] - sy, db = s_yrec[m] - sr;
      cyy += da * da; crr += db * db; cyr += da * db;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cyy += __shfl_xor(cyy, o); crr += __shfl_xor(crr, o); cyr += __shfl_xor(cyr, o); }
    double r2 = 0.0;
    if (M > 1 && cyy > 0.0 && crr > 0.0) {
      const double f = (double)(M - 1);
      double r = (cyr / f) / sqrt(cyy / f) / sqrt(crr / f);
      r = r > 1.0 ? 1.0 : (r < -1.0 ? -1.0 : r);
      r2 = r * r;
    }
    if (lane == 0) {  // params packing, mf.py:420-450
      double* out = a.params + (size_t)vox * a.num_params;
      double M0 = 0.0;
      for (int k = 0; k < Kp; ++k) M0 += wv[k];
      double nu[3];
      for (int k = 0; k < 3; ++k) nu[k] = (fabs(M0) > 0) ? wv[k] / M0 : wv[k];
      const int i_csf = 2 * a.maxfasc + 1, i_ear = 2 * a.maxfasc + a.csf_on + 1;
      out[0] = M0;
      if (K == 1) { out[1] = nu[0]; out[1 + a.maxfasc] = (double)ia; }
      if (has_csf) out[i_csf] = nu[K];
      if (E > 0) { out[i_ear] = nu[K + has_csf]; out[i_ear + 1] = (double)id_ear; }
      out[a.num_params - 2] = res / M;
      out[a.num_params - 1] = r2;
    }
  }
}
