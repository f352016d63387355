// solve_generic.hip -- explicit-dictionary solver: solve_exhaustive_posweights(A, y, dicsizes)
// (mf_utils.py:115-214) for ONE problem with arbitrary sub-dictionary sizes.
//
// The fused fit kernels never see an explicit dictionary; this path backs the drop-in mf_utils entry
// point (and reproduces the reference's own solver tests).  Three launches:
//   1. mfx_gram_kernel     G = A^T A and A^T y, every entry summed sequentially over the rows exactly as
//                          the reference's precompute loops (mf_utils.py:307-325, 503-535)  -> HBM
//   2. mfx_tuple_scan      all prod(sizes) index tuples, one thread per tuple (grid-stride); ranks by
//                          the NNLS optimum of the tuple's tiny Gram system; per-block best -> HBM
//   3. mfx_tuple_finalize  candidates within 1e-9*||y||^2 of the best are re-evaluated with the
//                          reference's exact per-tuple arithmetic (_1/_2 closed forms, _3 Cramer +
//                          explicit residual, _4up active-set optimum + explicit residual), first-hit
//                          rule in the reference's scan order, outputs written.
// HBM-bound gather work; meant for correctness and convenience, not for the 1e5-voxel loop.
#pragma once
#include "fit_small.hip"  // mfx_np_sumsq
#include "nnls_small.h"

#define MFX_GK 8  // max sub-dictionaries supported by the explicit solver

struct SolveArgs {
  const double* A;  // [M x lda] device copy
  long lda;
  int M, Kp, Ntot;
  long sizes[MFX_GK], start[MFX_GK];
  const double* y;  // [M]
  double* G;        // [Ntot x Ntot]
  double* Aty;      // [Ntot]
  double* ysq;      // [2]: sequential, pairwise
  long ntuples;
  double* blk_score;  // [nblocks]
  long* blk_tuple;    // [nblocks]
  int nblocks;
  // three-dictionary fast path (solve_k3.hip); all null / 0 otherwise
  const int* nblocks_dev;   // device-side number of entries in blk_score / blk_tuple (< 0: use nblocks)
  const int* scan_enable;   // the full scan below runs only when this device flag is set
  int gram_ranking_only;    // G was summed on the matrix pipe: the finalize stage re-sums what it needs sequentially
  const int* run_if;        // null, or a device flag: every kernel of the launch sequence exits at once while it is 0 (fit_k3.hip
                            // enqueues this path for every voxel of a batch as the fallback of a candidate-list overflow)
  // outputs
  double* w;       // [Kp]
  long* sub;       // [Kp]
  double* minobj;  // [1]
  double* yrec;    // [M]
};

__global__ void mfx_gram_kernel(SolveArgs a) {
  if (a.run_if && !*a.run_if) return;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nn = (long)a.Ntot * a.Ntot;
  if (idx < nn) {
    const int p = (int)(idx / a.Ntot), q = (int)(idx - (long)p * a.Ntot);
    double s = 0.0;
    for (int k = 0; k < a.M; ++k) s += a.A[k * a.lda + p] * a.A[k * a.lda + q];
    a.G[idx] = s;
  } else if (idx < nn + a.Ntot) {
    const int p = (int)(idx - nn);
    double s = 0.0;
    for (int k = 0; k < a.M; ++k) s += a.y[k] * a.A[k * a.lda + p];
    a.Aty[p] = s;
  } else if (idx == nn + a.Ntot) {
    double s = 0.0;
    for (int k = 0; k < a.M; ++k) s += a.y[k] * a.y[k];
    a.ysq[0] = s;
  } else if (idx == nn + a.Ntot + 1) {
    a.ysq[1] = mfx_np_sumsq(a.y, a.M);
  }
}

// decode tuple number t (itertools.product order: last index fastest) into absolute column indices
__device__ __forceinline__ void mfx_decode(const SolveArgs& a, long t, int col[MFX_GK]) {
  for (int k = a.Kp - 1; k >= 0; --k) {
    const long s = a.sizes[k];
    col[k] = (int)(a.start[k] + t % s);
    t /= s;
  }
}

// general feasible-support NNLS score for n <= MFX_GK columns (n >= 5 only; smaller n use closed forms)
__device__ inline double mfx_score_general(int n, const double* g /* n x n */, const double* y) {
  double best = 0.0;
  for (int mask = 1; mask < (1 << n); ++mask) {
    int idx[MFX_GK], c = 0;
    for (int k = 0; k < n; ++k)
      if (mask & (1 << k)) idx[c++] = k;
    double Am[MFX_GK][MFX_GK + 1];
    for (int p = 0; p < c; ++p) {
      for (int q = 0; q < c; ++q) Am[p][q] = g[idx[p] * n + idx[q]];
      Am[p][MFX_GK] = y[idx[p]];
    }
    bool ok = true;
    for (int p = 0; p < c && ok; ++p) {
      if (!(Am[p][p] > 1e-13 * g[idx[p] * n + idx[p]])) { ok = false; break; }
      for (int r = p + 1; r < c; ++r) {
        const double f = Am[r][p] / Am[p][p];
        for (int q = p; q < c; ++q) Am[r][q] -= f * Am[p][q];
        Am[r][MFX_GK] -= f * Am[p][MFX_GK];
      }
    }
    if (!ok) continue;
    double ws[MFX_GK], sc = 0.0;
    for (int p = c - 1; p >= 0; --p) {
      double t = Am[p][MFX_GK];
      for (int q = p + 1; q < c; ++q) t -= Am[p][q] * ws[q];
      ws[p] = t / Am[p][p];
    }
    for (int p = 0; p < c; ++p) { if (!(ws[p] >= 0.0)) ok = false; sc += ws[p] * y[idx[p]]; }
    if (ok && sc > best) best = sc;
  }
  return best;
}

__device__ inline double mfx_tuple_score(const SolveArgs& a, const int col[MFX_GK]) {
  const int n = a.Kp, N = a.Ntot;
  const double* G = a.G;
  if (n == 1) return score1(G[(long)col[0] * N + col[0]], a.Aty[col[0]]);
  if (n == 2)
    return score2(G[(long)col[0] * N + col[0]], G[(long)col[0] * N + col[1]], G[(long)col[1] * N + col[1]], a.Aty[col[0]],
                  a.Aty[col[1]]);
  if (n == 3)
    return score3(G[(long)col[0] * N + col[0]], G[(long)col[0] * N + col[1]], G[(long)col[0] * N + col[2]],
                  G[(long)col[1] * N + col[1]], G[(long)col[1] * N + col[2]], G[(long)col[2] * N + col[2]], a.Aty[col[0]],
                  a.Aty[col[1]], a.Aty[col[2]]);
  if (n == 4) {
    double g[10], y[4];
    int q = 0;
    for (int p = 0; p < 4; ++p) {
      y[p] = a.Aty[col[p]];
      for (int r = p; r < 4; ++r) g[q++] = G[(long)col[p] * N + col[r]];
    }
    return score4(g, y);
  }
  double g[MFX_GK * MFX_GK], y[MFX_GK];
  for (int p = 0; p < n; ++p) {
    y[p] = a.Aty[col[p]];
    for (int r = 0; r < n; ++r) g[p * n + r] = G[(long)col[p] * N + col[r]];
  }
  return mfx_score_general(n, g, y);
}

__global__ __launch_bounds__(256) void mfx_tuple_scan(SolveArgs a) {
  __shared__ double s_sc[256];
  __shared__ long s_t[256];
  if (a.run_if && !*a.run_if) return;
  if (a.scan_enable && !*a.scan_enable) return;   // (three-dictionary fast path: only after a candidate-list overflow)
  double best = 0.0;
  long bt = -1;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < a.ntuples; t += (long)gridDim.x * 256) {
    int col[MFX_GK];
    mfx_decode(a, t, col);
    const double s = mfx_tuple_score(a, col);
    if (s > best) { best = s; bt = t; }
  }
  s_sc[threadIdx.x] = best;
  s_t[threadIdx.x] = bt;
  __syncthreads();
  // keep EVERY thread's best that is within the tie window of the block maximum?  One per block is
  // enough for generic data; exact ties inside a block resolve to the smaller tuple number below.
  if (threadIdx.x == 0) {
    for (int i = 1; i < 256; ++i)
      if (s_sc[i] > best || (s_sc[i] == best && s_t[i] >= 0 && (bt < 0 || s_t[i] < bt))) { best = s_sc[i]; bt = s_t[i]; }
    a.blk_score[blockIdx.x] = best;
    a.blk_tuple[blockIdx.x] = bt;
  }
}

// reference scan-order key of a tuple: _3 iterates i3 -> i1 -> i2 (mf_utils.py:540-547), every other
// kernel in itertools/lexicographic order
__device__ __forceinline__ long mfx_order_key(const SolveArgs& a, long t) {
  if (a.Kp != 3) return t;
  const long i3 = t % a.sizes[2], i2 = (t / a.sizes[2]) % a.sizes[1], i1 = t / (a.sizes[2] * a.sizes[1]);
  return (i3 * a.sizes[0] + i1) * a.sizes[1] + i2;
}

__global__ __launch_bounds__(256) void mfx_tuple_finalize(SolveArgs a) {
  if (a.run_if && !*a.run_if) return;
  __shared__ double s_res[256];
  __shared__ long s_key[256], s_tt[256];
  __shared__ double s_w[256][MFX_GK];
  __shared__ double s_max;
  const int tid = threadIdx.x;
  const int n = a.Kp, N = a.Ntot, M = a.M;
  const double y_sq = (n == 1 || n >= 4) ? a.ysq[1] : a.ysq[0];  // np.sum(y**2) in _1/_4up, sequential in _2/_3
  const int nb = (a.nblocks_dev && *a.nblocks_dev >= 0) ? *a.nblocks_dev : a.nblocks;
  {
    double mx = 0.0;
    for (int b = tid; b < nb; b += 256) mx = fmax(mx, a.blk_score[b]);
    s_res[tid] = mx;
    __syncthreads();
    if (tid == 0) {
      for (int i = 1; i < 256; ++i) mx = fmax(mx, s_res[i]);
      s_max = mx;
    }
  }
  __syncthreads();
  const double thr = s_max - 1e-9 * y_sq;
  double bres = INFINITY, bw[MFX_GK];
  long bkey = -1, btt = -1;
  for (int k = 0; k < MFX_GK; ++k) bw[k] = 0.0;
  for (int b = tid; b < nb; b += 256) {
    const long t = a.blk_tuple[b];
    if (t < 0 || a.blk_score[b] < thr) continue;
    int col[MFX_GK];
    mfx_decode(a, t, col);
    double w[MFX_GK], res;
    for (int k = 0; k < MFX_GK; ++k) w[k] = 0.0;
    const double* G = a.G;
    if (n == 1) {
      nnls1_exact(y_sq, G[(long)col[0] * N + col[0]], a.Aty[col[0]], w[0], res);
    } else if (n == 2) {
      nnls2_exact(y_sq, G[(long)col[0] * N + col[0]], G[(long)col[0] * N + col[1]], G[(long)col[1] * N + col[1]],
                  a.Aty[col[0]], a.Aty[col[1]], w[0], w[1], res);
    } else if (n == 3) {
      auto explicit_res = [&](const double* ww) {
        double rr = 0.0;
        for (int k = 0; k < M; ++k) {
          const double t3 = (ww[0] * a.A[k * a.lda + col[0]] + ww[1] * a.A[k * a.lda + col[1]] + ww[2] * a.A[k * a.lda + col[2]] - a.y[k]);
          rr += t3 * t3;
        }
        return rr;
      };
      if (a.gram_ranking_only) {
        // the Gram came from the matrix pipe: the reference's sequential sums (mf_utils.py:503-535) for this triple
        double g11 = 0, g12 = 0, g13 = 0, g22 = 0, g23 = 0, g33 = 0, y1 = 0, y2 = 0, y3 = 0;
        for (int k = 0; k < M; ++k) {
          const double d1 = a.A[k * a.lda + col[0]], d2 = a.A[k * a.lda + col[1]], d3 = a.A[k * a.lda + col[2]], yk = a.y[k];
          g11 += d1 * d1; g22 += d2 * d2; g33 += d3 * d3; g12 += d1 * d2; g13 += d1 * d3; g23 += d2 * d3;
          y1 += yk * d1; y2 += yk * d2; y3 += yk * d3;
        }
        nnls3_cramer(y_sq, g11, g12, g13, g22, g23, g33, y1, y2, y3, explicit_res, w, res);
      } else
      nnls3_cramer(y_sq, G[(long)col[0] * N + col[0]], G[(long)col[0] * N + col[1]], G[(long)col[0] * N + col[2]],
                   G[(long)col[1] * N + col[1]], G[(long)col[1] * N + col[2]], G[(long)col[2] * N + col[2]], a.Aty[col[0]],
                   a.Aty[col[1]], a.Aty[col[2]], explicit_res, w, res);
    } else {
      // _4up: NNLS optimum from the Gram (feasible-support enumeration), residual explicitly
      double g[MFX_GK * MFX_GK], yy[MFX_GK];
      for (int p = 0; p < n; ++p) {
        yy[p] = a.Aty[col[p]];
        for (int r = 0; r < n; ++r) g[p * n + r] = G[(long)col[p] * N + col[r]];
      }
      double best = 0.0;
      for (int mask = 1; mask < (1 << n); ++mask) {
        int idx[MFX_GK], c = 0;
        for (int k = 0; k < n; ++k)
          if (mask & (1 << k)) idx[c++] = k;
        double Am[MFX_GK][MFX_GK + 1];
        for (int p = 0; p < c; ++p) {
          for (int q = 0; q < c; ++q) Am[p][q] = g[idx[p] * n + idx[q]];
          Am[p][MFX_GK] = yy[idx[p]];
        }
        bool ok = true;
        for (int p = 0; p < c && ok; ++p) {
          if (!(Am[p][p] > 1e-13 * g[idx[p] * n + idx[p]])) { ok = false; break; }
          for (int r = p + 1; r < c; ++r) {
            const double f = Am[r][p] / Am[p][p];
            for (int q = p; q < c; ++q) Am[r][q] -= f * Am[p][q];
            Am[r][MFX_GK] -= f * Am[p][MFX_GK];
          }
        }
        if (!ok) continue;
        double ws[MFX_GK], sc = 0.0;
        for (int p = c - 1; p >= 0; --p) {
          double tt = Am[p][MFX_GK];
          for (int q = p + 1; q < c; ++q) tt -= Am[p][q] * ws[q];
          ws[p] = tt / Am[p][p];
        }
        for (int p = 0; p < c; ++p) { if (!(ws[p] >= 0.0)) ok = false; sc += ws[p] * yy[idx[p]]; }
        if (ok && sc > best) {
          best = sc;
          for (int k = 0; k < n; ++k) w[k] = 0.0;
          for (int p = 0; p < c; ++p) w[idx[p]] = ws[p];
        }
      }
      double rr = 0.0;
      for (int k = 0; k < M; ++k) {
        double tt = -a.y[k];
        for (int p = 0; p < n; ++p) tt += w[p] * a.A[k * a.lda + col[p]];
        rr += tt * tt;
      }
      res = rr;
    }
    const long key = mfx_order_key(a, t);
    if (res < bres || (res == bres && key < bkey)) {
      bres = res; bkey = key; btt = t;
      for (int k = 0; k < MFX_GK; ++k) bw[k] = w[k];
    }
  }
  s_res[tid] = bres;
  s_key[tid] = bkey;
  s_tt[tid] = btt;
  for (int k = 0; k < MFX_GK; ++k) s_w[tid][k] = bw[k];
  __syncthreads();
  if (tid == 0) {
    // initial state of the reference: min_obj = ||y||^2, w = 0, every index 0; strict '<'
    double br = y_sq;
    long bk = -1, bt = -1;
    int bi = -1;
    for (int i = 0; i < 256; ++i) {
      if (s_key[i] < 0) continue;
      if (s_res[i] < br || (s_res[i] == br && bk >= 0 && s_key[i] < bk)) { br = s_res[i]; bk = s_key[i]; bt = s_tt[i]; bi = i; }
    }
    int col[MFX_GK];
    mfx_decode(a, bt < 0 ? 0 : bt, col);
    for (int k = 0; k < n; ++k) {
      a.w[k] = (bi >= 0) ? s_w[bi][k] : 0.0;
      a.sub[k] = col[k] - a.start[k];
    }
    a.minobj[0] = br;
    s_tt[0] = bt < 0 ? 0 : bt;
    s_key[0] = bi;
  }
  __syncthreads();
  {  // y_recons = A[:, tot] @ w   (w * column for one sub-dictionary, mf_utils.py:277)
    int col[MFX_GK];
    mfx_decode(a, s_tt[0], col);
    for (int k = tid; k < M; k += 256) {
      double t = 0.0;
      if (n == 1) t = a.w[0] * a.A[k * a.lda + col[0]];
      else
        for (int p = 0; p < n; ++p) t += a.A[k * a.lda + col[p]] * a.w[p];
      a.yrec[k] = t;
    }
  }
}


// ---- params packing for the voxel loop's generic classes (mf.py:420-450) from the solver's outputs: one 64-thread workgroup
struct PackArgs {
  const int* run_if;   // null, or a device flag: the kernel exits at once while it is 0
  const double* w; const long* sub; const double* minobj; const double* yrec; const double* y;
  int M, K, has_csf, E, maxfasc, csf_on, ear_on, num_params;
  double* out;   // the voxel's params row
};
// (one wave: the 64 lanes of the calling wave)
__device__ __forceinline__ void mfx_pack_params_body(const PackArgs& a) {
  const int lane = threadIdx.x & 63, M = a.M;
  const int Kp = a.K + a.has_csf + (a.E > 0);
  double sy = 0.0, sr = 0.0;
  for (int m = lane; m < M; m += 64) { sy += a.y[m]; sr += a.yrec[m]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sy += __shfl_xor(sy, o); sr += __shfl_xor(sr, o); }
  sy /= M; sr /= M;
  double cyy = 0.0, crr = 0.0, cyr = 0.0;
  for (int m = lane; m < M; m += 64) {
    const double da = a.y[m] - sy, db = a.yrec[m] - sr;
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
  if (lane == 0) {
    double M0 = 0.0;
    for (int k = 0; k < Kp; ++k) M0 += a.w[k];
    for (int k = 0; k < a.K; ++k) {
      a.out[1 + k] = (fabs(M0) > 0) ? a.w[k] / M0 : a.w[k];
      a.out[1 + a.maxfasc + k] = (double)a.sub[k];
    }
    a.out[0] = M0;
    const int i_csf = 2 * a.maxfasc + 1, i_ear = 2 * a.maxfasc + a.csf_on + 1;
    if (a.has_csf) a.out[i_csf] = (fabs(M0) > 0) ? a.w[a.K] / M0 : a.w[a.K];
    if (a.E > 0) {
      a.out[i_ear] = (fabs(M0) > 0) ? a.w[a.K + a.has_csf] / M0 : a.w[a.K + a.has_csf];
      a.out[i_ear + 1] = (double)a.sub[a.K + a.has_csf];
    }
    a.out[a.num_params - 2] = a.minobj[0] / M;
    a.out[a.num_params - 1] = r2;
  }
}
__global__ __launch_bounds__(64) void mfx_pack_params_kernel(PackArgs a) {
  if (a.run_if && !*a.run_if) return;
  mfx_pack_params_body(a);
}
