// fit_k2x.hip -- fused per-voxel kernel for two fascicles PLUS voxel-independent compartments:
//   [N,N,1] (CSF), [N,N,E] (EAR)          -> solve_exhaustive_posweights_3   (mf_utils.py:470-607)
//   [N,N,1,E] (CSF + EAR)                 -> solve_exhaustive_posweights_4up (mf_utils.py:612-657)
// Same skeleton as fit_k2.hip (one workgroup per voxel, D1 tile in registers, D2 tiles through LDS, FP64
// MFMA cross-Gram), but 16-atom chunks and a heavier epilogue: every (i1,i2) accumulator entry is
// combined with each extra column (3x3 / 4x4 normal equations, feasible-support ranking); the inner
// products of the rotated atoms with the extra columns are computed once per voxel into a per-workgroup
// HBM scratch slab and staged through LDS.  Short-listed tuples are re-evaluated with the reference's
// exact arithmetic (Cramer + explicit residual for _3; Gram-based active-set optimum + explicit residual
// for _4up, whose reference is the third-party scipy.optimize.nnls).  VALU-bound (the epilogue dominates).
//
// What is ranked where (no candidate is ever dropped silently):
//   * the pair scan ranks, per (i1,i2), only the supports that contain BOTH fascicle atoms ({1,2}, {1,2,f}, {1,2,x_t},
//     {1,2,f,x_t}); one best per (lane,row) slot goes to the short list, and a slot whose runner-up is also within
//     rounding distance of the optimum sends its whole slot row to the exact stage;
//   * supports with ONE fascicle atom are scored per atom (U_k[n][t], from the per-atom inner products), those with none
//     per extra tuple (Q[t]): when such a support is within rounding distance of the optimum, EVERY tuple sharing its
//     active atoms ties (a voxel fitted by one fascicle + CSF, a pure CSF voxel ...), so the exact stage evaluates the
//     whole family in the reference's arithmetic and scan order - as the reference's strict-'<' first hit demands -
//     instead of letting the ties flood the short list (one entry per slot and round: that list used to overflow and
//     drop entries silently).  When NO fascicle atom is active the tuples (i1, i2, t) of a given t all get the very
//     same arithmetic in the reference (its 2-variable fall-backs then depend on the extra column only), so the first
//     of them, (0, 0, t), stands for all; any tuple in which an atom does help is in the pair list or an atom family;
//   * every short-listed pair is evaluated exactly for ALL its extra tuples t (they tie when the extra column is inactive);
//   * a short list that still overflows triggers the exhaustive exact pass over all N*N*ntup tuples (slow, exact by
//     construction, counted in the launch's overflow counter).
#pragma once
#include "fit_small.hip"  // ExtrasDev, mfx_np_sumsq
#include "mfx_device.h"
#include "nnls_small.h"

typedef double d4x __attribute__((ext_vector_type(4)));

#define MFX_XWG 512
#define MFX_XMAXC 256
#define MFX_XS 16  // stride of the extras dimension in scratch/LDS

struct FitK2XArgs {
  TablesDev T;
  PlanDev P;
  ExtrasDev X;
  const double* Y;
  const double* peaks;
  int peaks_ld;
  const int* vox_list;
  double* params;
  double* ws;  // [gridDim.x][2][NP][MFX_XS] scratch: atom . extra-column inner products
  int num_params, maxfasc, csf_on, ear_on;
  int vox_base;  // first voxel (or first vox_list entry) of this launch
  int maxc;      // short-list size beyond which the exhaustive exact pass runs (MFX_XMAXC; tests lower it)
  int* ovf_count;  // [4] launch counters: [0] voxels that took the exhaustive pass, [2] short-listed pairs, [3] family items
};

#define MFX_XFAM 64   // family items (see below) per voxel before the exhaustive pass takes over
struct FamX {
  int type;   // 1: (i, all j, t)   2: (all i, j, t)   3: (0, 0, t): no fascicle atom active   4: (i, j = lc mod 16, all t)
  int a, t;
};

struct CandX {
  double score;
  int i, j, e, pad;
};

// NW waves per workgroup and NBUF LDS chunk buffers: (8, 2) for M <= 200, (4, 1) for long protocols
// (one wave per SIMD owns the 512-register file; see fit_k2.hip).
template <int KSTEPS, bool BRACKET, int NW = 8, int NBUF = 2>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void mfx_fit_k2x_kernel(FitK2XArgs a) {
  constexpr int MP = KSTEPS * 4;
  constexpr int WG = NW * 64;
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lg = lane >> 4, lc = lane & 15;
  const int M = a.P.M, N = a.T.N, ldn = a.T.ldn, NP = ldn, ntiles = NP >> 4;
  const int NX = a.X.NX, E = a.X.E, has_csf = a.X.has_csf;
  const int Kp = 2 + has_csf + (E > 0);
  const int ntup = (Kp == 3) ? NX : E;  // extra tuples per (i1,i2)
  const double2* __restrict__ tab = a.T.tab;
  const int vox = a.vox_list ? a.vox_list[a.vox_base + blockIdx.x] : a.vox_base + blockIdx.x;
  double* __restrict__ wsA = a.ws + (size_t)blockIdx.x * 2 * NP * MFX_XS;  // [k][n][e]

  double* sB = smem;                              // [NBUF][MP][16]
  double* s_y = sB + NBUF * MP * 16;              // [MP]
  double* s_t0 = s_y + MP;                        // [2][MP]
  double* s_t1 = s_t0 + 2 * MP;                   // [2][MP] (bracket)
  double* s_tG = s_t1 + (BRACKET ? 2 * MP : 0);
  double* s_dG = s_tG + (BRACKET ? MP : 0);
  double* s_A11 = s_dG + (BRACKET ? MP : 0);      // [NP] x4
  double* s_Y1 = s_A11 + NP;
  double* s_A22 = s_Y1 + NP;
  double* s_Y2 = s_A22 + NP;
  double* s_a1x = s_Y2 + NP;                      // [8 waves][16 rows][XS]
  double* s_a2x = s_a1x + NW * 16 * MFX_XS;       // [2 buf][16 cols][XS]
  double* s_Yx = s_a2x + 2 * 16 * MFX_XS;         // [XS]
  double* s_Gxx = s_Yx + MFX_XS;                  // [XS][XS]
  double* s_red = s_Gxx + MFX_XS * MFX_XS;        // [32]
  double* s_Qx = s_red + 32;                      // [XS]            best support made of extra columns only, per extra tuple
  CandX* s_cand = (CandX*)(s_Qx + MFX_XS);        // [XMAXC]
  FamX* s_fam = (FamX*)(s_cand + MFX_XMAXC);      // [XFAM]
  int* s_r0 = (int*)(s_fam + MFX_XFAM);           // [2][MP]
  int* s_r1 = s_r0 + 2 * MP;
  int* s_cnt = s_r1 + (BRACKET ? 2 * MP : 0);

  const double* __restrict__ yv = a.Y + (size_t)vox * M;
  const double* __restrict__ pk = a.peaks + (size_t)vox * a.peaks_ld;
  const double* __restrict__ xx = a.X.x;
  for (int m = tid; m < MP; m += WG) s_y[m] = (m < M) ? yv[m] : 0.0;
  for (int idx = tid; idx < 2 * MP; idx += WG) {
    const int k = idx / MP, m = idx - k * MP;
    RowDesc rd;
    rd.r0 = a.T.P; rd.t0 = 0.0; rd.r1 = -1; rd.t1 = 0.0;
    if (m < M) rd = mfx_row_desc(a.T, a.P, m, pk[3 * k], pk[3 * k + 1], pk[3 * k + 2]);
    s_r0[idx] = rd.r0;
    s_t0[idx] = rd.t0;
    if (BRACKET) {
      s_r1[idx] = rd.r1;
      s_t1[idx] = rd.t1;
      if (k == 0) { s_tG[m] = (m < M) ? a.P.tG[m] : 0.0; s_dG[m] = (m < M) ? a.P.dG[m] : 1.0; }
    }
  }
  for (int q = tid; q < MFX_XS * MFX_XS; q += WG) {
    const int p = q / MFX_XS, r = q - p * MFX_XS;
    s_Gxx[q] = (p < NX && r < NX) ? a.X.Gxx[p * NX + r] : 0.0;
  }
  if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
  if (tid < 2) mfx_check_dir(a.P, pk + 3 * tid, vox);
  __syncthreads();
  if (tid < MFX_XS) {
    double s = 0.0;
    if (tid < NX)
      for (int m = 0; m < M; ++m) s += s_y[m] * xx[(size_t)m * NX + tid];
    s_Yx[tid] = s;
  }
  if (tid == 64) s_red[16] = mfx_np_sumsq(s_y, M);  // _4up: min_obj starts at np.sum(y**2)
  __syncthreads();
  const bool HASF = (Kp == 4);   // a fixed CSF column (index 0) besides the varying extra column
  const int x0 = HASF ? 1 : 0;   // first varying extra column
  if (tid < ntup) {              // supports made of extra columns only
    const int cx = x0 + tid;
    double q = pos1(s_Gxx[cx * MFX_XS + cx], s_Yx[cx]);
    if (HASF) q = fmax(q, fmax(pos1(s_Gxx[0], s_Yx[0]), pos2(s_Gxx[0], s_Gxx[cx], s_Gxx[cx * MFX_XS + cx], s_Yx[0], s_Yx[cx])));
    s_Qx[tid] = q;
  }
  const double eps_abs_of = 1e-9;

  auto elem = [&](int k, int m, int n) -> double {
    if (BRACKET) {
      RowDesc rd;
      rd.r0 = s_r0[k * MP + m]; rd.t0 = s_t0[k * MP + m];
      rd.r1 = s_r1[k * MP + m]; rd.t1 = s_t1[k * MP + m];
      return mfx_eval_br(tab, ldn, rd, s_tG[m], s_dG[m], n);
    } else {
      return mfx_eval(tab, ldn, s_r0[k * MP + m], s_t0[k * MP + m], n);
    }
  };

  __syncthreads();   // s_Qx
  // best score over the supports {atom} + subset of {fixed, x_t}: every support with exactly ONE fascicle atom
  auto atom_best = [&](double a11, double y1, const double* ax /* atom . extras */, int t) -> double {
    const int cx = x0 + t;
    double r = fmax(pos1(a11, y1), pos2(a11, ax[cx], s_Gxx[cx * MFX_XS + cx], y1, s_Yx[cx]));
    if (HASF) r = fmax(r, fmax(pos2(a11, ax[0], s_Gxx[0], y1, s_Yx[0]),
                                pos3(a11, ax[0], ax[cx], s_Gxx[0], s_Gxx[cx], s_Gxx[cx * MFX_XS + cx], y1, s_Yx[0], s_Yx[cx])));
    return r;
  };
  // ---- phase 1: column statistics + inner products with the extra columns (sequential over rows)
  double y_sq_seq = 0.0;
  for (int m = 0; m < M; ++m) y_sq_seq += s_y[m] * s_y[m];
  double umax = 0.0;
  for (int col = tid; col < 2 * NP; col += WG) {
    const int k = col >= NP, n = col - k * NP;
    double a2 = 0.0, ay = 0.0, ax[MFX_XS];
#pragma unroll
    for (int e = 0; e < MFX_XS; ++e) ax[e] = 0.0;
    if (n < N) {
      for (int m = 0; m < M; ++m) {
        const double d = elem(k, m, n);
        a2 += d * d;
        ay += s_y[m] * d;
#pragma unroll
        for (int e = 0; e < MFX_XS; ++e)
          if (e < NX) ax[e] += d * xx[(size_t)m * NX + e];
      }
    }
    (k ? s_A22 : s_A11)[n] = a2;
    (k ? s_Y2 : s_Y1)[n] = ay;
    double* axp = wsA + ((size_t)k * NP + n) * MFX_XS;
#pragma unroll
    for (int e = 0; e < MFX_XS; ++e) axp[e] = ax[e];
    // (scored from the slab, not from ax[]: a dynamically indexed register array would live in scratch)
    if (n < N)
      for (int t = 0; t < ntup; ++t) umax = fmax(umax, atom_best(a2, ay, axp, t));
  }
  for (int t = 0; t < ntup; ++t) umax = fmax(umax, s_Qx[t]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) umax = fmax(umax, __shfl_xor(umax, o));
  if (lane == 0) s_red[wave] = umax;
  __syncthreads();
  const double y_sq = (Kp == 4) ? s_red[16] : y_sq_seq;
  // running best score (same value in every thread): starts from the supports with fewer than two fascicle atoms
  double gmax_run = s_red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) gmax_run = fmax(gmax_run, s_red[w]);
  __syncthreads();   // s_red is reused by the rounds

  auto gen_chunk = [&](int ch, int buf) {
    const int c = tid & 15, m0 = tid >> 4;  // WG/16 row groups
    const int n = ch * 16 + c;
    double* dst = sB + (size_t)buf * (MP * 16) + c;
    for (int m = m0; m < MP; m += WG / 16) dst[m * 16] = elem(1, m, n);
    if (tid < 16 * MFX_XS) {  // stage A2x of the chunk's 16 atoms
      const int cc = tid / MFX_XS, e = tid - cc * MFX_XS;
      s_a2x[(buf * 16 + cc) * MFX_XS + e] = wsA[((size_t)NP + ch * 16 + cc) * MFX_XS + e];
    }
  };

  const int nrounds = (ntiles + NW - 1) / NW;
  const double eps_abs = eps_abs_of * y_sq;

  for (int round = 0; round < nrounds; ++round) {
    const int rt = round * NW + wave;
    const bool rt_valid = rt < ntiles;
    double afr[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) afr[kk] = rt_valid ? elem(0, 4 * kk + lg, rt * 16 + lc) : 0.0;
    // stage this wave's A1x rows
    if (rt_valid) {
      for (int q = lane; q < 16 * MFX_XS; q += 64) s_a1x[wave * 16 * MFX_XS + q] = wsA[((size_t)rt * 16) * MFX_XS + q];
    }
    double bs[4], bs2[4];   // best and runner-up score of the (lane,row) slot
    int bj[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { bs[r] = 0.0; bs2[r] = 0.0; bj[r] = -1; }

    gen_chunk(0, 0);
    __syncthreads();
    for (int ch = 0; ch < ntiles; ++ch) {
      const int buf = (NBUF == 2) ? (ch & 1) : 0;
      if constexpr (NBUF == 2) {
        if (ch + 1 < ntiles) gen_chunk(ch + 1, buf ^ 1);
      } else if (ch > 0) {
        gen_chunk(ch, 0);
        __syncthreads();
      }
      if (rt_valid) {
        const double* bp = sB + (size_t)buf * (MP * 16) + lg * 16 + lc;
        d4x acc = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[kk], bp[kk * 64], acc, 0, 0, 0);
        const int j = ch * 16 + lc;
        if (j < N) {
          // Ranking by feasible supports: the NNLS optimum of a tuple is the best score among the supports whose
          // unconstrained solution is non-negative.  Only the supports with BOTH fascicle atoms are ranked here (the
          // others come from the per-atom scores, see the top of the file); per pair the {1,2}(+fixed) block is
          // eliminated once (LDL^T), per extra column only the last row is added.
          const double a22 = s_A22[j], y2 = s_Y2[j];
          const double* a2x = s_a2x + (buf * 16 + lc) * MFX_XS;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int il = lg + 4 * r;
            const int i = rt * 16 + il;
            if (i < N) {
              const double a11 = s_A11[i], y1 = s_Y1[i], a12 = acc[r];
              const double* a1x = s_a1x + (wave * 16 + il) * MFX_XS;
              // LDL^T of the {1,2} block
              const double ip1 = mfx_rcp(a11);
              const double l21 = a12 * ip1;
              const double p2 = fma(-l21, a12, a22);
              const double u2 = fma(-l21, y1, y2);
              const bool ok2 = p2 > 1e-8 * a22;
              const double ip2 = mfx_rcp(ok2 ? p2 : 1.0);
              const double v1 = y1 * ip1, v2 = u2 * ip2;
              const double S2u = fma(u2, v2, y1 * v1);
              const double w1p = fma(-l21, v2, v1);
              double s = (ok2 && w1p >= 0.0 && v2 >= 0.0) ? S2u : 0.0;   // {1,2}
              // fixed (CSF) column: extend the elimination by one row
              double l31 = 0.0, b23 = 0.0, l32 = 0.0, ip3 = 0.0, u3 = 0.0, v3 = 0.0, S3u = 0.0;
              bool ok3 = false;
              if (HASF) {
                const double a1f = a1x[0], a2f = a2x[0], aff = s_Gxx[0], yf = s_Yx[0];
                l31 = a1f * ip1;
                b23 = fma(-l21, a1f, a2f);
                l32 = b23 * ip2;
                const double p3 = fma(-l32, b23, fma(-l31, a1f, aff));
                u3 = fma(-l32, u2, fma(-l31, y1, yf));
                ok3 = ok2 && (p3 > 1e-8 * aff);
                ip3 = mfx_rcp(ok3 ? p3 : 1.0);
                v3 = u3 * ip3;
                S3u = fma(u3, v3, S2u);
                const double w2f = fma(-l32, v3, v2);
                const double w1f = fma(-l31, v3, fma(-l21, w2f, v1));
                s = (ok3 && w1f >= 0.0 && w2f >= 0.0 && v3 >= 0.0) ? fmax(s, S3u) : s;  // {1,2,f}
              }
              for (int t = 0; t < ntup; ++t) {
                const int cx = x0 + t;
                const double a1e = a1x[cx], a2e = a2x[cx], aee = s_Gxx[cx * MFX_XS + cx], ye = s_Yx[cx];
                // support {1,2,x}: last row of the LDL^T on top of the {1,2} block
                const double m1 = a1e * ip1;
                const double t2 = fma(-m1, a12, a2e);
                const double m2 = t2 * ip2;
                {
                  const double p3x = fma(-m2, t2, fma(-m1, a1e, aee));
                  const double u3x = fma(-m2, u2, fma(-m1, y1, ye));
                  const bool okx = ok2 && (p3x > 1e-8 * aee);
                  const double w3 = u3x * mfx_rcp(okx ? p3x : 1.0);
                  const double w2 = fma(-m2, w3, v2);
                  const double w1 = fma(-m1, w3, fma(-l21, w2, v1));
                  s = (okx && w1 >= 0.0 && w2 >= 0.0 && w3 >= 0.0) ? fmax(s, fma(u3x, w3, S2u)) : s;
                }
                if (HASF) {  // support {1,2,f,x}: last row on top of the {1,2,f} block
                  const double afe = s_Gxx[cx];
                  const double t3 = fma(-m2, b23, fma(-m1, a1x[0], afe));
                  const double m3 = t3 * ip3;
                  const double p4 = fma(-m3, t3, fma(-m2, t2, fma(-m1, a1e, aee)));
                  const double u4 = fma(-m3, u3, fma(-m2, u2, fma(-m1, y1, ye)));
                  const bool ok4 = ok3 && (p4 > 1e-8 * aee);
                  const double w4 = u4 * mfx_rcp(ok4 ? p4 : 1.0);
                  const double w3 = fma(-m3, w4, v3);
                  const double w2 = fma(-m2, w4, fma(-l32, w3, v2));
                  const double w1 = fma(-m1, w4, fma(-l31, w3, fma(-l21, w2, v1)));
                  s = (ok4 && w1 >= 0.0 && w2 >= 0.0 && w3 >= 0.0 && w4 >= 0.0) ? fmax(s, fma(u4, w4, S3u)) : s;
                }
              }
              // slot update: columns come in increasing j, strict '>' keeps the first of equal scores
              const bool better = s > bs[r];
              bs2[r] = better ? bs[r] : fmax(bs2[r], s);
              bj[r] = better ? j : bj[r];
              bs[r] = better ? s : bs[r];
            }
          }
        }
      }
      __syncthreads();
    }
    double lmax = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) lmax = fmax(lmax, bs[r]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = fmax(lmax, __shfl_xor(lmax, o));
    if (lane == 0) s_red[wave] = lmax;
    __syncthreads();
    double rmax = s_red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) rmax = fmax(rmax, s_red[w]);
    gmax_run = fmax(gmax_run, rmax);
    const double thr = gmax_run - eps_abs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (bj[r] >= 0 && bs[r] >= thr) {
        const int slot = atomicAdd(&s_cnt[0], 1);
        if (slot < MFX_XMAXC) {
          s_cand[slot].score = bs[r];
          s_cand[slot].i = rt * 16 + lg + 4 * r;
          s_cand[slot].j = bj[r];
          s_cand[slot].e = 0;
        }
        if (bs2[r] >= thr) {   // the slot's runner-up could be the reference's pick too: its whole slot row goes exact
          const int f = atomicAdd(&s_cnt[1], 1);
          if (f < MFX_XFAM) { s_fam[f].type = 4; s_fam[f].a = rt * 16 + lg + 4 * r; s_fam[f].t = lc; }
        }
      }
    }
    __syncthreads();
  }

  // ---- one-atom and no-atom supports within rounding distance of the optimum -> family items
  const double thr_final = gmax_run - eps_abs;
  for (int col = tid; col < 2 * NP; col += WG) {
    const int k = col >= NP, n = col - k * NP;
    if (n < N) {
      const double a2 = (k ? s_A22 : s_A11)[n], ay = (k ? s_Y2 : s_Y1)[n];
      const double* axp = wsA + ((size_t)k * NP + n) * MFX_XS;
      for (int t = 0; t < ntup; ++t)
        if (atom_best(a2, ay, axp, t) >= thr_final) {
          const int f = atomicAdd(&s_cnt[1], 1);
          if (f < MFX_XFAM) { s_fam[f].type = 1 + k; s_fam[f].a = n; s_fam[f].t = t; }
        }
    }
  }
  if (tid < ntup && s_Qx[tid] >= thr_final) {
    const int f = atomicAdd(&s_cnt[1], 1);
    if (f < MFX_XFAM) { s_fam[f].type = 3; s_fam[f].a = 0; s_fam[f].t = tid; }
  }
  __syncthreads();

  // ---- exact re-evaluation (reference arithmetic and summation order) of everything short-listed
  const int nappend = s_cnt[0], nfam_app = s_cnt[1];
  const int ncand = nappend > MFX_XMAXC ? MFX_XMAXC : nappend;
  // (a voxel in which no support scores above zero keeps the reference's initial state: nothing to evaluate)
  const bool nothing = !(gmax_run > 0.0);
  if (tid == 0 && a.ovf_count) {   // diagnostics: short-listed pairs and family items of the launch
    atomicAdd(a.ovf_count + 2, ncand);
    atomicAdd(a.ovf_count + 3, nfam_app);
  }
  const bool exhaustive = !nothing && (a.maxc == 0 || nappend > a.maxc || nfam_app > MFX_XFAM);   // workgroup-uniform
  __syncthreads();
  // scratch inside the (now idle) B buffers
  static_assert(NBUF * MP * 16 >= 8 * NW + 16 + MP, "B buffers too small for the exact-stage scratch");
  double* s_rres = (double*)sB;                  // [NW]
  long* s_rkey = (long*)(s_rres + NW);           // [NW]
  double* s_rw = (double*)(s_rkey + NW);         // [NW][4]
  double* s_win = s_rw + 4 * NW;                 // res, w0..w3, key
  double* s_yrec = s_win + 8;                    // [MP]
  double res = INFINITY, w[4] = {0.0, 0.0, 0.0, 0.0};
  long key = -1;
  // one tuple (i, j, t) exactly: _3 = Cramer test + explicit residual (mf_utils.py:554-593), _4up = active-set optimum
  // from a sequentially summed Gram + explicit residual (mf_utils.py:640-649); keeps the lexicographic (res, scan key) minimum
  auto consider = [&](int i, int j, int t) {
    const int c3 = (Kp == 3) ? t : 0, c4 = 1 + t;
    double a11 = 0.0, a22 = 0.0, a12 = 0.0, y1 = 0.0, y2 = 0.0, a13 = 0.0, a23 = 0.0, a14 = 0.0, a24 = 0.0;
    for (int m = 0; m < M; ++m) {
      const double d1 = elem(0, m, i), d2 = elem(1, m, j), ym = s_y[m];
      const double x3 = xx[(size_t)m * NX + c3];
      a11 += d1 * d1; a22 += d2 * d2; a12 += d1 * d2; y1 += ym * d1; y2 += ym * d2;
      a13 += d1 * x3; a23 += d2 * x3;
      if (Kp == 4) { const double x4 = xx[(size_t)m * NX + c4]; a14 += d1 * x4; a24 += d2 * x4; }
    }
    double r, u[4] = {0.0, 0.0, 0.0, 0.0};
    long k;
    if (Kp == 3) {
      auto explicit_res = [&](const double* ww) {
        double rr = 0.0;
        for (int m = 0; m < M; ++m) {
          const double tt = (ww[0] * elem(0, m, i) + ww[1] * elem(1, m, j) + ww[2] * xx[(size_t)m * NX + c3] - s_y[m]);
          rr += tt * tt;
        }
        return rr;
      };
      nnls3_cramer(y_sq, a11, a12, a13, a22, a23, s_Gxx[c3 * MFX_XS + c3], y1, y2, s_Yx[c3], explicit_res, u, r);
      k = ((long)t * N + i) * N + j;  // scan order of _3: i3 -> i1 -> i2
    } else {
      const double g[10] = {a11, a12, a13, a14, a22, a23, a24, s_Gxx[0], s_Gxx[c4], s_Gxx[c4 * MFX_XS + c4]};
      const double yy[4] = {y1, y2, s_Yx[0], s_Yx[c4]};
      nnls_gram_subsets(4, g, yy, u);
      double rr = 0.0;
      for (int m = 0; m < M; ++m) {
        const double tt = (u[0] * elem(0, m, i) + u[1] * elem(1, m, j) + u[2] * xx[(size_t)m * NX] + u[3] * xx[(size_t)m * NX + c4] - s_y[m]);
        rr += tt * tt;
      }
      r = rr;
      k = ((long)i * N + j) * E + t;  // itertools.product order, last index fastest
    }
    if (r < res || (r == res && k < key)) { res = r; key = k; w[0] = u[0]; w[1] = u[1]; w[2] = u[2]; w[3] = u[3]; }
  };
  if (exhaustive) {
    // The short list overflowed: nothing above can be trusted.  Last resort, exact by construction: every tuple
    // through the reference arithmetic (tens of milliseconds for this voxel).
    if (tid == 0 && a.ovf_count) atomicAdd(a.ovf_count, 1);
    const long nall = (long)N * N * ntup;
    for (long q = tid; q < nall; q += WG) {
      const int t = (int)(q % ntup);
      const long pr = q / ntup;
      consider((int)(pr / N), (int)(pr % N), t);
    }
  } else if (!nothing) {
    // short-listed pairs, every extra tuple (they tie when the extra column is inactive)
    for (int q = tid; q < ncand * ntup; q += WG) {
      const int c = q / ntup, t = q - c * ntup;
      if (s_cand[c].score >= thr_final) consider(s_cand[c].i, s_cand[c].j, t);
    }
    for (int f = 0; f < nfam_app; ++f) {   // workgroup-uniform loop
      const int type = s_fam[f].type, fa = s_fam[f].a, ft = s_fam[f].t;
      if (type == 1) { for (int n = tid; n < N; n += WG) consider(fa, n, ft); }
      else if (type == 2) { for (int n = tid; n < N; n += WG) consider(n, fa, ft); }
      else if (type == 3) { if (tid == 0) consider(0, 0, ft); }
      else {   // 4: row fa, columns ft, ft + 16, ...: the (lane,row) slot of the scan, all extra tuples
        const int ncol = (N - ft + 15) / 16;
        for (int q = tid; q < ncol * ntup; q += WG) consider(fa, ft + 16 * (q / ntup), q % ntup);
      }
    }
  }
  // lexicographic (res, key) minimum over the workgroup, folded into the reference's initial state (w = 0, indices 0,
  // min_obj = y_sq; strict '<')
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double r2 = __shfl_xor(res, o);
    const long k2 = __shfl_xor(key, o);
    double u[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) u[q] = __shfl_xor(w[q], o);
    const bool take = (k2 >= 0) && (key < 0 || r2 < res || (r2 == res && k2 < key));
    if (take) { res = r2; key = k2; w[0] = u[0]; w[1] = u[1]; w[2] = u[2]; w[3] = u[3]; }
  }
  if (lane == 0) { s_rres[wave] = res; s_rkey[wave] = key; for (int q = 0; q < 4; ++q) s_rw[4 * wave + q] = w[q]; }
  __syncthreads();
  if (tid == 0) {
    double br = y_sq, bw[4] = {0.0, 0.0, 0.0, 0.0};  // initial state of the reference: w = 0, indices 0
    long bk = -1;
    for (int c = 0; c < NW; ++c) {
      const double r = s_rres[c];
      const long k = s_rkey[c];
      if (k < 0) continue;
      if (r < br || (r == br && bk >= 0 && k < bk)) { br = r; bk = k; bw[0] = s_rw[4 * c]; bw[1] = s_rw[4 * c + 1]; bw[2] = s_rw[4 * c + 2]; bw[3] = s_rw[4 * c + 3]; }
    }
    s_win[0] = br; s_win[1] = bw[0]; s_win[2] = bw[1]; s_win[3] = bw[2]; s_win[4] = bw[3];
    ((long*)s_win)[5] = bk;
  }
  __syncthreads();
  if (wave == 0) {
    const double best = s_win[0];
    const double w0 = s_win[1], w1 = s_win[2], w2 = s_win[3], w3 = s_win[4];
    const long bk = ((long*)s_win)[5];
    int bi = 0, bjx = 0, bt = 0;
    if (bk >= 0) {
      if (Kp == 3) { bjx = (int)(bk % N); bi = (int)((bk / N) % N); bt = (int)(bk / ((long)N * N)); }
      else { bt = (int)(bk % E); bjx = (int)((bk / E) % N); bi = (int)(bk / ((long)E * N)); }
    }
    const int c3 = (Kp == 3) ? bt : 0, c4 = 1 + bt;
    double sy = 0.0, sr = 0.0;
    for (int m = lane; m < M; m += 64) {
      double yr = elem(0, m, bi) * w0 + elem(1, m, bjx) * w1 + xx[(size_t)m * NX + c3] * w2;
      if (Kp == 4) yr += xx[(size_t)m * NX + c4] * w3;
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
    if (lane == 0) {
      double* out = a.params + (size_t)vox * a.num_params;
      const double wv[4] = {w0, w1, w2, w3};
      double M0 = 0.0;
      for (int k = 0; k < Kp; ++k) M0 += wv[k];
      double nu[4];
      for (int k = 0; k < 4; ++k) nu[k] = (fabs(M0) > 0) ? wv[k] / M0 : wv[k];
      const int i_csf = 2 * a.maxfasc + 1, i_ear = 2 * a.maxfasc + a.csf_on + 1;
      out[0] = M0;
      out[1] = nu[0];
      out[2] = nu[1];
      out[1 + a.maxfasc] = (double)bi;
      out[2 + a.maxfasc] = (double)bjx;
      if (has_csf) out[i_csf] = nu[2];
      if (E > 0) { out[i_ear] = nu[2 + has_csf]; out[i_ear + 1] = (double)bt; }
      out[a.num_params - 2] = best / M;
      out[a.num_params - 1] = r2;
    }
  }
}
