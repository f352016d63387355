// fit_k2w.hip -- two-fascicle voxels, "wide" variant of the split-FP16 screening kernel (fit_k2s.hip): ONE wave per
// SIMD (256-thread workgroups, 512 registers per lane), TL row tiles of 32 atoms of D1 per wave.
//
// Same mathematics, margins, ring, exact stage and results as fit_k2s.hip (whose helpers and constants it includes);
// what differs is the sweep:
//   * a wave keeps TL row tiles (hi and lo halves, K = 16 KS) in registers - TL = 2 for protocols of up to 208
//     measurements, TL = 1 for long protocols (KS up to 35: 560 measurements, which do not fit the 256 registers of
//     the two-waves-per-SIMD kernel at all and used to fall back to the FP64 kernel) - and multiplies every B
//     fragment it reads from LDS with all of them (TL x fewer LDS fragment reads per MFMA);
//   * there are no wave groups: the four waves run the same program; the VALU work of a chunk (pair screen of the
//     PREVIOUS chunk's accumulators - the accumulators are double-buffered - and the FP16 conversion of the NEXT
//     chunk's image) is cut into slices that sit between the MFMAs of the SAME wave, in the issue slots the matrix
//     pipe leaves free (measured: ~2 cycles per VALU instruction sliced into an MFMA chain, tools/micro/
//     pingpong_overlap.hip), one workgroup barrier per chunk;
//   * NB = 2 chunk images: image (c+1) & 1 is written while image c & 1 is multiplied; NB = 1 (KS > 20: one image is
//     72 KB at KS = 35) converts and multiplies in turn, two barriers per chunk.
#pragma once
#include "fit_k2s.hip"

// BR: the protocol has G-bracketed rows (screening through the plan's virtual shells, exact stage as mfx_eval_br)
// TL: row tiles per wave; NB: LDS images of D2 chunks (2: one workgroup barrier per chunk; 1: two)
// XC: the [N, N, 1] form (one fixed extra column x projected out through the last padded measurement row; short lists for
// fit_k2x.hip's exact stage instead of an exact stage here) - the same changes as in fit_k2s.hip, see there and DESIGN.md 4.3b.
template <int KS, int TL, bool BR, int NB, bool XC = false>
__global__ __launch_bounds__(256, 1) void mfx_fit_k2w_kernel(FitK2Args a) {
  constexpr int WG = 256, NW = 4;
  constexpr int MP = KS * 16;  // padded measurement count
  extern __shared__ double smem[];
  int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int M = a.P.M, N = a.T.N, ldn = a.T.ldn;
  const int NP = (N + 31) & ~31;  // atoms padded to a multiple of 32
  const int ntiles = NP >> 5;
  const double2* __restrict__ tab = a.T.tab;
  const int vox = a.vox_list ? a.vox_list[a.vox_base + blockIdx.x] : a.vox_base + (int)blockIdx.x;

  // ---- LDS carve-up
  _Float16* sBh = (_Float16*)smem;                 // [NB][KS][64][8]  hi halves, fragment order
  _Float16* sBl = sBh + NB * KS * 512;             // [NB][KS][64][8]  lo halves
  double* s_y = (double*)(sBl + NB * KS * 512);    // [MP]
  double* s_t0 = s_y + MP;                         // [2][MP]
  double* s_red = s_t0 + 2 * MP;                   // [32] scratch
  Cand* s_cand = (Cand*)(s_red + 32);              // [MFX_S_CAP]
  unsigned long long* s_thr = (unsigned long long*)(s_cand + MFX_S_CAP);  // [0] threshold bits, [1] lost-entry max bits
  int* s_r0 = (int*)(s_thr + 2);                   // [2][MP] knot row * ldn (element offset of the row in the table)
  int* s_cnt = s_r0 + 2 * MP;                      // [4]
  float* s_t0f = (float*)(s_cnt + 4);              // [2][MP] FP32 copy of s_t0 for the screening passes
  float* s_Zf = s_t0f + 2 * MP;                    // [2][NP] Z1 | Z2 = d.y/|d| of the rotated atoms (-1e30 beyond N)
  float* s_cs = s_Zf + 2 * NP;                     // [2][NP] |d1| | |d2| (FP32, table units; 0: no such atom): accumulator = cosine |d1| |d2|
  float* s_yf = s_cs + 2 * NP;                     // [MP] FP32 copy of y (ranking statistics)
  float* s_pq = s_yf + MP;                         // [NW][TL][2][32] pair-screen constants of each wave's row tiles
  // bracketed protocols: exact-stage descriptors of the upper shell, and separate screening row offsets
  double* s_t1 = (double*)(s_pq + NW * TL * 64);   // [2][MP]
  double* s_tG = s_t1 + (BR ? 2 * MP : 0);         // [MP]
  double* s_dG = s_tG + (BR ? MP : 0);             // [MP]
  int* s_r1 = (int*)(s_dG + (BR ? MP : 0));        // [2][MP] upper-shell knot row * ldn, or -1
  int* s_rs = BR ? s_r1 + 2 * MP : s_r0;           // [2][MP] row offsets used by the screening passes
  int* s_evl4 = s_r1 + (BR ? 4 * MP : 0);          // [MFX_S_CAP] exact-stage compaction list, KS < 8 only (else inside the B image)
  float* s_xf = (float*)(s_evl4 + (KS < 8 ? MFX_S_CAP : 0));   // XC: [MP] the FP32 copy of x (0 beyond M)
  float* s_uf = s_xf + (XC ? MP : 0);              // XC: [2][NP] u = d.x^ of the rotated atoms

  MFX_STAMP(0);
  // ---- phase 0: y, knot-interval descriptors
  const double* __restrict__ yv = a.Y + (size_t)vox * M;
  const double* __restrict__ pk = a.peaks + (size_t)vox * a.peaks_ld;
  for (int m = tid; m < MP; m += WG) { const double v = (m < M) ? yv[m] : 0.0; s_y[m] = v; s_yf[m] = (float)v; }
  for (int idx = tid; idx < 2 * MP; idx += WG) {
    const int k = idx / MP, m = idx - k * MP;
    RowDesc rd;
    rd.r0 = a.T.P; rd.t0 = 0.0; rd.r1 = -1; rd.t1 = 0.0;  // padded rows -> the all-zero table row
    if (m < M) rd = mfx_row_desc(a.T, a.P, m, pk[3 * k], pk[3 * k + 1], pk[3 * k + 2]);
    s_r0[idx] = rd.r0 * ldn;
    s_t0[idx] = rd.t0;
    if constexpr (!BR) {
      s_t0f[idx] = (float)rd.t0;
    } else {
      s_r1[idx] = rd.r1 < 0 ? -1 : rd.r1 * ldn;
      s_t1[idx] = rd.t1;
      if (k == 0) { s_tG[m] = (m < M) ? a.P.tG[m] : 0.0; s_dG[m] = (m < M) ? a.P.dG[m] : 1.0; }
      // screening descriptor: the row's single (possibly virtual) shell of the plan's screening view
      int rs = a.T.P;
      double ts = 0.0;
      if (m < M) {
        const int sg = a.P.s_scr[m], st = a.P.offs[2 * sg], cn = a.P.offs[2 * sg + 1];
        const double u = mfx_absdot(a.P.g + 3 * m, pk[3 * k], pk[3 * k + 1], pk[3 * k + 2]);
        int j = mfx_searchsorted_left(a.P.xs + st, cn, u);
        j = j < 1 ? 1 : (j > cn - 1 ? cn - 1 : j);
        rs = st + j - 1;
        ts = u - a.P.xs[rs];
      }
      s_rs[idx] = rs * ldn;
      s_t0f[idx] = (float)ts;
    }
  }
  if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; s_thr[0] = 0ull; s_thr[1] = 0ull; }
  if constexpr (XC) {
    for (int m = tid; m < MP; m += WG) s_xf[m] = (m < M) ? (float)a.xc[m] : 0.0f;
    if (tid < 2) ((unsigned long long*)s_red)[24 + tid] = 0ull;   // max |d|^2/|d'|^2 of each dictionary (bits of a non-negative double)
  }
  if (tid < 2) mfx_check_dir(a.P, pk + 3 * tid, vox);
  __syncthreads();

  // table entry (row offset ro, atom n) through a 32-bit element offset: SGPR base + VGPR offset addressing
  auto tab_at = [&](int ro, int n) -> double2 { return *(const double2*)((const char*)tab + ((unsigned)(ro + n) << 4)); };
  // the FP32 copy of the table feeds everything that only RANKS pairs (half the L2 -> CU bytes)
  const float2* __restrict__ tab32 = a.P.tab32s;   // == a.T.tab32 unless the plan has virtual shells
  auto tab32_at = [&](int ro, int n) -> float2 { return *(const float2*)((const char*)tab32 + ((unsigned)(ro + n) << 3)); };
  // two adjacent atoms (n even) in one 16-byte load: {ylo_n, slope_n, ylo_n+1, slope_n+1}
  auto tab32x2_at = [&](int ro, int n) -> f32x4 { return *(const f32x4*)((const char*)tab32 + ((unsigned)(ro + n) << 3)); };
  // exact-arithmetic rotated dictionary entry: slope * t + y_lo, separate mul and add (mfx_eval)
  auto elem = [&](int k, int m, int n) -> double {
    const double2 e = tab_at(s_r0[k * MP + m], n);
    const double v0 = e.y * s_t0[k * MP + m] + e.x;
    if constexpr (BR) {   // linear interpolation in G between the two shell values, mf_utils.py:1950-1955 (mfx_eval_br)
      const int r1 = s_r1[k * MP + m];
      if (r1 < 0) return v0;
      const double2 f = tab_at(r1, n);
      const double v1 = f.y * s_t1[k * MP + m] + f.x;
      const double sl = (v1 - v0) / s_dG[m];
      return sl * s_tG[m] + v0;
    }
    return v0;
  };

  MFX_STAMP(1);
  // ---- phase 1: column statistics; y_sq sequential as mf_utils.py:307-325
  double y_sq_v = 0.0;
  for (int m = 0; m < M; ++m) y_sq_v += s_y[m] * s_y[m];
  // wave-uniform values that live through the whole kernel go to scalar registers (the vector file is full)
  const double y_sq = mfx_readlane_f64(y_sq_v, 0);
  double rsh_v = 1.0, yx_v = 0.0;   // XC: 1/|v| and y.v/|v| of v = the FP32 copy of x (see fit_k2s.hip)
  if constexpr (XC) {
    double h = 0.0, xy = 0.0;
    for (int m = 0; m < M; ++m) { const double xv = (double)s_xf[m]; h = fma(xv, xv, h); xy = fma(xv, (double)s_yf[m], xy); }
    rsh_v = h > 0.0 ? 1.0 / sqrt(h) : 0.0;
    yx_v = xy * rsh_v;
  }
  const double rsh = mfx_readlane_f64(rsh_v, 0), yx = mfx_readlane_f64(yx_v, 0);
  const double y_sq_p = XC ? fmax(y_sq - yx * yx, 0.0) : y_sq;   // |y'|^2
  double my_s[2] = {0.0, 0.0};
  int my_n[2] = {0, 0};
  {
    // The vector-memory pipe of a CU retires roughly one wave load per 20 cycles whatever its width (<= 16 B per
    // lane), and this kernel issues ~1e4 of them per voxel: table entries are therefore fetched two atoms at a
    // time (16 B: {ylo, slope} of atoms n, n+1).  A thread accumulates the column pairs v = tid + 512 p (atoms 2v,
    // 2v+1), all passes at once: independent loads in flight, and the per-row constants (knot row, offset, y)
    // come from LDS as one 16-byte broadcast read per four rows.
    // Ranking statistics only (FP32 table, fused ops): the exact stage re-sums in reference order.
    // Only D2 here: the statistics of D1 fall out of the A-operand generation of each round (the table is then
    // read once for both purposes: the L2 -> L1 fill rate, ~32 B/clk, is what bounds these passes).
    const int VH = ((N + 1) / 2 + 63) & ~63;
    const int npass = (VH + WG - 1) / WG;
#pragma unroll 1
    for (int kd = XC ? 0 : 1; kd < 2; ++kd) {   // (XC: both dictionaries, projected statistics)
    double ms_cur = 0.0, rm2_cur = 1.0;
    int mn_cur = 0;
    for (int p0 = 0; p0 < npass; p0 += 2) {
      int kq[2], nq[2];
      bool wact[2];
      double a2[2][2], ay[2][2], au[2][2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int v = tid + WG * (p0 + q);
        kq[q] = kd;
        nq[q] = 2 * v;
        wact[q] = __any((p0 + q < npass) && (nq[q] < N));
        a2[q][0] = a2[q][1] = ay[q][0] = ay[q][1] = 0.0;
        au[q][0] = au[q][1] = 0.0;
      }
      // software pipeline over groups of four rows: the 8 table loads of the next group are in flight while this
      // group is accumulated (the vector-memory pipe and the FP64 VALU work of this pass each take ~40 k cycles per
      // voxel: un-pipelined they simply add up)
      int ncl[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) ncl[q] = min(nq[q], ldn - 2);   // ldn is even: the pair stays inside the row
      f32x4 dbuf[2][2][4];
      auto issue = [&](int m4, auto stc) {
        constexpr int st = decltype(stc)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (wact[q]) {   // wave-uniform
            const i32x4 r = *(const i32x4*)(s_rs + kq[q] * MP + m4);
#pragma unroll
            for (int e = 0; e < 4; ++e) dbuf[st][q][e] = tab32x2_at(r[e], ncl[q]);
          }
        }
      };
      auto accumulate = [&](int m4, auto stc) {
        constexpr int st = decltype(stc)::value;
        const f32x4 yv = *(const f32x4*)(s_yf + m4);
        f32x4 xv = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (XC) xv = *(const f32x4*)(s_xf + m4);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (wact[q]) {
            const f32x4 t = *(const f32x4*)(s_t0f + kq[q] * MP + m4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x4 d = dbuf[st][q][e];
              const double d0 = (double)fmaf(d[1], t[e], d[0]), d1 = (double)fmaf(d[3], t[e], d[2]);
              const double ye = (double)yv[e];
              a2[q][0] = fma(d0, d0, a2[q][0]);
              ay[q][0] = fma(ye, d0, ay[q][0]);
              a2[q][1] = fma(d1, d1, a2[q][1]);
              ay[q][1] = fma(ye, d1, ay[q][1]);
              if constexpr (XC) {
                const double xe = (double)xv[e];
                au[q][0] = fma(xe, d0, au[q][0]);
                au[q][1] = fma(xe, d1, au[q][1]);
              }
            }
          }
        }
      };
      issue(0, std::integral_constant<int, 0>{});
      for (int m4 = 0; m4 < MP; m4 += 8) {   // MP is a multiple of 16
        issue(m4 + 4, std::integral_constant<int, 1>{});
        accumulate(m4, std::integral_constant<int, 0>{});
        if (m4 + 8 < MP) issue(m4 + 8, std::integral_constant<int, 0>{});
        accumulate(m4 + 4, std::integral_constant<int, 1>{});
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int k = kq[q], n = nq[q] + u;
          if (p0 + q < npass && n < NP) {
            const bool act = n < N;
            if constexpr (!XC) {
              const double inv = (act && a2[q][u] > 0.0) ? 1.0 / sqrt(a2[q][u]) : 0.0;
              const double z = ay[q][u] * inv;
              s_Zf[k * NP + n] = act ? (float)z : -1e30f;
              s_cs[NP + n] = (act && inv > 0.0) ? (float)sqrt(a2[q][u]) : 0.0f;
              const double s = z > 0.0 ? z * z : 0.0;
              if (act && s > ms_cur) { ms_cur = s; mn_cur = n; }   // increasing n per thread and dictionary
            } else {   // projected statistics (fit_k2s.hip)
              const double uu = au[q][u] * rsh;
              const double n2p = a2[q][u] - uu * uu;
              const bool ok = act && a2[q][u] > 0.0;
              if (ok && !(n2p > a2[q][u] * (1.0 / 16.0))) s_cnt[1] = 1;
              const bool okp = ok && n2p > 0.0;
              const double np = okp ? sqrt(n2p) : 0.0;
              const double inv = okp ? 1.0 / np : 0.0;
              const double z = (ay[q][u] - uu * yx) * inv;
              s_Zf[k * NP + n] = act ? (float)z : -1e30f;
              s_cs[k * NP + n] = okp ? (float)np : 0.0f;
              s_uf[k * NP + n] = okp ? (float)uu : 0.0f;
              if (okp) rm2_cur = fmax(rm2_cur, a2[q][u] * inv * inv);
              const bool feas = okp && z > 0.0 && (yx - z * inv * uu) >= 0.0;
              const double s = feas ? z * z : 0.0;
              if (s > ms_cur) { ms_cur = s; mn_cur = n; }
            }
          }
        }
      }
    }
    if (kd == 0) { my_s[0] = ms_cur; my_n[0] = mn_cur; } else { my_s[1] = ms_cur; my_n[1] = mn_cur; }
    if constexpr (XC) {
      if (rm2_cur > 1.0) atomicMax((unsigned long long*)s_red + 24 + kd, mfx_nonneg_bits(rm2_cur));
    }
    }
  }
  // best single atom of each dictionary (first index on ties): they stand for every pair whose optimum
  // has one active atom (mf_utils.py:357-379); the exact stage expands the winner's family.  D2's here, D1's
  // after the rounds (its statistics come with the A operands); the threshold starts from what is known.
  {
    double* s_bs = s_red;            // [2][8]
    int* s_bn = (int*)(s_red + 16);  // [2][8]
#pragma unroll
    for (int k = XC ? 0 : 1; k < 2; ++k) {
      double s = my_s[k];
      int n = my_n[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double s2 = __shfl_xor(s, o);
        const int n2 = __shfl_xor(n, o);
        const bool take = (s2 > s) || (s2 == s && n2 < n);
        s = take ? s2 : s;
        n = take ? n2 : n;
      }
      if (lane == 0) { s_bs[k * 8 + wave] = s; s_bn[k * 8 + wave] = n; }
    }
    __syncthreads();
  }
  // margins (XC: amplified, with the statistics' rounding term and the exact kernel's tie tolerance as floor: fit_k2s.hip)
  double ramp_v = 1.0;
  if constexpr (XC) {
    const unsigned long long* rw = (const unsigned long long*)s_red + 24;
    ramp_v = sqrt(fmax(1.0, __longlong_as_double((long long)rw[0]))) * sqrt(fmax(1.0, __longlong_as_double((long long)rw[1])));
  }
  const double ramp = mfx_readlane_f64(ramp_v, 0);
  const double dc_eff = XC ? MFX_S_DC * ramp : MFX_S_DC;
  const double mrg = mfx_readlane_f64(XC ? fmax(dc_eff * y_sq_p + 2e-6 * ramp * sqrt(y_sq * y_sq_p), 1e-9 * y_sq) : dc_eff * y_sq_p, 0);   // |S(c~) - S(c)| <= mrg
  const double etol = mfx_readlane_f64(XC ? dc_eff * sqrt(y_sq_p) + 2e-6 * ramp * sqrt(y_sq) : dc_eff * sqrt(y_sq_p), 0);   // |e(c~) - e(c)| <= etol
  {
    double* s_bs = s_red;            // [2][8]
    int* s_bn = (int*)(s_red + 16);  // [2][8]
    if (tid == 0) {
      double best1 = 0.0;
      for (int k = XC ? 0 : 1; k < 2; ++k) {   // (!XC) D1's best single atom is known after the last round (see there)
        double s = s_bs[k * 8];
        int n = s_bn[k * 8];
        for (int w = 1; w < NW; ++w) {
          const double s2 = s_bs[k * 8 + w];
          const int n2 = s_bn[k * 8 + w];
          if (s2 > s || (s2 == s && n2 < n)) { s = s2; n = n2; }
        }
        best1 = fmax(best1, s);
        if (!XC && s > 0.0) {   // (XC: supports with fewer than two fascicle atoms belong to the exact kernel's families)
          const int slot = s_cnt[0]++;
          s_cand[slot].score = s + mrg;   // exact single-atom score up to the statistics' rounding: evaluated only if it can win
          s_cand[slot].i = k ? 0 : n;
          s_cand[slot].j = (k ? n : 0) | MFX_S_BOUND;
        }
      }
      // single-atom scores are exact: a pair matters only if S(c) >= best1, i.e. S(c~) >= best1 - mrg
      s_thr[0] = mfx_nonneg_bits(best1 - mrg);
    }
  }
  __syncthreads();

  // ring append (rare path)
  auto push = [&](double S, int i, int j) {
    const int slot = atomicAdd(&s_cnt[0], 1);
    const int idx = slot & (a.scap - 1);
    if (slot >= a.scap) {   // overwriting: remember the best score that got lost
      const double old = s_cand[idx].score;
      atomicMax(&s_thr[1], mfx_nonneg_bits(fmin(old, 1e300)));
    }
    s_cand[idx].score = S;
    s_cand[idx].i = i;
    s_cand[idx].j = j;
  };

  MFX_STAMP(2);
  double bs1 = 0.0;   // best single atom of D1 among the row tiles this wave has generated
  int bn1 = 0;
  constexpr int TPR = NW * TL;   // row tiles per round
  const int nrounds = (ntiles + TPR - 1) / TPR;
  // generation items of a chunk image: (pair of adjacent atoms) x (the 8 rows of one MFMA fragment): 16 x 2 KS of them,
  // 8 KS per wave, a lane takes items l, l + 64, ... of its wave's share
  constexpr int IPW = 8 * KS;              // items per wave
  constexpr int IT = (IPW + 63) / 64;      // items per lane (the last one only for the first IPW - 64 (IT-1) lanes)
  for (int round = 0; round < nrounds; ++round) {
    // A last round with ONE row tile left (N = 782: 25 = 3*8 + 1) is shared by all waves: each keeps the same A tile
    // and takes every 4th column tile, generating its B operand straight into registers (no LDS image, no barrier).
    const bool tail = (ntiles - round * TPR == 1) && (ntiles > 1);
    int rts[TL];
    bool rtv[TL];
#pragma unroll
    for (int t = 0; t < TL; ++t) {
      const int rt = tail ? round * TPR : round * TPR + wave * TL + t;
      rtv[t] = (rt < ntiles) && !(tail && t > 0);   // wave-uniform
      rts[t] = rtv[t] ? rt : 0;
    }
    // A operands: TL x 32 atoms of D1, all KS k-steps, split in registers, UN-normalised like D2; the column statistics
    // I1 = 1/|d1|, Z1 = d1.y/|d1| fall out of the same read of the table (see fit_k2s.hip)
    h8 afh[TL][KS], afl[TL][KS];
    mfx_static_for<0, TL>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      const bool rt_valid = rtv[t];
      const int n = rts[t] * 32 + lr;
      const int nn = min(n, ldn - 1);
      double a2 = 0.0, ay = 0.0;
      float um1 = 0.0f;   // XC: the last padded row of the A operand carries -u1
      if constexpr (XC) um1 = (rt_valid && n < N) ? -s_uf[n] : 0.0f;
      mfx_static_for<0, KS>([&](auto kc) {
        constexpr int ks = decltype(kc)::value;
        float2 d[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = tab32_at(s_rs[16 * ks + 8 * lh + j], nn);
        h8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float fv = fmaf(d[j].y, s_t0f[16 * ks + 8 * lh + j], d[j].x);
          fv = rt_valid ? fv : 0.0f;
          if constexpr (XC) {
            if constexpr (ks == KS - 1) { if (j == 7) fv = lh ? um1 : fv; }
          } else {
            const double fd = (double)fv;
            a2 = fma(fd, fd, a2);
            ay = fma((double)s_yf[16 * ks + 8 * lh + j], fd, ay);
          }
          _Float16 x, y;
          mfx_split16(fv, x, y);
          vh[j] = x; vl[j] = y;
        }
        asm volatile("" : "+v"(vh), "+v"(vl));
        afh[t][ks] = vh; afl[t][ks] = vl;
      });
      if constexpr (!XC) {
      a2 += __shfl_xor(a2, 32);
      ay += __shfl_xor(ay, 32);
      const bool act = rt_valid && n < N;
      const double nrm = sqrt(a2);
      const double inv = (act && a2 > 0.0) ? 1.0 / nrm : 0.0;
      const double z = ay * inv;
      if (rt_valid && lh == 0) {
        s_Zf[n] = act ? (float)z : -1e30f;
        s_cs[n] = (act && a2 > 0.0) ? (float)nrm : 0.0f;
      }
      double sb = (act && z > 0.0) ? z * z : 0.0;
      int nb = n;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        const double s2 = __shfl_xor(sb, o);
        const int n2 = __shfl_xor(nb, o);
        const bool take = (s2 > sb) || (s2 == sb && n2 < nb);
        sb = take ? s2 : sb;
        nb = take ? n2 : nb;
      }
      sb = mfx_readlane_f64(sb, 0);
      nb = __builtin_amdgcn_readfirstlane(nb);
      if (sb > bs1) { bs1 = sb; bn1 = nb; }
      if (lane == 0 && sb - mrg > 0.0) atomicMax(&s_thr[0], mfx_nonneg_bits(sb - mrg));
      }
    });

    // ---- pair screen of one 32x32 accumulator tile (row tile t of this wave) against column tile ct: the fast FP32
    // pass (3 VALU per pair) and, for the flagged register groups, the FP64 criteria - see fit_k2s.hip for the maths.
    double thr = 0.0;
    double thr_rows[TL];
#pragma unroll
    for (int t = 0; t < TL; ++t) thr_rows[t] = -1.0;
    const float DCF = XC ? (float)(((double)MFX_S_DC + 2e-6) * ramp) * (1.0f + 2e-7f) : (float)MFX_S_DC + 2e-6f;
    auto pq_of = [&](float z, float rth, float& P, float& Q) {
      P = fminf(1.0f, fmaxf(z, 0.0f) * rth);
      Q = __builtin_amdgcn_sqrtf(fmaxf(0.0f, fmaf(-P, P, 1.0f) - 1.2e-7f)) * (1.0f - 3e-7f);
    };
    // column constants of the tile being screened (shared by the TL row tiles of the wave)
    float sc_p2 = 0.0f, sc_q2 = 0.0f, sc_rth = 1.0f;
    auto scan_cols = [&](int ct) {
      const int j = ct * 32 + lr;
      thr = fmax(thr, __longlong_as_double((long long)s_thr[0]));
      sc_rth = __builtin_amdgcn_rsqf(fmaxf((float)thr * (1.0f - 2e-7f), 1e-30f)) * (1.0f + 4e-7f);
      const float z2f = s_Zf[NP + j], n2 = s_cs[NP + j];
      float P2, Q2;
      pq_of(z2f, sc_rth, P2, Q2);
      const bool colok = n2 > 0.0f;
      sc_p2 = colok ? (P2 + DCF) * n2 : -1e18f;
      sc_q2 = colok ? Q2 * ((1.0f - DCF) * n2) : 1e18f;
    };
    // row constants of tile t, refreshed when the threshold has risen (wave-uniform branch)
    auto scan_rows = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      float* pqw = s_pq + (wave * TL + t) * 64;
      if (thr > thr_rows[t]) {
        thr_rows[t] = thr;
        if (lane < 32) {
          const float z1 = s_Zf[rts[t] * 32 + lane], n1 = s_cs[rts[t] * 32 + lane];
          float P, Q;
          pq_of(z1, sc_rth, P, Q);
          const bool ok = n1 > 0.0f;
          pqw[lane] = ok ? (P + DCF) * n1 : 0.0f;
          pqw[32 + lane] = ok ? Q * ((1.0f - DCF) * n1) : 1e18f;
        }
      }
    };
    // fast pass over register group q (accumulator entries 4q .. 4q+3): max of t over the four pairs (>= 0: some pair passes)
    auto scan_group = [&](const f32x16& acc, auto tc, auto qc) -> float {
      constexpr int t = decltype(tc)::value;
      constexpr int q = decltype(qc)::value;
      const float* pqw = s_pq + (wave * TL + t) * 64;
      const f32x4 p1q = *(const f32x4*)(pqw + 8 * q + 4 * lh);
      const f32x4 q1q = *(const f32x4*)(pqw + 32 + 8 * q + 4 * lh);
      float tt[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) tt[u] = fmaf(-q1q[u], sc_q2, fmaf(p1q[u], sc_p2, -acc[4 * q + u]));
      return fmaxf(fmaxf(tt[0], tt[1]), fmaxf(tt[2], tt[3]));
    };
    // FP64 criteria on the flagged groups of one tile (rare once thr is close to the optimum); mm[q] from scan_group
    auto scan_exact = [&](const f32x16& acc, int rt, int ct, const float* mm) {
      const int j = ct * 32 + lr;
      const double z2 = (double)s_Zf[NP + j], n2d = (double)s_cs[NP + j];
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {
        if (!__any(mm[q] >= 0.0f)) continue;
#pragma unroll 1
        for (int gg = 0; gg < 4; ++gg) {
          const int g = 4 * q + gg;
          const int i = rt * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
          const double n12 = (double)s_cs[i] * n2d;
          const double c = n12 > 0.0 ? (double)acc[g] / n12 : 0.0;
          const double z1 = (double)s_Zf[i];
          const double e1 = fma(-c, z2, z1);
          const double e2 = fma(-c, z1, z2);
          const double den = fma(-c, c, 1.0);
          const double num = fma(z2, e2, z1 * e1);
          const bool pos = (e1 > -etol) & (e2 > -etol);
          const bool wellc = (den >= MFX_S_DENMIN) & (c >= 0.0);
          const bool hit = pos & wellc & (fma(-thr, den, num) >= 0.0);
          const bool near = pos & !wellc;
          if (!__any(hit | near)) continue;
          double S = -1.0;
          if (hit) {
            S = num / den;
          } else if (near) {
            const double dlo = den - 2.0 * dc_eff - dc_eff * dc_eff;
            const double u1 = fabs(e1) + etol, u2 = fabs(e2) + etol;
            S = (dlo > 0.0 && c > -0.5) ? fmin(fma(z2, z2, u1 * u1 / dlo), fma(z1, z1, u2 * u2 / dlo)) + mrg : 1e300;
          }
          double sraise = hit ? S : 0.0;
          double slist = S;
          if constexpr (XC) {   // see fit_k2s.hip: only feasible scores raise the threshold
            const double q1 = (double)s_uf[i] * mfx_rcp_nr(fmax((double)s_cs[i], 1e-300));
            const double q2 = (double)s_uf[NP + j] * mfx_rcp_nr(fmax(n2d, 1e-300));
            const bool feas = hit && (fma(-e2, q2, fma(-e1, q1, yx * den)) >= 8.0 * etol);
            sraise = feas ? S : 0.0;
            if (__any(hit && !feas)) {
              const double n1p = (double)s_cs[i], u1 = (double)s_uf[i], u2 = (double)s_uf[NP + j];
              const double m1 = fma(u1, u1, n1p * n1p), m2 = fma(u2, u2, n2d * n2d);
              const double i1 = mfx_rcp_nr(fmax(sqrt(m1), 1e-300)), i2 = mfx_rcp_nr(fmax(sqrt(m2), 1e-300));
              const double w1 = fma(u1, yx, z1 * n1p) * i1, w2 = fma(u2, yx, z2 * n2d) * i2;
              const double c0 = fma(u1, u2, (double)acc[g]) * i1 * i2;
              const double f1 = fma(-c0, w2, w1), f2 = fma(-c0, w1, w2), den0 = fma(-c0, c0, 1.0);
              const bool ok0 = hit && !feas && (f1 > etol) && (f2 > etol) && (den0 >= MFX_S_DENMIN) && (c0 >= 0.0) && (m1 > 0.0) && (m2 > 0.0);
              if (ok0) sraise = fma(w2, f2, w1 * f1) * mfx_rcp_nr(den0) - yx * yx;
              // (fit_k2s.hip: x clearly inactive in the pair's optimum - listed with its plain score, or not at all)
              const bool xneg = hit && (e1 > etol) && (e2 > etol) && (fma(-e2, q2, fma(-e1, q1, yx * den)) <= -8.0 * etol) &&
                                (den0 >= MFX_S_DENMIN) && (c0 >= 0.0) && (m1 > 0.0) && (m2 > 0.0);
              if (xneg) {
                if (ok0) slist = sraise + mrg;
                else if (f1 < -etol || f2 < -etol) slist = -1.0;
              }
            }
          }
          const double smax = wave_max(fmax(sraise, 0.0));
          if (smax - 2.0 * mrg > thr) {
            thr = smax - 2.0 * mrg;
            if (lane == 0) atomicMax(&s_thr[0], mfx_nonneg_bits(thr));
          }
          if ((hit | near) && slist >= thr) push(slist, i, hit ? j : (j | MFX_S_BOUND));
        }
      }
    };
    // whole screen of one tile in one go (tail round, and the last chunk of a sweep)
    auto scan_tile = [&](const f32x16& acc, auto tc, int ct) {
      constexpr int t = decltype(tc)::value;
      scan_cols(ct);
      scan_rows(tc);
      float mm[4];
      mm[0] = scan_group(acc, tc, std::integral_constant<int, 0>{});
      mm[1] = scan_group(acc, tc, std::integral_constant<int, 1>{});
      mm[2] = scan_group(acc, tc, std::integral_constant<int, 2>{});
      mm[3] = scan_group(acc, tc, std::integral_constant<int, 3>{});
      if (__any(fmaxf(fmaxf(mm[0], mm[1]), fmaxf(mm[2], mm[3])) >= 0.0f)) scan_exact(acc, rts[t], ct, mm);
    };

    if (tail) {
      thr = __longlong_as_double((long long)s_thr[0]);
      for (int ct = wave; ct < ntiles; ct += NW) {
        const int n = ct * 32 + lr;
        const int nn = min(n, ldn - 1);
        float tail_u2 = 0.0f;
        if constexpr (XC) tail_u2 = s_uf[NP + n];
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
        float2 d[2][8];   // table entries of k-step ks (in use) and ks+1 (in flight)
#pragma unroll
        for (int j = 0; j < 8; ++j) d[0][j] = tab32_at(s_rs[MP + 8 * lh + j], nn);
        mfx_static_for<0, KS>([&](auto kc) {
          constexpr int ks = decltype(kc)::value;
          if constexpr (ks + 1 < KS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) d[(ks + 1) & 1][j] = tab32_at(s_rs[MP + 16 * (ks + 1) + 8 * lh + j], nn);
          }
          h8 bh, bl;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            _Float16 x, y;
            float fv = fmaf(d[ks & 1][j].y, s_t0f[MP + 16 * ks + 8 * lh + j], d[ks & 1][j].x);
            if constexpr (XC && ks == KS - 1) { if (j == 7) fv = lh ? tail_u2 : fv; }   // the spare row carries u2
            mfx_split16(fv, x, y);
            bh[j] = x; bl[j] = y;
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[0][ks], bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[0][ks], bl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afl[0][ks], bh, acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
        scan_tile(acc, std::integral_constant<int, 0>{}, ct);
      }
      __syncthreads();   // all appends of the round are in the ring
      continue;
    }

    // ---- LDS sweep.  Chunk c = column tile c of D2 (32 atoms), image in buffer c % NB, fragment order
    // [k-step][lane][8 halves] (ds_read_b128, conflict-free; 32-byte stores).
    // table entries of this lane's generation items: {ylo, slope} of two adjacent atoms, 8 rows (NB = 2 only: with one
    // image the items go through a two-stage pipeline of their own, gen_direct)
    f32x4 gd[NB == 2 ? IT : 1][8];
    auto item_of = [&](int it, int& rb, int& c0) -> bool {   // it-th item of this lane: fragment row block, first atom
      const int q = wave * IPW + 64 * it + lane;
      const bool ok = 64 * it + lane < IPW;
      const int qq = ok ? q : wave * IPW;      // idle lanes repeat a valid address
      rb = qq >> 4;
      c0 = 2 * (qq & 15);
      return ok;
    };
    int g_ch = 0;   // chunk whose table entries gd holds (XC: the spare row needs the atoms' u2)
    auto gen_load = [&](int ch) {
      g_ch = ch;
#pragma unroll
      for (int it = 0; it < (NB == 2 ? IT : 0); ++it) {
        int rb, c0;
        item_of(it, rb, c0);
        const int nn = min(ch * 32 + c0, ldn - 2);
        const i32x4 r0 = *(const i32x4*)(s_rs + MP + 8 * rb), r1 = *(const i32x4*)(s_rs + MP + 8 * rb + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { gd[it][e] = tab32x2_at(r0[e], nn); gd[it][4 + e] = tab32x2_at(r1[e], nn); }
      }
    };
    // conversion of rows e0 .. e0+3 of item `it` into halves (a slice of the generation work)
    h8 g_hi0[NB == 2 ? IT : 1], g_lo0[NB == 2 ? IT : 1], g_hi1[NB == 2 ? IT : 1], g_lo1[NB == 2 ? IT : 1];
    auto gen_convert = [&](auto itc, auto ec) {
      constexpr int it = decltype(itc)::value;
      constexpr int e0 = decltype(ec)::value;
      int rb, c0;
      item_of(it, rb, c0);
      const f32x4 tq = *(const f32x4*)(s_t0f + MP + 8 * rb + e0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 x, y;
        float v0 = fmaf(gd[it][e0 + e][1], tq[e], gd[it][e0 + e][0]), v1 = fmaf(gd[it][e0 + e][3], tq[e], gd[it][e0 + e][2]);
        if constexpr (XC) {   // the spare row (last row of the last fragment block) carries u2 of the item's two atoms
          if (e0 + e == 7) { const bool spare = (rb == 2 * KS - 1); v0 = spare ? s_uf[NP + g_ch * 32 + c0] : v0; v1 = spare ? s_uf[NP + g_ch * 32 + c0 + 1] : v1; }
        }
        mfx_split16(v0, x, y);
        g_hi0[it][e0 + e] = x; g_lo0[it][e0 + e] = y;
        mfx_split16(v1, x, y);
        g_hi1[it][e0 + e] = x; g_lo1[it][e0 + e] = y;
      }
    };
    auto gen_write = [&](auto itc, int buf) {
      constexpr int it = decltype(itc)::value;
      int rb, c0;
      if (item_of(it, rb, c0)) {
        const int off = (rb * 32 + c0) << 3;
        _Float16* dh = sBh + buf * KS * 512 + off;
        _Float16* dl = sBl + buf * KS * 512 + off;
        *(h8*)dh = g_hi0[it]; *(h8*)(dh + 8) = g_hi1[it];
        *(h8*)dl = g_lo0[it]; *(h8*)(dl + 8) = g_lo1[it];
      }
    };
    auto gen_store_all = [&](int buf) {   // un-sliced (prologue of the NB = 2 schedule)
      mfx_static_for<0, (NB == 2 ? IT : 0)>([&](auto itc) {
        gen_convert(itc, std::integral_constant<int, 0>{});
        gen_convert(itc, std::integral_constant<int, 4>{});
        gen_write(itc, buf);
      });
    };
    // NB = 1: the whole image of chunk ch in one go, item after item, the loads of the next item in flight while the
    // current one is converted (two register stages instead of IT)
    auto gen_direct = [&](int ch) {
      f32x4 st[2][8];
      auto ld = [&](auto itc, auto pc) {
        constexpr int it = decltype(itc)::value;
        constexpr int p = decltype(pc)::value;
        int rb, c0;
        item_of(it, rb, c0);
        const int nn = min(ch * 32 + c0, ldn - 2);
        const i32x4 r0 = *(const i32x4*)(s_rs + MP + 8 * rb), r1 = *(const i32x4*)(s_rs + MP + 8 * rb + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { st[p][e] = tab32x2_at(r0[e], nn); st[p][4 + e] = tab32x2_at(r1[e], nn); }
      };
      ld(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      mfx_static_for<0, IT>([&](auto itc) {
        constexpr int it = decltype(itc)::value;
        if constexpr (it + 1 < IT) ld(std::integral_constant<int, it + 1>{}, std::integral_constant<int, (it + 1) & 1>{});
        int rb, c0;
        const bool ok = item_of(it, rb, c0);
        h8 hi0, lo0, hi1, lo1;
        const f32x4 ta = *(const f32x4*)(s_t0f + MP + 8 * rb), tb = *(const f32x4*)(s_t0f + MP + 8 * rb + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float tq = e < 4 ? ta[e & 3] : tb[e & 3];
          _Float16 x, y;
          float v0 = fmaf(st[it & 1][e][1], tq, st[it & 1][e][0]), v1 = fmaf(st[it & 1][e][3], tq, st[it & 1][e][2]);
          if constexpr (XC) {
            if (e == 7) { const bool spare = (rb == 2 * KS - 1); v0 = spare ? s_uf[NP + ch * 32 + c0] : v0; v1 = spare ? s_uf[NP + ch * 32 + c0 + 1] : v1; }
          }
          mfx_split16(v0, x, y);
          hi0[e] = x; lo0[e] = y;
          mfx_split16(v1, x, y);
          hi1[e] = x; lo1[e] = y;
        }
        if (ok) {
          const int off = (rb * 32 + c0) << 3;
          *(h8*)(sBh + off) = hi0; *(h8*)(sBh + off + 8) = hi1;
          *(h8*)(sBl + off) = lo0; *(h8*)(sBl + off + 8) = lo1;
        }
      });
    };

    f32x16 accs[2][TL];   // double-buffered by chunk parity: chunk c accumulates into accs[c & 1] while accs[(c-1) & 1] is screened
    // VALU slices of a chunk: TL x 4 screen groups of the previous chunk + IT x 2 conversion halves + IT writes of the next
    constexpr int NSCR = TL * 4, NGEN = (NB == 2 ? IT * 2 : 0);
    constexpr int NSL = NSCR + NGEN;
    float mmv[TL][4];
    // slice s of the VALU work that rides along chunk c's MFMAs: screen of chunk c-1 (its accumulators: accs[P ^ 1]),
    // conversion of chunk c+1's items
    auto valu_slice = [&](auto sc, auto pc, bool do_scr, bool do_gen, int bufn) {
      constexpr int s = decltype(sc)::value;
      constexpr int P = decltype(pc)::value;
      if constexpr (s < NSCR) {
        constexpr int t = s / 4, q = s % 4;
        if (do_scr && rtv[t]) mmv[t][q] = scan_group(accs[P ^ 1][t], std::integral_constant<int, t>{}, std::integral_constant<int, q>{});
      } else if constexpr (s < NSL) {
        constexpr int u = s - NSCR, it = u / 2, half = u % 2;
        if (do_gen) {
          gen_convert(std::integral_constant<int, it>{}, std::integral_constant<int, 4 * half>{});
          if constexpr (half == 1) gen_write(std::integral_constant<int, it>{}, bufn);
        }
      }
    };
    // the TL x 3 KS MFMAs of chunk c (image in buffer buf), the VALU slices spread evenly over the k-steps
    auto mfma_chunk = [&](auto pc, int buf, bool do_scr, bool do_gen, int bufn) {
      constexpr int P = decltype(pc)::value;
#pragma unroll
      for (int t = 0; t < TL; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) accs[P][t][g] = 0.0f;
      const _Float16* bhp = sBh + buf * KS * 512 + lane * 8;
      const _Float16* blp = sBl + buf * KS * 512 + lane * 8;
      h8 bh = *(const h8*)bhp, bl = *(const h8*)blp;
      mfx_static_for<0, KS>([&](auto kc) {
        constexpr int ks = decltype(kc)::value;
        h8 bhn = bh, bln = bl;
        if constexpr (ks + 1 < KS) {   // fragments of the next k-step while this one multiplies
          bhn = *(const h8*)(bhp + (ks + 1) * 512);
          bln = *(const h8*)(blp + (ks + 1) * 512);
        }
        mfx_static_for<0, TL>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          accs[P][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[t][ks], bh, accs[P][t], 0, 0, 0);
        });
        mfx_static_for<0, TL>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          accs[P][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[t][ks], bl, accs[P][t], 0, 0, 0);
        });
        mfx_static_for<0, TL>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          accs[P][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afl[t][ks], bh, accs[P][t], 0, 0, 0);
        });
        // slices [ks NSL / KS, (ks+1) NSL / KS)
        constexpr int s0 = ks * NSL / KS, s1 = (ks + 1) * NSL / KS;
        mfx_static_for<s0, s1>([&](auto sc) { valu_slice(sc, pc, do_scr, do_gen, bufn); });
        bh = bhn; bl = bln;
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    // what is left of chunk c-1's screen after the slices: the FP64 criteria of the flagged groups (rare)
    auto screen_finish = [&](auto pc, int ct) {
      constexpr int P = decltype(pc)::value;
      mfx_static_for<0, TL>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if (rtv[t] && __any(fmaxf(fmaxf(mmv[t][0], mmv[t][1]), fmaxf(mmv[t][2], mmv[t][3])) >= 0.0f))
          scan_exact(accs[P ^ 1][t], rts[t], ct, mmv[t]);
      });
    };
    auto screen_begin = [&](int ct) {   // column / row constants for the screen of chunk ct (LDS reads fly behind the first MFMAs)
      scan_cols(ct);
      mfx_static_for<0, TL>([&](auto tc) { if (rtv[decltype(tc)::value]) scan_rows(tc); });
    };

    if (round == 0) MFX_STAMP(3);
    thr = __longlong_as_double((long long)s_thr[0]);
    if constexpr (NB == 2) {
      // one barrier per chunk: during chunk c (buffer c & 1) the slices convert chunk c+1 into buffer (c+1) & 1, which
      // every wave has finished reading before the barrier that opened chunk c
      gen_load(0);
      gen_store_all(0);
      if (ntiles > 1) gen_load(1);
      __syncthreads();
      if (round == 0) MFX_STAMP(4);
#ifdef MFX_STAMPS_W   // diagnostic: where a chunk's time goes (wave 0, chunks 10 and 11 of round 1), tools/dev_stamps_w.py
#define MFX_WSTAMP(k) do { if (a.stamps && round == 1 && (c == 10 || c == 11) && tid == 0) a.stamps[(size_t)blockIdx.x * 16 + (c - 10) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MFX_WSTAMP(k) do { } while (0)
#endif
      auto chunk = [&](auto pc, int c) {
        const bool do_scr = c >= 1, do_gen = c + 1 < ntiles;
        MFX_WSTAMP(0);
        if (do_scr) screen_begin(c - 1);
        MFX_WSTAMP(1);
        mfma_chunk(pc, c & 1, do_scr, do_gen, (c + 1) & 1);
        MFX_WSTAMP(2);
        if (do_scr) screen_finish(pc, c - 1);
        MFX_WSTAMP(3);
        if (c + 2 < ntiles) gen_load(c + 2);   // consumed by the slices of chunk c+1
        MFX_WSTAMP(4);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier: the table loads stay in flight
        MFX_WSTAMP(5);
      };
      for (int c = 0; c < ntiles; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        if (c + 1 < ntiles) chunk(std::integral_constant<int, 1>{}, c + 1);
      }
      // screen of the last chunk
      {
        const int c = ntiles - 1;
        if (c & 1) { mfx_static_for<0, TL>([&](auto tc) { if (rtv[decltype(tc)::value]) scan_tile(accs[1][decltype(tc)::value], tc, c); }); }
        else { mfx_static_for<0, TL>([&](auto tc) { if (rtv[decltype(tc)::value]) scan_tile(accs[0][decltype(tc)::value], tc, c); }); }
      }
      __syncthreads();   // all appends of the round are in the ring; the images are free
    } else {
      // one image: convert, barrier, multiply (+ screen of the previous chunk in the MFMA shadows), barrier
      auto chunk = [&](auto pc, int c) {
        gen_direct(c);
        __syncthreads();
        const bool do_scr = c >= 1;
        if (do_scr) screen_begin(c - 1);
        mfma_chunk(pc, 0, do_scr, false, 0);
        if (do_scr) screen_finish(pc, c - 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      };
      for (int c = 0; c < ntiles; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        if (c + 1 < ntiles) chunk(std::integral_constant<int, 1>{}, c + 1);
      }
      {
        const int c = ntiles - 1;
        if (c & 1) { mfx_static_for<0, TL>([&](auto tc) { if (rtv[decltype(tc)::value]) scan_tile(accs[1][decltype(tc)::value], tc, c); }); }
        else { mfx_static_for<0, TL>([&](auto tc) { if (rtv[decltype(tc)::value]) scan_tile(accs[0][decltype(tc)::value], tc, c); }); }
      }
      __syncthreads();
    }
    if (round == 0) MFX_STAMP(5);
  }
  // (the thread index is re-derived here instead of being kept - spilled, 4 KB of scratch per voxel - across the sweep)
  tid = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  // D1's best single atom (first index on ties) joins the candidates, like D2's after phase 1
  {
    double* s_bs = (double*)smem + 64;            // [8]  inside the chunk images, idle from here on (a constant
    int* s_bn = (int*)((double*)smem + 80);       // [8]  address: nothing to keep in a register across the sweep)
    if (lane == 0) { s_bs[wave] = bs1; s_bn[wave] = bn1; }
    __syncthreads();
    if (tid == 0) {
      double sb = s_bs[0];
      int nb = s_bn[0];
      for (int w = 1; w < NW; ++w) {
        const double s2 = s_bs[w];
        const int n2 = s_bn[w];
        if (s2 > sb || (s2 == sb && n2 < nb)) { sb = s2; nb = n2; }
      }
      s_cnt[3] = -1;
      if (sb > 0.0) { s_cnt[3] = s_cnt[0] & (a.scap - 1); push(sb + mrg, nb, MFX_S_BOUND); }   // [3]: its slot (diagnostics)
    }
    __syncthreads();
  }

  MFX_STAMP(6);
  // ---- exact stage (same as fit_k2.hip phase 3): reference arithmetic and order on the short list
  auto exact_pair = [&](int i, int j, double& w0, double& w1, double& res) {
    double a11 = 0.0, a22 = 0.0, a12 = 0.0, y1 = 0.0, y2 = 0.0;
#pragma unroll 4
    for (int m = 0; m < M; ++m) {
      const double d1 = elem(0, m, i), d2 = elem(1, m, j), ym = s_y[m];
      a11 += d1 * d1;
      a22 += d2 * d2;
      a12 += d1 * d2;
      y1 += ym * d1;
      y2 += ym * d2;
    }
    nnls2_exact(y_sq, a11, a12, a22, y1, y2, w0, w1, res);
  };
  double* s_rres = (double*)smem;        // [8] per-wave partials (B buffers are idle now)
  long* s_ridx = (long*)(s_rres + 8);    // [8]
  double* s_rw = (double*)(s_ridx + 8);  // [8][2]
  double* s_win = s_rw + 16;             // winner: res, w0, w1, (long) idx
  auto block_argmin = [&](double res, long idx, double w0, double w1) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double r2 = __shfl_xor(res, o), u0 = __shfl_xor(w0, o), u1 = __shfl_xor(w1, o);
      const long i2 = __shfl_xor(idx, o);
      const bool take = (r2 < res) || (r2 == res && i2 < idx);
      res = take ? r2 : res; idx = take ? i2 : idx; w0 = take ? u0 : w0; w1 = take ? u1 : w1;
    }
    __syncthreads();
    if (lane == 0) { s_rres[wave] = res; s_ridx[wave] = idx; s_rw[2 * wave] = w0; s_rw[2 * wave + 1] = w1; }
    __syncthreads();
    if (tid == 0) {
      double br = s_win[0], b0 = s_win[1], b1 = s_win[2];
      long bi = ((long*)s_win)[3];
      for (int w = 0; w < NW; ++w) {
        const double r = s_rres[w];
        const long ix = s_ridx[w];
        if (ix < 0) continue;
        if (r < br || (r == br && bi >= 0 && ix < bi)) { br = r; bi = ix; b0 = s_rw[2 * w]; b1 = s_rw[2 * w + 1]; }
      }
      s_win[0] = br; s_win[1] = b0; s_win[2] = b1; ((long*)s_win)[3] = bi;
    }
    __syncthreads();
  };
  const int nappend = s_cnt[0];
  const int ncand = nappend > a.scap ? a.scap : nappend;
  const double thr_fin = __longlong_as_double((long long)s_thr[0]);
  const double lost = __longlong_as_double((long long)s_thr[1]);
  const int xc_flag = XC ? s_cnt[1] : 0;   // an atom (nearly) inside span(x)
  __syncthreads();   // everyone has read the counters / is done with the B buffers
  if ((nappend > a.scap && lost >= thr_fin) || xc_flag) {
    // an entry that could still matter was overwritten: hand the voxel to the FP64 kernel
    if (tid == 0) {
      const int slot = atomicAdd(a.fb_count, 1);
      a.fb_list[slot] = vox;
      if constexpr (XC) a.xl_cnt[blockIdx.x] = -1;
    }
    return;
  }
  if constexpr (XC) {
    // short list of the voxel for fit_k2x.hip's exact stage: ring entries that reach the final threshold, then the single
    // atoms whose one-atom supports could tie with the optimum or whose relaxed score bounds all their pairs ("forced")
    int* s_evl = (KS >= 8) ? (int*)((char*)smem + 2048) : s_evl4;
    if (tid == 0) s_cnt[2] = 0;
    __syncthreads();
    for (int cix = tid; cix < ncand; cix += WG)
      if (s_cand[cix].score >= thr_fin) s_evl[atomicAdd(&s_cnt[2], 1)] = cix;
    __syncthreads();
    const int neval = s_cnt[2];
    Cand* dst = a.xl_cand + (size_t)blockIdx.x * a.xl_cap;
    for (int e = tid; e < min(neval, a.xl_cap); e += WG) {
      Cand c = s_cand[s_evl[e]];
      c.score += yx * yx;
      c.j &= ~MFX_S_BOUND;
      dst[e] = c;
    }
    __syncthreads();
    const double tcut = thr_fin - 2.0 * mrg;
    for (int q = tid; q < 2 * NP; q += WG) {
      const int k = q >= NP, n = q - k * NP;
      const double npr = (double)s_cs[q];
      if (n < N && npr > 0.0) {
        const double zp = (double)s_Zf[q], u = (double)s_uf[q];
        const double ayv = fma(u, yx, zp * npr);
        double s1 = (ayv > 0.0 ? ayv * ayv / fma(u, u, npr * npr) : 0.0) - yx * yx;
        if (zp > -sqrt(mrg)) s1 = fmax(s1, zp > 0.0 ? zp * zp : 0.0);
        const bool forced = zp > 0.0 && zp * zp >= tcut;
        if (s1 >= tcut || forced) {
          const int slot = atomicAdd(&s_cnt[2], 1);
          const int mark = forced ? -2 : -1;
          if (slot < a.xl_cap) { Cand c; c.score = s1 + yx * yx; c.i = k ? mark : n; c.j = k ? n : mark; dst[slot] = c; }
        }
      }
    }
    __syncthreads();
    const int nall = s_cnt[2];
    if (tid == 0) {
      if (nall > a.xl_cap) {   // too many near-ties for the list: the FP64 kernel of the class decides
        const int slot = atomicAdd(a.fb_count, 1);
        a.fb_list[slot] = vox;
        a.xl_cnt[blockIdx.x] = -1;
      } else {
        a.xl_cnt[blockIdx.x] = nall;
        a.xl_mrg[blockIdx.x] = mrg;
      }
    }
    return;
  }
  if (tid == 0) {    // mf_utils.py:327, 382: start from min_obj = y_sq at pair (0,0) with w = 0, strict '<'
    s_win[0] = y_sq; s_win[1] = 0.0; s_win[2] = 0.0; ((long*)s_win)[3] = -1;
  }
  {
    double res = INFINITY, w0 = 0.0, w1 = 0.0;
    long idx = -1;
#ifdef MFX_STAMPS
    double dbg_err = 0.0;
    int dbg_eval = 0;
#endif
    // compact list of the ring entries that reach the final threshold
    // [MFX_S_CAP] compaction list: inside the idle hi image (behind s_win / s_yrec) when that is large enough
    int* s_evl = (KS >= 8) ? (int*)((char*)smem + 2048) : s_evl4;
    double* s_stage = (double*)sBl + (size_t)wave * 2 * MP;       // [2][MP] per wave, inside the idle lo image
    if (tid == 0) s_cnt[2] = 0;
    __syncthreads();
    for (int cix = tid; cix < ncand; cix += WG)
      if (s_cand[cix].score >= thr_fin) s_evl[atomicAdd(&s_cnt[2], 1)] = cix;
    __syncthreads();
    const int neval = s_cnt[2];
    MFX_STAMP(9);
    if (neval <= 24) {
      // few candidates (the usual case): one WAVE per candidate.  A thread-per-candidate loop is bound by the
      // latency of its 2 x 200 dependent-address table loads (51 k cycles whatever the count); here the 64 lanes
      // fetch the rows side by side, then five lanes run the five sequential sums of mf_utils.py:307-325 from LDS.
      for (int e = wave; e < neval; e += NW) {
        const int cix = s_evl[e];
        const int i = s_cand[cix].i, jf = s_cand[cix].j, j = jf & ~MFX_S_BOUND;
#pragma unroll
        for (int mb = 0; mb < (MP + 63) / 64; ++mb) {   // all table loads of the pair in flight at once
          const int m = mb * 64 + lane;
          if (m < M) {
            s_stage[m] = elem(0, m, i);
            s_stage[MP + m] = elem(1, m, j);
          }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (e == 0) MFX_STAMP(15);
        // lane 0: a11 = sum d1*d1, 1: a22 = sum d2*d2, 2: a12 = sum d1*d2, 3: y1 = sum y*d1, 4: y2 = sum y*d2
        const double* pa = (lane == 1) ? s_stage + MP : (lane >= 3 ? s_y : s_stage);
        const double* pb = (lane == 0 || lane == 3) ? s_stage : s_stage + MP;
        double acc = 0.0;
        if (lane < 5) {
          // blocks of 8 rows: sixteen 16-byte LDS reads in flight, then the 8 dependent multiply-adds in row order
          // (a read per term leaves its ~100-cycle round trip exposed 200 times: 21 k cycles per candidate)
          int m = 0;
          for (; m + 8 <= M; m += 8) {
            double2 va[4], vb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { va[q] = *(const double2*)(pa + m + 2 * q); vb[q] = *(const double2*)(pb + m + 2 * q); }
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc += va[q].x * vb[q].x; acc += va[q].y * vb[q].y; }
          }
          for (; m < M; ++m) acc += pa[m] * pb[m];
        }
        const double a11 = mfx_readlane_f64(acc, 0), a22 = mfx_readlane_f64(acc, 1), a12 = mfx_readlane_f64(acc, 2),
                     y1 = mfx_readlane_f64(acc, 3), y2 = mfx_readlane_f64(acc, 4);
        double r, u0, u1;
        nnls2_exact(y_sq, a11, a12, a22, y1, y2, u0, u1, r);
        const long ix = (long)i * N + j;
        if (r < res || (r == res && ix < idx)) { res = r; idx = ix; w0 = u0; w1 = u1; }
        // run-time guard on the screening error (wave-uniform values)
        if (!(jf & MFX_S_BOUND) && u0 > 0.0 && u1 > 0.0 && fabs((y_sq - r) - s_cand[cix].score) > MFX_S_GUARD * mrg) s_cnt[1] = 1;
#ifdef MFX_STAMPS
        if (lane == 0) {
          ++dbg_eval;
          if (cix >= 1 && cix != s_cnt[3] && s_cand[cix].score < 1e299 && u0 > 0.0 && u1 > 0.0) dbg_err = fmax(dbg_err, fabs((y_sq - r) - s_cand[cix].score) / y_sq);
        }
#endif
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      for (int e = tid; e < neval; e += WG) {
        const int cix = s_evl[e];
        double r, u0, u1;
        const int i = s_cand[cix].i, jf = s_cand[cix].j, j = jf & ~MFX_S_BOUND;
        exact_pair(i, j, u0, u1, r);
        const long ix = (long)i * N + j;
        if (r < res || (r == res && ix < idx)) { res = r; idx = ix; w0 = u0; w1 = u1; }
        if (!(jf & MFX_S_BOUND) && u0 > 0.0 && u1 > 0.0 && fabs((y_sq - r) - s_cand[cix].score) > MFX_S_GUARD * mrg) s_cnt[1] = 1;
#ifdef MFX_STAMPS
        ++dbg_eval;
        if (cix >= 1 && cix != s_cnt[3] && s_cand[cix].score < 1e299 && u0 > 0.0 && u1 > 0.0) dbg_err = fmax(dbg_err, fabs((y_sq - r) - s_cand[cix].score) / y_sq);
#endif
      }
    }
#ifdef MFX_STAMPS
    MFX_STAMP(13);   // before the diagnostics below: 512 global atomics would count as exact-stage time
    if (a.stamps) {
      if (dbg_err > 0.0) atomicMax(&a.stamps[(size_t)blockIdx.x * 16 + 10], (unsigned long long)__double_as_longlong(dbg_err));
      if (dbg_eval > 0) atomicAdd(&a.stamps[(size_t)blockIdx.x * 16 + 11], (unsigned long long)dbg_eval);
      if (tid == 0) a.stamps[(size_t)blockIdx.x * 16 + 12] = (unsigned long long)nappend;
    }
#else
    MFX_STAMP(13);
#endif
    block_argmin(res, idx, w0, w1);
    MFX_STAMP(14);
  }
  if (s_cnt[1]) {   // workgroup-uniform (block_argmin ends with a barrier)
    // the split-FP16 Gram missed an exactly evaluated pair by more than the guard allows: do not trust the short
    // list, let the FP64 kernel redo the voxel
    if (tid == 0) {
      const int slot = atomicAdd(a.fb_count, 1);
      a.fb_list[slot] = vox;
      atomicAdd(a.fb_count + 1, 1);
    }
    return;
  }
  // near-zero second weight: evaluate the winner's whole row / column family exactly (see fit_k2.hip)
  for (int pass = 0; pass < 2; ++pass) {
    const double bw0 = s_win[1], bw1 = s_win[2];
    const long bidx = ((long*)s_win)[3];
    if (bidx < 0) break;
    const int bi = (int)(bidx / N), bj2 = (int)(bidx - (long)bi * N);
    const bool row_family = (pass == 0) && (bw1 <= 1e-7 * bw0);
    const bool col_family = (pass == 1) && (bw0 <= 1e-7 * bw1);
    if (!row_family && !col_family) continue;
    double res = INFINITY, w0 = 0.0, w1 = 0.0;
    long idx = -1;
    for (int n = tid; n < N; n += WG) {
      double r, u0, u1;
      const int i = row_family ? bi : n, j = row_family ? n : bj2;
      exact_pair(i, j, u0, u1, r);
      const long ix = (long)i * N + j;
      if (r < res || (r == res && ix < idx)) { res = r; idx = ix; w0 = u0; w1 = u1; }
    }
    block_argmin(res, idx, w0, w1);
  }
  MFX_STAMP(7);
  if (wave == 0) {
    const double best = s_win[0], w0 = s_win[1], w1 = s_win[2];
    const long bidx = ((long*)s_win)[3];
    const int bi = bidx < 0 ? 0 : (int)(bidx / N);
    const int bjx = bidx < 0 ? 0 : (int)(bidx - (long)bi * N);
    // params packing, mf.py:420-450
    const double M0 = w0 + w1;
    const double nu0 = (fabs(M0) > 0) ? w0 / M0 : w0;
    const double nu1 = (fabs(M0) > 0) ? w1 / M0 : w1;
    double* s_yrec = s_win + 8;  // [MP] scratch inside the (now idle) B buffers
    double sy = 0.0, sr = 0.0;
    for (int m = lane; m < M; m += 64) {
      const double yr = elem(0, m, bi) * w0 + elem(1, m, bjx) * w1;
      s_yrec[m] = yr;
      sy += s_y[m];
      sr += yr;
    }
    sy = wave_sum(sy) / M;
    sr = wave_sum(sr) / M;
    double cyy = 0.0, crr = 0.0, cyr = 0.0;
    for (int m = lane; m < M; m += 64) {
      const double da = s_y[m] - sy, db = s_yrec[m] - sr;
      cyy += da * da;
      crr += db * db;
      cyr += da * db;
    }
    cyy = wave_sum(cyy);
    crr = wave_sum(crr);
    cyr = wave_sum(cyr);
    double r2 = 0.0;
    if (M > 1 && cyy > 0.0 && crr > 0.0) {
      const double f = (double)(M - 1);
      double r = (cyr / f) / sqrt(cyy / f) / sqrt(crr / f);
      r = r > 1.0 ? 1.0 : (r < -1.0 ? -1.0 : r);
      r2 = r * r;
    }
    double* out = a.params + (size_t)vox * a.num_params;
    if (lane == 0) {
      out[0] = M0;
      out[1] = nu0;
      out[2] = nu1;
      out[1 + a.maxfasc] = (double)bi;
      out[2 + a.maxfasc] = (double)bjx;
      out[a.num_params - 2] = best / M;
      out[a.num_params - 1] = r2;
    }
  }
  MFX_STAMP(8);
}
