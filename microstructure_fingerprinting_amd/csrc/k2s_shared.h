// k2s_shared.h -- what the two split-FP16 screening kernels have in common besides their mathematics: the LDS layout, the
// per-voxel prologue (signal, knot-interval descriptors, column statistics of the rotated dictionaries, margins, starting
// threshold), the ring append and everything after the sweep (exact stage on the short list in the reference's arithmetic,
// short lists of the [N, N, 1] form, parameter packing).  fit_k2s.hip (two waves per SIMD, one row tile per wave) and
// fit_k2w.hip (one wave per SIMD, TL row tiles per wave) differ in their sweeps only; a parity fix lands here once.
// Included by fit_k2s.hip after its helpers (mfx_split16, mfx_readlane_f64, ...).
#pragma once

// LDS layout.  KS: k-steps of 16 measurements, NB: chunk images, BR: G-bracketed protocol, XC: one fixed extra column,
// PQF: floats of pair-screen row constants (all waves together)
template <int KS, int NB, bool BR, bool XC, int PQF>
struct K2sLds {
  static constexpr int MP = KS * 16;   // padded measurement count
  double* smem;
  _Float16* sBh;   // [NB][KS][64][8]  hi halves, fragment order
  _Float16* sBl;   // [NB][KS][64][8]  lo halves
  double* s_y;   // [MP]
  double* s_t0;   // [2][MP]
  double* s_red;   // [32] scratch
  Cand* s_cand;   // [MFX_S_CAP]
  unsigned long long* s_thr;   // [0] threshold bits, [1] lost-entry max bits
  int* s_r0;   // [2][MP] knot row * ldn (element offset of the row in the table)
  int* s_cnt;   // [4]
  float* s_t0f;   // [2][MP] FP32 copy of s_t0 for the screening passes
  float* s_Zf;   // [2][NP] Z1 | Z2 = d.y/|d| of the rotated atoms (-1e30 beyond N)
  float* s_cs;   // [2][NP] |d1| | |d2| (FP32, table units; 0: no such atom): accumulator = cosine |d1| |d2|
  float* s_yf;   // [MP] FP32 copy of y (ranking statistics)
  float* s_pq;   // [PQF] pair-screen constants of each wave's rows (fit_k2s.hip: [NW][2][32]; fit_k2w.hip: [NW][TL][2][32])
  double* s_t1;   // [2][MP]
  double* s_tG;   // [MP]
  double* s_dG;   // [MP]
  int* s_r1;   // [2][MP] upper-shell knot row * ldn, or -1
  int* s_rs;   // [2][MP] row offsets used by the screening passes
  int* s_evl4;   // [MFX_S_CAP] exact-stage compaction list, KS < 8 only (else inside the B image)
  float* s_xf;   // XC: [MP] x^ (unit extra column, 0 beyond M)
  float* s_uf;   // XC: [2][NP] u = d.x^ of the rotated atoms
  __device__ __forceinline__ K2sLds(double* smem_, int NP) : smem(smem_) {
    sBh = (_Float16*)smem;
    sBl = sBh + NB * KS * 512;
    s_y = (double*)(sBl + NB * KS * 512);
    s_t0 = s_y + MP;
    s_red = s_t0 + 2 * MP;
    s_cand = (Cand*)(s_red + 32);
    s_thr = (unsigned long long*)(s_cand + MFX_S_CAP);
    s_r0 = (int*)(s_thr + 2);
    s_cnt = s_r0 + 2 * MP;
    s_t0f = (float*)(s_cnt + 4);
    s_Zf = s_t0f + 2 * MP;
    s_cs = s_Zf + 2 * NP;
    s_yf = s_cs + 2 * NP;
    s_pq = s_yf + MP;
    s_t1 = (double*)(s_pq + PQF);
    s_tG = s_t1 + (BR ? 2 * MP : 0);
    s_dG = s_tG + (BR ? MP : 0);
    s_r1 = (int*)(s_dG + (BR ? MP : 0));
    s_rs = BR ? s_r1 + 2 * MP : s_r0;
    s_evl4 = s_r1 + (BR ? 4 * MP : 0);
    s_xf = (float*)(s_evl4 + (KS < 8 ? MFX_S_CAP : 0));
    s_uf = s_xf + (XC ? MP : 0);
  }
};
// the layout's pointers under the names the kernels use
#define K2S_UNPACK(L) \
  [[maybe_unused]] _Float16* const sBh = (L).sBh; \
  [[maybe_unused]] _Float16* const sBl = (L).sBl; \
  [[maybe_unused]] double* const s_y = (L).s_y; \
  [[maybe_unused]] double* const s_t0 = (L).s_t0; \
  [[maybe_unused]] double* const s_red = (L).s_red; \
  [[maybe_unused]] Cand* const s_cand = (L).s_cand; \
  [[maybe_unused]] unsigned long long* const s_thr = (L).s_thr; \
  [[maybe_unused]] int* const s_r0 = (L).s_r0; \
  [[maybe_unused]] int* const s_cnt = (L).s_cnt; \
  [[maybe_unused]] float* const s_t0f = (L).s_t0f; \
  [[maybe_unused]] float* const s_Zf = (L).s_Zf; \
  [[maybe_unused]] float* const s_cs = (L).s_cs; \
  [[maybe_unused]] float* const s_yf = (L).s_yf; \
  [[maybe_unused]] float* const s_pq = (L).s_pq; \
  [[maybe_unused]] double* const s_t1 = (L).s_t1; \
  [[maybe_unused]] double* const s_tG = (L).s_tG; \
  [[maybe_unused]] double* const s_dG = (L).s_dG; \
  [[maybe_unused]] int* const s_r1 = (L).s_r1; \
  [[maybe_unused]] int* const s_rs = (L).s_rs; \
  [[maybe_unused]] int* const s_evl4 = (L).s_evl4; \
  [[maybe_unused]] float* const s_xf = (L).s_xf; \
  [[maybe_unused]] float* const s_uf = (L).s_uf;

// what the prologue hands to the sweep and to the exact stage (wave-uniform: scalar registers)
struct K2sVoxel {
  double y_sq;     // |y|^2 summed sequentially (mf_utils.py:320-325)
  double yx;       // XC: y.x^ (0 otherwise)
  double ramp;     // XC: amplification of the screening error by the projection (1 otherwise)
  double dc_eff;   // bound on |c~ - c| in the units of the pair test
  double mrg;      // |S(c~) - S(c)| <= mrg
  double etol;     // |e(c~) - e(c)| <= etol
};

// Population audit of the screening product.  The run-time guard of the exact stage sees only pairs that were short-listed;
// a pair that was NOT listed because its c~ was off by more than the margin would go unnoticed.  So every voxel also
// audits ONE pseudo-random pair of its N^2 (a hash of the voxel index picks row tile, column chunk, accumulator register
// and lane): the wave that screens that accumulator tile parks the raw accumulator value in LDS, the exact stage sums the
// pair's Gram scalars in FP64 from the FP64 table and compares the two cosines.  Counted per launch (FitK2Args::audit): a
// 1e5-voxel launch audits 1e5 pairs of real operands of that very run, whatever was listed.
__device__ __forceinline__ unsigned k2s_audit_hash(int vox) {
  unsigned h = (unsigned)vox * 2654435761u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  return h;
}
// (row tile << 8) | column chunk of the audited pair, or -1: row tiles of the regular rounds only (ntiles_reg of them)
__device__ __forceinline__ int k2s_audit_key(unsigned h, int ntiles_reg, int ntiles) {
  return ntiles_reg > 0 ? (int)(((h & 0xffu) % (unsigned)ntiles_reg) << 8) | (int)(((h >> 8) & 0xffu) % (unsigned)ntiles) : -1;
}

// ring append (rare path)
template <class LDS>
__device__ __forceinline__ void k2s_push(const LDS& L, int scap, double S, int i, int j) {
  const int slot = atomicAdd(&L.s_cnt[0], 1);
  const int idx = slot & (scap - 1);
  if (slot >= scap) {   // overwriting: remember the best score that got lost
    const double old = L.s_cand[idx].score;
    atomicMax(&L.s_thr[1], mfx_nonneg_bits(fmin(old, 1e300)));
  }
  L.s_cand[idx].score = S;
  L.s_cand[idx].i = i;
  L.s_cand[idx].j = j;
}

// ---- phases 0 and 1 of a voxel: signal and knot-interval descriptors -> LDS, column statistics, margins, starting threshold
template <int KS, int NB, bool BR, bool XC, int PQF, int WG>
__device__ __forceinline__ K2sVoxel k2s_prologue(const FitK2Args& a, const K2sLds<KS, NB, BR, XC, PQF>& L, const int vox, const int tid) {
  constexpr int MP = KS * 16, NW = WG / 64;
  const int lane = tid & 63, wave = tid >> 6;
  const int M = a.P.M, N = a.T.N, ldn = a.T.ldn;
  const int NP = (N + 31) & ~31;
  [[maybe_unused]] double* const smem = L.smem;
  K2S_UNPACK(L);

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
  if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; s_thr[0] = 0ull; s_thr[1] = 0ull; ((float*)(s_red + 30))[0] = __int_as_float(0x7fc00000); }   // (audit slot: nothing parked yet)
  if constexpr (XC) {
    for (int m = tid; m < MP; m += WG) s_xf[m] = (m < M) ? (float)a.xc[m] : 0.0f;
    if (tid < 2) ((unsigned long long*)s_red)[24 + tid] = 0ull;   // max |d|^2/|d'|^2 of each dictionary (bits of a non-negative double)
  }
  if (tid < 2) mfx_check_dir(a.P, pk + 3 * tid, vox);
  __syncthreads();

  // the FP32 copy of the table feeds everything that only RANKS pairs (half the L2 -> CU bytes); two adjacent atoms (n even)
  // in one 16-byte load: {ylo_n, slope_n, ylo_n+1, slope_n+1}
  const float2* __restrict__ tab32 = a.P.tab32s;   // == a.T.tab32 unless the plan has virtual shells
  auto tab32x2_at = [&](int ro, int n) -> f32x4 { return *(const f32x4*)((const char*)tab32 + ((unsigned)(ro + n) << 3)); };

  MFX_STAMP(1);
  // ---- phase 1: column statistics; y_sq sequential as mf_utils.py:307-325
  double y_sq_v = 0.0;
  for (int m = 0; m < M; ++m) y_sq_v += s_y[m] * s_y[m];
  // wave-uniform values that live through the whole kernel go to scalar registers (the vector file is full)
  const double y_sq = mfx_readlane_f64(y_sq_v, 0);
  // XC: v = the FP32 copy of x; everything is projected on the complement of v: with h = |v|^2, u = d.v / sqrt(h),
  // yx = y.v / sqrt(h) (y as its FP32 ranking copy, like the statistics)
  double rsh_v = 1.0, yx_v = 0.0;
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
    // (XC: both dictionaries here - the projected statistics and the amplification of the margin must be known
    // before the sweep - with u = d.v accumulated beside |d|^2 and d.y)
#pragma unroll 1
    for (int kd = XC ? 0 : 1; kd < 2; ++kd) {
    double ms_cur = 0.0, rm2_cur = 1.0;   // this dictionary's best single score / max |d|^2/|d'|^2 (no runtime-indexed arrays: scratch)
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
            } else {
              // projected statistics: |d'|^2 = |d|^2 - u^2, d'.y' = d.y - u yx.  An atom (nearly) inside span(x) -
              // |d'| < |d|/4 - would amplify the margin beyond use: the voxel goes to the FP64 kernel of the class
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
              // {d, x} with both weights non-negative is a feasible support: its score starts the threshold
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
  // margins.  XC: the screening error of the cross product is relative to |d1||d2|, the test works in units of
  // |d1'||d2'|: amplified by max |d|/|d'| of either dictionary (<= 4 each, see the statistics)
  double ramp_v = 1.0;
  if constexpr (XC) {
    const unsigned long long* rw = (const unsigned long long*)s_red + 24;
    ramp_v = sqrt(fmax(1.0, __longlong_as_double((long long)rw[0]))) * sqrt(fmax(1.0, __longlong_as_double((long long)rw[1])));
  }
  const double ramp = mfx_readlane_f64(ramp_v, 0);
  const double dc_eff = XC ? mfx_s_dc<KS>() * ramp : mfx_s_dc<KS>();     // bound on |c~ - c| in the units of the test
  // XC: the projected statistics cancel - z' |d'| = d.y - u yx - so the FP32 rounding of table, signal and column
  // (<= 3.6e-7 |d||y| in that difference) is no longer negligible when most of the signal is x: + 2e-6 ramp |y||y'| in a
  // score, + 2e-6 ramp |y| in e.  And never below the exact kernel's own tie tolerance, 1e-9 |y|^2: what it would
  // treat as a tie must reach its list.
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
#ifdef MFX_STAMPS_RND   // experiment: a starting threshold handed in by the tool (slot 15), to price the threshold's convergence
      if (a.stamps) {
        const double t0 = __longlong_as_double((long long)a.stamps[(size_t)blockIdx.x * 16 + 15]);
        if (t0 > best1 - mrg) s_thr[0] = mfx_nonneg_bits(t0);
      }
#endif
    }
  }
  __syncthreads();
  return K2sVoxel{y_sq, yx, ramp, dc_eff, mrg, etol};
}

// ---- everything after the sweep: D1's best single atom joins the candidates; hand-back decisions; XC: the voxel's short list;
// otherwise the exact stage (reference arithmetic and order on the ring entries that reach the final threshold, strict-'<'
// first hit, family expansion, run-time guard on the screening error) and the parameters (mf.py:420-450)
template <int KS, int NB, bool BR, bool XC, int PQF, int WG>
__device__ __forceinline__ void k2s_finish(const FitK2Args& a, const K2sLds<KS, NB, BR, XC, PQF>& L, const K2sVoxel& VX, const int vox,
                                           const int wave, const int lane, const double bs1, const int bn1) {
  constexpr int MP = KS * 16, NW = WG / 64;
  const int M = a.P.M, N = a.T.N;
  const int NP = (N + 31) & ~31;
  const double2* __restrict__ tab = a.T.tab;
  [[maybe_unused]] double* const smem = L.smem;
  K2S_UNPACK(L);
  const double y_sq = VX.y_sq, yx = VX.yx, mrg = VX.mrg;
  auto push = [&](double S, int i, int j) { k2s_push(L, a.scap, S, i, j); };
  auto tab_at = [&](int ro, int n) -> double2 { return *(const double2*)((const char*)tab + ((unsigned)(ro + n) << 4)); };
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
  [[maybe_unused]] int tid;

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
    // population audit of the [N, N, 1] form (see k2s_audit_hash): its accumulator holds d1.d2 - u1 u2 with the FP32 projections
    // u = d.x/|x| of the statistics pass in the spare row; wave 0 compares the parked value of the voxel's audited pair, in units
    // of |d1||d2|, with the FP64 sum over the FP64 table minus the product of the same two projections
    if (a.audit && wave == 0) {
      const float accv = ((const float*)(s_red + 30))[0];
      if (accv == accv) {
        const unsigned h = k2s_audit_hash(vox);
        const int ntl = NP >> 5;
        const int key = k2s_audit_key(h, (ntl % NW == 1 && ntl > 1) ? ntl - 1 : ntl, ntl);
        const int ga = (h >> 16) & 15, la = (h >> 20) & 63;
        const int i = (key >> 8) * 32 + (ga & 3) + 8 * (ga >> 2) + 4 * (la >> 5), j = (key & 0xff) * 32 + (la & 31);
        if (i < N && j < N) {   // (wave-uniform)
          double a11 = 0.0, a22 = 0.0, a12 = 0.0;
          for (int m = lane; m < M; m += 64) {
            const double d1 = elem(0, m, i), d2 = elem(1, m, j);
            a11 += d1 * d1; a22 += d2 * d2; a12 += d1 * d2;
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) { a11 += __shfl_xor(a11, o); a22 += __shfl_xor(a22, o); a12 += __shfl_xor(a12, o); }
          const double u1 = (double)s_uf[i], u2 = (double)s_uf[NP + j], p1 = (double)s_cs[i], p2 = (double)s_cs[NP + j];
          const double n12 = sqrt((p1 * p1 + u1 * u1) * (p2 * p2 + u2 * u2));     // |d1||d2| in the accumulator's units
          if (n12 > 0.0 && a11 > 0.0 && a22 > 0.0) {
            const double err = fabs((double)accv / n12 - (a12 / (sqrt(a11) * sqrt(a22)) - (u1 * u2) / n12));
            if (lane == 0) {
              atomicAdd(a.audit + 2, 1);
              atomicMax(a.audit + 1, (int)fmin(err * 1e11, 2.0e9));
              if (err > 0.25 * mfx_s_dc<KS>()) atomicAdd(a.audit, 1);
            }
          }
        }
      }
    }
    // the ring entries that reach the final threshold -> this voxel's short list (scores in projected units, + yx^2 in all)
    int* s_evl = (KS >= 8) ? (int*)((char*)smem + 2048) : s_evl4;
    if (tid == 0) s_cnt[2] = 0;
    __syncthreads();
    for (int cix = tid; cix < ncand; cix += WG)
      if (s_cand[cix].score >= thr_fin) s_evl[atomicAdd(&s_cnt[2], 1)] = cix;
    __syncthreads();
    const int neval = s_cnt[2];
    if (neval > a.xl_cap) {   // too many near-ties for the list: the FP64 kernel of the class decides
      if (tid == 0) {
        const int slot = atomicAdd(a.fb_count, 1);
        a.fb_list[slot] = vox;
        a.xl_cnt[blockIdx.x] = -1;
      }
      return;
    }
    Cand* dst = a.xl_cand + (size_t)blockIdx.x * a.xl_cap;
    for (int e = tid; e < neval; e += WG) {
      Cand c = s_cand[s_evl[e]];
      c.score += yx * yx;
      c.j &= ~MFX_S_BOUND;
      dst[e] = c;
    }
    // single atoms whose best support with ONE fascicle atom ({d} or {d, x}) could tie with the optimum (the exact
    // kernel's family rule needs them; it computes their statistics itself): approximate scores from the projected
    // statistics, |d|^2 = |d'|^2 + u^2, d.y = z' |d'| + u yx, everything in projected units (minus yx^2)
    __syncthreads();   // s_cnt[2] == neval has been read by everybody
    const double tcut = thr_fin - 2.0 * mrg;
    for (int q = tid; q < 2 * NP; q += WG) {
      const int k = q >= NP, n = q - k * NP;
      const double npr = (double)s_cs[q];
      if (n < N && npr > 0.0) {
        const double zp = (double)s_Zf[q], u = (double)s_uf[q];
        const double ayv = fma(u, yx, zp * npr);
        double s1 = (ayv > 0.0 ? ayv * ayv / fma(u, u, npr * npr) : 0.0) - yx * yx;
        // ({d, x}: whether d's weight is positive is decided by the exact kernel - a z' within the statistics' error of
        // zero, every atom of a voxel whose signal is all x, counts as positive here)
        if (zp > -sqrt(mrg)) s1 = fmax(s1, zp > 0.0 ? zp * zp : 0.0);
        // A pair whose projected two-atom solution has a non-positive weight is bounded by its better projected SINGLE
        // atom, z'^2 (x free, even negative) - not the score of any support, so unlike section 4.1 the best single atom
        // does not stand for such pairs: every atom whose z'^2 reaches the threshold takes ALL its pairs to the exact
        // kernel ("forced" family: -2 instead of -1 in the list entry)
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
    if (nall > a.xl_cap) {
      if (tid == 0) {
        const int slot = atomicAdd(a.fb_count, 1);
        a.fb_list[slot] = vox;
        a.xl_cnt[blockIdx.x] = -1;
      }
      return;
    }
    if (tid == 0) { a.xl_cnt[blockIdx.x] = nall; a.xl_mrg[blockIdx.x] = mrg; }
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
    // one WAVE sums the five Gram scalars of the pair (i, j) in the reference's order (mf_utils.py:307-325): the 64 lanes fetch
    // the 2 x M table entries side by side (all loads of the pair in flight at once), then five lanes run the five sequential
    // sums from LDS - lane 0: a11 = sum d1*d1, 1: a22 = sum d2*d2, 2: a12 = sum d1*d2, 3: y1 = sum y*d1, 4: y2 = sum y*d2
    auto wave_sums = [&](int i, int j, double& a11, double& a22, double& a12, double& y1, double& y2) {
#pragma unroll
      for (int mb = 0; mb < (MP + 63) / 64; ++mb) {
        const int m = mb * 64 + lane;
        if (m < M) {
          s_stage[m] = elem(0, m, i);
          s_stage[MP + m] = elem(1, m, j);
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
      a11 = mfx_readlane_f64(acc, 0); a22 = mfx_readlane_f64(acc, 1); a12 = mfx_readlane_f64(acc, 2);
      y1 = mfx_readlane_f64(acc, 3); y2 = mfx_readlane_f64(acc, 4);
      __builtin_amdgcn_wave_barrier();
    };
    if (neval <= 24) {
      // few candidates (the usual case): one WAVE per candidate.  A thread-per-candidate loop is bound by the
      // latency of its 2 x 200 dependent-address table loads (51 k cycles whatever the count); here the 64 lanes
      // fetch the rows side by side, then five lanes run the five sequential sums of mf_utils.py:307-325 from LDS.
      for (int e = wave; e < neval; e += NW) {
        const int cix = s_evl[e];
        const int i = s_cand[cix].i, jf = s_cand[cix].j, j = jf & ~MFX_S_BOUND;
        double a11, a22, a12, y1, y2;
        wave_sums(i, j, a11, a22, a12, y1, y2);
        if (e == 0) MFX_STAMP(15);
        double r, u0, u1;
        nnls2_exact(y_sq, a11, a12, a22, y1, y2, u0, u1, r);
        const long ix = (long)i * N + j;
        if (r < res || (r == res && ix < idx)) { res = r; idx = ix; w0 = u0; w1 = u1; }
        // run-time guard on the screening error (wave-uniform values)
        if (!(jf & MFX_S_BOUND) && u0 > 0.0 && u1 > 0.0 && fabs((y_sq - r) - s_cand[cix].score) > MFX_S_GUARD * mrg) s_cnt[1] = 1;
#ifdef MFX_STAMPS
        if (lane == 0) {
          ++dbg_eval;
          if (!(jf & MFX_S_BOUND) && u0 > 0.0 && u1 > 0.0) dbg_err = fmax(dbg_err, fabs((y_sq - r) - s_cand[cix].score) / y_sq);
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
        if (!(jf & MFX_S_BOUND) && u0 > 0.0 && u1 > 0.0) dbg_err = fmax(dbg_err, fabs((y_sq - r) - s_cand[cix].score) / y_sq);
#endif
      }
    }
    // population audit (see k2s_audit_hash): the next wave in turn compares the parked accumulator value of this voxel's
    // audited pair with the cosine from FP64 sums over the FP64 table
    if constexpr (!XC) {
      const float accv = ((const float*)(s_red + 30))[0];
      if (a.audit && accv == accv && wave == (neval <= 24 ? neval % NW : 0)) {
        const unsigned h = k2s_audit_hash(vox);
        const int ntl = NP >> 5;
        const int key = k2s_audit_key(h, (ntl % NW == 1 && ntl > 1) ? ntl - 1 : ntl, ntl);
        const int ga = (h >> 16) & 15, la = (h >> 20) & 63;
        const int i = (key >> 8) * 32 + (ga & 3) + 8 * (ga >> 2) + 4 * (la >> 5), j = (key & 0xff) * 32 + (la & 31);
        const double n12 = (double)s_cs[min(i, NP - 1)] * (double)s_cs[NP + min(j, NP - 1)];
        if (i < N && j < N && n12 > 0.0) {   // (wave-uniform)
          double a11, a22, a12, y1, y2;
          wave_sums(i, j, a11, a22, a12, y1, y2);
          const double err = fabs((double)accv / n12 - a12 / (sqrt(a11) * sqrt(a22)));
          if (lane == 0) {
            atomicAdd(a.audit + 2, 1);
            atomicMax(a.audit + 1, (int)fmin(err * 1e11, 2.0e9));
            if (err > 0.25 * mfx_s_dc<KS>()) atomicAdd(a.audit, 1);
          }
        }
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
