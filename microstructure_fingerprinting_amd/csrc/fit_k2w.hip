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
  [[maybe_unused]] const int M = a.P.M;
  const int N = a.T.N, ldn = a.T.ldn;
  const int NP = (N + 31) & ~31;  // atoms padded to a multiple of 32
  const int ntiles = NP >> 5;
  const int vox = a.vox_list ? a.vox_list[a.vox_base + blockIdx.x] : a.vox_base + (int)blockIdx.x;

  // ---- LDS layout, prologue (phases 0 and 1), ring append: shared with fit_k2s.hip (k2s_shared.h)
  const K2sLds<KS, NB, BR, XC, NW * TL * 64> L(smem, NP);
  K2S_UNPACK(L);
  const K2sVoxel VX = k2s_prologue<KS, NB, BR, XC, NW * TL * 64, WG>(a, L, vox, tid);
  const double y_sq = VX.y_sq, yx = VX.yx, ramp = VX.ramp, dc_eff = VX.dc_eff, mrg = VX.mrg, etol = VX.etol;
  (void)y_sq; (void)yx;
  const float2* __restrict__ tab32 = a.P.tab32s;
  auto tab32_at = [&](int ro, int n) -> float2 { return *(const float2*)((const char*)tab32 + ((unsigned)(ro + n) << 3)); };
  auto tab32x2_at = [&](int ro, int n) -> f32x4 { return *(const f32x4*)((const char*)tab32 + ((unsigned)(ro + n) << 3)); };
  auto push = [&](double S, int i, int j) { k2s_push(L, a.scap, S, i, j); };

  MFX_STAMP(2);
  double bs1 = 0.0;   // best single atom of D1 among the row tiles this wave has generated
  int bn1 = 0;
  constexpr int TPR = NW * TL;   // row tiles per round
  const int nrounds = (ntiles + TPR - 1) / TPR;
  // the voxel's audited pair: the key k2s_finish evaluates (its rule for the shared last tile: workgroup waves, not TPR)
  const int aud_key = a.audit ? k2s_audit_key(k2s_audit_hash(vox), (ntiles % NW == 1 && ntiles > 1) ? ntiles - 1 : ntiles, ntiles) : -1;
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
    const float DCF = XC ? (float)((mfx_s_dc<KS>() + 2e-6) * ramp) * (1.0f + 2e-7f) : (float)mfx_s_dc<KS>() + 2e-6f;
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
    // the voxel's audited pair (k2s_shared.h, as in fit_k2s.hip): when its tile's accumulators are complete, the raw value of
    // its register and lane is parked in LDS for the exact stage to compare with the FP64 cosine
    auto audit_park = [&](auto pc, int ct) {
      constexpr int P = decltype(pc)::value;
      if (aud_key < 0) return;
      mfx_static_for<0, TL>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if (rtv[t] && ((rts[t] << 8) | ct) == aud_key) {
          const unsigned h = k2s_audit_hash(vox);
          const int ga = (h >> 16) & 15;
          float v = accs[P][t][0];
#pragma unroll
          for (int g = 1; g < 16; ++g) v = (ga == g) ? accs[P][t][g] : v;
          if (lane == (int)((h >> 20) & 63)) ((float*)(s_red + 30))[0] = v;
        }
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
        audit_park(pc, c);
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
        audit_park(pc, c);
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
  k2s_finish<KS, NB, BR, XC, NW * TL * 64, WG>(a, L, VX, vox, wave, lane, bs1, bn1);
}
