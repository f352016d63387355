// fit_k2s.hip -- two-fascicle voxels, protocols of up to 256 measurements (exact-G or G-bracketed rows):
// split-FP16 MFMA SCREENING of all atom pairs followed by the exact FP64 evaluation of the short list.
// Same inputs, outputs and results as mfx_fit_k2_kernel (fit_k2.hip), which stays the path for longer
// protocols and the fallback for the rare voxel whose short list overflows.
//
// Why: the 2*N1*N2*M cross-Gram of solve_exhaustive_posweights_2 (mf_utils.py:307-325) only RANKS the
// pairs; the reference's answer is decided by the few pairs within rounding distance of the best one.
// So the Gram is computed here from operands split into two FP16 halves (mfx_split16),
//        a = hi + lo (+ r),  |r| <= 2^-21 |a|,      c~ = hi.hi + hi.lo + lo.hi
// on v_mfma_f32_32x32x16_f16 (3 instructions per 32x32x16 block, 32x the FP64 MFMA rate), with unit-norm
// columns so that c~ is the cosine of the pair up to |c~ - c| <= DC (MFX_S_DC below: 1.5e-5, a bound under one measured
// assumption about the matrix pipe's FP32 summation; 13x the largest error measured).
// For a pair whose optimum has two positive weights the score S = |y|^2 - residual obeys dS/dc = -2 w1 w2
// with w1 w2 <= |y|^2 / 2 (c >= 0), so |S(c~) - S(c)| <= DC |y|^2 =: m.  Every pair with S(c~) >= thr is
// appended to a ring in LDS, thr = (largest S(c~) seen) - 2m, which can only drop pairs that are not the
// optimum; nearly collinear pairs (1 - c~^2 < 1e-3), where S(c) is ill-conditioned, go through an
// interval upper bound instead.  The survivors (typically < 20 of 611 524) are evaluated by the same
// exact stage as in fit_k2.hip (reference arithmetic and order, strict-'<' first hit, family expansion).
// A ring entry that is overwritten while it could still matter raises a flag and the voxel is redone
// by the FP64 kernel (host side, mfx_api.hip), so the result never depends on the ring size.
//
// Structure: one 512-thread workgroup per voxel, 8 waves = 8 row tiles of 32 atoms of D1 per round; a
// wave keeps its tile (hi and lo, K = 16*KS) in 8*KS VGPRs; D2 is generated 32 atoms at a time into a
// double-buffered LDS image laid out in MFMA fragment order (ds_read_b128, conflict-free), waves 0-3 and
// 4-7 alternating between the MFMAs of a chunk and VALU work (pair screen + generation) half-step by
// half-step; a last round with one row tile left is shared by all waves.  Everything that only ranks reads
// an FP32 copy of the table, two adjacent atoms per 16-byte load (the vector-memory pipe, not the matrix
// pipe, is what this kernel saturates next to VALU issue: DESIGN.md 4.0/4.1).
//
// XC = true: the same sweep for the class with one fixed extra column ([N, N, 1]: two fascicles + CSF), the column projected
// out through a spare measurement row; it writes per-voxel short lists for fit_k2x.hip's exact stage instead of running
// one (see the comment at the kernel and DESIGN.md 4.3b).
#pragma once
#include "fit_k2.hip"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Bound on |c~ - c| (cosine units).  A statement under ONE measured assumption:
//   operands: every rotated entry is one FP32 fma of FP32 table values (<= 3 roundings): |da| <= 2.4e-7 |a|  -> 4.8e-7
//   split:    a = hi + lo + r, |r| <= 2^-21 |a|; dropped lo.lo + r.b + a.r <= 2^-19 |a b| per term           -> 1.9e-6
//   matrix pipe: ASSUMPTION - one v_mfma_f32_32x32x16_f16 returns its 17 addends' sum (16 exact products + accumulator)
//             within kappa 2^-24 (|c| + sum |a_k b_k|), kappa <= 5.1: the largest value tools/micro/mfma_sum_model.hip
//             finds over 15 adversarial operand families, 1.6e7 cases (profiles/r03_micro_mfma_sum_model.txt: 5.02;
//             2.0 on positive dictionary-like operands; the pipe does NOT round once - 12-46 % of the results differ
//             from the correctly rounded sum); 39 dependent instructions, every partial sum <= sum |a_i b_i| <= |a||b|
//             (Cauchy-Schwarz), the hi.lo / lo.hi addends 2^-10 of it:  39 x 5.1 x 2^-24 x 1.002                -> 1.19e-5
//   total 1.43e-5 <= MFX_S_DC.  (Measured: worst 1.14e-6 over 46 M adversarial pairs, tools/micro/split_mfma_error.hip;
//   and every voxel audits one pseudo-random pair of its own, k2s_shared.h.)
#ifndef MFX_S_DC
#define MFX_S_DC 1.5e-5
#endif
// The matrix-pipe term grows with the number of dependent instructions, 3 per k-step: longer protocols (KS = 16: up to 256
// measurements here, KS = 24 / 35: the wide kernel) get the margin the same statement gives for them, with the same 5 % of head
// room: 1.05 (2.38e-6 + 3 KS x 3.046e-7) = 1.79e-5, 2.55e-5, 3.61e-5.  (The audit of the wide kernel at 302 / 551 measurements:
// largest |c~ - c| 9.9e-7 / 1.19e-6.)
template <int KS>
__host__ __device__ constexpr double mfx_s_dc() { return KS <= 13 ? (double)MFX_S_DC : 1.05 * (2.38e-6 + 3.0 * KS * 3.046e-7); }
#define MFX_S_DENMIN 1e-3   // below this 1 - c~^2 the pair goes through the interval bound
#define MFX_S_BOUND 0x40000000   // ring entry flag (in .j): .score is an upper bound (interval bound, single atom), not S(c~)
#define MFX_S_GUARD 0.25    // run-time guard: an exactly evaluated pair whose screening score was off by more than this
                            // fraction of the margin DC |y|^2 sends the voxel to the FP64 kernel (and is counted)

// bit pattern of max(x, +0): non-negative doubles order like unsigned integers (LDS atomicMax)
__device__ __forceinline__ unsigned long long mfx_nonneg_bits(double x) {
  return (unsigned long long)__double_as_longlong(x > 0.0 ? x : 0.0);
}

// wave-wide maximum of NON-NEGATIVE values, rounded down to FP32, without LDS traffic: four DPP butterfly steps inside
// the rows of 16 lanes, then the four row values through scalar registers.  (wave_max's __shfl_xor is six dependent
// ds_bpermute round trips, ~800 of the ~2 600 cycles a register group cost in the FP64 pass while seven waves wait at
// the period's barrier; the result only feeds the threshold, where 6e-8 relative is 0.6 % of the margin.)
template <int CTRL>
__device__ __forceinline__ float mfx_dpp_max_f32(float v) {
  const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return fmaxf(v, __int_as_float(o));
}
__device__ __forceinline__ double mfx_wave_max_down(double x) {
  float v = (float)x * (1.0f - 1.2e-7f);   // <= x
  v = mfx_dpp_max_f32<0xb1>(v);    // quad_perm [1,0,3,2]
  v = mfx_dpp_max_f32<0x4e>(v);    // quad_perm [2,3,0,1]
  v = mfx_dpp_max_f32<0x141>(v);   // row_half_mirror
  v = mfx_dpp_max_f32<0x140>(v);   // row_mirror
  const int b = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16)),
              r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
  return (double)fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// 1/x for a positive, normal x to ~1e-14 relative: FP32 reciprocal + one Newton step (an IEEE FP64 division is ~10
// quarter-rate instructions; the ranking quantities of the FP64 pass carry a margin of 1e-5)
__device__ __forceinline__ double mfx_rcp_nr(double x) {
  const double r0 = (double)__builtin_amdgcn_rcpf((float)x);
  return fma(r0, fma(-x, r0, 1.0), r0);
}

// value of lane `l` (compile-time constant) in every lane: v_readlane_b32 x 2, no ds_bpermute index arithmetic
__device__ __forceinline__ double mfx_readlane_f64(double v, int l) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Neither operand is normalised: the FP32 screening table is pre-scaled on the host so that its values are <= 128,
// the accumulator holds cosine * |d1| * |d2| and the norms sit in the per-atom constants of the pair screen.  Both
// operands are split WITHOUT rescaling the low half:  f = hi + lo + r,  hi = f with its mantissa truncated to 10 bits
// (exact in FP16),  lo = fp16(f - hi),  |r| <= 2^-21 |f|.  With |f| <= 128 the low halves sit around 2^-10 |f|:
// normal FP16 numbers for all but negligible entries (an FP16 subnormal still resolves 6e-8, i.e. ~1e-10 of a column
// norm), so hi.hi, hi.lo and lo.hi can share ONE FP32 accumulator.
__device__ __forceinline__ void mfx_split16(float f, _Float16& hi, _Float16& lo) {
  // f must be ONE rounded FP32 value for both uses below (the compiler may otherwise fold the producing multiply
  // into a mixed-precision FMA for one use and not for the other: the halves then miss f by an FP16 ulp)
  asm("" : "+v"(f));   // not volatile: an opaque value, free to schedule
  const float h = __uint_as_float(__float_as_uint(f) & 0xffffe000u);
  hi = (_Float16)h;
  lo = (_Float16)(f - h);
}

#include "k2s_shared.h"

// BR: the protocol has G-bracketed rows (screening through the plan's virtual shells, exact stage as mfx_eval_br)
// NB: LDS images of D2 chunks: 3 (one workgroup barrier per chunk) where they fit beside the rest, else 2 (two barriers)
// XC: the voxel class has one fixed extra column x besides the two fascicles (sub-dictionaries [N, N, 1]: CSF).  Leaving
//     x unconstrained turns the problem into a two-atom problem in the orthogonal complement of x: atoms
//     d' = d - u x^ (u = d.x^, x^ = x/|x|), signal y' = y - (y.x^) x^, and  y.x^^2 + S2(d1', d2'; y')  bounds the score of
//     every support made of d1, d2 and x from above.  The projected cross product d1'.d2' = d1.d2 - u1 u2 comes out of
//     the SAME MFMAs: one spare padded measurement row carries -u1 in the A operand and u2 in the B image.  Statistics,
//     pair screen, FP64 criteria and ring then run unchanged on projected quantities (norms |d'|, z' = d'.y'/|d'|, the
//     margin DC amplified by max|d|/|d'| of the two dictionaries); the threshold only rises with pairs whose
//     relaxed solution keeps x's weight non-negative (then it IS a feasible score).  There is no exact stage here: the
//     ring entries that reach the final threshold go to a per-voxel list for fit_k2x.hip's exact stage (list mode),
//     which also owns every support with fewer than two fascicle atoms.
template <int KS, bool BR = false, int NB = 3, bool XC = false>
__global__ __launch_bounds__(512, 2) void mfx_fit_k2s_kernel(FitK2Args a) {
  constexpr int WG = 512, NW = 8;
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

  // ---- LDS layout, prologue (phases 0 and 1: signal, descriptors, column statistics, margins, starting threshold): k2s_shared.h
  const K2sLds<KS, NB, BR, XC, NW * 64> L(smem, NP);
  K2S_UNPACK(L);
  const K2sVoxel VX = k2s_prologue<KS, NB, BR, XC, NW * 64, WG>(a, L, vox, tid);
  const double y_sq = VX.y_sq, yx = VX.yx, ramp = VX.ramp, dc_eff = VX.dc_eff, mrg = VX.mrg, etol = VX.etol;
  (void)y_sq; (void)yx;
  const float2* __restrict__ tab32 = a.P.tab32s;   // FP32 copy of the table (== a.T.tab32 unless the plan has virtual shells): everything that only RANKS
  auto tab32_at = [&](int ro, int n) -> float2 { return *(const float2*)((const char*)tab32 + ((unsigned)(ro + n) << 3)); };
  // two adjacent atoms (n even) in one 16-byte load: {ylo_n, slope_n, ylo_n+1, slope_n+1}
  auto tab32x2_at = [&](int ro, int n) -> f32x4 { return *(const f32x4*)((const char*)tab32 + ((unsigned)(ro + n) << 3)); };
  auto push = [&](double S, int i, int j) { k2s_push(L, a.scap, S, i, j); };

  MFX_STAMP(2);
#ifdef MFX_STAMPS_RND
  if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 16] = __builtin_amdgcn_s_memtime();
#endif
  double bs1 = 0.0;   // best single atom of D1 among the row tiles this wave has generated
  int bn1 = 0;
  const int nrounds = (ntiles + NW - 1) / NW;
  // the voxel's audited pair (k2s_shared.h): its row tile and column chunk; the shared last row tile is not audited
#ifdef MFX_EXP_NOAUDIT   // timing experiment
  const int aud_key = -1;
#else
  const int aud_key = a.audit ? k2s_audit_key(k2s_audit_hash(vox), (ntiles % NW == 1 && ntiles > 1) ? ntiles - 1 : ntiles, ntiles) : -1;
#endif
  for (int round = 0; round < nrounds; ++round) {
    // A last round with ONE row tile left (N = 782: 25 = 3*8 + 1) is shared by all waves: each keeps the same
    // A tile and takes every 8th column tile, generating its B operand straight into registers (no LDS image,
    // no workgroup barrier), instead of 7 waves idling through a full D2 sweep.
    const bool tail = (ntiles - round * NW == 1) && (ntiles > 1);
    const int rt = tail ? round * NW : round * NW + wave;
    const bool rt_valid = rt < ntiles;  // wave-uniform
    const int rtc = rt_valid ? rt : 0;
    // A operand: this wave's 32 atoms of D1, all KS k-steps, split in registers - UN-normalised like D2 (the table
    // is pre-scaled to values <= 128), so that the column statistics I1 = 1/|d1|, Z1 = d1.y/|d1| fall out of the
    // same read of the table (ranking only, as in phase 1; a lane sums its half of the rows, the halves meet
    // through one cross-lane exchange): the L2 -> L1 fill rate bounds these passes, not the arithmetic.
    h8 afh[KS], afl[KS];
    {
      const int n = rtc * 32 + lr;
      const int nn = min(n, ldn - 1);
      double a2 = 0.0, ay = 0.0;
      // XC: the LAST padded measurement row (MP - 1 > M - 1: the launcher sees to that; the table's padded rows are
      // zero) carries -u1 in the A operand and u2 in the B images - a compile-time position: a test against M per element
      // is loop-invariant, gets hoisted out of the rounds and costs ~100 spilled registers
      float um1 = 0.0f;
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
        asm volatile("" : "+v"(vh), "+v"(vl));   // materialise here: the conversions must not sink below all loads
        afh[ks] = vh; afl[ks] = vl;
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (!XC) {   // (XC: the statistics of both dictionaries come from phase 1, and one-atom supports are not this kernel's)
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
      // best single atom of D1 so far (first index on ties)
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
      if (sb > bs1 || (sb == bs1 && sb > 0.0 && nb < bn1)) { bs1 = sb; bn1 = nb; }
      // a pair matters only if it beats every single atom
      if (lane == 0 && sb - mrg > 0.0) atomicMax(&s_thr[0], mfx_nonneg_bits(sb - mrg));
      }
    }

    // pair screen of one 32x32 accumulator tile against column tile ct (used by the LDS sweep and by the tail round)
    //
    // Fast pass in FP32, 3 VALU per pair.  With unit atoms at angle phi (c = cos phi), the projection of y on
    // their plane has squared length S and makes the angles alpha1, alpha2 with them (z_i = sqrt(S) cos alpha_i);
    // both weights are positive exactly when it lies between the atoms: phi = alpha1 + alpha2.  For any T <= S,
    // theta_i = acos(min(1, max(z_i, 0) / sqrt(T))) <= alpha_i, hence
    //     two positive weights and S >= T   ==>   c <= cos(theta1 + theta2) = P1 P2 - Q1 Q2,
    // P_i = cos theta_i, Q_i = sin theta_i (the other branch of S(c) >= T, c >= cos(theta1 - theta2), has a
    // non-positive weight).  The test is therefore  c~ - DC <= P1 P2 - Q1 Q2  with T = the running threshold;
    // a STALE (lower) threshold, P rounded up and Q rounded down only let more pairs through.  In accumulator
    // units (acc = c |d1| |d2|, neither operand is normalised) the margin is folded into the constants,
    //     (P1 + D)(P2 + D) - (1 - D)^2 Q1 Q2  >=  P1 P2 - Q1 Q2 + D      for all P, Q = sqrt(1 - P^2) in [0, 1]
    // (the difference is D (P1 + P2 + (2 - D) Q1 Q2 - 1) + D^2 and P1 + P2 + 1.99 Q1 Q2 >= 1 on the unit square),
    // at the price of a margin up to ~3 D instead of D:
    //     t = ((P1 + D) n1)((P2 + D) n2) - ((1 - D) Q1 n1)((1 - D) Q2 n2) - acc,      pass when max t >= 0:
    // two FMAs and a max per pair, from two constants per row and two per column.  Rows and columns beyond N get
    // constants that do not pass (but for the corner cases noted below).  Only a register group with a passing
    // pair runs the FP64 criteria below (one out-of-line copy).
    double thr = 0.0, thr_rows = -1.0;
#ifdef MFX_STAMPS_RND
    int dbg_flagged = 0, dbg_groups = 0;   // accumulator tiles / register groups of this wave that reached the FP64 criteria
#endif
    const float DCF = XC ? (float)((mfx_s_dc<KS>() + 2e-6) * ramp) * (1.0f + 2e-7f)
                         : (float)mfx_s_dc<KS>() + 2e-6f;   // + the FP32 evaluation error of t (< 1e-6 in cosine units)
    auto pq_of = [&](float z, float rth, float& P, float& Q) {
      P = fminf(1.0f, fmaxf(z, 0.0f) * rth);
      Q = __builtin_amdgcn_sqrtf(fmaxf(0.0f, fmaf(-P, P, 1.0f) - 1.2e-7f)) * (1.0f - 3e-7f);
    };
    // scan_pre only ISSUES the LDS reads of a tile's column (threshold, column statistics): a screening wave has
    // no partner to cover an LDS round trip, so they fly behind the generation work; scan_main does the rest
    unsigned long long sc_thr = 0ull;
    float sc_z2 = 0.0f, sc_s = 0.0f;
    auto scan_pre = [&](int ct) {
      const int j = ct * 32 + lr;
      sc_thr = s_thr[0];
      sc_z2 = s_Zf[NP + j];
      sc_s = s_cs[NP + j];
    };
#ifdef MFX_STAMPS_SCAN   // diagnostic builds: wave 0's first pair screen of a voxel in detail (tools/dev_stamps_scan.py)
    int dbg_calls = 0;
#define MFX_SCAN_T(k) do { if (a.stamps && round == 0 && dbg_calls == 1 && wave == 0 && lane == 0) a.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MFX_SCAN_T(k) do { } while (0)
#endif
    auto scan_main = [&](const f32x16& acc, int ct) {
#ifdef MFX_STAMPS_SCAN
      ++dbg_calls;
#endif
      MFX_SCAN_T(0);
      {
        if (((rt << 8) | ct) == aud_key) {   // once per voxel, one wave: park the raw accumulator value of the audited pair (k2s_shared.h)
          const unsigned h = k2s_audit_hash(vox);
          const int ga = (h >> 16) & 15;
          float v = acc[0];
#pragma unroll
          for (int g = 1; g < 16; ++g) v = (ga == g) ? acc[g] : v;
          if (lane == (int)((h >> 20) & 63)) ((float*)(s_red + 30))[0] = v;
        }
      }
      const int j = ct * 32 + lr;
      // row i = rt*32 + (g&3) + 8(g>>2) + 4 lh, column j = ct*32 + lr
      thr = fmax(thr, __longlong_as_double((long long)sc_thr));
      // 1/sqrt(T), T = thr rounded down, the reciprocal root rounded up (v_rsq_f32: 1 ulp)
      const float rth = __builtin_amdgcn_rsqf(fmaxf((float)thr * (1.0f - 2e-7f), 1e-30f)) * (1.0f + 4e-7f);
      float* pqw = s_pq + wave * 64;
      if (thr > thr_rows) {   // wave-uniform: the threshold rose since this wave's row constants were made
        thr_rows = thr;
        if (lane < 32) {
          const float z1 = s_Zf[rtc * 32 + lane], n1 = s_cs[rtc * 32 + lane];
          float P, Q;
          pq_of(z1, rth, P, Q);
          // beyond N: (0, 1e18) for rows, (-1e18, 1e18) for columns: t < 0 but for a padded row against a column
          // with Q2 = 0 (its atom alone reaches the threshold), which the FP64 criteria reject
          const bool ok = n1 > 0.0f;
          pqw[lane] = ok ? (P + DCF) * n1 : 0.0f;
          pqw[32 + lane] = ok ? Q * ((1.0f - DCF) * n1) : 1e18f;
        }
      }
      const float z2f = sc_z2, n2 = sc_s;
      float P2, Q2;
      pq_of(z2f, rth, P2, Q2);
      const bool colok = n2 > 0.0f;
      const float p2 = colok ? (P2 + DCF) * n2 : -1e18f, q2 = colok ? Q2 * ((1.0f - DCF) * n2) : 1e18f;
      float mm[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 p1q = *(const f32x4*)(pqw + 8 * q + 4 * lh);        // rows (g&3) + 8q + 4 lh, g = 4q..4q+3
        const f32x4 q1q = *(const f32x4*)(pqw + 32 + 8 * q + 4 * lh);
        // plain (not packed) FP32: beside the other wave's MFMAs a v_pk_*_f32 costs several plain ones
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = fmaf(-q1q[u], q2, fmaf(p1q[u], p2, -acc[4 * q + u]));
        mm[q] = fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3]));
      }
      MFX_SCAN_T(1);
      if (__any(fmaxf(fmaxf(mm[0], mm[1]), fmaxf(mm[2], mm[3])) >= 0.0f)) {
        MFX_SCAN_T(2);
#ifdef MFX_STAMPS_RND
        ++dbg_flagged;
#endif
        // ---- exact FP64 pass over the flagged register groups (rare once thr is close to the optimum)
        // the column statistics are ranking-grade anyway (FP32 table): their FP32 copies serve here too
        // (6e-8 relative: ~4e-7 |y|^2 in a score, against the margin of 1e-5 |y|^2)
        const double z2 = (double)s_Zf[NP + j], n2d = (double)s_cs[NP + j];
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
          if (!__any(mm[q] >= 0.0f)) continue;
          // the quad's four register groups one by one: the FP32 test again (mm[q] is their maximum), so that only
          // groups with a passing pair pay the ~40 FP64 instructions below - the seven other waves of the workgroup
          // wait at the period's barrier for a wave that is in here
          const f32x4 p1q = *(const f32x4*)(pqw + 8 * q + 4 * lh);
          const f32x4 q1q = *(const f32x4*)(pqw + 32 + 8 * q + 4 * lh);
#pragma unroll 1
          for (int gg = 0; gg < 4; ++gg) {
            const int g = 4 * q + gg;
            if (!__any(fmaf(-q1q[gg], q2, fmaf(p1q[gg], p2, -acc[g])) >= 0.0f)) continue;
#ifdef MFX_STAMPS_RND
            ++dbg_groups;
#endif
            const int i = rt * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
            const double n12 = (double)s_cs[i] * n2d;
            const double c = n12 > 0.0 ? (double)acc[g] * mfx_rcp_nr(n12) : 0.0;
            const double z1 = (double)s_Zf[i];
            const double e1 = fma(-c, z2, z1);
            const double e2 = fma(-c, z1, z2);
            const double den = fma(-c, c, 1.0);
            const double num = fma(z2, e2, z1 * e1);
            const bool pos = (e1 > -etol) & (e2 > -etol);      // false for padded atoms (z = -inf)
            // (dS/dc = -2 w1 w2 <= |y|^2 needs c >= 0; pairs at an obtuse angle - no physical dictionary has them -
            // go through the interval bound like the ill-conditioned ones)
            const bool wellc = (den >= MFX_S_DENMIN) & (c >= 0.0);
            const bool hit = pos & wellc & (fma(-thr, den, num) >= 0.0);
            const bool near = pos & !wellc;
            if (!__any(hit | near)) continue;
            double S = -1.0;
            if (hit) {
              S = num * mfx_rcp_nr(den);
            } else if (near) {
              // ill-conditioned pair: interval upper bound of S over |c - c~| <= DC;
              // S = z2^2 + e1^2/den = z1^2 + e2^2/den for two positive weights
              const double dlo = den - 2.0 * dc_eff - dc_eff * dc_eff;
              const double u1 = fabs(e1) + etol, u2 = fabs(e2) + etol;
              S = (dlo > 0.0 && c > -0.5) ? fmin(fma(z2, z2, u1 * u1 / dlo), fma(z1, z1, u2 * u2 / dlo)) + mrg : 1e300;
            }
            // raise the threshold with the best SCORE of this wave instruction first (an interval bound is not
            // a score and never raises it), then append only what still reaches it: no burst of stale entries
            bool feas = hit;
            double q1x = 0.0, q2x = 0.0;   // (XC) u / |d'| of the two atoms
            if constexpr (XC) {
              // a relaxed score may raise the threshold only if it is the score of a feasible support: x's weight
              // w_x = yx - w1 u1 - w2 u2, w_i = e_i / (den |d_i'|), clearly non-negative (e carries an error <= etol,
              // u/|d'| <= 4)
              q1x = (double)s_uf[i] * mfx_rcp_nr(fmax((double)s_cs[i], 1e-300));
              q2x = (double)s_uf[NP + j] * mfx_rcp_nr(fmax(n2d, 1e-300));
              feas = hit && (fma(-e2, q2x, fma(-e1, q1x, yx * den)) >= 8.0 * etol);
            }
            double sraise = feas ? S : 0.0;
            double slist = S;   // what the pair is listed with: an upper bound of the best score of its supports with BOTH atoms
            if constexpr (XC) {
              // x would get a negative weight: the pair's PLAIN two-atom score (x left out) is feasible and raises the
              // threshold instead - without it a voxel with no x signal never leaves the single-atom threshold, floods
              // the ring and ends up on the FP64 kernel.  Unprojected statistics from the projected ones:
              // |d|^2 = |d'|^2 + u^2, d.y = z' |d'| + u yx, d1.d2 = d1'.d2' + u1 u2; scores here are minus yx^2.
              if (__any(hit && !feas)) {
                const double n1p = (double)s_cs[i], u1 = (double)s_uf[i], u2 = (double)s_uf[NP + j];
                const double m1 = fma(u1, u1, n1p * n1p), m2 = fma(u2, u2, n2d * n2d);
                const double i1 = mfx_rcp_nr(fmax(sqrt(m1), 1e-300)), i2 = mfx_rcp_nr(fmax(sqrt(m2), 1e-300));
                const double w1 = fma(u1, yx, z1 * n1p) * i1, w2 = fma(u2, yx, z2 * n2d) * i2;   // d.y / |d|
                const double c0 = fma(u1, u2, (double)acc[g]) * i1 * i2;
                const double f1 = fma(-c0, w2, w1), f2 = fma(-c0, w1, w2), den0 = fma(-c0, c0, 1.0);
                const bool ok0 = hit && !feas && (f1 > etol) && (f2 > etol) && (den0 >= MFX_S_DENMIN) && (c0 >= 0.0) && (m1 > 0.0) && (m2 > 0.0);
                if (ok0) sraise = fma(w2, f2, w1 * f1) * mfx_rcp_nr(den0) - yx * yx;
                // A pair whose relaxed optimum CLEARLY gives x a negative weight (both atoms clearly active in it) has x
                // inactive in its NNLS optimum: that optimum is the plain two-atom problem's.  Listed with the plain score
                // then, or - one atom clearly inactive there - not at all (supports with one atom are the list kernel's
                // own).  Without this a flagged voxel WITHOUT x signal lists every pair its loose relaxed bound lets
                // through: 17 % of such voxels overflowed their list and went to the FP64 kernel.
                const bool xneg = hit && (e1 > etol) && (e2 > etol) && (fma(-e2, q2x, fma(-e1, q1x, yx * den)) <= -8.0 * etol) &&
                                  (den0 >= MFX_S_DENMIN) && (c0 >= 0.0) && (m1 > 0.0) && (m2 > 0.0);
                if (xneg) {
                  if (ok0) slist = sraise + mrg;
                  else if (f1 < -etol || f2 < -etol) slist = -1.0;
                }
              }
            }
            const double smax = mfx_wave_max_down(fmax(sraise, 0.0));
            if (smax - 2.0 * mrg > thr) {
              thr = smax - 2.0 * mrg;
              if (lane == 0) atomicMax(&s_thr[0], mfx_nonneg_bits(thr));
            }
            if ((hit | near) && slist >= thr) push(slist, i, hit ? j : (j | MFX_S_BOUND));
#ifdef MFX_STAMPS_SCAN
            if (a.stamps && round == 0 && dbg_calls == 1 && wave == 0 && lane == 0) {
              unsigned long long* st = a.stamps + (size_t)blockIdx.x * 16;
              if (st[15] < 8) st[4 + st[15]] = __builtin_amdgcn_s_memtime();   // end of the first 8 evaluated groups
              ++st[15];
            }
#endif
          }
        }
        MFX_SCAN_T(3);
      }
    };
    auto scan_tile = [&](const f32x16& acc, int ct) { scan_pre(ct); scan_main(acc, ct); };

#ifdef MFX_EXP_NOTAIL   // timing experiment (wrong results): what the shared last row tile costs
    if (tail) continue;
#endif
    if (tail) {
#ifdef MFX_STAMPS_RND
      if (a.stamps && tid == 0 && round < 4) a.stamps[(size_t)blockIdx.x * 16 + 2 * round + 1] = __builtin_amdgcn_s_memtime();
#endif
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
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[ks], bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[ks], bl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afl[ks], bh, acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
        scan_tile(acc, ct);
      }
      __syncthreads();   // all appends of the round are in the ring
#ifdef MFX_STAMPS_RND
      if (a.stamps && tid == 0 && round < 4) a.stamps[(size_t)blockIdx.x * 16 + 2 * round + 2] = __builtin_amdgcn_s_memtime();
#endif
      continue;
    }

    // ---- LDS sweep, ping-pong between the two wave groups (waves 0-3 | 4-7: one wave of each per SIMD): while one
    // group runs the 3*KS MFMAs of a column chunk the other one does VALU work (pair screen of its previous
    // accumulator tile + its half of the generation of a coming chunk), so the matrix pipe and the vector ALUs of
    // a SIMD are busy at the same time.  Generation item = (pair of adjacent atoms, the 8 rows of one MFMA
    // fragment); a group owns KS of the 2*KS fragment row blocks of a chunk, one item per thread: gen_load issues
    // its table loads, gen_store converts and writes the FP16 hi/lo fragments.  The two schedules are below.
    const int grp = wave >> 2, tg = tid & 255;
    static_assert(KS <= 16, "one generation item per thread");
    f32x4 gd[8];   // item = (pair of adjacent atoms, the 8 rows of one fragment): 16-byte loads, see phase 1
    const int gq = KS * grp + (tg >> 4);      // fragment row block (k-step, half) = gq; rows 8 gq .. 8 gq + 7
    const bool gact = (tg >> 4) < KS;
    auto gen_load = [&](int ch) {
      const int nn = min(ch * 32 + 2 * (tg & 15), ldn - 2);
      const int q = gact ? gq : KS * grp;     // idle threads repeat a valid address, gen_store skips them
#pragma unroll
      for (int e = 0; e < 8; ++e) gd[e] = tab32x2_at(s_rs[MP + 8 * q + e], nn);
    };
    // the item's interpolation offsets are the same for every chunk of the round: registers, not LDS
    // (not for KS = 16: its A tile takes 128 registers and every further one spills)
#ifndef MFX_XC_GT
#define MFX_XC_GT 0   // (the CSF form needs the registers: with the offsets in registers it spilled 26-29, 11 scratch accesses per period)
#endif
    constexpr bool GT = KS <= 13 && (!XC || MFX_XC_GT);
    float g_t[GT ? 8 : 1];
    if constexpr (GT) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g_t[e] = s_t0f[MP + 8 * (gact ? gq : KS * grp) + e];
    }
    auto gen_store = [&](int ch, int buf) {
      if (gact) {
        const int c0 = 2 * (tg & 15);
        h8 hi0, lo0, hi1, lo1;
        float gu0 = 0.0f, gu1 = 0.0f;   // XC: u2 of the item's two atoms, for the spare row M
        if constexpr (XC) { gu0 = s_uf[NP + ch * 32 + c0]; gu1 = s_uf[NP + ch * 32 + c0 + 1]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = GT ? g_t[GT ? e : 0] : s_t0f[MP + 8 * gq + e];
          _Float16 x, y;
          float v0 = fmaf(gd[e][1], t, gd[e][0]), v1 = fmaf(gd[e][3], t, gd[e][2]);
          if constexpr (XC) { if (e == 7) { const bool spare = (gq == 2 * KS - 1); v0 = spare ? gu0 : v0; v1 = spare ? gu1 : v1; } }
          mfx_split16(v0, x, y);
          hi0[e] = x; lo0[e] = y;
          mfx_split16(v1, x, y);
          hi1[e] = x; lo1[e] = y;
        }
        // fragments of atoms c0, c0+1 are adjacent: 32 contiguous bytes per lane, conflict-free
        const int off = (gq * 32 + c0) << 3;
        _Float16* dh = sBh + buf * KS * 512 + off;
        _Float16* dl = sBl + buf * KS * 512 + off;
        *(h8*)dh = hi0; *(h8*)(dh + 8) = hi1;
        *(h8*)dl = lo0; *(h8*)(dl + 8) = lo1;
      }
    };
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
    // the 3*KS MFMAs of this wave's row tile against the chunk image in buffer buf
    auto mfma_chunk = [&](int buf) {
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
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
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[ks], bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[ks], bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afl[ks], bh, acc, 0, 0, 0);
        bh = bhn; bl = bln;
        __builtin_amdgcn_sched_barrier(0);
      });
    };

    if (round == 0) MFX_STAMP(3);
#ifdef MFX_STAMPS_RND   // diagnostic builds: start and end of every round's sweep (tools/dev_stamps_rnd.py)
    if (a.stamps && tid == 0 && round < 4) a.stamps[(size_t)blockIdx.x * 16 + 2 * round + 1] = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (NB == 3) {
      // ---- one barrier per chunk.  In period c both groups multiply chunk c - group 0 first, group 1 second - and
      // do their VALU work in the other phase: group 0 then screens chunk c and generates its half of chunk c+2,
      // group 1 first screens chunk c-1 and generates its half of chunk c+1.  Chunk c lives in buffer c mod 3:
      // during period c nobody writes it, group 1 writes buffer (c+1) mod 3 (last read in period c-2) and group 0
      // buffer (c+2) mod 3 (last read in period c-1), so the two phases of a period need no barrier between them:
      // a wave moves on to its second phase as soon as its first is done instead of waiting for the slower group
      // (the screening wave has nobody to cover its LDS round trips and in-order table loads: ~2 250 cycles against
      // 1 330 for the MFMA chain when the half-steps were coupled by a barrier).
      // Table loads of a group's next item go at the end of its VALU phase and are consumed one period later.
      gen_load(0);
      gen_store(0, 0);
      if (grp == 0) {
        if (ntiles > 1) { gen_load(1); gen_store(1, 1); }
        if (ntiles > 2) gen_load(2);
      } else if (ntiles > 1) {
        gen_load(1);
      }
      __syncthreads();
      if (round == 0) MFX_STAMP(4);
      thr = __longlong_as_double((long long)s_thr[0]);
      int b0 = 0;   // buffer of chunk c
#ifdef MFX_EXP_NOGEN   // timing experiment (wrong results): rounds after the first find their chunk images ready-made
      const bool exp_gen = round == 0;
#else
      constexpr bool exp_gen = true;
#endif
      for (int c = 0; c <= ntiles; ++c) {
#ifdef MFX_STAMPS_HS   // diagnostic builds: start of 16 periods as seen by wave 0 (tools/dev_stamps_hs.py; default: periods 10..25 of round 1)
#ifndef MFX_HS_ROUND
#define MFX_HS_ROUND 1
#define MFX_HS_C0 10
#endif
        if (a.stamps && round == MFX_HS_ROUND && c >= MFX_HS_C0 && c < MFX_HS_C0 + 16 && tid == 0) a.stamps[(size_t)blockIdx.x * 16 + c - MFX_HS_C0] = __builtin_amdgcn_s_memtime();
#endif
        const int b1 = b0 == 2 ? 0 : b0 + 1, b2 = b1 == 2 ? 0 : b1 + 1;   // buffers of chunks c+1, c+2
        if (grp == 0) {
          if (c < ntiles) {
            if (rt_valid) { mfma_chunk(b0); scan_pre(c); }
            if (exp_gen && c + 2 < ntiles) gen_store(c + 2, b2);
            if (rt_valid) scan_main(acc, c);
            if (exp_gen && c + 3 < ntiles) gen_load(c + 3);
          }
        } else {
          if (c >= 1 && rt_valid) scan_pre(c - 1);
          if (exp_gen && c + 1 < ntiles) gen_store(c + 1, b1);
          if (c >= 1 && rt_valid) scan_main(acc, c - 1);
          if (exp_gen && c + 2 < ntiles) gen_load(c + 2);
          if (c < ntiles && rt_valid) mfma_chunk(b0);
        }
        // LDS-only workgroup barrier: __syncthreads() would also wait for the table loads just issued (vmcnt)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        b0 = b1;
      }
    } else {
      // ---- two chunk images: a barrier after every half-step.  Group g multiplies chunk c in half-step 2c+g and
      // screens it in half-step 2c+g+1; chunk c lives in buffer c&1, read in half-steps 2c, 2c+1 and written in
      // 2c-2 (group 1's half) and 2c-1 (group 0's half), i.e. while buffer (c-1)&1 is being read.  gen_load issues
      // the table loads at the start of the group's MFMA half-step, gen_store converts and writes the FP16 hi/lo
      // fragments in its next VALU half-step (the loads fly behind the MFMAs).
      gen_load(0);
      gen_store(0, 0);
      if (grp == 1 && ntiles > 1) { gen_load(1); gen_store(1, 1); }
      __syncthreads();
      if (round == 0) MFX_STAMP(4);
      thr = __longlong_as_double((long long)s_thr[0]);
      for (int hs = 0; hs <= 2 * ntiles; ++hs) {
        if ((hs & 1) == grp) {
          const int c = (hs - grp) >> 1;
          if (c < ntiles) {
            if (c + 1 + grp < ntiles) gen_load(c + 1 + grp);   // consumed in this group's next half-step
            if (rt_valid) mfma_chunk(c & 1);
          }
        } else {
          const int c = (hs - 1 - grp) >> 1;
          if (c >= 0 && c < ntiles) {
            if (c + 1 + grp < ntiles) gen_store(c + 1 + grp, (c + 1 + grp) & 1);
            if (rt_valid) scan_tile(acc, c);
          }
        }
        __syncthreads();
      }
    }
    if (round == 0) MFX_STAMP(5);
#ifdef MFX_STAMPS_RND
    if (a.stamps && tid == 0 && round < 4) a.stamps[(size_t)blockIdx.x * 16 + 2 * round + 2] = __builtin_amdgcn_s_memtime();
    if (a.stamps && lane == 0 && round < 3) atomicAdd(&a.stamps[(size_t)blockIdx.x * 16 + 9 + round], (unsigned long long)dbg_flagged + ((unsigned long long)dbg_groups << 32));
#endif
  }
  k2s_finish<KS, NB, BR, XC, NW * 64, WG>(a, L, VX, vox, wave, lane, bs1, bn1);
}
