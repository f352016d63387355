// fit_k2x.hip -- fused per-voxel kernel for two fascicles PLUS voxel-independent compartments (and, with LIST = true,
// the exact stage behind the screening kernel's short lists for the [N, N, 1] class: fit_k2s.hip XC = true):
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
//   * in front of the per-tuple scoring (~30 FP64 instructions per tuple) sits a FILTER of 5 instructions per tuple: with
//     the extra columns R_t = {f, x_t} of tuple t left UNCONSTRAINED the problem is a two-atom problem in the orthogonal
//     complement of R_t (projected atoms d' = d - proj_R d, projected signal y'), whose score + the score q0_t of R_t alone
//     is an upper bound of the tuple's NNLS score.  Two positive atom weights with that bound >= T needs
//     cos(d1', d2') <= cos(theta1 + theta2), theta_i = acos(min(1, z_i' / sqrt(T - q0_t))) - the angle-sum test of
//     fit_k2s.hip, here with the cross term updated per tuple in FP64 (a12' = a12 - u_f1 u_f2 - u_t1 u_t2: no cancellation
//     error) and the test itself in FP32 from two constants per (atom, tuple); an atom whose own projected score reaches
//     T - q0_t passes with every partner.  T = the best NNLS score found so far (workgroup-wide, LDS atomicMax) minus the
//     tie tolerance: only tuples that pass are scored, typically < 1 %;
//   * a short list that still overflows triggers the exhaustive exact pass over all N*N*ntup tuples (slow, exact by
//     construction, counted in the launch's overflow counter).
#pragma once
#include "fit_k2.hip"     // mfx_static_for, MFX_STAMP
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
  int xx_in_lds; // the extra columns are staged in LDS (when the 160 KB allow it)
  unsigned long long* stamps;  // diagnostic builds only: [gridDim.x][16] s_memtime stamps (null otherwise)
  int* ovf_count;  // [4] launch counters: [0] voxels that took the exhaustive pass, [2] short-listed pairs, [3] family items
  // list mode (LIST = true: the pair scan is replaced by the short lists of the screening kernel, fit_k2s.hip XC = true)
  const Cand* xl_cand;     // [gridDim.x][xl_cap] short-listed pairs; .score bounds the pair's best score from above (up to xl_mrg)
  const int* xl_cnt;       // [gridDim.x] their number, or -1: the voxel is not this launch's
  const double* xl_mrg;    // [gridDim.x] the screening margin of the voxel (score units)
  int xl_cap;
  int* fb_count;           // hand-back list of the pipeline: a pair that beats its bound sends the voxel to the plain kernel
  int* fb_list;
  const int* list_count;   // null, or device word: blocks beyond it exit (plain kernel over a device-side voxel list)
};

#define MFX_XFAM 64   // family items (see below) per voxel before the exhaustive pass takes over
struct FamX {
  int type;   // 1: (i, all j, t)   2: (all i, j, t)   3: (0, 0, t): no fascicle atom active
  int a, t;
};

struct ProjC {   // filter constants of one (atom, extra tuple): see the top of the file
  double u;      // d . x_t' / |x_t'| with x_t' = x_t minus its projection on the fixed column
  float pn, qn;  // (P + D) |d'|, (1 - D) Q |d'|
};
#define MFX_XD 2e-6f   // margin folded into the filter constants: covers the FP32 evaluation of the test
typedef unsigned int mfx_u32x4 __attribute__((ext_vector_type(4)));
// one 16-byte LDS read per (atom, tuple) record (the records sit at 16-byte aligned LDS addresses: see the carve-up)
__device__ __forceinline__ ProjC mfx_ldc(const ProjC* p) {
  const mfx_u32x4 r = *(const mfx_u32x4*)__builtin_assume_aligned(p, 16);
  ProjC c;
  c.u = __longlong_as_double((long long)(((unsigned long long)r[1] << 32) | r[0]));
  c.pn = __uint_as_float(r[2]);
  c.qn = __uint_as_float(r[3]);
  return c;
}
#define MFX_XQ 64   // entries of a wave's queue of passing tuples: one scoring pass with every lane busy
struct QItem {   // a tuple that passed the filter: the owner lane writes (v = a1.a2, code), the scoring lane answers v = score
  double v;
  int code, pad;   // row in the wave's tile | column in the chunk << 4 | tuple << 8
};
struct ProjB {   // threshold-independent part of the filter constants of one (atom, extra tuple), computed once per voxel
  double u;      // as ProjC::u (entry ntup of an atom: its d . f / |f|)
  float np, zp;  // |d'| (0: d' vanishes, the atom passes with every partner) and d'.y' / |d'|
};

// d.f of a row / column with the atom's "passes with every partner" flags of the ntup tuples in its 16 lowest mantissa bits
// (flag of tuple t in bit ntup - 1 - t: the layout of the filter's shift registers)
__device__ __forceinline__ double mfx_pack_alw(double u, unsigned flags_by_tuple, int ntup) {
  const unsigned rev = __builtin_bitreverse32(flags_by_tuple) >> (32 - ntup);
  return __longlong_as_double((__double_as_longlong(u) & ~0xffffll) | (long long)rev);
}

struct CandX {   // short-list entry: a pair of atoms and its ranking score
  double score;
  int i, j;
};

// NW waves per workgroup and NBUF LDS chunk buffers: (8, 2) for M <= 200, (4, 1) for long protocols
// (one wave per SIMD owns the 512-register file; see fit_k2.hip).
template <int KSTEPS, bool BRACKET, int NW = 8, int NBUF = 2, bool LIST = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void mfx_fit_k2x_kernel(FitK2XArgs a) {
  constexpr int MP = KSTEPS * 4;
  constexpr int WG = NW * 64;
  extern __shared__ double smem[];
  if (a.list_count && a.vox_base + (int)blockIdx.x >= *a.list_count) return;   // workgroup-uniform
  if constexpr (LIST) { if (a.xl_cnt[blockIdx.x] < 0) return; }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lg = lane >> 4, lc = lane & 15;
  const int M = a.P.M, N = a.T.N, ldn = a.T.ldn, NP = ldn, ntiles = NP >> 4;
  const int NX = a.X.NX, E = a.X.E, has_csf = a.X.has_csf;
  const int Kp = 2 + has_csf + (E > 0);
  const int ntup = (Kp == 3) ? NX : E;  // extra tuples per (i1,i2)
  const double2* __restrict__ tab = a.T.tab;
  const int vox = a.vox_list ? a.vox_list[a.vox_base + blockIdx.x] : a.vox_base + blockIdx.x;
  double* __restrict__ wsA = a.ws + (size_t)blockIdx.x * 2 * NP * (MFX_XS + 2 * (ntup + 1));  // [k][n][e] atom . extra inner products
  ProjB* __restrict__ wsP = (ProjB*)(wsA + (size_t)2 * NP * MFX_XS);                            // [k][n][ntup + 1]

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
  double* s_red = s_Gxx + MFX_XS * MFX_XS;        // [24] (per-wave partials 0..NW-1, [16]: np.sum(y**2))
  double* s_Qx = s_red + 24;                      // [XS]            best support made of extra columns only, per extra tuple
  CandX* s_cand = (CandX*)(s_Qx + MFX_XS);        // [NW][XMAXC / NW] every wave's own short list
  FamX* s_fam = (FamX*)(s_cand + MFX_XMAXC);      // [XFAM]
  double* s_tc = (double*)(s_fam + MFX_XFAM);     // [XS][4] per extra tuple: l_t, 1/r_t, y.x_t', q0_t   + [2]: 1/|f|, y.f/|f|
  double* s_rowf = s_tc + 4 * MFX_XS + 2;         // [NW][16] d1 . f / |f| of each wave's rows
  double* s_colf = s_rowf + NW * 16;              // [2][16]  d2 . f / |f| of the chunk's columns
  unsigned long long* s_thr = (unsigned long long*)(s_colf + 2 * 16);   // [2] best NNLS score so far (bits of a non-negative double)
  ProjC* s_rowc = (ProjC*)(s_thr + 2);            // [NW][16][ntup]
  ProjC* s_colc = s_rowc + NW * 16 * ntup;        // [2][16][ntup]
  QItem* s_q = (QItem*)(s_colc + 2 * 16 * ntup);  // [NW][MFX_XQ] per wave: tuples that passed the filter, waiting for their score
  double* s_xx = (double*)(s_q + NW * MFX_XQ);    // [M][NX] the extra columns (read by every row of every sum: LDS, not global loads)
  int* s_r0 = (int*)(s_xx + (a.xx_in_lds ? (size_t)M * NX : 0));      // [2][MP]
  int* s_r1 = s_r0 + 2 * MP;
  int* s_cnt = s_r1 + (BRACKET ? 2 * MP : 0);     // [4]
  int* s_qn = s_cnt + 4;                          // [NW][2] per wave: a 64-bit scratch word of the queue logic (8-byte aligned)

  MFX_STAMP(0);
  const double* __restrict__ yv = a.Y + (size_t)vox * M;
  const double* __restrict__ pk = a.peaks + (size_t)vox * a.peaks_ld;
  if (a.xx_in_lds) for (int q = tid; q < M * NX; q += WG) s_xx[q] = a.X.x[q];
  const double* xx = a.xx_in_lds ? s_xx : a.X.x;
  for (int m = tid; m < MP; m += WG) s_y[m] = (m < M) ? yv[m] : 0.0;
  for (int idx = tid; idx < 2 * MP; idx += WG) {
    const int k = idx / MP, m = idx - k * MP;
    RowDesc rd;
    rd.r0 = a.T.P; rd.t0 = 0.0; rd.r1 = -1; rd.t1 = 0.0;
    if (m < M) rd = mfx_row_desc(a.T, a.P, m, pk[3 * k], pk[3 * k + 1], pk[3 * k + 2]);
    s_r0[idx] = rd.r0 * ldn;            // element offset of the knot row (SGPR base + 32-bit offset addressing)
    s_t0[idx] = rd.t0;
    if (BRACKET) {
      s_r1[idx] = rd.r1 < 0 ? -1 : rd.r1 * ldn;
      s_t1[idx] = rd.t1;
      if (k == 0) { s_tG[m] = (m < M) ? a.P.tG[m] : 0.0; s_dG[m] = (m < M) ? a.P.dG[m] : 1.0; }
    }
  }
  for (int q = tid; q < MFX_XS * MFX_XS; q += WG) {
    const int p = q / MFX_XS, r = q - p * MFX_XS;
    s_Gxx[q] = (p < NX && r < NX) ? a.X.Gxx[p * NX + r] : 0.0;
  }
  if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; s_cnt[2] = 0; s_cnt[3] = 0; }
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
    // relaxation of tuple t: Gram-Schmidt of its extra columns R_t = {f, x_t} (f only in the four-column class)
    const double gf = HASF ? s_Gxx[0] : 1.0, rf = sqrt(gf);
    const double l = HASF ? s_Gxx[cx] / rf : 0.0;
    const double gtt = s_Gxx[cx * MFX_XS + cx] - l * l;
    const bool dep = !(gtt > 1e-12 * s_Gxx[cx * MFX_XS + cx]);   // x_t inside span(f): R_t = {f}
    const double irt = dep ? 0.0 : 1.0 / sqrt(gtt);
    const double yf = HASF ? s_Yx[0] / rf : 0.0;
    const double yt = (s_Yx[cx] - yf * l) * irt;
    s_tc[4 * tid] = l; s_tc[4 * tid + 1] = irt; s_tc[4 * tid + 2] = yt; s_tc[4 * tid + 3] = yf * yf + yt * yt;
    if (tid == 0) { s_tc[4 * MFX_XS] = HASF ? 1.0 / rf : 0.0; s_tc[4 * MFX_XS + 1] = yf; }
  }
  const double eps_abs_of = 1e-9;

  // table entry at a 32-bit element offset: SGPR base + VGPR offset addressing (64-bit address arithmetic per element
  // costs several quarter-rate VALU instructions, and beside FP64 MFMAs every VALU instruction counts in full)
  auto tab_at = [&](int eo) -> double2 { return *(const double2*)((const char*)tab + ((unsigned)eo << 4)); };
  auto elem = [&](int k, int m, int n) -> double {
    const double2 e = tab_at(s_r0[k * MP + m] + n);
    const double v0 = e.y * s_t0[k * MP + m] + e.x;          // mfx_eval: separate multiply and add
    if constexpr (BRACKET) {                                 // mfx_eval_br
      const int r1 = s_r1[k * MP + m];
      if (r1 < 0) return v0;
      const double2 f = tab_at(r1 + n);
      const double v1 = f.y * s_t1[k * MP + m] + f.x;
      const double sl = (v1 - v0) / s_dG[m];
      return sl * s_tG[m] + v0;
    }
    return v0;
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
  MFX_STAMP(1);
  // ---- phase 1: column statistics + inner products with the extra columns (sequential over rows)
  double y_sq_seq = 0.0;
  for (int m = 0; m < M; ++m) y_sq_seq += s_y[m] * s_y[m];
  double umax = 0.0;
  // The inner products of the rotated atoms with the extra columns and the signal are a dense product
  // [x_0 .. x_{NX-1} | y]^T D (at most 16 rows: one MFMA row tile) - on FP64 MFMA, 16 atoms at a time, a wave taking
  // every NW-th atom tile of the two dictionaries and reading its B operand (the 16 rotated atoms) straight from the
  // table into registers; |d|^2 comes from the same fragments.  (Thread-per-atom sequential sums took 0.4-1.8 M
  // cycles per voxel here: LDS instruction issue for the extra columns, 64-bit addressing.)  These sums only rank and
  // set thresholds - their order differs from the reference's - every reported number is re-summed in the exact stage.
  {
    double afx[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
      const int m = 4 * kk + lg;
      afx[kk] = (m < M) ? (lc < NX ? xx[(size_t)m * NX + lc] : (lc == NX ? s_y[m] : 0.0)) : 0.0;
    }
    // (list mode: no statistics pass - the screening kernel lists the single atoms that matter, see below)
    for (int tile = wave; tile < (LIST ? 0 : 2 * ntiles); tile += NW) {
      const int k = tile >= ntiles, n0t = (tile - k * ntiles) * 16;
      const int n = n0t + lc;
      double bfr[KSTEPS];
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) bfr[kk] = (n < N) ? elem(k, 4 * kk + lg, n) : 0.0;   // rows beyond M: the zero table row
      d4x c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
      double a2 = 0.0;
      mfx_static_for<0, KSTEPS>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        if constexpr (kk % 2 == 0) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(afx[kk], bfr[kk], c0, 0, 0, 0);
        else c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(afx[kk], bfr[kk], c1, 0, 0, 0);
        a2 = fma(bfr[kk], bfr[kk], a2);
      });
      a2 += __shfl_xor(a2, 16);
      a2 += __shfl_xor(a2, 32);
      // lane (lg, lc): rows lg, lg+4, lg+8, lg+12 of column lc = inner products of atom n with extra columns e = lg + 4r
      double* axp = wsA + ((size_t)k * NP + n) * MFX_XS;
      double ayv = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = lg + 4 * r;
        const double v = c0[r] + c1[r];
        axp[e] = (e < NX) ? v : 0.0;
        if (e == NX) ayv = v;
      }
      // the y row sits in lane group NX & 3: bring d.y to every lane of the column
      ayv = __shfl(ayv, ((NX & 3) << 4) | lc);
      if (lg == 0) {
        (k ? s_A22 : s_A11)[n] = (n < N) ? a2 : 0.0;
        (k ? s_Y2 : s_Y1)[n] = (n < N) ? ayv : 0.0;
      }
    }
    __syncthreads();   // statistics and slab complete (workgroup scope)
    for (int col = tid; col < (LIST ? 0 : 2 * NP); col += WG) {
      const int k = col >= NP, n = col - k * NP;
      if (n < N) {
        const double* axp = wsA + (size_t)col * MFX_XS;
        const double a2v = (k ? s_A22 : s_A11)[n], ayv = (k ? s_Y2 : s_Y1)[n];
        for (int t = 0; t < ntup; ++t) umax = fmax(umax, atom_best(a2v, ayv, axp, t));
      }
    }
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
  if (tid == 0) s_thr[0] = (unsigned long long)__double_as_longlong(gmax_run > 0.0 ? gmax_run : 0.0);
  __syncthreads();   // s_red is reused by the rounds
  // ---- filter constants.  Once per voxel: the threshold-independent part of every (atom, tuple) -> slab wsP
  for (int q = tid; q < (LIST ? 0 : 2 * NP * (ntup + 1)); q += WG) {
    const int kn = q / (ntup + 1), t = q - kn * (ntup + 1);
    const int k = kn >= NP, n = kn - k * NP;
    ProjB pb;
    pb.u = 0.0; pb.np = 0.0f; pb.zp = 0.0f;
    if (n < N) {
      const double a2 = (k ? s_A22 : s_A11)[n], ay = (k ? s_Y2 : s_Y1)[n];
      const double* axp = wsA + (size_t)kn * MFX_XS;
      const double uf = HASF ? axp[0] * s_tc[4 * MFX_XS] : 0.0;
      if (t == ntup) {
        pb.u = uf;
      } else {
        const int cx = x0 + t;
        const double ut = (axp[cx] - uf * s_tc[4 * t]) * s_tc[4 * t + 1];
        const double n2 = a2 - uf * uf - ut * ut;                            // |d'|^2
        const double zn = ay - uf * s_tc[4 * MFX_XS + 1] - ut * s_tc[4 * t + 2];   // d' . y'
        pb.u = ut;
        if (n2 > 1e-10 * a2) {
          const double np = sqrt(n2);
          pb.np = (float)np;
          pb.zp = (float)(zn / np);
        }
      }
    }
    wsP[q] = pb;
  }
  __syncthreads();   // (workgroup-scope visibility of the slab, as for wsA)
  // for a threshold T: (P + D) |d'|, (1 - D) Q |d'| in FP32, P rounded up, Q down (only lets more pairs through)
  // (alw: the atom passes with EVERY partner for this tuple - its own projected score reaches the threshold, or it has no
  // projection left.  The flags travel as bit masks in the low mantissa bits of the row's / column's d.f entry (s_rowf,
  // s_colf: 16 of 52 bits of a ranking value) and are applied to the tests' sign bits in one instruction per row: a constant
  // like pn = 1e18 works only while every partner's pn is positive, and an atom with a negative projection has pn < 0.)
  auto proj_c = [&](const ProjB& pb, int t, double T, bool& alw) -> ProjC {
    ProjC c;
    c.u = pb.u;
    const float Tp = (float)(T - s_tc[4 * t + 3]) * (1.0f - 2e-7f);   // what the two projected atoms must reach (rounded down)
    const float z = pb.zp;
    const bool always = !(pb.np > 0.0f) || !(Tp > 0.0f) || (z > 0.0f && z * z >= Tp * (1.0f - 1e-6f));
    const float rth = __builtin_amdgcn_rsqf(fmaxf(Tp, 1e-30f)) * (1.0f + 4e-7f);
    // P = cos theta over the WHOLE range of z (clamping z at 0 keeps the test valid but lets every sufficiently obtuse partner
    // of an atom with a negative projection through; after the projection on the complement of the extra columns such atoms
    // are common: fit_k3.hip counted 1e7 .. 3e8 false passes per voxel).  An atom whose projection on the signal is negative
    // can still carry a positive weight beside a partner at an obtuse angle, and S(c) = T at c = cos(theta1 + theta2) holds
    // for either sign (cos^2 a + cos^2 b - 2 cos(a + b) cos a cos b = sin^2(a + b)).
    //  * Thresholds: the test is valid for atoms of either sign of z only if BOTH atoms' constants come from the same
    //    threshold (for z > 0 a lower threshold lets more pairs through, for z < 0 - theta decreases with T - fewer): every
    //    chunk has ONE threshold for all its constants, published in s_red[20 + chunk mod 3] two chunks ahead of its use:
    //    the chunk's column constants are made from it, and every wave re-makes its row constants when it differs from theirs.
    //  * Margin: at most one atom of a feasible pair has z < 0.  With P' = P + D, Q' = (1 - D) Q for the positive one and
    //    P' = P + 2 D, Q' = Q - 2 D for the negative one,
    //       P1' P2' - Q1' Q2' - (P1 P2 - Q1 Q2) = D (P1 + Q1 Q2) + 2 D (P2 + D) + 2 D (1 - D) Q2 >= D (-1 + 2 (P2 + Q2)) >= D
    //    for theta1 in [90, 180] and theta2 in [0, 90] degrees (the folding for two positive z needs P1 + P2 + 1.99 Q1 Q2 >= 1,
    //    which fails here: the flood-voxel parity test caught that).
    const bool neg = !(z > 0.0f);
    const float P = fmaxf(-1.0f, fminf(1.0f, z * rth));
    const float Q = __builtin_amdgcn_sqrtf(fmaxf(0.0f, fmaf(-P, P, 1.0f) - 1.2e-7f)) * (1.0f - 3e-7f);
    float pnv = (P + (neg ? 2.0f * MFX_XD : MFX_XD)) * pb.np;
    pnv += fabsf(pnv) * 2e-7f;                                    // rounded up
    float qnv = neg ? (Q - 2.0f * MFX_XD) * pb.np : Q * ((1.0f - MFX_XD) * pb.np);
    qnv -= fabsf(qnv) * 2e-7f;                                    // rounded down
    c.pn = always ? 0.0f : pnv;
    c.qn = always ? 0.0f : qnv;
    alw = always;
    return c;
  };

  // generation of one 16-atom chunk of D2: the table entries are LOADED before the MFMAs of the current chunk and turned
  // into LDS entries after them (the loads fly behind the matrix work instead of in front of it)
  constexpr int NEL = (MP * 16 + WG - 1) / WG;   // elements per thread
  double2 ge0[NEL], ge1[BRACKET ? NEL : 1];
  ProjB gpb;
  gpb.u = 0.0; gpb.np = 0.0f; gpb.zp = 0.0f;
  double ga2x = 0.0;
  auto gen_load = [&](int ch) {
    if (tid < 16 * MFX_XS) ga2x = wsA[((size_t)NP + ch * 16) * MFX_XS + tid];   // A2x of the chunk's 16 atoms
    const int c = tid & 15, m0 = tid >> 4;
    const int n = ch * 16 + c;
    if (tid >= WG - 256) {   // filter base values of the chunk's columns: 16 lanes per column, lane t: tuple t (t = ntup: the fixed column)
      const int q = tid - (WG - 256), cc = q >> 4, t = q & 15;
      if (t <= ntup) gpb = wsP[((size_t)NP + ch * 16 + cc) * (ntup + 1) + t];
    }
#pragma unroll
    for (int p = 0; p < NEL; ++p) {
      const int m = min(m0 + p * (WG / 16), MP - 1);
      ge0[p] = tab_at(s_r0[MP + m] + n);
      if constexpr (BRACKET) {
        const int r1 = s_r1[MP + m];
        ge1[p] = tab_at((r1 < 0 ? a.T.P * ldn : r1) + n);
      }
    }
  };
  auto gen_store = [&](int ch, int buf) {
    const int c = tid & 15, m0 = tid >> 4;
    double* dst = sB + (size_t)buf * (MP * 16) + c;
#pragma unroll
    for (int p = 0; p < NEL; ++p) {
      const int m = m0 + p * (WG / 16);
      if (m < MP) {
        double v = ge0[p].y * s_t0[MP + m] + ge0[p].x;       // mfx_eval: separate multiply and add
        if constexpr (BRACKET) {
          if (s_r1[MP + m] >= 0) {                           // mfx_eval_br
            const double v1 = ge1[p].y * s_t1[MP + m] + ge1[p].x;
            const double sl = (v1 - v) / s_dG[m];
            v = sl * s_tG[m] + v;
          }
        }
        dst[m * 16] = v;
      }
    }
    if (tid < 16 * MFX_XS) s_a2x[buf * 16 * MFX_XS + tid] = ga2x;
    if (tid >= WG - 256) {   // filter constants of the chunk's columns for the current threshold (whole waves: ballot below)
      const int q = tid - (WG - 256), cc = q >> 4, t = q & 15;
      bool alw = false;
      if (t < ntup) s_colc[(buf * 16 + cc) * ntup + t] = proj_c(gpb, t, s_red[20 + ch % 3], alw);   // the chunk's threshold
      const unsigned long long am = __builtin_amdgcn_ballot_w64(alw);
      if (t == ntup) s_colf[buf * 16 + cc] = mfx_pack_alw(gpb.u, (unsigned)(am >> (lane & 48)) & 0xffffu, ntup);
    }
  };
  auto gen_chunk = [&](int ch, int buf) { gen_load(ch); gen_store(ch, buf); };

  const int nrounds = LIST ? 0 : (ntiles + NW - 1) / NW;
  const double eps_abs = eps_abs_of * y_sq;
  MFX_STAMP(2);

  // Every wave keeps its own short list: each scored tuple within the tie tolerance of the best score known at that
  // moment is appended (so everything within the tolerance of the FINAL best is in some list); a full list is compacted
  // against the current threshold, and if that does not make room the voxel takes the exhaustive pass.
  constexpr int WCAP = MFX_XMAXC / NW;
  static_assert(WCAP <= 64, "a wave compacts its list in one pass");
  const int wcap = a.maxc >= MFX_XMAXC ? WCAP : max(1, a.maxc / NW);   // (tests lower maxc to force the exhaustive pass)
  CandX* wl = s_cand + wave * WCAP;
  int wn = 0, wovf = 0;   // wave-uniform: entries in the wave's list, overflow seen

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
    // filter constants of this wave's 16 rows for the threshold thr_rows (refreshed when the threshold has risen)
    double thr_rows = -INFINITY;   // (T = threshold - 1e-9 |y|^2 is below -1 for a large weak signal: no finite sentinel)
    auto row_consts = [&](double T) {
      thr_rows = T;
#pragma unroll 1
      for (int it = 0; it < 4; ++it) {   // 16 lanes per row, lane t: tuple t (t = ntup: the fixed column)
        const int il = 4 * it + (lane >> 4), t = lane & 15;
        ProjB pb;
        pb.u = 0.0; pb.np = 0.0f; pb.zp = 0.0f;
        if (rt_valid && t <= ntup) pb = wsP[((size_t)rt * 16 + il) * (ntup + 1) + t];   // (atoms beyond N: np = 0, they pass and are skipped by the scan)
        bool alw = false;
        if (t < ntup) s_rowc[(wave * 16 + il) * ntup + t] = proj_c(pb, t, T, alw);
        const unsigned long long am = __builtin_amdgcn_ballot_w64(alw);
        if (t == ntup) s_rowf[wave * 16 + il] = mfx_pack_alw(pb.u, (unsigned)(am >> (lane & 48)) & 0xffffu, ntup);
      }
      __builtin_amdgcn_wave_barrier();   // (a wave's LDS operations execute in order: its later reads see these writes)
    };

    if (round == 0) MFX_STAMP(3);
    if (tid == 0) s_red[20] = s_red[21] = __longlong_as_double((long long)s_thr[0]) - eps_abs;   // thresholds of chunks 0 and 1 (slot = chunk mod 3)
    __syncthreads();
    gen_chunk(0, 0);
    __syncthreads();
    if (round == 0) MFX_STAMP(4);
    for (int ch = 0; ch < ntiles; ++ch) {
      const int buf = (NBUF == 2) ? (ch & 1) : 0;
      {
        const double T = s_red[20 + ch % 3];   // the chunk's threshold (its column constants were made from it)
        if (T != thr_rows) row_consts(T);
      }
#ifdef MFX_STAMPS_W   // diagnostic: where a chunk's time goes (chunk 10 of round 1; waves 0 and 7), tools/dev_stamps_k2x.py w
#define MFX_XSTAMP(k) do { if (a.stamps && round == 1 && ch == 10 && (tid == 0 || tid == WG - 64)) a.stamps[(size_t)blockIdx.x * 16 + (tid ? 8 : 0) + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MFX_XSTAMP(k) do { } while (0)
#endif
      MFX_XSTAMP(0);
      if constexpr (NBUF == 2) {
        if (ch + 1 < ntiles) gen_load(ch + 1);
      } else if (ch > 0) {
        gen_chunk(ch, 0);
        __syncthreads();
      }
      MFX_XSTAMP(1);
      d4x acc = {0, 0, 0, 0};
      if (rt_valid) {
        // B operands read PD k-steps ahead into their own registers (a read right in front of its MFMA leaves the
        // matrix pipe idle for an LDS round trip per k-step), two accumulator chains (a dependent FP64 MFMA chain
        // issues every 72 cycles instead of 64); the sum of the chains is ranking-grade like everything in the scan
        const double* bp = sB + (size_t)buf * (MP * 16) + lg * 16 + lc;
        constexpr int PD = 4;
        double bb[PD + 1];
#pragma unroll
        for (int q = 0; q < PD; ++q) bb[q] = bp[q * 64];
        d4x acc1 = {0, 0, 0, 0};
        mfx_static_for<0, KSTEPS>([&](auto kc) {
          constexpr int kk = decltype(kc)::value;
          if constexpr (kk + PD < KSTEPS) bb[(kk + PD) % (PD + 1)] = bp[(kk + PD) * 64];
          if constexpr (kk % 2 == 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[kk], bb[kk % (PD + 1)], acc, 0, 0, 0);
          else acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[kk], bb[kk % (PD + 1)], acc1, 0, 0, 0);
        });
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += acc1[r];
      }
      MFX_XSTAMP(2);
      if constexpr (NBUF == 2) {
        if (ch + 1 < ntiles) gen_store(ch + 1, buf ^ 1);   // nobody reads that buffer during this chunk; the loads have landed
      }
      MFX_XSTAMP(3);
      if (rt_valid) {
        const int j = ch * 16 + lc;
        // ---- filter: which tuples of the lane's four pairs can reach the threshold at all (see the top of the file).
        // fm[r]: shift register of the tests' sign bits (set = the tuple fails), tuple t ends up in bit ntup - 1 - t
        unsigned fm[4] = {~0u, ~0u, ~0u, ~0u};
        if (j < N) {
          const ProjC* cc = s_colc + (buf * 16 + lc) * ntup;
          const ProjC* rc = s_rowc + (wave * 16 + lg) * ntup;
          const double uf2 = s_colf[buf * 16 + lc];
          double accf[4];
          unsigned alwm[4];   // tuples this pair passes whatever the test says (bit layout of fm)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double uf1 = s_rowf[wave * 16 + lg + 4 * r];
            accf[r] = HASF ? fma(-uf1, uf2, acc[r]) : acc[r];
            alwm[r] = ((unsigned)__double_as_longlong(uf1) | (unsigned)__double_as_longlong(uf2)) & 0xffffu;
          }
          const ProjC* r0p = rc, *r1p = rc + 4 * ntup, *r2p = rc + 8 * ntup, *r3p = rc + 12 * ntup;
#pragma unroll 2
          for (int t = 0; t < ntup; ++t) {
            const ProjC c2 = mfx_ldc(cc + t);
            const ProjC c10 = mfx_ldc(r0p + t), c11 = mfx_ldc(r1p + t), c12 = mfx_ldc(r2p + t), c13 = mfx_ldc(r3p + t);
            const float b0 = fmaf(-c10.qn, c2.qn, fmaf(c10.pn, c2.pn, -(float)fma(-c10.u, c2.u, accf[0])));
            const float b1 = fmaf(-c11.qn, c2.qn, fmaf(c11.pn, c2.pn, -(float)fma(-c11.u, c2.u, accf[1])));
            const float b2 = fmaf(-c12.qn, c2.qn, fmaf(c12.pn, c2.pn, -(float)fma(-c12.u, c2.u, accf[2])));
            const float b3 = fmaf(-c13.qn, c2.qn, fmaf(c13.pn, c2.pn, -(float)fma(-c13.u, c2.u, accf[3])));
            // one instruction per test: the sign bit (clear = the tuple passes) is shifted in
            fm[0] = __builtin_amdgcn_alignbit(fm[0], __float_as_uint(b0), 31);
            fm[1] = __builtin_amdgcn_alignbit(fm[1], __float_as_uint(b1), 31);
            fm[2] = __builtin_amdgcn_alignbit(fm[2], __float_as_uint(b2), 31);
            fm[3] = __builtin_amdgcn_alignbit(fm[3], __float_as_uint(b3), 31);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) fm[r] &= ~alwm[r];
        }
        // ---- the passing tuples of the wave's 256 pairs are COMPACTED before they are scored: a few percent of the
        // tuples pass, scattered over the lanes - scored in place, the ~100-instruction FP64 scoring ran for almost every
        // (row group, tuple) with a handful of lanes active (45 % of a config-4 voxel's time).  The wave goes through the
        // (row group, tuple) combinations some lane passes: the passing lanes append (a1.a2, who they are) to the wave's
        // 64-entry LDS queue; a queue that cannot take the next combination is scored first, all lanes busy.
        MFX_XSTAMP(6);
        const unsigned allt = (1u << ntup) - 1u;
        unsigned long long pmask = 0ull;   // bit 16 r + (ntup - 1 - t): tuple t of the lane's pair r passes
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pmask |= (unsigned long long)((j < N && rt * 16 + lg + 4 * r < N) ? (~fm[r] & allt) : 0u) << (16 * r);
#ifdef MFX_STAMPS
        if (a.stamps && pmask) atomicAdd(&a.stamps[(size_t)blockIdx.x * 16 + 10], (unsigned long long)__builtin_popcountll(pmask));   // diagnostics: tuples that pass the filter
#endif
        if (__builtin_amdgcn_ballot_w64(pmask != 0ull)) {   // wave-uniform
          // the combinations present in the wave: OR of the lanes' masks (a wave's LDS operations execute in order and the
          // word is the wave's own: no barrier)
          unsigned long long* wor_p = (unsigned long long*)(s_qn + 2 * wave);
          if (lane == 0) *wor_p = 0ull;
          if (pmask) atomicOr(wor_p, pmask);
          unsigned long long wor = *wor_p;
          wor = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wor >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)wor);
          QItem* q = s_q + wave * MFX_XQ;
          int qn = 0;   // wave-uniform fill of the queue
#ifdef MFX_STAMPS_W
          unsigned long long dbg_total = 0;
#endif
          for (bool more = true; more;) {
            // next combination (or none left: the last pass only scores what is queued)
            const bool have = wor != 0ull;
            const int kb = have ? __ffsll((long long)wor) - 1 : 0;
            wor &= wor - 1ull;
            const bool p = have && ((pmask >> kb) & 1ull);
            const unsigned long long pb = __builtin_amdgcn_ballot_w64(p);
            const int n = __builtin_popcountll(pb);
            more = have;
            if (qn + n <= MFX_XQ && have) {   // room: append and go on
              if (p) {
                const int r = kb >> 4, t = ntup - 1 - (kb & 15);
                const int slot = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(pb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pb, 0u));
                q[slot].v = r == 0 ? acc[0] : (r == 1 ? acc[1] : (r == 2 ? acc[2] : acc[3]));
                q[slot].code = (lg + 4 * r) | (lc << 4) | (t << 8);
              }
              qn += n;
              continue;
            }
            if (have) wor |= 1ull << kb;   // the combination waits for the queue to drain
            more = have;
            const int nq = qn;
            qn = 0;
            if (nq == 0) continue;
#ifdef MFX_STAMPS_W
            dbg_total += nq;
#endif
            __builtin_amdgcn_wave_barrier();
            double sc = -1.0;
            int ci = 0, cj = 0;
            if (lane < nq) {
              const int code = q[lane].code;
              const double a12 = q[lane].v;
              const int il = code & 15, cl = (code >> 4) & 15, t = code >> 8;
              const int i = rt * 16 + il, jj = ch * 16 + cl;
              ci = i; cj = jj;
              const double* a1x = s_a1x + (wave * 16 + il) * MFX_XS;
              const double* a2x = s_a2x + (buf * 16 + cl) * MFX_XS;
              const int cx = x0 + t;
              // every operand is read up front, in one batch of LDS reads (reads scattered between the dependent FP64
              // steps cost a wave that scores alone an LDS round trip each)
              const double a11 = s_A11[i], y1 = s_Y1[i], a22 = s_A22[jj], y2 = s_Y2[jj];
              const double a1f = a1x[0], a2f = a2x[0], aff = s_Gxx[0], yf = s_Yx[0];
              const double a1e = a1x[cx], a2e = a2x[cx], aee = s_Gxx[cx * MFX_XS + cx], ye = s_Yx[cx], afe = s_Gxx[cx];
              // Ranking by feasible supports: the NNLS optimum of a tuple is the best score among the supports whose
              // unconstrained solution is non-negative.  Only the supports with BOTH fascicle atoms are ranked here (the
              // others come from the per-atom scores, see the top of the file): LDL^T of the {1,2}(+fixed) block, the
              // tuple's extra column as the last row.
              const double ip1 = mfx_rcp(a11);
              const double l21 = a12 * ip1;
              const double p2 = fma(-l21, a12, a22);
              const double u2 = fma(-l21, y1, y2);
              const bool ok2 = p2 > 1e-8 * a22;
              const double ip2 = mfx_rcp(ok2 ? p2 : 1.0);
              const double v1 = y1 * ip1, v2 = u2 * ip2;
              const double S2u = fma(u2, v2, y1 * v1);
              const double w1p = fma(-l21, v2, v1);
              sc = (ok2 && w1p >= 0.0 && v2 >= 0.0) ? S2u : 0.0;   // {1,2}
              // fixed (CSF) column: extend the elimination by one row
              double l31 = 0.0, b23 = 0.0, l32 = 0.0, ip3 = 0.0, u3 = 0.0, v3 = 0.0, S3u = 0.0;
              bool ok3 = false;
              if (HASF) {
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
                sc = (ok3 && w1f >= 0.0 && w2f >= 0.0 && v3 >= 0.0) ? fmax(sc, S3u) : sc;  // {1,2,f}
              }
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
                sc = (okx && w1 >= 0.0 && w2 >= 0.0 && w3 >= 0.0) ? fmax(sc, fma(u3x, w3, S2u)) : sc;
              }
              if (HASF) {  // support {1,2,f,x}: last row on top of the {1,2,f} block
                const double t3 = fma(-m2, b23, fma(-m1, a1f, afe));
                const double m3 = t3 * ip3;
                const double p4 = fma(-m3, t3, fma(-m2, t2, fma(-m1, a1e, aee)));
                const double u4 = fma(-m3, u3, fma(-m2, u2, fma(-m1, y1, ye)));
                const bool ok4 = ok3 && (p4 > 1e-8 * aee);
                const double w4 = u4 * mfx_rcp(ok4 ? p4 : 1.0);
                const double w3 = fma(-m3, w4, v3);
                const double w2 = fma(-m2, w4, fma(-l32, w3, v2));
                const double w1 = fma(-m1, w4, fma(-l31, w3, fma(-l21, w2, v1)));
                sc = (ok4 && w1 >= 0.0 && w2 >= 0.0 && w3 >= 0.0 && w4 >= 0.0) ? fmax(sc, fma(u4, w4, S3u)) : sc;
              }
              // a better score raises the workgroup's threshold (the filter constants follow at their next refresh)
              if (sc > __longlong_as_double((long long)s_thr[0])) atomicMax(&s_thr[0], (unsigned long long)__double_as_longlong(sc));
            }
            // ---- short list: whatever is within the tie tolerance of the best score so far (wave-uniform code)
            const double Tn = __longlong_as_double((long long)s_thr[0]) - eps_abs;
            const bool want = sc > 0.0 && sc >= Tn;
            const unsigned long long wb = __builtin_amdgcn_ballot_w64(want);
            if (wb) {
              const int need = __builtin_popcountll(wb);
              if (wn + need > wcap) {   // make room: drop what the risen threshold has left behind
                CandX ent;
                ent.score = 0.0; ent.i = 0; ent.j = 0;
                bool keep = false;
                if (lane < wn) { ent = wl[lane]; keep = ent.score >= Tn; }
                const unsigned long long kb = __builtin_amdgcn_ballot_w64(keep);
                // (every lane has read its entry before any lane writes: one instruction stream)
                if (keep) wl[__builtin_amdgcn_mbcnt_hi((unsigned)(kb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)kb, 0u))] = ent;
                wn = __builtin_popcountll(kb);
              }
              if (wn + need > wcap) {
                wovf = 1;
              } else {
                if (want) {
                  CandX ent;
                  ent.score = sc; ent.i = ci; ent.j = cj;
                  wl[wn + __builtin_amdgcn_mbcnt_hi((unsigned)(wb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wb, 0u))] = ent;
                }
                wn += need;
              }
            }
            __builtin_amdgcn_wave_barrier();   // the next combinations overwrite the queue
          }
#ifdef MFX_STAMPS_W
          if (a.stamps && round == 1 && ch == 10 && (tid == 0 || tid == WG - 64)) a.stamps[(size_t)blockIdx.x * 16 + (tid ? 8 : 0) + 7] = dbg_total;
#endif
        }
      }
      MFX_XSTAMP(4);
      // the threshold of chunk ch + 2 (its column constants are made during chunk ch + 1, after this barrier); its slot held
      // chunk ch - 1's, last read before the barrier that ended that chunk
      if (tid == 0) s_red[20 + (ch + 2) % 3] = __longlong_as_double((long long)s_thr[0]) - eps_abs;
      __syncthreads();
      MFX_XSTAMP(5);
    }
    if (round == 0) MFX_STAMP(5);
  }
  if constexpr (!LIST) {
    if (lane == 0) {
      s_qn[2 * wave] = wn;   // (the queues are idle now: their scratch words carry the list lengths to the exact stage)
      if (wovf) s_cnt[3] = 1;
      atomicAdd(&s_cnt[0], wn);
    }
    __syncthreads();
    gmax_run = fmax(gmax_run, __longlong_as_double((long long)s_thr[0]));   // best pair score of the scan
  }

  MFX_STAMP(6);
  // list mode: |d|^2, d.y and the inner products with the extra columns of ONE rotated atom, summed on demand
  auto single_stats = [&](int k, int n, double& a2, double& ay, double* ax) {
    a2 = 0.0; ay = 0.0;
    for (int e = 0; e < MFX_XS; ++e) ax[e] = 0.0;
    for (int m = 0; m < M; ++m) {
      const double d = elem(k, m, n);
      a2 = fma(d, d, a2);
      ay = fma(s_y[m], d, ay);
      for (int e = 0; e < NX; ++e) ax[e] = fma(d, xx[(size_t)m * NX + e], ax[e]);
    }
  };
  // ---- one-atom and no-atom supports within rounding distance of the optimum -> family items
  auto family_detection = [&]() {
  const double thr_final = gmax_run - eps_abs;
  if constexpr (LIST) {   // only the single atoms the screening kernel listed can be that close
    const int cnt = a.xl_cnt[blockIdx.x];
    const Cand* lst = a.xl_cand + (size_t)blockIdx.x * a.xl_cap;
    for (int q = tid; q < cnt; q += WG) {
      const Cand e = lst[q];
      if (e.i >= 0 && e.j >= 0) continue;
      const int k = e.i < 0, n = k ? e.j : e.i;
      double a2, ay, ax[MFX_XS];
      single_stats(k, n, a2, ay, ax);
      // forced (-2): the screening kernel's bound of the atom's pairs is its relaxed single score, all of them must be seen
      const bool forced = (k ? e.i : e.j) == -2;
      for (int t = 0; t < ntup; ++t)
        if (forced || atom_best(a2, ay, ax, t) >= thr_final) {
          const int f = atomicAdd(&s_cnt[1], 1);
          if (f < MFX_XFAM) { s_fam[f].type = 1 + k; s_fam[f].a = n; s_fam[f].t = t; }
        }
    }
  }
  for (int col = tid; col < (LIST ? 0 : 2 * NP); col += WG) {
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
  };
  if constexpr (!LIST) family_detection();

  MFX_STAMP(7);
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
  // (col1(m), col2(m): the rotated atoms' entries - through the table, or from columns staged in LDS)
  auto consider_cols = [&](int i, int j, int t, auto&& col1, auto&& col2) -> double {
    const int c3 = (Kp == 3) ? t : 0, c4 = 1 + t;
    double a11 = 0.0, a22 = 0.0, a12 = 0.0, y1 = 0.0, y2 = 0.0, a13 = 0.0, a23 = 0.0, a14 = 0.0, a24 = 0.0;
    for (int m = 0; m < M; ++m) {
      const double d1 = col1(m), d2 = col2(m), ym = s_y[m];
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
          const double tt = (ww[0] * col1(m) + ww[1] * col2(m) + ww[2] * xx[(size_t)m * NX + c3] - s_y[m]);
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
        const double tt = (u[0] * col1(m) + u[1] * col2(m) + u[2] * xx[(size_t)m * NX] + u[3] * xx[(size_t)m * NX + c4] - s_y[m]);
        rr += tt * tt;
      }
      r = rr;
      k = ((long)i * N + j) * E + t;  // itertools.product order, last index fastest
    }
    if (r < res || (r == res && k < key)) { res = r; key = k; w[0] = u[0]; w[1] = u[1]; w[2] = u[2]; w[3] = u[3]; }
    return r;
  };
  auto consider = [&](int i, int j, int t) -> double {
    return consider_cols(i, j, t, [&](int m) { return elem(0, m, i); }, [&](int m) { return elem(1, m, j); });
  };
  if constexpr (LIST) {
    // ---- list mode: the screening kernel's short list stands for the pair scan.  Every listed pair through the
    // reference arithmetic at once (the lists are short); the best of their scores joins the one-/no-atom supports
    // in the running maximum the family rule needs.  A pair that beats the upper bound it was listed with by more than
    // the screening margin cannot happen if the margin holds: the voxel then goes to the plain kernel (and is counted).
    const int cnt = a.xl_cnt[blockIdx.x];
    const Cand* lst = a.xl_cand + (size_t)blockIdx.x * a.xl_cap;
    const double xmrg = a.xl_mrg[blockIdx.x];
    bool beaten = false;
    double lbs = 0.0;   // best one-atom support among the listed single atoms
    if (Kp == 3 && NX == 1) {
      // The screening pipeline's class ([N, N, 1]).  A thread that walks its pair's 2 x M rows through the table alone pays
      // an L2 round trip per row block and sums seven products per row in one instruction stream (156 k of a voxel's
      // 200 k cycles).  Instead, per batch of PB listed entries: the whole workgroup stages the rotated columns in the
      // (idle) chunk buffers; EIGHT lanes per pair take one of the reference's seven sequential sums each (the sums stay
      // sequential over the rows, only different sums run side by side); lane 0 of the group solves the triple and sums
      // the explicit residual from the staged columns.
      constexpr int PB = NBUF * 8;   // listed entries per batch: two columns of MP rows each
      double* s_sum = s_a1x;         // [PB][8] (the row staging of the scan is idle in list mode)
      static_assert(PB * 8 <= NW * 16 * MFX_XS && PB * 8 <= WG, "list-mode batch");
      for (int c0 = 0; c0 < cnt; c0 += PB) {
        const int nb = min(PB, cnt - c0);
        for (int q = tid; q < nb * 2 * MP; q += WG) {
          const int c = q / (2 * MP), r = q - c * (2 * MP), k = r / MP, m = r - k * MP;
          const Cand e = lst[c0 + c];
          const int n = k ? e.j : e.i;   // (negative: a single atom of the other dictionary)
          sB[q] = (n >= 0 && m < M) ? elem(k, m, n & 0x3fffffff) : 0.0;
        }
        __syncthreads();
        if (tid < nb * 8) {
          const int c = tid >> 3, sl = tid & 7;
          const Cand e = lst[c0 + c];
          const double* d1s = sB + (size_t)c * (2 * MP);
          const double* d2s = d1s + MP;
          const bool pair = e.i >= 0 && e.j >= 0;
          if (pair && sl < 7) {
            // sum sl of mf_utils.py:548-553's Gram scalars: a11 a22 a12 y1 y2 a13 a23 (factor order as in consider())
            const double* p = (sl == 1 || sl == 4 || sl == 6) ? d2s : ((sl == 3) ? s_y : d1s);
            const double* q = (sl == 0) ? d1s : ((sl == 1 || sl == 2) ? d2s : ((sl == 3) ? d1s : ((sl == 4) ? s_y : xx)));
            // (y1 += ym * d1, y2 += ym * d2: the products commute exactly)
            double acc = 0.0;
#pragma unroll 8
            for (int m = 0; m < M; ++m) acc += p[m] * q[m];   // (NX == 1: the extra column has stride 1; unrolled: the LDS reads of eight rows fly ahead of the sequential adds)
            s_sum[tid] = acc;
          }
          __builtin_amdgcn_wave_barrier();   // (a group sits inside one wave; its LDS operations execute in order)
          if (sl == 0) {
            if (pair) {
              const int i = e.i, j = e.j & 0x3fffffff;
              const double a11 = s_sum[tid], a22 = s_sum[tid + 1], a12 = s_sum[tid + 2], y1 = s_sum[tid + 3], y2 = s_sum[tid + 4],
                           a13 = s_sum[tid + 5], a23 = s_sum[tid + 6];
              double u[3], r;
              auto explicit_res = [&](const double* ww) {
                double rr = 0.0;
#pragma unroll 8
                for (int m = 0; m < M; ++m) {
                  const double tt = (ww[0] * d1s[m] + ww[1] * d2s[m] + ww[2] * xx[m] - s_y[m]);
                  rr += tt * tt;
                }
                return rr;
              };
              nnls3_cramer(y_sq, a11, a12, a13, a22, a23, s_Gxx[0], y1, y2, s_Yx[0], explicit_res, u, r);
              const long k = (long)i * N + j;   // scan order of _3 with one extra tuple: i1 -> i2
              if (r < res || (r == res && k < key)) { res = r; key = k; w[0] = u[0]; w[1] = u[1]; w[2] = u[2]; w[3] = 0.0; }
              beaten |= (y_sq - r) > e.score + 1.25 * xmrg;
            } else {   // a single atom of dictionary 0 (j < 0) or 1 (i < 0): its best one-atom support (ranking only)
              const double* ds = (e.i < 0) ? d2s : d1s;
              double a2 = 0.0, ay = 0.0, ax0 = 0.0;
              for (int m = 0; m < M; ++m) {
                const double d = ds[m];
                a2 = fma(d, d, a2);
                ay = fma(s_y[m], d, ay);
                ax0 = fma(d, xx[m], ax0);
              }
              const double ax1[1] = {ax0};
              lbs = fmax(lbs, atom_best(a2, ay, ax1, 0));
            }
          }
        }
        __syncthreads();   // the next batch overwrites the staged columns
      }
    } else {
    for (int q = tid; q < cnt * ntup; q += WG) {
      const int c = q / ntup, t = q - c * ntup;
      const Cand e = lst[c];
      if (e.i >= 0 && e.j >= 0) {
        const double r = consider(e.i, e.j & 0x3fffffff, t);
        beaten |= (y_sq - r) > e.score + 1.25 * xmrg;
      } else if (t == 0) {   // a single atom of dictionary 0 (j < 0) or 1 (i < 0)
        const int k = e.i < 0, n = k ? e.j : e.i;
        double a2, ay, ax[MFX_XS];
        single_stats(k, n, a2, ay, ax);
        for (int tt = 0; tt < ntup; ++tt) lbs = fmax(lbs, atom_best(a2, ay, ax, tt));
      }
    }
    }
    MFX_STAMP(12);   // (diagnostic builds: listed pairs and single atoms evaluated)
    double lb = fmax((key >= 0) ? y_sq - res : 0.0, lbs);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lb = fmax(lb, __shfl_xor(lb, o));
    if (lane == 0) s_red[wave] = lb;
    if (beaten) s_cnt[2] = 1;
    __syncthreads();
    for (int c = 0; c < NW; ++c) gmax_run = fmax(gmax_run, s_red[c]);
    if (s_cnt[2] && tid == 0) {
      const int slot = atomicAdd(a.fb_count, 1);
      a.fb_list[slot] = vox;
      atomicAdd(a.fb_count + 1, 1);
    }
    __syncthreads();   // s_red is free again
    MFX_STAMP(13);
    family_detection();
    MFX_STAMP(14);
    // more families than the exact stage takes (a voxel whose signal is mostly the extra column: every atom's relaxed
    // score reaches the threshold): not this kernel's exhaustive pass over all tuples - the plain kernel has its own scan
    // and decides what is really needed
    if (s_cnt[1] > MFX_XFAM) {
      if (tid == 0 && !s_cnt[2]) {
        const int slot = atomicAdd(a.fb_count, 1);
        a.fb_list[slot] = vox;
      }
      return;
    }
  }
  // ---- exact re-evaluation (reference arithmetic and summation order) of everything short-listed
  const double thr_final = gmax_run - eps_abs;
  const int ncand = s_cnt[0], nfam_app = s_cnt[1];   // short-listed (pair, tuple) entries of all waves; family items
  // (a voxel in which no support scores above zero keeps the reference's initial state: nothing to evaluate)
  const bool nothing = !(gmax_run > 0.0);
  if (tid == 0 && a.ovf_count) {   // diagnostics: short-listed pairs and family items of the launch
    atomicAdd(a.ovf_count + 2, LIST ? a.xl_cnt[blockIdx.x] : ncand);
    atomicAdd(a.ovf_count + 3, nfam_app);
  }
  const bool exhaustive = !nothing && (a.maxc == 0 || s_cnt[3] != 0 || nfam_app > MFX_XFAM);   // workgroup-uniform
  __syncthreads();
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
    // short-listed pairs, every extra tuple (they tie when the extra column is inactive).  A pair is listed once per
    // tuple that scored near the optimum: the later copies are struck out first.
    if constexpr (!LIST) {
      for (int c = tid; c < NW * WCAP; c += WG) {
        const int w = c / WCAP, k = c - w * WCAP;
        if (k < s_qn[2 * w] && s_cand[c].score >= thr_final) {
          bool dup = false;
          for (int c2 = 0; c2 < c && !dup; ++c2) {
            const int w2 = c2 / WCAP;
            dup = (c2 - w2 * WCAP) < s_qn[2 * w2] && s_cand[c2].i == s_cand[c].i && s_cand[c2].j == s_cand[c].j && s_cand[c2].score >= thr_final;
          }
          if (dup) s_cand[c].score = -2.0;
        }
      }
      __syncthreads();
      // A thread that walks its tuple's 2 x M rows through the table alone pays an L2 round trip per row block, twice (Gram
      // scalars, explicit residual): 370 k of a config-4 voxel's cycles.  Instead, per batch of PB listed pairs the whole
      // workgroup stages the two rotated columns of every pair in the (idle) chunk buffers and one thread per (pair, tuple)
      // runs the reference's arithmetic on the staged columns - the same values in the same order.
      constexpr int PB = NBUF * 8;          // pairs per batch: two columns of MP rows each
      int* s_pl = (int*)s_a1x;              // [NW * WCAP] slots of the surviving pairs, compacted; [NW * WCAP]: their number
      static_assert(NW * WCAP + 1 <= NW * 16 * MFX_XS * 2, "pair list in the row staging area");
      if (wave == 0) {
        int n = 0;
        for (int c0 = 0; c0 < NW * WCAP; c0 += 64) {
          const int c = c0 + lane, w = c / WCAP, k = c - w * WCAP;
          const bool ok = c < NW * WCAP && k < s_qn[2 * w] && s_cand[c].score >= thr_final;
          const unsigned long long mk = __ballot(ok);
          if (ok) s_pl[n + __popcll(mk & ((1ull << lane) - 1ull))] = c;
          n += __popcll(mk);
        }
        if (lane == 0) s_pl[NW * WCAP] = n;
      }
      __syncthreads();
      const int npl = s_pl[NW * WCAP];
      for (int c0 = 0; c0 < npl; c0 += PB) {
        const int nb = min(PB, npl - c0);
        for (int q = tid; q < nb * 2 * MP; q += WG) {
          const int c = q / (2 * MP), r = q - c * (2 * MP), k = r / MP, m = r - k * MP;
          const CandX e = s_cand[s_pl[c0 + c]];
          sB[q] = (m < M) ? elem(k, m, k ? e.j : e.i) : 0.0;
        }
        __syncthreads();
        for (int q = tid; q < nb * ntup; q += WG) {
          const int c = q / ntup, t = q - c * ntup;
          const CandX e = s_cand[s_pl[c0 + c]];
          const double* d1s = sB + (size_t)c * (2 * MP);
          const double* d2s = d1s + MP;
          consider_cols(e.i, e.j, t, [&](int m) { return d1s[m]; }, [&](int m) { return d2s[m]; });
        }
        __syncthreads();   // the next batch overwrites the staged columns
      }
    }
    for (int f = 0; f < nfam_app; ++f) {   // workgroup-uniform loop
      const int type = s_fam[f].type, fa = s_fam[f].a, ft = s_fam[f].t;
      if (type == 1) { for (int n = tid; n < N; n += WG) consider(fa, n, ft); }
      else if (type == 2) { for (int n = tid; n < N; n += WG) consider(n, fa, ft); }
      else if (type == 3) { if (tid == 0) consider(0, 0, ft); }
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
  MFX_STAMP(8);
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
  MFX_STAMP(9);
}
