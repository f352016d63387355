// fit_k2.hip -- fused per-voxel kernel for voxels with two fascicles (sub-dictionary sizes [N, N]).
//
// Replaces, for one voxel per workgroup, the reference chain
//   _fit_voxel (mf.py:340-461) -> 2 x interp_PGSE_from_multishell (mf_utils.py:1693-1956)
//   -> solve_exhaustive_posweights_2 (mf_utils.py:288-392) -> params_vox packing (mf.py:420-450).
//
// Structure (one 512-thread workgroup = 8 waves, 2 per SIMD, one voxel):
//   phase 0  y -> LDS; per (direction,row) knot-interval descriptors (binary search) -> LDS
//   phase 1  column statistics ||d||^2 and d.y of both rotated dictionaries, one thread per atom
//   phase 2  the cross-Gram D1^T D2 on FP64 MFMA (v_mfma_f64_16x16x4_f64): each wave keeps the
//            A operand (its 16 atoms of D1 over all M rows) in registers, generated straight from
//            the L2-resident table; D2 is generated 32 atoms at a time into a double-buffered LDS
//            tile shared by the 8 waves.  The 2x2 NNLS of mf_utils.py:341-379 is evaluated on the
//            accumulator tile in registers, division-free (candidates are compared as fractions
//            by cross-multiplication), keeping one best candidate per (lane,row).
//   phase 3  short-listed candidates are re-evaluated in the reference's exact arithmetic and loop
//            order, the reference's strict-'<' first-hit rule picks the winner, and the voxel's
//            parameters are written.
// Rotated dictionaries are never materialised in HBM: per voxel the kernel reads y (M doubles),
// two directions, and writes num_params doubles.
//
// Two code paths share phases 0 and 3:
//   FAST (exact-G protocols): measured on MI355X, FP64 MFMA and VALU instructions do NOT overlap on a
//        SIMD (the DGEMM MFMA runs on the FP64 vector datapath: SIMD time = 64.4 cycles per MFMA +
//        ~4.5 cycles per VALU instruction, profiles/r01_micro_mfma_model.txt), so the path minimises
//        instruction COUNT: both operands are pre-normalised (the accumulator is the cosine c of the
//        atom pair), the scan handles only the two-positive-weights case (16 VALU per entry); the
//        single-active cases are represented by the two best single atoms and resolved by phase 3's
//        family expansion; the next D2 chunk is generated in slices inside the MFMA k-loop (hides the
//        L2 latency of the table loads); no register spills (scratch traffic would show up as HBM bytes).
//   GENERIC (protocols with G-bracketed rows, or MFX_K2_PIPE=0): straightforward chunk loop.
#pragma once
#include <type_traits>

#include "mfx_device.h"
#include "nnls_small.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define MFX_WG 512
#define MFX_MAXC 256
#define MFX_S_CAP 2048      // screening kernel (fit_k2s.hip): ring entries (power of two): two full single-atom families (2 x 782) fit
#define MFX_DET_REL 1e-8   // pairs with 1 - cos^2 below this are ranked as single atoms
#define MFX_A12_REL 4e-14  // bound on the relative rounding difference of an MFMA-summed Gram entry

struct FitK2Args {
  TablesDev T;
  PlanDev P;
  const double* Y;      // [V x M]
  const double* peaks;  // [V x peaks_ld]
  int peaks_ld;
  const int* vox_list;  // [nvox] or null (identity)
  const int* list_count;  // null, or device word holding the number of valid vox_list entries (blocks beyond it exit)
  double* params;       // [V x num_params]
  int num_params;
  int maxfasc;
  int csf_on, ear_on;
  unsigned long long* stamps;  // diagnostic builds only: [gridDim.x][16] s_memtime stamps (null otherwise)
  int* fb_count;        // screening kernel (fit_k2s.hip) only: number of voxels handed back to the FP64 kernel ...
  int* fb_list;         // ... and their voxel indices
  int maxc;             // FP64 kernel: short-list size beyond which the exhaustive exact pass runs (MFX_MAXC; tests lower it)
  int scap;             // screening kernel: ring entries in use (MFX_S_CAP, a power of two; tests lower it to force hand-backs)
  // screening kernel in its [N, N, 1] form (XC: two fascicles + one fixed extra column, fit_k2s.hip): the column, the
  // per-voxel short lists it writes for fit_k2x.hip's exact stage, and the first voxel of the launch
  const double* xc;     // [M] the extra column (CSF signal)
  struct Cand* xl_cand; // [gridDim.x][xl_cap] short-listed pairs of every voxel of the launch
  int* xl_cnt;          // [gridDim.x] their number, or -1: voxel handed back
  double* xl_mrg;       // [gridDim.x] the voxel's screening margin (score units)
  int xl_cap;
  int vox_base;         // voxel (or vox_list entry) of block 0
  // screening kernels: population audit of the split-FP16 cross product (k2s_shared.h) - [0] audited pairs with
  // |c~ - c| > MFX_S_DC / 4, [1] the largest |c~ - c| in units of 1e-11, [2] audited pairs; null: no audit
  int* audit;
};

#ifdef MFX_STAMPS
#define MFX_STAMP(i) do { if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MFX_STAMP(i) do { } while (0)
#endif

struct Cand {
  double score;  // upper bound of the candidate's score
  int i, j;
};

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int I, int N, typename F>
__device__ __forceinline__ void mfx_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    mfx_static_for<I + 1, N>(f);
  }
}

// NW waves per workgroup, TILES 16-atom column tiles per D2 chunk, NBUF LDS buffers for the chunks.
// (8, 2, 2) is the tuned configuration (M <= 200: the A operand fits in 100 VGPRs, 2 waves per SIMD);
// (4, 1, 1) serves long protocols (M up to 560): one wave per SIMD owns the whole 512-register file.
template <int KSTEPS, bool BRACKET, bool PIPE, int NW = 8, int TILES = 2, int NBUF = 2>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void mfx_fit_k2_kernel(FitK2Args a) {
  constexpr bool FAST = PIPE && !BRACKET;
  constexpr int WG = NW * 64;
  static_assert(!FAST || (NW == 8 && TILES == 2 && NBUF == 2), "the pipelined path is written for 8 waves, 2 tiles, 2 buffers");
  static_assert(TILES == 1 || TILES == 2, "TILES");
  static_assert(NBUF == 1 || NBUF == 2, "NBUF");
  constexpr int MP = KSTEPS * 4;              // padded measurement count
  constexpr int MPS = ((MP + 15) / 16) * 16;  // rows of one LDS D2 tile (lets the pipelined writer skip a bounds test)
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lg = lane >> 4, lc = lane & 15;
  const int M = a.P.M, N = a.T.N, ldn = a.T.ldn;
  const int NP = ldn;  // atoms padded to a multiple of 16
  const int ntiles = NP >> 4;
  const double2* __restrict__ tab = a.T.tab;
  if (a.list_count && (int)blockIdx.x >= *a.list_count) return;   // device-side hand-back list (fit_k2s.hip): workgroup-uniform
  const int vox = a.vox_list ? a.vox_list[blockIdx.x] : blockIdx.x;

  // ---- LDS carve-up
  double* sB = smem;                             // [NBUF][TILES][MPS][16]
  double* s_y = sB + NBUF * TILES * MPS * 16;    // [MP]
  double* s_t0 = s_y + MP;                       // [2][MP]
  double* s_t1 = s_t0 + 2 * MP;                  // [2][MP] (bracket only)
  double* s_tG = s_t1 + (BRACKET ? 2 * MP : 0);  // [MP]
  double* s_dG = s_tG + (BRACKET ? MP : 0);      // [MP]
  // six [NP] statistic arrays.  GENERIC: A11,Y1,A22,Y2,S2,S1.  FAST: I1=1/|d1|, Z1=d1.y/|d1|, I2, Z2.
  double* s_A11 = s_dG + (BRACKET ? MP : 0);
  double* s_Y1 = s_A11 + NP;
  double* s_A22 = s_Y1 + NP;
  double* s_Y2 = s_A22 + NP;
  double* s_S2 = s_Y2 + NP;
  double* s_S1 = s_S2 + NP;
  double* s_red = s_S1 + NP;                    // [32] scratch
  Cand* s_cand = (Cand*)(s_red + 32);           // [MFX_MAXC]
  int* s_r0 = (int*)(s_cand + MFX_MAXC);        // [2][MP]
  int* s_r1 = s_r0 + 2 * MP;                    // [2][MP] (bracket only)
  int* s_cnt = s_r1 + (BRACKET ? 2 * MP : 0);   // [4] counters

  MFX_STAMP(0);
  // ---- phase 0: y, descriptors
  const double* __restrict__ yv = a.Y + (size_t)vox * M;
  const double* __restrict__ pk = a.peaks + (size_t)vox * a.peaks_ld;
  for (int m = tid; m < MP; m += WG) s_y[m] = (m < M) ? yv[m] : 0.0;
  for (int idx = tid; idx < 2 * MP; idx += WG) {
    const int k = idx / MP, m = idx - k * MP;
    RowDesc rd;
    rd.r0 = a.T.P; rd.t0 = 0.0; rd.r1 = -1; rd.t1 = 0.0;  // padded rows -> the all-zero table row
    if (m < M) rd = mfx_row_desc(a.T, a.P, m, pk[3 * k], pk[3 * k + 1], pk[3 * k + 2]);
    s_r0[idx] = rd.r0;
    s_t0[idx] = rd.t0;
    if (BRACKET) {
      s_r1[idx] = rd.r1;
      s_t1[idx] = rd.t1;
      if (k == 0) { s_tG[m] = (m < M) ? a.P.tG[m] : 0.0; s_dG[m] = (m < M) ? a.P.dG[m] : 1.0; }
    }
  }
  if (tid == 0) s_cnt[0] = 0;
  if (tid < 2) mfx_check_dir(a.P, pk + 3 * tid, vox);
  __syncthreads();

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

  MFX_STAMP(1);
  // ---- phase 1: column statistics (sequential over the measurements, as mf_utils.py:307-325), y_sq likewise
  double y_sq = 0.0;
  for (int m = 0; m < M; ++m) y_sq += s_y[m] * s_y[m];
  double my_s[2] = {0.0, 0.0};  // FAST: this thread's best single-atom score per dictionary ...
  int my_n[2] = {0, 0};         // ... and its (first) atom
  for (int col = tid; col < 2 * NP; col += WG) {
    const int k = col >= NP, n = col - k * NP;
    double a2 = 0.0, ay = 0.0;
    if (n < N) {
#pragma unroll 8
      for (int m = 0; m < M; ++m) {
        if constexpr (FAST) {   // ranking statistics only (the exact stage re-sums in reference order): fused ops
          const double2 e = tab[(size_t)s_r0[k * MP + m] * ldn + n];
          const double d = fma(e.y, s_t0[k * MP + m], e.x);
          a2 = fma(d, d, a2);
          ay = fma(s_y[m], d, ay);
        } else {
          const double d = elem(k, m, n);
          a2 += d * d;
          ay += s_y[m] * d;
        }
      }
    }
    if constexpr (FAST) {
      const double inv = (n < N && a2 > 0.0) ? 1.0 / sqrt(a2) : 0.0;
      const double z = ay * inv;
      (k ? s_A22 : s_A11)[n] = inv;  // I1 / I2
      (k ? s_Y2 : s_Y1)[n] = z;      // Z1 / Z2
      const double s = z > 0.0 ? z * z : 0.0;
      if (s > my_s[k]) { my_s[k] = s; my_n[k] = n; }  // columns are visited in increasing n per thread
    } else {
      (k ? s_A22 : s_A11)[n] = a2;
      (k ? s_Y2 : s_Y1)[n] = ay;
      const double s = (n < N && ay > 0.0) ? (ay * ay) / a2 : 0.0;
      (k ? s_S2 : s_S1)[n] = s;
      if (s > my_s[k]) { my_s[k] = s; my_n[k] = n; }
    }
  }
  const double eps_abs = 1e-9 * y_sq;
  double glb_run = 0.0;  // running best lower bound on the score (same value in every thread)
  {
    // best single atom of each dictionary (first index on ties): they stand for every pair whose
    // optimum has one active atom (mf_utils.py:357-379); phase 3 expands the winner's family exactly.
    // (Both paths: ranking the single-active cases inside the scan floods the short list with one entry per
    // (lane,row) slot whenever one atom suffices -- 513 entries > MFX_MAXC on a 0.14 % / 99.86 % voxel.)
    double* s_bs = s_red;            // [2][8] per-wave bests
    int* s_bn = (int*)(s_red + 16);  // [2][8]
#pragma unroll
    for (int k = 0; k < 2; ++k) {
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
    double best1 = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      double s = s_bs[k * 8];
      int n = s_bn[k * 8];
      for (int w = 1; w < NW; ++w) {
        const double s2 = s_bs[k * 8 + w];
        const int n2 = s_bn[k * 8 + w];
        if (s2 > s || (s2 == s && n2 < n)) { s = s2; n = n2; }
      }
      best1 = fmax(best1, s);
      if (tid == 0 && s > 0.0) {
        const int slot = s_cnt[0]++;
        s_cand[slot].score = s + eps_abs;
        s_cand[slot].i = k ? 0 : n;
        s_cand[slot].j = k ? n : 0;
      }
    }
    glb_run = best1;
  }
  __syncthreads();
  if (tid == 0) s_cnt[1] = s_cnt[0];   // the single-atom representatives appended so far (0..2); kept in LDS, not in a register

  // generation of one (16*TILES)-atom chunk of D2 into LDS buffer `buf`: thread -> (atom c, rows m0 + RS*p)
  constexpr int CW = 16 * TILES;  // atoms per chunk
  constexpr int RS = WG / CW;     // row stride of one thread
  auto gen_chunk = [&](int ch, int buf) {
    const int c = tid % CW, m0 = tid / CW;
    const int n = ch * CW + c;
    double* dst = sB + (size_t)buf * (TILES * MPS * 16) + (c >> 4) * (MPS * 16) + (c & 15);
    if (n < NP) {
      const double sc = FAST ? s_A22[n] : 1.0;  // FAST: normalised columns
#pragma unroll 4
      for (int m = m0; m < MP; m += RS) dst[m * 16] = FAST ? elem(1, m, n) * sc : elem(1, m, n);
    } else {
      for (int m = m0; m < MP; m += RS) dst[m * 16] = 0.0;
    }
  };

  MFX_STAMP(2);
  const int nchunks = (ntiles + TILES - 1) / TILES;
  const int nrounds = (ntiles + NW - 1) / NW;

  for (int round = 0; round < nrounds; ++round) {
    // mode of this wave in this round: 0 idle, 1 both column tiles of every chunk, 2 / 3 only tile 0 / 1.
    // A last round with ONE row tile left (N = 782: 49 = 6*8 + 1) is shared by waves 0 and 1 (they sit on
    // different SIMDs): each takes one of the two column tiles -> that round costs half the MFMA time.
    int rt = round * NW + wave;
    int mode = (rt < ntiles) ? 1 : 0;
    if (FAST && ntiles - round * NW == 1) {
      rt = round * NW;
      mode = (wave == 0) ? 2 : ((wave == 1) ? 3 : 0);
    }
    const bool rt_valid = mode != 0;  // wave-uniform
    const int rtc = rt_valid ? rt : 0;
    // A operand: this wave's 16 atoms of D1, all KSTEPS k-steps, in registers
    double afr[KSTEPS];
    {
      const double asc = FAST ? (rt_valid ? s_A11[rtc * 16 + lc] : 0.0) : 1.0;  // FAST: normalised atoms
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        const double v = rt_valid ? elem(0, 4 * kk + lg, rtc * 16 + lc) : 0.0;
        afr[kk] = FAST ? v * asc : v;
      }
    }
    double bp[4], bq[4];
    int bj[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { bp[r] = 0.0; bq[r] = 1.0; bj[r] = -1; }

    if (round == 0) MFX_STAMP(3);
    gen_chunk(0, 0);
    __syncthreads();
    if (round == 0) MFX_STAMP(4);

    if constexpr (FAST) {
      double z1r[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) z1r[r] = s_Y1[rtc * 16 + lg + 4 * r];
      // two-positive-weights case of one accumulator entry; both MFMA operands are normalised when they
      // are generated: c = cos(atom i, atom j), z = d.y/|d|
      // (returns true for a pair with two positive weights whose 1 - c^2 is too small to be ranked as a fraction: see ill_pairs)
      auto scan_one = [&](double c, int r, int j, double z2) -> bool {
        const double e1 = fma(-c, z2, z1r[r]);
        const double e2 = fma(-c, z1r[r], z2);
        const double den = fma(-c, c, 1.0);
        const double num = fma(z2, e2, z1r[r] * e1);
        const bool pos = (e1 > 0.0) & (e2 > 0.0);
        const bool better = pos & (den > MFX_DET_REL) & (num * bq[r] > bp[r] * den);
        bp[r] = better ? num : bp[r];
        bq[r] = better ? den : bq[r];
        bj[r] = better ? j : bj[r];
        return pos & !(den > MFX_DET_REL);
      };
      // Nearly collinear atom pairs (1 - c^2 <= MFX_DET_REL) whose two-atom solution has two positive weights cannot be
      // ranked - the score's error grows like 1/(1 - c^2) - but the reference does solve them (mf_utils.py:348-356: no test on
      // Det), and down to 1 - c^2 ~ 1e-12 its answer is well defined.  They go to the short list unranked (always
      // evaluated exactly; with them their slot row, see phase 3).  Real dictionaries have none (HCP: 1 - c^2 >= 8e-4); a
      // dictionary of two-parameter decays has dozens per voxel, and a voxel whose list overflows takes the exhaustive pass.
      auto ill_pairs = [&](const d4& acc, int j, double z2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double c = acc[r];
          const bool ill = (fma(-c, z2, z1r[r]) > 0.0) && (fma(-c, z1r[r], z2) > 0.0) && !(fma(-c, c, 1.0) > MFX_DET_REL);
          const int i = rtc * 16 + lg + 4 * r;
          if (ill && rt_valid && i < N && j < N) {
            const int slot = atomicAdd(&s_cnt[0], 1);
            if (slot < MFX_MAXC) { s_cand[slot].score = 1e300; s_cand[slot].i = i; s_cand[slot].j = j; }
          }
        }
      };
      constexpr int NEL = (MP + 15) / 16;                                  // D2 elements per thread per chunk
      constexpr int GS = (KSTEPS - 5) / NEL > 0 ? (KSTEPS - 5) / NEL : 1;  // k-steps between two element loads
      constexpr int GD = 4;                                                // load -> use distance in k-steps
      constexpr int PD = 3;                                                // B operand read-ahead in k-steps
      static_assert(GS * (NEL - 1) + GD < KSTEPS, "generation slices do not fit in the k-loop");
      const int gc = tid & 31, gm0 = tid >> 5;
      const int* gr = s_r0 + MP;  // direction-1 descriptors
      const double* gt = s_t0 + MP;
      for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        const int n_next = (ch + 1) * 32 + gc;
        const double gsc = (n_next < NP) ? s_A22[n_next] : 0.0;  // 0 -> columns beyond the dictionary stay zero
        double* gdst = sB + (size_t)(buf ^ 1) * (2 * MPS * 16) + (gc >> 4) * (MPS * 16) + (gc & 15);
        const double2* gsrc = tab + min(n_next, NP - 1);
        const double* b0p = sB + (size_t)buf * (2 * MPS * 16) + lg * 16 + lc;
        const double* b1p = b0p + MPS * 16;
        d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        double2 gl[3];
        double gtv[3];
        auto body = [&](auto mode_c) {
          constexpr int MODE = decltype(mode_c)::value;
          constexpr bool T0 = (MODE == 1 || MODE == 2), T1 = (MODE == 1 || MODE == 3);
          double bb0[PD + 1], bb1[PD + 1];
#pragma unroll
          for (int q = 0; q < PD; ++q) {
            if constexpr (T0) bb0[q] = b0p[q * 64];
            if constexpr (T1) bb1[q] = b1p[q * 64];
          }
          mfx_static_for<0, KSTEPS>([&](auto kc) {
            constexpr int kk = decltype(kc)::value;
            if constexpr (kk + PD < KSTEPS) {
              if constexpr (T0) bb0[(kk + PD) % (PD + 1)] = b0p[(kk + PD) * 64];
              if constexpr (T1) bb1[(kk + PD) % (PD + 1)] = b1p[(kk + PD) * 64];
            }
            if constexpr (T0) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[kk], bb0[kk % (PD + 1)], acc0, 0, 0, 0);
            if constexpr (T1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[kk], bb1[kk % (PD + 1)], acc1, 0, 0, 0);
            // slice of the next chunk's generation: issue the load of element p ...
            if constexpr (kk % GS == 0 && kk / GS < NEL) {
              constexpr int p = kk / GS;
              const int m = min(gm0 + 16 * p, MP - 1);
              gtv[p % 3] = gt[m];
              gl[p % 3] = gsrc[(size_t)gr[m] * ldn];
            }
            // ... and GD k-steps later turn it into a (normalised) D2 entry in the other LDS buffer;
            // rows beyond MP land in the tile's padding, the last chunk writes a dummy successor
            if constexpr (kk >= GD && (kk - GD) % GS == 0 && (kk - GD) / GS < NEL) {
              constexpr int p = (kk - GD) / GS;
              gdst[(gm0 + 16 * p) * 16] = fma(gl[p % 3].y, gtv[p % 3], gl[p % 3].x) * gsc;
            }
          });
          // pair scan of the accumulator tile(s) (VALU work cannot hide behind FP64 MFMAs anyway)
          const int j0 = ch * 32 + lc, j1 = j0 + 16;
          if constexpr (T0) {
            const double z20 = (j0 < NP) ? s_Y2[min(j0, NP - 1)] : 0.0;
            bool ill = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) ill |= scan_one(acc0[r], r, j0, z20);
            if (__any(ill)) ill_pairs(acc0, j0, z20);
          }
          if constexpr (T1) {
            const double z21 = (j1 < NP) ? s_Y2[min(j1, NP - 1)] : 0.0;
            bool ill = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) ill |= scan_one(acc1[r], r, j1, z21);
            if (__any(ill)) ill_pairs(acc1, j1, z21);
          }
        };
        if (mode == 1) body(std::integral_constant<int, 1>{});
        else if (mode == 0) body(std::integral_constant<int, 0>{});
        else if (mode == 2) body(std::integral_constant<int, 2>{});
        else body(std::integral_constant<int, 3>{});
        __syncthreads();
      }
    } else {
      double A11r[4], Y1r[4], s1r[4];
      bool rowok[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = rtc * 16 + lg + 4 * r;
        rowok[r] = rt_valid && (i < N);
        A11r[r] = s_A11[i];
        Y1r[r] = s_Y1[i];
        s1r[r] = s_S1[i];
      }
      // pair scan of one accumulator entry (row r of the lane, column j), all cases of mf_utils.py:348-379
      auto scan_one = [&](double A12, int r, int j, bool colok, double A22, double Y2, double s2) {
        const double d1 = fma(-A12, Y2, A22 * Y1r[r]);
        const double d2 = fma(-A12, Y1r[r], A11r[r] * Y2);
        const double pd = A11r[r] * A22;
        const double Det = fma(-A12, A12, pd);
        const double num = fma(Y2, d2, Y1r[r] * d1);
        // nearly collinear atom pairs cannot be ranked as a fraction (MFX_DET_REL bounds the score error of every pair
        // that is); with two positive weights they go to the short list unranked, as in the FAST variant (rare: a branch)
        if ((d1 > 0.0) && (d2 > 0.0) && !(Det > MFX_DET_REL * pd) && colok && rowok[r]) {
          const int slot = atomicAdd(&s_cnt[0], 1);
          if (slot < MFX_MAXC) { s_cand[slot].score = 1e300; s_cand[slot].i = rtc * 16 + lg + 4 * r; s_cand[slot].j = j; }
        }
        const bool both = (d1 > 0.0) & (d2 > 0.0) & (Det > MFX_DET_REL * pd);
        double p = both ? num : 0.0;   // single-active cases: the two best single atoms, see phase 1
        const double q = both ? Det : 1.0;
        p = (colok & rowok[r]) ? p : 0.0;
        const bool better = p * bq[r] > bp[r] * q;
        bp[r] = better ? p : bp[r];
        bq[r] = better ? q : bq[r];
        bj[r] = better ? j : bj[r];
      };
      for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = (NBUF == 2) ? (ch & 1) : 0;
        if constexpr (NBUF == 2) {
          if (ch + 1 < nchunks) gen_chunk(ch + 1, buf ^ 1);
        } else if (ch > 0) {
          gen_chunk(ch, 0);   // single buffer: generate, barrier, consume, barrier
          __syncthreads();
        }
        if (rt_valid) {
          const double* b0p = sB + (size_t)buf * (TILES * MPS * 16) + lg * 16 + lc;
          const double* b1p = b0p + (TILES == 2 ? MPS * 16 : 0);
          d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
          for (int kk = 0; kk < KSTEPS; ++kk) {
            const double b0 = b0p[kk * 64];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[kk], b0, acc0, 0, 0, 0);
            if constexpr (TILES == 2) {
              const double b1 = b1p[kk * 64];
              acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[kk], b1, acc1, 0, 0, 0);
            }
          }
#pragma unroll
          for (int t = 0; t < TILES; ++t) {
            const d4 acc = t ? acc1 : acc0;
            const int j = ch * CW + t * 16 + lc;
            const bool colok = j < N;
            const int jq = colok ? j : 0;
            const double A22 = s_A22[jq], Y2 = s_Y2[jq], s2 = s_S2[jq];
#pragma unroll
            for (int r = 0; r < 4; ++r) scan_one(acc[r], r, j, colok, A22, Y2, s2);
          }
        }
        __syncthreads();
      }
    }
    if (round == 0) MFX_STAMP(5);
    // ---- round end: short-list by interval: a candidate stays if its upper bound reaches the best
    // lower bound seen so far.  Bounds: eps_abs (formula/rounding differences between the fraction
    // form and the reference's expression) plus the conditioning-dependent error of num/Det.
    double sc[4], er[4];
    double llb = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[r] = -1.0;
      er[r] = 0.0;
      const int i = rtc * 16 + lg + 4 * r;
      if (bj[r] >= 0 && rt_valid && i < N) {
        sc[r] = bp[r] / bq[r];
        if constexpr (FAST) {
          er[r] = sc[r] * (MFX_A12_REL / bq[r]);  // bq = 1 - c^2
        } else {
          const double pd = s_A11[i] * s_A22[bj[r]];
          er[r] = (bq[r] == 1.0) ? 0.0 : sc[r] * (MFX_A12_REL * pd / bq[r]);
        }
        llb = fmax(llb, sc[r] - er[r]);
      }
    }
    llb = wave_max(llb);
    if (lane == 0) s_red[wave] = llb;
    __syncthreads();
    double rlb = s_red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) rlb = fmax(rlb, s_red[w]);
    glb_run = fmax(glb_run, rlb);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (sc[r] > 0.0 && sc[r] + er[r] + eps_abs >= glb_run) {
        const int slot = atomicAdd(&s_cnt[0], 1);
        if (slot < MFX_MAXC) {
          s_cand[slot].score = sc[r] + er[r] + eps_abs;  // upper bound
          s_cand[slot].i = rtc * 16 + lg + 4 * r;
          s_cand[slot].j = bj[r];
        }
      }
    }
    if (round == 0) MFX_STAMP(9);
    __syncthreads();
  }

  MFX_STAMP(6);
  // ---- phase 3: exact re-evaluation of the short list (reference arithmetic and order)
  // exact (res, w) of one pair: sequential sums over the measurements as mf_utils.py:307-325, then
  // the case analysis of mf_utils.py:341-379
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
  // lexicographic (res, idx) minimum over the workgroup; idx = i*N + j is the reference's scan order
  double* s_rres = (double*)sB;          // [8] per-wave partials (B buffers are idle now)
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
      // fold into the current winner (strict '<' on res, ties -> earlier pair in scan order)
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
  const int ncand = nappend > MFX_MAXC ? MFX_MAXC : nappend;
  __syncthreads();   // everyone has read s_cnt / is done with the B buffers
  if (tid == 0) {    // mf_utils.py:327, 382: start from min_obj = y_sq at pair (0,0) with w = 0, strict '<'
    s_win[0] = y_sq; s_win[1] = 0.0; s_win[2] = 0.0; ((long*)s_win)[3] = -1;
  }
  {
    double res = INFINITY, w0 = 0.0, w1 = 0.0;
    long idx = -1;
    if (nappend <= a.maxc) {
      // A scan candidate is the best pair of its (lane,row) slot - row i, the columns j = lc (mod 16).  A second pair
      // of that slot within rounding distance of the optimum was never seen by the short list (one entry per slot):
      // the whole slot row of every listed candidate is therefore evaluated exactly (<= N/16 pairs each; a slot that
      // is not listed cannot hold a near-optimal pair: its best is below the threshold).  The first entries of the
      // list are the single-atom representatives of phase 1: themselves only.
      const int NJ = (N + 15) >> 4, nsingle = s_cnt[1];
      for (int q = tid; q < ncand * NJ; q += WG) {
        const int c = q / NJ, u = q - c * NJ;
        if (!(s_cand[c].score >= glb_run)) continue;
        const int ci = s_cand[c].i, cj = s_cand[c].j;
        const int jj = (c < nsingle) ? cj : (cj & 15) + 16 * u;
        if ((c < nsingle && u > 0) || jj >= N) continue;
        double r, u0, u1;
        exact_pair(ci, jj, u0, u1, r);
        const long ix = (long)ci * N + jj;
        if (r < res || (r == res && ix < idx)) { res = r; idx = ix; w0 = u0; w1 = u1; }
      }
    } else {
      // The short list overflowed (more than MFX_MAXC slot bests within rounding distance of the optimum: massive
      // near-ties): entries were dropped, so nothing above can be trusted.  Last resort, exact by construction:
      // every pair through the reference arithmetic (milliseconds for this voxel; no test or benchmark voxel gets here).
      const long npairs = (long)N * N;
      for (long pr = tid; pr < npairs; pr += WG) {
        double r, u0, u1;
        exact_pair((int)(pr / N), (int)(pr % N), u0, u1, r);
        if (r < res || (r == res && pr < idx)) { res = r; idx = pr; w0 = u0; w1 = u1; }
      }
    }
    block_argmin(res, idx, w0, w1);
  }
  // Near-zero second weight: every pair sharing the active atom fits equally well up to rounding
  // (a single-fascicle signal fitted with two fascicles, or any voxel whose optimum has one active atom).
  // The reference then returns the first pair of that row/column attaining the minimum of its own
  // rounded residual: evaluate the whole family exactly.  Wave-uniform branch on the broadcast winner.
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
    // y_rec = A[:, tot] @ w and R^2 = corrcoef(y, y_rec)[0,1]^2 (mf.py:449-450)
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
