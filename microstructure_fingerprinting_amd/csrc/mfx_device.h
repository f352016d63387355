// mfx_device.h -- device-side data layout shared by the HIP kernels and the host C-ABI.
//
// HBM layout
//   tables  tab[(P+1) x ldn] of double2 {Ylo, slope}: row r = knot row (global index over all
//           shells), column n = atom.  slope[r][n] = (Y[r+1][n]-Y[r][n]) / (x[r+1]-x[r]) is the
//           per-interval slope of SciPy's interp1d._call_linear (reference call sites
//           mf_utils.py:1423, 2038, 2074), precomputed once because it is direction independent.
//           Row P is all zeros (used for padded measurement rows).  ldn = N rounded up to 16 so
//           that 16-atom MFMA tiles never need a bounds check (padded atoms are zero).
//   plan    per protocol row m: unit gradient g[m][3], shell index s_lo[m], and for rows whose
//           G lies strictly between two table shells (mf_utils.py:1827-1839) s_hi[m] >= 0 with
//           tG = G - G_lo, dG = G_hi - G_lo.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct TablesDev {
  int S, N, ldn, P;
  const double* x;     // [P]
  const int* off;      // [S+1]
  const double2* tab;  // [(P+1) x ldn]
  const float2* tab32; // [(P+1) x ldn] the same entries rounded to FP32 (pair screening only, fit_k2s.hip)
  const double* G_un;  // [S]
};

struct PlanDev {
  int M;
  const double* g;   // [M x 3]
  const int* s_lo;   // [M]
  const int* s_hi;   // [M]  (-1: exact shell)
  const double* tG;  // [M]
  const double* dG;  // [M]
  int any_bracket;
  int normalise;     // explicit (rotate_atom) plans: a fascicle direction is divided by its norm first (mf_utils.py:1262-1270)
  // Screening view of the protocol (fit_k2s.hip only ranks with it; exact arithmetic never uses it): every row maps
  // to ONE knot table.  Exact-shell rows use the table's own shells; a G-bracketed row uses a "virtual shell" built
  // once per plan: the blend (1-w) shell_lo + w shell_hi, w = (G-G_lo)/(G_hi-G_lo), which is itself piecewise linear
  // in u on the union of the two shells' knots.  Without bracketed rows tab32s aliases TablesDev::tab32 and the other three are unused.
  const float2* tab32s;  // [(Ps_rows) x ldn] FP32 {ylo, slope}; rows 0..P as TablesDev::tab32 (row P zero), then virtual rows
  const double* xs;      // knot value per row of tab32s (bracketed plans only)
  const int* offs;       // [n_shells][2] first row, knot count of the table shells followed by the virtual shells
  const int* s_scr;      // [M] shell of the row in (xs, offs)
  // deferred error channel of the asynchronous entry points (mfx_plan_status): [0] flag word, [1] a flagged voxel
  int* status;
};

#define MFX_ST_DIR_NORM 1   // a fascicle direction failed |1 - |d|| <= 1e-3 (mf_utils.py:1798-1802)

// the reference's per-voxel direction check (interp_PGSE_from_multishell, mf_utils.py:1798-1802), once per voxel and
// fascicle: flags the plan's status word instead of raising (the device cannot), the voxel is still computed
__device__ __forceinline__ void mfx_check_dir(const PlanDev& P, const double* __restrict__ pk, int vox) {
  if (P.normalise) return;   // rotate_atom normalises the direction itself and has no such check
  const double nrm = sqrt((pk[0] * pk[0] + pk[1] * pk[1]) + pk[2] * pk[2]);
  if (!(fabs(1.0 - nrm) <= 1e-3) && P.status) {
    atomicOr(P.status, MFX_ST_DIR_NORM);
    P.status[1] = vox;
  }
}

// per-(direction,row) evaluation descriptor: which knot interval, and the offset inside it
struct RowDesc {
  int r0, r1;     // knot rows (r1 = -1 when the row maps to exactly one shell)
  double t0, t1;  // u - x[r]
};

// np.searchsorted(x, v, side='left'): first index with x[idx] >= v
__device__ __forceinline__ int mfx_searchsorted_left(const double* __restrict__ x, int n, double v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (x[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// interp1d._call_linear index logic: j = clip(searchsorted(x,u), 1, P_s-1); interval (j-1, j)
__device__ __forceinline__ void mfx_shell_locate(const TablesDev& T, int s, double u, int& r, double& t) {
  const int o = T.off[s];
  const int Ps = T.off[s + 1] - o;
  int j = mfx_searchsorted_left(T.x + o, Ps, u);
  j = j < 1 ? 1 : j;
  j = j > Ps - 1 ? Ps - 1 : j;
  r = o + j - 1;
  t = u - T.x[r];
}

// |g . d| with the reference's operation order ((g0 d0 + g1 d1) + g2 d2), mf_utils.py:1810
__device__ __forceinline__ double mfx_absdot(const double* __restrict__ g, double d0, double d1, double d2) {
  return fabs((g[0] * d0 + g[1] * d1) + g[2] * d2);
}

__device__ __forceinline__ RowDesc mfx_row_desc(const TablesDev& T, const PlanDev& P, int m, double d0, double d1,
                                                double d2) {
  RowDesc rd;
  if (P.normalise) {   // rotate_atom: newdir / |newdir| (mf_utils.py:1262, 1269)
    const double nn = sqrt((d0 * d0 + d1 * d1) + d2 * d2);
    d0 /= nn; d1 /= nn; d2 /= nn;
  }
  const double u = mfx_absdot(P.g + 3 * m, d0, d1, d2);
  mfx_shell_locate(T, P.s_lo[m], u, rd.r0, rd.t0);
  rd.r1 = -1;
  rd.t1 = 0.0;
  const int sh = P.s_hi[m];
  if (sh >= 0) mfx_shell_locate(T, sh, u, rd.r1, rd.t1);
  return rd;
}

// one rotated dictionary entry, exact-shell row:  slope * (u - x_lo) + y_lo  (separate mul and
// add, as NumPy evaluates it; the translation unit is compiled with -ffp-contract=off)
__device__ __forceinline__ double mfx_eval(const double2* __restrict__ tab, int ldn, int r, double t, int n) {
  const double2 e = tab[(size_t)r * ldn + n];
  return e.y * t + e.x;
}

// bracketed row: linear interpolation in G between the two shell values, mf_utils.py:1950-1955
__device__ __forceinline__ double mfx_eval_br(const double2* __restrict__ tab, int ldn, const RowDesc& rd, double tG,
                                              double dG, int n) {
  const double v0 = mfx_eval(tab, ldn, rd.r0, rd.t0, n);
  if (rd.r1 < 0) return v0;
  const double v1 = mfx_eval(tab, ldn, rd.r1, rd.t1, n);
  const double sl = (v1 - v0) / dG;
  return sl * tG + v0;
}
