// nnls_small.h -- per-tuple solvers shared by the kernels.
//
// "exact" functions reproduce the reference's arithmetic for ONE tuple of atoms, given Gram scalars
// that were summed in the reference's order:
//   nnls1_exact   mf_utils.py:255-267   (solve_exhaustive_posweights_1, one column)
//   nnls2_exact   mf_utils.py:341-379   (= lsqnonneg_2var_opt, mf_utils.py:404-459)
//   nnls3_cramer  mf_utils.py:554-593   (solve_exhaustive_posweights_3, Cramer test + 2-var fallbacks);
//                 the all-positive branch needs the explicit residual over the M rows, which the
//                 caller supplies through a functor.
// "score" functions give the NNLS optimum of a small Gram system by feasible-subset enumeration
// (max over supports S with a positive unconstrained solution of y'A_S w_S); they are used for
// ranking only, every reported weight/residual comes from an exact function.
#pragma once
#include <hip/hip_runtime.h>

#define MFX_TOL3 (100 * 2.2204e-16)  // mf_utils.py:480-481

__device__ __forceinline__ void nnls1_exact(double y_sq, double asq, double adoty, double& w, double& res) {
  w = 0.0;
  res = y_sq;
  if (adoty >= 0) {
    w = adoty / asq;
    res -= w * adoty;
  }
}

__device__ __forceinline__ void nnls2_exact(double y_sq, double A11, double A12, double A22, double Y1, double Y2,
                                            double& w0, double& w1, double& res) {
  const double d1 = A22 * Y1 - A12 * Y2;
  const double d2 = A11 * Y2 - A12 * Y1;
  w0 = 0.0;
  w1 = 0.0;
  res = y_sq;
  if (d1 > 0.0 && d2 > 0.0) {
    const double Det = A11 * A22 - A12 * A12;
    w0 = d1 / Det;
    w1 = d2 / Det;
    res = (res + w0 * w0 * A11 + w1 * w1 * A22 + 2 * (w0 * w1 * A12 - w0 * Y1 - w1 * Y2));
  } else if (d1 >= 0.0 && d2 <= 0.0) {
    if (Y1 >= 0.0) { w0 = Y1 / A11; res = res - Y1 * w0; }
  } else if (d1 <= 0.0 && d2 >= 0.0) {
    if (Y2 >= 0.0) { w1 = Y2 / A22; res = res - Y2 * w1; }
  } else if (d1 < 0.0 && d2 < 0.0) {
    if (Y1 > 0) { w0 = Y1 / A11; res -= Y1 * w0; }
    else if (Y2 > 0) { w1 = Y2 / A22; res -= Y2 * w1; }
  }
}

// One triple of solve_exhaustive_posweights_3.  `explicit_res(w)` must return
// sum_k (w0*a1[k] + w1*a2[k] + w2*a3[k] - y[k])^2 accumulated sequentially over k (mf_utils.py:569-573).
template <typename F>
__device__ __forceinline__ void nnls3_cramer(double y_sq, double a11, double a12, double a13, double a22, double a23,
                                             double a33, double Y1, double Y2, double Y3, F explicit_res, double w[3],
                                             double& res) {
  const double D1 = (Y1 * (a22 * a33 - a23 * a23) - Y2 * (a12 * a33 - a23 * a13) + Y3 * (a12 * a23 - a22 * a13));
  const double D2 = (-Y1 * (a12 * a33 - a13 * a23) + Y2 * (a11 * a33 - a13 * a13) - Y3 * (a11 * a23 - a12 * a13));
  const double D3 = (Y1 * (a12 * a23 - a13 * a22) - Y2 * (a11 * a23 - a12 * a13) + Y3 * (a11 * a22 - a12 * a12));
  if (D1 >= -MFX_TOL3 && D2 >= -MFX_TOL3 && D3 >= -MFX_TOL3) {
    const double D = (a11 * (a22 * a33 - a23 * a23) - a12 * (a12 * a33 - a23 * a13) + a13 * (a12 * a23 - a22 * a13));
    w[0] = D1 / D;
    w[1] = D2 / D;
    w[2] = D3 / D;
    res = explicit_res(w);
  } else {
    double u0, u1, r;
    nnls2_exact(y_sq, a11, a12, a22, Y1, Y2, u0, u1, r);
    w[0] = u0; w[1] = u1; w[2] = 0.0; res = r;
    nnls2_exact(y_sq, a11, a13, a33, Y1, Y3, u0, u1, r);
    if (r < res) { w[0] = u0; w[1] = 0.0; w[2] = u1; res = r; }
    nnls2_exact(y_sq, a22, a23, a33, Y2, Y3, u0, u1, r);
    if (r < res) { w[0] = 0.0; w[1] = u0; w[2] = u1; res = r; }
  }
}

// ---------------------------------------------------------------------------------------------
// ranking scores (NNLS optimum of tiny Gram systems); return y'Aw >= 0, larger is better
__device__ __forceinline__ double score1(double a, double y) { return (y > 0.0) ? (y * y) / a : 0.0; }

__device__ __forceinline__ double score2(double a11, double a12, double a22, double y1, double y2) {
  const double d1 = a22 * y1 - a12 * y2, d2 = a11 * y2 - a12 * y1, det = a11 * a22 - a12 * a12;
  if (d1 > 0.0 && d2 > 0.0 && det > 1e-8 * (a11 * a22)) return (y1 * d1 + y2 * d2) / det;
  return fmax(score1(a11, y1), score1(a22, y2));
}

__device__ __forceinline__ double score3(double a11, double a12, double a13, double a22, double a23, double a33,
                                         double y1, double y2, double y3) {
  const double c11 = a22 * a33 - a23 * a23, c12 = a13 * a23 - a12 * a33, c13 = a12 * a23 - a13 * a22;
  const double c22 = a11 * a33 - a13 * a13, c23 = a12 * a13 - a11 * a23, c33 = a11 * a22 - a12 * a12;
  const double det = a11 * c11 + a12 * c12 + a13 * c13;
  const double D1 = y1 * c11 + y2 * c12 + y3 * c13;
  const double D2 = y1 * c12 + y2 * c22 + y3 * c23;
  const double D3 = y1 * c13 + y2 * c23 + y3 * c33;
  if (det > 1e-12 * (a11 * a22 * a33) && D1 >= 0.0 && D2 >= 0.0 && D3 >= 0.0) return (y1 * D1 + y2 * D2 + y3 * D3) / det;
  return fmax(score2(a11, a12, a22, y1, y2), fmax(score2(a11, a13, a33, y1, y3), score2(a22, a23, a33, y2, y3)));
}

// 4x4 SPD solve by elimination; returns false if a pivot is (numerically) non-positive
__device__ __forceinline__ bool solve4_spd(const double g[10] /* a11 a12 a13 a14 a22 a23 a24 a33 a34 a44 */,
                                           const double y[4], double w[4]) {
  const double a11 = g[0], a12 = g[1], a13 = g[2], a14 = g[3], a22 = g[4], a23 = g[5], a24 = g[6], a33 = g[7],
               a34 = g[8], a44 = g[9];
  if (!(a11 > 0.0)) return false;
  const double i1 = 1.0 / a11;
  const double l21 = a12 * i1, l31 = a13 * i1, l41 = a14 * i1;
  const double b22 = a22 - l21 * a12, b23 = a23 - l21 * a13, b24 = a24 - l21 * a14;
  const double b33 = a33 - l31 * a13, b34 = a34 - l31 * a14, b44 = a44 - l41 * a14;
  const double y2 = y[1] - l21 * y[0], y3 = y[2] - l31 * y[0], y4 = y[3] - l41 * y[0];
  if (!(b22 > 1e-10 * a22)) return false;
  const double i2 = 1.0 / b22;
  const double l32 = b23 * i2, l42 = b24 * i2;
  const double c33 = b33 - l32 * b23, c34 = b34 - l32 * b24, c44 = b44 - l42 * b24;
  const double z3 = y3 - l32 * y2, z4 = y4 - l42 * y2;
  if (!(c33 > 1e-10 * a33)) return false;
  const double l43 = c34 / c33;
  const double d44 = c44 - l43 * c34;
  const double v4 = z4 - l43 * z3;
  if (!(d44 > 1e-10 * a44)) return false;
  w[3] = v4 / d44;
  w[2] = (z3 - c34 * w[3]) / c33;
  w[1] = (y2 - b23 * w[2] - b24 * w[3]) * i2;
  w[0] = (y[0] - a12 * w[1] - a13 * w[2] - a14 * w[3]) * i1;
  return true;
}

// NNLS optimum (score = y'Aw) of a 4-column system given its Gram; ranking only
__device__ __forceinline__ double score4(const double g[10], const double y[4]) {
  double w[4];
  if (solve4_spd(g, y, w) && w[0] >= 0.0 && w[1] >= 0.0 && w[2] >= 0.0 && w[3] >= 0.0)
    return w[0] * y[0] + w[1] * y[1] + w[2] * y[2] + w[3] * y[3];
  // support has at most 3 columns: best of the four 3-subsets (each falls back to pairs/singles itself)
  const double s123 = score3(g[0], g[1], g[2], g[4], g[5], g[7], y[0], y[1], y[2]);
  const double s124 = score3(g[0], g[1], g[3], g[4], g[6], g[9], y[0], y[1], y[3]);
  const double s134 = score3(g[0], g[2], g[3], g[7], g[8], g[9], y[0], y[2], y[3]);
  const double s234 = score3(g[4], g[5], g[6], g[7], g[8], g[9], y[1], y[2], y[3]);
  return fmax(fmax(s123, s124), fmax(s134, s234));
}

// Exact-stage NNLS for n <= 4 columns from a sequentially summed Gram: the support with the largest
// y'A_S w_S among supports whose unconstrained solution is non-negative (= the NNLS optimum that
// scipy.optimize.nnls reaches at mf_utils.py:640).  g: upper triangle row-major (n(n+1)/2), w out.
__device__ inline void nnls_gram_subsets(int n, const double* g, const double* y, double* w) {
  auto G = [&](int p, int q) { if (p > q) { int t = p; p = q; q = t; } return g[p * n - p * (p - 1) / 2 + (q - p)]; };
  double best = 0.0;
  for (int k = 0; k < n; ++k) w[k] = 0.0;
  for (int mask = 1; mask < (1 << n); ++mask) {
    int idx[4], c = 0;
    for (int k = 0; k < n; ++k) if (mask & (1 << k)) idx[c++] = k;
    // eliminate in an order that does not depend on where a column sits in the tuple (largest A'y first, as the
    // Lawson-Hanson iteration admits them): two tuples holding the SAME columns in swapped positions (identical peak
    // directions) then get bit-identical weights and residuals, an exact tie as in the reference, and its first hit wins
    for (int p = 1; p < c; ++p) {
      const int v = idx[p];
      int q = p - 1;
      while (q >= 0 && (y[idx[q]] < y[v] || (y[idx[q]] == y[v] && G(idx[q], idx[q]) > G(v, v)))) { idx[q + 1] = idx[q]; --q; }
      idx[q + 1] = v;
    }
    double A[4][5];
    for (int p = 0; p < c; ++p) { for (int q = 0; q < c; ++q) A[p][q] = G(idx[p], idx[q]); A[p][4] = y[idx[p]]; }
    bool ok = true;
    for (int p = 0; p < c && ok; ++p) {  // Gaussian elimination (SPD: no pivoting)
      if (!(A[p][p] > 1e-13 * G(idx[p], idx[p]))) { ok = false; break; }
      for (int r = p + 1; r < c; ++r) {
        const double f = A[r][p] / A[p][p];
        for (int q = p; q < c; ++q) A[r][q] -= f * A[p][q];
        A[r][4] -= f * A[p][4];
      }
    }
    if (!ok) continue;
    double ws[4];
    for (int p = c - 1; p >= 0; --p) {
      double t = A[p][4];
      for (int q = p + 1; q < c; ++q) t -= A[p][q] * ws[q];
      ws[p] = t / A[p][p];
    }
    double sc = 0.0;
    for (int p = 0; p < c; ++p) { if (!(ws[p] >= 0.0)) ok = false; sc += ws[p] * y[idx[p]]; }
    if (ok && sc > best) {
      best = sc;
      for (int k = 0; k < n; ++k) w[k] = 0.0;
      for (int p = 0; p < c; ++p) w[idx[p]] = ws[p];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// building blocks of the feasible-support ranking used by fit_k2x.hip: score of ONE support if its
// unconstrained least-squares solution is non-negative, else 0 (the NNLS optimum of a tuple is the
// maximum of these over all supports)
__device__ __forceinline__ double mfx_rcp(double x) {  // 1/x to ~1 ulp without the IEEE division sequence
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double pos1(double a, double y) { return (y > 0.0 && a > 0.0) ? (y * y) / a : 0.0; }
__device__ __forceinline__ double pos2(double a11, double a12, double a22, double y1, double y2) {
  const double d1 = a22 * y1 - a12 * y2, d2 = a11 * y2 - a12 * y1, det = a11 * a22 - a12 * a12;
  return (d1 >= 0.0 && d2 >= 0.0 && det > 1e-8 * (a11 * a22)) ? (y1 * d1 + y2 * d2) / det : 0.0;
}
__device__ __forceinline__ double pos3(double a11, double a12, double a13, double a22, double a23, double a33,
                                       double y1, double y2, double y3) {
  const double c11 = a22 * a33 - a23 * a23, c12 = a13 * a23 - a12 * a33, c13 = a12 * a23 - a13 * a22;
  const double c22 = a11 * a33 - a13 * a13, c23 = a12 * a13 - a11 * a23, c33 = a11 * a22 - a12 * a12;
  const double det = a11 * c11 + a12 * c12 + a13 * c13;
  const double D1 = y1 * c11 + y2 * c12 + y3 * c13;
  const double D2 = y1 * c12 + y2 * c22 + y3 * c23;
  const double D3 = y1 * c13 + y2 * c23 + y3 * c33;
  return (det > 1e-12 * (a11 * a22 * a33) && D1 >= 0.0 && D2 >= 0.0 && D3 >= 0.0) ? (y1 * D1 + y2 * D2 + y3 * D3) / det : 0.0;
}
