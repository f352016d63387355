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
