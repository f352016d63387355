/*
 * mf_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle / CPU baseline).
 *
 * Plain-C restatement of the reference's per-voxel hot path, written from the
 * reference's algorithm description, loop order and branch structure so that it
 * is bit-comparable with the reference's Python/Numba code.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file.
 * The product (microstructure_fingerprinting_amd) never imports or links it.
 *
 * Pinned against the reference: tests/test_oracle_golden.py checks every function
 * here against tests/golden/ (.npz files), which tests/golden/gen_golden.py produced by
 * running the reference's own source in the build container.
 *
 * Citations: "mfu" = /root/reference/microstructure_fingerprinting/mf_utils.py,
 *            "mf"  = /root/reference/microstructure_fingerprinting/mf.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 *        (no FMA contraction: the reference's CPython/Numba-without-fastmath
 *        arithmetic rounds every product and sum separately).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_ERR_G_RANGE 1   /* mfu:1829-1836 ValueError: extrapolation in G not supported */
#define ORC_ERR_ARG 2
#define ORC_ERR_NNLS_ITER 3 /* scipy.optimize.nnls RuntimeError: max iterations */

/* ------------------------------------------------------------------------- */
/* 2-variable NNLS from precomputed scalars: mfu:404-459 (lsqnonneg_2var_opt),
 * same case analysis as the inlined copy in mfu:341-379.                      */
static void nnls2_scalars(double y_sq, double A11, double A12, double A22, double Y1, double Y2,
                          double w[2], double* res_out) {
  double A21 = A12;
  double w1d = A22 * Y1 - A12 * Y2; /* mfu:425 */
  double w2d = A11 * Y2 - A21 * Y1; /* mfu:426 */
  double resnorm = y_sq;
  w[0] = 0.0;
  w[1] = 0.0;
  if (w1d > 0.0 && w2d > 0.0) { /* mfu:432-439 */
    double Det = A11 * A22 - A21 * A12;
    w[0] = w1d / Det;
    w[1] = w2d / Det;
    resnorm = (resnorm + w[0] * w[0] * A11 + w[1] * w[1] * A22 +
               2 * (w[0] * w[1] * A12 - w[0] * Y1 - w[1] * Y2));
  } else if (w1d >= 0.0 && w2d <= 0.0) { /* mfu:440-444 */
    if (Y1 >= 0.0) {
      w[0] = Y1 / A11;
      resnorm = resnorm - Y1 * w[0];
    }
  } else if (w1d <= 0.0 && w2d >= 0.0) { /* mfu:445-449 */
    if (Y2 >= 0.0) {
      w[1] = Y2 / A22;
      resnorm = resnorm - Y2 * w[1];
    }
  } else if (w1d < 0.0 && w2d < 0.0) { /* mfu:450-458 */
    if (Y1 > 0) {
      w[0] = Y1 / A11;
      resnorm -= Y1 * w[0];
    } else if (Y2 > 0) {
      w[1] = Y2 / A22;
      resnorm -= Y2 * w[1];
    }
  }
  *res_out = resnorm;
}

/* np.sum(y**2): NumPy pairwise summation (mfu:248, mfu:630).  For n < 8 a plain
 * loop; otherwise 8 accumulators over blocks of <=128, recursive halving above. */
static double np_pairwise_sumsq(const double* a, long n) {
  if (n < 8) {
    double res = 0.;
    for (long i = 0; i < n; i++) res += a[i] * a[i];
    return res;
  } else if (n <= 128) {
    double r[8];
    long i;
    for (i = 0; i < 8; i++) r[i] = a[i] * a[i];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; j++) r[j] += a[i + j] * a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i] * a[i];
    return res;
  } else {
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sumsq(a, n2) + np_pairwise_sumsq(a + n2, n - n2);
  }
}

/* ------------------------------------------------------------------------- */
/* Per-thread scratch for the Gram arrays of solve2 / solve3 (4.9 MB per config-2 voxel): grown once per thread and kept
 * across the voxels of a batch, released when the thread leaves orc_fit_batch / orc_solve_exhaustive.  A calloc + free
 * per voxel is an mmap + munmap of fresh pages per voxel; with one worker per host core they serialise in the kernel
 * (the reference's process pool, mf.py:978-1009, lets Numba allocate per call too, but its workers do not share an
 * address space).  What needs zeros is cleared explicitly below. */
static __thread double* tl_scratch = NULL;
static __thread size_t tl_scratch_cap = 0;
static double* scratch_get(size_t n) {
  if (n > tl_scratch_cap) {
    free(tl_scratch);
    tl_scratch = (double*)malloc(n * sizeof(double));
    tl_scratch_cap = tl_scratch ? n : 0;
  }
  return tl_scratch;
}
static void scratch_release(void) {
  free(tl_scratch);
  tl_scratch = NULL;
  tl_scratch_cap = 0;
}

/* ------------------------------------------------------------------------- */
/* solve_exhaustive_posweights_1: mfu:225-278 */
static void solve1(const double* A, long lda, int M, long N, const double* y, double* w_out, long* sub,
                   double* min_obj_out) {
  double w_nneg = 0.0;
  long ind = 0;
  double y_sq = np_pairwise_sumsq(y, M); /* mfu:248 */
  double min_obj = y_sq;
  for (long i1 = 0; i1 < N; i1++) {
    double adoty = 0.0, w = 0.0, resnorm = y_sq;
    for (int i = 0; i < M; i++) adoty = adoty + A[i * lda + i1] * y[i]; /* mfu:258-259 */
    if (adoty >= 0) {
      double asq = 0.0;
      for (int i = 0; i < M; i++) asq = asq + A[i * lda + i1] * A[i * lda + i1];
      w = adoty / asq;
      resnorm -= w * adoty; /* mfu:267 */
    }
    if (resnorm < min_obj) { /* mfu:270 strict */
      ind = i1;
      min_obj = resnorm;
      w_nneg = w;
    }
  }
  *w_out = w_nneg;
  *sub = ind;
  *min_obj_out = min_obj;
}

/* solve_exhaustive_posweights_2: mfu:288-392 */
static int solve2(const double* A, long lda, int M, long N1, long N2, const double* y, double* w_out, long* sub,
                  double* min_obj_out) {
  double* A11 = scratch_get((size_t)N1 + N2 + (size_t)N1 * N2 + N1 + N2);
  if (!A11) return ORC_ERR_ARG;
  double* A22 = A11 + N1;
  double* Ady = A22 + N2;
  double* A12 = Ady + N1 + N2;                                   /* every entry is assigned below */
  memset(A11, 0, sizeof(double) * (size_t)(2 * (N1 + N2)));      /* A11, A22, Ady: np.zeros, mfu:300-304 */
  for (long i1 = 0; i1 < N1; i1++) /* mfu:307-310 */
    for (int k = 0; k < M; k++) A11[i1] += A[k * lda + i1] * A[k * lda + i1];
  for (long i2 = 0; i2 < N2; i2++) /* mfu:311-314 */
    for (int k = 0; k < M; k++) A22[i2] += A[k * lda + N1 + i2] * A[k * lda + N1 + i2];
  for (long i1 = 0; i1 < N1; i1++) /* mfu:315-319 */
    for (long i2 = 0; i2 < N2; i2++) {
      double s = 0.0;
      for (int k = 0; k < M; k++) s += A[k * lda + i1] * A[k * lda + N1 + i2];
      A12[i1 * N2 + i2] = s;
    }
  double y_sq = 0.0;
  for (int k = 0; k < M; k++) { /* mfu:320-325 */
    y_sq += y[k] * y[k];
    for (long i = 0; i < N1 + N2; i++) Ady[i] += y[k] * A[k * lda + i];
  }
  double min_obj = y_sq;
  double wb[2] = {0, 0};
  long s0 = 0, s1 = 0;
  for (long i1 = 0; i1 < N1; i1++) /* mfu:329-386 */
    for (long i2 = 0; i2 < N2; i2++) {
      double w[2], res;
      nnls2_scalars(y_sq, A11[i1], A12[i1 * N2 + i2], A22[i2], Ady[i1], Ady[N1 + i2], w, &res);
      if (res < min_obj) {
        s0 = i1;
        s1 = i2;
        min_obj = res;
        wb[0] = w[0];
        wb[1] = w[1];
      }
    }
  w_out[0] = wb[0];
  w_out[1] = wb[1];
  sub[0] = s0;
  sub[1] = s1;
  *min_obj_out = min_obj;
  return ORC_OK;
}

/* solve_exhaustive_posweights_3: mfu:470-607 */
static int solve3(const double* A, long lda, int M, long N1, long N2, long N3, const double* y, double* w_out,
                  long* sub, double* min_obj_out) {
  const long s_ind[3] = {0, N1, N1 + N2};
  const double eps = 2.2204e-16; /* mfu:480 */
  const double tol = 100 * eps;
  const size_t nsum = (size_t)N1 + N2 + N3;
  double* A11 = scratch_get(2 * nsum + (size_t)N1 * N2 + (size_t)N1 * N3 + (size_t)N2 * N3);
  if (!A11) return ORC_ERR_ARG;
  double* A22 = A11 + N1;
  double* A33 = A22 + N2;
  double* Ady = A33 + N3;
  double* A12 = Ady + nsum;                                      /* the three cross-Grams are assigned entry by entry */
  double* A13 = A12 + (size_t)N1 * N2;
  double* A23 = A13 + (size_t)N1 * N3;
  memset(A11, 0, sizeof(double) * 2 * nsum);                     /* norms and Ady: np.zeros, mfu:488-497 */
#define COL(k, c) A[(k) * lda + (c)]
  for (long i = 0; i < N1; i++)
    for (int k = 0; k < M; k++) A11[i] += COL(k, s_ind[0] + i) * COL(k, s_ind[0] + i);
  for (long i = 0; i < N2; i++)
    for (int k = 0; k < M; k++) A22[i] += COL(k, s_ind[1] + i) * COL(k, s_ind[1] + i);
  for (long i = 0; i < N3; i++)
    for (int k = 0; k < M; k++) A33[i] += COL(k, s_ind[2] + i) * COL(k, s_ind[2] + i);
  for (long i1 = 0; i1 < N1; i1++)
    for (long i2 = 0; i2 < N2; i2++) {
      double s = 0.0;
      for (int k = 0; k < M; k++) s += COL(k, s_ind[0] + i1) * COL(k, s_ind[1] + i2);
      A12[i1 * N2 + i2] = s;
    }
  for (long i3 = 0; i3 < N3; i3++)
    for (long i1 = 0; i1 < N1; i1++) {
      double s = 0.0;
      for (int k = 0; k < M; k++) s += COL(k, s_ind[0] + i1) * COL(k, s_ind[2] + i3);
      A13[i1 * N3 + i3] = s;
    }
  for (long i3 = 0; i3 < N3; i3++)
    for (long i2 = 0; i2 < N2; i2++) {
      double s = 0.0;
      for (int k = 0; k < M; k++) s += COL(k, s_ind[1] + i2) * COL(k, s_ind[2] + i3);
      A23[i2 * N3 + i3] = s;
    }
  double y_sq = 0.0;
  for (int k = 0; k < M; k++) {
    y_sq += y[k] * y[k];
    for (long i = 0; i < N1 + N2 + N3; i++) Ady[i] += y[k] * COL(k, i);
  }
  double min_obj = y_sq;
  double wb[3] = {0, 0, 0};
  long sb[3] = {0, 0, 0};
  for (long i3 = 0; i3 < N3; i3++) { /* mfu:540: i3 outermost */
    double a33 = A33[i3], Y3 = Ady[s_ind[2] + i3];
    for (long i1 = 0; i1 < N1; i1++) {
      double a11 = A11[i1], a13 = A13[i1 * N3 + i3], Y1 = Ady[s_ind[0] + i1];
      for (long i2 = 0; i2 < N2; i2++) {
        double a12 = A12[i1 * N2 + i2], a22 = A22[i2], a23 = A23[i2 * N3 + i3], Y2 = Ady[s_ind[1] + i2];
        double D1 = (Y1 * (a22 * a33 - a23 * a23) - Y2 * (a12 * a33 - a23 * a13) + Y3 * (a12 * a23 - a22 * a13));
        double D2 = (-Y1 * (a12 * a33 - a13 * a23) + Y2 * (a11 * a33 - a13 * a13) - Y3 * (a11 * a23 - a12 * a13));
        double D3 = (Y1 * (a12 * a23 - a13 * a22) - Y2 * (a11 * a23 - a12 * a13) + Y3 * (a11 * a22 - a12 * a12));
        double w[3], res;
        if (D1 >= -tol && D2 >= -tol && D3 >= -tol) { /* mfu:562-573 */
          double D = (a11 * (a22 * a33 - a23 * a23) - a12 * (a12 * a33 - a23 * a13) + a13 * (a12 * a23 - a22 * a13));
          w[0] = D1 / D;
          w[1] = D2 / D;
          w[2] = D3 / D;
          res = 0.0;
          for (int k = 0; k < M; k++) {
            double t = (w[0] * COL(k, s_ind[0] + i1) + w[1] * COL(k, s_ind[1] + i2) + w[2] * COL(k, s_ind[2] + i3) - y[k]);
            res += t * t;
          }
        } else { /* mfu:574-593 */
          double w2[2], r2;
          nnls2_scalars(y_sq, a11, a12, a22, Y1, Y2, w2, &r2);
          w[0] = w2[0]; w[1] = w2[1]; w[2] = 0; res = r2;
          nnls2_scalars(y_sq, a11, a13, a33, Y1, Y3, w2, &r2);
          if (r2 < res) { w[0] = w2[0]; w[1] = 0; w[2] = w2[1]; res = r2; }
          nnls2_scalars(y_sq, a22, a23, a33, Y2, Y3, w2, &r2);
          if (r2 < res) { w[0] = 0; w[1] = w2[0]; w[2] = w2[1]; res = r2; }
        }
        if (res < min_obj) { /* mfu:596 */
          sb[0] = i1; sb[1] = i2; sb[2] = i3;
          min_obj = res;
          wb[0] = w[0]; wb[1] = w[1]; wb[2] = w[2];
        }
      }
    }
  }
#undef COL
  for (int i = 0; i < 3; i++) { w_out[i] = wb[i]; sub[i] = sb[i]; }
  *min_obj_out = min_obj;
  return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* Lawson-Hanson NNLS: the algorithm behind scipy.optimize.nnls (call site mfu:640).  Third-party dependency of the
 * reference, unpinned there (requirements.txt); this container has SciPy 1.15.3, whose `_cython_nnls._nnls` is a
 * port of the authors' published FORTRAN 77 routine NNLS (Lawson & Hanson, "Solving Least Squares Problems", 1974,
 * SIAM 1995, appendix C; netlib lawson-hanson/all) with its helpers H12 (Householder), G1/G2 (Givens).  This is a
 * restatement of that published routine, statement for statement where the arithmetic is concerned: the candidate
 * column is tested for sufficient independence (`unorm + |a|*0.01 > unorm`) and for a positive trial coefficient
 * before it enters the passive set (that is what keeps duplicate columns and zero-residual problems from cycling),
 * `itmax = 3 n` counts the inner loop, and rnorm is the norm of the TRANSFORMED right-hand side below the passive
 * block - not an explicit ||A x - b||.  Pinned by tests/golden/nnls_cases.npz (SciPy 1.15.3 outputs on random,
 * duplicate-column, zero-residual and all-negative problems, generated by tests/golden/gen_golden.py --only nnls)
 * and by solver_cases.npz k4_*, k5_*.  Where a sum's order is BLAS-internal in SciPy the results agree to rounding
 * (1e-12), not bit for bit: the reference's own pick among rounding-level ties of _4up is not reproducible either.  */
#define NNLS_MAXN 16
/* H12: construct (mode 1) / apply (mode 2) the Householder transformation Q = I + u u'/b; 0-based lpivot < l1 <= m.
 * u has stride 1; c: ncv vectors, element stride ice, vector stride icv. */
static void h12(int mode, int lpivot, int l1, int m, double* u, double* up, double* c, int ice, int icv, int ncv) {
  if (lpivot < 0 || lpivot >= l1 || l1 > m) return;
  double cl = fabs(u[lpivot]);
  if (mode != 2) {
    for (int j = l1; j < m; j++) cl = fmax(fabs(u[j]), cl);
    if (cl <= 0) return;
    const double clinv = 1.0 / cl;
    double sm = (u[lpivot] * clinv) * (u[lpivot] * clinv);
    for (int j = l1; j < m; j++) sm += (u[j] * clinv) * (u[j] * clinv);
    cl = cl * sqrt(sm);
    if (u[lpivot] > 0) cl = -cl;
    *up = u[lpivot] - cl;
    u[lpivot] = cl;
  } else if (cl <= 0) {
    return;
  }
  if (ncv <= 0) return;
  double bb = (*up) * u[lpivot];
  if (bb >= 0) return;
  bb = 1.0 / bb;
  for (int j = 0; j < ncv; j++) {
    double* cj = c + (long)j * icv;
    double sm = cj[(long)lpivot * ice] * (*up);
    for (int i = l1; i < m; i++) sm += cj[(long)i * ice] * u[i];
    if (sm != 0) {
      sm *= bb;
      cj[(long)lpivot * ice] += sm * (*up);
      for (int i = l1; i < m; i++) cj[(long)i * ice] += sm * u[i];
    }
  }
}
static void g1(double a, double b, double* cterm, double* sterm, double* sig) {
  if (fabs(a) > fabs(b)) {
    const double xr = b / a, yr = sqrt(1.0 + xr * xr);
    *cterm = copysign(1.0 / yr, a);
    *sterm = (*cterm) * xr;
    *sig = fabs(a) * yr;
  } else if (b != 0) {
    const double xr = a / b, yr = sqrt(1.0 + xr * xr);
    *sterm = copysign(1.0 / yr, b);
    *cterm = (*sterm) * xr;
    *sig = fabs(b) * yr;
  } else {
    *sig = 0; *cterm = 0; *sterm = 1;
  }
}

static int nnls_lh(const double* As /* M x n row-major */, int M, int n, const double* b_in, double* x, double* rnorm,
                   double* work /* M*(n+2) */) {
  const int m = M;
  double* A = work;                       /* column-major m x n working copy */
  double* b = work + (size_t)m * n;       /* transformed right-hand side */
  double* zz = b + m;
  double w[NNLS_MAXN];
  int index[NNLS_MAXN];
  for (int j = 0; j < n; j++)
    for (int i = 0; i < m; i++) A[(size_t)j * m + i] = As[(size_t)i * n + j];
  for (int i = 0; i < m; i++) b[i] = b_in[i];
  const double factor = 0.01;
  const int itmax = 3 * n;
  int iter = 0, mode = 1;
  for (int i = 0; i < n; i++) { x[i] = 0; index[i] = i; w[i] = 0; }
  int iz2 = n - 1, iz1 = 0, nsetp = 0, npp1 = 0;   /* npp1: 0-based row of the next pivot */
  double up = 0;
  for (;;) {   /* main loop */
    if (iz1 > iz2 || nsetp >= m) break;
    for (int iz = iz1; iz <= iz2; iz++) {   /* dual (negative gradient) vector on set Z */
      const int j = index[iz];
      double sm = 0;
      for (int l = npp1; l < m; l++) sm += A[(size_t)j * m + l] * b[l];
      w[j] = sm;
    }
    int j = -1, iz = -1, accepted = 0;
    for (;;) {
      double wmax = 0;
      int izmax = -1;
      for (int q = iz1; q <= iz2; q++) {
        const int jq = index[q];
        if (w[jq] > wmax) { wmax = w[jq]; izmax = q; }
      }
      if (wmax <= 0) break;   /* Kuhn-Tucker conditions hold */
      iz = izmax;
      j = index[iz];
      double* Aj = A + (size_t)j * m;
      const double asave = Aj[npp1];
      h12(1, npp1, npp1 + 1, m, Aj, &up, NULL, 1, 1, 0);
      double unorm = 0;
      for (int l = 0; l < nsetp; l++) unorm += Aj[l] * Aj[l];
      unorm = sqrt(unorm);
      volatile double t1 = unorm + fabs(Aj[npp1]) * factor;   /* DIFF(): the comparison must not be optimised away */
      if (t1 - unorm > 0) {
        for (int l = 0; l < m; l++) zz[l] = b[l];
        h12(2, npp1, npp1 + 1, m, Aj, &up, zz, 1, 1, 1);
        const double ztest = zz[npp1] / Aj[npp1];
        if (ztest > 0) { accepted = 1; break; }
      }
      Aj[npp1] = asave;   /* reject j as a candidate */
      w[j] = 0;
    }
    if (!accepted) break;
    /* move j from set Z to set P */
    for (int l = 0; l < m; l++) b[l] = zz[l];
    index[iz] = index[iz1];
    index[iz1] = j;
    iz1++;
    nsetp = npp1 + 1;
    npp1++;
    {
      double* Aj = A + (size_t)j * m;
      for (int jz = iz1; jz <= iz2; jz++) {
        const int jj = index[jz];
        h12(2, nsetp - 1, npp1, m, Aj, &up, A + (size_t)jj * m, 1, m, 1);
      }
      if (nsetp != m) for (int l = npp1; l < m; l++) Aj[l] = 0;
    }
    w[j] = 0;
#define NNLS_SOLVE_TRI()                                                              \
    do {                                                                              \
      int jj_ = 0;                                                                    \
      for (int l = 0; l < nsetp; l++) {                                               \
        const int ip = nsetp - 1 - l;                                                 \
        if (l != 0) for (int ii = 0; ii <= ip; ii++) zz[ii] -= A[(size_t)jj_ * m + ii] * zz[ip + 1]; \
        jj_ = index[ip];                                                              \
        zz[ip] /= A[(size_t)jj_ * m + ip];                                            \
      }                                                                               \
    } while (0)
    NNLS_SOLVE_TRI();
    int fail = 0;
    for (;;) {   /* secondary loop */
      if (++iter > itmax) { mode = 3; fail = 1; break; }
      double alpha = 2.0;
      int jj = -1;
      for (int ip = 0; ip < nsetp; ip++) {
        const int l = index[ip];
        if (zz[ip] <= 0) {
          const double t = -x[l] / (zz[ip] - x[l]);
          if (alpha > t) { alpha = t; jj = ip; }
        }
      }
      if (alpha == 2.0) break;   /* all trial coefficients feasible */
      for (int ip = 0; ip < nsetp; ip++) {
        const int l = index[ip];
        x[l] += alpha * (zz[ip] - x[l]);
      }
      int i = index[jj];
      for (;;) {   /* move coefficient i from set P to set Z */
        x[i] = 0;
        if (jj != nsetp - 1) {
          for (int jq = jj + 1; jq < nsetp; jq++) {
            const int ii = index[jq];
            index[jq - 1] = ii;
            double cc, ss, sig;
            g1(A[(size_t)ii * m + jq - 1], A[(size_t)ii * m + jq], &cc, &ss, &sig);
            A[(size_t)ii * m + jq - 1] = sig;
            A[(size_t)ii * m + jq] = 0;
            for (int l = 0; l < n; l++) {
              if (l != ii) {
                const double t = A[(size_t)l * m + jq - 1];
                A[(size_t)l * m + jq - 1] = cc * t + ss * A[(size_t)l * m + jq];
                A[(size_t)l * m + jq] = -ss * t + cc * A[(size_t)l * m + jq];
              }
            }
            const double t = b[jq - 1];
            b[jq - 1] = cc * t + ss * b[jq];
            b[jq] = -ss * t + cc * b[jq];
          }
        }
        npp1 = nsetp - 1;
        nsetp--;
        iz1--;
        index[iz1] = i;
        /* the remaining coefficients in set P should be feasible; any that are not (round-off) go to set Z too */
        int again = 0;
        for (int q = 0; q < nsetp; q++) {
          i = index[q];
          if (x[i] <= 0) { jj = q; again = 1; break; }
        }
        if (!again) break;
      }
      for (int l = 0; l < m; l++) zz[l] = b[l];
      NNLS_SOLVE_TRI();
    }
    if (fail) break;
    for (int ip = 0; ip < nsetp; ip++) x[index[ip]] = zz[ip];
  }
  double sm = 0;
  if (npp1 < m) for (int i = npp1; i < m; i++) sm += b[i] * b[i];
  *rnorm = sqrt(sm);
  return mode == 3 ? ORC_ERR_NNLS_ITER : ORC_OK;
}

/* solve_exhaustive_posweights_4up: mfu:612-657 (itertools.product order = last index fastest) */
static int solve4up(const double* A, long lda, int M, const long* sizes, int Kp, const double* y, double* w_out,
                    long* sub, double* min_obj_out) {
  if (Kp > NNLS_MAXN) return ORC_ERR_ARG;
  long st[NNLS_MAXN], idx[NNLS_MAXN];
  long acc = 0;
  for (int k = 0; k < Kp; k++) { st[k] = acc; acc += sizes[k]; idx[k] = 0; }
  double* As = (double*)malloc(sizeof(double) * M * Kp);
  double* work = (double*)malloc(sizeof(double) * M * (Kp + 2));
  double y_sq = np_pairwise_sumsq(y, M); /* mfu:630 */
  double min_obj = y_sq;
  double wb[NNLS_MAXN];
  long sb[NNLS_MAXN];
  for (int k = 0; k < Kp; k++) { wb[k] = 0; sb[k] = 0; }
  int rc = ORC_OK;
  for (;;) {
    for (int i = 0; i < M; i++)
      for (int k = 0; k < Kp; k++) As[i * Kp + k] = A[i * lda + st[k] + idx[k]];
    double x[NNLS_MAXN], rn;
    rc = nnls_lh(As, M, Kp, y, x, &rn, work);
    if (rc) break;
    double obj = rn * rn; /* mfu:641 */
    if (obj < min_obj) {
      min_obj = obj;
      for (int k = 0; k < Kp; k++) { wb[k] = x[k]; sb[k] = idx[k]; }
    }
    int k = Kp - 1;
    while (k >= 0) {
      if (++idx[k] < sizes[k]) break;
      idx[k] = 0;
      k--;
    }
    if (k < 0) break;
  }
  for (int k = 0; k < Kp; k++) { w_out[k] = wb[k]; sub[k] = sb[k]; }
  *min_obj_out = min_obj;
  free(As); free(work);
  return rc;
}

/* scipy.optimize.nnls(A, b) on its own (A row-major M x n, n <= 16): the unit the goldens of nnls_cases.npz pin */
int orc_nnls(const double* A, int M, int n, const double* b, double* x, double* rnorm) {
  if (n > NNLS_MAXN || n < 1 || M < 1) return ORC_ERR_ARG;
  double* work = (double*)malloc(sizeof(double) * (size_t)M * (n + 2));
  if (!work) return ORC_ERR_ARG;
  const int rc = nnls_lh(A, M, n, b, x, rnorm, work);
  free(work);
  return rc;
}

/* solve_exhaustive_posweights: mfu:115-214 (dispatch; validation lives in the Python wrapper) */
static int solve_exhaustive(const double* A, long lda, int M, const long* sizes, int Kp, const double* y, double* w,
                            long* sub, long* tot, double* min_obj, double* y_rec) {
  int rc = ORC_OK;
  if (Kp == 1) solve1(A, lda, M, sizes[0], y, w, sub, min_obj);
  else if (Kp == 2) rc = solve2(A, lda, M, sizes[0], sizes[1], y, w, sub, min_obj);
  else if (Kp == 3) rc = solve3(A, lda, M, sizes[0], sizes[1], sizes[2], y, w, sub, min_obj);
  else rc = solve4up(A, lda, M, sizes, Kp, y, w, sub, min_obj);
  if (rc) return rc;
  long acc = 0;
  for (int k = 0; k < Kp; k++) { tot[k] = acc + sub[k]; acc += sizes[k]; }
  /* y_recons = A[:, tot] @ w  (mfu:277, 391, 606, 655) */
  for (int i = 0; i < M; i++) {
    double t = 0.0;
    for (int k = 0; k < Kp; k++) t += A[i * lda + tot[k]] * w[k];
    y_rec[i] = (Kp == 1) ? w[0] * A[i * lda + tot[0]] : t;
  }
  return ORC_OK;
}
int orc_solve_exhaustive(const double* A, long lda, int M, const long* sizes, int Kp, const double* y, double* w,
                         long* sub, long* tot, double* min_obj, double* y_rec) {
  const int rc = solve_exhaustive(A, lda, M, sizes, Kp, y, w, sub, tot, min_obj, y_rec);
  scratch_release();   /* a single problem: nothing to keep the Gram arrays for */
  return rc;
}

/* ------------------------------------------------------------------------- */
/* Rotation: interp_PGSE_from_multishell fast mode, mfu:1810-1955, on a flat
 * table: S shells, sorted G_un[S], knots x[off[s]..off[s+1]) ascending,
 * values Yk[(off[s]+j)*N + n].                                                */
static inline long searchsorted_left(const double* x, long n, double v) {
  long lo = 0, hi = n; /* first index with x[idx] >= v */
  while (lo < hi) {
    long mid = (lo + hi) / 2;
    if (x[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* SciPy interp1d._call_linear for one new abscissa, all N columns */
static void shell_eval(const double* x, const double* Yk, long P, int N, double u, double* out) {
  long j = searchsorted_left(x, P, u);
  if (j < 1) j = 1;
  if (j > P - 1) j = P - 1;
  const double* ylo = Yk + (j - 1) * N;
  const double* yhi = Yk + j * N;
  double dx = x[j] - x[j - 1], t = u - x[j - 1];
  for (int n = 0; n < N; n++) {
    double slope = (yhi[n] - ylo[n]) / dx;
    out[n] = slope * t + ylo[n];
  }
}

int orc_interp(int S, int N, const double* G_un, const int* off, const double* x, const double* Yk,
               const double* sch, int M, const double* dir, double* out /* M x ldo */, long ldo, double* tmp /* 2N */) {
  for (int m = 0; m < M; m++) {
    const double* g = sch + 7 * m;
    double u = fabs((g[0] * dir[0] + g[1] * dir[1]) + g[2] * dir[2]); /* mfu:1810 */
    double G = g[3];
    int sx = -1;
    for (int s = 0; s < S; s++) if (G_un[s] == G) { sx = s; break; } /* mfu:1822 exact equality */
    double* o = out + (long)m * ldo;
    if (sx >= 0) {
      shell_eval(x + off[sx], Yk + (long)off[sx] * N, off[sx + 1] - off[sx], N, u, o);
    } else {
      int ih = 0; /* np.argmax(Gms_un > Gnew), mfu:1829 */
      for (int s = 0; s < S; s++) if (G_un[s] > G) { ih = s; break; }
      if (ih == 0) return ORC_ERR_G_RANGE;
      int il = ih - 1;
      shell_eval(x + off[il], Yk + (long)off[il] * N, off[il + 1] - off[il], N, u, tmp);
      shell_eval(x + off[ih], Yk + (long)off[ih] * N, off[ih + 1] - off[ih], N, u, tmp + N);
      double dG = G_un[ih] - G_un[il], tG = G - G_un[il];
      for (int n = 0; n < N; n++) { /* mfu:1950-1955 */
        double slope = (tmp[N + n] - tmp[n]) / dG;
        o[n] = slope * tG + tmp[n];
      }
    }
  }
  return ORC_OK;
}

/* rotate_atom's per-direction evaluation (mfu:1423-1426) on pre-built per-row shell tables:
 * row m uses shell shell_of_row[m] (-1 = b0 row: copy sig, mfu:1298-1300); u = |g/|g| . n/|n||. */
int orc_rotate_eval(int N, const int* off, const double* x, const double* Yk, const double* sch, int ldsch, int M,
                    const int* shell_of_row, const double* sig, const double* newdir, double* out) {
  double nn = sqrt((newdir[0] * newdir[0] + newdir[1] * newdir[1]) + newdir[2] * newdir[2]);
  double d[3] = {newdir[0] / nn, newdir[1] / nn, newdir[2] / nn};
  for (int m = 0; m < M; m++) {
    const double* g = sch + (long)ldsch * m;
    int s = shell_of_row[m];
    if (s < 0) {
      for (int n = 0; n < N; n++) out[(long)m * N + n] = sig[(long)m * N + n];
      continue;
    }
    double gn = sqrt((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]);
    if (gn == 0) gn = INFINITY; /* mfu:1266 */
    double u = fabs(((g[0] / gn) * d[0] + (g[1] / gn) * d[1]) + (g[2] / gn) * d[2]);
    shell_eval(x + off[s], Yk + (long)off[s] * N, off[s + 1] - off[s], N, u, out + (long)m * N);
  }
  return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* _fit_voxel: mf:340-461 */
static double np_pairwise_sum(const double* a, long n) {
  /* NumPy's pairwise summation (np.add.reduce on a contiguous double vector) */
  if (n < 8) {
    double res = 0.;
    for (long i = 0; i < n; i++) res += a[i];
    return res;
  } else if (n <= 128) {
    double r[8];
    long i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
  } else {
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
  }
}
static double np_mean(const double* a, int n) { return np_pairwise_sum(a, n) / n; }

static int fit_voxel(int S, int N, const double* G_un, const int* off, const double* x, const double* Yk,
                     const double* sch, int M, const double* y, int K, int csf_i, int ear_i, const double* peaks,
                     int maxfasc, int csf_on, int ear_on, const double* sig_csf, const double* sig_ear, int E,
                     double* D /* M x maxdic scratch */, long maxdic, double* tmp, double* y_rec, double* params) {
  int num_params = 1 + 2 * maxfasc + csf_on + 2 * ear_on + 2; /* mf:381 */
  for (int i = 0; i < num_params; i++) params[i] = 0.0;
  if (K + csf_i + ear_i == 0) return ORC_OK; /* mf:387-388 */
  long sizes[4];
  int Kp = 0;
  long dicsize = (long)K * N + (csf_i > 0) + (ear_i > 0) * E; /* mf:371-373 */
  for (int k = 0; k < K; k++) { /* mf:391-397 */
    int rc = orc_interp(S, N, G_un, off, x, Yk, sch, M, peaks + 3 * k, D + (long)k * N, maxdic, tmp);
    if (rc) return rc;
    sizes[Kp++] = N;
  }
  if (csf_i) { /* mf:401-403 */
    for (int m = 0; m < M; m++) D[(long)m * maxdic + (long)K * N] = sig_csf[m];
    sizes[Kp++] = 1;
  }
  if (ear_i) { /* mf:404-408 */
    long st = (long)K * N + (csf_i > 0);
    for (int m = 0; m < M; m++) for (int e = 0; e < E; e++) D[(long)m * maxdic + st + e] = sig_ear[(long)m * E + e];
    sizes[Kp++] = E;
  }
  (void)dicsize;
  double w[4], SoS;
  long sub[4], tot[4];
  int rc = solve_exhaustive(D, maxdic, M, sizes, Kp, y, w, sub, tot, &SoS, y_rec);
  if (rc) return rc;
  double M0 = 0.0;
  for (int k = 0; k < Kp; k++) M0 += w[k]; /* np.sum of <=4 values: plain loop */
  double nu[4];
  for (int k = 0; k < Kp; k++) nu[k] = (fabs(M0) > 0) ? w[k] / M0 : w[k]; /* mf:420-425 */
  int i_csf = 2 * maxfasc + 1, i_ear = 2 * maxfasc + csf_on + 1;
  int i_mse = 2 * maxfasc + csf_on + 2 * ear_on + 1, i_R2 = i_mse + 1;
  params[0] = M0;
  for (int k = 0; k < K; k++) { params[1 + k] = nu[k]; params[1 + maxfasc + k] = (double)sub[k]; }
  if (csf_i) params[i_csf] = nu[K];
  if (ear_i) { params[i_ear] = nu[K + (csf_i > 0)]; params[i_ear + 1] = (double)sub[K + (csf_i > 0)]; }
  params[i_mse] = SoS / M;
  /* R2 = corrcoef(y, y_rec)[0,1]**2 if M>1 and both std > 0 (mf:449-450) */
  if (M > 1) {
    double my = np_mean(y, M), mr = np_mean(y_rec, M);
    double cyy = 0, crr = 0, cyr = 0;
    for (int m = 0; m < M; m++) {
      double a = y[m] - my, b = y_rec[m] - mr;
      cyy += a * a; crr += b * b; cyr += a * b;
    }
    if (cyy > 0 && crr > 0) {
      double f = (double)(M - 1);
      double c00 = cyy / f, c11 = crr / f, c01 = cyr / f;
      double r = c01 / sqrt(c00) / sqrt(c11);
      if (r > 1) r = 1; if (r < -1) r = -1;
      params[i_R2] = r * r;
    }
  }
  return ORC_OK;
}

/* Voxel loop of MFModel.fit (mf:1017-1028 serial; mf:978-1009 process pool == nthreads>1 here) */
int orc_fit_batch(int S, int N, const double* G_un, const int* off, const double* x, const double* Yk,
                  const double* sch, int M, const double* Y /* V x M */, const int* Kv, const unsigned char* csfv,
                  const unsigned char* earv, const double* peaks /* V x 3*maxfasc */, int maxfasc, int csf_on,
                  int ear_on, const double* sig_csf, const double* sig_ear, int E, long V, double* params_out,
                  int nthreads) {
  int num_params = 1 + 2 * maxfasc + csf_on + 2 * ear_on + 2;
  long maxdic = (long)maxfasc * N + csf_on + (long)ear_on * E;
  if (maxdic < 1) maxdic = 1;
  int err = 0;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
  {
    double* D = (double*)calloc((size_t)M * maxdic, sizeof(double));
    double* tmp = (double*)malloc(sizeof(double) * 2 * (N > 0 ? N : 1));
    double* yrec = (double*)malloc(sizeof(double) * M);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (long v = 0; v < V; v++) {
      int rc = fit_voxel(S, N, G_un, off, x, Yk, sch, M, Y + v * M, Kv[v], csfv ? csfv[v] : 0, earv ? earv[v] : 0,
                         peaks + v * 3 * maxfasc, maxfasc, csf_on, ear_on, sig_csf, sig_ear, E, D, maxdic, tmp, yrec,
                         params_out + v * num_params);
      if (rc) err = rc;
    }
    free(D); free(tmp); free(yrec);
    scratch_release();   /* this thread's Gram arrays, kept across its voxels */
  }
  return err;
}

/* ---------------------------------------------------------------------------
 * Monte-Carlo average of the spins' dephasing: mfu:2762-2810, loop for loop
 * (sequential sum over spins, products accumulated in dimension order, libm cos).
 * nthreads > 1 splits the SEQUENCES (independent outputs) over OpenMP threads.   */
int orc_monte_carlo_average(const double* sim_phases, long n_entries, int dim, const long* delta_mapping,
                            const double* gscaling, double Dscaling, long num_spins, long n_seq, double* signal,
                            int nthreads) {
  for (long i = 0; i < n_seq; ++i)
    if (delta_mapping[i] < 0 || (delta_mapping[i] + 1) * num_spins > n_entries) return 2;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (long iseq = 0; iseq < n_seq; ++iseq) {
    const long start = delta_mapping[iseq] * num_spins;      /* mfu:2801-2802 */
    double acc = 0.0;
    for (long ispin = 0; ispin < num_spins; ++ispin) {
      double ph_final = 0.0;                                 /* mfu:2804 */
      const double* p = sim_phases + (start + ispin) * dim;
      for (int idim = 0; idim < dim; ++idim) ph_final += gscaling[iseq * dim + idim] * p[idim];   /* mfu:2806-2807 */
      acc += cos(Dscaling * ph_final);                       /* mfu:2808 */
    }
    signal[iseq] = acc / (double)num_spins;                  /* mfu:2809 */
  }
  return 0;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
