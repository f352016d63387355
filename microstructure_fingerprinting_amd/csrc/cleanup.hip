// cleanup.hip -- per-voxel selection of 0, 1 or 2 of two detected fascicle orientations: the voxel loop of the
// reference's cleanup_2fascicles (ref mf.py:170-335; NeuroImage 184 (2019) 964-980), one thread per ROI voxel.
// Every step is a handful of IEEE additions, multiplications, one division and one square root in the reference's
// order (-ffp-contract=off), so the outputs equal the reference's bit for bit.
#pragma once
#include <hip/hip_runtime.h>

struct CleanupArgs {
  const double* f1;      // [n] weight of population 0
  const double* f2;      // [n] weight of population 1
  const double* p1;      // [n x 3] direction of population 0
  const double* p2;      // [n x 3]
  long n;
  double cos_min;        // cos of the merge angle (ref: np.cos(ANG_MIN * np.pi / 180), computed by the caller)
  double ratio, w_keep, w_small;
  double* peaks_out;     // [n x 6]
  double* count_out;     // [n]
};

__global__ __launch_bounds__(256) void mfx_cleanup_kernel(CleanupArgs a) {
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= a.n) return;
  double f0 = a.f1[v], f1 = a.f2[v];
  double p0[3] = {a.p1[3 * v], a.p1[3 * v + 1], a.p1[3 * v + 2]};
  double p1[3] = {a.p2[3 * v], a.p2[3 * v + 1], a.p2[3 * v + 2]};
  double count = 2.0;
  // 1. merge nearly parallel peaks into slot 0 (sign-aware sum, weights added)
  const double dp = (p0[0] * p1[0] + p0[1] * p1[1]) + p0[2] * p1[2];
  const double dpc = dp < -1.0 ? -1.0 : (dp > 1.0 ? 1.0 : dp);       // np.clip (NaN stays NaN: no merge)
  if (fabs(dpc) > a.cos_min) {
    const double sg = dp > 0.0 ? 1.0 : (dp < 0.0 ? -1.0 : dp);       // np.sign
    double s[3];
    for (int k = 0; k < 3; ++k) s[k] = p0[k] + p1[k] * sg;
    const double nrm = sqrt((s[0] * s[0] + s[1] * s[1]) + s[2] * s[2]);
    for (int k = 0; k < 3; ++k) { p0[k] = s[k] / nrm; p1[k] = 0.0; }
    f0 = f0 + f1;
    f1 = 0.0;
    count = 1.0;
  }
  // 2. relatively small population 0: population 1 takes its slot
  if ((f1 > a.ratio * f0) && (f0 < a.w_keep)) {
    for (int k = 0; k < 3; ++k) { p0[k] = p1[k]; p1[k] = 0.0; }
    f0 = f1;
    f1 = 0.0;
    count = f0 > 0.0 ? 1.0 : 0.0;
  }
  // 3. relatively small population 1: dropped, weight not transferred
  if ((f0 > a.ratio * f1) && (f1 < a.w_keep)) {
    for (int k = 0; k < 3; ++k) p1[k] = 0.0;
    f1 = 0.0;
    count = f0 > 0.0 ? 1.0 : 0.0;
  }
  // 4./5. small absolute weights
  if (f0 < a.w_small) {
    for (int k = 0; k < 3; ++k) p0[k] = 0.0;
    f0 = 0.0;
    count = count - 1.0;
  }
  if (f1 < a.w_small) {
    for (int k = 0; k < 3; ++k) p1[k] = 0.0;
    f1 = 0.0;
    count = f0 > 0.0 ? 1.0 : 0.0;
  }
  // 6. heavier population first; the reference's reversed ascending argsort puts slot 1 first on ties
  const bool swap = f1 >= f0;
  double* o = a.peaks_out + 6 * v;
  for (int k = 0; k < 3; ++k) { o[k] = swap ? p1[k] : p0[k]; o[3 + k] = swap ? p0[k] : p1[k]; }
  a.count_out[v] = count;
}
