// rotate.hip -- batched dictionary rotation (materialising variant).
//
// Replaces the evaluation part of interp_PGSE_from_multishell (mf_utils.py:1810-1955) and of
// rotate_atom (mf_utils.py:1423-1426) for B directions at once.  The fused fit kernels never call
// this (they generate rotated atoms on the fly); it backs the drop-in mf_utils API and synthetic
// data generation.  HBM-write bound: B*M*N*8 bytes out, table reads served from L2.
#pragma once
#include "mfx_device.h"

#define MFX_ROT_ROWS 16
#define MFX_ROT_WG 256

// out[b * bstride + m * rstride + n] (b-major [B x M x N]: bstride = M N, rstride = N; sub-dictionary b of a row-major
// [M x K N] matrix: bstride = N, rstride = K N), one workgroup per (direction b, block of MFX_ROT_ROWS protocol rows)
__device__ __forceinline__ void mfx_rotate_block(const TablesDev& T, const PlanDev& P, const double* __restrict__ dir3,
                                                 int normalise, double* __restrict__ out, long rstride, int m0) {
  __shared__ RowDesc s_rd[MFX_ROT_ROWS];
  __shared__ double s_tG[MFX_ROT_ROWS], s_dG[MFX_ROT_ROWS];
  const int M = P.M, N = T.N, ldn = T.ldn;
  if (threadIdx.x < MFX_ROT_ROWS) {
    const int m = m0 + threadIdx.x;
    if (m < M) {
      double d0 = dir3[0], d1 = dir3[1], d2 = dir3[2];
      if (normalise && !P.normalise) {  // rotate_atom: newdir / |newdir| (mf_utils.py:1262,1269); explicit plans do it in mfx_row_desc
        const double nn = sqrt((d0 * d0 + d1 * d1) + d2 * d2);
        d0 /= nn; d1 /= nn; d2 /= nn;
      }
      s_rd[threadIdx.x] = mfx_row_desc(T, P, m, d0, d1, d2);
      s_tG[threadIdx.x] = P.tG[m];
      s_dG[threadIdx.x] = P.dG[m];
    }
  }
  __syncthreads();
  const int rows = min(MFX_ROT_ROWS, M - m0);
  for (int idx = threadIdx.x; idx < rows * N; idx += MFX_ROT_WG) {
    const int r = idx / N, n = idx - r * N;
    out[(size_t)(m0 + r) * rstride + n] = mfx_eval_br(T.tab, ldn, s_rd[r], s_tG[r], s_dG[r], n);
  }
}

__global__ __launch_bounds__(MFX_ROT_WG) void mfx_rotate_kernel(TablesDev T, PlanDev P, const double* __restrict__ dirs,
                                                                int normalise, double* __restrict__ out, long bstride,
                                                                long rstride) {
  const int b = blockIdx.y;
  mfx_rotate_block(T, P, dirs + 3 * (size_t)b, normalise, out + (size_t)b * bstride, rstride, blockIdx.x * MFX_ROT_ROWS);
}

// The K rotated dictionaries of a batch of voxels, each voxel's as one row-major [M x K N] matrix (grid (rows, K, voxels)):
// voxel vox[z]'s fascicle y along peaks[vox[z] * peaks_ld + 3 y ..], into out + z * M * K * N, columns y N ..
__global__ __launch_bounds__(MFX_ROT_WG) void mfx_rotate_voxels_kernel(TablesDev T, PlanDev P, const double* __restrict__ peaks,
                                                                       int peaks_ld, const int* __restrict__ vox,
                                                                       double* __restrict__ out) {
  const long LD = (long)gridDim.y * T.N;
  mfx_rotate_block(T, P, peaks + (size_t)vox[blockIdx.z] * peaks_ld + 3 * blockIdx.y, 0,
                   out + (size_t)blockIdx.z * P.M * LD + (size_t)blockIdx.y * T.N, LD, blockIdx.x * MFX_ROT_ROWS);
}

// out[b][m] = rotated atom cols[b] only (one atom per direction); thread per (b, m)
__global__ __launch_bounds__(256) void mfx_rotate_cols_kernel(TablesDev T, PlanDev P, const double* __restrict__ dirs,
                                                              const int* __restrict__ cols, int64_t B, int normalise,
                                                              double* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int M = P.M;
  if (idx >= B * M) return;
  const int64_t b = idx / M;
  const int m = (int)(idx - b * M);
  double d0 = dirs[3 * b], d1 = dirs[3 * b + 1], d2 = dirs[3 * b + 2];
  if (normalise && !P.normalise) {
    const double nn = sqrt((d0 * d0 + d1 * d1) + d2 * d2);
    d0 /= nn; d1 /= nn; d2 /= nn;
  }
  const RowDesc rd = mfx_row_desc(T, P, m, d0, d1, d2);
  out[idx] = mfx_eval_br(T.tab, T.ldn, rd, P.tG[m], P.dG[m], cols[b]);
}
