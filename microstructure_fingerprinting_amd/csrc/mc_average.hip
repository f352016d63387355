// mc_average.hip -- Monte-Carlo signal synthesis from stored spin phases (dictionary generation,
// upstream of fitting):  S_i = (1/n_spin) * sum_l cos(Dscaling * sum_n gscaling[i,n] * phi[map(i)*n_spin + l, n])
// Reference: mf_utils.py:2758-2810 (monte_carlo_average), called by get_PGSE_from_phases (mf_utils.py:2813-3015).
//
// Bound: FP64 VALU (one double-precision cosine per (sequence, spin) pair, ~60 VALU instructions);
// HBM traffic is small because a workgroup keeps its spins' phases in registers and reuses them
// for a tile of MFX_MC_TS sequences that share the same simulated (Delta, delta) acquisition.
//
// Arithmetic per (sequence, spin) term follows the reference order exactly (products accumulated in
// dimension order, then one multiply by Dscaling, then cos; no FMA contraction).  Only the order in
// which the n_spin terms are ADDED differs (per-thread strided partial sums -> wave -> workgroup ->
// fixed-order sum over chunks; deterministic, and at least as accurate as the sequential sum).
#pragma once
#include <hip/hip_runtime.h>

#define MFX_MC_TS 8          // sequences per workgroup tile
#define MFX_MC_THREADS 256
#define MFX_MC_SP 16         // spins per thread
#define MFX_MC_CHUNK (MFX_MC_THREADS * MFX_MC_SP)

struct McArgs {
  const double* ph;          // phases, element (entry e, dimension d) at ph[e*spin_stride + d*dim_stride]
  long spin_stride, dim_stride;
  int dim;                   // 1..3
  const int* tile_first;     // [n_tiles] first (sorted) sequence of the tile
  const int* tile_cnt;       // [n_tiles] sequences in the tile (1..MFX_MC_TS)
  const long* tile_start;    // [n_tiles] first phase entry of the tile's reference acquisition
  const double* gs;          // [n_seq][3] gradient scaling, sorted order, zero padded
  const int* order;          // [n_seq] sorted position -> caller's sequence index
  double Ds;
  long num_spins;
  int n_tiles;
  int nchunks;
  int n_seq;
  double* partial;           // [n_seq][nchunks]
  double* signal;            // [n_seq] in the caller's order
};

__device__ __forceinline__ double mfx_mc_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__global__ __launch_bounds__(MFX_MC_THREADS) void mfx_mc_partial_kernel(McArgs a) {
  __shared__ double s_part[MFX_MC_THREADS / 64][MFX_MC_TS];
  const long b = blockIdx.x;
  const int tile = (int)(b % a.n_tiles);      // tiles of one spin chunk are neighbours in launch order: L2 reuse
  const long chunk = b / a.n_tiles;
  const int first = a.tile_first[tile], cnt = a.tile_cnt[tile];
  const long start = a.tile_start[tile];
  const int dim = a.dim;
  double g[MFX_MC_TS][3];
#pragma unroll
  for (int t = 0; t < MFX_MC_TS; ++t)
#pragma unroll
    for (int d = 0; d < 3; ++d) g[t][d] = (t < cnt) ? a.gs[(long)(first + t) * 3 + d] : 0.0;
  double acc[MFX_MC_TS];
#pragma unroll
  for (int t = 0; t < MFX_MC_TS; ++t) acc[t] = 0.0;

  const long spin0 = chunk * MFX_MC_CHUNK + threadIdx.x;
#pragma unroll 2
  for (int it = 0; it < MFX_MC_SP; ++it) {
    const long spin = spin0 + (long)it * MFX_MC_THREADS;
    if (spin < a.num_spins) {
      const double* q = a.ph + (start + spin) * a.spin_stride;
      const double p0 = q[0];
      const double p1 = dim > 1 ? q[a.dim_stride] : 0.0;
      const double p2 = dim > 2 ? q[2 * a.dim_stride] : 0.0;
#pragma unroll
      for (int t = 0; t < MFX_MC_TS; ++t) {
        if (t < cnt) {
          double phs = g[t][0] * p0;                 // ref:2804-2807 (0 + x == x)
          if (dim > 1) phs += g[t][1] * p1;
          if (dim > 2) phs += g[t][2] * p2;
          acc[t] += cos(a.Ds * phs);                 // ref:2808
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < MFX_MC_TS; ++t) {
    const double v = mfx_mc_wave_sum(acc[t]);
    if (lane == 0) s_part[wave][t] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < cnt) {
    double v = s_part[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < MFX_MC_THREADS / 64; ++w) v += s_part[w][threadIdx.x];
    a.partial[(long)(first + threadIdx.x) * a.nchunks + chunk] = v;
  }
}

// one wave per sequence: fixed-order sum over the spin chunks, then the mean (ref:2809)
__global__ __launch_bounds__(64) void mfx_mc_finalize_kernel(McArgs a) {
  const int q = blockIdx.x;
  if (q >= a.n_seq) return;
  double v = 0.0;
  for (int c = threadIdx.x; c < a.nchunks; c += 64) v += a.partial[(long)q * a.nchunks + c];
  v = mfx_mc_wave_sum(v);
  if (threadIdx.x == 0) a.signal[a.order[q]] = v / (double)a.num_spins;
}
