// extras.hip -- the voxel-independent CSF / EAR columns of a voxel class (mf.py:401-408, 918-925)
#pragma once
#include <hip/hip_runtime.h>

// Builds the extra-column block of one voxel class on the device (one 256-thread workgroup): x [M x NX] = [csf | ear_0 ..
// ear_{E-1}] and its Gram Gxx [NX x NX], every entry summed sequentially over the rows like the reference's Gram loops
// (mf_utils.py:311-319, 517-531).  Inputs are device pointers, so the asynchronous entry point needs no host round trip.
__global__ __launch_bounds__(256) void mfx_extras_kernel(const double* __restrict__ sig_csf, const double* __restrict__ sig_ear,
                                                         int M, int has_csf, int E, double* __restrict__ x, double* __restrict__ G) {
  const int NX = has_csf + E;
  for (int q = threadIdx.x; q < M * NX; q += 256) {
    const int m = q / NX, c = q - m * NX;
    x[q] = (has_csf && c == 0) ? sig_csf[m] : sig_ear[(size_t)m * E + (c - has_csf)];
  }
  __syncthreads();
  for (int q = threadIdx.x; q < NX * NX; q += 256) {
    const int p = q / NX, r = q - p * NX;
    double acc = 0.0;
    for (int m = 0; m < M; ++m) acc += x[(size_t)m * NX + p] * x[(size_t)m * NX + r];
    G[q] = acc;
  }
}

