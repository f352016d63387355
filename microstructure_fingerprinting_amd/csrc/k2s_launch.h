// k2s_launch.h -- launcher of the screening kernel (fit_k2s.hip), shared by the tu_k2s_*.hip translation units
// (one per k-step count: the instantiations are what takes the compile time).
#pragma once
#include "mfx_host.h"
#include "fit_k2s.hip"

template <int KS, bool BR, int NB>
static int launch_k2s_t(const FitK2Args& a, int nvox, hipStream_t st) {
  MfxThread& T = mfx_thread();
  const size_t lds = mfx_k2s_lds_bytes(KS, a.T.N, BR, NB);
  auto kern = mfx_fit_k2s_kernel<KS, BR, NB>;
  HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  StreamMem fbm(st);   // [0] hand-back count, [1] guard count, [2..4] audit, [8..] voxel list
  HIPCHK(fbm.alloc(sizeof(int) * ((size_t)nvox + 8)));
  int* fb = fbm.as<int>();
  HIPCHK(hipMemsetAsync(fb, 0, 8 * sizeof(int), st));
  if (int rc = mfx_prof_begin(st)) return rc;
  FitK2Args aa = a;
  aa.stamps = T.stamps;
  aa.fb_count = fb;
  aa.fb_list = fb + 8;
  aa.audit = fb + 2;   // [2] audited pairs beyond DC/4, [3] largest error (1e-11), [4] audited pairs
  aa.scap = T.k2s_cap ? T.k2s_cap : MFX_S_CAP;
  hipLaunchKernelGGL(kern, dim3(nvox), dim3(512), lds, st, aa);
  HIPCHK(hipGetLastError());
  if (int rc = mfx_prof_end(st)) return rc;
  // Voxels the screening kernel could not decide (short-list ring overflow, screening-error guard) are redone by the
  // FP64 kernel straight from the device-side list: no host read, the call stays asynchronous.  The launch covers
  // nvox blocks; all but the first *fb exit at once.
  FitK2Args ab = a;
  ab.vox_list = fb + 8;
  ab.list_count = fb;
  if (int rc = mfx_launch_k2_f64(ab, nvox, st, false)) return rc;
  if (int rc = mfx_fb_accumulate(fb, 2, st)) return rc;
  return mfx_fb_accumulate_audit(fb + 2, st);
}

#define MFX_K2S_TU(KS_, NAME_)                                                            \
  int NAME_(const FitK2Args& a, int nvox, hipStream_t st, bool br, int NB) {              \
    if (NB == 3) return br ? launch_k2s_t<KS_, true, 3>(a, nvox, st) : launch_k2s_t<KS_, false, 3>(a, nvox, st); \
    return br ? launch_k2s_t<KS_, true, 2>(a, nvox, st) : launch_k2s_t<KS_, false, 2>(a, nvox, st);              \
  }
