// k2w_launch.h -- launcher of the wide screening kernel (fit_k2w.hip), shared by the tu_k2w_*.hip translation units
#pragma once
#include "mfx_host.h"
#include "fit_k2w.hip"

template <int KS, int TL, bool BR, int NB>
static int launch_k2w_t(const FitK2Args& a, int nvox, hipStream_t st) {
  MfxThread& T = mfx_thread();
  const size_t lds = mfx_k2w_lds_bytes(KS, a.T.N, BR, NB, TL);
  if (lds > 160 * 1024) return mfx_fail(MFX_ERR_UNSUPPORTED, "wide K=2 kernel needs %zu B of LDS (> 160 KiB): N=%d too large", lds, a.T.N);
  auto kern = mfx_fit_k2w_kernel<KS, TL, BR, NB>;
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
  hipLaunchKernelGGL(kern, dim3(nvox), dim3(256), lds, st, aa);
  HIPCHK(hipGetLastError());
  if (int rc = mfx_prof_end(st)) return rc;
  // hand-backs (ring overflow, screening-error guard) go to the FP64 kernel through the device-side list, as in k2s_launch.h
  FitK2Args ab = a;
  ab.vox_list = fb + 8;
  ab.list_count = fb;
  if (int rc = mfx_launch_k2_f64(ab, nvox, st, false)) return rc;
  if (int rc = mfx_fb_accumulate(fb, 2, st)) return rc;
  return mfx_fb_accumulate_audit(fb + 2, st);
}

#define MFX_K2W_TU(KS_, TL_, NB_, NAME_)                                                    \
  int NAME_(const FitK2Args& a, int nvox, hipStream_t st, bool br) {                        \
    return br ? launch_k2w_t<KS_, TL_, true, NB_>(a, nvox, st) : launch_k2w_t<KS_, TL_, false, NB_>(a, nvox, st); \
  }
