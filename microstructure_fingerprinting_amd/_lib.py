"""ctypes binding of the C-ABI shared library ``libmfx.so`` (include/mfx.h).

The library is the product's only compute path.  If it is missing or no MI355X is
visible, calls raise -- there is no NumPy/CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmfx.so")

MFX_OK, MFX_ERR_ARG, MFX_ERR_G_RANGE, MFX_ERR_NO_DEVICE, MFX_ERR_HIP, MFX_ERR_UNSUPPORTED, MFX_ERR_DIR_NORM = range(7)

_lib = None

# every symbol include/mfx.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "mfx_last_error", "mfx_device_count", "mfx_abi_version",
    "mfx_tables_create", "mfx_tables_destroy", "mfx_tables_num_atoms",
    "mfx_plan_create_multishell", "mfx_plan_create_explicit", "mfx_plan_destroy", "mfx_plan_status",
    "mfx_rotate", "mfx_rotate_dev", "mfx_rotate_cols", "mfx_rotate_cols_dev", "mfx_fit_batch", "mfx_fit_batch_rows", "mfx_fit_batch_volume", "mfx_volume_rows", "mfx_thread_release", "mfx_fit_batch_dev",
    "mfx_solve_exhaustive", "mfx_monte_carlo_average", "mfx_monte_carlo_average_dev", "mfx_cleanup_2fascicles", "mfx_cleanup_2fascicles_dev", "mfx_last_kernel_ms", "mfx_set_profiling", "mfx_debug_set_stamps", "mfx_debug_last_fallback_count", "mfx_debug_last_guard_count", "mfx_debug_last_counter", "mfx_debug_set_k2_screen", "mfx_debug_set_k2_wide", "mfx_debug_set_k2x_screen", "mfx_debug_set_k2_maxc", "mfx_debug_set_k2x_maxc", "mfx_debug_set_k2s_cap", "mfx_debug_set_k2s_images", "mfx_debug_set_k3_cap", "mfx_debug_set_force_generic", "mfx_debug_set_k3_screen",
]


class MfxError(RuntimeError):
    pass


def lib():
    """Load libmfx.so (fails loudly if the HIP extension has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MfxError("HIP extension %s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(or `make -C microstructure_fingerprinting_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    lp = C.POINTER(C.c_int64)
    bp = C.POINTER(C.c_uint8)
    vp = C.c_void_p
    L.mfx_last_error.restype = C.c_char_p
    L.mfx_device_count.restype = C.c_int
    L.mfx_abi_version.restype = C.c_int
    L.mfx_tables_create.argtypes = [dp, ip, dp, dp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.mfx_tables_destroy.argtypes = [vp]
    L.mfx_tables_destroy.restype = None
    L.mfx_tables_num_atoms.argtypes = [vp]
    L.mfx_plan_create_multishell.argtypes = [vp, dp, C.c_int, C.POINTER(vp)]
    L.mfx_plan_create_explicit.argtypes = [vp, dp, ip, C.c_int, C.POINTER(vp)]
    L.mfx_plan_destroy.argtypes = [vp]
    L.mfx_plan_destroy.restype = None
    L.mfx_plan_status.argtypes = [vp, vp]
    L.mfx_rotate.argtypes = [vp, dp, C.c_int64, C.c_int, dp]
    L.mfx_rotate_dev.argtypes = [vp, vp, C.c_int64, C.c_int, vp, vp]
    L.mfx_rotate_cols.argtypes = [vp, dp, ip, C.c_int64, C.c_int, dp]
    L.mfx_rotate_cols_dev.argtypes = [vp, vp, vp, C.c_int64, C.c_int, vp, vp]
    L.mfx_fit_batch.argtypes = [vp, dp, ip, bp, bp, dp, C.c_int, C.c_int, C.c_int, dp, dp, C.c_int, C.c_int64, dp]
    L.mfx_fit_batch_rows.argtypes = [vp, dp, lp, ip, bp, bp, dp, C.c_int, C.c_int, C.c_int, dp, dp, C.c_int, C.c_int64, dp]
    L.mfx_fit_batch_volume.argtypes = [vp, vp, C.c_int, C.c_double, C.c_double, C.c_int64, lp, ip, bp, bp, dp, C.c_int, C.c_int,
                                       C.c_int, dp, dp, C.c_int, C.c_int64, dp]
    L.mfx_volume_rows.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_int64, C.c_int, lp, C.c_int64, dp, C.c_int]
    L.mfx_fit_batch_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int64, vp, vp]
    L.mfx_solve_exhaustive.argtypes = [dp, C.c_int64, C.c_int, lp, C.c_int, dp, dp, lp, lp, dp, dp]
    L.mfx_monte_carlo_average.argtypes = [dp, C.c_int64, C.c_int, lp, dp, C.c_double, C.c_int64, C.c_int64, dp, C.c_int]
    L.mfx_monte_carlo_average_dev.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, C.c_int, lp, dp, C.c_double, C.c_int64,
                                              C.c_int64, dp, vp]
    L.mfx_cleanup_2fascicles.argtypes = [dp, dp, dp, dp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp, C.c_int]
    L.mfx_cleanup_2fascicles_dev.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, vp, vp, vp]
    L.mfx_last_kernel_ms.restype = C.c_double
    L.mfx_set_profiling.argtypes = [C.c_int]
    L.mfx_set_profiling.restype = None
    L.mfx_debug_set_stamps.argtypes = [vp]
    L.mfx_debug_set_stamps.restype = None
    L.mfx_debug_last_fallback_count.restype = C.c_int
    L.mfx_debug_last_guard_count.restype = C.c_int
    L.mfx_debug_last_counter.argtypes = [C.c_int]
    L.mfx_debug_last_counter.restype = C.c_int
    L.mfx_debug_set_k2x_maxc.argtypes = [C.c_int]
    L.mfx_debug_set_k2x_maxc.restype = None
    L.mfx_debug_set_k2_screen.argtypes = [C.c_int]
    L.mfx_debug_set_k2_screen.restype = None
    L.mfx_debug_set_k2_wide.argtypes = [C.c_int]
    L.mfx_debug_set_k2_wide.restype = None
    L.mfx_debug_set_k2x_screen.argtypes = [C.c_int]
    L.mfx_debug_set_k2x_screen.restype = None
    L.mfx_debug_set_k2_maxc.argtypes = [C.c_int]
    L.mfx_debug_set_k2_maxc.restype = None
    L.mfx_debug_set_k2s_cap.argtypes = [C.c_int]
    L.mfx_debug_set_k2s_cap.restype = None
    L.mfx_debug_set_k3_cap.argtypes = [C.c_int]
    L.mfx_debug_set_k3_cap.restype = None
    L.mfx_debug_set_force_generic.argtypes = [C.c_int]
    L.mfx_debug_set_force_generic.restype = None
    L.mfx_debug_set_k3_screen.argtypes = [C.c_int]
    L.mfx_debug_set_k3_screen.restype = None
    L.mfx_debug_set_k2s_images.argtypes = [C.c_int]
    L.mfx_debug_set_k2s_images.restype = None
    _lib = L
    return L


def check(rc):
    """Map a C status code to the exception class the reference raises for that condition."""
    if rc == MFX_OK:
        return
    msg = lib().mfx_last_error().decode("utf-8", "replace")
    if rc in (MFX_ERR_G_RANGE, MFX_ERR_DIR_NORM):
        raise ValueError(msg)             # mf_utils.py:1798-1802, 1829-1836
    if rc == MFX_ERR_ARG:
        raise ValueError(msg)
    if rc == MFX_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise MfxError(msg)


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def lptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def bptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def f64c(a):
    return np.ascontiguousarray(a, dtype=np.float64)
