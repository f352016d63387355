"""Thin object layer over the C ABI: device tables, protocol plans, batched fit."""
import ctypes as C

import numpy as np

from . import _lib as L


class DeviceTables:
    """Per-shell knot tables resident in HBM (mfx_tables)."""

    def __init__(self, xs, Ys, G_un, device=0):
        self.xs = [np.ascontiguousarray(x, dtype=np.float64) for x in xs]
        self.Ys = [np.ascontiguousarray(y, dtype=np.float64) for y in Ys]
        self.G_un = np.ascontiguousarray(G_un, dtype=np.float64)
        self.N = int(self.Ys[0].shape[1])
        self.S = len(self.xs)
        self.device = device
        self._h = None
        self.off = np.concatenate([[0], np.cumsum([x.size for x in self.xs])]).astype(np.int32)

    def handle(self):
        if self._h is None:
            x = np.ascontiguousarray(np.concatenate(self.xs))
            Y = np.ascontiguousarray(np.concatenate(self.Ys, axis=0))
            h = C.c_void_p()
            L.check(L.lib().mfx_tables_create(L.dptr(x), L.iptr(self.off), L.dptr(Y), L.dptr(self.G_un), self.S,
                                             self.N, self.device, C.byref(h)))
            self._h = h
        return self._h

    def close(self):
        if self._h is not None:
            L.lib().mfx_tables_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """Per-protocol row->shell mapping resident in HBM (mfx_plan)."""

    def __init__(self, tables, scheme=None, gdirs=None, shell_of_row=None):
        self.tables = tables
        h = C.c_void_p()
        if scheme is not None:
            sch = L.f64c(scheme)
            if sch.ndim != 2 or sch.shape[1] != 7:
                raise ValueError("pgse_scheme should have 7 columns")
            self.M = sch.shape[0]
            L.check(L.lib().mfx_plan_create_multishell(tables.handle(), L.dptr(sch), self.M, C.byref(h)))
        else:
            g = L.f64c(gdirs)
            s = np.ascontiguousarray(shell_of_row, dtype=np.int32)
            self.M = g.shape[0]
            L.check(L.lib().mfx_plan_create_explicit(tables.handle(), L.dptr(g), L.iptr(s), self.M, C.byref(h)))
        self._h = h

    def handle(self):
        return self._h

    def close(self):
        if self._h is not None:
            L.lib().mfx_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def num_params(maxfasc, csf_on, ear_on):
    return 1 + 2 * maxfasc + int(csf_on) + 2 * int(ear_on) + 2   # mf.py:381


class FileOrderVolume(object):
    """A 4-D data volume in the layout NIfTI files (and nibabel's arrays) have: ``array`` is Fortran-contiguous with
    shape (..., M), i.e. one 3-D image per measurement, in one of the scalar types of ``NIFTI_CODES``; ``slope`` /
    ``inter`` are the header's scaling (slope 0: none).  ``get_fdata()`` gives what nib.load(f).get_fdata() would."""
    NIFTI_CODES = {'u1': 2, 'i2': 4, 'i4': 8, 'f4': 16, 'f8': 64, 'i1': 256, 'u2': 512, 'u4': 768}

    def __init__(self, array, slope=0.0, inter=0.0):
        self.array, self.slope, self.inter = array, float(slope), float(inter)
        self.shape = array.shape

    @staticmethod
    def accepts(a):
        return (isinstance(a, np.ndarray) and a.ndim >= 2 and a.flags.f_contiguous and not a.flags.c_contiguous
                and a.dtype.isnative and a.dtype.str[1:] in FileOrderVolume.NIFTI_CODES)

    @property
    def scaled(self):
        return self.slope != 0.0 and not (self.slope == 1.0 and self.inter == 0.0) and np.isfinite(self.slope)

    def get_fdata(self):
        d = self.array.astype(np.float64)
        return d * self.slope + self.inter if self.scaled else d

    def file_order_index(self, c_flat_index):
        """Positions inside one 3-D image of the voxels with C-order flat indices ``c_flat_index`` of the image grid."""
        grid = self.shape[:-1]
        return np.ravel_multi_index(np.unravel_index(c_flat_index, grid), grid, order='F').astype(np.int64)


def volume_rows(vol, vox, device=0):
    """mfx_volume_rows: ``vol.get_fdata().reshape(-1, n, order='F')[vox]`` (float64 [V x n]) with the conversion, scaling
    and gather on the device - for the per-voxel quantities MFModel.fit indexes with the mask beside the data."""
    a = vol.array
    n = a.shape[-1]
    nvox = int(np.prod(a.shape[:-1]))
    vox = np.ascontiguousarray(vox, dtype=np.int64)
    V = vox.shape[0]
    if V and (vox.min() < 0 or vox.max() >= nvox):
        raise ValueError("voxel indices out of range")
    out = np.zeros((V, n))
    L.check(L.lib().mfx_volume_rows(a.ctypes.data, FileOrderVolume.NIFTI_CODES[a.dtype.str[1:]], vol.slope, vol.inter, nvox, n,
                                    L.lptr(vox), V, L.dptr(out), int(device)))
    return out


def fit_batch_volume(plan, vol, vox, K, csf, ear, peaks, maxfasc, csf_on, ear_on, sig_csf=None, sig_ear=None, E=0):
    """mfx_fit_batch_volume: like ``fit_batch(plan, vol.get_fdata().reshape(-1, M, order='F')[vox], ...)`` with the
    volume uploaded in its own layout and type and the conversion, scaling and ROI gather (mf.py:623-657) on the device."""
    a = vol.array
    M = a.shape[-1]
    if M != plan.M:
        raise ValueError("data has %d measurements, protocol has %d" % (M, plan.M))
    nvox = int(np.prod(a.shape[:-1]))
    vox = np.ascontiguousarray(vox, dtype=np.int64)
    V = vox.shape[0]
    if V and (vox.min() < 0 or vox.max() >= nvox):
        raise ValueError("voxel indices out of range")
    K = np.ascontiguousarray(K, dtype=np.int32)
    if K.shape[0] != V:
        raise ValueError("K should have one entry per voxel")
    csf_a = np.ascontiguousarray(csf, dtype=np.uint8) if csf is not None else np.zeros(V, np.uint8)
    ear_a = np.ascontiguousarray(ear, dtype=np.uint8) if ear is not None else np.zeros(V, np.uint8)
    pk = L.f64c(peaks).reshape(V, -1) if maxfasc > 0 else np.zeros((V, 3))
    if maxfasc > 0 and pk.shape[1] != 3 * maxfasc:
        raise ValueError("peaks should have %d columns" % (3 * maxfasc))
    out = np.zeros((V, num_params(maxfasc, csf_on, ear_on)))
    sc = L.f64c(sig_csf) if sig_csf is not None else None
    se = L.f64c(sig_ear) if sig_ear is not None else None
    L.check(L.lib().mfx_fit_batch_volume(plan.handle(), a.ctypes.data, FileOrderVolume.NIFTI_CODES[a.dtype.str[1:]],
                                         vol.slope, vol.inter, nvox, L.lptr(vox), L.iptr(K), L.bptr(csf_a), L.bptr(ear_a),
                                         L.dptr(pk), int(maxfasc), int(csf_on), int(ear_on),
                                         L.dptr(sc) if sc is not None else None, L.dptr(se) if se is not None else None,
                                         int(E), V, L.dptr(out)))
    return out


def fit_batch(plan, Y, K, csf, ear, peaks, maxfasc, csf_on, ear_on, sig_csf=None, sig_ear=None, E=0, rows=None):
    """Host-buffer voxel loop (mfx_fit_batch_rows): returns params_in_mask [V x num_params] (mf.py:1018-1028).

    ``rows`` (int64 [V], optional): voxel v's signal is ``Y[rows[v]]`` -- the ROI gather ``data[mask > 0]`` of the
    reference (mf.py:644) done by the library while it stages the upload; Y then is the whole [n_vox_total x M] volume."""
    Y = L.f64c(Y)
    if Y.ndim != 2:
        raise ValueError("Y should be a 2-D array [voxels x measurements]")
    M = Y.shape[1]
    if rows is not None:
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        V = rows.shape[0]
        if V and (rows.min() < 0 or rows.max() >= Y.shape[0]):
            raise ValueError("rows out of range")
    else:
        V = Y.shape[0]
    if M != plan.M:
        raise ValueError("data has %d measurements, protocol has %d" % (M, plan.M))
    K = np.ascontiguousarray(K, dtype=np.int32)
    if K.shape[0] != V:
        raise ValueError("K should have one entry per voxel")
    csf_a = np.ascontiguousarray(csf, dtype=np.uint8) if csf is not None else np.zeros(V, np.uint8)
    ear_a = np.ascontiguousarray(ear, dtype=np.uint8) if ear is not None else np.zeros(V, np.uint8)
    pk = L.f64c(peaks).reshape(V, -1) if maxfasc > 0 else np.zeros((V, 3))
    if maxfasc > 0 and pk.shape[1] != 3 * maxfasc:
        raise ValueError("peaks should have %d columns" % (3 * maxfasc))
    out = np.zeros((V, num_params(maxfasc, csf_on, ear_on)))
    sc = L.f64c(sig_csf) if sig_csf is not None else None
    se = L.f64c(sig_ear) if sig_ear is not None else None
    L.check(L.lib().mfx_fit_batch_rows(plan.handle(), L.dptr(Y), L.lptr(rows) if rows is not None else None, L.iptr(K),
                                       L.bptr(csf_a), L.bptr(ear_a), L.dptr(pk), int(maxfasc), int(csf_on), int(ear_on),
                                       L.dptr(sc) if sc is not None else None, L.dptr(se) if se is not None else None,
                                       int(E), V, L.dptr(out)))
    return out


def fit_batch_dev(plan, d_Y, d_peaks, maxfasc, csf_on=False, ear_on=False, d_sig_csf=None, d_sig_ear=None, E=0,
                  out=None, check=True):
    """Device-resident voxel loop (mfx_fit_batch_dev) on torch CUDA tensors of ONE voxel class (every voxel:
    K == maxfasc, csf == csf_on, ear == ear_on).  Enqueues on torch's current stream and returns the [V x num_params]
    output tensor without waiting; ``check=True`` then waits and raises what the kernels flagged (a fascicle
    direction that is not a unit vector: the reference's per-voxel ValueError, mf_utils.py:1798-1802)."""
    import torch
    assert d_Y.is_cuda and d_Y.dtype == torch.float64 and d_Y.is_contiguous()
    V = d_Y.shape[0]
    if d_Y.shape[1] != plan.M:
        raise ValueError("data has %d measurements, protocol has %d" % (d_Y.shape[1], plan.M))
    if maxfasc > 0:
        assert d_peaks.is_cuda and d_peaks.dtype == torch.float64 and d_peaks.is_contiguous()
        if tuple(d_peaks.shape) != (V, 3 * maxfasc):
            raise ValueError("peaks should have shape (%d, %d)" % (V, 3 * maxfasc))
    if out is None:
        out = torch.zeros((V, num_params(maxfasc, csf_on, ear_on)), dtype=torch.float64, device=d_Y.device)
    st = torch.cuda.current_stream(d_Y.device).cuda_stream
    L.check(L.lib().mfx_fit_batch_dev(plan.handle(), d_Y.data_ptr(), d_peaks.data_ptr() if maxfasc > 0 else None,
                                      int(maxfasc), int(bool(csf_on)), int(bool(ear_on)),
                                      d_sig_csf.data_ptr() if d_sig_csf is not None else None,
                                      d_sig_ear.data_ptr() if d_sig_ear is not None else None, int(E), V,
                                      out.data_ptr(), st))
    if check:
        L.check(L.lib().mfx_plan_status(plan.handle(), st))
    return out


def rotate_columns_dev(plan, d_dirs, d_cols, normalise=False):
    """Device-resident single-atom rotation: torch CUDA tensors dirs [B,3] f64, cols [B] i32 -> [B,M] f64."""
    import torch
    assert d_dirs.is_cuda and d_dirs.dtype == torch.float64 and d_dirs.is_contiguous()
    cols = d_cols.to(torch.int32).contiguous()
    B = d_dirs.shape[0]
    out = torch.empty((B, plan.M), dtype=torch.float64, device=d_dirs.device)
    st = torch.cuda.current_stream(d_dirs.device).cuda_stream
    L.check(L.lib().mfx_rotate_cols_dev(plan.handle(), d_dirs.data_ptr(), cols.data_ptr(), B, int(normalise),
                                        out.data_ptr(), st))
    return out


def cleanup_select(f1, f2, p1, p2, cos_min, ratio, w_keep, w_small, device=0):
    """Voxel loop of cleanup_2fascicles (mfx_cleanup_2fascicles; ref mf.py:170-335): weights f1, f2 [n] and directions p1, p2
    [n x 3] of the ROI voxels -> (peaks [n x 6], count [n])."""
    f1, f2, p1, p2 = L.f64c(f1), L.f64c(f2), L.f64c(p1), L.f64c(p2)
    n = f1.shape[0]
    if f2.shape != (n,) or p1.shape != (n, 3) or p2.shape != (n, 3):
        raise ValueError("cleanup_select: f1, f2 should have shape (n,), p1, p2 (n, 3)")
    peaks = np.zeros((n, 6))
    count = np.zeros(n)
    L.check(L.lib().mfx_cleanup_2fascicles(L.dptr(f1), L.dptr(f2), L.dptr(p1), L.dptr(p2), n, float(cos_min), float(ratio),
                                           float(w_keep), float(w_small), L.dptr(peaks), L.dptr(count), int(device)))
    return peaks, count
