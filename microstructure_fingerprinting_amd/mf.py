"""DIPY-style user API: ``MFModel(dictionary).fit(...) -> MFModelFit``.

Mirror of the reference's ``microstructure_fingerprinting/mf.py`` (cited as ``ref``) for the fitting
path: same constructor, same ``fit`` signature / accepted inputs / exceptions, same ``params`` layout
and the same attributes on the fit object -- but the per-voxel loop of ``ref:976-1032`` (serial or
``multiprocessing.Pool`` over ``_fit_voxel``, ``ref:340-461``) is ONE batched call into the HIP
library (``engine.fit_batch`` -> ``mfx_fit_batch``); ``parallel=True`` shards the ROI over all
visible GPUs instead of over CPU processes.
"""
import os
import threading
import time

import numpy as np

from . import _lib as L
from . import dist as mdist
from . import engine
from . import mf_utils as mfu
from . import nifti


def _load_volume(x):
    """str -> (array, affine) from a NIfTI file; ndarray -> (x, None)."""
    if isinstance(x, str):
        return nifti.load(x)
    return x, None


# ---------------------------------------------------------------------------------------------
# peak clean-up ahead of the fit (post-processing of a 2-tensor / 2-peak estimate)
# ---------------------------------------------------------------------------------------------
CLEANUP_RATIO = 2.5        # dominant/minor weight ratio beyond which the minor fascicle is dropped ...
CLEANUP_W_KEEP = 0.20      # ... unless its own weight reaches this value
CLEANUP_W_SMALL = 0.075    # absolute weight under which a fascicle is always dropped
CLEANUP_ANG_MIN = 15       # crossing angle [deg] under which two peaks are merged


def _directions_from(mu, peakmode):
    """(n, 2|3|6) orientation descriptors -> (n, 3) direction vectors."""
    if peakmode == 'colat_longit':
        st = np.sin(mu[..., 0])
        return np.stack([st * np.cos(mu[..., 1]), st * np.sin(mu[..., 1]), np.cos(mu[..., 0])], axis=-1)
    if peakmode == 'peaks':
        return np.array(mu, dtype=np.float64)
    return mfu.DT_vec_to_peaks(mu, 'column')


def cleanup_2fascicles(frac1, frac2, peakmode, mu1, mu2, mask, frac12=None):
    """Select 0, 1 or 2 of two detected fascicle orientations per voxel (same arguments, thresholds and
    results as ref mf.py:36-335; NeuroImage 184 (2019) 964-980).

    Per mask voxel, in this order: peaks closer than 15 deg are merged into population 0 (sign-aware sum,
    weights added); a population more than 2.5x lighter than the other one and lighter than 0.20 is
    dropped (population 1 moves to slot 0 when population 0 goes); populations lighter than 0.075 are
    dropped; the survivors are ordered by descending weight.  Returns ``(peaks_out, num_fasc_out)`` with
    shapes ``mask.shape + (6,)`` and ``mask.shape``."""
    if (frac1 is None or frac2 is None) and frac12 is None:
        raise ValueError("If fractions of first and second fascicles set to None, argument frac12 is required "
                         "to specify both fractions simultanously. A total of 6 arguments should be passed, not 5.")
    mask = _load_volume(mask)[0]
    frac1 = _load_volume(frac1)[0]
    frac2 = _load_volume(frac2)[0]
    if frac12 is not None:
        frac12 = _load_volume(frac12)[0]
        if frac12.shape[-1] < 2:
            raise ValueError("Last dimension of frac12 should have size at least 2.")
        if frac12.shape[mask.ndim] == 1:           # (nx, ny, nz, 1, 2)
            frac1, frac2 = frac12[..., 0, 0], frac12[..., 0, 1]
        else:
            frac1, frac2 = frac12[..., 0], frac12[..., 1]
    if frac1.shape != mask.shape:
        raise ValueError("frac1 should have the same shape as mask")
    if frac2.shape != mask.shape:
        raise ValueError("frac2 should have the same shape as mask")
    mu1 = _load_volume(mu1)[0]
    mu2 = _load_volume(mu2)[0]
    width = {'colat_longit': 2, 'peaks': 3, 'tensor': 6}.get(peakmode)
    if width is None:
        raise ValueError('Unknown peak mode %s' % peakmode)
    if peakmode == 'tensor':                        # tensor files are often (nx, ny, nz, 1, 6)
        if mu1.shape[mask.ndim] == 1:
            mu1 = mu1[..., 0, :]
        if mu2.shape[mask.ndim] == 1:
            mu2 = mu2[..., 0, :]
    if mu1.shape[-1] != width or mu2.shape[-1] != width:
        raise ValueError("In '%s' peak mode, last dimension of mu1 and mu2 should have size %d. Detected %d and %d."
                         % (peakmode, width, mu1.shape[-1], mu2.shape[-1]))

    roi = mask > 0
    f1 = np.ascontiguousarray(frac1[roi], dtype=np.float64)
    f2 = np.ascontiguousarray(frac2[roi], dtype=np.float64)
    p1 = np.ascontiguousarray(_directions_from(mu1[roi], peakmode), dtype=np.float64)
    p2 = np.ascontiguousarray(_directions_from(mu2[roi], peakmode), dtype=np.float64)
    # the voxel loop (ref:170-335) runs on the device, one thread per ROI voxel (csrc/cleanup.hip)
    peaks, count = engine.cleanup_select(f1, f2, p1, p2, np.cos(CLEANUP_ANG_MIN * np.pi / 180), CLEANUP_RATIO,
                                         CLEANUP_W_KEEP, CLEANUP_W_SMALL)

    peaks_out = np.zeros(mask.shape + (6,))
    peaks_out[roi] = peaks
    num_fasc_out = np.zeros(mask.shape)
    num_fasc_out[roi] = count
    return peaks_out, num_fasc_out


class MFModel():
    r"""Microstructure Fingerprinting model (ref:464-1051)."""
    MAX_FASC = 2          # ref:467
    SHARD_DEVICES = None  # parallel=True: device of every shard; None: each visible GPU once (a device may be named twice)
    MAX_PROG_LINES = 100  # ref:468 (kept for API compatibility; progress is per batch here)
    DFT_DISP_ITVL = 5     # ref:469

    def __init__(self, dictionary, device=0):
        if isinstance(dictionary, str):
            self.dic = mfu.loadmat(dictionary)
        elif isinstance(dictionary, dict):
            self.dic = dictionary
        else:
            raise ValueError("Dictionary should either be a valid path to a Matlab-like mat file or a "
                             "Python dictionary.")
        self.device = device
        # per-shell knot tables, built once on the host (ref:506-509); uploaded to HBM on first use
        self.ms_interpolator = mfu.init_PGSE_multishell_interp(self.dic['dictionary'], self.dic['sch_mat'],
                                                               self.dic['orientation'], device=device)
        print("Initiated model based on dictionary with %d single-fascicle fingerprint(s) and %d "
              "fingerprint(s) for the extra-axonal restricted (EAR) compartment."
              % (self.dic['num_atom'], self.dic['num_ear']))

    # ------------------------------------------------------------------------------------------
    def fit(self, data, mask, numfasc, *, peaks=None, colat_longit=None, tensors=None, pgse_scheme=None,
            bvals=None, bvecs=None, csf_mask=None, ear_mask=None, verbose=1, parallel=False):
        r"""Fingerprinting on the pre-computed dictionary (ref:516-1051; same arguments)."""
        VRB = verbose
        nii_affine = None
        t0 = time.time()
        if isinstance(data, str) and VRB >= 2:
            print("Loading data from file %s..." % data)
        # a volume in file order (a NIfTI file, or nibabel's Fortran-ordered array) stays as it is: conversion to
        # float64, the header's scaling and the ROI gather (ref:623-657) run on the device (mfx_fit_batch_volume)
        vol = None
        if isinstance(data, str):
            raw, slope, inter, aff = nifti.load_raw(data)
            if engine.FileOrderVolume.accepts(raw):
                vol = engine.FileOrderVolume(raw, slope, inter)
                data_arr = raw
            else:
                data_arr = np.array(raw, dtype=np.float64)
                if slope != 0.0 and not (slope == 1.0 and inter == 0.0) and np.isfinite(slope):
                    data_arr = data_arr * slope + inter
        else:
            data_arr, aff = data, None
            if engine.FileOrderVolume.accepts(data_arr):
                vol = engine.FileOrderVolume(data_arr)
        nii_affine = aff
        if isinstance(data, str) and VRB >= 2:
            print("Data loaded in %g s." % (time.time() - t0))
        mask_arr, aff = _load_volume(mask)
        if nii_affine is None:
            nii_affine = aff
        img_shape = mask_arr.shape
        roi = mask_arr > 0
        roi_index = np.flatnonzero(roi.reshape(-1))     # ROI order == np.where(mask > 0) (C order)
        ROI_size = int(roi_index.shape[0])
        _fidx = []

        def file_order_roi():   # the ROI voxels' positions inside one 3-D image of a file-order volume (computed once)
            if not _fidx:
                _fidx.append(np.ravel_multi_index(np.unravel_index(roi_index, img_shape), img_shape, order='F').astype(np.int64))
            return _fidx[0]

        def roi_rows(arr):      # arr[mask > 0] of a (grid x n) volume as float64 rows
            if engine.FileOrderVolume.accepts(arr) and arr.shape[:-1] == img_shape and ROI_size >= 4096:
                # a file-order (Fortran) volume: indexing it with the C-ordered mask, or reshaping it to C-order rows,
                # first copies the whole volume on one host core (0.2-0.6 s at 1e6 ROI voxels) - the device gathers it
                # like the data (mfx_volume_rows)
                return engine.volume_rows(engine.FileOrderVolume(arr), file_order_roi(), device=self.ms_interpolator.device)
            return np.asarray(arr.reshape(-1, arr.shape[-1])[roi_index], dtype=np.float64)
        if ROI_size == 0:
            raise ValueError("No voxel detected in mask. Please provide a non-empty mask.")
        if data_arr.shape[:-1] != img_shape:
            raise ValueError("Data and mask not compatible. Based on data, mask should have shape (%s), got (%s) "
                             "instead." % (" ".join("%d" % x for x in data_arr.shape[:-1]),
                                           " ".join("%d" % x for x in img_shape)))
        # ---- number of fascicles (ref:660-687)
        if np.isscalar(numfasc) and not isinstance(numfasc, str):
            numfasc_roi = np.full(ROI_size, numfasc, dtype=int)
        else:
            nf, _ = _load_volume(numfasc)
            if mask_arr.shape != nf.shape:
                raise ValueError("Data and argument numfasc not compatible.  Based on data, numfasc should have "
                                 "shape (%s), got (%s) instead." % (" ".join("%d" % x for x in img_shape),
                                                                   " ".join("%d" % x for x in nf.shape)))
            numfasc_roi = nf[roi].astype(int)
        maxfasc = int(np.max(numfasc_roi))
        if maxfasc > MFModel.MAX_FASC:
            raise ValueError("Detected %d mask voxel(s) in numfasc with number of axon populations greater than "
                             "allowed maximum of %d." % (np.sum(numfasc_roi > MFModel.MAX_FASC), MFModel.MAX_FASC))
        # ---- fascicle directions: peaks | colat_longit | tensors (ref:693-815)
        if peaks is not None:
            pk, aff = _load_volume(peaks)
            if nii_affine is None:
                nii_affine = aff
            if pk.shape[:-1] != img_shape:
                raise ValueError("Arg. peaks not compatible. Based on data, it should have shape (%s x), with x a "
                                 "multiple of 3. Got (%s) instead." % (" ".join("%d" % x for x in img_shape),
                                                                       " ".join("%d" % x for x in pk.shape)))
            if pk.shape[-1] % 3 != 0:
                raise ValueError("Size of last dimension of arg. peaks should be a multiple of 3, got %d instead."
                                 % pk.shape[-1])
            if pk.shape[-1] > maxfasc * 3 and VRB >= 1:
                print("Ignoring last %d value(s) along last dimension of peaks, as max number of axon populations "
                      "in mask is %d." % (pk.shape[-1] - maxfasc * 3, maxfasc))
            peaks_roi = np.ascontiguousarray(roi_rows(np.asarray(pk))[:, :3 * maxfasc], dtype=np.float64)
        elif colat_longit is not None or tensors is not None:
            arg = colat_longit if colat_longit is not None else tensors
            dims = ((2,),) if colat_longit is not None else ((6,), (1, 6))
            arg = arg if isinstance(arg, list) else [arg]
            peaks_roi = np.zeros((ROI_size, 3 * len(arg)))
            if len(arg) > maxfasc and VRB >= 1:
                print("Ignoring %d peak orientation argument(s) because max number of axon populations in mask "
                      "is %d." % (len(arg) - maxfasc, maxfasc))
            for i in range(min(len(arg), maxfasc)):
                a_i, aff = _load_volume(arg[i])
                if nii_affine is None:
                    nii_affine = aff
                if a_i.shape not in [img_shape + d for d in dims]:
                    want = " or ".join("(" + " ".join("%d" % x for x in img_shape + d) + ")" for d in dims)
                    raise ValueError("Peak orientation arg. %d of %d seems incompatible. Based on data, it should "
                                     "have shape %s, got (%s) instead."
                                     % (i + 1, len(arg), want, " ".join("%d" % x for x in a_i.shape)))
                if colat_longit is not None:
                    ang = roi_rows(a_i)
                    th, ph = ang[:, 0], ang[:, 1]
                    peaks_roi[:, 3 * i + 0] = np.sin(th) * np.cos(ph)
                    peaks_roi[:, 3 * i + 1] = np.sin(th) * np.sin(ph)
                    peaks_roi[:, 3 * i + 2] = np.cos(th)
                else:
                    if a_i.shape[mask_arr.ndim] == 1:
                        a_i = a_i[(slice(None),) * mask_arr.ndim + (0, slice(None))]
                    # NIfTI 'column' order of the upper triangle; principal eigenvector, 0 for zero tensors
                    peaks_roi[:, 3 * i:3 * i + 3] = mfu.DT_vec_to_peaks(roi_rows(a_i), 'column')
            peaks_roi = np.ascontiguousarray(peaks_roi[:, :3 * maxfasc])
            if peaks_roi.shape[1] < 3 * maxfasc:
                peaks_roi = np.concatenate([peaks_roi, np.zeros((ROI_size, 3 * maxfasc - peaks_roi.shape[1]))], axis=1)
        else:
            raise RuntimeError("At least one of peaks, colat_longit and tensors must be specified.")
        for k in range(maxfasc):   # missing peak where numfasc demands one (ref:803-815)
            zero_k = ~np.any(peaks_roi[:, 3 * k:3 * k + 3], axis=1)
            n0 = int(np.count_nonzero(zero_k & (numfasc_roi >= k + 1))) if zero_k.any() else 0
            if n0 > 0:
                raise ValueError("Detected %d voxel(s) in which the main orientation of axon population %d/%d was a "
                                 "zero vector, although numfasc specifies the presence of that population."
                                 % (n0, k + 1, maxfasc))
        # ---- protocol (ref:821-846)
        if pgse_scheme is not None:
            if isinstance(pgse_scheme, str):
                pgse_scheme = np.loadtxt(pgse_scheme, skiprows=1)
            if pgse_scheme.shape[1] != 7:
                raise ValueError("pgse_scheme should have 7 columns,  detected %d instead." % (pgse_scheme.shape[1],))
        else:
            if bvals is None or bvecs is None:
                raise TypeError("If no schemefile is provided, then both bvals and bvecs must be specified.")
            pgse_scheme = mfu.get_PGSE_scheme_from_bval_bvec_dense(self.dic['sch_mat'], bvals, bvecs, 1e-3)
        pgse_scheme = np.ascontiguousarray(pgse_scheme, dtype=np.float64)
        num_seq = pgse_scheme.shape[0]
        gam = mfu.get_gyromagnetic_ratio('H')
        G, Delta, delta, TE = pgse_scheme[:, 3], pgse_scheme[:, 4], pgse_scheme[:, 5], pgse_scheme[:, 6]
        b = (gam * G * delta) ** 2 * (Delta - delta / 3)
        # ---- optional compartments (ref:852-894)
        csf_mask, aff = self._roi_flags(csf_mask, roi, img_shape, ROI_size, "csf_mask")
        if nii_affine is None:
            nii_affine = aff
        ear_mask, aff = self._roi_flags(ear_mask, roi, img_shape, ROI_size, "ear_mask")
        if nii_affine is None:
            nii_affine = aff
        csf_on = bool(np.any(csf_mask > 0))
        ear_on = bool(np.any(ear_mask > 0))
        n_empty = int(np.sum((numfasc_roi + csf_mask + ear_mask) == 0))
        if n_empty > 0 and VRB >= 2:
            print("WARNING: detected %d voxel(s) in mask with zero  axon population, no cerebrospinal fluid (CSF) "
                  "and no extra-axonal restricted (EAR) compartment specified. No estimation will be performed "
                  "there." % (n_empty,))
        sig_csf = sig_ear = None
        num_ear = int(self.dic['num_ear'])
        if csf_on:   # ref:918-920
            sig_csf = np.exp(-TE / self.dic['T2_csf']) * np.exp(-b * self.dic['DIFF_csf'])
        if ear_on:   # ref:921-925
            DIFF_ear = np.atleast_1d(self.dic['DIFF_ear'])
            sig_ear = np.zeros((num_seq, num_ear))
            for i in range(num_ear):
                sig_ear[:, i] = np.exp(-TE / self.dic['T2_ear']) * np.exp(-b * DIFF_ear[i])
        if data_arr.shape[-1] != num_seq:
            raise ValueError("Data has %d measurements per voxel but the protocol has %d." % (data_arr.shape[-1],
                                                                                                num_seq))
        # ---- what the reference checks in every voxel with a fascicle, through interp_PGSE_from_multishell
        # (mf_utils.py:1786-1789 and 1804-1807), checked once per fit here: the protocol's timing must be the
        # dictionary's, its gradient directions zero or unit vectors
        if maxfasc > 0:
            if not np.all(np.isclose(self.ms_interpolator['scheme_DeldelTE'], pgse_scheme[:, 4:7])):
                raise ValueError("Delta, delta and TE values should all be identical to those in the multi-shell "
                                 "sampling.")
            mfu._check_gnorms(pgse_scheme)
        # ---- the voxel loop, batched on the device (replaces ref:976-1032)
        # ROI order == np.where(mask > 0).  A float64 C-contiguous volume is handed over as it is with the ROI's row
        # numbers: the library gathers the rows while it stages the upload (the reference's data[mask > 0], ref:644)
        if vol is not None:
            Y = vol
            rows = file_order_roi()
        elif isinstance(data_arr, np.ndarray) and data_arr.dtype == np.float64 and data_arr.flags.c_contiguous:
            Y = data_arr.reshape(-1, num_seq)
            rows = roi_index.astype(np.int64, copy=False)
        else:
            Y = np.ascontiguousarray(data_arr[roi], dtype=np.float64)
            rows = None
        st = time.time()
        if VRB >= 2:
            print("Starting estimation in %d voxel(s) on the GPU%s." % (ROI_size, "s (sharded)" if parallel else ""))
        args = (numfasc_roi, csf_mask, ear_mask, peaks_roi, maxfasc, csf_on, ear_on, sig_csf, sig_ear, num_ear)
        devs = list(range(L.lib().mfx_device_count())) if self.SHARD_DEVICES is None else list(self.SHARD_DEVICES)
        if parallel and len(devs) > 1 and ROI_size >= 2 * len(devs):
            params_in_mask = self._fit_sharded(pgse_scheme, Y, rows, args, devs)
        else:
            plan = self.ms_interpolator.plan_for(pgse_scheme)
            params_in_mask = (engine.fit_batch_volume(plan, Y, rows, *args) if vol is not None
                              else engine.fit_batch(plan, Y, *args, rows=rows))
        if VRB >= 2:
            print("Estimation performed in %g second(s)." % (time.time() - st))
        fitinfo = {'maxfasc': maxfasc, 'csf_on': csf_on, 'ear_on': ear_on, 'affine': nii_affine, 'mask': mask_arr,
                   'fasc_propnames': [x.strip() for x in self.dic['fasc_propnames']], 'peaks_roi': peaks_roi,
                   'roi_index': roi_index}
        for n in fitinfo['fasc_propnames']:
            fitinfo['_dict_' + n] = self.dic[n]
        if ear_on:
            fitinfo['DIFF_ear'] = np.atleast_1d(self.dic['DIFF_ear'])
        return MFModelFit(fitinfo, params_in_mask, verbose=VRB)

    @staticmethod
    def _roi_flags(m, roi, img_shape, ROI_size, name):
        """csf_mask / ear_mask argument -> bool[ROI_size] (ref:852-894)."""
        aff = None
        if m is None:
            return np.zeros(ROI_size, dtype=bool), aff
        if np.isscalar(m) and not isinstance(m, str):
            return np.full(ROI_size, m > 0, dtype=bool), aff
        m, aff = _load_volume(m)
        if m.shape != img_shape:
            raise ValueError("Arg. %s incomptabible. Based on data, it should have shape (%s), detected (%s) instead."
                             % (name, " ".join("%d" % x for x in img_shape), " ".join("%d" % x for x in m.shape)))
        return (m[roi] > 0), aff

    def _fit_sharded(self, pgse_scheme, Y, rows, args, devs):
        """parallel=True: one host thread per GPU (ctypes releases the GIL), each device with its own copy of the
        tables.  The voxels of every class (numfasc, CSF, EAR: their cost differs by up to 40x) are dealt round-robin
        over the devices so that each gets the same mix (reference: mp.Pool over voxels, ref:978-1009).
        (Multi-process / multi-node runs use dist.py.)"""
        numfasc_roi, csf_mask, ear_mask, peaks_roi, maxfasc, csf_on, ear_on, sig_csf, sig_ear, num_ear = args
        V = numfasc_roi.shape[0]
        ndev = len(devs)
        out = [None] * ndev
        idx = [mdist.balanced_shard_indices(numfasc_roi, csf_mask, ear_mask, d, ndev) for d in range(ndev)]
        errs = []

        def work(d):
            try:
                ix = idx[d]
                ms = mfu.MultiShellInterpolator(self.ms_interpolator['scheme_DeldelTE'], self.ms_interpolator['Gms_un'],
                                                self.ms_interpolator['interpolators'], device=devs[d])
                part = (numfasc_roi[ix], csf_mask[ix], ear_mask[ix], peaks_roi[ix], maxfasc, csf_on, ear_on, sig_csf,
                        sig_ear, num_ear)
                if isinstance(Y, engine.FileOrderVolume):
                    out[d] = engine.fit_batch_volume(ms.plan_for(pgse_scheme), Y, rows[ix], *part)
                else:
                    out[d] = engine.fit_batch(ms.plan_for(pgse_scheme), Y, *part, rows=(ix if rows is None else rows[ix]))
            except Exception as e:   # re-raised below, like pool.get() (ref:1006-1008)
                errs.append(e)
        th = [threading.Thread(target=work, args=(d,)) for d in range(ndev)]
        [t.start() for t in th]
        [t.join() for t in th]
        if errs:
            raise errs[0]
        full = np.zeros((V, out[0].shape[1]))
        for d in range(ndev):
            full[idx[d]] = out[d]
        return full


class MFModelFit():
    """Fit object: one ndarray attribute per estimated map + ``param_names`` (ref:1054-1229)."""

    def __init__(self, fitinfo, model_params, verbose=0):
        self.affine = fitinfo['affine']
        nf, csf_on, ear_on, mask = fitinfo['maxfasc'], fitinfo['csf_on'], fitinfo['ear_on'], fitinfo['mask']
        ROI_size = model_params.shape[0]
        flat = fitinfo.get('roi_index')            # flat indices of the ROI voxels (== np.where(mask > 0) order)
        if flat is None:
            flat = np.flatnonzero(np.asarray(mask) > 0)
        assert ROI_size == flat.shape[0], 'Inconsistent mask and model parameter array'
        self.params_in_mask = model_params
        whole = ROI_size == int(np.prod(mask.shape))

        def to_map(vals, extra=()):
            if whole:     # every voxel is in the ROI: the map is the parameter column itself, reshaped
                return np.array(vals, dtype=np.float64).reshape(mask.shape + tuple(extra))
            m = np.zeros(mask.shape + tuple(extra))
            m.reshape((-1,) + tuple(extra))[flat] = vals
            return m
        names = ['M0']
        self.M0 = to_map(model_params[:, 0])
        for k in range(nf):
            setattr(self, 'frac_f%d' % k, to_map(model_params[:, k + 1]))
            setattr(self, 'peak_f%d' % k, to_map(fitinfo['peaks_roi'][:, 3 * k:3 * (k + 1)], (3,)))
            names += ['frac_f%d' % k, 'peak_f%d' % k]
        IDs = [model_params[:, 1 + nf + k].astype(np.intp) for k in range(nf)]
        active = [model_params[:, k + 1] > 0 for k in range(nf)]
        for prop in fitinfo['fasc_propnames']:     # per-fascicle properties and nu-weighted totals (ref:1106-1129)
            tot = np.zeros(ROI_size)
            for k in range(nf):
                nu_k = model_params[:, k + 1]
                prop_k = fitinfo['_dict_' + prop][IDs[k]] * active[k]
                tot += nu_k * prop_k
                setattr(self, prop + '_f%d' % k, to_map(prop_k))
                names.append(prop + '_f%d' % k)
            setattr(self, prop + '_tot', to_map(tot))
            names.append(prop + '_tot')
        if csf_on:
            self.frac_csf = to_map(model_params[:, 2 * nf + 1])
            names.append('frac_csf')
        if ear_on:
            nu_e = model_params[:, 2 * nf + csf_on + 1]
            self.frac_ear = to_map(nu_e)
            ID_e = model_params[:, 2 * nf + csf_on + 2].astype(int)
            self.D_ear = to_map(fitinfo['DIFF_ear'][ID_e] * (nu_e > 0))
            names += ['frac_ear', 'D_ear']
        self.MSE = to_map(model_params[:, -2])
        self.R2 = to_map(model_params[:, -1])
        names += ['MSE', 'R2']
        self.param_names = names
        if verbose >= 2:
            print("Microstructure Fingerprinting fit object constructed; maps: %s" % ", ".join(names))

    def write_nifti(self, output_basename, affine=None):
        """One NIfTI file per parameter map, ``<stem>_<param><ext>``; returns the file names (reference mf.py:1177-1229).
        ``output_basename`` may end in .nii.gz (kept), .nii or nothing (both give .nii); any other extension is refused."""
        xfm = self.affine if affine is None else affine
        if xfm is None:
            raise ValueError("Argument affine must be explicitely passed  because no affine transform matrix was "
                             "found during model fitting. Expecting NumPy array with shape (4, 4).")
        stem, ext = _nifti_stem(output_basename)
        written = []
        for name in self.param_names:
            target = "%s_%s%s" % (stem, name, ext)
            nifti.save(getattr(self, name), xfm, target)
            written.append(target)
        return written


def _nifti_stem(output_basename):
    """('dir/name', '.nii' | '.nii.gz') of an output name for MFModelFit.write_nifti."""
    if output_basename.endswith('.nii.gz') and len(output_basename) > len('.nii.gz'):
        return output_basename[:-len('.nii.gz')], '.nii.gz'
    stem, ext = os.path.splitext(output_basename)
    if ext and ext != '.nii':
        raise ValueError("Unknown NIfTI extension %s in output %s" % (ext, output_basename))
    return stem, '.nii'
