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


class MFModel():
    r"""Microstructure Fingerprinting model (ref:464-1051)."""
    MAX_FASC = 2          # ref:467
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
        data_arr, aff = _load_volume(data)
        nii_affine = aff
        if isinstance(data, str) and VRB >= 2:
            print("Data loaded in %g s." % (time.time() - t0))
        mask_arr, aff = _load_volume(mask)
        if nii_affine is None:
            nii_affine = aff
        img_shape = mask_arr.shape
        roi = mask_arr > 0
        ROI_size = int(np.sum(roi))
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
            peaks_roi = np.ascontiguousarray(pk[roi, :3 * maxfasc], dtype=np.float64)
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
                    th, ph = a_i[roi, 0], a_i[roi, 1]
                    peaks_roi[:, 3 * i + 0] = np.sin(th) * np.cos(ph)
                    peaks_roi[:, 3 * i + 1] = np.sin(th) * np.sin(ph)
                    peaks_roi[:, 3 * i + 2] = np.cos(th)
                else:
                    if a_i.shape[mask_arr.ndim] == 1:
                        a_i = a_i[(slice(None),) * mask_arr.ndim + (0, slice(None))]
                    v = a_i[roi, :]       # NIfTI 'column' order of the upper triangle: xx xy yy xz yz zz
                    T = np.zeros((ROI_size, 3, 3))
                    T[:, 0, 0], T[:, 0, 1], T[:, 0, 2] = v[:, 0], v[:, 1], v[:, 3]
                    T[:, 1, 0], T[:, 1, 1], T[:, 1, 2] = v[:, 1], v[:, 2], v[:, 4]
                    T[:, 2, 0], T[:, 2, 1], T[:, 2, 2] = v[:, 3], v[:, 4], v[:, 5]
                    d, eigv = np.linalg.eigh(T)
                    nz = (np.abs(d)[..., -1] > 0)[:, np.newaxis]
                    peaks_roi[:, 3 * i:3 * i + 3] = eigv[..., -1] * nz   # principal eigenvector, 0 for zero tensors
            peaks_roi = np.ascontiguousarray(peaks_roi[:, :3 * maxfasc])
            if peaks_roi.shape[1] < 3 * maxfasc:
                peaks_roi = np.concatenate([peaks_roi, np.zeros((ROI_size, 3 * maxfasc - peaks_roi.shape[1]))], axis=1)
        else:
            raise RuntimeError("At least one of peaks, colat_longit and tensors must be specified.")
        for k in range(maxfasc):   # missing peak where numfasc demands one (ref:803-815)
            l1 = np.sum(np.abs(peaks_roi[numfasc_roi >= k + 1, 3 * k:3 * k + 3]), axis=1)
            n0 = int(np.sum(l1 == 0))
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
        # ---- the voxel loop, batched on the device (replaces ref:976-1032)
        Y = np.ascontiguousarray(data_arr[roi], dtype=np.float64)     # ROI order == np.where(mask > 0)
        st = time.time()
        if VRB >= 2:
            print("Starting estimation in %d voxel(s) on the GPU%s." % (ROI_size, "s (sharded)" if parallel else ""))
        args = (numfasc_roi, csf_mask, ear_mask, peaks_roi, maxfasc, csf_on, ear_on, sig_csf, sig_ear, num_ear)
        ndev = L.lib().mfx_device_count()
        if parallel and ndev > 1 and ROI_size >= 2 * ndev:
            params_in_mask = self._fit_sharded(pgse_scheme, Y, args, ndev)
        else:
            plan = self.ms_interpolator.plan_for(pgse_scheme)
            params_in_mask = engine.fit_batch(plan, Y, *args)
        if VRB >= 2:
            print("Estimation performed in %g second(s)." % (time.time() - st))
        fitinfo = {'maxfasc': maxfasc, 'csf_on': csf_on, 'ear_on': ear_on, 'affine': nii_affine, 'mask': mask_arr,
                   'fasc_propnames': [x.strip() for x in self.dic['fasc_propnames']], 'peaks_roi': peaks_roi}
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

    def _fit_sharded(self, pgse_scheme, Y, args, ndev):
        """parallel=True: contiguous ROI shards, one host thread per GPU (ctypes releases the GIL); each
        device gets its own copy of the tables.  (Multi-process / multi-node runs use dist.py.)"""
        numfasc_roi, csf_mask, ear_mask, peaks_roi, maxfasc, csf_on, ear_on, sig_csf, sig_ear, num_ear = args
        V = Y.shape[0]
        out = [None] * ndev
        errs = []

        def work(d):
            try:
                lo, hi = mdist.shard_range(V, d, ndev)
                ms = mfu.MultiShellInterpolator(self.ms_interpolator['scheme_DeldelTE'], self.ms_interpolator['Gms_un'],
                                                self.ms_interpolator['interpolators'], device=d)
                out[d] = engine.fit_batch(ms.plan_for(pgse_scheme), Y[lo:hi], numfasc_roi[lo:hi], csf_mask[lo:hi],
                                          ear_mask[lo:hi], peaks_roi[lo:hi], maxfasc, csf_on, ear_on, sig_csf, sig_ear,
                                          num_ear)
            except Exception as e:   # re-raised below, like pool.get() (ref:1006-1008)
                errs.append(e)
        th = [threading.Thread(target=work, args=(d,)) for d in range(ndev)]
        [t.start() for t in th]
        [t.join() for t in th]
        if errs:
            raise errs[0]
        return np.concatenate(out, axis=0)


class MFModelFit():
    """Fit object: one ndarray attribute per estimated map + ``param_names`` (ref:1054-1229)."""

    def __init__(self, fitinfo, model_params, verbose=0):
        self.affine = fitinfo['affine']
        nf, csf_on, ear_on, mask = fitinfo['maxfasc'], fitinfo['csf_on'], fitinfo['ear_on'], fitinfo['mask']
        roi = mask > 0
        ROI_size = model_params.shape[0]
        assert ROI_size == np.sum(roi), 'Inconsistent mask and model parameter array'
        self.params_in_mask = model_params

        def to_map(vals, extra=()):
            m = np.zeros(mask.shape + tuple(extra))
            m[roi] = vals
            return m
        names = ['M0']
        self.M0 = to_map(model_params[:, 0])
        for k in range(nf):
            setattr(self, 'frac_f%d' % k, to_map(model_params[:, k + 1]))
            setattr(self, 'peak_f%d' % k, to_map(fitinfo['peaks_roi'][:, 3 * k:3 * (k + 1)], (3,)))
            names += ['frac_f%d' % k, 'peak_f%d' % k]
        for prop in fitinfo['fasc_propnames']:     # per-fascicle properties and nu-weighted totals (ref:1106-1129)
            tot = np.zeros(ROI_size)
            for k in range(nf):
                nu_k = model_params[:, k + 1]
                ID_k = model_params[:, 1 + nf + k].astype(int)
                prop_k = fitinfo['_dict_' + prop][ID_k] * (nu_k > 0)
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
        """Export every map as ``<basename>_<param>.nii[.gz]`` (ref:1177-1229); returns the file names."""
        if affine is None:
            affine = self.affine
        if affine is None:
            raise ValueError("Argument affine must be explicitely passed  because no affine transform matrix was "
                             "found during model fitting. Expecting NumPy array with shape (4, 4).")
        niigz = '.nii.gz'
        if len(output_basename) > len(niigz) and output_basename[-len(niigz):] == niigz:
            path, fname = os.path.split(output_basename[:-len(niigz)])
            ext = niigz
        else:
            path, tail = os.path.split(output_basename)
            fname, ext = os.path.splitext(tail)
            if ext not in ['', '.nii']:
                raise ValueError("Unknown NIfTI extension %s in output %s" % (ext, output_basename))
            ext = '.nii'
        base = os.path.join(path, fname)
        fnames = []
        for p in self.param_names:
            fn = '%s_%s%s' % (base, p, ext)
            nifti.save(getattr(self, p), affine, fn)
            fnames.append(fn)
        return fnames
