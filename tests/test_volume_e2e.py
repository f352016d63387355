"""Row N1 at size: NIfTI files in -> MFModel.fit (mixed K in {0, 1, 2}, CSF / EAR masks) -> write_nifti, with the
volume handed to the library in its file layout (mfx_fit_batch_volume: conversion, scaling and ROI gather of
mf.py:623-657 on the device).  Checked against the host-array path, the CPU restatement of the reference on a sample
of every voxel class, and the files written back."""
import os

import numpy as np
import pytest

Z = np.array([0.0, 0.0, 1.0])


def _model(N, E=4):
    import microstructure_fingerprinting_amd as mf
    from microstructure_fingerprinting_amd import synth
    sch, dic, rng = synth.make_model("C2", N=N)
    md = {"dictionary": dic, "sch_mat": sch, "orientation": Z, "num_atom": N, "num_ear": E, "T2_csf": 2.0,
          "DIFF_csf": 3e-9, "T2_ear": 0.08, "DIFF_ear": np.linspace(0.2e-9, 1.2e-9, E), "fasc_propnames": ["rad", "fin"],
          "rad": rng.uniform(0.2e-6, 2e-6, N), "fin": rng.uniform(0.2, 0.9, N)}
    return mf.MFModel(md), sch, rng


@pytest.mark.gpu
def test_volume_files_to_maps(tmp_path):
    import microstructure_fingerprinting_amd as mf
    from microstructure_fingerprinting_amd import nifti, synth, engine
    from oracle import oracle as orc
    model, sch, rng = _model(130)
    grid = (40, 36, 30)
    ph = synth.make_phantom(model, grid, rng)
    aff = np.diag([2.0, 2.0, 2.5, 1.0]); aff[:3, 3] = [-40, -36, -30]
    files = {}
    for k, a in ph.items():
        files[k] = str(tmp_path / (k + ".nii"))
        nifti.save(a, aff, files[k])
    np.savetxt(str(tmp_path / "scheme.txt"), sch, header="VERSION: 1", comments="")
    kw = dict(peaks=files["peaks"], pgse_scheme=str(tmp_path / "scheme.txt"), csf_mask=files["csf_mask"],
              ear_mask=files["ear_mask"], verbose=0)
    fit = model.fit(files["data"], files["mask"], files["numfasc"], **kw)
    V = fit.params_in_mask.shape[0]
    assert V == int(ph["mask"].sum()) and 0.45 < V / np.prod(grid) < 0.6
    # (1) the same volume as C-ordered float64 host arrays (the mfx_fit_batch_rows path): identical rows
    kwa = dict(peaks=ph["peaks"], pgse_scheme=sch, csf_mask=ph["csf_mask"], ear_mask=ph["ear_mask"], verbose=0)
    data64 = np.ascontiguousarray(ph["data"], dtype=np.float64)
    fit_a = model.fit(data64, ph["mask"], ph["numfasc"], **kwa)
    assert np.array_equal(fit.params_in_mask, fit_a.params_in_mask)
    assert fit.param_names == fit_a.param_names and np.allclose(fit.affine, aff)
    # (2) nibabel-style Fortran-ordered float64 array (get_fdata()) takes the volume path too
    fit_f = model.fit(np.asfortranarray(data64), ph["mask"], ph["numfasc"], **kwa)
    assert np.array_equal(fit_f.params_in_mask, fit.params_in_mask)
    # (2b) directions as colatitude / longitude volumes and as tensor volumes in file order: the device gathers them too
    # (mfx_volume_rows); the same rows as from C-ordered arrays
    pkv = ph["peaks"]
    cl = [np.stack([np.arccos(np.clip(pkv[..., 3 * k + 2], -1, 1)), np.arctan2(pkv[..., 3 * k + 1], pkv[..., 3 * k])], axis=-1) for k in range(2)]
    from microstructure_fingerprinting_amd import mf_utils as mfu
    dt = mfu.peaks_to_DT_vec(pkv.reshape(pkv.shape[:-1] + (2, 3)).copy(), 'column')          # one (grid x 6) array per fascicle
    dirs_only = np.any(pkv[..., :3], axis=-1) & np.any(pkv[..., 3:], axis=-1) & (ph["mask"] > 0)   # both directions present
    kw2 = dict(pgse_scheme=sch, csf_mask=ph["csf_mask"], ear_mask=ph["ear_mask"], verbose=0)
    for key, vols in (("colat_longit", cl), ("tensors", dt)):
        fc = model.fit(data64, dirs_only.astype(float), ph["numfasc"], **{key: [np.ascontiguousarray(v) for v in vols]}, **kw2)
        ff = model.fit(data64, dirs_only.astype(float), ph["numfasc"], **{key: [np.asfortranarray(v) for v in vols]}, **kw2)
        assert np.array_equal(fc.params_in_mask, ff.params_in_mask), key
        assert fc.params_in_mask.shape[0] >= 4096
    # (3) parallel=True over two shards (the one GPU named twice: _fit_sharded with one host thread per shard)
    model.SHARD_DEVICES = [0, 0]
    try:
        fit_p = model.fit(files["data"], files["mask"], files["numfasc"], parallel=True, **kw)
        fit_pa = model.fit(data64, ph["mask"], ph["numfasc"], parallel=True, **kwa)
    finally:
        model.SHARD_DEVICES = None
    assert np.array_equal(fit_p.params_in_mask, fit.params_in_mask)
    assert np.array_equal(fit_pa.params_in_mask, fit.params_in_mask)
    # (4) a sample of every voxel class against the CPU restatement of the reference
    roi = ph["mask"] > 0
    Kv = ph["numfasc"][roi].astype(int); cm = ph["csf_mask"][roi] > 0; em = ph["ear_mask"][roi] > 0
    cls = Kv * 4 + cm * 2 + em
    pick = np.concatenate([np.flatnonzero(cls == q)[:6] for q in range(12)])
    assert len(np.unique(cls[pick])) == 12
    ms = model.ms_interpolator
    T = {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat,
         "scheme_DeldelTE": ms["scheme_DeldelTE"]}
    b = (orc.GAMMA_H * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3e-9)
    sig_ear = np.stack([np.exp(-sch[:, 6] / 0.08) * np.exp(-b * x) for x in model.dic["DIFF_ear"]], axis=1)
    Ys = data64[roi][pick]
    ref = orc.fit_batch(T, sch, Ys, Kv[pick], cm[pick], em[pick], ph["peaks"][roi][pick], 2, True, True, sig_csf, sig_ear,
                        sig_ear.shape[1], nthreads=8)
    got = fit.params_in_mask[pick].copy()
    for col_nu, col_id in ((1, 3), (2, 4), (6, 7)):      # index of a compartment with zero weight: see test_fit_gpu
        off = ref[:, col_nu] <= 1e-9
        got[off, col_id] = 0; ref[off, col_id] = 0
    assert np.array_equal(got[:, [3, 4, 7]], ref[:, [3, 4, 7]])
    assert np.allclose(got, ref, rtol=1e-7, atol=1e-9)
    # (5) maps written as NIfTI and read back: same values, zero outside the mask, the affine of the input
    written = fit.write_nifti(str(tmp_path / "out.nii.gz"))
    assert [os.path.basename(w) for w in written] == ["out_%s.nii.gz" % n for n in fit.param_names]
    for n, w in zip(fit.param_names, written):
        m, a2 = nifti.load(w)
        assert np.array_equal(m, getattr(fit, n)) and np.allclose(a2, aff)
        assert not np.any(m[~roi])
    assert np.array_equal(fit.M0[roi], fit.params_in_mask[:, 0])


@pytest.mark.gpu
def test_volume_scalar_types_and_scaling():
    """mfx_fit_batch_volume on integer volumes with the header's scl_slope / scl_inter against mfx_fit_batch on
    get_fdata(): the same two roundings, so the same rows; ROI order arbitrary (not sorted, with repeats)."""
    from microstructure_fingerprinting_amd import engine
    model, sch, rng = _model(48)
    plan = model.ms_interpolator.plan_for(sch)
    M = sch.shape[0]
    grid = (7, 5, 9)
    nvox = int(np.prod(grid))
    clean = 400 * model.dic["dictionary"][:, rng.integers(0, 48, nvox)].T + rng.normal(0, 10, (nvox, M))
    for dt, slope, inter in ((np.int16, 0.0173, -3.5), (np.uint16, 0.0, 0.0), (np.uint8, 2.25, 0.0), (np.int32, 1.0, 0.0),
                             (np.float32, 1.0, 0.125), (np.float64, 0.0, 0.0), (np.int8, 3.1, 7.0), (np.uint32, 0.5, 0.0)):
        info = np.iinfo(dt) if np.issubdtype(dt, np.integer) else None
        sc = slope if slope != 0 else 1.0
        q = (clean - inter) / sc
        if info is not None:
            q = np.clip(np.rint(q), max(info.min, -30000), min(info.max, 30000))
        raw = np.asfortranarray(q.T.reshape((M,) + grid[::-1]).T.astype(dt))      # (grid x M) in file order
        assert raw.flags.f_contiguous and raw.shape == grid + (M,)
        vol = engine.FileOrderVolume(raw, slope, inter)
        vox = rng.integers(0, nvox, 50).astype(np.int64)
        Kv = np.full(50, 1)
        pk = np.tile(Z, (50, 1))
        got = engine.fit_batch_volume(plan, vol, vox, Kv, None, None, pk, 1, False, False)
        Yf = vol.get_fdata().reshape(-1, M, order="F")[vox]
        ref = engine.fit_batch(plan, Yf, Kv, None, None, pk, 1, False, False)
        assert np.array_equal(got, ref), dt
        assert np.array_equal(engine.volume_rows(vol, vox), Yf), dt      # mfx_volume_rows: the gather alone, rows back on the host
    with pytest.raises(ValueError):
        engine.fit_batch_volume(plan, vol, np.array([nvox]), np.array([1]), None, None, Z[None], 1, False, False)


def test_file_order_volume_host_logic(tmp_path):
    """FileOrderVolume / nifti.load_raw without a GPU: layouts accepted, index mapping, get_fdata == nifti.load."""
    from microstructure_fingerprinting_amd import nifti
    from microstructure_fingerprinting_amd.engine import FileOrderVolume
    rng = np.random.default_rng(5)
    a = rng.normal(0, 100, (4, 5, 6, 7)).astype(np.float32)
    p = str(tmp_path / "a.nii")
    nifti.save(a, np.eye(4), p)
    raw, slope, inter, aff = nifti.load_raw(p)
    assert raw.dtype == np.float32 and raw.flags.f_contiguous and np.array_equal(raw, a)
    assert FileOrderVolume.accepts(raw) and not FileOrderVolume.accepts(a) and not FileOrderVolume.accepts(a.astype(">f4"))
    assert not FileOrderVolume.accepts(np.asfortranarray(a).astype(np.float16))
    full, _ = nifti.load(p)
    vol = FileOrderVolume(raw, slope, inter)
    assert full.dtype == np.float64 and np.array_equal(vol.get_fdata(), full)
    pgz = str(tmp_path / "a.nii.gz")
    nifti.save(a, np.eye(4), pgz)
    assert np.array_equal(nifti.load_raw(pgz)[0], a)
    mask = rng.random((4, 5, 6)) < 0.4
    cflat = np.flatnonzero(mask.reshape(-1))
    fi = vol.file_order_index(cflat)
    planes = np.asarray(raw).reshape(-1, 7, order="F")            # [voxel in file order, measurement]
    assert np.array_equal(planes[fi], a[mask])
    assert np.array_equal(FileOrderVolume(raw, 2.0, 1.0).get_fdata(), a.astype(np.float64) * 2.0 + 1.0)
