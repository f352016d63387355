"""MFModel / MFModelFit host mirror (mf.py:464-1229 of the reference): argument handling, map scatter,
NIfTI output.  CPU part needs no GPU (the device is only touched inside fit's batched call)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
Z = np.array([0.0, 0.0, 1.0])


def _model_dict(d):
    return {"dictionary": d["dictionary"], "sch_mat": d["sch_ms"], "orientation": Z, "num_atom": int(d["N"]),
            "num_ear": int(d["E"]), "T2_csf": float(d["T2_csf"]), "DIFF_csf": float(d["DIFF_csf"]),
            "T2_ear": float(d["T2_ear"]), "DIFF_ear": d["DIFF_ear"], "fasc_propnames": ["rad ", "fin"],
            "rad": d["rad"], "fin": d["fin"]}


def test_fit_object_maps_from_params_match_reference(tmp_path):
    """MFModelFit fed with the ORACLE's params rows reproduces every map the reference's fit object holds,
    then write_nifti / load round-trips them."""
    import microstructure_fingerprinting_amd as mf
    from microstructure_fingerprinting_amd import nifti
    from oracle import oracle as orc
    d = np.load(os.path.join(G, "fit_cases.npz"))
    sch = d["sch"]
    T = orc.init_tables(d["dictionary"], d["sch_ms"], Z)
    b = (orc.GAMMA_H * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / float(d["T2_csf"])) * np.exp(-b * float(d["DIFF_csf"]))
    sig_ear = np.stack([np.exp(-sch[:, 6] / float(d["T2_ear"])) * np.exp(-b * x) for x in d["DIFF_ear"]], axis=1)
    P = orc.fit_batch(T, sch, d["Y"], d["numfasc"], d["csf"], d["ear"], d["peaks"], 2, True, True, sig_csf, sig_ear,
                      int(d["E"]))
    mask = np.ones((4, 6))
    fitinfo = {"maxfasc": 2, "csf_on": True, "ear_on": True, "affine": None, "mask": mask,
               "fasc_propnames": ["rad", "fin"], "peaks_roi": d["peaks"], "_dict_rad": d["rad"], "_dict_fin": d["fin"],
               "DIFF_ear": d["DIFF_ear"]}
    fit = mf.MFModelFit(fitinfo, P)
    assert fit.param_names == list(d["param_names"])
    for name in fit.param_names:
        assert np.allclose(getattr(fit, name), d["map_" + name], rtol=1e-8, atol=1e-12), name
    with pytest.raises(ValueError):
        fit.write_nifti(str(tmp_path / "x"))          # no affine known (ref:1198-1203)
    with pytest.raises(ValueError):
        fit.write_nifti(str(tmp_path / "x.txt"), affine=np.eye(4))
    aff = np.diag([2.0, 2.0, 2.5, 1.0]); aff[:3, 3] = [10, -5, 3]
    files = fit.write_nifti(str(tmp_path / "sub01.nii.gz"), affine=aff)
    assert len(files) == len(fit.param_names) and all(f.endswith(".nii.gz") for f in files)
    arr, a2 = nifti.load(files[fit.param_names.index("peak_f1")])
    assert arr.shape == (4, 6, 3) and np.allclose(arr, fit.peak_f1) and np.allclose(a2, aff)
    files = fit.write_nifti(str(tmp_path / "plain"), affine=aff)
    arr, _ = nifti.load(files[0])
    assert files[0].endswith("plain_M0.nii") and np.array_equal(arr, fit.M0)


def test_fit_argument_errors():
    import microstructure_fingerprinting_amd as mf
    d = np.load(os.path.join(G, "fit_cases.npz"))
    with pytest.raises(ValueError):
        mf.MFModel(42)
    model = mf.MFModel(_model_dict(d))
    M = d["sch"].shape[0]
    data = np.ones((2, 3, M)); mask = np.ones((2, 3)); pk = np.tile([0, 0, 1.0, 1, 0, 0], (2, 3, 1))
    kw = dict(peaks=pk, pgse_scheme=d["sch"], verbose=0)
    with pytest.raises(ValueError):
        model.fit(data, np.zeros((2, 3)), 1, **kw)                       # empty mask (ref:647-649)
    with pytest.raises(ValueError):
        model.fit(data, np.ones((3, 2)), 1, **kw)                        # shape mismatch (ref:651-657)
    with pytest.raises(ValueError):
        model.fit(data, mask, 3, **kw)                                   # > MAX_FASC (ref:682-687)
    with pytest.raises(ValueError):
        model.fit(data, mask, np.ones((3, 3)), **kw)                     # numfasc shape (ref:672-677)
    with pytest.raises(RuntimeError):
        model.fit(data, mask, 1, pgse_scheme=d["sch"], verbose=0)        # no directions (ref:730-731)
    with pytest.raises(ValueError):
        model.fit(data, mask, 1, peaks=np.ones((2, 3, 4)), pgse_scheme=d["sch"], verbose=0)   # not multiple of 3
    with pytest.raises(ValueError):
        model.fit(data, mask, 2, peaks=np.zeros((2, 3, 6)), pgse_scheme=d["sch"], verbose=0)  # zero peak (ref:803-815)
    with pytest.raises(TypeError):
        model.fit(data, mask, 1, peaks=pk, bvals=np.ones(M), verbose=0)  # bvecs missing (ref:832-834)
    with pytest.raises(ValueError):
        model.fit(data, mask, 1, peaks=pk, pgse_scheme=d["sch"][:, :6], verbose=0)             # 7 columns
    with pytest.raises(ValueError):
        model.fit(data, mask, 1, csf_mask=np.ones((3, 3)), **kw)        # csf_mask shape (ref:864-869)


@pytest.mark.gpu
def test_mfmodel_fit_end_to_end_vs_reference():
    """MFModel(dict).fit(ndarrays) against the maps the reference's own MFModel.fit produced."""
    import microstructure_fingerprinting_amd as mf
    d = np.load(os.path.join(G, "fit_cases.npz"))
    model = mf.MFModel(_model_dict(d))
    for par in (False, True):
        fit = model.fit(d["Y"].reshape(4, 6, -1), np.ones((4, 6)), d["numfasc"].reshape(4, 6),
                        peaks=d["peaks"].reshape(4, 6, 6), pgse_scheme=d["sch"],
                        csf_mask=d["csf"].reshape(4, 6).astype(float), ear_mask=d["ear"].reshape(4, 6).astype(float),
                        verbose=0, parallel=par)
        assert fit.param_names == list(d["param_names"])
        for name in fit.param_names:
            assert np.allclose(getattr(fit, name), d["map_" + name], rtol=1e-5, atol=1e-9), name
    # K=1 everywhere, scalar numfasc, 1-D "image" (ref: numfasc scalar path, mf.py:660-662)
    k1 = np.load(os.path.join(G, "fit_cases_k1.npz"))
    fit = model.fit(k1["Y"], np.ones(k1["Y"].shape[0]), 1, peaks=k1["peaks"], pgse_scheme=d["sch"], verbose=0)
    assert fit.param_names == list(k1["param_names"])
    for name in fit.param_names:
        assert np.allclose(getattr(fit, name), k1["map_" + name], rtol=1e-5, atol=1e-12), name
    # masked-out voxels stay zero, colatitude/longitude input gives the same peaks
    th = np.arccos(k1["peaks"][:, 2]); ph = np.arctan2(k1["peaks"][:, 1], k1["peaks"][:, 0])
    msk = np.ones(k1["Y"].shape[0]); msk[3] = 0
    fit2 = model.fit(k1["Y"], msk, 1, colat_longit=np.stack([th, ph], axis=1), pgse_scheme=d["sch"], verbose=0)
    assert fit2.M0[3] == 0 and np.allclose(np.delete(fit2.M0, 3), np.delete(fit.M0, 3), rtol=1e-9)


# ---- input normalisation either side of the path, pinned on outputs of the reference itself
def test_scheme_from_bvals_bvecs_matches_reference(tmp_path):
    """get_PGSE_scheme_from_bval_bvec_dense (ref mf_utils.py:2197-2300) on the UKBB subject fixture, and
    import_PGSE_scheme (ref:2128-2192) re-reading a scheme file written from the golden array."""
    from microstructure_fingerprinting_amd import mf_utils as mfu
    d = np.load(os.path.join(G, "input_cases.npz"))
    got = mfu.get_PGSE_scheme_from_bval_bvec_dense(d["uk_dense"], d["uk_bvals"], d["uk_bvecs"], 1e-3)
    assert got.shape == d["uk_scheme_from_bvals"].shape
    assert np.allclose(got, d["uk_scheme_from_bvals"], rtol=1e-12, atol=0)
    # bvecs given as (3, n) or (n, 3), bvals as column: same result (ref:2226-2245)
    got2 = mfu.get_PGSE_scheme_from_bval_bvec_dense(d["uk_dense"], d["uk_bvals"][:, None], d["uk_bvecs"].T, 1e-3)
    assert np.array_equal(got, got2)
    # scheme text file: header line + 7 columns, %.17g keeps doubles exact
    p = tmp_path / "x.scheme"
    with open(p, "w") as f:
        f.write("VERSION: STEJSKALTANNER\n")
        np.savetxt(f, d["hcp_scheme"], fmt="%.17g")
    assert np.array_equal(mfu.import_PGSE_scheme(str(p)), d["hcp_scheme"])
    assert np.array_equal(mfu.import_PGSE_scheme(d["hcp_scheme"]), d["hcp_scheme"])


@pytest.mark.gpu
def test_fit_from_tensors_and_colat_matches_reference():
    """MFModel.fit fed through tensors= + bvals/bvecs and through colat_longit= (ref mf.py:693-860) against
    the maps the reference produced from the same arrays."""
    import microstructure_fingerprinting_amd as mf
    d = np.load(os.path.join(G, "input_cases.npz"))
    N = d["m_dictionary"].shape[1]
    md = {"dictionary": d["m_dictionary"], "sch_mat": d["m_sch_ms"], "orientation": Z, "num_atom": N,
          "num_ear": len(d["m_DIFF_ear"]), "T2_csf": 2.0, "DIFF_csf": 3.0e-9, "T2_ear": 0.08,
          "DIFF_ear": d["m_DIFF_ear"], "fasc_propnames": ["rad ", "fin"], "rad": d["m_rad"], "fin": d["m_fin"]}
    model = mf.MFModel(md)
    V = d["m_Y"].shape[0]
    fit = model.fit(d["m_Y"], np.ones(V), 2, tensors=[d["m_T1"], d["m_T2"]], bvals=d["m_bvals"], bvecs=d["m_bvecs"],
                    verbose=0)
    assert fit.param_names == list(d["t_names"])
    for name in fit.param_names:
        ref = d["t_" + name]
        if name.startswith("peak"):        # eigenvector sign is LAPACK's choice; the fit only sees |g.dir|
            s = np.sign(np.sum(getattr(fit, name) * ref, axis=-1, keepdims=True))
            assert np.allclose(getattr(fit, name) * s, ref, rtol=1e-6, atol=1e-9), name
        else:
            assert np.allclose(getattr(fit, name), ref, rtol=1e-5, atol=1e-12), name
    fit = model.fit(d["m_Y"], np.ones(V), 1, colat_longit=d["m_colat"], pgse_scheme=d["m_sch_ms"], verbose=0)
    assert fit.param_names == list(d["c_names"])
    for name in fit.param_names:
        assert np.allclose(getattr(fit, name), d["c_" + name], rtol=1e-5, atol=1e-12), name


def test_fit_checks_protocol_timing_and_gradient_norms():
    """What the reference raises from inside every voxel with a fascicle (interp_PGSE_from_multishell,
    mf_utils.py:1786-1789 and 1804-1807) is raised once per fit: a protocol whose Delta/delta/TE differ from the
    dictionary's, and gradient directions that are neither zero nor unit vectors.  No GPU needed: both are argument
    errors found before the device is touched."""
    from microstructure_fingerprinting_amd import mf, synth
    rng = np.random.default_rng(3)
    sch = synth.make_scheme(rng, 2, [1000, 2000], [12, 12])
    N = 10
    model = {"dictionary": synth.make_dictionary(rng, sch, N), "sch_mat": sch, "orientation": np.array([0.0, 0.0, 1.0]),
             "num_atom": N, "num_ear": 2, "T2_csf": 2.0, "DIFF_csf": 3e-9, "T2_ear": 0.08, "DIFF_ear": np.array([0.3e-9, 0.9e-9]),
             "fasc_propnames": ["rad"], "rad": np.linspace(1e-6, 2e-6, N)}
    m = mf.MFModel(model)
    V = 4
    Y = rng.uniform(1, 2, (V, sch.shape[0]))
    pk = synth.unit_vectors(rng, V)
    bad = sch.copy(); bad[:, 4] *= 1.1                     # other Delta
    with pytest.raises(ValueError, match="Delta, delta and TE"):
        m.fit(Y, np.ones(V, int), 1, peaks=pk, pgse_scheme=bad, verbose=0)
    bad = sch.copy(); bad[5, :3] *= 1.5                    # not a unit vector
    with pytest.raises(ValueError, match="zero or unit norm"):
        m.fit(Y, np.ones(V, int), 1, peaks=pk, pgse_scheme=bad, verbose=0)
