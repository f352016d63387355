"""Pin the CPU oracle (oracle/) against the reference.

* known-answer tables copied as DATA from the reference's own tests
  (tests/integration/test_exhaustive_fingerprinting.py:38-89),
* golden vectors produced by running the reference itself (tests/golden/gen_golden.py).
No GPU needed.
"""
import os

import numpy as np
import pytest

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")


def test_known_answers_1var():
    # reference test_boundary_cases_1d (w_exp / obj_exp tables)
    s2 = np.sqrt(2.0)
    A = np.array([[0.], [1.], [0.]])
    Y = np.array([[1, 0, s2 / 2, 0, s2 / 2], [0, 0, -s2 / 2, 2, s2 / 2], [0, 1, 0, 0, 0]], dtype=float)
    w_exp = [0, 0, 0, 2, s2 / 2]
    obj_exp = [1, 1, 1, 0, 0.5]
    for i in range(5):
        w, sub, tot, obj, yrec = orc.solve_exhaustive_posweights(A, Y[:, i].copy(), np.array([1]))
        assert np.isclose(w[0], w_exp[i]) and np.isclose(obj, obj_exp[i])


def test_known_answers_2var():
    # reference test_boundary_cases_2d
    s2, s3 = np.sqrt(2.0), np.sqrt(3.0)
    A = np.array([[0.5, s3 * 0.5], [s3 * 0.5, 0.5]])
    Y = np.array([[-s3 / 2, 0.5, -1, -s3 / 2, 0.5001, 0.5, s3 / 2, s2 / 2, -s2 / 2.0],
                  [0.5, -s3 / 2, 0, 0.5001, -s3 / 2, s3 / 2, 0.5, s2 / 2, -s2 / 2.0]])
    w_exp = np.array([[0, 0], [0, 0], [0, 0], [8.66025404e-05, 0], [0, 8.66025404e-05], [1, 0], [0, 1],
                      [0.51763809, 0.51763809], [0, 0]]).T
    obj_exp = np.array([1, 1, 1, 1.0001000025, 1.0001000025, 0, 0, 0, 1])
    for i in range(Y.shape[1]):
        w, sub, tot, obj, yrec = orc.solve_exhaustive_posweights(A, Y[:, i].copy(), np.array([1, 1]))
        assert np.all(np.isclose(w, w_exp[:, i])) and np.isclose(obj, obj_exp[i])


def test_solver_goldens():
    d = np.load(os.path.join(G, "solver_cases.npz"))
    for nm in d["names"]:
        A, y, sizes = d[nm + "_A"], d[nm + "_y"], d[nm + "_sizes"]
        w, sub, tot, obj, yrec = orc.solve_exhaustive_posweights(A, y, sizes)
        assert np.array_equal(sub, d[nm + "_sub"]), nm
        assert np.array_equal(tot, d[nm + "_tot"]), nm
        if sizes.size <= 3:
            # same loop nests, same arithmetic order: bit-exact w and objective
            assert np.array_equal(w, d[nm + "_w"]), nm
            assert obj == float(d[nm + "_obj"]), nm
        else:
            # third-party scipy.optimize.nnls (QR based): agree to rounding
            assert np.allclose(w, d[nm + "_w"], rtol=1e-9, atol=1e-12), nm
            assert np.isclose(obj, float(d[nm + "_obj"]), rtol=1e-9, atol=1e-12), nm
        assert np.allclose(yrec, d[nm + "_yrec"], rtol=1e-12, atol=1e-13), nm
    # boundary tables as executed by the reference itself
    for i in range(d["b1_Y"].shape[1]):
        w, _, _, obj, _ = orc.solve_exhaustive_posweights(d["b1_A"], d["b1_Y"][:, i].copy(), np.array([1]))
        assert w[0] == d["b1_w"][i] and obj == d["b1_obj"][i]
    for i in range(d["b2_Y"].shape[1]):
        w, _, _, obj, _ = orc.solve_exhaustive_posweights(d["b2_A"], d["b2_Y"][:, i].copy(), np.array([1, 1]))
        assert np.array_equal(w, d["b2_w"][:, i]) and obj == d["b2_obj"][i]


@pytest.mark.parametrize("pre", ["syn", "uk"])
def test_tables_match_reference(pre):
    d = np.load(os.path.join(G, "rotation_cases.npz"))
    T = orc.init_tables(d[pre + "_dic"], d[pre + "_sch_ms"], d[pre + "_ordir"])
    assert np.array_equal(T["G_un"], d[pre + "_Gms_un"])
    for s in range(T["S"]):
        assert np.array_equal(T["xs"][s], d["%s_x_%d" % (pre, s)])
        assert np.array_equal(T["Ys"][s], d["%s_y_%d" % (pre, s)])


def test_interp_goldens_synthetic():
    d = np.load(os.path.join(G, "rotation_cases.npz"))
    T = orc.init_tables(d["syn_dic"], d["syn_sch_ms"], d["syn_ordir"])
    for i, dr in enumerate(d["syn_dirs"]):
        a = orc.interp(d["syn_schA"], dr, T)
        b = orc.interp(d["syn_schB"], dr, T)
        # |g.d| may differ from BLAS by 1 ulp -> allow 1e-13 relative
        assert np.allclose(a, d["syn_outA"][i], rtol=1e-12, atol=1e-14)
        assert np.allclose(b, d["syn_outB"][i], rtol=1e-12, atol=1e-14)


def test_interp_goldens_ukbb():
    d = np.load(os.path.join(G, "rotation_cases.npz"))
    T = orc.init_tables(d["uk_dic"], d["uk_sch_ms"], d["uk_ordir"])
    for i, dr in enumerate(d["uk_dirs"]):
        a = orc.interp(d["uk_sch_subj"], dr, T)   # 9 subject G values vs 4 table shells: G-bracketing
        b = orc.interp(d["uk_sch_ms"], dr, T)     # exact-G
        assert np.allclose(a, d["uk_out_subj"][i], rtol=1e-12, atol=1e-14)
        assert np.allclose(b, d["uk_out_dense"][i], rtol=1e-12, atol=1e-14)
    # the reference's own accuracy gate (test_PGSE_from_multishell.py:262-267): < 1e-2 vs Monte-Carlo truth
    a0 = orc.interp(d["uk_sch_subj"], d["uk_dirs"][0], T)
    assert np.max(np.abs(a0 - d["uk_mc_truth"])) < 1e-2


def test_interp_errors():
    d = np.load(os.path.join(G, "rotation_cases.npz"))
    T = orc.init_tables(d["syn_dic"], d["syn_sch_ms"], d["syn_ordir"])
    sch = d["syn_schA"].copy()
    sch[3, 3] = T["G_un"][-1] * 1.5   # above the largest table G: no extrapolation (mfu:1829-1836)
    with pytest.raises(ValueError):
        orc.interp(sch, d["syn_dirs"][0], T)
    with pytest.raises(ValueError):
        orc.interp(d["syn_schA"], np.array([0, 0, 1.1]), T)   # non-unit direction (mfu:1798-1802)
    sch = d["syn_schA"].copy()
    sch[:, 4] *= 1.1
    with pytest.raises(ValueError):
        orc.interp(sch, d["syn_dirs"][0], T)                  # Delta mismatch (mfu:1786-1789)


def test_rotate_atom_goldens():
    d = np.load(os.path.join(G, "rotation_cases.npz"))
    for i, dr in enumerate(d["hcp_dirs"]):
        r = orc.rotate_atom(d["hcp_sig"], d["hcp_sch"], d["hcp_refdir"], dr, float(d["hcp_DIFF"]), d["hcp_S0"])
        assert np.allclose(r, d["hcp_rot"][i], rtol=1e-11, atol=1e-13)
    r1 = orc.rotate_atom(d["hcp_sig"][:, 3].copy(), d["hcp_sch"], d["hcp_refdir"], d["hcp_dirs"][1],
                         float(d["hcp_DIFF"]), d["hcp_S0"][:, 3].copy())
    assert r1.shape == d["hcp_rot_1d"].shape
    assert np.allclose(r1, d["hcp_rot_1d"], rtol=1e-11, atol=1e-13)


def _params_to_maps(params, maxfasc, csf_on, ear_on, props, DIFF_ear):
    """MFModelFit.__init__ semantics (mf.py:1068-1157) on flat ROI rows -> dict name -> array."""
    out = {"M0": params[:, 0]}
    for k in range(maxfasc):
        out["frac_f%d" % k] = params[:, 1 + k]
    for name, tab in props.items():
        tot = np.zeros(params.shape[0])
        for k in range(maxfasc):
            nu = params[:, 1 + k]
            ID = params[:, 1 + maxfasc + k].astype(int)
            pk = tab[ID] * (nu > 0)
            tot += nu * pk
            out["%s_f%d" % (name, k)] = pk
        out[name + "_tot"] = tot
    if csf_on:
        out["frac_csf"] = params[:, 2 * maxfasc + 1]
    if ear_on:
        nu_e = params[:, 2 * maxfasc + csf_on + 1]
        out["frac_ear"] = nu_e
        out["D_ear"] = DIFF_ear[params[:, 2 * maxfasc + csf_on + 2].astype(int)] * (nu_e > 0)
    out["MSE"] = params[:, -2]
    out["R2"] = params[:, -1]
    return out


def _csf_ear_sigs(sch, d):
    b = (orc.GAMMA_H * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / float(d["T2_csf"])) * np.exp(-b * float(d["DIFF_csf"]))
    sig_ear = np.stack([np.exp(-sch[:, 6] / float(d["T2_ear"])) * np.exp(-b * x) for x in d["DIFF_ear"]], axis=1)
    return sig_csf, sig_ear


def test_fit_goldens_mixed():
    d = np.load(os.path.join(G, "fit_cases.npz"))
    T = orc.init_tables(d["dictionary"], d["sch_ms"], np.array([0, 0, 1.0]))
    sch = d["sch"]
    sig_csf, sig_ear = _csf_ear_sigs(sch, d)
    P = orc.fit_batch(T, sch, d["Y"], d["numfasc"], d["csf"], d["ear"], d["peaks"], 2, True, True, sig_csf, sig_ear,
                      int(d["E"]), nthreads=2)
    maps = _params_to_maps(P, 2, 1, 1, {"rad": d["rad"], "fin": d["fin"]}, d["DIFF_ear"])
    for name, arr in maps.items():
        ref = d["map_" + name].reshape(-1)
        # indices / IDs exact -> property maps exact up to the weights' rounding
        assert np.allclose(arr, ref, rtol=1e-8, atol=1e-12), name
    # the empty voxel (K=0, no CSF, no EAR) returns all zeros (mf.py:387-388)
    assert np.all(P[21] == 0)


def test_fit_goldens_k1_and_c2():
    d0 = np.load(os.path.join(G, "fit_cases.npz"))
    T = orc.init_tables(d0["dictionary"], d0["sch_ms"], np.array([0, 0, 1.0]))
    d = np.load(os.path.join(G, "fit_cases_k1.npz"))
    V = d["Y"].shape[0]
    P = orc.fit_batch(T, d0["sch"], d["Y"], np.ones(V, int), np.zeros(V, bool), np.zeros(V, bool), d["peaks"], 1,
                      False, False, None, None, 0)
    maps = _params_to_maps(P, 1, 0, 0, {"rad": d0["rad"], "fin": d0["fin"]}, None)
    for name, arr in maps.items():
        assert np.allclose(arr, d["map_" + name].reshape(-1), rtol=1e-9, atol=1e-12), name

    c = np.load(os.path.join(G, "fit_c2_small.npz"))
    T2 = orc.init_tables(c["dictionary"], c["sch_ms"], np.array([0, 0, 1.0]))
    V = c["Y"].shape[0]
    P = orc.fit_batch(T2, c["sch_ms"], c["Y"], np.full(V, 2), np.zeros(V, bool), np.zeros(V, bool), c["peaks"], 2,
                      False, False, None, None, 0, nthreads=2)
    maps = _params_to_maps(P, 2, 0, 0, {"rad": c["rad"], "fin": c["fin"]}, None)
    for name, arr in maps.items():
        assert np.allclose(arr, c["map_" + name].reshape(-1), rtol=1e-9, atol=1e-12), name


def test_nnls_restatement_vs_scipy_goldens():
    """The oracle's Lawson-Hanson routine against SciPy's own outputs (tests/golden/nnls_cases.npz), incl. problems
    with duplicate / dependent columns, zero residual and nothing to fit, where a plain active-set loop cycles (the
    published routine rejects such candidates).  Exact ties (two identical columns: kinds 1, 6) may put the weight on
    either twin - SciPy's own choice depends on its BLAS - so there the twins' weights are compared as a sum."""
    d = np.load(os.path.join(G, "nnls_cases.npz"))
    for k in range(int(d["count"])):
        A, b, xs, rs, kind = d["A_%d" % k], d["b_%d" % k], d["x_%d" % k], float(d["rn_%d" % k]), int(d["kind_%d" % k])
        x, rn = orc.nnls(A, b)
        assert abs(rn - rs) <= 1e-9 * max(1.0, rs), (k, kind, rn, rs)
        if kind in (1, 6):
            x, xs = x.copy(), xs.copy()
            x[0] += x[1]; x[1] = 0.0; xs[0] += xs[1]; xs[1] = 0.0
        assert np.allclose(x, xs, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(xs).max())), (k, kind, x, xs)
