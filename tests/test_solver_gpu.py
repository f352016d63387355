"""GPU parity of mf_utils.solve_exhaustive_posweights (C ABI mfx_solve_exhaustive).  These read like the
reference's own tests (tests/integration/test_exhaustive_fingerprinting.py): its known-answer tables,
its seeded synthetic recovery test and its HCP-dictionary test, plus the golden problems executed by
the reference itself."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_boundary_cases_1d():
    from microstructure_fingerprinting_amd import mf_utils as mfu
    sqrt2 = np.sqrt(2.0)
    A = np.array([[0], [1], [0]])
    Y = np.array([[1, 0, sqrt2 / 2, 0, sqrt2 / 2], [0, 0, -sqrt2 / 2, 2, sqrt2 / 2], [0, 1, 0, 0, 0]])
    w_exp = [0, 0, 0, 2, sqrt2 / 2]
    obj_exp = [1, 1, 1, 0, 0.5]
    for i in range(Y.shape[1]):
        (w, ind_subdic, ind_totdic, obj, y_rec) = mfu.solve_exhaustive_posweights(A, Y[:, i].copy(), np.array([1]))
        assert w.shape == (1,) and ind_subdic.shape == (1,) and y_rec.shape == (3,)
        assert np.isclose(w[0], w_exp[i]) and np.isclose(obj, obj_exp[i])


def test_boundary_cases_2d():
    from microstructure_fingerprinting_amd import mf_utils as mfu
    sqrt2, sqrt3 = np.sqrt(2.0), np.sqrt(3.0)
    A = np.array([[0.5, sqrt3 * 0.5], [sqrt3 * 0.5, 0.5]])
    Y = np.array([[-sqrt3 / 2, 0.5, -1, -sqrt3 / 2, 0.5001, 0.5, sqrt3 / 2, sqrt2 / 2, -sqrt2 / 2.0],
                  [0.5, -sqrt3 / 2, 0, 0.5001, -sqrt3 / 2, sqrt3 / 2, 0.5, sqrt2 / 2, -sqrt2 / 2.0]])
    w_exp = np.array([[0, 0], [0, 0], [0, 0], [8.66025404e-05, 0], [0, 8.66025404e-05], [1, 0], [0, 1],
                      [0.51763809, 0.51763809], [0, 0]]).transpose()
    obj_exp = np.array([1, 1, 1, 1.0001000025, 1.0001000025, 0, 0, 0, 1])
    for i in range(Y.shape[1]):
        (w, _, _, obj, _) = mfu.solve_exhaustive_posweights(A, Y[:, i].copy(), np.array([1, 1]))
        assert np.all(np.isclose(w, w_exp[:, i])) and np.isclose(obj, obj_exp[i])


def test_reference_golden_problems():
    """33 random problems (K' = 1..5) solved by the reference itself."""
    from microstructure_fingerprinting_amd import mf_utils as mfu
    d = np.load(os.path.join(G, "solver_cases.npz"))
    for nm in d["names"]:
        sizes = d[nm + "_sizes"]
        w, sub, tot, obj, yrec = mfu.solve_exhaustive_posweights(d[nm + "_A"], d[nm + "_y"], sizes)
        assert np.array_equal(sub, d[nm + "_sub"]), nm
        assert np.array_equal(tot, d[nm + "_tot"]), nm
        if sizes.size <= 3:      # same arithmetic, same order: bit-exact
            assert np.array_equal(w, d[nm + "_w"]), nm
            assert obj == float(d[nm + "_obj"]), nm
            assert sub.dtype == np.int32
        else:                    # reference = scipy.optimize.nnls (third party): 1e-5 relative (north_star)
            assert np.allclose(w, d[nm + "_w"], rtol=1e-5, atol=1e-10), nm
            assert np.isclose(obj, float(d[nm + "_obj"]), rtol=1e-5, atol=1e-10), nm
        assert np.allclose(yrec, d[nm + "_yrec"], rtol=1e-5, atol=1e-10), nm


def test_synthetic_data():
    """Reference test_synthetic_data: randn dictionary 200 x (2*700+1), ground-truth indices recovered and
    objective below the noise norm (seed as in the reference)."""
    from microstructure_fingerprinting_amd import mf_utils as mfu
    np.random.seed(141414)
    Nfasc, iso_on, Natoms, N_mris, Nvox = 2, 1, 700, 200, 5
    A = np.random.randn(N_mris * (Nfasc * Natoms + iso_on)).reshape((N_mris, Nfasc * Natoms + iso_on), order='F')
    ID_gt = np.zeros((Nfasc + iso_on, Nvox), dtype=int)
    ID_gt[0, :] = np.random.randint(0, Natoms, (Nvox))
    ID_gt[1, :] = np.random.randint(0, Natoms, (Nvox)) + Natoms
    ID_gt[Nfasc, :] = Nfasc * Natoms
    w_gt = np.random.rand(Nfasc + iso_on, Nvox)
    Y = np.zeros((N_mris, Nvox))
    for i in range(Nvox):
        Y[:, i] = np.dot(A[:, ID_gt[:, i]], w_gt[:, i])
    noise = 0.1 * (2.0 * np.random.rand(N_mris, Nvox) - 1.0)
    Ynoisy = Y + noise
    noise_sq_nrm = np.sum(noise ** 2, axis=0)
    diclengths = np.append(np.tile(Natoms, Nfasc), 1)
    ID_est = np.zeros((Nfasc + iso_on, Nvox))
    min_obj = np.zeros(Nvox)
    for i in range(Nvox):
        (w, ID_subdic, ID_est[:, i], min_obj[i], y_rec) = mfu.solve_exhaustive_posweights(A, Ynoisy[:, i], diclengths)
    assert np.all(ID_gt == ID_est)
    assert np.all(min_obj < noise_sq_nrm)


def test_hcp_rotate_then_solve():
    """Reference test_hcp_dict on the committed HCP sub-sample: rotate_atom twice, add CSF, solve with 3
    sub-dictionaries; the generating atom is recovered for both fascicles and nu ~= nu_gt."""
    from microstructure_fingerprinting_amd import mf_utils as mfu
    d = np.load(os.path.join(G, "rotation_cases.npz"))
    rng = np.random.default_rng(141414)
    sig, S0, sch = d["hcp_sig"], d["hcp_S0"], d["hcp_sch"]
    refdir, DIFF = d["hcp_refdir"], float(d["hcp_DIFF"])
    Natoms = sig.shape[1]
    i_gt = 7
    fascdirs = rng.standard_normal((3, 2)); fascdirs /= np.sqrt(np.sum(fascdirs ** 2, axis=0, keepdims=True))
    nu_gt = rng.random(3); nu_gt /= nu_gt.sum()
    gam = mfu.get_gyromagnetic_ratio('H')
    b = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-b * 3.0e-9)
    T = mfu.RotateAtomTables(sig, sch, refdir, DIFF, S0, warnings=False)
    D = T.rotate(fascdirs.T)                                     # (2, M, N) in one device call
    dictionary = np.concatenate([D[0], D[1], sig_csf[:, None]], axis=1)
    y = 500 * (nu_gt[0] * D[0][:, i_gt] + nu_gt[1] * D[1][:, i_gt] + nu_gt[2] * sig_csf)
    single = mfu.rotate_atom(sig[:, i_gt].copy(), sch, refdir, fascdirs[:, 0].copy(), DIFF, S0[:, i_gt].copy(), warnings=False)
    assert np.allclose(single, D[0][:, i_gt], rtol=1e-12)
    (w, ind_subdic, ind_totdic, min_obj, y_rec) = mfu.solve_exhaustive_posweights(dictionary, y, np.array([Natoms, Natoms, 1]))
    nu = w / np.sum(w)
    assert ind_subdic[0] == i_gt and ind_subdic[1] == i_gt and ind_subdic[2] == 0
    assert np.all(np.isclose(nu_gt, nu))
    assert np.array_equal(ind_totdic, [i_gt, Natoms + i_gt, 2 * Natoms])


def test_solver_vs_oracle_two_dictionaries_ragged():
    """K'=2 with unequal sizes, wide-ish matrices, ties: oracle parity (bit-exact)."""
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(77)
    for (M, n1, n2) in [(30, 37, 5), (64, 1, 90), (17, 150, 149)]:
        A = np.abs(rng.standard_normal((M, n1 + n2)))
        y = A[:, 3 % n1] * 0.7 + A[:, n1 + 2] * 0.2 + 0.01 * rng.standard_normal(M)
        ref = orc.solve_exhaustive_posweights(A, y, np.array([n1, n2]))
        got = mfu.solve_exhaustive_posweights(A, y, np.array([n1, n2]))
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[0], ref[0]) and got[3] == ref[3]
    # duplicated columns -> exact ties -> first hit in scan order
    A = np.abs(rng.standard_normal((12, 6)))
    A[:, 4] = A[:, 3]
    y = 2.0 * A[:, 0] + 1.0 * A[:, 3]
    ref = orc.solve_exhaustive_posweights(A, y, np.array([3, 3]))
    got = mfu.solve_exhaustive_posweights(A, y, np.array([3, 3]))
    assert np.array_equal(got[1], ref[1])
