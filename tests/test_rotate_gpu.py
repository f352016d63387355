"""GPU parity: rotation entry points (mfx_rotate / mfx_rotate_cols through the C ABI) against the
reference's golden outputs and the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-11   # float64 path; |g.d| may differ from BLAS by one ulp


@pytest.fixture(scope="module")
def rot():
    return np.load(os.path.join(G, "rotation_cases.npz"))


def test_interp_synthetic_exact_and_bracketed(rot):
    from microstructure_fingerprinting_amd import mf_utils as mfu
    ms = mfu.init_PGSE_multishell_interp(rot["syn_dic"], rot["syn_sch_ms"], rot["syn_ordir"])
    for i, dr in enumerate(rot["syn_dirs"]):
        a = mfu.interp_PGSE_from_multishell(rot["syn_schA"], dr, msinterp=ms)
        b = mfu.interp_PGSE_from_multishell(rot["syn_schB"], dr, msinterp=ms)
        assert a.shape == rot["syn_outA"][i].shape
        assert np.allclose(a, rot["syn_outA"][i], rtol=RTOL, atol=1e-14)
        assert np.allclose(b, rot["syn_outB"][i], rtol=RTOL, atol=1e-14)
    # uninitialised ("slow") mode agrees with the initialised one (reference test_interp_initialized_*: <= 1e-7)
    slow = mfu.interp_PGSE_from_multishell(rot["syn_schA"], rot["syn_dirs"][2], rot["syn_dic"], rot["syn_sch_ms"],
                                           rot["syn_ordir"])
    assert np.max(np.abs(slow - rot["syn_outA"][2])) <= 1e-7


def test_interp_ukbb_fixture(rot):
    from microstructure_fingerprinting_amd import mf_utils as mfu
    ms = mfu.init_PGSE_multishell_interp(rot["uk_dic"], rot["uk_sch_ms"], rot["uk_ordir"])
    for i, dr in enumerate(rot["uk_dirs"]):
        a = mfu.interp_PGSE_from_multishell(rot["uk_sch_subj"], dr, msinterp=ms)   # G-bracketing rows
        b = mfu.interp_PGSE_from_multishell(rot["uk_sch_ms"], dr, msinterp=ms)
        assert np.allclose(a, rot["uk_out_subj"][i], rtol=RTOL, atol=1e-14)
        assert np.allclose(b, rot["uk_out_dense"][i], rtol=RTOL, atol=1e-14)
    # reference gate test_interp_from_dense_vs_monte_carlo: max abs err < 1e-2 vs Monte-Carlo truth
    a0 = mfu.interp_PGSE_from_multishell(rot["uk_sch_subj"], rot["uk_dirs"][0], msinterp=ms)
    assert np.max(np.abs(a0 - rot["uk_mc_truth"])) < 1e-2


def test_interp_g_out_of_range_raises(rot):
    from microstructure_fingerprinting_amd import mf_utils as mfu
    ms = mfu.init_PGSE_multishell_interp(rot["syn_dic"], rot["syn_sch_ms"], rot["syn_ordir"])
    sch = rot["syn_schA"].copy()
    sch[3, 3] = ms["Gms_un"][-1] * 1.5
    with pytest.raises(ValueError):
        mfu.interp_PGSE_from_multishell(sch, rot["syn_dirs"][0], msinterp=ms)


def test_rotate_atom_hcp_fixture(rot):
    from microstructure_fingerprinting_amd import mf_utils as mfu
    for i, dr in enumerate(rot["hcp_dirs"]):
        r = mfu.rotate_atom(rot["hcp_sig"], rot["hcp_sch"], rot["hcp_refdir"], dr, float(rot["hcp_DIFF"]),
                            rot["hcp_S0"], warnings=False)
        assert r.shape == rot["hcp_rot"][i].shape
        assert np.allclose(r, rot["hcp_rot"][i], rtol=1e-10, atol=1e-13)
    r1 = mfu.rotate_atom(rot["hcp_sig"][:, 3].copy(), rot["hcp_sch"], rot["hcp_refdir"], rot["hcp_dirs"][1],
                         float(rot["hcp_DIFF"]), rot["hcp_S0"][:, 3].copy(), warnings=False)
    assert r1.shape == (rot["hcp_sch"].shape[0],)
    assert np.allclose(r1, rot["hcp_rot_1d"], rtol=1e-10, atol=1e-13)
    # batched directions in one device call give the same rows
    T = mfu.RotateAtomTables(rot["hcp_sig"], rot["hcp_sch"], rot["hcp_refdir"], float(rot["hcp_DIFF"]), rot["hcp_S0"],
                             warnings=False)
    allr = T.rotate(rot["hcp_dirs"] * 3.0)   # rotate_atom normalises the direction itself
    assert np.allclose(allr, rot["hcp_rot"], rtol=1e-10, atol=1e-13)


def test_rotate_vs_oracle_random_and_single_columns(rot):
    """Batched rotation (full and one-atom-per-direction) against the oracle on random directions."""
    import ctypes as C
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(5)
    ms = mfu.init_PGSE_multishell_interp(rot["uk_dic"], rot["uk_sch_ms"], rot["uk_ordir"])
    T = orc.init_tables(rot["uk_dic"], rot["uk_sch_ms"], rot["uk_ordir"])
    sch = rot["uk_sch_subj"]
    plan = ms.plan_for(sch)
    B = 37
    d = rng.standard_normal((B, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    out = np.zeros((B, sch.shape[0], ms.num_subs))
    L.check(L.lib().mfx_rotate(plan.handle(), L.dptr(d), B, 0, L.dptr(out)))
    ref = np.stack([orc.interp(sch, x, T) for x in d])
    assert np.array_equal(out, ref)          # same formulas, same order, no FMA: bit-exact
    cols = rng.integers(0, ms.num_subs, B).astype(np.int32)
    oc = np.zeros((B, sch.shape[0]))
    L.check(L.lib().mfx_rotate_cols(plan.handle(), L.dptr(d), L.iptr(cols), B, 0, L.dptr(oc)))
    assert np.array_equal(oc, ref[np.arange(B), :, cols])
