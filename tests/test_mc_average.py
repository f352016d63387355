"""Monte-Carlo signal synthesis (ref mf_utils.py:2758-3015): oracle and HIP path against outputs of the
reference (tests/golden/mc_cases.npz).  Per-term arithmetic is identical; only the cosine's last ulp
(NumPy vs libm vs device) and the order of the spin sum differ, so the bar is 1e-13 absolute on signals
that live in [-1, 1]."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-13


@pytest.fixture(scope="module")
def d():
    return np.load(os.path.join(G, "mc_cases.npz"))


def test_oracle_mc_average_vs_reference(d):
    from oracle import oracle as orc
    n = int(d["n_spin"])
    for nt in (1, 3):
        assert np.allclose(orc.monte_carlo_average(d["phases"], d["dm"], d["gs"], 1.0, n, nthreads=nt), d["sig_direct"],
                           rtol=0, atol=TOL)
    assert np.allclose(orc.monte_carlo_average(d["phases"], d["dm"], d["gs"], float(np.sqrt(2.0e-9 / 3.0e-9)), n),
                       d["sig_direct_D"], rtol=0, atol=TOL)
    assert np.allclose(orc.monte_carlo_average(d["phases"][:, :2], d["dm"], d["gs"][:, :2], 1.0, n), d["sig_dim2"],
                       rtol=0, atol=TOL)
    assert d["sig_direct"][0] == 1.0                     # zero gradient scaling -> cos(0) everywhere
    with pytest.raises(IndexError):
        orc.monte_carlo_average(d["phases"], np.array([3]), d["gs"][:1], 1.0, n)


def _write_phase_files(tmp_path, ph, ext, dtype):
    for i, nm in enumerate("xyz"):
        ph[:, i].astype(dtype).tofile(str(tmp_path / ("sim_phase_%s.%s" % (nm, ext))))
    return str(tmp_path / ("sim_phase_x.%s" % ext))


def test_get_pgse_from_phases_argument_errors(d, tmp_path):
    """Checks that run before any device work (ref:2841-2943)."""
    from microstructure_fingerprinting_amd import mf_utils as mfu
    f = _write_phase_files(tmp_path, d["phases"], "bdouble", ">f8")
    with pytest.raises(NameError):
        mfu.get_PGSE_from_phases(f, d["sch_sim"], d["sch_new"], D=2e-9)
    with pytest.raises(ValueError):
        mfu.get_PGSE_from_phases(f, d["sch_sim"], d["sch_new"], dim=4)
    bad = d["sch_new"].copy(); bad[2, 4] = 21e-3
    with pytest.raises(ValueError, match="not used to simulate"):
        mfu.get_PGSE_from_phases(f, d["sch_sim"], bad)
    with pytest.raises(RuntimeError):
        mfu.get_PGSE_from_phases(str(tmp_path / "nope_phase_x.bdouble"), d["sch_sim"], d["sch_new"])
    for name in ("p_phase_x", "p_phase_x.xdouble", "p_phase_x.bint"):
        (tmp_path / name).write_bytes(b"\0" * 48)
        with pytest.raises(ValueError):
            mfu.get_PGSE_from_phases(str(tmp_path / name), d["sch_sim"], d["sch_new"])
    (tmp_path / "q_phase_x.bdouble").write_bytes(b"\0" * 40)      # 5 items, 3 acquisitions
    with pytest.raises(RuntimeError, match="corrupted"):
        mfu.get_PGSE_from_phases(str(tmp_path / "q_phase_x.bdouble"), d["sch_sim"], d["sch_new"])


@pytest.mark.gpu
def test_mc_average_gpu_vs_reference(d):
    from microstructure_fingerprinting_amd import mf_utils as mfu
    n = int(d["n_spin"])
    assert np.allclose(mfu.monte_carlo_average(d["phases"], d["dm"], d["gs"], 1.0, n), d["sig_direct"], rtol=0, atol=TOL)
    assert np.allclose(mfu.monte_carlo_average(d["phases"], d["dm"], d["gs"], float(np.sqrt(2.0e-9 / 3.0e-9)), n),
                       d["sig_direct_D"], rtol=0, atol=TOL)
    assert np.allclose(mfu.monte_carlo_average(d["phases"][:, :2], d["dm"], d["gs"][:, :2], 1.0, n), d["sig_dim2"],
                       rtol=0, atol=TOL)
    assert mfu.monte_carlo_average(d["phases"], d["dm"][:0], d["gs"][:0], 1.0, n).shape == (0,)
    with pytest.raises(ValueError):
        mfu.monte_carlo_average(d["phases"], np.array([3]), d["gs"][:1], 1.0, n)      # outside the phase table
    with pytest.raises(ValueError):
        mfu.monte_carlo_average(d["phases"], d["dm"], d["gs"][:, :2], 1.0, n)         # gscaling / phases dim mismatch


@pytest.mark.gpu
def test_get_pgse_from_phases_gpu_vs_reference(d, tmp_path):
    from microstructure_fingerprinting_amd import mf_utils as mfu
    fb = _write_phase_files(tmp_path, d["phases"], "bdouble", ">f8")
    fl = _write_phase_files(tmp_path, d["phases"], "lfloat", "<f4")
    assert np.allclose(mfu.get_PGSE_from_phases(fb, d["sch_sim"], d["sch_new"]), d["sig_files_bdouble"], rtol=0, atol=TOL)
    assert np.allclose(mfu.get_PGSE_from_phases(fl, d["sch_sim"], d["sch_new"], D_sim=3.0e-9, D=2.0e-9),
                       d["sig_files_lfloat_D"], rtol=0, atol=TOL)
    assert np.allclose(mfu.get_PGSE_from_phases(fb, d["sch_sim"], d["sch_new_xy"], dim=2), d["sig_files_dim2"],
                       rtol=0, atol=TOL)


@pytest.mark.gpu
def test_mc_average_gpu_large_vs_oracle_and_properties():
    """Ragged sizes (spins not a multiple of the chunk, 37 sequences over 5 acquisitions -> partial tiles)
    against the oracle, plus size-independent properties: S(0)=1, S(g)=S(-g), |S|<=1, and for Gaussian
    phases S -> exp(-sigma^2 g^2 / 2) within Monte-Carlo error."""
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(5)
    n_ref, n_spin, n_seq = 5, 50021, 37
    ph = rng.standard_normal((n_ref * n_spin, 3)) * 2.0
    dm = rng.integers(0, n_ref, n_seq)
    gs = rng.uniform(-1.5, 1.5, (n_seq, 3))
    gs[5] = 0
    got = mfu.monte_carlo_average(ph, dm, gs, 0.9, n_spin)
    ref = orc.monte_carlo_average(ph, dm, gs, 0.9, n_spin, nthreads=8)
    assert np.allclose(got, ref, rtol=0, atol=TOL)
    assert got[5] == 1.0 and np.all(np.abs(got) <= 1.0)
    assert np.allclose(mfu.monte_carlo_average(ph, dm, -gs, 0.9, n_spin), got, rtol=0, atol=1e-15)
    expect = np.exp(-0.5 * (2.0 * 0.9) ** 2 * np.sum(gs ** 2, axis=1))
    assert np.max(np.abs(got - expect)) < 5.0 / np.sqrt(n_spin)
