"""Host-side (NumPy) logic of the product: knot tables, argument checks, protocol helpers, shard
maths.  No GPU needed: tables are only uploaded lazily."""
import os

import numpy as np
import pytest

from microstructure_fingerprinting_amd import dist as mdist
from microstructure_fingerprinting_amd import mf_utils as mfu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("pre", ["syn", "uk"])
def test_tables_equal_reference(pre):
    d = np.load(os.path.join(G, "rotation_cases.npz"))
    ms = mfu.init_PGSE_multishell_interp(d[pre + "_dic"], d[pre + "_sch_ms"], d[pre + "_ordir"])
    assert set(ms.keys()) == {"scheme_DeldelTE", "num_subs", "Gms_un", "interpolators"}   # ref:2081-2085
    assert np.array_equal(ms["Gms_un"], d[pre + "_Gms_un"])
    for s, f in enumerate(ms["interpolators"]):
        assert np.array_equal(f.x, d["%s_x_%d" % (pre, s)])
        assert np.array_equal(f.y, d["%s_y_%d" % (pre, s)])
    hdr, flat = ms.pack()
    ms2 = mfu.MultiShellInterpolator.unpack(hdr, flat)
    assert np.array_equal(ms2.x_flat, ms.x_flat) and np.array_equal(ms2.Y_flat, ms.Y_flat)
    assert np.array_equal(ms2["scheme_DeldelTE"], ms["scheme_DeldelTE"])


def test_init_and_interp_argument_errors():
    d = np.load(os.path.join(G, "rotation_cases.npz"))
    with pytest.raises(ValueError):
        mfu.init_PGSE_multishell_interp(d["syn_dic"], d["syn_sch_ms"], np.array([0, 0, 1.2]))
    sch = d["syn_sch_ms"].copy()
    sch[5, 4] *= 2
    with pytest.raises(ValueError):
        mfu.init_PGSE_multishell_interp(d["syn_dic"], sch, d["syn_ordir"])
    dic = d["syn_dic"].copy()
    b0 = np.where(d["syn_sch_ms"][:, 3] == 0)[0]
    dic[b0[1], 0] += 1.0
    with pytest.raises(ValueError):
        mfu.init_PGSE_multishell_interp(dic, d["syn_sch_ms"], d["syn_ordir"])
    ms = mfu.init_PGSE_multishell_interp(d["syn_dic"], d["syn_sch_ms"], d["syn_ordir"])
    with pytest.raises(ValueError):     # checks run before anything touches the device
        mfu.interp_PGSE_from_multishell(d["syn_schA"], np.array([0, 0, 1.1]), msinterp=ms)
    with pytest.raises(ValueError):
        mfu.interp_PGSE_from_multishell(d["syn_schA"], np.array([0, 0, 1.0, 0]), msinterp=ms)
    with pytest.raises(ValueError):
        mfu.interp_PGSE_from_multishell(d["syn_schA"], np.array([0, 0, 1.0]))


def test_solver_argument_assertions():
    A = np.ones((4, 3))
    for bad in (lambda: mfu.solve_exhaustive_posweights([[1.0]], np.ones(1), np.array([1])),
                lambda: mfu.solve_exhaustive_posweights(np.zeros((4, 3)), np.ones(4), np.array([3])),
                lambda: mfu.solve_exhaustive_posweights(A, np.ones(5), np.array([3])),
                lambda: mfu.solve_exhaustive_posweights(A, np.ones(4), np.array([2])),
                lambda: mfu.solve_exhaustive_posweights(A, np.ones(4), np.array([3, 0]))):
        with pytest.raises(AssertionError):
            bad()


def test_scheme_helpers(tmp_path):
    d = np.load(os.path.join(G, "rotation_cases.npz"))
    sch = d["uk_sch_ms"]
    assert mfu.import_PGSE_scheme(sch) is sch
    p = tmp_path / "a.scheme"
    with open(p, "w") as f:
        f.write("VERSION: STEJSKALTANNER\n")
        np.savetxt(f, sch)
    assert np.allclose(mfu.import_PGSE_scheme(str(p)), sch)
    bad = sch.copy(); bad[3, :3] *= 1.01
    with pytest.raises(ValueError):
        mfu.import_PGSE_scheme(bad)
    with pytest.raises(RuntimeError):
        mfu.import_PGSE_scheme(sch[:, :6])
    # bval/bvec -> scheme: G snapped onto the dense scheme's shells
    gam = mfu.get_gyromagnetic_ratio('H')
    b = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3) / 1e6
    out = mfu.get_PGSE_scheme_from_bval_bvec_dense(sch, b * (1 + 1e-4), sch[:, :3].T.copy(), Gtol=1e-3)
    assert out.shape == sch.shape and np.array_equal(out[:, 3], sch[:, 3]) and np.allclose(out[:, 4:], sch[:, 4:])
    with pytest.raises(ValueError):
        mfu.get_PGSE_scheme_from_bval_bvec_dense(sch, b * 1.5, sch[:, :3], Gtol=1e-4)
    assert np.isclose(mfu.get_gyromagnetic_ratio('H'), 2 * np.pi * 42.577480e6)
    with pytest.raises(ValueError):
        mfu.get_gyromagnetic_ratio('X')


def test_shard_range_partitions():
    for V in (0, 1, 7, 100000, 100003):
        for W in (1, 2, 3, 8):
            bounds = [mdist.shard_range(V, r, W) for r in range(W)]
            assert bounds[0][0] == 0 and bounds[-1][1] == V
            assert all(bounds[i][1] == bounds[i + 1][0] for i in range(W - 1))
            sizes = [b[1] - b[0] for b in bounds]
            assert max(sizes) - min(sizes) <= 1
