"""cleanup_2fascicles (ref mf.py:36-335) and the diffusion-tensor helpers (ref mf_utils.py:865-1135) against
outputs of the reference itself (tests/golden/cleanup_cases.npz, made by gen_golden.py --only cleanup)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def d():
    return np.load(os.path.join(G, "cleanup_cases.npz"))


def test_oracle_cleanup_selection_vs_reference_goldens(d):
    """The oracle's restatement of the voxel loop (the referee of the device kernel) against the reference's outputs."""
    from oracle import oracle as orc
    roi = d["mask"] > 0
    pk, nf = orc.cleanup_select(d["f1"][roi], d["f2"][roi], d["d1"][roi], d["d2"][roi])
    assert np.array_equal(nf, d["peaks_nf"][roi]) and np.array_equal(pk, d["peaks_pk"][roi])


@pytest.mark.gpu
def test_cleanup_device_kernel_vs_oracle_large():
    """2e5 random voxels (all outcomes, exact ties, zero weights, anti-parallel and identical peaks) through the device
    kernel: bit-identical to the oracle."""
    from microstructure_fingerprinting_amd import engine, mf as mfm, synth
    from oracle import oracle as orc
    rng = np.random.default_rng(8)
    n = 200000
    p1, p2 = synth.unit_vectors(rng, n), synth.unit_vectors(rng, n)
    near = rng.random(n) < 0.3                     # crossings around the 15 degree merge angle, both signs
    ang = np.deg2rad(rng.uniform(0, 30, n))
    ortho = np.cross(p1, synth.unit_vectors(rng, n)); ortho /= np.linalg.norm(ortho, axis=1)[:, None]
    p2[near] = (np.cos(ang)[:, None] * p1 + np.sin(ang)[:, None] * ortho)[near] * rng.choice([-1.0, 1.0], n)[near, None]
    p2[::1000] = p1[::1000]; p2[1::1000] = -p1[1::1000]
    f1, f2 = rng.uniform(0, 0.6, n), rng.uniform(0, 0.6, n)
    f2[::7] = f1[::7]; f1[::11] = 0.0; f2[::13] = 0.0; f1[::17] = 0.075; f2[::19] = 0.2
    got = engine.cleanup_select(f1, f2, p1, p2, np.cos(mfm.CLEANUP_ANG_MIN * np.pi / 180), mfm.CLEANUP_RATIO, mfm.CLEANUP_W_KEEP,
                                mfm.CLEANUP_W_SMALL)
    ref = orc.cleanup_select(f1, f2, p1, p2)
    assert np.array_equal(got[1], ref[1]) and np.array_equal(got[0], ref[0])
    assert set(np.unique(got[1])) == {0.0, 1.0, 2.0}


@pytest.mark.gpu
def test_cleanup_peaks_mode_bit_exact(d):
    import microstructure_fingerprinting_amd as mf
    pk, nf = mf.cleanup_2fascicles(d["f1"], d["f2"], 'peaks', d["d1"].copy(), d["d2"].copy(), d["mask"])
    assert np.array_equal(nf, d["peaks_nf"])
    assert np.array_equal(pk, d["peaks_pk"])
    assert set(np.unique(nf)) == {0.0, 1.0, 2.0}                     # every outcome is exercised
    assert np.all(pk[d["mask"] == 0] == 0) and np.all(nf[d["mask"] == 0] == 0)


@pytest.mark.gpu
def test_cleanup_colat_and_frac12(d):
    import microstructure_fingerprinting_amd as mf
    f12 = np.stack([d["f1"], d["f2"]], -1)
    for frac12 in (f12, f12[..., None, :]):                           # (..., 2) and (..., 1, 2) layouts
        pk, nf = mf.cleanup_2fascicles(None, None, 'colat_longit', d["cl1"], d["cl2"], d["mask"], frac12=frac12)
        assert np.array_equal(nf, d["colat_nf"])
        assert np.allclose(pk, d["colat_pk"], rtol=0, atol=1e-15)


@pytest.mark.gpu
def test_cleanup_tensor_mode(d):
    import microstructure_fingerprinting_amd as mf
    pk, nf = mf.cleanup_2fascicles(d["f1"], d["f2"], 'tensor', d["T1"][..., None, :], d["T2"], d["mask"])
    assert np.array_equal(nf, d["tensor_nf"])
    assert np.allclose(pk, d["tensor_pk"], rtol=0, atol=1e-14)


def test_cleanup_argument_errors(d):
    import microstructure_fingerprinting_amd as mf
    a = (d["f1"], d["f2"], 'peaks', d["d1"], d["d2"], d["mask"])
    with pytest.raises(ValueError):
        mf.cleanup_2fascicles(None, d["f2"], 'peaks', d["d1"], d["d2"], d["mask"])         # ref:97-103
    with pytest.raises(ValueError):
        mf.cleanup_2fascicles(a[0][:-1], *a[1:])                                            # ref:127-130
    with pytest.raises(ValueError):
        mf.cleanup_2fascicles(a[0], a[1], 'odf', *a[3:])                                    # ref:150-151
    with pytest.raises(ValueError):
        mf.cleanup_2fascicles(a[0], a[1], 'colat_longit', *a[3:])                           # ref:154-158
    with pytest.raises(ValueError):
        mf.cleanup_2fascicles(None, None, 'peaks', d["d1"], d["d2"], d["mask"], frac12=d["f1"][..., None])


@pytest.mark.gpu
def test_cleanup_from_nifti_files(d, tmp_path):
    import microstructure_fingerprinting_amd as mf
    from microstructure_fingerprinting_amd import nifti
    names = {}
    for k in ("f1", "f2", "d1", "d2", "mask"):
        names[k] = str(tmp_path / (k + ".nii.gz"))
        nifti.save(d[k], np.eye(4), names[k])
    pk, nf = mf.cleanup_2fascicles(names["f1"], names["f2"], 'peaks', names["d1"], names["d2"], names["mask"])
    assert np.array_equal(nf, d["peaks_nf"]) and np.array_equal(pk, d["peaks_pk"])


def test_dt_helpers(d):
    from microstructure_fingerprinting_amd import mf_utils as mfu
    for order in ("row", "column", "diagonal"):
        A = mfu.DT_vec_to_2Darray(d["T1"], order)
        assert np.array_equal(A, d["dt2d_" + order])
        assert np.array_equal(mfu.DT_array_to_vec(A, order), d["dtvec_" + order])
        assert np.array_equal(mfu.DT_array_to_vec(A, order), d["T1"])
        assert np.allclose(mfu.DT_vec_to_peaks(d["T2"], order, d["mask"]), d["dtpk_" + order], rtol=0, atol=1e-14)
    # stick tensors: same as the reference's (whose perpendicular pair is random) to rounding
    T = mfu.peaks_to_DT_vec(np.stack([d["d1"], d["d2"]], axis=-2).copy(), 'column')
    assert len(T) == 2 and np.allclose(T[0], d["T1"], rtol=0, atol=1e-17)
    one = mfu.DT_vec_to_peaks(d["T1"][0, 0, 0], 'column')
    assert one.shape == (3,) and np.allclose(np.abs(one @ d["d1"][0, 0, 0]), 1.0)
    for bad in (lambda: mfu.DT_vec_to_2Darray(d["T1"], 'zigzag'), lambda: mfu.DT_array_to_vec(np.zeros((3, 2))),
                lambda: mfu.DT_vec_to_2Darray(np.zeros((4, 5)), 'row'),
                lambda: mfu.peaks_to_DT_vec(np.zeros((2, 3)), 'row', 1e-3, 2e-3)):
        with pytest.raises(ValueError):
            bad()
