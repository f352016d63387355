"""GPU parity of the voxel loop (mfx_fit_batch through the C ABI) against the reference goldens and
the CPU oracle.  Bar: atom indices bit-exact, weights / derived parameters within 1e-5 relative
(BASELINE.json north_star); the fused kernel in fact reproduces the oracle to ~1e-14."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL_W = 1e-5          # tolerance stated by north_star for fitted weights
Z = np.array([0.0, 0.0, 1.0])


def _oracle_tables(ms):
    return {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat}


def _check(got, ref, maxfasc):
    ids = slice(1 + maxfasc, 1 + 2 * maxfasc)
    assert np.array_equal(got[:, ids], ref[:, ids]), "selected atom indices differ"
    assert np.allclose(got, ref, rtol=RTOL_W, atol=1e-10)
    return float(np.max(np.abs(got - ref) / (np.abs(ref) + 1e-300)))


def test_k2_reference_golden_small():
    """6 config-2-shaped voxels (2 fascicles, M=200, N=48) fitted by the reference itself."""
    from microstructure_fingerprinting_amd import engine
    from microstructure_fingerprinting_amd import mf_utils as mfu
    c = np.load(os.path.join(G, "fit_c2_small.npz"))
    ms = mfu.init_PGSE_multishell_interp(c["dictionary"], c["sch_ms"], Z)
    V = c["Y"].shape[0]
    P = engine.fit_batch(ms.plan_for(c["sch_ms"]), c["Y"], np.full(V, 2), None, None, c["peaks"], 2, False, False)
    assert np.allclose(P[:, 0], c["map_M0"], rtol=RTOL_W)
    assert np.allclose(P[:, 1], c["map_frac_f0"], rtol=RTOL_W, atol=1e-12)
    assert np.allclose(P[:, 2], c["map_frac_f1"], rtol=RTOL_W, atol=1e-12)
    # atom ids, through the property tables exactly as MFModelFit does (mf.py:1111-1117)
    for k in range(2):
        ids = P[:, 3 + k].astype(int)
        assert np.array_equal(c["rad"][ids] * (P[:, 1 + k] > 0), c["map_rad_f%d" % k])
        assert np.array_equal(c["fin"][ids] * (P[:, 1 + k] > 0), c["map_fin_f%d" % k])
    assert np.allclose(P[:, -2], c["map_MSE"], rtol=RTOL_W)
    assert np.allclose(P[:, -1], c["map_R2"], rtol=RTOL_W)


@pytest.mark.parametrize("N,V,seed", [(48, 64, 11), (100, 32, 12), (257, 16, 13), (782, 12, 1)])
def test_k2_vs_oracle(N, V, seed):
    """Seeded synthetic voxels at several dictionary sizes (ragged tiles: 48=3x16, 100, 257, 782)."""
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    c = synth.config("C2")
    sch = synth.make_scheme(rng, c["n_b0"], c["shells_b"], c["dirs"])
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    T = _oracle_tables(ms)
    peaks, Y, _, _ = synth.make_voxels(rng, V, 2, lambda d: np.stack([orc.interp(sch, x, T | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}) for x in d]), N)
    z = np.zeros(V, bool)
    ref = orc.fit_batch(T, sch, Y, np.full(V, 2), z, z, peaks, 2, False, False, None, None, 0, nthreads=8)
    got = engine.fit_batch(ms.plan_for(sch), Y, np.full(V, 2), None, None, peaks, 2, False, False)
    err = _check(got, ref, 2)
    assert err < 1e-9


def test_k2_bracketed_protocol_and_small_M():
    """Subject protocol with G between table shells (bracketing rows) and M=60 (KSTEPS bucket 16)."""
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(21)
    sch_ms = synth.make_scheme(rng, 2, [1000, 2000, 3000], [30, 30, 30])
    dic = synth.make_dictionary(rng, sch_ms, 40)
    ms = mfu.init_PGSE_multishell_interp(dic, sch_ms, Z)
    sch = sch_ms[rng.permutation(sch_ms.shape[0])[:60]].copy()
    nz = sch[:, 3] > 0
    Gs = ms["Gms_un"]
    sch[nz, 3] = rng.uniform(Gs[1], Gs[-1], nz.sum())
    sch[np.where(nz)[0][:4], 3] = Gs[2]
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    V = 24
    peaks, Y, _, _ = synth.make_voxels(rng, V, 2, lambda d: np.stack([orc.interp(sch, x, T) for x in d]), 40)
    z = np.zeros(V, bool)
    ref = orc.fit_batch(T, sch, Y, np.full(V, 2), z, z, peaks, 2, False, False, None, None, 0)
    got = engine.fit_batch(ms.plan_for(sch), Y, np.full(V, 2), None, None, peaks, 2, False, False)
    _check(got, ref, 2)


def test_k2_degenerate_voxels():
    """Edge cases of the 2-variable case analysis (mf_utils.py:348-379) inside the fused kernel:
    pure single-fascicle signal (single-active branch, first-hit tie break over i2), all-negative
    signal (w = 0, indices (0,0), min_obj = ||y||^2), zero signal, identical directions."""
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(31)
    c = synth.config("C2")
    sch = synth.make_scheme(rng, c["n_b0"], c["shells_b"], c["dirs"])
    N = 64
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    V = 6
    peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
    Y = np.zeros((V, sch.shape[0]))
    D0 = orc.interp(sch, peaks[0, :3], T)
    Y[0] = 300.0 * D0[:, 17]                           # exactly one atom of fascicle 0
    D1 = orc.interp(sch, peaks[1, 3:], T)
    Y[1] = 200.0 * D1[:, 5]                            # exactly one atom of fascicle 1
    Y[2] = -np.abs(rng.normal(100, 10, sch.shape[0]))  # nothing fits: w = 0
    Y[3] = 0.0                                         # zero signal
    peaks[4, 3:] = peaks[4, :3]                        # both fascicles along the same direction
    D4 = orc.interp(sch, peaks[4, :3], T)
    Y[4] = 100 * D4[:, 3] + 150 * D4[:, 40] + rng.normal(0, 5, sch.shape[0])
    Y[5] = rng.normal(0, 50, sch.shape[0])             # pure noise, mixed signs
    z = np.zeros(V, bool)
    ref = orc.fit_batch(T, sch, Y, np.full(V, 2), z, z, peaks, 2, False, False, None, None, 0)
    got = engine.fit_batch(ms.plan_for(sch), Y, np.full(V, 2), None, None, peaks, 2, False, False)
    _check(got, ref, 2)
    assert got[2, 0] == 0 and got[2, 3] == 0 and got[2, 4] == 0
    assert np.all(got[3] == 0)


def test_k2_full_size_properties():
    """Size-independent properties at BASELINE's C2 shape (782 atoms x 200 measurements, 2000 voxels):
    scale equivariance (y -> 2y doubles M0, keeps ids/fractions), fascicle-swap symmetry (swapping
    the two peaks swaps ids and fractions), determinism, and parity with the oracle on a sample."""
    import torch
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    sch, dic, rng = synth.make_model("C2")
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    V, N, M = 2000, ms.num_subs, sch.shape[0]
    dev = torch.device("cuda", 0)
    peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
    atoms = rng.integers(0, N, (V, 2)).astype(np.int32)
    nu = rng.dirichlet(np.ones(2), V)
    d_pk = torch.from_numpy(peaks).to(dev)
    d_Y = torch.zeros((V, M), dtype=torch.float64, device=dev)
    for k in range(2):
        col = engine.rotate_columns_dev(plan, d_pk[:, 3 * k:3 * k + 3].contiguous(), torch.from_numpy(atoms[:, k].copy()).to(dev))
        d_Y += 500.0 * torch.from_numpy(nu[:, k:k + 1].copy()).to(dev) * col
    d_Y += torch.from_numpy(rng.normal(0, 500 / 30.0, (V, M))).to(dev)

    def run(dY, dpk):
        out = torch.zeros((V, 7), dtype=torch.float64, device=dev)
        L.check(L.lib().mfx_fit_batch_dev(plan.handle(), dY.data_ptr(), dpk.data_ptr(), 2, 0, 0, None, None, 0, V,
                                          out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.synchronize(dev)
        return out.cpu().numpy()

    P = run(d_Y, d_pk)
    assert np.array_equal(P, run(d_Y, d_pk))                                  # deterministic
    P2 = run(2.0 * d_Y, d_pk)
    assert np.array_equal(P2[:, 3:5], P[:, 3:5])
    assert np.allclose(P2[:, 0], 2 * P[:, 0], rtol=1e-12) and np.allclose(P2[:, 1:3], P[:, 1:3], rtol=1e-10, atol=1e-13)
    assert np.allclose(P2[:, 5], 4 * P[:, 5], rtol=1e-10)
    d_pk_sw = torch.cat([d_pk[:, 3:], d_pk[:, :3]], dim=1).contiguous()
    Ps = run(d_Y, d_pk_sw)
    assert np.array_equal(Ps[:, 3], P[:, 4]) and np.array_equal(Ps[:, 4], P[:, 3])
    assert np.allclose(Ps[:, 1], P[:, 2], rtol=1e-9, atol=1e-12) and np.allclose(Ps[:, 5:], P[:, 5:], rtol=1e-9)
    # fractions sum to one wherever something was fitted; R2 in [0, 1]
    fitted = P[:, 0] > 0
    assert np.allclose(P[fitted, 1] + P[fitted, 2], 1.0, rtol=1e-12)
    assert np.all((P[:, 6] >= 0) & (P[:, 6] <= 1 + 1e-12))
    # the exhaustive optimum fits at least as well as the generating pair: MSE <= noise power (sigma^2)
    sig2 = (500 / 30.0) ** 2
    assert 0.7 * sig2 < np.mean(P[:, 5]) < 1.05 * sig2
    # oracle parity on a sample of the same full-size voxels
    ns = 10
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    z = np.zeros(ns, bool)
    ref = orc.fit_batch(T, sch, d_Y[:ns].cpu().numpy(), np.full(ns, 2), z, z, peaks[:ns], 2, False, False, None, None, 0,
                        nthreads=8)
    _check(P[:ns], ref, 2)


def test_k2_screening_kernel_equals_fp64_kernel_c2():
    """The split-FP16 screening kernel (fit_k2s.hip) against the FP64 kernel (fit_k2.hip) on 30 000 C2-shaped
    voxels: every output bit-identical.  Includes voxels built to stress the short list: single-fascicle
    signals (second weight ~0), nearly parallel peaks (ill-conditioned pairs -> interval bound) and identical
    peaks (every diagonal pair collinear -> hundreds of interval-bound candidates)."""
    import torch
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    sch, dic, rng = synth.make_model("C2")
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    V, N, M = 30000, ms.num_subs, sch.shape[0]
    dev = torch.device("cuda", 0)
    p1, p2 = synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)
    ax = np.cross(p1, [0.3, -0.5, 0.8]); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    for lo, hi, deg in ((0, 300, 16.0), (300, 600, 3.0), (600, 700, 0.2)):       # close crossings
        th = np.deg2rad(deg)
        p2[lo:hi] = p1[lo:hi] * np.cos(th) + ax[lo:hi] * np.sin(th)
    p2[700:760] = p1[700:760]                                                     # identical peaks
    peaks = np.concatenate([p1, p2], axis=1)
    atoms = rng.integers(0, N, (V, 2)).astype(np.int32)
    nu = rng.dirichlet(np.ones(2), V)
    nu[760:1100] = [1.0, 0.0]                                                     # one fascicle only
    nu[1100:1200] = [0.0, 1.0]
    d_pk = torch.from_numpy(peaks).to(dev)
    d_Y = torch.zeros((V, M), dtype=torch.float64, device=dev)
    for k in range(2):
        col = engine.rotate_columns_dev(plan, d_pk[:, 3 * k:3 * k + 3].contiguous(), torch.from_numpy(atoms[:, k].copy()).to(dev))
        d_Y += 500.0 * torch.from_numpy(nu[:, k:k + 1].copy()).to(dev) * col
    noise = rng.normal(0, 500 / 30.0, (V, M))
    noise[1200:1300] = 0.0                                                        # noise-free: exact atoms recoverable
    d_Y += torch.from_numpy(noise).to(dev)
    lib = L.lib()
    res = []
    nfb = []
    try:
        # screening kernel | FP64 kernel | screening kernel with a 32-entry short list (forces hand-backs to the FP64 kernel)
        for screen, cap in ((1, 0), (0, 0), (1, 32)):
            lib.mfx_debug_set_k2_screen(screen)
            lib.mfx_debug_set_k2s_cap(cap)
            out = torch.zeros((V, 7), dtype=torch.float64, device=dev)
            L.check(lib.mfx_fit_batch_dev(plan.handle(), d_Y.data_ptr(), d_pk.data_ptr(), 2, 0, 0, None, None, 0, V,
                                          out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
            torch.cuda.synchronize(dev)
            nfb.append(lib.mfx_debug_last_fallback_count() if screen else 0)
            res.append(out.cpu().numpy())
    finally:
        lib.mfx_debug_set_k2_screen(1)
        lib.mfx_debug_set_k2s_cap(0)
    bad = np.where(np.any(res[0] != res[1], axis=1))[0]
    assert bad.size == 0, "screening kernel differs from the FP64 kernel in voxels %s" % bad[:10]
    assert 0 <= nfb[0] < 0.02 * V, nfb     # hand-backs to the FP64 kernel stay rare (and are bit-identical anyway)
    bad = np.where(np.any(res[2] != res[1], axis=1))[0]
    assert bad.size == 0, "hand-back path differs from the FP64 kernel in voxels %s" % bad[:10]
    assert nfb[2] > 100, nfb               # the small short list did overflow: the hand-back path was exercised
    # noise-free two-fascicle voxels: the generating pair is recovered (or an exactly equivalent fit)
    assert np.max(res[0][1200:1300, 5]) < 1e-12 * 500 ** 2


def test_k2_full_size_bench_workload_screening_equals_fp64_kernel():
    """BASELINE config 2 at its full size, the very voxels bench.py times (1e5 voxels, 782 atoms x 200 measurements,
    seeds as in bench.py): every output of the screening kernel bit-identical to the FP64 kernel's; properties that
    do not depend on the size: fractions in [0, 1] summing to 1, atom ids inside the dictionary, MSE close to the
    noise variance, and a sample equal to the CPU oracle."""
    import torch
    import bench
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from oracle import oracle as orc
    V = 100000
    sch, dic, ms = bench.build_model(782)
    dev = torch.device("cuda", 0)
    ms.device = 0
    plan = engine.Plan(ms.device_tables(), scheme=sch)
    M, N = sch.shape[0], ms.num_subs
    rng = np.random.default_rng(1000)
    peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
    atoms = rng.integers(0, N, (V, 2)).astype(np.int32)
    nu = rng.dirichlet(np.ones(2), V)
    d_pk = torch.from_numpy(peaks).to(dev)
    d_Y = torch.zeros((V, M), dtype=torch.float64, device=dev)
    for k in range(2):
        col = engine.rotate_columns_dev(plan, d_pk[:, 3 * k:3 * k + 3].contiguous(), torch.from_numpy(atoms[:, k].copy()).to(dev))
        d_Y += 500.0 * torch.from_numpy(nu[:, k:k + 1].copy()).to(dev) * col
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    d_Y += torch.randn((V, M), dtype=torch.float64, device=dev, generator=gen) * (500.0 / 30.0)
    lib = L.lib()
    res = []
    try:
        for screen in (1, 0):
            lib.mfx_debug_set_k2_screen(screen)
            out = torch.zeros((V, 7), dtype=torch.float64, device=dev)
            L.check(lib.mfx_fit_batch_dev(plan.handle(), d_Y.data_ptr(), d_pk.data_ptr(), 2, 0, 0, None, None, 0, V,
                                          out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
            torch.cuda.synchronize(dev)
            res.append(out.cpu().numpy())
    finally:
        lib.mfx_debug_set_k2_screen(1)
    a, b = res
    bad = np.where(np.any(a != b, axis=1))[0]
    assert bad.size == 0, "screening kernel differs from the FP64 kernel in voxels %s" % bad[:10]
    # layout [M0, nu1, nu2, id1, id2, MSE, R2] (mf.py:375-450)
    assert np.all(a[:, 0] > 0) and np.all((a[:, 1:3] >= 0) & (a[:, 1:3] <= 1))
    np.testing.assert_allclose(a[:, 1] + a[:, 2], 1.0, rtol=0, atol=1e-12)
    assert np.all((a[:, 3:5] >= 0) & (a[:, 3:5] < N)) and np.all(a[:, 3:5] == np.round(a[:, 3:5]))
    assert abs(a[:, 5].mean() / (500.0 / 30.0) ** 2 - 1.0) < 0.05
    sel = np.arange(0, V, V // 6)[:6]
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    z = np.zeros(sel.size, bool)
    ref = orc.fit_batch(T, sch, d_Y[torch.from_numpy(sel).to(dev)].cpu().numpy(), np.full(sel.size, 2), z, z,
                        np.ascontiguousarray(peaks[sel]), 2, False, False, None, None, 0, nthreads=8)
    _check(a[sel], ref, 2)


def test_k2_screening_kernel_bracketed_protocol():
    """Protocol whose gradient strengths fall BETWEEN the dictionary's shells (UKBB-like: a handful of distinct G values,
    mf_utils.py:1827-1839): the screening kernel ranks through the plan's virtual shells (blend of the two bracketing
    shells on the union of their knots) and evaluates the short list with the reference's bracketing arithmetic.
    8 000 voxels, 782 atoms: bit-identical to the FP64 kernel, and equal to the oracle on a sample."""
    import torch
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    sch_ms, dic, rng = synth.make_model("C2")
    ms = mfu.init_PGSE_multishell_interp(dic, sch_ms, Z)
    Gs = ms["Gms_un"]
    sch = sch_ms[rng.permutation(sch_ms.shape[0])[:150]].copy()       # subject protocol: 150 of the 200 directions
    nz = np.where(sch[:, 3] > 0)[0]
    between = [0.3 * Gs[1] + 0.7 * Gs[2], 0.55 * Gs[2] + 0.45 * Gs[3], 0.9 * Gs[2] + 0.1 * Gs[3], 0.5 * (Gs[1] + Gs[2])]
    sch[nz[::2], 3] = rng.choice(between, size=nz[::2].size)          # half of the rows bracketed, 4 distinct G
    plan = ms.plan_for(sch)
    V, N, M = 8000, ms.num_subs, sch.shape[0]
    dev = torch.device("cuda", 0)
    peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
    atoms = rng.integers(0, N, (V, 2)).astype(np.int32)
    nu = rng.dirichlet(np.ones(2), V)
    d_pk = torch.from_numpy(peaks).to(dev)
    d_Y = torch.zeros((V, M), dtype=torch.float64, device=dev)
    for k in range(2):
        col = engine.rotate_columns_dev(plan, d_pk[:, 3 * k:3 * k + 3].contiguous(), torch.from_numpy(atoms[:, k].copy()).to(dev))
        d_Y += 500.0 * torch.from_numpy(nu[:, k:k + 1].copy()).to(dev) * col
    d_Y += torch.from_numpy(rng.normal(0, 500 / 30.0, (V, M))).to(dev)
    lib = L.lib()
    res = []
    try:
        for screen in (1, 0):
            lib.mfx_debug_set_k2_screen(screen)
            out = torch.zeros((V, 7), dtype=torch.float64, device=dev)
            L.check(lib.mfx_fit_batch_dev(plan.handle(), d_Y.data_ptr(), d_pk.data_ptr(), 2, 0, 0, None, None, 0, V,
                                          out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
            torch.cuda.synchronize(dev)
            if screen:
                nfb = lib.mfx_debug_last_fallback_count()
            res.append(out.cpu().numpy())
    finally:
        lib.mfx_debug_set_k2_screen(1)
    bad = np.where(np.any(res[0] != res[1], axis=1))[0]
    assert bad.size == 0, "screening kernel differs from the FP64 kernel in voxels %s" % bad[:10]
    assert nfb < 0.02 * V
    ns = 6
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    z = np.zeros(ns, bool)
    ref = orc.fit_batch(T, sch, d_Y[:ns].cpu().numpy(), np.full(ns, 2), z, z, peaks[:ns], 2, False, False, None, None, 0,
                        nthreads=8)
    _check(res[0][:ns], ref, 2)


def test_k2_fp64_kernel_exhaustive_last_resort():
    """The FP64 kernel's last resort when its short list overflows (every pair through the reference arithmetic),
    forced for every voxel by lowering the overflow threshold to 0: still the oracle's answer, exact-G and bracketed."""
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(5)
    sch_ms = synth.make_scheme(rng, 2, [1000, 2000, 3000], [20, 20, 20])
    dic = synth.make_dictionary(rng, sch_ms, 48)
    ms = mfu.init_PGSE_multishell_interp(dic, sch_ms, Z)
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    lib = L.lib()
    try:
        lib.mfx_debug_set_k2_screen(0)
        lib.mfx_debug_set_k2_maxc(0)
        for bracket in (False, True):
            sch = sch_ms.copy()
            if bracket:
                nz = np.where(sch[:, 3] > 0)[0]
                Gs = ms["Gms_un"]
                sch[nz[::3], 3] = 0.5 * (Gs[1] + Gs[2])
            V = 12
            peaks, Y, _, _ = synth.make_voxels(rng, V, 2, lambda d: np.stack([orc.interp(sch, x, T) for x in d]), 48)
            Y[0] = 300 * orc.interp(sch, peaks[0, :3], T)[:, 7]            # one atom suffices: massive near-ties
            z = np.zeros(V, bool)
            ref = orc.fit_batch(T, sch, Y, np.full(V, 2), z, z, peaks, 2, False, False, None, None, 0)
            got = engine.fit_batch(ms.plan_for(sch), Y, np.full(V, 2), None, None, peaks, 2, False, False)
            _check(got, ref, 2)
    finally:
        lib.mfx_debug_set_k2_screen(1)
        lib.mfx_debug_set_k2_maxc(-1)


@pytest.mark.parametrize("unit", [1e4, 1e-5])
def test_k2_dictionary_units(unit):
    """The screening kernel feeds D2 to the FP16 matrix pipe un-normalised: whatever units the dictionary is stored in
    (here x1e4 and x1e-5, signals scaled alike), the host pre-scales the FP32 screening table into the FP16 range and
    the results still equal the oracle's."""
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(77)
    sch = synth.make_scheme(rng, 2, [1000, 2000, 3000], [30, 30, 30])
    dic = synth.make_dictionary(rng, sch, 100) * unit
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    V = 32
    peaks, Y, _, _ = synth.make_voxels(rng, V, 2, lambda d: np.stack([orc.interp(sch, x, T) for x in d]), 100, snr=30.0 / unit)
    z = np.zeros(V, bool)
    ref = orc.fit_batch(T, sch, Y, np.full(V, 2), z, z, peaks, 2, False, False, None, None, 0)
    got = engine.fit_batch(ms.plan_for(sch), Y, np.full(V, 2), None, None, peaks, 2, False, False)
    ids = slice(3, 5)
    assert np.array_equal(got[:, ids], ref[:, ids]), "selected atom indices differ"
    assert np.allclose(got, ref, rtol=RTOL_W, atol=0)


@pytest.mark.parametrize("N,dirs", [(31, [20, 20, 20]), (32, [9, 9, 9]), (33, [40, 41, 42]), (64, [60, 60, 60]), (255, [20, 21, 20]),
                                    (257, [66, 66, 66]), (288, [30, 30, 30]), (289, [12, 12, 12]), (513, [40, 40, 40]),
                                    (545, [66, 67, 66]), (800, [33, 33, 33]), (100, [84, 84, 84]), (1100, [33, 33, 33])])
def test_k2_screening_kernel_shapes(N, dirs):
    """Tile/round/tail logic of the screening kernel across dictionary sizes (1 round, full rounds, a single leftover
    row tile shared by all waves: 9, 17 and 25 tiles) and protocol lengths (the KS = 4, 8, 13, 16 instantiations):
    bit-identical to the FP64 kernel on 256 voxels, equal to the oracle on a few."""
    import torch
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(1000 + N)
    sch = synth.make_scheme(rng, 2, [1000, 2000, 3000], dirs)
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    V, M = 256, sch.shape[0]
    dev = torch.device("cuda", 0)
    peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
    atoms = rng.integers(0, N, (V, 2)).astype(np.int32)
    nu = rng.dirichlet(np.ones(2), V)
    nu[:16] = [1.0, 0.0]
    d_pk = torch.from_numpy(peaks).to(dev)
    d_Y = torch.zeros((V, M), dtype=torch.float64, device=dev)
    for k in range(2):
        col = engine.rotate_columns_dev(plan, d_pk[:, 3 * k:3 * k + 3].contiguous(), torch.from_numpy(atoms[:, k].copy()).to(dev))
        d_Y += 500.0 * torch.from_numpy(nu[:, k:k + 1].copy()).to(dev) * col
    d_Y += torch.from_numpy(rng.normal(0, 500 / 30.0, (V, M))).to(dev)
    lib = L.lib()
    res = []
    try:
        # screening kernel (as many chunk images as fit: the one-barrier-per-chunk schedule) | FP64 kernel |
        # screening kernel forced to its two-image schedule
        for screen, images in ((1, 0), (0, 0), (1, 2)):
            lib.mfx_debug_set_k2_screen(screen)
            lib.mfx_debug_set_k2s_images(images)
            out = torch.zeros((V, 7), dtype=torch.float64, device=dev)
            L.check(lib.mfx_fit_batch_dev(plan.handle(), d_Y.data_ptr(), d_pk.data_ptr(), 2, 0, 0, None, None, 0, V,
                                          out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
            torch.cuda.synchronize(dev)
            res.append(out.cpu().numpy())
    finally:
        lib.mfx_debug_set_k2_screen(1)
        lib.mfx_debug_set_k2s_images(0)
    for r, what in ((res[0], "screening kernel"), (res[2], "screening kernel (two chunk images)")):
        bad = np.where(np.any(r != res[1], axis=1))[0]
        assert bad.size == 0, "%s differs from the FP64 kernel in voxels %s" % (what, bad[:10])
    ns = 4
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    z = np.zeros(ns, bool)
    ref = orc.fit_batch(T, sch, d_Y[:ns].cpu().numpy(), np.full(ns, 2), z, z, peaks[:ns], 2, False, False, None, None, 0,
                        nthreads=8)
    _check(res[0][:ns], ref, 2)


def test_bad_direction_raises():
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    rng = np.random.default_rng(3)
    sch = synth.make_scheme(rng, 1, [1000, 2000], [20, 20])
    ms = mfu.init_PGSE_multishell_interp(synth.make_dictionary(rng, sch, 20), sch, Z)
    peaks = np.array([[0, 0, 1.0, 0, 0, 1.2]])
    with pytest.raises(ValueError):   # mf_utils.py:1798-1802
        engine.fit_batch(ms.plan_for(sch), np.ones((1, sch.shape[0])), np.array([2]), None, None, peaks, 2, False, False)


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


# ---------------------------------------------------------------------------------------------
# mixed compartment classes (reference MFModel.fit goldens, mf.py:340-461 packing)
def _maps_from_params(P, maxfasc, csf_on, ear_on, props, DIFF_ear):
    """MFModelFit.__init__ semantics (mf.py:1068-1157) on flat ROI rows."""
    out = {"M0": P[:, 0]}
    for k in range(maxfasc):
        out["frac_f%d" % k] = P[:, 1 + k]
    for name, tab in props.items():
        tot = np.zeros(P.shape[0])
        for k in range(maxfasc):
            nu = P[:, 1 + k]
            pk = tab[P[:, 1 + maxfasc + k].astype(int)] * (nu > 0)
            tot += nu * pk
            out["%s_f%d" % (name, k)] = pk
        out[name + "_tot"] = tot
    if csf_on:
        out["frac_csf"] = P[:, 2 * maxfasc + 1]
    if ear_on:
        nu_e = P[:, 2 * maxfasc + csf_on + 1]
        out["frac_ear"] = nu_e
        out["D_ear"] = DIFF_ear[P[:, 2 * maxfasc + csf_on + 2].astype(int)] * (nu_e > 0)
    out["MSE"] = P[:, -2]
    out["R2"] = P[:, -1]
    return out


def _mixed_case():
    from oracle import oracle as orc
    d = np.load(os.path.join(G, "fit_cases.npz"))
    sch = d["sch"]
    b = (orc.GAMMA_H * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / float(d["T2_csf"])) * np.exp(-b * float(d["DIFF_csf"]))
    sig_ear = np.stack([np.exp(-sch[:, 6] / float(d["T2_ear"])) * np.exp(-b * x) for x in d["DIFF_ear"]], axis=1)
    return d, sch, sig_csf, sig_ear


def test_mixed_classes_reference_golden():
    """24-voxel ROI fitted by the reference's MFModel.fit: K in {0,1,2} x CSF x EAR (exercises _1, _2, _3,
    _4up), bracketed protocol rows, one all-zero voxel, one single-active tie voxel."""
    from microstructure_fingerprinting_amd import engine
    from microstructure_fingerprinting_amd import mf_utils as mfu
    d, sch, sig_csf, sig_ear = _mixed_case()
    ms = mfu.init_PGSE_multishell_interp(d["dictionary"], d["sch_ms"], Z)
    P = engine.fit_batch(ms.plan_for(sch), d["Y"], d["numfasc"], d["csf"], d["ear"], d["peaks"], 2, True, True,
                         sig_csf, sig_ear, int(d["E"]))
    maps = _maps_from_params(P, 2, 1, 1, {"rad": d["rad"], "fin": d["fin"]}, d["DIFF_ear"])
    for name, arr in maps.items():
        assert np.allclose(arr, d["map_" + name].reshape(-1), rtol=RTOL_W, atol=1e-9), name
    assert np.all(P[21] == 0)     # K=0, no CSF, no EAR: zeros (mf.py:387-388)


def test_host_pipeline_chunks_rows_and_buffer_pool():
    """mfx_fit_batch_rows on a mixed-class ROI large enough for several chunks (growing chunk sizes), with the ROI
    gather through `rows`, on re-used and re-grown device buffers, and after mfx_thread_release: every variant returns
    the rows of the one-call result."""
    from microstructure_fingerprinting_amd import _lib as L, engine
    from microstructure_fingerprinting_amd import mf_utils as mfu
    d, sch, sig_csf, sig_ear = _mixed_case()
    ms = mfu.init_PGSE_multishell_interp(d["dictionary"], d["sch_ms"], Z)
    plan = ms.plan_for(sch)
    args = lambda idx: (d["numfasc"][idx], d["csf"][idx], d["ear"][idx], d["peaks"][idx], 2, True, True, sig_csf, sig_ear, int(d["E"]))
    V0 = d["Y"].shape[0]
    base = engine.fit_batch(plan, d["Y"], *args(np.arange(V0)))
    rng = np.random.default_rng(3)
    for V in (3000, 9000):                    # chunks of 1024, 2048, ... voxels; the second call grows the pool
        idx = rng.integers(0, V0, V)
        got = engine.fit_batch(plan, d["Y"][idx], *args(idx))
        assert np.array_equal(got, base[idx])
    idx = rng.integers(0, V0, 5000)           # ROI gather: voxel v's signal is row rows[v] of a larger volume
    vol = np.concatenate([d["Y"], d["Y"][::-1]], axis=0)
    rows = np.where(rng.random(5000) < 0.5, idx, 2 * V0 - 1 - idx).astype(np.int64)
    got = engine.fit_batch(plan, vol, *args(idx), rows=rows)
    assert np.array_equal(got, base[idx])
    L.check(L.lib().mfx_thread_release())
    L.check(L.lib().mfx_thread_release())     # idempotent
    got = engine.fit_batch(plan, d["Y"][idx], *args(idx))
    assert np.array_equal(got, base[idx])


def test_k1_reference_golden_and_oracle():
    from microstructure_fingerprinting_amd import engine
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    d0, sch, sig_csf, sig_ear = _mixed_case()
    d = np.load(os.path.join(G, "fit_cases_k1.npz"))
    ms = mfu.init_PGSE_multishell_interp(d0["dictionary"], d0["sch_ms"], Z)
    V = d["Y"].shape[0]
    P = engine.fit_batch(ms.plan_for(sch), d["Y"], np.ones(V, int), None, None, d["peaks"], 1, False, False)
    maps = _maps_from_params(P, 1, 0, 0, {"rad": d0["rad"], "fin": d0["fin"]}, None)
    for name, arr in maps.items():
        assert np.allclose(arr, d["map_" + name].reshape(-1), rtol=RTOL_W, atol=1e-12), name
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    z = np.zeros(V, bool)
    ref = orc.fit_batch(T, sch, d["Y"], np.ones(V, int), z, z, d["peaks"], 1, False, False, None, None, 0)
    assert np.array_equal(P[:, 2], ref[:, 2]) and np.allclose(P, ref, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("K,c,e", [(1, 0, 0), (1, 1, 0), (1, 0, 1), (1, 1, 1), (0, 1, 0), (0, 0, 1), (0, 1, 1)])
def test_small_classes_vs_oracle(K, c, e):
    """Every K<=1 class on 40 random voxels (C1 shape: 100 atoms x 60 measurements) against the oracle,
    homogeneous batches through the device-pointer entry as well."""
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    sch, dic, rng = synth.make_model("C1")
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    N, E, V = ms.num_subs, 5, 40
    b = (orc.GAMMA_H * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3e-9)
    sig_ear = np.stack([np.exp(-sch[:, 6] / 0.08) * np.exp(-b * x) for x in np.linspace(0.2e-9, 1.2e-9, E)], axis=1)
    maxfasc = max(K, 1)
    peaks = synth.unit_vectors(rng, V)
    Y = rng.normal(0, 15.0, (V, sch.shape[0]))
    for v in range(V):
        comps = []
        if K:
            comps.append(orc.interp(sch, peaks[v], T)[:, rng.integers(0, N)])
        if c:
            comps.append(sig_csf)
        if e:
            comps.append(sig_ear[:, rng.integers(0, E)])
        nu = rng.dirichlet(np.ones(len(comps)))
        if v % 7 == 3:
            nu[rng.integers(0, len(comps))] = 0.0     # missing compartment -> single-active branches
        Y[v] += 400 * np.stack(comps, 1) @ nu
    Y[5] = -np.abs(Y[5])                                # nothing fits
    Kv = np.full(V, K)
    cm = np.full(V, bool(c)); em = np.full(V, bool(e))
    ref = orc.fit_batch(T, sch, Y, Kv, cm, em, peaks, maxfasc, bool(c), bool(e), sig_csf if c else None,
                        sig_ear if e else None, E if e else 0)
    got = engine.fit_batch(ms.plan_for(sch), Y, Kv, cm, em, peaks, maxfasc, bool(c), bool(e),
                           sig_csf if c else None, sig_ear if e else None, E if e else 0)
    assert np.allclose(got, ref, rtol=1e-12, atol=1e-12)
    ids = [1 + maxfasc + k for k in range(K)] + ([2 * maxfasc + c + 2] if e else [])
    assert np.array_equal(got[:, ids], ref[:, ids])


@pytest.mark.parametrize("c,e,N,V", [(1, 0, 48, 24), (0, 1, 48, 24), (1, 1, 48, 16), (1, 0, 130, 8)])
def test_k2_with_compartments_vs_oracle(c, e, N, V):
    """Two fascicles + CSF and/or EAR ([N,N,1], [N,N,E] -> _3; [N,N,1,E] -> _4up) against the oracle."""
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(100 + 10 * c + e)
    cfg = synth.config("C2")
    sch = synth.make_scheme(rng, cfg["n_b0"], cfg["shells_b"], cfg["dirs"])
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    E = 4
    b = (orc.GAMMA_H * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3e-9)
    sig_ear = np.stack([np.exp(-sch[:, 6] / 0.08) * np.exp(-b * x) for x in np.linspace(0.2e-9, 1.2e-9, E)], axis=1)
    peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
    Y = rng.normal(0, 500 / 30.0, (V, sch.shape[0]))
    for v in range(V):
        comps = [orc.interp(sch, peaks[v, :3], T)[:, rng.integers(0, N)], orc.interp(sch, peaks[v, 3:], T)[:, rng.integers(0, N)]]
        if c:
            comps.append(sig_csf)
        if e:
            comps.append(sig_ear[:, rng.integers(0, E)])
        nu = rng.dirichlet(np.ones(len(comps)))
        if v % 5 == 2:
            nu[rng.integers(0, len(comps))] = 0.0
        Y[v] += 500 * np.stack(comps, 1) @ nu
    Kv = np.full(V, 2)
    cm = np.full(V, bool(c)); em = np.full(V, bool(e))
    ref = orc.fit_batch(T, sch, Y, Kv, cm, em, peaks, 2, bool(c), bool(e), sig_csf if c else None,
                        sig_ear if e else None, E if e else 0, nthreads=8)
    got = engine.fit_batch(ms.plan_for(sch), Y, Kv, cm, em, peaks, 2, bool(c), bool(e),
                           sig_csf if c else None, sig_ear if e else None, E if e else 0)
    ids = [3, 4] + ([5 + c + 1] if e else [])
    if c and e:
        # _4up: the reference's residual comes out of scipy.optimize.nnls (norm of a Householder-transformed vector,
        # BLAS dnrm2): tuples that differ only in the atom of an INACTIVE compartment tie up to its rounding, so the
        # index of a compartment with zero weight is not reproducible (and is multiplied by zero in every map)
        got, ref = got.copy(), ref.copy()
        for col_nu, col_id in ((1, 3), (2, 4), (6, 7)):
            off = ref[:, col_nu] <= 1e-9
            got[off, col_id] = 0; ref[off, col_id] = 0
    assert np.array_equal(got[:, ids], ref[:, ids])
    tol = 1e-9 if not (c and e) else 1e-7     # _4up: third-party NNLS in the reference, Gram-based optimum here
    assert np.allclose(got, ref, rtol=tol, atol=1e-9)


@pytest.mark.parametrize("dirs,c,bracket", [([100, 100, 100], 0, False), ([137, 137, 137, 137], 0, False),
                                            ([100, 100, 100], 1, False), ([137, 137, 137, 137], 1, True)])
def test_k2_long_protocols(dirs, c, bracket):
    """Protocols longer than 200 measurements (M = 302 and M = 552, the HCP-MGH length): the one-wave-per-SIMD
    variants of the fused kernels, with and without CSF, exact-G and G-bracketed."""
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(500 + len(dirs) + 7 * c)
    shells = [1000, 2000, 3000, 5000][:len(dirs)]
    sch_ms = synth.make_scheme(rng, 2 + 2 * (len(dirs) - 3), shells, dirs)
    N, V = 40, 10
    dic = synth.make_dictionary(rng, sch_ms, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch_ms, Z)
    sch = sch_ms.copy()
    if bracket:
        nz = np.where(sch[:, 3] > 0)[0]
        Gs = ms["Gms_un"]
        sch[nz[::3], 3] = rng.uniform(Gs[1], Gs[-1], nz[::3].size)
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    b = (orc.GAMMA_H * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3e-9)
    peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
    Y = rng.normal(0, 500 / 30.0, (V, sch.shape[0]))
    for v in range(V):
        comps = [orc.interp(sch, peaks[v, :3], T)[:, rng.integers(0, N)], orc.interp(sch, peaks[v, 3:], T)[:, rng.integers(0, N)]]
        if c:
            comps.append(sig_csf)
        Y[v] += 500 * np.stack(comps, 1) @ rng.dirichlet(np.ones(len(comps)))
    Kv = np.full(V, 2)
    cm = np.full(V, bool(c)); z = np.zeros(V, bool)
    ref = orc.fit_batch(T, sch, Y, Kv, cm, z, peaks, 2, bool(c), False, sig_csf if c else None, None, 0, nthreads=8)
    got = engine.fit_batch(ms.plan_for(sch), Y, Kv, cm, z, peaks, 2, bool(c), False, sig_csf if c else None, None, 0)
    assert np.array_equal(got[:, 3:5], ref[:, 3:5])
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-9)


def _extras(sch, E):
    from oracle import oracle as orc
    b = (orc.GAMMA_H * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3e-9)
    sig_ear = np.stack([np.exp(-sch[:, 6] / 0.08) * np.exp(-b * x) for x in np.linspace(0.2e-9, 1.2e-9, E)], axis=1)
    return sig_csf, sig_ear


def _mixed_voxels(rng, ms, sch, T, N, V, sig_csf, sig_ear, kmax=2):
    from microstructure_fingerprinting_amd import synth
    from oracle import oracle as orc
    Kv = rng.integers(0, kmax + 1, V)
    cm = rng.random(V) < 0.5
    em = rng.random(V) < 0.5
    peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
    Y = rng.normal(0, 12.0, (V, sch.shape[0]))
    for v in range(V):
        comps = [orc.interp(sch, peaks[v, 3 * k:3 * k + 3], T)[:, rng.integers(0, N)] for k in range(Kv[v])]
        if cm[v]:
            comps.append(sig_csf)
        if em[v]:
            comps.append(sig_ear[:, rng.integers(0, sig_ear.shape[1])])
        if comps:
            Y[v] += 400 * np.stack(comps, 1) @ rng.dirichlet(np.ones(len(comps)))
    return Kv, cm, em, peaks, Y


def _mask_inactive_ids(got, ref):
    got, ref = got.copy(), ref.copy()
    for col_nu, col_id in ((1, 3), (2, 4), (6, 7)):
        off = ref[:, col_nu] <= 1e-9
        got[off, col_id] = 0; ref[off, col_id] = 0
    return got, ref


def test_out_of_limits_fallback_forced_on_every_class():
    """The path shapes beyond the fused kernels' limits take (voxel by voxel through the explicit-dictionary solver),
    forced on a small mixed ROI: every class K in {0, 1, 2} x CSF x EAR equals the fused kernels and the CPU restatement."""
    from microstructure_fingerprinting_amd import engine, synth, _lib as L
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(77)
    c = synth.config("C1")
    sch = synth.make_scheme(rng, c["n_b0"], c["shells_b"], c["dirs"])
    N, E, V = 40, 4, 72
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    sig_csf, sig_ear = _extras(sch, E)
    Kv, cm, em, peaks, Y = _mixed_voxels(rng, ms, sch, T, N, V, sig_csf, sig_ear)
    assert len(set(zip(Kv, cm, em))) == 12
    plan = ms.plan_for(sch)
    fused = engine.fit_batch(plan, Y, Kv, cm, em, peaks, 2, True, True, sig_csf, sig_ear, E)
    L.lib().mfx_debug_set_force_generic(1)
    try:
        slow = engine.fit_batch(plan, Y, Kv, cm, em, peaks, 2, True, True, sig_csf, sig_ear, E)
    finally:
        L.lib().mfx_debug_set_force_generic(0)
    ref = orc.fit_batch(T, sch, Y, Kv, cm, em, peaks, 2, True, True, sig_csf, sig_ear, E, nthreads=8)
    for got in (slow, fused):
        g, r = _mask_inactive_ids(got, ref)
        assert np.array_equal(g[:, [3, 4, 7]], r[:, [3, 4, 7]])
        assert np.allclose(g, r, rtol=1e-7, atol=1e-9)
    empty = (Kv == 0) & ~cm & ~em
    assert empty.any() and np.all(slow[empty] == 0)


@pytest.mark.parametrize("case", ["many_ear_columns", "long_protocol", "large_dictionary"])
def test_shapes_beyond_the_fused_kernels_limits(case):
    """Shapes the fused kernels refuse (more than 16 CSF+EAR columns; more than 560 measurements; a dictionary whose
    rotated chunks do not fit the LDS) are fitted all the same, against the CPU restatement."""
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng({"many_ear_columns": 5, "long_protocol": 6, "large_dictionary": 7}[case])
    if case == "many_ear_columns":
        N, E, V, dirs, kmax = 24, 20, 36, [29, 29], 2
    elif case == "long_protocol":
        N, E, V, dirs, kmax = 24, 3, 24, [200, 200, 200], 2
    else:
        N, E, V, dirs, kmax = 3000, 2, 3, [66, 66, 66], 2
    sch = synth.make_scheme(rng, 2, [1000, 2000, 3000][:len(dirs)], dirs)
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    T = _oracle_tables(ms) | {"scheme_DeldelTE": ms["scheme_DeldelTE"]}
    sig_csf, sig_ear = _extras(sch, E)
    Kv, cm, em, peaks, Y = _mixed_voxels(rng, ms, sch, T, N, V, sig_csf, sig_ear, kmax)
    if case == "large_dictionary":
        Kv[:] = 2; cm[:] = [False, True, False]; em[:] = False
        for v in range(V):
            Y[v] = 400 * (0.6 * orc.interp(sch, peaks[v, :3], T)[:, 7 + v] + 0.4 * orc.interp(sch, peaks[v, 3:], T)[:, 2900 - v]) + rng.normal(0, 10, sch.shape[0])
    got = engine.fit_batch(ms.plan_for(sch), Y, Kv, cm, em, peaks, 2, True, True, sig_csf, sig_ear, E)
    ref = orc.fit_batch(T, sch, Y, Kv, cm, em, peaks, 2, True, True, sig_csf, sig_ear, E, nthreads=8)
    g, r = _mask_inactive_ids(got, ref)
    assert np.array_equal(g[:, [3, 4, 7]], r[:, [3, 4, 7]])
    assert np.allclose(g, r, rtol=1e-7, atol=1e-9)
