"""Oracle-refereed parity at FULL dictionary size, in the regimes that stress the kernels' short lists.

The fused kernels rank in reduced precision (fit_k2s: split-FP16 MFMA) or through feasible-support scores
(fit_k2x) and decide on a short list in the reference's arithmetic; these tests let the CPU oracle -- not another
HIP kernel -- referee them where that matters:
  * >= 2400 config-2 voxels (782 atoms x 200 measurements) from the regimes of tools/dev_regimes.py;
  * the reference tests' own real dictionaries (tests/golden/real_*.npz: UKBB 271 x 986 with near-duplicate atoms and a
    G-bracketed subject protocol, through MFModel.fit; HCP-MGH 552 x 782 through rotate_atom tables);
  * config 4 ([782, 782, 1, 10]) at full size incl. noise-free and pure-CSF voxels, and the flood cases of the
    two-fascicle + CSF/EAR kernel at small size.
Bar: atom indices bit-exact, weights within 1e-5 relative (north_star); the kernels in fact reproduce the oracle to
the last bits for _1/_2/_3 and to ~1e-9 for _4up (third-party Lawson-Hanson arithmetic in the reference).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
Z = np.array([0.0, 0.0, 1.0])
NTHREADS = max(1, min(os.cpu_count() or 1, 64))


def _tables(ms):
    return {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat,
            "scheme_DeldelTE": ms["scheme_DeldelTE"]}


def _rotate_cols(plan, dirs, cols):
    """[B, M] rotated single atoms through the library's host entry point (signal synthesis for the tests)."""
    from microstructure_fingerprinting_amd import _lib as L
    dirs = np.ascontiguousarray(dirs, dtype=np.float64)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    out = np.zeros((dirs.shape[0], plan.M))
    L.check(L.lib().mfx_rotate_cols(plan.handle(), L.dptr(dirs), L.iptr(cols), dirs.shape[0], 0, L.dptr(out)))
    return out


def _second_peak(rng, p1, deg):
    r = rng.standard_normal(p1.shape)
    r -= (r * p1).sum(1, keepdims=True) * p1
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    a = np.deg2rad(deg)
    p2 = np.cos(a) * p1 + np.sin(a) * r
    return p2 / np.linalg.norm(p2, axis=1, keepdims=True)


def _assert_rows(got, ref, maxfasc, what, rtol=1e-5, ids_where_active=False, ear=None):
    """ids_where_active: compare an atom index only where the oracle gives its compartment a positive weight.  Used for
    the _4up class only: there the reference's residual is the norm of a Householder-transformed vector inside
    scipy.optimize.nnls, whose last bits (and with them the winner among EXACTLY tied tuples - all of them when a
    compartment is inactive) depend on SciPy's BLAS (dnrm2, dlarfgp: tests/golden/nnls_cases.npz pins the solver to
    rounding, not bit for bit); the index of an inactive compartment is multiplied by zero in every map the reference
    produces (mf.py:1111-1117, 1148).  ear = (column of nu_ear, column of the EAR index)."""
    ids = slice(1 + maxfasc, 1 + 2 * maxfasc)
    if ids_where_active:
        got, ref = got.copy(), ref.copy()
        off = ref[:, 1:1 + maxfasc] <= 1e-9   # (Lawson-Hanson leaves ~1e-14 on atoms it could have dropped)
        got[:, ids][off] = 0; ref[:, ids][off] = 0
        if ear is not None:
            off = ref[:, ear[0]] <= 1e-9
            got[off, ear[1]] = 0; ref[off, ear[1]] = 0
    bad = np.where(np.any(got[:, ids] != ref[:, ids], axis=1))[0]
    assert bad.size == 0, "%s: atom indices differ from the oracle in voxels %s (got %s, oracle %s)" % (
        what, bad[:8], got[bad[:3], ids], ref[bad[:3], ids])
    assert np.allclose(got, ref, rtol=rtol, atol=1e-9), "%s: parameters differ from the oracle" % what
    return float(np.max(np.abs(got - ref) / (np.abs(ref) + 1e-300) * (np.abs(ref) > 1e-6)))


REGIMES = [  # name, snr (0: noise-free), crossing angle in degrees (None: random), nu0 (None: Dirichlet)
    ("snr30", 30, None, None), ("snr10", 10, None, None), ("snr100", 100, None, None), ("noise_free", 0, None, None),
    ("cross15", 30, 15.0, None), ("cross3", 30, 3.0, None), ("cross0.2", 30, 0.2, None), ("identical", 30, 0.0, None),
    ("nu95", 30, None, 0.95), ("one_fascicle", 30, None, 1.0), ("one_fascicle_noise_free", 0, None, 1.0),
    ("second_only", 30, None, 0.0),
]


def test_k2_stress_regimes_vs_oracle_full_size():
    """2400 voxels at BASELINE config 2's size (782 atoms x 200 measurements) from twelve regimes, screening kernel
    (the default path) against the CPU oracle: one dominant fascicle, nu = (0.95, 0.05), crossings of 0.2 / 3 / 15
    degrees, identical peaks, noise-free data, SNR 10 / 30 / 100.  Every row must agree."""
    import torch
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    sch, dic, rng = synth.make_model("C2")
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    N, M = ms.num_subs, sch.shape[0]
    per = int(os.environ.get("MFX_STRESS_PER_REGIME", "200"))   # (one-off deep runs: e.g. 2000)
    pk, Ys, names = [], [], []
    for name, snr, deg, nu0 in REGIMES:
        p1 = synth.unit_vectors(rng, per)
        p2 = synth.unit_vectors(rng, per) if deg is None else (p1.copy() if deg == 0.0 else _second_peak(rng, p1, deg))
        atoms = rng.integers(0, N, (per, 2))
        nu = rng.dirichlet(np.ones(2), per) if nu0 is None else np.tile([nu0, 1.0 - nu0], (per, 1))
        Y = 500.0 * nu[:, :1] * _rotate_cols(plan, p1, atoms[:, 0]) + 500.0 * nu[:, 1:] * _rotate_cols(plan, p2, atoms[:, 1])
        if snr:
            Y = Y + rng.normal(0, 500.0 / snr, Y.shape)
        pk.append(np.concatenate([p1, p2], axis=1)); Ys.append(Y); names += [name] * per
    peaks, Y = np.concatenate(pk), np.concatenate(Ys)
    V = Y.shape[0]
    assert V >= 2000
    dev = torch.device("cuda", 0)
    lib = L.lib()
    lib.mfx_debug_set_k2_screen(1)
    out = engine.fit_batch_dev(plan, torch.from_numpy(Y).to(dev), torch.from_numpy(peaks).to(dev), 2)
    got = out.cpu().numpy()
    nfb, ngd = lib.mfx_debug_last_fallback_count(), lib.mfx_debug_last_guard_count()
    aud_over, aud_max, aud_n = lib.mfx_debug_last_counter(8), lib.mfx_debug_last_counter(9) * 1e-11, lib.mfx_debug_last_counter(10)
    z = np.zeros(V, bool)
    ref = orc.fit_batch(_tables(ms), sch, Y, np.full(V, 2), z, z, peaks, 2, False, False, None, None, 0, nthreads=NTHREADS)
    names = np.array(names)
    for name, *_ in REGIMES:
        sel = names == name
        _assert_rows(got[sel], ref[sel], 2, "regime %s" % name)
    assert np.array_equal(got[:, 3:5], ref[:, 3:5])
    print("stress regimes: %d voxels, ids exact, max rel err %.2e, handed back %d (guard %d)"
          % (V, float(np.max(np.abs(got - ref) / (np.abs(ref) + 1e-300) * (np.abs(ref) > 1e-6))), nfb, ngd))
    assert nfb < 0.05 * V
    # population audit: (nearly) every voxel compared one pseudo-random pair's split-FP16 cosine with its FP64 value
    print("screen audit: %d pairs, max |c~ - c| %.2e, beyond a quarter of the margin: %d" % (aud_n, aud_max, aud_over))
    assert aud_n >= 0.9 * (V - nfb) and aud_over == 0 and 0.0 < aud_max < 2.5e-6


def test_k2_worst_operand_family_as_dictionary_vs_oracle():
    """The operand family with the largest measured split-FP16 product error (tools/micro/split_mfma_error.hip, family 1:
    smooth positive decays over the measurement index at table scale, near-collinear pairs; worst |c~ - c| 1.05e-6 at
    K = 208) fed through the WHOLE two-fascicle fit as a dictionary, against the oracle.  The atoms
    are isotropic - a_n(m) = A_n exp(-3 p_n s(m) / S) with s(m) the shell of measurement m, 67 shells x 3 directions -
    so that every rotation returns exactly these vectors and both rotated dictionaries are the family itself: 782 atoms x
    200 measurements, the screening kernel's config-2 instantiation.  Mixtures of two atoms at SNR 30 / 100, noise-free,
    one atom only.  Indices must equal the oracle's (pairs (i, j) and (j, i) tie exactly: first hit), weights to 1e-9;
    the run-time guard must stay silent."""
    import torch
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(2024)
    S, N = 67, 782                                    # a b0 shell (2 rows) + 66 shells x 3 directions = 200 measurements
    sch = synth.make_scheme(rng, 2, list(np.linspace(200.0, 10000.0, S - 1)), [3] * (S - 1))
    az = rng.uniform(0, 2 * np.pi, sch.shape[0])
    uz = np.tile([0.15, 0.5, 0.85], S)[1:]            # |g.z| of the three directions of a shell: well separated knots
    sch[2:, 0], sch[2:, 1], sch[2:, 2] = (np.sqrt(1 - uz ** 2) * np.cos(az))[2:], (np.sqrt(1 - uz ** 2) * np.sin(az))[2:], uz[2:]
    M = sch.shape[0]
    assert M == 200
    shell = np.concatenate([[0, 0], np.repeat(np.arange(1, S), 3)])
    amp, rate = 0.3 + 0.7 * rng.random(N), rng.random(N)
    dic = amp[None, :] * np.exp(-3.0 * rate[None, :] * shell[:, None] / S)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    V = int(os.environ.get("MFX_WORST_FAMILY_V", "384"))
    peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
    atoms = rng.integers(0, N, (V, 2))
    nu = rng.dirichlet(np.ones(2), V)
    nu[V // 2: V // 2 + V // 8] = [1.0, 0.0]
    Y = 500.0 * (nu[:, :1] * dic[:, atoms[:, 0]].T + nu[:, 1:] * dic[:, atoms[:, 1]].T)
    q = V // 4
    Y[:q] += rng.normal(0, 500.0 / 30.0, (q, M))
    Y[q:2 * q] += rng.normal(0, 500.0 / 100.0, (q, M))
    Y[3 * q:] += rng.normal(0, 500.0 / 30.0, (V - 3 * q, M))
    dev = torch.device("cuda", 0)
    lib = L.lib()
    lib.mfx_debug_set_k2_screen(1)
    got = engine.fit_batch_dev(plan, torch.from_numpy(Y).to(dev), torch.from_numpy(peaks).to(dev), 2).cpu().numpy()
    nfb, ngd = lib.mfx_debug_last_fallback_count(), lib.mfx_debug_last_guard_count()
    aud_over, aud_max, aud_n = lib.mfx_debug_last_counter(8), lib.mfx_debug_last_counter(9) * 1e-11, lib.mfx_debug_last_counter(10)
    z = np.zeros(V, bool)
    ref = orc.fit_batch(_tables(ms), sch, Y, np.full(V, 2), z, z, peaks, 2, False, False, None, None, 0, nthreads=NTHREADS)
    _assert_rows(got, ref, 2, "worst operand family", rtol=1e-9)
    assert ngd == 0, "screening-error guard tripped on %d voxels" % ngd
    print("worst operand family: %d voxels identical to the oracle, handed back %d, guard %d; audit: %d pairs, max |c~ - c| %.2e"
          % (V, nfb, ngd, aud_n, aud_max))
    assert aud_over == 0 and aud_max < 2.5e-6


def _ukbb_model():
    d = np.load(os.path.join(G, "real_ukbb.npz"))
    model = {k: d[k] for k in d.files if k != "sch_subj"}
    for k in ("num_atom", "num_ear"):
        model[k] = int(model[k])
    for k in ("T2_csf", "DIFF_csf", "T2_ear"):
        model[k] = float(model[k])
    model["fasc_propnames"] = ["rad", "fin"]
    return model, np.ascontiguousarray(d["sch_subj"])


def test_ukbb_real_dictionary_through_mfmodel_vs_oracle():
    """The reference tests' UKBB dictionary (271 x 986 atoms, near-duplicate atoms, 10 EAR columns) with subject
    1000521's protocol (105 rows, G-bracketed) through MFModel.fit: 640 two-fascicle voxels + 32 with CSF (reference
    _3) + 8 with CSF and EAR (_4up), incl. voxels generated from the dictionary's most similar atom pairs, against the
    oracle; and the voxels the reference itself fitted (tests/golden/real_ukbb_fit_*.npz)."""
    from microstructure_fingerprinting_amd import mf, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    model, sch = _ukbb_model()
    m = mf.MFModel(model)
    ms = m.ms_interpolator
    plan = ms.plan_for(sch)
    N, M, E = model["num_atom"], sch.shape[0], model["num_ear"]
    rng = np.random.default_rng(5)
    # the most similar atom pairs of the canonical dictionary (cosine closest to 1)
    D = model["dictionary"] / np.linalg.norm(model["dictionary"], axis=0)
    C = D.T @ D
    np.fill_diagonal(C, 0.0)
    twins = np.argsort(C.max(axis=1))[::-1][:64]
    nK2, nC, nCE = 640, 32, 8
    V = nK2 + nC + nCE
    p1, p2 = synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)
    p2[:40] = _second_peak(rng, p1[:40], 2.0)
    p2[40:60] = p1[40:60]
    atoms = rng.integers(0, N, (V, 2))
    atoms[60:124, 0] = twins                      # a twin atom in fascicle 0 ...
    atoms[124:188, 1] = twins                     # ... in fascicle 1
    csf = np.zeros(V, bool); csf[nK2:] = True
    ear = np.zeros(V, bool); ear[nK2 + nC:] = True
    gam = mfu.get_gyromagnetic_ratio('H')
    b = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / model["T2_csf"]) * np.exp(-b * model["DIFF_csf"])
    sig_ear = np.stack([np.exp(-sch[:, 6] / model["T2_ear"]) * np.exp(-b * Dk) for Dk in model["DIFF_ear"]], axis=1)
    nu = rng.dirichlet(np.ones(4), V)
    nu[:, 2] *= csf; nu[:, 3] *= ear
    nu[188:220, 1] = 0.0                          # one fascicle only
    nu /= nu.sum(1, keepdims=True)
    Y = 500.0 * (nu[:, :1] * _rotate_cols(plan, p1, atoms[:, 0]) + nu[:, 1:2] * _rotate_cols(plan, p2, atoms[:, 1])
                 + nu[:, 2:3] * sig_csf + nu[:, 3:4] * sig_ear[:, rng.integers(0, E, V)].T)
    Y += rng.normal(0, 500.0 / 30.0, Y.shape)
    peaks = np.concatenate([p1, p2], axis=1)
    mask = np.ones(V, dtype=int)
    fit = m.fit(Y, mask, 2, peaks=peaks, pgse_scheme=sch, csf_mask=csf.astype(int), ear_mask=ear.astype(int), verbose=0)
    got = fit.params_in_mask
    ref = orc.fit_batch(_tables(ms), sch, Y, np.full(V, 2), csf, ear, peaks, 2, True, True, sig_csf, sig_ear, E,
                        nthreads=NTHREADS)
    _assert_rows(got[:nK2], ref[:nK2], 2, "UKBB K=2")
    _assert_rows(got[nK2:nK2 + nC], ref[nK2:nK2 + nC], 2, "UKBB K=2+CSF")
    _assert_rows(got[nK2 + nC:], ref[nK2 + nC:], 2, "UKBB K=2+CSF+EAR", ids_where_active=True, ear=(6, 7))
    act = ref[:, 6] > 1e-9
    assert np.array_equal(got[act, 7], ref[act, 7])   # EAR atom
    # voxels fitted by the reference itself
    for name in ("k2", "k2csf", "k2csfear"):
        fn = os.path.join(G, "real_ukbb_fit_%s.npz" % name)
        if not os.path.exists(fn):
            continue
        c = np.load(fn)
        Vr = c["Y"].shape[0]
        one = np.ones(Vr, dtype=int)
        f = m.fit(c["Y"], one, 2, peaks=c["peaks"], pgse_scheme=sch, csf_mask=(one if int(c["csf"]) else None),
                  ear_mask=(one if int(c["ear"]) else None), verbose=0)
        for pn in c["param_names"]:
            a, r = np.asarray(getattr(f, str(pn))), c["map_" + str(pn)]
            assert np.allclose(a, r, rtol=1e-5, atol=1e-12), "reference golden %s: map %s differs" % (name, pn)


def test_hcp_real_dictionary_long_protocol_vs_oracle():
    """The reference's HCP-MGH dictionary (552 rows x 782 atoms: test_hcp_dict's shape of use) as a dense dictionary
    fitted on its own 552-row protocol: two-fascicle voxels against the oracle (M = 552 > 256: the long-protocol path)."""
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    d = np.load(os.path.join(G, "real_hcp.npz"))
    sch, dic = np.ascontiguousarray(d["sch_mat"]), np.ascontiguousarray(d["dictionary"])
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    N = ms.num_subs
    rng = np.random.default_rng(9)
    V = 96
    p1, p2 = synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)
    p2[:8] = _second_peak(rng, p1[:8], 3.0)
    atoms = rng.integers(0, N, (V, 2))
    atoms[8:16] = 86                               # the atom the reference's own test recovers
    nu = rng.dirichlet(np.ones(2), V)
    nu[16:24] = [1.0, 0.0]
    Y = 500.0 * (nu[:, :1] * _rotate_cols(plan, p1, atoms[:, 0]) + nu[:, 1:] * _rotate_cols(plan, p2, atoms[:, 1]))
    Y += rng.normal(0, 500.0 / 30.0, Y.shape)
    Y[24:32] = 500.0 * (nu[24:32, :1] * _rotate_cols(plan, p1[24:32], atoms[24:32, 0]) + nu[24:32, 1:] * _rotate_cols(plan, p2[24:32], atoms[24:32, 1]))
    peaks = np.concatenate([p1, p2], axis=1)
    got = engine.fit_batch(plan, Y, np.full(V, 2), None, None, peaks, 2, False, False)
    z = np.zeros(V, bool)
    ref = orc.fit_batch(_tables(ms), sch, Y, np.full(V, 2), z, z, peaks, 2, False, False, None, None, 0, nthreads=NTHREADS)
    _assert_rows(got, ref, 2, "HCP 552 x 782")


def _c4_model(N, E, rng=None):
    from microstructure_fingerprinting_amd import synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    sch, dic, rng = synth.make_model("C2", N=N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    gam = mfu.get_gyromagnetic_ratio('H')
    b = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3.0e-9)
    sig_ear = np.stack([np.exp(-sch[:, 6] / 0.08) * np.exp(-b * Dk) for Dk in np.linspace(0.2e-9, 1.2e-9, E)], axis=1)
    return sch, ms, sig_csf, sig_ear, rng


def _flood_voxels(rng, plan, N, E, sig_csf, sig_ear, V):
    """Voxels whose optimum leaves a fascicle atom (or both) inactive: every tuple sharing the active atoms ties."""
    from microstructure_fingerprinting_amd import synth
    M = plan.M
    p1, p2 = synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)
    atoms = rng.integers(0, N, (V, 2))
    nu = rng.dirichlet(np.ones(4), V)              # f0, f1, csf, ear
    kind = np.arange(V) % 8
    nu[kind == 0] = [0, 0, 1, 0]                   # pure CSF
    nu[kind == 1, 1] = 0                           # fascicle 0 + CSF + EAR only
    nu[kind == 2, 0] = 0                           # fascicle 1 + ...
    nu[kind == 3, 2:] = 0                          # no CSF / EAR signal although the columns are offered
    nu[kind == 4] = [0, 0, 0, 1]                   # pure EAR
    p2[kind == 5] = p1[kind == 5]                  # identical peaks
    nu /= nu.sum(1, keepdims=True)
    e_id = rng.integers(0, E, V)
    Y = 500.0 * (nu[:, :1] * _rotate_cols(plan, p1, atoms[:, 0]) + nu[:, 1:2] * _rotate_cols(plan, p2, atoms[:, 1])
                 + nu[:, 2:3] * sig_csf + nu[:, 3:4] * sig_ear[:, e_id].T)
    noise = rng.normal(0, 500.0 / 30.0, (V, M))
    noise[kind == 6] = 0.0                         # noise-free
    nf_csf = (kind == 0) & (np.arange(V) % 16 == 0)
    noise[nf_csf] = 0.0                            # noise-free pure CSF: EVERY tuple ties -> the exhaustive pass is the answer
    return np.concatenate([p1, p2], axis=1), Y + noise, int(nf_csf.sum())


@pytest.mark.parametrize("c,e", [(1, 0), (0, 1), (1, 1)])
def test_k2x_flood_voxels_vs_oracle(c, e):
    """Two fascicles + CSF and/or EAR at N = 64, E = 4 on voxels whose optimum has inactive atoms (pure CSF / EAR, one
    fascicle, noise-free, identical peaks): the ties used to flood the 256-entry short list, which dropped entries
    silently; now the families are evaluated exactly.  Only a voxel in which EVERY tuple ties (noise-free pure CSF) may
    need the exhaustive last resort."""
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine
    from oracle import oracle as orc
    N, E, V = 64, 4, 64
    sch, ms, sig_csf, sig_ear, rng = _c4_model(N, E)
    plan = ms.plan_for(sch)
    peaks, Y, n_all_tie = _flood_voxels(rng, plan, N, E, sig_csf * c, sig_ear * e, V)
    one = np.ones(V, bool)
    got = engine.fit_batch(plan, Y, np.full(V, 2), one * c, one * e, peaks, 2, bool(c), bool(e), sig_csf if c else None,
                           sig_ear if e else None, E if e else 0)
    nfb = L.lib().mfx_debug_last_fallback_count()
    ref = orc.fit_batch(_tables(ms), sch, Y, np.full(V, 2), one * c, one * e, peaks, 2, bool(c), bool(e),
                        sig_csf if c else None, sig_ear if e else None, E if e else 0, nthreads=NTHREADS)
    _assert_rows(got, ref, 2, "flood voxels csf=%d ear=%d" % (c, e), rtol=1e-5 if (c and e) else 1e-9,
                 ids_where_active=bool(c and e), ear=(6, 7) if (c and e) else None)
    assert nfb <= n_all_tie, "exhaustive pass taken by %d voxels" % nfb


def test_k2x_exhaustive_last_resort():
    """A short list that overflows (forced by a zero cap) must trigger the exhaustive exact pass, never a silent drop."""
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine
    from oracle import oracle as orc
    N, E, V = 40, 3, 12
    sch, ms, sig_csf, sig_ear, rng = _c4_model(N, E)
    plan = ms.plan_for(sch)
    peaks, Y, _ = _flood_voxels(rng, plan, N, E, sig_csf, sig_ear, V)
    one = np.ones(V, bool)
    lib = L.lib()
    try:
        lib.mfx_debug_set_k2x_maxc(0)
        got = engine.fit_batch(plan, Y, np.full(V, 2), one, one, peaks, 2, True, True, sig_csf, sig_ear, E)
        nfb = lib.mfx_debug_last_fallback_count()
    finally:
        lib.mfx_debug_set_k2x_maxc(-1)
    ref = orc.fit_batch(_tables(ms), sch, Y, np.full(V, 2), one, one, peaks, 2, True, True, sig_csf, sig_ear, E, nthreads=NTHREADS)
    _assert_rows(got, ref, 2, "exhaustive pass", ids_where_active=True, ear=(6, 7))
    assert nfb == V


def test_c4_full_size_vs_oracle():
    """BASELINE config 4 at full size: sub-dictionaries [782, 782, 1, 10], 200 measurements (reference _4up: one NNLS
    per tuple, 6.1e6 tuples per voxel -> tens of seconds per voxel for the oracle, one voxel per host thread), and the
    two _3 classes [782, 782, 1] / [782, 782, 10]; incl. noise-free and pure-CSF voxels."""
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine
    from oracle import oracle as orc
    N, E = 782, 10
    sch, ms, sig_csf, sig_ear, rng = _c4_model(N, E)
    plan = ms.plan_for(sch)
    V = max(8, min(NTHREADS, 16))
    peaks, Y, n_all_tie = _flood_voxels(rng, plan, N, E, sig_csf, sig_ear, V)
    one = np.ones(V, bool)
    got = engine.fit_batch(plan, Y, np.full(V, 2), one, one, peaks, 2, True, True, sig_csf, sig_ear, E)
    nfb = L.lib().mfx_debug_last_fallback_count()
    ref = orc.fit_batch(_tables(ms), sch, Y, np.full(V, 2), one, one, peaks, 2, True, True, sig_csf, sig_ear, E, nthreads=NTHREADS)
    _assert_rows(got, ref, 2, "config 4", ids_where_active=True, ear=(6, 7))
    act = ref[:, 6] > 1e-9
    assert np.array_equal(got[act, 7], ref[act, 7]), "EAR atom differs"
    assert nfb <= n_all_tie
    for c, e in ((1, 0), (0, 1)):
        pk2, Y2, _ = _flood_voxels(rng, plan, N, E, sig_csf * c, sig_ear * e, V)
        g2 = engine.fit_batch(plan, Y2, np.full(V, 2), one * c, one * e, pk2, 2, bool(c), bool(e), sig_csf if c else None,
                              sig_ear if e else None, E if e else 0)
        r2 = orc.fit_batch(_tables(ms), sch, Y2, np.full(V, 2), one * c, one * e, pk2, 2, bool(c), bool(e),
                           sig_csf if c else None, sig_ear if e else None, E if e else 0, nthreads=NTHREADS)
        _assert_rows(g2, r2, 2, "[782,782,%s]" % ("1" if c else "10"), rtol=1e-9)


def test_dev_entry_point_is_asynchronous_and_reports_bad_directions():
    """mfx_fit_batch_dev only enqueues (include/mfx.h): a non-unit fascicle direction -- the reference's per-voxel
    ValueError (mf_utils.py:1798-1802) -- is flagged by the kernels and raised by mfx_plan_status."""
    import torch
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    sch, dic, rng = synth.make_model("C2", N=64)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    dev = torch.device("cuda", 0)
    V = 16
    peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
    Y = rng.uniform(10, 100, (V, sch.shape[0]))
    good = engine.fit_batch_dev(plan, torch.from_numpy(Y).to(dev), torch.from_numpy(peaks).to(dev), 2)
    assert bool(torch.isfinite(good).all())
    peaks[5, 3:] *= 1.01
    with pytest.raises(ValueError, match="unit norm"):
        engine.fit_batch_dev(plan, torch.from_numpy(Y).to(dev), torch.from_numpy(peaks).to(dev), 2)
    # the flag is cleared by the report: the next clean batch passes
    peaks[5, 3:] /= 1.01
    engine.fit_batch_dev(plan, torch.from_numpy(Y).to(dev), torch.from_numpy(peaks).to(dev), 2)
    # one fascicle, CSF + EAR on the device path (extras built on the device, no host round trip)
    gam = mfu.get_gyromagnetic_ratio('H')
    b = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3.0e-9)
    sig_ear = np.stack([np.exp(-sch[:, 6] / 0.08) * np.exp(-b * Dk) for Dk in (0.3e-9, 0.9e-9)], axis=1)
    out = engine.fit_batch_dev(plan, torch.from_numpy(Y).to(dev), torch.from_numpy(peaks[:, :3].copy()).to(dev), 1, True, True,
                               torch.from_numpy(sig_csf).to(dev), torch.from_numpy(np.ascontiguousarray(sig_ear)).to(dev), 2)
    host = engine.fit_batch(plan, Y, np.ones(V, int), np.ones(V, bool), np.ones(V, bool), peaks[:, :3], 1, True, True,
                            sig_csf, sig_ear, 2)
    assert np.array_equal(out.cpu().numpy(), host)


def test_dev_entry_point_does_not_block_with_calls_in_flight():
    """Four back-to-back mfx_fit_batch_dev calls on one stream (config 2's size, 20 000 voxels each: ~18 ms of kernel
    per call) must all return long before the first one has finished - the host only enqueues (include/mfx.h) - and give
    the same rows as a call made alone.  (Regression: scratch memory released with hipFreeAsync blocked the host for a
    kernel's duration from the second call in flight on; the library's scratch now comes from per-stream arenas.)"""
    import time
    import torch
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    sch, dic, rng = synth.make_model("C2", N=782)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    dev = torch.device("cuda", 0)
    V = 20000
    peaks, Y = _c2_like_voxels(rng, plan, 782, V)
    dY, dpk = torch.from_numpy(Y).to(dev), torch.from_numpy(peaks).to(dev)
    alone = engine.fit_batch_dev(plan, dY, dpk, 2)
    torch.cuda.synchronize()
    outs = [torch.zeros_like(alone) for _ in range(4)]
    lib, st = L.lib(), torch.cuda.current_stream(dev).cuda_stream
    t0 = time.perf_counter()
    for o in outs:
        L.check(lib.mfx_fit_batch_dev(plan.handle(), dY.data_ptr(), dpk.data_ptr(), 2, 0, 0, None, None, 0, V, o.data_ptr(), st))
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("4 calls enqueued in %.2f ms, finished after %.1f ms" % (t_host * 1e3, t_all * 1e3))
    assert t_host < 0.25 * t_all, "the device entry point blocked: %.1f ms of %.1f ms on the host" % (t_host * 1e3, t_all * 1e3)
    for o in outs:
        assert torch.equal(o, alone)


def test_two_host_threads_share_one_device():
    """MFModel.fit(parallel=True) drives every GPU from its own host thread (mf.py: _fit_sharded); all library state that
    is not owned by a handle is per thread (streams, staging and device buffers, scratch arenas, counters).  Two threads
    fitting DIFFERENT mixed ROIs on the same device at the same time must each get what they get alone."""
    import threading
    from microstructure_fingerprinting_amd import engine, synth
    N = 782
    sch, ms, sig_csf, _, rng = _c4_model(N, 4)
    plan = ms.plan_for(sch)
    jobs = []
    for seed, V in ((11, 3000), (12, 2200)):
        r = np.random.default_rng(seed)
        peaks, Y = _c2_like_voxels(r, plan, N, V)
        K = np.where(r.random(V) < 0.3, 1, 2).astype(np.int32)
        csf = r.random(V) < 0.5
        Y = Y + 500.0 * 0.2 * csf[:, None] * sig_csf
        jobs.append((plan, Y, K, csf, None, peaks, 2, True, False, sig_csf))
    alone = [engine.fit_batch(*j) for j in jobs]
    got = [[None] * 3, [None] * 3]
    errs = []

    def work(t):
        try:
            for it in range(3):
                got[t][it] = engine.fit_batch(*jobs[t])
        except Exception as e:   # noqa: BLE001 (reported below)
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for t in range(2):
        for it in range(3):
            assert np.array_equal(got[t][it], alone[t]), "thread %d, call %d" % (t, it + 1)


def _c2_like_voxels(rng, plan, N, V):
    from microstructure_fingerprinting_amd import synth
    p1, p2 = synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)
    q = V // 10
    p2[:q] = _second_peak(rng, p1[:q], 3.0)
    p2[q:q + q // 2] = p1[q:q + q // 2]
    atoms = rng.integers(0, N, (V, 2))
    nu = rng.dirichlet(np.ones(2), V)
    nu[2 * q:3 * q] = [1.0, 0.0]
    nu[3 * q:3 * q + q // 2] = [0.0, 1.0]
    Y = 500.0 * (nu[:, :1] * _rotate_cols(plan, p1, atoms[:, 0]) + nu[:, 1:] * _rotate_cols(plan, p2, atoms[:, 1]))
    noise = rng.normal(0, 500.0 / 30.0, Y.shape)
    noise[4 * q:4 * q + q // 2] = 0.0
    return np.concatenate([p1, p2], axis=1), Y + noise


@pytest.mark.parametrize("N,shells,dirs", [(782, [1000, 2000, 3000], [66, 66, 66]), (100, [1000, 2000, 3000], [66, 66, 66]),
                                            (33, [1000, 2000], [70, 80]), (416, [1000, 2000, 3000, 4000], [60, 60, 60, 60])])
def test_k2_wide_kernel_vs_default_and_oracle(N, shells, dirs):
    """The wide screening kernel (fit_k2w.hip: one wave per SIMD, two row tiles per wave) forced onto protocols of 129..256
    measurements: outputs bit-identical to the default two-waves-per-SIMD screening kernel on 6 000 voxels (close
    crossings, identical peaks, one fascicle, noise-free included), and equal to the oracle on a sample; ragged
    dictionary sizes (one tile + tail, partial rounds)."""
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(900 + N)
    sch = synth.make_scheme(rng, 2, shells, dirs)
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    V = 6000
    peaks, Y = _c2_like_voxels(rng, plan, N, V)
    lib = L.lib()
    K = np.full(V, 2)
    res = []
    try:
        for wide in (1, 0):
            lib.mfx_debug_set_k2_wide(wide)
            res.append(engine.fit_batch(plan, Y, K, None, None, peaks, 2, False, False))
    finally:
        lib.mfx_debug_set_k2_wide(-1)
    bad = np.where(np.any(res[0] != res[1], axis=1))[0]
    assert bad.size == 0, "wide kernel differs from the default kernel in voxels %s" % bad[:10]
    ns = 48
    sel = np.arange(0, V, V // ns)[:ns]
    z = np.zeros(ns, bool)
    ref = orc.fit_batch(_tables(ms), sch, Y[sel], K[:ns], z, z, peaks[sel], 2, False, False, None, None, 0, nthreads=NTHREADS)
    _assert_rows(res[0][sel], ref, 2, "wide kernel N=%d" % N)


@pytest.mark.parametrize("N,dirs,bracket", [(782, [100, 100, 100], False), (782, [137, 137, 137, 137], False),
                                             (100, [137, 137, 137, 137], True), (65, [90, 90, 90], True)])
def test_k2_long_protocols_wide_kernel_vs_fp64_kernel_and_oracle(N, dirs, bracket):
    """Protocols of 257..560 measurements (the reference's HCP-MGH fixture has 552 rows) run on the wide screening kernel
    (one row tile per wave, KS = 24 / 35): bit-identical to the FP64 kernel on 3 000 voxels, equal to the oracle on a
    sample; exact-G and G-bracketed rows."""
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(700 + N + len(dirs))
    shells = [1000, 2000, 3000, 5000][:len(dirs)]
    sch_ms = synth.make_scheme(rng, 2, shells, dirs)
    dic = synth.make_dictionary(rng, sch_ms, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch_ms, Z)
    sch = sch_ms.copy()
    if bracket:
        nz = np.where(sch[:, 3] > 0)[0]
        Gs = ms["Gms_un"]
        sch[nz[::3], 3] = rng.choice([0.4 * Gs[1] + 0.6 * Gs[2], 0.5 * (Gs[-2] + Gs[-1])], size=nz[::3].size)
    plan = ms.plan_for(sch)
    V = 3000
    peaks, Y = _c2_like_voxels(rng, plan, N, V)
    lib = L.lib()
    K = np.full(V, 2)
    res = []
    try:
        for screen in (1, 0):
            lib.mfx_debug_set_k2_screen(screen)
            res.append(engine.fit_batch(plan, Y, K, None, None, peaks, 2, False, False))
    finally:
        lib.mfx_debug_set_k2_screen(1)
    bad = np.where(np.any(res[0] != res[1], axis=1))[0]
    assert bad.size == 0, "wide kernel differs from the FP64 kernel in voxels %s" % bad[:10]
    ns = 32
    sel = np.arange(0, V, V // ns)[:ns]
    z = np.zeros(ns, bool)
    ref = orc.fit_batch(_tables(ms), sch, Y[sel], K[:ns], z, z, peaks[sel], 2, False, False, None, None, 0, nthreads=NTHREADS)
    _assert_rows(res[0][sel], ref, 2, "long protocol M=%d" % sch.shape[0])


def test_three_fascicles_generic_class_vs_oracle():
    """BASELINE config 5's class (three fascicles; opt-in on the C ABI, MFModel.fit stops at two like the reference):
    sub-dictionaries [96, 96, 96] and [40, 40, 40, 1] against the oracle's solve_exhaustive_posweights_3 / _4up on
    rotated dictionaries, incl. a voxel with two fascicles only."""
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    for N, c in ((96, 0), (40, 1)):
        rng = np.random.default_rng(40 + N)
        sch = synth.make_scheme(rng, 2, [1000, 2000, 3000], [30, 30, 30])
        dic = synth.make_dictionary(rng, sch, N)
        ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
        plan = ms.plan_for(sch)
        T = _tables(ms)
        V, M = 6, sch.shape[0]
        peaks = np.concatenate([synth.unit_vectors(rng, V) for _ in range(3)], axis=1)
        atoms = rng.integers(0, N, (V, 3))
        nu = rng.dirichlet(np.ones(3 + c), V)
        nu[1, 2] = 0.0
        gam = mfu.get_gyromagnetic_ratio('H')
        b = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
        sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3.0e-9)
        Y = rng.normal(0, 500.0 / 30.0, (V, M))
        for k in range(3):
            Y += 500.0 * nu[:, k:k + 1] * _rotate_cols(plan, peaks[:, 3 * k:3 * k + 3], atoms[:, k])
        if c:
            Y += 500.0 * nu[:, 3:4] * sig_csf
        got = engine.fit_batch(plan, Y, np.full(V, 3), np.full(V, bool(c)), None, peaks, 3, bool(c), False, sig_csf if c else None)
        for v in range(V):
            A = np.concatenate([orc.interp(sch, peaks[v, 3 * k:3 * k + 3], T) for k in range(3)] + ([sig_csf[:, None]] if c else []), axis=1)
            w, sub, tot, mo, yrec = orc.solve_exhaustive_posweights(np.ascontiguousarray(A), Y[v], np.array([N, N, N] + ([1] if c else [])))
            act = w[:3] > 1e-9
            assert np.array_equal(got[v, 4:7][act], sub[:3][act].astype(float)), (v, got[v], sub, w)
            assert np.allclose(got[v, 0], w.sum(), rtol=1e-6) and np.allclose(got[v, 1:4], w[:3] / w.sum(), rtol=1e-6, atol=1e-9)
            assert np.isclose(got[v, -2], mo / M, rtol=1e-6)


def test_three_fascicles_in_a_mixed_roi_host_path():
    """A ROI with one, two and three fascicles per voxel through ONE mfx_fit_batch call (class binning, voxel lists, the
    batched three-fascicle path on a list of scattered voxels, 70 three-fascicle voxels = three batches): every row equals
    the row of the same voxel fitted in a call of its own class."""
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    rng = np.random.default_rng(4242)
    sch = synth.make_scheme(rng, 2, [1000, 2000, 3000], [30, 30, 30])
    N = 96
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    V, M = 210, sch.shape[0]
    Kv = np.tile([1, 2, 3], V // 3)
    rng.shuffle(Kv)
    peaks = np.concatenate([synth.unit_vectors(rng, V) for _ in range(3)], axis=1)
    atoms = rng.integers(0, N, (V, 3))
    nu = rng.dirichlet(np.ones(3), V) * (np.arange(3)[None, :] < Kv[:, None])
    Y = rng.normal(0, 500.0 / 30.0, (V, M))
    for k in range(3):
        Y += 500.0 * nu[:, k:k + 1] * _rotate_cols(plan, peaks[:, 3 * k:3 * k + 3], atoms[:, k])
    got = engine.fit_batch(plan, Y, Kv, None, None, peaks, 3, False, False)
    for K in (1, 2, 3):
        sel = np.flatnonzero(Kv == K)
        alone = engine.fit_batch(plan, Y[sel], np.full(sel.size, K), None, None, peaks[sel], 3, False, False)
        assert np.array_equal(got[sel], alone), K
    assert np.all(got[Kv == 1][:, 2:4] == 0) and np.all(got[Kv == 3][:, 0] > 0)


def test_fit_over_rotate_atom_plan_vs_oracle():
    """The voxel loop driven by rotate_atom tables (an explicit row plan: the dictionary is sampled on the subject's own
    protocol, mf_utils.py:1205-1437 - the reference's test_hcp_dict shape of use) on the HCP-MGH fixture (552 rows x 782
    atoms): two-fascicle voxels against orc.rotate_atom + the oracle's solve_exhaustive_posweights_2, and a
    three-fascicle voxel on an atom subset against solve_exhaustive_posweights_3."""
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    d = np.load(os.path.join(G, "real_hcp.npz"))
    sch, sig, S0, DIFF = np.ascontiguousarray(d["sch_mat"]), np.ascontiguousarray(d["dictionary"]), np.ascontiguousarray(d["S0"]), float(d["WM_DIFF"])
    rng = np.random.default_rng(17)
    N = sig.shape[1]
    RT = mfu.RotateAtomTables(sig, sch, Z, DIFF, S0, warnings=False)
    V = 12
    peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
    peaks[:, :3] *= 1.0 + 1e-4 * rng.standard_normal((V, 1))        # rotate_atom normalises: any length is legal
    atoms = rng.integers(0, N, (V, 2)); atoms[0] = 86
    nu = rng.dirichlet(np.ones(2), V)
    D = [RT.rotate(peaks[:, 3 * k:3 * k + 3]) for k in range(2)]       # [V, M, N] each
    Y = rng.normal(0, 500.0 / 30.0, (V, sch.shape[0]))
    for k in range(2):
        Y += 500.0 * nu[:, k:k + 1] * np.take_along_axis(D[k], atoms[:, k][:, None, None], axis=2)[:, :, 0]
    got = engine.fit_batch(RT.plan, Y, np.full(V, 2), None, None, peaks, 2, False, False)
    for v in range(V):
        A = np.concatenate([orc.rotate_atom(sig, sch, Z, peaks[v, 3 * k:3 * k + 3], DIFF, S0) for k in range(2)], axis=1)
        w, sub, tot, mo, yrec = orc.solve_exhaustive_posweights(np.ascontiguousarray(A), Y[v], np.array([N, N]))
        assert np.array_equal(got[v, 3:5], sub.astype(float)), (v, got[v], sub)
        assert np.allclose(got[v, 1:3], w / w.sum(), rtol=1e-9) and np.isclose(got[v, 0], w.sum(), rtol=1e-9)
    # three fascicles on 60 atoms of the same dictionary
    sel = np.arange(0, N, 13)[:60]
    RT3 = mfu.RotateAtomTables(np.ascontiguousarray(sig[:, sel]), sch, Z, DIFF, np.ascontiguousarray(S0[:, sel]), warnings=False)
    pk3 = np.concatenate([synth.unit_vectors(rng, 2) for _ in range(3)], axis=1)
    at3 = rng.integers(0, sel.size, (2, 3))
    nu3 = rng.dirichlet(np.ones(3), 2)
    Y3 = rng.normal(0, 500.0 / 30.0, (2, sch.shape[0]))
    for k in range(3):
        Dk = RT3.rotate(pk3[:, 3 * k:3 * k + 3])
        Y3 += 500.0 * nu3[:, k:k + 1] * np.take_along_axis(Dk, at3[:, k][:, None, None], axis=2)[:, :, 0]
    g3 = engine.fit_batch(RT3.plan, Y3, np.full(2, 3), None, None, pk3, 3, False, False)
    for v in range(2):
        A = np.concatenate([orc.rotate_atom(np.ascontiguousarray(sig[:, sel]), sch, Z, pk3[v, 3 * k:3 * k + 3], DIFF, np.ascontiguousarray(S0[:, sel])) for k in range(3)], axis=1)
        w, sub, tot, mo, yrec = orc.solve_exhaustive_posweights(np.ascontiguousarray(A), Y3[v], np.array([sel.size] * 3))
        assert np.array_equal(g3[v, 4:7], sub.astype(float)), (v, g3[v], sub)
        assert np.allclose(g3[v, 1:4], w / w.sum(), rtol=1e-6, atol=1e-9)


def test_three_dictionaries_fast_path_vs_oracle():
    """solve_exhaustive_posweights(A, y, [N, N, N]) on the fast three-dictionary path (solve_k3.hip: Gram on FP64 MFMA,
    relaxed-bound screen of all N^3 triples, candidate list, exact finalize) against the oracle's
    solve_exhaustive_posweights_3: N = 128 (2.1e6 triples) incl. a signal made of two atoms only and of one atom only (every
    triple sharing the active atoms ties: the first hit in the reference's i3 -> i1 -> i2 order must win), noise-free data."""
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from microstructure_fingerprinting_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(33)
    sch = synth.make_scheme(rng, 2, [1000, 2000, 3000], [30, 30, 30])
    N, M = 128, sch.shape[0]
    dic = synth.make_dictionary(rng, sch, N)
    T = orc.init_tables(dic, sch, Z)
    for case in range(6):
        dirs = synth.unit_vectors(rng, 3)
        A = np.ascontiguousarray(np.concatenate([orc.interp(sch, d, T) for d in dirs], axis=1))
        at = rng.integers(0, N, 3)
        nu = rng.dirichlet(np.ones(3))
        if case == 1: nu = np.array([0.6, 0.4, 0.0])
        if case == 2: nu = np.array([0.0, 1.0, 0.0])
        y = 500.0 * sum(nu[k] * A[:, k * N + at[k]] for k in range(3))
        if case != 3:
            y = y + rng.normal(0, 500.0 / 30.0, M)
        sizes = np.array([N, N, N])
        w, sub, tot, mo, yrec = mfu.solve_exhaustive_posweights(A, y, sizes)
        wr, subr, totr, mor, yrecr = orc.solve_exhaustive_posweights(A, y, sizes)
        assert np.array_equal(sub, subr), (case, sub, subr, w, wr)
        assert np.allclose(w, wr, rtol=1e-9, atol=1e-9) and np.isclose(mo, mor, rtol=1e-9, atol=1e-9 * float(y @ y))
        assert np.allclose(yrec, yrecr, rtol=1e-9, atol=1e-9)


def test_three_fascicles_oracle_refereed_at_n400():
    """Config 5's class where the triple screen is under pressure: three fascicles, 400 atoms x 300
    measurements (6.4e7 triples per voxel, ~100 s per voxel for the oracle's solve_exhaustive_posweights_3: one voxel per
    host thread), through mfx_fit_batch with maxfasc = 3.  Voxels: generic mixtures at SNR 30, signals made of two atoms
    and of one atom only (every triple sharing the active atoms nearly ties: the reference's first hit in its
    i3 -> i1 -> i2 order must win), two peaks 3 degrees apart, identical peaks, noise-free data.  Atom indices must equal
    the oracle's, weights and objective to 1e-9."""
    from concurrent.futures import ThreadPoolExecutor
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    N = int(os.environ.get("MFX_K3_REFEREE_N", "400"))
    rng = np.random.default_rng(77)
    sch = synth.make_scheme(rng, 1, [1000, 2000, 3000, 4000], [75, 75, 75, 74])
    M = sch.shape[0]
    assert M == 300
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    T = _tables(ms)
    kinds = ["generic"] * 5 + ["two_atoms", "two_atoms", "one_atom", "cross3", "cross3", "identical", "noise_free"]
    V = len(kinds)
    p = [synth.unit_vectors(rng, V) for _ in range(3)]
    nu = rng.dirichlet(np.ones(3) * 2, V)
    for v, kd in enumerate(kinds):
        if kd == "two_atoms": nu[v] = [0.55, 0.45, 0.0] if v % 2 else [0.0, 0.3, 0.7]
        if kd == "one_atom": nu[v] = [0.0, 1.0, 0.0]
        if kd == "cross3": p[1][v] = _second_peak(rng, p[0][v:v + 1], 3.0)[0]
        if kd == "identical": p[2][v] = p[0][v]
    peaks = np.concatenate(p, axis=1)
    atoms = rng.integers(0, N, (V, 3))
    Y = np.zeros((V, M))
    for k in range(3):
        Y += 500.0 * nu[:, k:k + 1] * _rotate_cols(plan, peaks[:, 3 * k:3 * k + 3], atoms[:, k])
    noisy = np.array([kd != "noise_free" for kd in kinds])
    Y[noisy] += rng.normal(0, 500.0 / 30.0, (int(noisy.sum()), M))
    got = engine.fit_batch(plan, Y, np.full(V, 3), None, None, peaks, 3, False, False)
    sizes = np.array([N, N, N])

    def referee(v):
        A = np.ascontiguousarray(np.concatenate([orc.interp(sch, peaks[v, 3 * k:3 * k + 3], T) for k in range(3)], axis=1))
        return orc.solve_exhaustive_posweights(A, Y[v], sizes)
    with ThreadPoolExecutor(max_workers=min(V, NTHREADS)) as ex:     # the C oracle runs outside the GIL
        refs = list(ex.map(referee, range(V)))
    for v, (w, sub, tot, mo, yrec) in enumerate(refs):
        assert np.array_equal(got[v, 4:7], sub.astype(float)), (v, kinds[v], got[v], sub, w)
        ws = w.sum()
        assert np.isclose(got[v, 0], ws, rtol=1e-9) and np.allclose(got[v, 1:4], w / ws if abs(ws) > 0 else w, rtol=1e-9, atol=1e-12)
        assert np.isclose(got[v, -2], mo / M, rtol=1e-9, atol=1e-12 * float(Y[v] @ Y[v]) / M), (v, kinds[v], got[v, -2] * M, mo)


def test_three_fascicles_batched_path_overflow_fallback():
    """The batched three-fascicle path (fit_k3.hip) with its candidate list cut to 4 entries: every voxel overflows and is
    redone by the voxel-by-voxel path, enqueued behind the batch and gated on the device by the overflow flag.  Results
    must equal the un-cut run's (which the oracle referees in the tests above), incl. a one-atom signal whose N^2 ties fill
    any list."""
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    rng = np.random.default_rng(78)
    sch = synth.make_scheme(rng, 1, [1000, 2000, 3000], [40, 40, 39])
    N, M, V = 96, sch.shape[0], 11
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    peaks = np.concatenate([synth.unit_vectors(rng, V) for _ in range(3)], axis=1)
    atoms = rng.integers(0, N, (V, 3))
    nu = rng.dirichlet(np.ones(3) * 2, V)
    nu[3] = [0.0, 1.0, 0.0]
    Y = rng.normal(0, 500.0 / 30.0, (V, M))
    for k in range(3):
        Y += 500.0 * nu[:, k:k + 1] * _rotate_cols(plan, peaks[:, 3 * k:3 * k + 3], atoms[:, k])
    lib = L.lib()
    full = engine.fit_batch(plan, Y, np.full(V, 3), None, None, peaks, 3, False, False)
    try:
        lib.mfx_debug_set_k3_cap(4)
        cut = engine.fit_batch(plan, Y, np.full(V, 3), None, None, peaks, 3, False, False)
    finally:
        lib.mfx_debug_set_k3_cap(0)
    assert np.array_equal(full[:, 4:7], cut[:, 4:7]), (full[:, 4:7], cut[:, 4:7])
    assert np.allclose(full, cut, rtol=1e-12, atol=1e-12)


def test_c5_full_size_properties():
    """BASELINE config 5 at its full size - three fascicles, 1500 atoms x 300 measurements, 3.4e9 triples per voxel -
    through mfx_fit_batch_dev (two voxels in flight).  The oracle would need hours per voxel here, so the checks are
    size-independent properties: (i) a noise-free voxel made of three atoms returns exactly those atoms, its weights and a
    zero residual; (ii) for noisy voxels the returned triple's objective, recomputed on the host from the oracle's
    rotation and a three-column NNLS, equals the reported MSE, and no triple obtained by exchanging one atom of the
    planted or of the returned solution for a neighbour does better (local optimality of an exhaustive search)."""
    import torch
    from microstructure_fingerprinting_amd import engine
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from microstructure_fingerprinting_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(55)
    sch = synth.make_scheme(rng, 1, [1000, 2000, 3000, 4000], [75, 75, 75, 74])
    N, M = 1500, sch.shape[0]
    assert M == 300
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    T = orc.init_tables(dic, sch, Z)
    V = 6
    peaks = np.concatenate([synth.unit_vectors(rng, V) for _ in range(3)], axis=1)
    atoms = rng.integers(0, N, (V, 3))
    nu = rng.dirichlet(np.ones(3) * 3, V)
    D = [[orc.interp(sch, peaks[v, 3 * k:3 * k + 3], T) for k in range(3)] for v in range(V)]     # [V][3] of [M, N]
    Y = np.stack([500.0 * sum(nu[v, k] * D[v][k][:, atoms[v, k]] for k in range(3)) for v in range(V)])
    Y[2:] += rng.normal(0, 500.0 / 30.0, (V - 2, M))
    dev = torch.device("cuda", 0)
    out = engine.fit_batch_dev(plan, torch.from_numpy(Y).to(dev), torch.from_numpy(peaks).to(dev), 3).cpu().numpy()

    def objective(v, ids):
        A = np.stack([D[v][k][:, ids[k]] for k in range(3)], axis=1)
        w, _ = orc.nnls(A, Y[v])
        r = Y[v] - A @ w
        return float(r @ r), w
    for v in range(V):
        M0, fr, ids, mse = out[v, 0], out[v, 1:4], out[v, 4:7].astype(int), out[v, 7]
        res, w = objective(v, ids)
        assert np.isclose(mse * M, res, rtol=1e-7, atol=1e-9 * float(Y[v] @ Y[v])), (v, mse * M, res)
        assert np.allclose(M0 * fr, w, rtol=1e-6, atol=1e-7 * M0)
        if v < 2:     # noise-free: the planted triple, exactly
            assert np.array_equal(ids, atoms[v]), (v, ids, atoms[v])
            assert np.allclose(M0 * fr, 500.0 * nu[v], rtol=1e-8) and res <= 1e-12 * float(Y[v] @ Y[v])
        else:         # nobody in the neighbourhood of the planted or the returned triple does better
            assert res <= objective(v, atoms[v])[0] * (1 + 1e-12)
            for base in (atoms[v], ids):
                for k in range(3):
                    for dlt in (-2, -1, 1, 2):
                        alt = np.array(base).copy()
                        alt[k] = (alt[k] + dlt) % N
                        assert res <= objective(v, alt)[0] * (1 + 1e-12), (v, base, k, dlt)


def test_c5_full_size_screen_vs_unscreened_scan():
    """BASELINE config 5 at its full size: the batched path (triple test on the matrix pipe, second test, candidate list)
    against a scan of ALL 3.4e9 triples of every voxel with no screen at all (mfx_debug_set_k3_screen(0): one thread per
    triple, every triple scored from the FP64 Gram, the same exact stage) - whether any screening stage pruned the global
    optimum is exactly what this compares.  40 voxels: bench-style mixtures at SNR 30, four at SNR 10, two noise-free, four
    whose signal holds only TWO fascicles (flat optimum: the third atom is inactive), two with ONE."""
    import torch
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    import bench
    rng = np.random.default_rng(77)
    sch = synth.make_scheme(rng, 1, [1000, 2000, 3000, 4000], [75, 75, 75, 74])
    N, M = 1500, sch.shape[0]
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    dev = torch.device("cuda", 0)
    V = 40
    _, dpk, dY = bench.synth_voxels(plan, V, N, M, dev, 11, K=3)
    Y = dY.cpu().numpy()
    clean = bench.synth_voxels(plan, V, N, M, dev, 11, K=3, snr=1e12)[2].cpu().numpy()
    Y[28:32] = clean[28:32] + 3.0 * (Y[28:32] - clean[28:32])    # SNR 10
    Y[32:34] = clean[32:34]                                      # noise-free

    def atom(v, k, n):
        return engine.rotate_columns_dev(plan, dpk[v:v + 1, 3 * k:3 * k + 3].contiguous(),
                                         torch.tensor([n], dtype=torch.int32, device=dev)).cpu().numpy()[0]
    for v in range(34, 38):                                      # two fascicles only (flat optimum: the third atom is inactive)
        Y[v] = 300.0 * atom(v, 0, 17 + v) + 200.0 * atom(v, (v % 2) + 1, 1203 - v) + rng.normal(0, 500.0 / 30.0, M)
    for v in range(38, 40):                                      # one fascicle only
        Y[v] = 500.0 * atom(v, v % 3, 700 + v) + rng.normal(0, 500.0 / 30.0, M)
    dY = torch.from_numpy(Y).to(dev)
    got = engine.fit_batch_dev(plan, dY, dpk, 3).cpu().numpy()
    lib = L.lib()
    lib.mfx_debug_set_k3_screen(0)
    try:
        ref = engine.fit_batch_dev(plan, dY, dpk, 3).cpu().numpy()
    finally:
        lib.mfx_debug_set_k3_screen(1)
    assert np.array_equal(got[:, 4:7], ref[:, 4:7]), (got[:, 4:7], ref[:, 4:7])
    assert np.array_equal(got, ref)
    assert np.all(got[:, 0] > 0)


@pytest.mark.parametrize("N,dirs", [(64, [15, 15, 15, 15]), (100, [19, 19, 19, 19]), (257, [8, 8, 8, 8]), (130, [75, 75, 75, 74]), (513, [13, 12, 12, 12])])
def test_three_fascicles_ragged_shapes_screen_vs_unscreened_scan(N, dirs):
    """The batched three-fascicle path on shapes that are not multiples of its tiles (256 x 128 pairs per workgroup, blocks
    of 4 third atoms, Gram tiles of 128 x 64 atoms and 16 measurement rows per group): dictionaries of 64 .. 513 atoms,
    33 .. 300 measurements, 24 voxels each (mixtures, noise-free, one and two fascicles only) - against the unscreened scan
    of all triples (mfx_debug_set_k3_screen(0)); rows must be identical."""
    import torch
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    import bench
    rng = np.random.default_rng(N)
    sch = synth.make_scheme(rng, 1, [1000, 2000, 3000, 4000], dirs)
    M = sch.shape[0]
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    dev = torch.device("cuda", 0)
    V = 24
    _, dpk, dY = bench.synth_voxels(plan, V, N, M, dev, 5, K=3)
    Y = dY.cpu().numpy()
    clean = bench.synth_voxels(plan, V, N, M, dev, 5, K=3, snr=1e12)[2].cpu().numpy()
    Y[16:18] = clean[16:18]

    def atom(v, k, n):
        return engine.rotate_columns_dev(plan, dpk[v:v + 1, 3 * k:3 * k + 3].contiguous(),
                                         torch.tensor([n], dtype=torch.int32, device=dev)).cpu().numpy()[0]
    for v in range(18, 22):
        Y[v] = 300.0 * atom(v, 0, (7 * v) % N) + 200.0 * atom(v, (v % 2) + 1, N - 1 - v) + rng.normal(0, 500.0 / 30.0, M)
    for v in range(22, 24):
        Y[v] = 500.0 * atom(v, v % 3, (11 * v) % N) + rng.normal(0, 500.0 / 30.0, M)
    dY = torch.from_numpy(Y).to(dev)
    got = engine.fit_batch_dev(plan, dY, dpk, 3).cpu().numpy()
    lib = L.lib()
    lib.mfx_debug_set_k3_screen(0)
    try:
        ref = engine.fit_batch_dev(plan, dY, dpk, 3).cpu().numpy()
    finally:
        lib.mfx_debug_set_k3_screen(1)
    assert np.array_equal(got[:, 4:7], ref[:, 4:7]), (got[:, 4:7], ref[:, 4:7])
    assert np.array_equal(got, ref)


def test_k2_csf_screening_pipeline_vs_plain_kernel_and_oracle():
    """[782, 782, 1] (two fascicles + CSF) at BASELINE config 2's size through the screening pipeline (split-FP16 screening
    kernel with the CSF column projected out -> per-voxel short lists -> exact stage in list mode -> plain kernel for the
    handed-back voxels): 6 144 voxels - generic mixtures, weak / absent CSF signal, one fascicle + CSF, pure CSF, 3 degree
    crossings, identical peaks, noise-free - must equal the plain FP64 kernel of the class bit for bit, and the oracle on a
    sample; most voxels must be decided by the pipeline itself and the bound check must never fire."""
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from oracle import oracle as orc
    N, V = 782, int(os.environ.get("MFX_STRESS_CSF_V", "6144"))   # (one-off deep runs: e.g. 61440)
    sch, ms, sig_csf, _, rng = _c4_model(N, 4)
    plan = ms.plan_for(sch)
    M = plan.M
    p1, p2 = synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)
    atoms = rng.integers(0, N, (V, 2))
    nu = rng.dirichlet(np.ones(3), V)              # f0, f1, csf
    kind = np.arange(V) % 12
    nu[kind == 0, 2] *= 0.02                       # weak CSF signal
    nu[kind == 1, 2] = 0                           # none, although the column is offered
    nu[kind == 2, 1] = 0                           # fascicle 0 + CSF
    nu[kind == 3, 0] = 0                           # fascicle 1 + CSF
    nu[kind == 4] = [0, 0, 1]                      # pure CSF
    p2[kind == 5] = _second_peak(rng, p1[kind == 5], 3.0)
    p2[kind == 6] = p1[kind == 6]                  # identical peaks
    nu /= nu.sum(1, keepdims=True)
    Y = 500.0 * (nu[:, :1] * _rotate_cols(plan, p1, atoms[:, 0]) + nu[:, 1:2] * _rotate_cols(plan, p2, atoms[:, 1]) + nu[:, 2:3] * sig_csf)
    noise = rng.normal(0, 500.0 / 30.0, (V, M))
    noise[kind == 7] = 0.0                         # noise-free
    noise[kind == 8] *= 3.0                        # SNR 10
    Y = Y + noise
    peaks = np.concatenate([p1, p2], axis=1)
    one, zero = np.ones(V, bool), np.zeros(V, bool)
    lib = L.lib()
    args = (plan, Y, np.full(V, 2), one, zero, peaks, 2, True, False, sig_csf, None, 0)
    got = engine.fit_batch(*args)
    cn = [lib.mfx_debug_last_counter(q) for q in range(6)]
    aud = [lib.mfx_debug_last_counter(q) for q in (8, 9, 10)]
    # population audit of the [N, N, 1] screening kernel: one pseudo-random pair per screened voxel, d1.d2 - u1 u2 against FP64
    assert aud[2] > 0.5 * V and aud[0] == 0 and aud[1] * 1e-11 < 0.25 * 1.5e-5, aud
    lib.mfx_debug_set_k2x_screen(0)
    try:
        plain = engine.fit_batch(*args)
    finally:
        lib.mfx_debug_set_k2x_screen(1)
    assert np.array_equal(got, plain), "rows differ: %s" % np.flatnonzero(np.any(got != plain, axis=1))[:10]
    print("[782,782,1] screening pipeline: %d of %d voxels handed to the plain kernel, %d listed pairs, %d family items" % (cn[4], V, cn[2], cn[3]))
    assert cn[5] == 0, "bound check fired for %d voxels" % cn[5]
    assert cn[4] <= 0.35 * V, "%d of %d voxels handed back to the plain kernel" % (cn[4], V)
    ns = 384
    ref = orc.fit_batch(_tables(ms), sch, Y[:ns], np.full(ns, 2), one[:ns], zero[:ns], peaks[:ns], 2, True, False, sig_csf, None, 0,
                        nthreads=NTHREADS)
    _assert_rows(got[:ns], ref, 2, "[782,782,1] screening pipeline", rtol=1e-9)


def test_repeated_small_host_calls_are_reproducible():
    """The same 64 two-fascicle + CSF voxels through the host entry point thirty times, the screening pipeline switched on
    and off in between: every call must return the oracle's rows.  (Regression: with scratch memory from HIP's default
    stream-ordered pool - unmapped at every synchronisation, mapped again by the next call - some of these calls came back
    with blocks of about 8 voxels fitted on stale data; the library now owns a pool that keeps its memory, mfx_host.h.)"""
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from oracle import oracle as orc
    N, V = 782, 64
    sch, ms, sig_csf, _, _ = _c4_model(N, 4)
    plan = ms.plan_for(sch)
    M = plan.M
    rng = np.random.default_rng(5)
    p1, p2 = synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)
    atoms = rng.integers(0, N, (V, 2))
    Y = 500.0 * (0.4 * _rotate_cols(plan, p1, atoms[:, 0]) + 0.4 * _rotate_cols(plan, p2, atoms[:, 1]) + 0.2 * sig_csf) + rng.normal(0, 16, (V, M))
    peaks = np.concatenate([p1, p2], axis=1)
    one, zero = np.ones(V, bool), np.zeros(V, bool)
    args = (plan, Y, np.full(V, 2), one, zero, peaks, 2, True, False, sig_csf, None, 0)
    ref = orc.fit_batch(_tables(ms), sch, Y, np.full(V, 2), one, zero, peaks, 2, True, False, sig_csf, None, 0, nthreads=NTHREADS)
    lib = L.lib()
    try:
        for rep in range(3):
            for scr in (1, 0):
                lib.mfx_debug_set_k2x_screen(scr)
                for k in range(5):
                    _assert_rows(engine.fit_batch(*args), ref, 2, "repeat %d, screening %d, call %d" % (rep, scr, k + 1), rtol=1e-9)
    finally:
        lib.mfx_debug_set_k2x_screen(1)


def test_mixed_classes_full_size_vs_oracle():
    """A mixed ROI at BASELINE config 2's size (782 atoms x 200 measurements) through the host entry point: voxels with 0, 1
    or 2 fascicles, with and without the CSF flag, interleaved - every class goes through its own kernel from per-chunk
    voxel lists (the two-fascicle + CSF class through the screening pipeline with a voxel list) and must equal the oracle
    row by row."""
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from oracle import oracle as orc
    N, V = 782, 720
    sch, ms, sig_csf, _, rng = _c4_model(N, 4)
    plan = ms.plan_for(sch)
    M = plan.M
    K = rng.integers(0, 3, V)
    csf = rng.random(V) < 0.5
    csf[K == 0] = True                              # (K = 0 without CSF: nothing to estimate, rows stay zero)
    csf[::17] = False
    p1, p2 = synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)
    atoms = rng.integers(0, N, (V, 2))
    nu = rng.dirichlet(np.ones(3), V)
    nu[K < 2, 1] = 0
    nu[K < 1, 0] = 0
    nu[~csf, 2] = 0
    nu[nu.sum(1) == 0] = [0, 0, 1]
    nu /= nu.sum(1, keepdims=True)
    Y = 500.0 * (nu[:, :1] * _rotate_cols(plan, p1, atoms[:, 0]) + nu[:, 1:2] * _rotate_cols(plan, p2, atoms[:, 1]) + nu[:, 2:3] * sig_csf)
    Y = Y + rng.normal(0, 500.0 / 30.0, (V, M))
    peaks = np.concatenate([p1, p2], axis=1)
    zero = np.zeros(V, bool)
    got = engine.fit_batch(plan, Y, K, csf, zero, peaks, 2, True, False, sig_csf, None, 0)
    cn = [L.lib().mfx_debug_last_counter(q) for q in range(6)]
    ref = orc.fit_batch(_tables(ms), sch, Y, K, csf, zero, peaks, 2, True, False, sig_csf, None, 0, nthreads=NTHREADS)
    _assert_rows(got, ref, 2, "mixed ROI at 782 x 200", rtol=1e-9)
    assert cn[5] == 0
    # the same rows through the reference's ROI gather (rows argument): the volume holds every voxel twice
    vol = np.concatenate([Y[::-1], Y], axis=0)
    rows = (V + np.arange(V)).astype(np.int64)
    got2 = engine.fit_batch(plan, vol, K, csf, zero, peaks, 2, True, False, sig_csf, None, 0, rows=rows)
    assert np.array_equal(got, got2)


@pytest.mark.parametrize("dirs", [[100, 100, 100], [137, 137, 137, 139]])
def test_k2_csf_pipeline_long_protocols_vs_plain_kernel_and_oracle(dirs):
    """[782, 782, 1] with 302 / 552 measurements (HCP-MGH length): the wide screening kernel in its XC form + list-mode exact
    stage against the plain FP64 kernel of the class (bit-identical, incl. pure-CSF and one-fascicle voxels) and the oracle."""
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    rng = np.random.default_rng(31)
    sch = synth.make_scheme(rng, 2, [1000, 2000, 3000, 5000][:len(dirs)], dirs)
    N, V = 782, 1536
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    M = plan.M
    gam = mfu.get_gyromagnetic_ratio('H')
    b = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3.0e-9)
    p1, p2 = synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)
    atoms = rng.integers(0, N, (V, 2))
    nu = rng.dirichlet(np.ones(3), V)
    kind = np.arange(V) % 8
    nu[kind == 1, 2] = 0
    nu[kind == 2, 1] = 0
    nu[kind == 3] = [0, 0, 1]
    p2[kind == 4] = _second_peak(rng, p1[kind == 4], 3.0)
    nu /= nu.sum(1, keepdims=True)
    Y = 500.0 * (nu[:, :1] * _rotate_cols(plan, p1, atoms[:, 0]) + nu[:, 1:2] * _rotate_cols(plan, p2, atoms[:, 1]) + nu[:, 2:3] * sig_csf)
    noise = rng.normal(0, 500.0 / 30.0, (V, M))
    noise[kind == 5] = 0.0
    Y = Y + noise
    peaks = np.concatenate([p1, p2], axis=1)
    one, zero = np.ones(V, bool), np.zeros(V, bool)
    lib = L.lib()
    args = (plan, Y, np.full(V, 2), one, zero, peaks, 2, True, False, sig_csf, None, 0)
    got = engine.fit_batch(*args)
    cn = [lib.mfx_debug_last_counter(q) for q in range(6)]
    lib.mfx_debug_set_k2x_screen(0)
    try:
        plain = engine.fit_batch(*args)
    finally:
        lib.mfx_debug_set_k2x_screen(1)
    assert np.array_equal(got, plain), "rows differ: %s" % np.flatnonzero(np.any(got != plain, axis=1))[:10]
    assert cn[5] == 0 and cn[4] <= 0.4 * V
    ns = 48
    ref = orc.fit_batch(_tables(ms), sch, Y[:ns], np.full(ns, 2), one[:ns], zero[:ns], peaks[:ns], 2, True, False, sig_csf, None, 0,
                        nthreads=NTHREADS)
    _assert_rows(got[:ns], ref, 2, "[782,782,1] screening pipeline, %d rows" % M, rtol=1e-9)
