#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference). Nothing here is
imported by the product or shipped as code to the GPU box: the outputs are
small ``.npz`` data files (inputs + the reference's outputs), committed.

How the reference is executed: ``numba`` is not installed
in this image, and ``mf_utils.py`` applies ``@nba.jit(...)`` at import time, so
the package cannot be imported as-is (ordinary ImportError/TripWireError, not a
permission denial).  We put a *pass-through* ``numba`` module (decorators return
the undecorated function) in a temporary directory in front of ``sys.path``; the
reference's own, unmodified Python source then runs as plain CPython with
IEEE-754 double arithmetic in source order, which is what Numba's ``nopython``
mode without ``fastmath`` computes as well.

Usage:  python tests/golden/gen_golden.py [--only NAME]
"""
import argparse
import os
import sys
import tempfile
import textwrap

import numpy as np

REF = "/root/reference"
FIX = os.path.join(REF, "tests", "integration", "fixtures")
OUT = os.path.dirname(os.path.abspath(__file__))

NUMBA_STUB = textwrap.dedent('''
    """pass-through stand-in: decorators are the identity."""
    class _T:
        def __getitem__(self, k): return self
        def __call__(self, *a, **k): return self
    class _Types:
        def Tuple(self, *a, **k): return _T()
        def UniTuple(self, *a, **k): return _T()
    types = _Types()
    float64 = int32 = int64 = int8 = float32 = _T()
    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not isinstance(a[0], _T) and not k:
            return a[0]
        return lambda f: f
    njit = jit
''')


def import_reference():
    d = tempfile.mkdtemp(prefix="numba_passthrough_")
    with open(os.path.join(d, "numba.py"), "w") as f:
        f.write(NUMBA_STUB)
    sys.path.insert(0, d)
    sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    import microstructure_fingerprinting as mfpkg  # noqa
    from microstructure_fingerprinting import mf_utils as mfu
    from microstructure_fingerprinting import mf as mfmod
    return mfu, mfmod


# --------------------------------------------------------------------------
# shared synthetic generators (kept in sync with the product's synth module by
# storing the generated arrays themselves in the fixtures, not the recipe)
# --------------------------------------------------------------------------
GAM = 2 * np.pi * 42.577480e6


def synth_scheme(rng, n_b0, shells_b, dirs_per_shell, Delta=43.1e-3, delta=10.6e-3, TE=92e-3):
    rows = []
    for _ in range(n_b0):
        rows.append([0, 0, 0, 0.0, Delta, delta, TE])
    for b, nd in zip(shells_b, dirs_per_shell):
        G = np.sqrt(b * 1e6 / (Delta - delta / 3)) / (GAM * delta)
        g = rng.standard_normal((nd, 3))
        g /= np.linalg.norm(g, axis=1, keepdims=True)
        for i in range(nd):
            rows.append([g[i, 0], g[i, 1], g[i, 2], G, Delta, delta, TE])
    return np.array(rows)


def synth_dictionary(rng, sch, N):
    """Smooth positive single-fascicle signals along z for N atoms on scheme sch."""
    G, Dl, dl = sch[:, 3], sch[:, 4], sch[:, 5]
    b = (GAM * G * dl) ** 2 * (Dl - dl / 3)
    u = np.abs(sch[:, 2])
    f = rng.uniform(0.3, 0.9, N)
    dpar = rng.uniform(1.5e-9, 2.5e-9, N)
    dperp = rng.uniform(0.1e-9, 0.8e-9, N)
    Diso = rng.uniform(0.5e-9, 1.5e-9, N)
    s0 = rng.uniform(0.5, 1.0, N)
    u2 = (u ** 2)[:, None]
    bb = b[:, None]
    sig = s0 * (f * np.exp(-bb * dpar * u2) * np.exp(-bb * dperp * (1 - u2))
                + (1 - f) * np.exp(-bb * Diso))
    return sig


def unit(rng, n):
    v = rng.standard_normal((n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


# --------------------------------------------------------------------------
def gen_solver(mfu):
    """Random (A, y, dicsizes) problems for _1/_2/_3/_4up + the reference's
    boundary tables (test_exhaustive_fingerprinting.py:38-89)."""
    rng = np.random.default_rng(20261004)
    out = {}
    cases = [
        ("k1_a", 30, [17]), ("k1_b", 12, [40]),
        ("k2_a", 24, [13, 9]), ("k2_b", 40, [33, 35]), ("k2_c", 7, [5, 6]),
        ("k3_a", 20, [9, 8, 1]), ("k3_b", 25, [7, 6, 5]), ("k3_c", 30, [12, 12, 3]),
        ("k4_a", 20, [6, 5, 1, 3]), ("k4_b", 16, [4, 4, 2, 2]), ("k5_a", 14, [3, 3, 1, 2, 2]),
    ]
    names = []
    for name, M, sizes in cases:
        for variant in range(3):
            nm = "%s_v%d" % (name, variant)
            A = np.abs(rng.standard_normal((M, sum(sizes)))) if variant != 2 else rng.standard_normal((M, sum(sizes)))
            st = np.concatenate([[0], np.cumsum(sizes)[:-1]])
            gt = st + np.array([rng.integers(0, s) for s in sizes])
            w = rng.uniform(0.1, 1.0, len(sizes))
            if variant == 1:  # some weights zero -> exercises single-active branches
                w[rng.integers(0, len(sizes))] = 0.0
            y = A[:, gt] @ w + 0.05 * rng.standard_normal(M)
            if variant == 2:
                y = rng.standard_normal(M)  # arbitrary sign pattern
            ds = np.array(sizes)
            (wn, isub, itot, mo, yr) = mfu.solve_exhaustive_posweights(A, y, ds)
            out[nm + "_A"] = A
            out[nm + "_y"] = y
            out[nm + "_sizes"] = ds
            out[nm + "_w"] = np.asarray(wn, dtype=np.float64)
            out[nm + "_sub"] = np.asarray(isub, dtype=np.int64)
            out[nm + "_tot"] = np.asarray(itot, dtype=np.int64)
            out[nm + "_obj"] = np.float64(mo)
            out[nm + "_yrec"] = np.asarray(yr, dtype=np.float64)
            names.append(nm)
    out["names"] = np.array(names)

    # reference boundary tables, re-run through the reference
    s2, s3 = np.sqrt(2.0), np.sqrt(3.0)
    A1 = np.array([[0.], [1.], [0.]])
    Y1 = np.array([[1, 0, s2 / 2, 0, s2 / 2], [0, 0, -s2 / 2, 2, s2 / 2], [0, 1, 0, 0, 0]], dtype=float)
    w1, o1 = [], []
    for i in range(Y1.shape[1]):
        r = mfu.solve_exhaustive_posweights(A1, Y1[:, i], np.array([1]))
        w1.append(float(r[0][0])); o1.append(float(r[3]))
    out["b1_A"], out["b1_Y"], out["b1_w"], out["b1_obj"] = A1, Y1, np.array(w1), np.array(o1)
    A2 = np.array([[0.5, s3 * 0.5], [s3 * 0.5, 0.5]])
    Y2 = np.array([[-s3 / 2, 0.5, -1, -s3 / 2, 0.5001, 0.5, s3 / 2, s2 / 2, -s2 / 2.0],
                   [0.5, -s3 / 2, 0, 0.5001, -s3 / 2, s3 / 2, 0.5, s2 / 2, -s2 / 2.0]])
    w2, o2 = [], []
    for i in range(Y2.shape[1]):
        r = mfu.solve_exhaustive_posweights(A2, Y2[:, i], np.array([1, 1]))
        w2.append(np.array(r[0], dtype=float)); o2.append(float(r[3]))
    out["b2_A"], out["b2_Y"], out["b2_w"], out["b2_obj"] = A2, Y2, np.array(w2).T, np.array(o2)
    np.savez_compressed(os.path.join(OUT, "solver_cases.npz"), **out)
    print("solver_cases.npz:", len(names), "cases")


def gen_rotation(mfu):
    """init_PGSE_multishell_interp tables + interp_PGSE_from_multishell outputs
    (exact-G and G-bracketing) on a synthetic table and on the UKBB fixture
    (atoms sub-sampled), + rotate_atom on the HCP fixture (atoms sub-sampled)."""
    rng = np.random.default_rng(7)
    out = {}
    # --- synthetic table: 2 b0 + 3 shells
    sch_ms = synth_scheme(rng, 2, [1000, 2000, 3000], [33, 34, 35])
    # add a near-perpendicular cluster in shell 1 to exercise the left-edge merge
    idx = np.where(sch_ms[:, 3] > 0)[0][:3]
    for t, i in enumerate(idx):
        v = np.array([np.cos(0.3 * t), np.sin(0.3 * t), 2e-4 * (t + 1)])
        sch_ms[i, :3] = v / np.linalg.norm(v)
    N = 23
    dic = synth_dictionary(rng, sch_ms, N)
    b0 = np.where(sch_ms[:, 3] == 0)[0]
    dic[b0, :] = dic[b0[0], :]
    ordir = np.array([0.0, 0.0, 1.0])
    ms = mfu.init_PGSE_multishell_interp(dic, sch_ms, ordir)
    out["syn_sch_ms"], out["syn_dic"], out["syn_ordir"] = sch_ms, dic, ordir
    out["syn_Gms_un"] = ms["Gms_un"]
    for s, f in enumerate(ms["interpolators"]):
        out["syn_x_%d" % s] = np.asarray(f.x, dtype=float)
        out["syn_y_%d" % s] = np.asarray(f.y, dtype=float)
    # subject scheme A: subset/reorder of dense scheme rows (exact-G)
    perm = rng.permutation(sch_ms.shape[0])[:60]
    schA = sch_ms[perm].copy()
    schA[:, :3] = np.where(schA[:, 3:4] > 0, unit(rng, 60), 0.0)
    # subject scheme B: G values between shells (bracketing) + b0
    schB = schA.copy()
    Gs = np.unique(sch_ms[:, 3])
    nz = schB[:, 3] > 0
    schB[nz, 3] = rng.uniform(Gs[1], Gs[-1], nz.sum())
    schB[nz, 3][:5] = Gs[2]
    # a very small but nonzero G between the b0 shell and shell 1 is legal too
    dirs = unit(rng, 6)
    dirs[0] = ordir
    dirs[1] = np.array([1.0, 0, 0])  # perpendicular: left extrapolation region
    outA = np.stack([mfu.interp_PGSE_from_multishell(schA, d, msinterp=ms) for d in dirs])
    outB = np.stack([mfu.interp_PGSE_from_multishell(schB, d, msinterp=ms) for d in dirs])
    # slow (uninitialised) path must agree (reference's own test idea)
    slowA = mfu.interp_PGSE_from_multishell(schA, dirs[2], dic, sch_ms, ordir)
    assert np.max(np.abs(slowA - outA[2])) < 1e-12
    out["syn_schA"], out["syn_schB"], out["syn_dirs"] = schA, schB, dirs
    out["syn_outA"], out["syn_outB"] = outA, outB

    # --- UKBB fixture (271 x 986), atoms subsampled to keep file small
    uk = mfu.loadmat(os.path.join(FIX, "ukbb_90_dirs_dictionary_hcp_deltas.mat"))
    sel = np.arange(0, uk["dictionary"].shape[1], 29)  # 34 atoms
    dic_uk = np.ascontiguousarray(uk["dictionary"][:, sel])
    ms_uk = mfu.init_PGSE_multishell_interp(dic_uk, uk["sch_mat"], uk["orientation"])
    out["uk_sch_ms"], out["uk_dic"], out["uk_ordir"] = uk["sch_mat"], dic_uk, np.asarray(uk["orientation"], float)
    out["uk_Gms_un"] = ms_uk["Gms_un"]
    for s, f in enumerate(ms_uk["interpolators"]):
        out["uk_x_%d" % s] = np.asarray(f.x, dtype=float)
        out["uk_y_%d" % s] = np.asarray(f.y, dtype=float)
    bvals = np.loadtxt(os.path.join(FIX, "1000521_bvals.txt"))
    bvecs = np.loadtxt(os.path.join(FIX, "1000521_bvecs.txt"))
    sch_subj = np.zeros((bvals.size, 7))
    sch_subj[:, :3] = bvecs.T
    sch_subj[:, 4:7] = uk["sch_mat"][0, 4:7]
    sch_subj[:, 3] = np.sqrt(bvals * 1e6 / (sch_subj[:, 4] - sch_subj[:, 5] / 3)) / (GAM * sch_subj[:, 5])
    sch_subj[:, 3] = np.minimum(sch_subj[:, 3], np.max(uk["sch_mat"][:, 3]))
    dirs_uk = unit(rng, 4)
    dirs_uk[0] = np.asarray(uk["orientation"], float)
    out["uk_sch_subj"], out["uk_dirs"] = sch_subj, dirs_uk
    out["uk_out_subj"] = np.stack([mfu.interp_PGSE_from_multishell(sch_subj, d, msinterp=ms_uk) for d in dirs_uk])
    out["uk_out_dense"] = np.stack([mfu.interp_PGSE_from_multishell(uk["sch_mat"], d, msinterp=ms_uk) for d in dirs_uk])
    # MC ground truth for the same atoms (reference test gate: max abs err < 1e-2)
    gt = mfu.loadmat(os.path.join(FIX, "1000521_dictionary_hcp_deltas.mat"))
    out["uk_mc_truth"] = np.ascontiguousarray(gt["dictionary"][:, sel])

    # --- HCP fixture: rotate_atom (552 x 782 with 40 b0 prepended)
    hcp = mfu.loadmat(os.path.join(FIX, "MC_dictionary_hcp.mat"))
    sch = mfu.import_PGSE_scheme(os.path.join(FIX, "hcp_mgh_1003.scheme1"))
    nb0 = 40
    sch_b0 = np.vstack((np.zeros((nb0, sch.shape[1])), sch))
    sch_b0[:nb0, 4:] = sch[0, 4:]
    selh = np.arange(2, 782, 41)  # 20 atoms, includes 86? -> add explicitly
    selh = np.unique(np.concatenate([selh, [86]]))
    sig = np.ascontiguousarray(hcp["dic_fascicle_refdir"][:, selh])
    S0 = np.ascontiguousarray(hcp["S0_fascicle"][:, selh])
    dirs_h = unit(rng, 3)
    refdir = np.array([0.0, 0.0, 1.0])
    out["hcp_sch"], out["hcp_sig"], out["hcp_S0"] = sch_b0, sig, S0
    out["hcp_DIFF"] = np.float64(hcp["WM_DIFF"])
    out["hcp_dirs"], out["hcp_refdir"] = dirs_h, refdir
    out["hcp_rot"] = np.stack([mfu.rotate_atom(sig, sch_b0, refdir, d, hcp["WM_DIFF"], S0) for d in dirs_h])
    out["hcp_rot_1d"] = mfu.rotate_atom(sig[:, 3], sch_b0, refdir, dirs_h[1], hcp["WM_DIFF"], S0[:, 3])
    np.savez_compressed(os.path.join(OUT, "rotation_cases.npz"), **out)
    print("rotation_cases.npz written")


def make_model_dict(rng, sch_ms, N, num_ear):
    dic = synth_dictionary(rng, sch_ms, N)
    b0 = np.where(sch_ms[:, 3] == 0)[0]
    dic[b0, :] = dic[b0[0], :]
    return {
        "dictionary": dic, "sch_mat": sch_ms, "orientation": np.array([0.0, 0.0, 1.0]),
        "num_atom": N, "num_ear": num_ear, "T2_csf": 2.0, "DIFF_csf": 3.0e-9,
        "T2_ear": 0.08, "DIFF_ear": np.linspace(0.2e-9, 1.2e-9, num_ear),
        "fasc_propnames": ["rad ", "fin"],
        "rad": rng.uniform(0.2e-6, 2e-6, N), "fin": rng.uniform(0.2, 0.9, N),
    }


def gen_fit(mfu, mfmod):
    """MFModel.fit end-to-end through the reference on a small mixed ROI:
    K in {0,1,2}, CSF / EAR on subsets (exercises _1, _2, _3, _4up and the
    params_vox packing, mf.py:340-461)."""
    rng = np.random.default_rng(99)
    sch_ms = synth_scheme(rng, 1, [1000, 2000, 3000], [20, 21, 22])
    N, E = 14, 3
    md = make_model_dict(rng, sch_ms, N, E)
    model = mfmod.MFModel(dict(md))
    ms = model.ms_interpolator
    # subject scheme: exact-G rows + a few bracketed rows
    sch = sch_ms.copy()
    nzr = np.where(sch[:, 3] > 0)[0]
    sch[nzr, :3] = unit(rng, nzr.size)
    Gs = np.unique(sch_ms[:, 3])
    sch[nzr[:7], 3] = rng.uniform(Gs[1], Gs[2], 7)
    V = 24
    numfasc = np.array([2] * 10 + [1] * 8 + [0] * 6)
    csf = np.zeros(V, bool); ear = np.zeros(V, bool)
    csf[[1, 2, 5, 11, 12, 18, 19, 22]] = True
    ear[[2, 3, 6, 12, 13, 19, 20, 23]] = True  # voxel 21: K=0, no csf, no ear -> zeros
    peaks = np.zeros((V, 6))
    peaks[:, :3] = unit(rng, V); peaks[:, 3:] = unit(rng, V)
    peaks[numfasc < 2, 3:] = 0
    peaks[numfasc < 1, :3] = 0
    b = (GAM * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / md["T2_csf"]) * np.exp(-b * md["DIFF_csf"])
    sig_ear = np.stack([np.exp(-sch[:, 6] / md["T2_ear"]) * np.exp(-b * d) for d in md["DIFF_ear"]], axis=1)
    Y = np.zeros((V, sch.shape[0]))
    for v in range(V):
        comps = []
        for k in range(numfasc[v]):
            Dk = mfu.interp_PGSE_from_multishell(sch, peaks[v, 3 * k:3 * k + 3], msinterp=ms)
            comps.append(Dk[:, rng.integers(0, N)])
        if csf[v]:
            comps.append(sig_csf)
        if ear[v]:
            comps.append(sig_ear[:, rng.integers(0, E)])
        if comps:
            nu = rng.dirichlet(np.ones(len(comps)))
            Y[v] = 500 * np.stack(comps, 1) @ nu
        Y[v] += rng.normal(0, 500 / 30.0, sch.shape[0])
    # voxel 9 (K=2): pure single-fascicle signal -> single-active tie-break path
    D0 = mfu.interp_PGSE_from_multishell(sch, peaks[9, :3], msinterp=ms)
    Y[9] = 400 * D0[:, 4]
    mask = np.ones((4, 6))
    data = Y.reshape(4, 6, -1)
    fit = model.fit(data, mask, numfasc.reshape(4, 6), peaks=peaks.reshape(4, 6, 6), pgse_scheme=sch,
                    csf_mask=csf.reshape(4, 6).astype(float), ear_mask=ear.reshape(4, 6).astype(float), verbose=0)
    out = {"sch_ms": sch_ms, "sch": sch, "dictionary": md["dictionary"], "N": N, "E": E,
           "T2_csf": md["T2_csf"], "DIFF_csf": md["DIFF_csf"], "T2_ear": md["T2_ear"], "DIFF_ear": md["DIFF_ear"],
           "rad": md["rad"], "fin": md["fin"],
           "numfasc": numfasc, "csf": csf, "ear": ear, "peaks": peaks, "Y": Y}
    out["param_names"] = np.array(fit.param_names)
    for p in fit.param_names:
        out["map_" + p] = getattr(fit, p)
    # raw params_in_mask rows via _fit_voxel semantics: re-run serial loop pieces
    np.savez_compressed(os.path.join(OUT, "fit_cases.npz"), **out)
    print("fit_cases.npz written; params:", fit.param_names)

    # second case: no CSF/EAR anywhere, maxfasc = 1  (config-1-like plumbing)
    V2 = 12
    pk = unit(rng, V2)
    Y2 = np.zeros((V2, sch.shape[0]))
    for v in range(V2):
        Dk = mfu.interp_PGSE_from_multishell(sch, pk[v], msinterp=ms)
        Y2[v] = 300 * Dk[:, rng.integers(0, N)] + rng.normal(0, 5, sch.shape[0])
    fit2 = model.fit(Y2, np.ones(V2), 1, peaks=pk, pgse_scheme=sch, verbose=0)
    out2 = {"peaks": pk, "Y": Y2, "param_names": np.array(fit2.param_names)}
    for p in fit2.param_names:
        out2["map_" + p] = getattr(fit2, p)
    np.savez_compressed(os.path.join(OUT, "fit_cases_k1.npz"), **out2)
    print("fit_cases_k1.npz written; params:", fit2.param_names)


def gen_c2_small(mfu, mfmod):
    """A handful of config-2-shaped voxels (2 fascicles, M=200) at reduced N so the
    CPython reference finishes in seconds; full params via MFModel.fit."""
    rng = np.random.default_rng(1)
    sch_ms = synth_scheme(rng, 2, [1000, 2000, 3000], [66, 66, 66])
    N = 48
    md = make_model_dict(rng, sch_ms, N, 2)
    model = mfmod.MFModel(dict(md))
    V = 6
    peaks = np.concatenate([unit(rng, V), unit(rng, V)], axis=1)
    ms = model.ms_interpolator
    Y = np.zeros((V, sch_ms.shape[0]))
    for v in range(V):
        D1 = mfu.interp_PGSE_from_multishell(sch_ms, peaks[v, :3], msinterp=ms)
        D2 = mfu.interp_PGSE_from_multishell(sch_ms, peaks[v, 3:], msinterp=ms)
        nu = rng.dirichlet([1, 1])
        Y[v] = 500 * (nu[0] * D1[:, rng.integers(0, N)] + nu[1] * D2[:, rng.integers(0, N)])
        Y[v] += rng.normal(0, 500 / 30.0, sch_ms.shape[0])
    fit = model.fit(Y, np.ones(V), 2, peaks=peaks, pgse_scheme=sch_ms, verbose=0)
    out = {"sch_ms": sch_ms, "dictionary": md["dictionary"], "peaks": peaks, "Y": Y,
           "param_names": np.array(fit.param_names), "rad": md["rad"], "fin": md["fin"]}
    for p in fit.param_names:
        out["map_" + p] = getattr(fit, p)
    np.savez_compressed(os.path.join(OUT, "fit_c2_small.npz"), **out)
    print("fit_c2_small.npz written")


def gen_inputs(mfu, mfmod):
    """Input-normalisation callers of the path, executed by the reference:
    bvals/bvecs -> scheme (mfu:2197-2300), tensors / colat-longit -> peaks inside MFModel.fit (mf.py:723-800),
    scheme file import (mfu:2128-2192)."""
    rng = np.random.default_rng(2024)
    out = {}
    # UKBB fixture: subject bvals/bvecs snapped onto the dense scheme
    dense = mfu.import_PGSE_scheme(os.path.join(FIX, "ukbb_scheme_90_dirs.scheme"))
    bvals = np.loadtxt(os.path.join(FIX, "1000521_bvals.txt"))
    bvecs = np.loadtxt(os.path.join(FIX, "1000521_bvecs.txt"))
    out["uk_dense"] = dense
    out["uk_bvals"], out["uk_bvecs"] = bvals, bvecs
    out["uk_scheme_from_bvals"] = mfu.get_PGSE_scheme_from_bval_bvec_dense(dense, bvals, bvecs, 1e-3)
    out["hcp_scheme"] = mfu.import_PGSE_scheme(os.path.join(FIX, "hcp_mgh_1003.scheme1"))
    # MFModel.fit through tensors=, colat_longit= and bvals/bvecs on a synthetic model
    sch_ms = synth_scheme(rng, 1, [1000, 2000], [24, 25])
    N, E = 12, 2
    md = make_model_dict(rng, sch_ms, N, E)
    model = mfmod.MFModel(dict(md))
    V = 10
    d1, d2 = unit(rng, V), unit(rng, V)
    ms = model.ms_interpolator
    Y = np.zeros((V, sch_ms.shape[0]))
    for v in range(V):
        A1 = mfu.interp_PGSE_from_multishell(sch_ms, d1[v], msinterp=ms)
        A2 = mfu.interp_PGSE_from_multishell(sch_ms, d2[v], msinterp=ms)
        Y[v] = 300 * A1[:, rng.integers(0, N)] + 200 * A2[:, rng.integers(0, N)] + rng.normal(0, 10, sch_ms.shape[0])
    # tensors with principal eigenvector d (NIfTI 'column' order xx xy yy xz yz zz), shape (V, 1, 6) and (V, 6)
    def tens(d, lam=(1.7e-3, 0.4e-3, 0.3e-3)):
        T = np.zeros((d.shape[0], 6))
        for v in range(d.shape[0]):
            e1 = d[v]; tmp = np.cross(e1, [0.3, 0.5, 0.8]); e2 = tmp / np.linalg.norm(tmp); e3 = np.cross(e1, e2)
            Dm = lam[0] * np.outer(e1, e1) + lam[1] * np.outer(e2, e2) + lam[2] * np.outer(e3, e3)
            T[v] = [Dm[0, 0], Dm[0, 1], Dm[1, 1], Dm[0, 2], Dm[1, 2], Dm[2, 2]]
        return T
    T1, T2 = tens(d1)[:, None, :], tens(d2)
    gam = GAM
    bv = (gam * sch_ms[:, 3] * sch_ms[:, 5]) ** 2 * (sch_ms[:, 4] - sch_ms[:, 5] / 3) / 1e6
    fit_t = model.fit(Y, np.ones(V), 2, tensors=[T1, T2], bvals=bv, bvecs=sch_ms[:, :3].T.copy(), verbose=0)
    th1 = np.arccos(d1[:, 2]); ph1 = np.arctan2(d1[:, 1], d1[:, 0])
    fit_c = model.fit(Y, np.ones(V), 1, colat_longit=np.stack([th1, ph1], 1), pgse_scheme=sch_ms, verbose=0)
    out.update({"m_sch_ms": sch_ms, "m_dictionary": md["dictionary"], "m_rad": md["rad"], "m_fin": md["fin"],
                "m_DIFF_ear": md["DIFF_ear"], "m_Y": Y, "m_T1": T1, "m_T2": T2, "m_bvals": bv, "m_bvecs": sch_ms[:, :3].T.copy(),
                "m_colat": np.stack([th1, ph1], 1), "t_names": np.array(fit_t.param_names), "c_names": np.array(fit_c.param_names)})
    for p in fit_t.param_names:
        out["t_" + p] = getattr(fit_t, p)
    for p in fit_c.param_names:
        out["c_" + p] = getattr(fit_c, p)
    np.savez_compressed(os.path.join(OUT, "input_cases.npz"), **out)
    print("input_cases.npz written")


def gen_cleanup(mfu, mfmod):
    """cleanup_2fascicles (mf.py:36-335) in its three peak modes + DT helper round trips (mfu:865-1135)."""
    rng = np.random.default_rng(99)
    shape = (6, 5, 4)
    n = int(np.prod(shape))
    mask = (rng.uniform(size=shape) > 0.15).astype(float)
    # weights drawn to hit every branch: tiny, lopsided, comparable, tied
    f1 = rng.choice([0.0, 0.03, 0.07, 0.1, 0.19, 0.3, 0.5, 0.8], size=shape) + rng.uniform(0, 0.01, shape) * (rng.uniform(size=shape) > 0.3)
    f2 = rng.choice([0.0, 0.03, 0.07, 0.1, 0.19, 0.3, 0.5, 0.8], size=shape) + rng.uniform(0, 0.01, shape) * (rng.uniform(size=shape) > 0.3)
    d1 = unit(rng, n).reshape(shape + (3,))
    d2 = unit(rng, n).reshape(shape + (3,))
    close = rng.uniform(size=shape) < 0.3           # crossing angle below / around 15 deg, both signs
    pert = d1 + 0.25 * rng.uniform(size=shape + (1,)) * unit(rng, n).reshape(shape + (3,))
    pert /= np.linalg.norm(pert, axis=-1, keepdims=True)
    d2[close] = (pert * rng.choice([-1.0, 1.0], size=shape + (1,)))[close]
    out = {"mask": mask, "f1": f1, "f2": f2, "d1": d1, "d2": d2}
    pk, nf = mfmod.cleanup_2fascicles(f1, f2, 'peaks', d1.copy(), d2.copy(), mask)
    out["peaks_pk"], out["peaks_nf"] = pk, nf
    cl1 = np.stack([np.arccos(d1[..., 2]), np.arctan2(d1[..., 1], d1[..., 0])], -1)
    cl2 = np.stack([np.arccos(d2[..., 2]), np.arctan2(d2[..., 1], d2[..., 0])], -1)
    pk, nf = mfmod.cleanup_2fascicles(None, None, 'colat_longit', cl1, cl2, mask, frac12=np.stack([f1, f2], -1)[..., None, :])
    out["cl1"], out["cl2"], out["colat_pk"], out["colat_nf"] = cl1, cl2, pk, nf
    T = mfu.peaks_to_DT_vec(np.stack([d1, d2], axis=-2).copy(), 'column')
    zero = rng.uniform(size=shape) < 0.1            # zero tensors -> zero peaks
    T[1][zero] = 0
    pk, nf = mfmod.cleanup_2fascicles(f1, f2, 'tensor', T[0][..., None, :], T[1], mask)
    out["T1"], out["T2"], out["tensor_pk"], out["tensor_nf"] = T[0], T[1], pk, nf
    for order in ("row", "column", "diagonal"):
        A = mfu.DT_vec_to_2Darray(T[0], order)
        out["dt2d_" + order] = A
        out["dtvec_" + order] = mfu.DT_array_to_vec(A, order)
        out["dtpk_" + order] = mfu.DT_vec_to_peaks(T[1], order, mask)
    np.savez_compressed(os.path.join(OUT, "cleanup_cases.npz"), **out)
    print("cleanup_cases.npz written; num_fasc histogram", np.bincount(out["peaks_nf"].astype(int).ravel()))


def gen_mc(mfu):
    """monte_carlo_average (mfu:2758-2810) and get_PGSE_from_phases (mfu:2813-3015) on a small synthetic
    phase table: 3 simulated (Delta, delta) acquisitions x 700 spins x 3 components."""
    rng = np.random.default_rng(4242)
    n_ref, n_spin = 3, 700
    Dl = np.array([20e-3, 35e-3, 50e-3]); dl = np.array([8e-3, 10e-3, 12e-3])
    Gsim = 0.05
    sch_sim = np.zeros((n_ref, 7))
    sch_sim[:, :3] = 1 / np.sqrt(3.0)
    sch_sim[:, 3], sch_sim[:, 4], sch_sim[:, 5], sch_sim[:, 6] = Gsim, Dl, dl, Dl + dl + 5e-3
    # phases ~ gamma*G*delta*displacement, a few rad; heavier tails on one component
    ph = rng.standard_normal((n_ref * n_spin, 3)) * np.array([1.5, 1.0, 4.0])
    out = {"sch_sim": sch_sim, "phases": ph, "n_spin": np.int64(n_spin)}
    # direct kernel call
    n_seq = 23
    dm = rng.integers(0, n_ref, n_seq).astype(np.int64)
    gs = rng.uniform(-2, 2, (n_seq, 3))
    gs[0] = 0.0
    out["dm"], out["gs"] = dm, gs
    out["sig_direct"] = mfu.monte_carlo_average(ph, dm, gs, 1.0, n_spin)
    out["sig_direct_D"] = mfu.monte_carlo_average(ph, dm, gs, float(np.sqrt(2.0e-9 / 3.0e-9)), n_spin)
    out["sig_dim2"] = mfu.monte_carlo_average(np.ascontiguousarray(ph[:, :2]), dm, np.ascontiguousarray(gs[:, :2]), 1.0, n_spin)
    # through phase files: big-endian double and little-endian float
    sch = np.zeros((17, 7))
    g = unit(rng, 17)
    sch[:, :3] = g
    pick = rng.integers(0, n_ref, 17)
    sch[:, 3] = rng.uniform(0.0, 0.08, 17)
    sch[:, 4], sch[:, 5], sch[:, 6] = Dl[pick], dl[pick], (Dl + dl)[pick] + 5e-3
    sch[3, :4] = 0.0      # a b0
    out["sch_new"] = sch
    d = tempfile.mkdtemp(prefix="mcphases_")
    for i, nm in enumerate("xyz"):
        ph[:, i].astype(">f8").tofile(os.path.join(d, "sim_phase_%s.bdouble" % nm))
        ph[:, i].astype("<f4").tofile(os.path.join(d, "sim_phase_%s.lfloat" % nm))
    out["sig_files_bdouble"] = mfu.get_PGSE_from_phases(os.path.join(d, "sim_phase_x.bdouble"), sch_sim, sch)
    out["sig_files_lfloat_D"] = mfu.get_PGSE_from_phases(os.path.join(d, "sim_phase_x.lfloat"), sch_sim, sch,
                                                          D_sim=3.0e-9, D=2.0e-9)
    sch2 = sch.copy(); sch2[:, 2] = 0
    n2 = np.linalg.norm(sch2[:, :3], axis=1); sch2[n2 > 0, :3] /= n2[n2 > 0][:, None]
    out["sch_new_xy"] = sch2
    out["sig_files_dim2"] = mfu.get_PGSE_from_phases(os.path.join(d, "sim_phase_x.bdouble"), sch_sim, sch2, dim=2)
    np.savez_compressed(os.path.join(OUT, "mc_cases.npz"), **out)
    print("mc_cases.npz written", out["sig_direct"][:4], out["sig_files_bdouble"][:4])


def gen_nnls():
    """scipy.optimize.nnls (the third-party solver under solve_exhaustive_posweights_4up, mf_utils.py:640) on seeded
    4- to 6-column problems, incl. the degenerate ones that make a naive Lawson-Hanson loop cycle: duplicate columns,
    dependent columns, zero residual, nothing to fit.  Outputs of SciPy as installed in the build container."""
    import scipy
    import scipy.optimize as so
    rng = np.random.default_rng(77)
    M = 40
    t = np.linspace(0, 3, M)
    out = {"scipy_version": np.array(scipy.__version__)}
    k = 0
    for it in range(160):
        n = int(rng.integers(4, 7))
        kind = it % 8
        A = rng.uniform(0.05, 1, (M, n)) * np.exp(-np.outer(t, rng.uniform(0, 2, n)))
        xt = rng.uniform(0, 300, n) * (rng.uniform(size=n) < 0.6)
        b = A @ xt + rng.normal(0, 10, M)
        if kind == 1: A[:, 1] = A[:, 0]
        if kind == 2: b = A @ xt
        if kind == 3: b = -np.abs(b)
        if kind == 4: A[:, 2] = 0.5 * A[:, 0] + 0.5 * A[:, 1]
        if kind == 5: A[:, 1] = A[:, 0] * (1 + 1e-9)
        if kind == 6: A[:, 1] = A[:, 0]; b = 100 * A[:, 0]
        x, rn = so.nnls(A, b)
        out["A_%d" % k], out["b_%d" % k], out["x_%d" % k], out["rn_%d" % k], out["kind_%d" % k] = A, b, x, np.float64(rn), np.int64(kind)
        k += 1
    out["count"] = np.int64(k)
    np.savez_compressed(os.path.join(OUT, "nnls_cases.npz"), **out)
    print("nnls_cases.npz written (%d problems)" % k)


def gen_real(mfu, mfmod, part=None):
    """The real dictionaries held by the reference's own tests, stored as DATA (arrays only), plus a few voxels fitted
    by the reference itself at full dictionary size:
      * UKBB 90-direction dictionary (271 x 986 atoms, 10 EAR columns; near-duplicate atoms) with the protocol of
        subject 1000521 (105 rows, 9 distinct G values against 4 table shells -> G-bracketing) ->
        real_ukbb.npz; reference fits: 4 voxels K=2, 1 voxel K=2+CSF (_3), 1 voxel K=2+CSF+EAR (_4up) -> real_ukbb_fits.npz
      * HCP-MGH dictionary (552 x 782) and its scheme with 40 b0 rows (test_hcp_dict's inputs) -> real_hcp.npz
    The reference's pure-Python kernels take minutes per voxel here; `part` = "data" | "k2" | "k2csf" | "k2csfear"."""
    uk = mfu.loadmat(os.path.join(FIX, "ukbb_90_dirs_dictionary_hcp_deltas.mat"))
    subj = mfu.loadmat(os.path.join(FIX, "1000521_dictionary_hcp_deltas.mat"))
    sch_subj = np.ascontiguousarray(subj["sch_mat"], dtype=float)
    model = {"dictionary": np.ascontiguousarray(uk["dictionary"], dtype=float), "sch_mat": np.asarray(uk["sch_mat"], float),
             "orientation": np.asarray(uk["orientation"], float), "num_atom": int(uk["Nsubs"]), "num_ear": int(uk["Near"]),
             "T2_csf": float(uk["T2_csf"]), "DIFF_csf": float(uk["CSF_DIFF"]), "T2_ear": float(uk["T2_ear"]),
             "DIFF_ear": np.asarray(uk["Dear"], float), "fasc_propnames": ["rad", "fin"],
             "rad": np.asarray(uk["rad"], float), "fin": np.asarray(uk["fin"], float)}
    if part in (None, "data"):
        np.savez_compressed(os.path.join(OUT, "real_ukbb.npz"), sch_subj=sch_subj,
                            **{k: np.asarray(v) for k, v in model.items() if k != "fasc_propnames"})
        hcp = mfu.loadmat(os.path.join(FIX, "MC_dictionary_hcp.mat"))
        sch = mfu.import_PGSE_scheme(os.path.join(FIX, "hcp_mgh_1003.scheme1"))
        nb0 = 40
        sch_b0 = np.vstack((np.zeros((nb0, sch.shape[1])), sch))
        sch_b0[:nb0, 4:] = sch[0, 4:]
        np.savez_compressed(os.path.join(OUT, "real_hcp.npz"), dictionary=np.asarray(hcp["dic_fascicle_refdir"], float),
                            S0=np.asarray(hcp["S0_fascicle"], float), sch_mat=sch_b0, sig_csf=np.asarray(hcp["sig_csf"], float),
                            WM_DIFF=np.float64(hcp["WM_DIFF"]), CSF_DIFF=np.float64(hcp["CSF_DIFF"]))
        print("real_ukbb.npz, real_hcp.npz written")
    cases = {"k2": (4, 0, 0, 101), "k2csf": (1, 1, 0, 102), "k2csfear": (1, 1, 1, 103)}
    for name, (V, c, e, seed) in cases.items():
        if part not in (None, name):
            continue
        rng = np.random.default_rng(seed)
        ms = mfu.init_PGSE_multishell_interp(model["dictionary"], model["sch_mat"], model["orientation"])
        N = model["num_atom"]
        peaks = np.concatenate([unit(rng, V), unit(rng, V)], axis=1)
        atoms = rng.integers(0, N, (V, 2))
        gam = mfu.get_gyromagnetic_ratio('H')
        b = (gam * sch_subj[:, 3] * sch_subj[:, 5]) ** 2 * (sch_subj[:, 4] - sch_subj[:, 5] / 3)
        sig_csf = np.exp(-sch_subj[:, 6] / model["T2_csf"]) * np.exp(-b * model["DIFF_csf"])
        sig_ear = np.stack([np.exp(-sch_subj[:, 6] / model["T2_ear"]) * np.exp(-b * D) for D in model["DIFF_ear"]], axis=1)
        Y = np.zeros((V, sch_subj.shape[0]))
        for v in range(V):
            nu = rng.dirichlet(np.ones(2 + c + e))
            for k in range(2):
                Dk = mfu.interp_PGSE_from_multishell(sch_subj, peaks[v, 3 * k:3 * k + 3], msinterp=ms)
                Y[v] += 500.0 * nu[k] * Dk[:, atoms[v, k]]
            if c:
                Y[v] += 500.0 * nu[2] * sig_csf
            if e:
                Y[v] += 500.0 * nu[2 + c] * sig_ear[:, int(rng.integers(0, model["num_ear"]))]
        Y += rng.normal(0, 500.0 / 30.0, Y.shape)
        mask = np.ones(V, dtype=int)
        m = mfmod.MFModel(model)
        fit = m.fit(Y, mask, 2, peaks=peaks, pgse_scheme=sch_subj, csf_mask=(mask if c else None),
                    ear_mask=(mask if e else None), verbose=0)
        out = {"Y": Y, "peaks": peaks, "csf": np.int64(c), "ear": np.int64(e), "param_names": np.array(fit.param_names)}
        for pn in fit.param_names:
            out["map_" + pn] = np.asarray(getattr(fit, pn))
        np.savez_compressed(os.path.join(OUT, "real_ukbb_fit_%s.npz" % name), **out)
        print("real_ukbb_fit_%s.npz written" % name)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--part", default=None, help="gen_real only: data | k2 | k2csf | k2csfear")
    a = ap.parse_args()
    mfu, mfmod = import_reference()
    todo = {"solver": lambda: gen_solver(mfu), "rotation": lambda: gen_rotation(mfu),
            "fit": lambda: gen_fit(mfu, mfmod), "c2": lambda: gen_c2_small(mfu, mfmod),
            "inputs": lambda: gen_inputs(mfu, mfmod), "cleanup": lambda: gen_cleanup(mfu, mfmod),
            "mc": lambda: gen_mc(mfu), "real": lambda: gen_real(mfu, mfmod, a.part), "nnls": gen_nnls}
    for k, fn in todo.items():
        if a.only in (None, k):
            fn()
