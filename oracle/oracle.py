"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Python/ctypes front-end of ``oracle/mf_oracle.c`` plus NumPy restatements of the
voxel-independent table builders of the reference.  It is the *checker* for the
HIP path (tests/, ``__graft_entry__.smoke()``) and the timed CPU baseline
(``bench.py`` ``cpu_baseline`` leg).  The product package never imports it.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every entry point
against ``tests/golden/*.npz`` (outputs of the reference's own source run in the
build container by ``tests/golden/gen_golden.py``) and against the reference's
known-answer tables (test_exhaustive_fingerprinting.py:38-89).

Citations: mfu = reference ``microstructure_fingerprinting/mf_utils.py``,
           mf  = reference ``microstructure_fingerprinting/mf.py``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmf_oracle.so")
_LIB = None

GAMMA_H = 2 * np.pi * 42.577480e6  # mfu:1142


def build(force=False):
    """Compile the C restatement (gcc, strict IEEE, no FMA contraction)."""
    src = os.path.join(_HERE, "mf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared", "-fPIC",
                               src, "-o", _SO, "-lm"])
    return _SO


def lib():
    global _LIB
    if _LIB is None:
        build()
        _LIB = C.CDLL(_SO)
        dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_long)
        _LIB.orc_solve_exhaustive.argtypes = [dp, C.c_long, C.c_int, lp, C.c_int, dp, dp, lp, lp, dp, dp]
        _LIB.orc_solve_exhaustive.restype = C.c_int
        _LIB.orc_interp.argtypes = [C.c_int, C.c_int, dp, ip, dp, dp, dp, C.c_int, dp, dp, C.c_long, dp]
        _LIB.orc_interp.restype = C.c_int
        _LIB.orc_rotate_eval.argtypes = [C.c_int, ip, dp, dp, dp, C.c_int, C.c_int, ip, dp, dp, dp]
        _LIB.orc_rotate_eval.restype = C.c_int
        _LIB.orc_fit_batch.argtypes = [C.c_int, C.c_int, dp, ip, dp, dp, dp, C.c_int, dp, ip,
                                       C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte), dp, C.c_int, C.c_int, C.c_int,
                                       dp, dp, C.c_int, C.c_long, dp, C.c_int]
        _LIB.orc_fit_batch.restype = C.c_int
        _LIB.orc_monte_carlo_average.argtypes = [dp, C.c_long, C.c_int, lp, dp, C.c_double, C.c_long, C.c_long, dp,
                                                 C.c_int]
        _LIB.orc_monte_carlo_average.restype = C.c_int
        _LIB.orc_max_threads.restype = C.c_int
        _LIB.orc_nnls.argtypes = [dp, C.c_int, C.c_int, dp, dp, dp]
        _LIB.orc_nnls.restype = C.c_int
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_long))


# ---------------------------------------------------------------------------
# solver: mfu:115-214
# ---------------------------------------------------------------------------
def solve_exhaustive_posweights(A, y, dicsizes):
    assert isinstance(A, np.ndarray) and A.ndim == 2            # mfu:159-160
    assert not np.any(np.all(A == 0, axis=0))                    # mfu:162
    A = np.ascontiguousarray(A, dtype=np.float64)
    assert isinstance(y, np.ndarray)
    y = np.ascontiguousarray(y, dtype=np.float64)
    assert A.size > 0 and y.size > 0
    assert A.shape[0] == y.size                                  # mfu:175
    assert isinstance(dicsizes, np.ndarray) and np.all(dicsizes > 0)
    sizes = np.ascontiguousarray(dicsizes, dtype=np.int64)
    assert A.shape[1] == np.sum(sizes)                           # mfu:188
    Kp = sizes.size
    w = np.zeros(Kp)
    sub = np.zeros(Kp, dtype=np.int64)
    tot = np.zeros(Kp, dtype=np.int64)
    mo = np.zeros(1)
    yrec = np.zeros(A.shape[0])
    rc = lib().orc_solve_exhaustive(_dp(A), A.shape[1], A.shape[0], _lp(sizes), Kp, _dp(y), _dp(w), _lp(sub),
                                    _lp(tot), _dp(mo), _dp(yrec))
    if rc == 3:
        raise RuntimeError("Maximum number of iterations reached.")
    if rc:
        raise RuntimeError("oracle solver error %d" % rc)
    return w, sub, tot, float(mo[0]), yrec


# ---------------------------------------------------------------------------
# knot tables
# ---------------------------------------------------------------------------
def _merge_left_cluster(x, Y):
    """mfu:2059-2072 (same code at mfu:1398-1412): left-edge cluster -> its mean."""
    almost_perp = np.abs(x - x[0]) < 1e-3
    cs = int(np.sum(almost_perp))
    if cs > 1:
        x = np.append(np.mean(x[almost_perp]), x[cs:])
        Y = np.append(np.mean(Y[almost_perp, :], axis=0, keepdims=True), Y[cs:, :], axis=0)
    return x, Y


def init_tables(sig_ms, sch_mat_ms, ordir):
    """init_PGSE_multishell_interp, mfu:1959-2085 -> flat tables."""
    ordir = np.squeeze(np.asarray(ordir, dtype=np.float64))
    if ordir.size != 3:
        raise ValueError("ordir should have 3 entries")
    if not np.all(np.isclose(sch_mat_ms[0, 4:7], sch_mat_ms[:, 4:7])):
        raise ValueError("Delta, delta and TE values should all be identical in multi-shell sampling.")
    sig_ms = np.asarray(sig_ms, dtype=np.float64)
    if sig_ms.ndim == 1:
        sig_ms = sig_ms.reshape((sig_ms.size, 1))
    if np.abs(1 - np.sqrt((ordir ** 2).sum())) > 1e-3:
        raise ValueError("Orientation vector of the multi-shell signal must have unit norm.")
    gn = np.sqrt(np.sum(sch_mat_ms[:, 0:3] ** 2, axis=1))
    if np.any(np.abs(1 - gn[gn > 0]) > 1e-3):
        raise ValueError("Gradient directions should all either have zero or unit norm.")
    dots = np.abs(np.dot(sch_mat_ms[:, 0:3], ordir))
    G_un, i_G = np.unique(sch_mat_ms[:, 3], return_inverse=True)
    xs, Ys = [], []
    for i in range(G_un.shape[0]):
        ind = np.where(i_G == i)[0]
        if G_un[i] == 0:
            chk = np.all(np.isclose(sig_ms[ind, :], sig_ms[ind[0], :]), axis=0)
            if np.any(~chk):
                raise ValueError("Distinct signal values in provided multi-shell sampling for zero gradients")
            xs.append(np.array([0.0, 1.0]))
            Ys.append(np.repeat([sig_ms[ind[0], :]], 2, axis=0))
            continue
        xu, iu = np.unique(dots[ind], return_index=True)
        Yd = sig_ms[ind, :][iu, :]
        xu, Yd = _merge_left_cluster(xu, Yd)
        xs.append(np.asarray(xu, dtype=np.float64))
        Ys.append(np.asarray(Yd, dtype=np.float64))
    off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.int32)
    return {"S": len(xs), "N": sig_ms.shape[1], "G_un": np.ascontiguousarray(G_un, dtype=np.float64), "off": off,
            "x": np.ascontiguousarray(np.concatenate(xs)), "Y": np.ascontiguousarray(np.concatenate(Ys, axis=0)),
            "scheme_DeldelTE": np.array(sch_mat_ms[0, 4:7], dtype=np.float64), "xs": xs, "Ys": Ys}


def interp(sch_mat, newdir, T):
    """interp_PGSE_from_multishell fast mode, mfu:1785-1956 (checks + evaluation)."""
    sch_mat = np.ascontiguousarray(sch_mat, dtype=np.float64)
    if not np.all(np.isclose(T["scheme_DeldelTE"], sch_mat[:, 4:7])):
        raise ValueError("Delta, delta and TE values should all be identical to those in the multi-shell sampling.")
    newdir = np.squeeze(np.asarray(newdir, dtype=np.float64))
    if newdir.size != 3:
        raise ValueError("newdir should have 3 entries.")
    if np.abs(1 - np.sqrt((newdir ** 2).sum())) > 1e-3:
        raise ValueError("Orientation vector of the new signal must have unit norm.")
    gn = np.sqrt(np.sum(sch_mat[:, 0:3] ** 2, axis=1))
    if np.any(np.abs(1 - gn[gn > 0]) > 1e-3):
        raise ValueError("Gradient directions should all either have zero or unit norm.")
    M, N = sch_mat.shape[0], T["N"]
    out = np.zeros((M, N))
    tmp = np.zeros(2 * N)
    nd = np.ascontiguousarray(newdir)
    rc = lib().orc_interp(T["S"], N, _dp(T["G_un"]), _ip(T["off"]), _dp(T["x"]), _dp(T["Y"]), _dp(sch_mat), M,
                          _dp(nd), _dp(out), N, _dp(tmp))
    if rc == 1:
        raise ValueError("Gradient intensity not in the range spanned by the multi-shell sampling. "
                         "Extrapolation not supported.")
    return np.squeeze(out)


def rotate_tables(sig, sch_mat, ordir, DIFF, S0):
    """Direction-independent part of rotate_atom, mfu:1233-1412: per-(G,Del,del) shell knots."""
    sig = np.asarray(sig, dtype=np.float64)
    S0 = np.asarray(S0, dtype=np.float64)
    if sig.ndim == 1:
        sig = sig.reshape((sig.size, 1))
    if S0.ndim == 1:
        S0 = S0[:, np.newaxis]
    if sch_mat.shape[1] < 6:
        raise ValueError("sch_mat must be a N-by-6 or7 matrix")
    if sch_mat.shape[0] != sig.shape[0]:
        raise ValueError("sch_mat and sig must have the same number of rows")
    assert sig.shape == S0.shape
    ordir = np.asarray(ordir, dtype=np.float64)
    gnorm = np.sqrt((sch_mat[:, 0:3] ** 2).sum(axis=1, keepdims=True))
    gnorm[gnorm == 0] = np.inf
    ordots = np.abs(np.dot(sch_mat[:, 0:3] / gnorm, ordir / np.sqrt((ordir ** 2).sum())))
    bvals = ((GAMMA_H * sch_mat[:, 3] * sch_mat[:, 5]) ** 2 * (sch_mat[:, 4] - sch_mat[:, 5] / 3))
    GdD_un, i_un = np.unique(sch_mat[:, 3:6], return_inverse=True, axis=0)
    i_un = np.asarray(i_un).reshape(-1)
    shell_of_row = np.full(sch_mat.shape[0], -1, dtype=np.int32)
    xs, Ys = [], []
    for i in range(GdD_un.shape[0]):
        ind = np.where(i_un == i)[0]
        bval = bvals[ind[0]]
        if bval == 0:
            continue
        if ind.size < 2:
            raise ValueError("Fewer than 2 identical (G, Del, del) triplets detected, probably not a HARDI shell.")
        ok = np.all(np.isclose(S0[ind, :], S0[ind[0], :]), axis=0)
        if np.any(~ok):
            raise ValueError("Distinct values in provided S0 image for shell")
        xu, iu = np.unique(ordots[ind], return_index=True)
        Yd = sig[ind, :][iu, :]
        if not np.any(xu == 1):
            xu = np.append(xu, [1])
            free = np.exp(-bval * DIFF) * S0[ind[0], :]
            Yd = np.append(Yd, np.reshape(free, (1, -1)), axis=0)
        xu, Yd = _merge_left_cluster(xu, Yd)
        shell_of_row[ind] = len(xs)
        xs.append(np.asarray(xu, dtype=np.float64))
        Ys.append(np.asarray(Yd, dtype=np.float64))
    off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.int32)
    return {"N": sig.shape[1], "off": off, "x": np.ascontiguousarray(np.concatenate(xs)) if xs else np.zeros(1),
            "Y": np.ascontiguousarray(np.concatenate(Ys, axis=0)) if Ys else np.zeros((1, sig.shape[1])),
            "shell_of_row": shell_of_row, "sig": np.ascontiguousarray(sig)}


def rotate_atom(sig, sch_mat, ordir, newdir, DIFF, S0):
    """rotate_atom, mfu:1205-1437."""
    shp = np.asarray(sig).shape
    sch_mat = np.ascontiguousarray(sch_mat, dtype=np.float64)
    T = rotate_tables(sig, sch_mat, ordir, DIFF, S0)
    M, N = T["sig"].shape
    out = np.zeros((M, N))
    nd = np.ascontiguousarray(np.asarray(newdir, dtype=np.float64))
    lib().orc_rotate_eval(N, _ip(T["off"]), _dp(T["x"]), _dp(T["Y"]), _dp(sch_mat), sch_mat.shape[1], M,
                          _ip(T["shell_of_row"]), _dp(T["sig"]), _dp(nd), _dp(out))
    if np.any(np.isnan(out)):
        raise ValueError("Nan detected after rotation of substrate(s)")
    return out.reshape(shp)


# ---------------------------------------------------------------------------
# voxel loop: mf:340-461 + mf:1017-1028
# ---------------------------------------------------------------------------
def fit_batch(T, sch_mat, Y, K, csf, ear, peaks, maxfasc, csf_on, ear_on, sig_csf, sig_ear, E, nthreads=1):
    sch_mat = np.ascontiguousarray(sch_mat, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    V, M = Y.shape
    K = np.ascontiguousarray(K, dtype=np.int32)
    csf = np.ascontiguousarray(csf, dtype=np.uint8)
    ear = np.ascontiguousarray(ear, dtype=np.uint8)
    peaks = np.ascontiguousarray(peaks, dtype=np.float64).reshape(V, -1)
    assert peaks.shape[1] == 3 * maxfasc or (maxfasc == 0)
    if maxfasc == 0:
        peaks = np.zeros((V, 1))
    npar = 1 + 2 * maxfasc + int(csf_on) + 2 * int(ear_on) + 2
    out = np.zeros((V, npar))
    sc = np.ascontiguousarray(sig_csf, dtype=np.float64) if sig_csf is not None else np.zeros(M)
    se = np.ascontiguousarray(sig_ear, dtype=np.float64) if sig_ear is not None else np.zeros((M, max(E, 1)))
    rc = lib().orc_fit_batch(T["S"], T["N"], _dp(T["G_un"]), _ip(T["off"]), _dp(T["x"]), _dp(T["Y"]), _dp(sch_mat),
                             M, _dp(Y), _ip(K), csf.ctypes.data_as(C.POINTER(C.c_ubyte)),
                             ear.ctypes.data_as(C.POINTER(C.c_ubyte)), _dp(peaks), int(maxfasc), int(csf_on),
                             int(ear_on), _dp(sc), _dp(se), int(E), V, _dp(out), int(nthreads))
    if rc == 1:
        raise ValueError("Gradient intensity outside the multi-shell range; extrapolation not supported.")
    if rc:
        raise RuntimeError("oracle fit error %d" % rc)
    return out


# ---------------------------------------------------------------------------
# Monte-Carlo signal synthesis: mfu:2758-2810
def monte_carlo_average(sim_phases, delta_mapping, gscaling, Dscaling, num_spins, nthreads=1):
    ph = np.ascontiguousarray(sim_phases, dtype=np.float64)
    dm = np.ascontiguousarray(delta_mapping, dtype=np.int64)
    gs = np.ascontiguousarray(gscaling, dtype=np.float64)
    out = np.zeros(dm.size)
    rc = lib().orc_monte_carlo_average(_dp(ph), ph.shape[0], ph.shape[1], _lp(dm), _dp(gs), float(Dscaling),
                                       int(num_spins), dm.size, _dp(out), int(nthreads))
    if rc:
        raise IndexError("delta_mapping points outside the phase table")
    return out


def nnls(A, b):
    """scipy.optimize.nnls(A, b) restated (mfu:640): returns (x, rnorm); RuntimeError like SciPy when 3n iterations do not suffice."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(A.shape[1])
    rn = np.zeros(1)
    rc = lib().orc_nnls(_dp(A), A.shape[0], A.shape[1], _dp(b), _dp(x), _dp(rn))
    if rc == 3:
        raise RuntimeError("Maximum number of iterations reached.")
    if rc:
        raise RuntimeError("oracle nnls error %d" % rc)
    return x, float(rn[0])


def max_threads():
    return int(lib().orc_max_threads())


def cleanup_select(f1, f2, p1, p2, ang_min_deg=15, ratio=2.5, w_keep=0.20, w_small=0.075):
    """The voxel loop of cleanup_2fascicles restated with NumPy selects (ref mf.py:170-335): weights f1, f2 [n] and unit
    directions p1, p2 [n x 3] of the ROI voxels -> (peaks [n x 6], count [n]).  Checked against outputs of the reference
    (tests/golden/cleanup_cases.npz) in tests/test_cleanup.py; the referee of the device kernel (csrc/cleanup.hip)."""
    n = f1.shape[0]
    f = np.stack([f1, f2], axis=1).astype(np.float64)
    p = [np.array(p1, dtype=np.float64), np.array(p2, dtype=np.float64)]
    count = np.full(n, 2.0)

    def drop(slot, where):
        p[slot][where] = 0.0
        f[where, slot] = 0.0
    dp = np.sum(p[0] * p[1], axis=-1)
    merge = np.abs(np.clip(dp, -1, 1)) > np.cos(ang_min_deg * np.pi / 180)
    if np.any(merge):
        summed = p[0][merge] + p[1][merge] * np.sign(dp[merge])[:, np.newaxis]
        p[0][merge] = summed / np.sqrt(np.sum(summed ** 2, axis=1))[:, np.newaxis]
        f[merge, 0] = f1[merge] + f2[merge]
        drop(1, merge)
        count[merge] = 1
    rel0 = (f[:, 1] > ratio * f[:, 0]) & (f[:, 0] < w_keep)
    if np.any(rel0):
        p[0][rel0] = p[1][rel0]
        f[rel0, 0] = f[rel0, 1]
        drop(1, rel0)
        count[rel0] = (f[rel0, 0] > 0) * 1
    rel1 = (f[:, 0] > ratio * f[:, 1]) & (f[:, 1] < w_keep)
    if np.any(rel1):
        drop(1, rel1)
        count[rel1] = (f[rel1, 0] > 0) * 1
    abs0 = f[:, 0] < w_small
    if np.any(abs0):
        drop(0, abs0)
        count[abs0] = count[abs0] - 1
    abs1 = f[:, 1] < w_small
    if np.any(abs1):
        drop(1, abs1)
        count[abs1] = (f[abs1, 0] > 0) * 1
    swap = (f[:, 1] >= f[:, 0])[:, np.newaxis]     # the reference's reversed ascending argsort puts slot 1 first on ties
    peaks = np.concatenate([np.where(swap, p[1], p[0]), np.where(swap, p[0], p[1])], axis=1)
    return peaks, count
