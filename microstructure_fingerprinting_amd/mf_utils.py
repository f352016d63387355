"""Host-side mirror of the reference's ``mf_utils`` names for the fingerprinting hot path.

Same names, argument meaning and error behaviour as
``microstructure_fingerprinting/mf_utils.py`` of the reference (cited below as ``ref``), but
every numeric evaluation runs in the HIP library ``libmfx.so`` (include/mfx.h):

=============================================  ================================================
reference function (ref line)                  what runs where here
=============================================  ================================================
solve_exhaustive_posweights (ref:115)          checks on host, search on device (mfx_solve_exhaustive)
init_PGSE_multishell_interp (ref:1959)         per-shell knot tables built once on host (NumPy),
                                               uploaded lazily to HBM (mfx_tables_create)
interp_PGSE_from_multishell (ref:1693)         checks on host, evaluation on device (mfx_rotate)
rotate_atom (ref:1205)                         per-shell knot tables on host, evaluation on device
import_PGSE_scheme (ref:2128)                  host (input normalisation, once per fit)
get_PGSE_scheme_from_bval_bvec_dense (2197)    host
monte_carlo_average (ref:2762)                 device (mfx_monte_carlo_average)
get_PGSE_from_phases (ref:2813)                file parsing / (Delta, delta) mapping on host, phase planes
                                               uploaded once, cosine reduction on device
DT_* / peaks_to_DT_vec (ref:865-1135)          host (orientation input normalisation)
loadmat (ref:3026)                             host (SciPy)
=============================================  ================================================

Nothing in this module falls back to a NumPy evaluation when the device library is missing.
"""
import ctypes as C
import hashlib

import numpy as np

from . import _lib as L
from . import engine

__all__ = ["get_gyromagnetic_ratio", "solve_exhaustive_posweights", "init_PGSE_multishell_interp",
           "interp_PGSE_from_multishell", "rotate_atom", "RotateAtomTables", "import_PGSE_scheme",
           "get_PGSE_scheme_from_bval_bvec_dense", "loadmat", "MultiShellInterpolator"]


def get_gyromagnetic_ratio(element='H'):
    """Gyromagnetic ratio [rad/s/T] (ref:1138-1150)."""
    table = {('hydrogen', 'H', 'proton'): 42.577480e6, ('carbon', 'C'): 10.7084e6, ('phosphorus', 'P'): 17.235e6}
    for names, mhz in table.items():
        if element in names:
            return 2 * np.pi * mhz
    raise ValueError('Gyromagnetic ratio for nucleus of element %s unknown.' % element)


# ---------------------------------------------------------------------------------------------
# knot tables
# ---------------------------------------------------------------------------------------------
class ShellKnots:
    """Knots of one shell's 1-D linear interpolant: ``x`` ascending (P,), ``y`` (P, N).
    Attribute names follow SciPy's interp1d objects, which the reference stores (ref:2074-2080)."""

    def __init__(self, x, y):
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.y = np.ascontiguousarray(y, dtype=np.float64)


def _shell_knots(dots, rows, extra_knot=None):
    """Sorted de-duplicated knots of one shell + left-edge cluster merge.

    dots: |g.ordir| of the shell's samples; rows: their signals (n, N).
    np.unique keeps the first occurrence of exactly repeated abscissae (ref:2048-2053); knots closer
    than 1e-3 to the smallest one are replaced by their centre of mass (ref:2059-2072, 1398-1412).
    extra_knot: optional (x, row) appended when no knot equals x exactly (rotate_atom, ref:1384-1394).
    """
    xu, first = np.unique(dots, return_index=True)
    Y = rows[first, :]
    if extra_knot is not None and not np.any(xu == extra_knot[0]):
        xu = np.append(xu, [extra_knot[0]])
        Y = np.append(Y, np.reshape(extra_knot[1], (1, -1)), axis=0)
    near = np.abs(xu - xu[0]) < 1e-3
    c = int(np.sum(near))
    if c > 1:
        xu = np.append(np.mean(xu[near]), xu[c:])
        Y = np.append(np.mean(Y[near, :], axis=0, keepdims=True), Y[c:, :], axis=0)
    return ShellKnots(xu, Y)


def _check_gnorms(sch, tol=1e-3):
    gn = np.sqrt(np.sum(sch[:, 0:3] ** 2, axis=1))
    if np.any(np.abs(1 - gn[gn > 0]) > tol):
        raise ValueError("Gradient directions in multi-shell scheme matrix should all either have zero or "
                         "unit norm.")


class MultiShellInterpolator(dict):
    """Result of :func:`init_PGSE_multishell_interp`.

    Behaves like the reference's dict (keys ``scheme_DeldelTE``, ``num_subs``, ``Gms_un``,
    ``interpolators``; ref:2081-2085) and additionally owns the HBM copy of the tables."""

    def __init__(self, scheme_DeldelTE, Gms_un, shells, device=0):
        super().__init__()
        self['scheme_DeldelTE'] = np.array(scheme_DeldelTE, dtype=np.float64)
        self['num_subs'] = int(shells[0].y.shape[1])
        self['Gms_un'] = np.ascontiguousarray(Gms_un, dtype=np.float64)
        self['interpolators'] = list(shells)
        self.device = device
        self._tables = None
        self._plans = {}

    # convenience views used by bench/tests
    num_subs = property(lambda self: self['num_subs'])
    Gms_un = property(lambda self: self['Gms_un'])
    S = property(lambda self: len(self['interpolators']))
    off = property(lambda self: np.concatenate([[0], np.cumsum([s.x.size for s in self['interpolators']])])
                   .astype(np.int32))
    x_flat = property(lambda self: np.ascontiguousarray(np.concatenate([s.x for s in self['interpolators']])))
    Y_flat = property(lambda self: np.ascontiguousarray(np.concatenate([s.y for s in self['interpolators']], axis=0)))

    @classmethod
    def from_mapping(cls, m, device=0):
        """Accept the reference's own dict (SciPy interp1d objects expose ``.x`` / ``.y`` too)."""
        if isinstance(m, cls):
            return m
        if m['Gms_un'].size != len(m['interpolators']):
            raise ValueError("msinterp['Gms_un'] has size %d vs expected %d to match "
                             "len(msinterp['interpolators'])" % (m['Gms_un'].size, len(m['interpolators'])))
        shells = [ShellKnots(f.x, np.asarray(f.y).reshape(len(f.x), -1)) for f in m['interpolators']]
        if shells[0].y.shape[1] != m['num_subs']:
            raise ValueError("Inconsistency in msinterp regarding number of substrates. Make sure the "
                             "interpolator was initialized on the right dictionary.")
        return cls(m['scheme_DeldelTE'], m['Gms_un'], shells, device=device)

    def device_tables(self):
        if self._tables is None or self._tables.device != self.device:
            self._tables = engine.DeviceTables([s.x for s in self['interpolators']],
                                               [s.y for s in self['interpolators']], self['Gms_un'],
                                               device=self.device)
            self._plans = {}
        return self._tables

    def plan_for(self, sch_mat):
        """Device row plan of a protocol (cached per distinct scheme matrix)."""
        sch = L.f64c(sch_mat)
        key = (sch.shape, hashlib.sha1(sch.tobytes()).hexdigest(), self.device)
        p = self._plans.get(key)
        if p is None:
            if len(self._plans) > 8:
                self._plans.clear()
            p = engine.Plan(self.device_tables(), scheme=sch)
            self._plans[key] = p
        return p

    # --- flat (de)serialisation, used to broadcast the dictionary over RCCL
    def pack(self):
        hdr = {"sizes": [int(s.x.size) for s in self['interpolators']], "N": self['num_subs'],
               "S": len(self['interpolators'])}
        flat = np.concatenate([self['scheme_DeldelTE'], self['Gms_un'], self.x_flat, self.Y_flat.reshape(-1)])
        return hdr, np.ascontiguousarray(flat, dtype=np.float64)

    @classmethod
    def unpack(cls, hdr, flat, device=0):
        S, N, sizes = hdr["S"], hdr["N"], hdr["sizes"]
        P = int(sum(sizes))
        tim, flat = flat[:3], flat[3:]
        G, flat = flat[:S], flat[S:]
        x, Y = flat[:P], flat[P:P + P * N].reshape(P, N)
        shells, o = [], 0
        for n in sizes:
            shells.append(ShellKnots(x[o:o + n], Y[o:o + n]))
            o += n
        return cls(tim, G, shells, device=device)


def init_PGSE_multishell_interp(sig_ms, sch_mat_ms, ordir, device=0):
    """Per-shell interpolation tables of a dense multi-shell dictionary (ref:1959-2085).

    Returns a :class:`MultiShellInterpolator` (a dict with the reference's keys)."""
    ordir = np.asarray(ordir, dtype=np.float64)
    if ordir.size != 3:
        raise ValueError("Direction of dictionary computed with dense sampling (ordir) should have 3 entries.")
    ordir = np.squeeze(ordir) if ordir.ndim > 1 else ordir
    sch_mat_ms = np.asarray(sch_mat_ms, dtype=np.float64)
    if not np.all(np.isclose(sch_mat_ms[0, 4:7], sch_mat_ms[:, 4:7])):
        raise ValueError("Delta, delta and TE values should all be identical in multi-shell sampling.")
    sig_ms = np.asarray(sig_ms, dtype=np.float64)
    if sig_ms.ndim == 1:
        sig_ms = sig_ms.reshape((sig_ms.size, 1))
    nrm = np.sqrt((ordir ** 2).sum())
    if np.abs(1 - nrm) > 1e-3:
        raise ValueError("Orientation vector of the multi-shell signal must have unit norm. Detected %g." % nrm)
    _check_gnorms(sch_mat_ms)
    dots = np.abs(np.dot(sch_mat_ms[:, 0:3], ordir))
    G_un, which = np.unique(sch_mat_ms[:, 3], return_inverse=True)
    shells = []
    for s, G in enumerate(G_un):
        rows = np.where(which == s)[0]
        if G == 0:
            # b0 shell: constant interpolant through the first b0 row (ref:2019-2046)
            same = np.all(np.isclose(sig_ms[rows, :], sig_ms[rows[0], :]), axis=0)
            if np.any(~same):
                bad = np.where(~same)[0]
                raise ValueError('Distinct signal values in provided multi-shell sampling for zero gradients '
                                 '(b0 acquistions), for %d substrate(s) [%s]'
                                 % (bad.shape[0], " ".join("{:d}".format(b) for b in bad)))
            shells.append(ShellKnots([0.0, 1.0], np.repeat([sig_ms[rows[0], :]], 2, axis=0)))
        else:
            shells.append(_shell_knots(dots[rows], sig_ms[rows, :]))
    return MultiShellInterpolator(sch_mat_ms[0, 4:7], G_un, shells, device=device)


def interp_PGSE_from_multishell(sch_mat, newdir, sig_ms=None, sch_mat_ms=None, ordir=None, msinterp=None):
    """Single-fascicle PGSE signals rotated to ``newdir`` and resampled on ``sch_mat`` (ref:1693-1956).

    Returns ``np.squeeze`` of the (Nseq, Nsub) array, like the reference."""
    if msinterp is None:
        if sig_ms is None or sch_mat_ms is None or ordir is None:
            raise ValueError("If msinterp is not specified, sig_ms, sch_mat_ms and ordir must all be specified.")
        if np.asarray(sch_mat_ms).shape[0] != np.asarray(sig_ms).shape[0]:
            raise ValueError("Number of lines in dense multishell scheme (%d) does not match number of signal "
                             "values per substrate (%d)." % (np.asarray(sch_mat_ms).shape[0],
                                                             np.asarray(sig_ms).shape[0]))
        ms = init_PGSE_multishell_interp(sig_ms, sch_mat_ms, ordir)
    else:
        ms = MultiShellInterpolator.from_mapping(msinterp)
    sch_mat = np.asarray(sch_mat, dtype=np.float64)
    if not np.all(np.isclose(ms['scheme_DeldelTE'], sch_mat[:, 4:7])):          # ref:1786-1789
        raise ValueError("Delta, delta and TE values should all be identical to those in the multi-shell "
                         "sampling.")
    newdir = np.asarray(newdir, dtype=np.float64)
    if newdir.size != 3:
        raise ValueError("Direction of fascicle for new signal (newdir) should have 3 entries.")
    newdir = np.ascontiguousarray(newdir.reshape(3))
    nrm = np.sqrt((newdir ** 2).sum())
    if np.abs(1 - nrm) > 1e-3:                                                   # ref:1798-1802
        raise ValueError("Orientation vector of the new signal must have unit norm. Detected %g." % nrm)
    _check_gnorms(sch_mat)                                                       # ref:1804-1807
    plan = ms.plan_for(sch_mat)               # raises ValueError outside the table's G range (ref:1829-1836)
    out = np.zeros((sch_mat.shape[0], ms['num_subs']))
    L.check(L.lib().mfx_rotate(plan.handle(), L.dptr(newdir), 1, 0, L.dptr(out)))
    return np.squeeze(out)


class RotateAtomTables:
    """Direction-independent part of :func:`rotate_atom` (ref:1233-1412), hoisted out of the call:
    per-(G, Delta, delta) shell knots incl. the free-diffusion knot at |g.n| = 1, resident in HBM.
    ``rotate(newdirs)`` evaluates B directions in one device call."""

    def __init__(self, sig, sch_mat, ordir, DIFF, S0, warnings=True, device=0):
        assert isinstance(sig, np.ndarray), "Input sig should be a NumPy ndarray"
        assert isinstance(sch_mat, np.ndarray), "Input sch_mat should be a NumPy ndarray"
        assert isinstance(ordir, np.ndarray), "Input ordir should be a NumPy ndarray"
        assert isinstance(S0, np.ndarray), "Input S0 should be a NumPy ndarray"
        self.sig_shape = sig.shape
        Dv = np.asarray(DIFF, dtype=np.float64).reshape(-1)   # scalar, or one value per substrate
        Dv = Dv[0] if Dv.size == 1 else Dv
        sig = np.asarray(sig, dtype=np.float64)
        S0 = np.asarray(S0, dtype=np.float64)
        if sig.ndim == 1:
            sig = sig.reshape((sig.size, 1))
        if S0.ndim == 1:
            S0 = S0[:, np.newaxis]
        if sch_mat.shape[1] < 6:
            raise ValueError('sch_mat must be a N-by-6 or7 matrix')
        if sch_mat.shape[0] != sig.shape[0]:
            raise ValueError('sch_mat and sig must have the same number of rows')
        assert sig.shape == S0.shape, "The S0 matrix should have the same size as the signal matrix"
        sch = np.asarray(sch_mat, dtype=np.float64)
        M = sch.shape[0]
        gn = np.sqrt((sch[:, 0:3] ** 2).sum(axis=1, keepdims=True))
        gn[gn == 0] = np.inf
        ghat = np.ascontiguousarray(sch[:, 0:3] / gn)                 # b0 rows -> zero vector -> |g.n| = 0
        odir = np.asarray(ordir, dtype=np.float64).reshape(3)
        dots = np.abs(np.dot(ghat, odir / np.sqrt((odir ** 2).sum())))
        gam = get_gyromagnetic_ratio('H')
        bvals = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
        trip, which = np.unique(sch[:, 3:6], return_inverse=True, axis=0)
        which = np.asarray(which).reshape(-1)
        shells, shell_of_row = [], np.zeros(M, dtype=np.int32)
        for s in range(trip.shape[0]):
            rows = np.where(which == s)[0]
            b = bvals[rows[0]]
            if b == 0:
                # no rotation for b0 rows (ref:1298-1300): one constant 2-knot shell per row
                for m in rows:
                    shell_of_row[m] = len(shells)
                    shells.append(ShellKnots([0.0, 1.0], np.repeat(sig[m:m + 1, :], 2, axis=0)))
                continue
            if rows.size < 2:
                raise ValueError("Fewer than 2 identical (G, Del, del) triplets detected for triplet %d/%d "
                                 "(%g, %g, %g), b=%g s/mm^2, probably not a HARDI shell."
                                 % (s + 1, trip.shape[0], trip[s, 0], trip[s, 1], trip[s, 2], b / 1e6))
            if rows.size < 10 and warnings:
                print("WARNING: rotate_atom: fewer than 10 data points detected for acquisition parameters "
                      "(G, Del, del) %d/%d (%g, %g, %g), b=%g s/mm^2.\nQuality of approximation may be poor."
                      % (s + 1, trip.shape[0], trip[s, 0], trip[s, 1], trip[s, 2], b / 1e6))
            ok = np.all(np.isclose(S0[rows, :], S0[rows[0], :]), axis=0)
            if np.any(~ok):
                bad = np.where(~ok)[0]
                raise ValueError('Distinct values in provided S0 image for shell  %d/%d (b=%g s/mm^2) for %d '
                                 'substrate(s) [%s]' % (s + 1, trip.shape[0], b / 1e6, bad.shape[0],
                                                        " ".join("{:d}".format(x) for x in bad)))
            free = np.exp(-b * Dv) * S0[rows[0], :]
            shell_of_row[rows] = len(shells)
            shells.append(_shell_knots(dots[rows], sig[rows, :], extra_knot=(1.0, free)))
        self.N = sig.shape[1]
        self.M = M
        self.tables = engine.DeviceTables([k.x for k in shells], [k.y for k in shells],
                                          np.arange(len(shells), dtype=np.float64), device=device)
        self.plan = engine.Plan(self.tables, gdirs=ghat, shell_of_row=shell_of_row)

    def rotate(self, newdirs):
        d = L.f64c(np.asarray(newdirs, dtype=np.float64).reshape(-1, 3))
        out = np.zeros((d.shape[0], self.M, self.N))
        L.check(L.lib().mfx_rotate(self.plan.handle(), L.dptr(d), d.shape[0], 1, L.dptr(out)))
        return out


def rotate_atom(sig, sch_mat, ordir, newdir, DIFF, S0, warnings=True):
    """Rotate HARDI signals of single fascicles from ``ordir`` to ``newdir`` (ref:1205-1437)."""
    assert isinstance(newdir, np.ndarray), "Input newdir should be a NumPy ndarray"
    T = RotateAtomTables(sig, sch_mat, ordir, DIFF, S0, warnings=warnings)
    out = T.rotate(np.asarray(newdir, dtype=np.float64).reshape(1, 3))[0]
    if np.any(np.isnan(out)):                                                    # ref:1428-1436
        bad = np.where(np.any(np.isnan(out), axis=0))[0]
        raise ValueError('Nan detected after rotation of substrate(s) for %d substrate(s): [%s]'
                         % (bad.shape[0], " ".join("%d" % b for b in bad)))
    return np.reshape(out, T.sig_shape)


# ---------------------------------------------------------------------------------------------
# solver
# ---------------------------------------------------------------------------------------------
def solve_exhaustive_posweights(A, y, dicsizes, printmsg=None):
    """Combinatorial NNLS with exactly one atom per sub-dictionary (ref:115-214).

    Returns ``(w_nneg, ind_atoms_subdic, ind_atoms_totdic, min_obj, y_recons)``."""
    if printmsg is not None:
        print(printmsg, end="")
    # input checks: same conditions and AssertionError as ref:157-188
    assert isinstance(A, np.ndarray), "A should be a NumPy ndarray"
    assert A.ndim == 2, "A should be a 2D array"
    assert not np.any(np.all(A == 0, axis=0)), "All-zero columns detected in A"
    assert isinstance(y, np.ndarray), "y should be a NumPy ndarray"
    assert A.size > 0 and y.size > 0, "A and y should not be empty arrays"
    assert A.shape[0] == y.size, ("Number of rows in A (%d) should match number of elements in y (%d)"
                                  % (A.shape[0], y.size))
    assert isinstance(dicsizes, np.ndarray), "dicsizes should be a NumPy ndarray"
    assert np.all(dicsizes > 0), "All entries of dicsizes should be > 0"
    assert A.shape[1] == np.sum(dicsizes), ("Number of columns of A (%d) does not equal sum of size of "
                                            "sub-matrices in diclengths array (%d)"
                                            % (A.shape[1], np.sum(dicsizes)))
    A64 = L.f64c(A)
    y64 = L.f64c(y).reshape(-1)
    sizes = np.ascontiguousarray(dicsizes, dtype=np.int64).reshape(-1)
    Kp = sizes.size
    w = np.zeros(Kp)
    sub = np.zeros(Kp, dtype=np.int64)
    tot = np.zeros(Kp, dtype=np.int64)
    obj = np.zeros(1)
    yrec = np.zeros(A64.shape[0])
    L.check(L.lib().mfx_solve_exhaustive(L.dptr(A64), A64.shape[1], A64.shape[0], L.lptr(sizes), Kp, L.dptr(y64),
                                         L.dptr(w), L.lptr(sub), L.lptr(tot), L.dptr(obj), L.dptr(yrec)))
    if Kp <= 3:
        # the reference's Numba kernels return int32 index arrays (ref:218-224, 284-286, 466-468)
        sub, tot = sub.astype(np.int32), tot.astype(np.int32)
    return w, sub, tot, float(obj[0]), yrec


# ---------------------------------------------------------------------------------------------
# protocol helpers (input normalisation before the voxel loop)
# ---------------------------------------------------------------------------------------------
def import_PGSE_scheme(scheme):
    """Load / validate a PGSE scheme ``[gx gy gz G Delta delta TE]`` (ref:2128-2192)."""
    if isinstance(scheme, str):
        with open(scheme, 'r') as f:
            skip = 1 if 'version' in f.readline().lower() else 0
        sch = np.loadtxt(scheme, skiprows=skip)
    elif isinstance(scheme, np.ndarray):
        sch = scheme
    else:
        raise TypeError("Unable to import a PGSE scheme matrix from input")
    if sch.ndim == 1:
        sch = sch[np.newaxis, :]
    if sch.shape[1] != 7:
        raise RuntimeError("Detected %s instead of expected 7 colums in PGSE scheme matrix." % sch.shape[1])
    gn = np.sqrt(np.sum(sch[:, :3] ** 2, axis=1))
    nbad = np.sum(np.abs(1 - gn[gn > 0]) > 1e-4)
    if nbad > 0:
        raise ValueError("Detected %d non-zero gradients which did not have unit norm. Please normalize." % nbad)
    G, Dl, dl, TE = sch[:, 3], sch[:, 4], sch[:, 5], sch[:, 6]
    for arr, what in ((G, 'gradient intensity (4th column)'), (Dl, 'gradient separation Delta (5th column)'),
                      (dl, 'gradient duration delta (6th column)'), (TE, 'echo time TE (7th column)')):
        if np.any(arr < 0):
            raise ValueError('Detected %d sequence(s) with negative %s.' % (np.sum(arr < 0), what))
    if np.any(dl > Dl):
        raise ValueError('Detected %d sequence(s) in which delta (6th column) was greater than Delta '
                         '(5th column).' % np.sum(dl > Dl))
    if np.any(TE < (Dl + dl) * 0.999):
        raise ValueError('Detected %d sequence(s) in which TE (7th column) was lower than Delta+delta.'
                         % np.sum(TE < (Dl + dl)))
    return sch


def _as_text_table(x):
    """A b-value / b-vector argument: a path to a whitespace-separated text file, or array-like."""
    return np.loadtxt(x) if isinstance(x, str) else np.asarray(x)


def _gradient_rows(bvecs, n):
    """b-vectors as an [n x 3] array of unit rows (zero rows stay zero).  Accepts 3 x n (FSL) and n x 3; a 3 x 3 input is
    read as 3 x n like the reference does (its test on shape[0] comes first, ref:2253-2259)."""
    if bvecs.shape[0] == 3:
        g = np.array(bvecs.T, dtype=np.float64)
    elif bvecs.shape[1] == 3:
        g = np.array(bvecs, dtype=np.float64)
    else:
        raise ValueError("Vectors in bvecs should be 3-dimensional. However, detected no dimension with size 3.")
    length = np.sqrt(np.sum(g ** 2, axis=1))
    nz = length > 0
    g[nz] = g[nz] / length[nz][:, np.newaxis]
    return g


def _snap_to_shells(G, shells, Gtol):
    """Each gradient intensity in G replaced by the dense scheme's shell value within Gtol of it; the number of
    (value, shell) matches must equal G.size - a value near no shell, or near two, is the reference's mapping error.
    Where two shells match the larger one would win (the reference assigns shell after shell in ascending order)."""
    near = np.abs(G[:, np.newaxis] - shells[np.newaxis, :]) < Gtol         # [n, S]
    n_matches = int(np.count_nonzero(near))
    last_match = near.shape[1] - 1 - np.argmax(near[:, ::-1], axis=1)
    return np.where(near.any(axis=1), shells[last_match], 0.0), n_matches


def get_PGSE_scheme_from_bval_bvec_dense(sch_mat_dense, bvals, bvecs, Gtol=1e-3):
    """Subject protocol [gx gy gz G Delta delta TE] from b-values [s/mm^2] and b-vectors, for a dictionary simulated on the
    dense multi-shell scheme ``sch_mat_dense``: timing (Delta, delta, TE) is the dense scheme's - which must be a single
    one - and every b-value is turned into a gradient intensity and snapped onto the dense scheme's shell within ``Gtol``
    [T/m] of it (reference mf_utils.py:2197-2300; same ValueErrors)."""
    dense = import_PGSE_scheme(sch_mat_dense)
    b_si = np.asarray(_as_text_table(bvals), dtype=np.float64).ravel() * 1e6          # s/mm^2 -> s/m^2
    vec = _as_text_table(bvecs)
    if isinstance(bvecs, str):
        vec = np.atleast_2d(vec)
    n = b_si.size
    if np.ndim(vec) != 2:
        raise ValueError("bvecs array should have 2 dimensions, detected %d." % np.ndim(vec))
    if n not in vec.shape:
        raise ValueError("Number of b-vectors does not match number of b-values (%d)" % n)
    timing = dense[:, 4:6]
    if not np.all(timing == timing[0]):
        raise ValueError('Detected different pairs of (Delta, delta) values in reference scheme matrix '
                         '(note that zeros count as values), which is currently not supported.')
    Delta, delta, TE = dense[0, 4:7]
    G = np.sqrt(b_si / (Delta - delta / 3)) / (get_gyromagnetic_ratio('H') * delta)     # b = (gamma G delta)^2 (Delta - delta/3)
    G_shell, n_matches = _snap_to_shells(G, np.unique(dense[:, 3]), Gtol)
    g = _gradient_rows(vec, n)
    if n_matches != n:
        raise ValueError('Mismatch between reference scheme matrix and bvals.  Could only map %d/%d b-values '
                         '(equivalently, gradient intensities G) from the specified bvals to the b-values '
                         'contained in the reference scheme matrix. You may want to change the tolerance on '
                         'gradient intensity G (currently %g T/m).' % (n_matches, n, Gtol))
    return np.column_stack([g, G_shell, np.full(n, Delta), np.full(n, delta), np.full(n, TE)])


# ---------------------------------------------------------------------------------------------
# diffusion-tensor <-> peak helpers (orientation inputs of MFModel.fit / cleanup_2fascicles)
# ---------------------------------------------------------------------------------------------
# position of each upper-triangle element (row, col) in the 6-vector, per storage convention
_DT_ORDER = {
    'row':      ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)),     # xx xy xz yy yz zz  (.nrrd)
    'column':   ((0, 0), (0, 1), (1, 1), (0, 2), (1, 2), (2, 2)),     # xx xy yy xz yz zz  (NIfTI)
    'diagonal': ((0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (0, 2)),     # xx yy zz xy yz xz
}


def _dt_order(order):
    try:
        return _DT_ORDER[order]
    except (KeyError, TypeError):
        raise ValueError('Unknown order "%s".' % (order,))


def DT_array_to_vec(DT, order='row'):
    """(..., 3, 3) symmetric tensors -> (..., 6) vectors in the given element order (ref:865-898)."""
    DT = np.asarray(DT)
    if DT.ndim < 2:
        raise ValueError('DT should have at least 2 dimensions. Detected %d.' % DT.ndim)
    if DT.shape[-2:] != (3, 3):
        raise ValueError('Last 2 dimensions of DT should be (3, 3). Detected (%d, %d).' % DT.shape[-2:])
    pos = _dt_order(order)
    return np.stack([DT[..., r, c] for (r, c) in pos], axis=-1)


def DT_vec_to_2Darray(DT_vec, order):
    """(..., 6) vectors -> (..., 3, 3) symmetric tensors (ref:901-957)."""
    DT_vec = np.asarray(DT_vec)
    if DT_vec.shape[-1] != 6:
        raise ValueError("Last dimension of input should have size 6, detected %d." % DT_vec.shape[-1])
    pos = _dt_order(order)
    out = np.zeros(DT_vec.shape[:-1] + (3, 3))
    for e, (r, c) in enumerate(pos):
        out[..., r, c] = DT_vec[..., e]
        out[..., c, r] = DT_vec[..., e]
    return out


def DT_vec_to_peaks(DT_vec, order, mask=None):
    """Principal direction of every tensor in a (..., 6) array: the unit eigenvector of the largest eigenvalue, a zero
    vector where that eigenvalue is zero (an all-zero tensor) and outside ``mask`` (reference mf_utils.py:960-1019).
    A single 6-vector gives a single 3-vector."""
    tensors = np.asarray(DT_vec)
    single = tensors.ndim == 1
    if single:
        tensors = tensors[np.newaxis, :]
    if tensors.shape[-1] != 6:
        raise ValueError('DT_vec should have size 6 along last dimension. Detected %d.' % (tensors.shape[-1],))
    grid = tensors.shape[:-1]
    inside = np.ones(grid, dtype=bool) if mask is None else np.asarray(mask)
    if inside.ndim != len(grid):
        raise ValueError('mask should have %d dimension(s) since DT_vec has %d, detected %d instead.'
                         % (len(grid), len(grid) + 1, inside.ndim))
    inside = inside > 0
    peaks = np.zeros(grid + (3,))
    if np.any(inside):
        evals, evecs = np.linalg.eigh(DT_vec_to_2Darray(tensors[inside], order))      # eigenvalues ascending
        principal = evecs[:, :, 2]
        principal[np.abs(evals[:, 2]) == 0] = 0.0                # eigh hands back the identity for a zero tensor
        peaks[inside] = principal
    return peaks[0] if single and peaks.shape[0] == 1 else (np.squeeze(peaks) if single else peaks)


def peaks_to_DT_vec(peaks, order, lam_par=2e-3, lam_perp=0.1e-3):
    """Stick-like tensors ``lam_par v v' + lam_perp (I - v v')`` for display (ref:1022-1135).

    The reference draws a random perpendicular pair (p1, p2) and sums ``lam_perp (p1 p1' + p2 p2')``;
    for any orthonormal completion that sum is ``lam_perp (I - v v')``, which is what is formed here
    (same tensor up to rounding, and deterministic).  Returns a list with one (..., 6) array per
    peak; like the reference, non-zero input peaks are normalised in place."""
    if peaks.ndim < 2:
        raise ValueError('peaks array should have at least 2 dimensions. Detected %d.' % peaks.ndim)
    if peaks.shape[-1] != 3:
        raise ValueError('Last dimension of peaks should have size 3, detected %d.' % (peaks.shape[-1]))
    if lam_par < lam_perp:
        raise ValueError('Parallel diffusivity should be greater than or equal to perpendicular diffusivity.')
    pos = _dt_order(order)
    nrm = np.sqrt(np.sum(peaks ** 2, axis=-1))
    nz = nrm > 0
    peaks[nz, :] = peaks[nz, :] / nrm[nz][:, np.newaxis]
    v = peaks[nz, :]
    DT = (lam_par - lam_perp) * v[:, :, np.newaxis] * v[:, np.newaxis, :] + lam_perp * np.eye(3)
    tens = np.zeros(peaks.shape[:-1] + (6,))
    tens[nz, :] = np.stack([DT[:, r, c] for (r, c) in pos], axis=-1)
    return [tens[..., k, :] for k in range(peaks.shape[-2])]


# ---------------------------------------------------------------------------------------------
# Monte-Carlo signal synthesis from stored spin phases (dictionary generation)
# ---------------------------------------------------------------------------------------------
def monte_carlo_average(sim_phases, delta_mapping, gscaling, Dscaling, num_spins, device=0):
    """``S_i = mean_l cos(Dscaling * sum_n gscaling[i,n] * sim_phases[delta_mapping[i]*num_spins + l, n])``
    (ref:2758-2810), evaluated on the GPU.  Per-term arithmetic follows the reference order; the
    spins are summed in a fixed tree order instead of sequentially."""
    ph = L.f64c(sim_phases)
    if ph.ndim != 2:
        raise ValueError("sim_phases should have 2 dimensions (n_spin*n_ref, n_dim), detected %d." % ph.ndim)
    dm = np.ascontiguousarray(delta_mapping, dtype=np.int64).reshape(-1)
    gs = L.f64c(gscaling)
    if gs.ndim != 2 or gs.shape[0] != dm.size or gs.shape[1] != ph.shape[1]:
        raise ValueError("gscaling should have shape (%d, %d), got %s." % (dm.size, ph.shape[1], gs.shape))
    out = np.zeros(dm.size)
    L.check(L.lib().mfx_monte_carlo_average(L.dptr(ph), ph.shape[0], ph.shape[1], L.lptr(dm), L.dptr(gs),
                                            float(Dscaling), int(num_spins), dm.size, L.dptr(out), int(device)))
    return out


def _phase_file_format(phasefile):
    """('<'|'>', 'f4'|'f8', bytes per item, directory, basename, extension) from the file name
    (ref:2904-2934): extension = endianness letter (b/l) + storage type (single|float|double)."""
    import os
    folder, tail = os.path.split(phasefile)
    base, ext = os.path.splitext(tail)
    if not ext:
        raise ValueError("Phase file extension not found.\nAborting as there is no way to tell which level of "
                         "precision was used to store the phase values (e.g., float, double, ...).")
    endian = {'b': '>', 'l': '<'}.get(ext[1].lower())
    if endian is None:
        raise ValueError("Phase file extension (after the dot) should start with a b for big endian or with a l "
                         "for little endian. Detected: \"%s\"." % ext[1])
    if ext[2:] in ('single', 'float'):
        kind, width = 'f4', 4
    elif ext[2:] == 'double':
        kind, width = 'f8', 8
    else:
        raise ValueError("Data type of phase file specified in file extension (\"%s\") not supported." % ext[2:])
    return endian, kind, width, folder, base, ext


def get_PGSE_from_phases(phasefile, sch_mat_sim, sch_mat, dim=None, D_sim=None, D=None, device=0):
    """PGSE signal for protocol ``sch_mat`` from the spins' phases of a reference Monte-Carlo run
    (ref:2813-3015; same arguments, checks and messages).  The phase files (one per gradient
    component, ``*_phase_x|y|z.<b|l><single|float|double>``) are staged to HBM as one plane per
    component -- no host-side interleaving -- and reduced there."""
    import os
    import torch
    names = ['x', 'y', 'z']
    MAXDIM = 3
    D_ratio_sqrt = 1.0
    if D is not None:
        if D_sim is None:
            raise NameError("Simulation diffusivity should be specified if new signal diffusivity is set.")
        D_ratio_sqrt = float(np.sqrt(D / D_sim))
    if dim is None:
        dim = MAXDIM
    elif dim > MAXDIM:
        raise ValueError("dim should be less than or equal to %d." % MAXDIM)
    sch_sim = import_PGSE_scheme(sch_mat_sim)
    sch = import_PGSE_scheme(sch_mat)
    if np.any(sch[:, dim:MAXDIM] != 0):
        print("WARNING get_PGSE_from_phases: detected non-zero entries in gradient components after dimension %d.\n"
              "Those components will be ignored but make sure the right acquisition protocol was provided.\n"
              "It is common for such protocols to contain zeros in those gradient components, for instance after "
              "projection into the xy-plane of a 3D protocol.\n" % dim)
    num_seq, num_ref = sch.shape[0], sch_sim.shape[0]
    g_sim = sch_sim[:, :3] * sch_sim[:, 3][:, np.newaxis]
    g_new = sch[:, :3] * sch[:, 3][:, np.newaxis]
    # each new sequence -> LAST simulated sequence with the same (Delta, delta) (ref:2874-2880)
    delta_mapping = np.full(num_seq, -1, dtype=np.int64)
    for i in range(num_ref):
        delta_mapping[np.all(sch[:, 4:6] == sch_sim[i, 4:6], axis=1)] = i
    bad = np.where(delta_mapping < 0)[0]
    if bad.size > 0:
        listing = '\n'.join('\t%4d -- %5g -- %5g' % (b, sch[b, 4] * 1e3, sch[b, 5] * 1e3) for b in bad)
        raise ValueError('Acquisition protocol contains %d (Delta,delta) pair(s) (out of %d) not used to simulate the '
                         'directional phases in the Monte Carlo simulation. List of unmatched sequences:\nSequ. no. '
                         '-- Delta [ms] -- delta [ms]\n%s' % (bad.size, num_seq, listing))
    with np.errstate(divide='ignore', invalid='ignore'):
        gscaling = np.ascontiguousarray(g_new[:, :dim] / g_sim[delta_mapping, :dim])
    if not os.path.isfile(phasefile):
        raise RuntimeError("File %s does not exist." % phasefile)
    nbytes = os.path.getsize(phasefile)
    endian, kind, width, folder, base, ext = _phase_file_format(phasefile)
    if nbytes % (num_ref * width) != 0:
        raise RuntimeError("Phase file %s is either corrupted or inconsistently named. Storage precision of items "
                           "(%d bytes) times number of reference simulation sequences (%d) does not divide total "
                           "size (%d bytes)." % (phasefile, width, num_ref, nbytes))
    num_entries = nbytes // width
    num_spins = num_entries // num_ref
    lib = L.lib()
    if lib.mfx_device_count() <= 0:
        raise L.MfxError("no HIP device available (this library has no CPU path)")
    dev = torch.device("cuda", int(device))
    planes = torch.empty((dim, num_entries), dtype=torch.float64, device=dev)     # one plane per component
    for i in range(dim):
        f_i = os.path.join(folder, base[:-len(names[i])] + names[i] + ext)
        if not os.path.isfile(f_i):
            raise RuntimeError("Phase file %s not found." % f_i)
        raw = np.fromfile(f_i, dtype=endian + kind, count=num_entries, sep="")
        planes[i].copy_(torch.from_numpy(raw.astype(np.float64)))
    out = np.zeros(num_seq)
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev)
        L.check(lib.mfx_monte_carlo_average_dev(planes.data_ptr(), num_entries, 1, num_entries, dim, L.lptr(delta_mapping),
                                                L.dptr(gscaling), D_ratio_sqrt, num_spins, num_seq, L.dptr(out),
                                                st.cuda_stream))
    return out


def loadmat(filename):
    """``scipy.io.loadmat`` with MATLAB structs converted to nested dicts (ref:3026-3087)."""
    import scipy.io
    try:
        from scipy.io.matlab import mat_struct
    except ImportError:  # older SciPy
        from scipy.io.matlab.mio5_params import mat_struct

    def conv(o):
        if isinstance(o, mat_struct):
            return {k: conv(v) for k, v in o.__dict__.items() if k != '_fieldnames'}
        return o

    data = scipy.io.loadmat(filename, struct_as_record=False, squeeze_me=True)
    return {k: conv(v) for k, v in data.items()}
