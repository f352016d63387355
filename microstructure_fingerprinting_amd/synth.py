"""Seeded synthetic protocols / dictionaries / voxels (SURVEY.md section 8d recipe).

Used by bench.py and the tests to build inputs of BASELINE.json's shapes; there is no
network, so no real dictionary can be downloaded.
"""
import numpy as np

GAMMA_H = 2 * np.pi * 42.577480e6  # proton gyromagnetic ratio (reference mf_utils.py:1142)


def unit_vectors(rng, n):
    v = rng.standard_normal((n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def make_scheme(rng, n_b0, shells_b, dirs_per_shell, Delta=43.1e-3, delta=10.6e-3, TE=92e-3):
    """PGSE scheme rows [gx gy gz G Delta delta TE] (SI units); b in s/mm^2."""
    rows = [[0, 0, 0, 0.0, Delta, delta, TE] for _ in range(n_b0)]
    for b, nd in zip(shells_b, dirs_per_shell):
        G = np.sqrt(b * 1e6 / (Delta - delta / 3)) / (GAMMA_H * delta)
        g = unit_vectors(rng, nd)
        rows += [[g[i, 0], g[i, 1], g[i, 2], G, Delta, delta, TE] for i in range(nd)]
    return np.array(rows)


def make_dictionary(rng, sch, N):
    """Smooth, positive, distinct single-fascicle signals for a fascicle along z."""
    G, Dl, dl = sch[:, 3], sch[:, 4], sch[:, 5]
    b = ((GAMMA_H * G * dl) ** 2 * (Dl - dl / 3))[:, None]
    u2 = (sch[:, 2] ** 2)[:, None]
    f = rng.uniform(0.3, 0.9, N)
    dpar = rng.uniform(1.5e-9, 2.5e-9, N)
    dperp = rng.uniform(0.1e-9, 0.8e-9, N)
    Diso = rng.uniform(0.5e-9, 1.5e-9, N)
    s0 = rng.uniform(0.5, 1.0, N)
    sig = s0 * (f * np.exp(-b * dpar * u2) * np.exp(-b * dperp * (1 - u2)) + (1 - f) * np.exp(-b * Diso))
    b0 = np.where(G == 0)[0]
    if b0.size:
        sig[b0, :] = sig[b0[0], :]
    return sig


def config(name):
    """(n_b0, shells_b, dirs_per_shell, N, seed) for BASELINE.json configs."""
    if name == "C1":   # 1k voxels, 1 fascicle, 100 atoms x 60 measurements
        return dict(n_b0=2, shells_b=[1000, 2000], dirs=[29, 29], N=100, seed=0, K=1)
    if name == "C2":   # 1e5 voxels, 2 fascicles, 782 atoms x 200 measurements
        return dict(n_b0=2, shells_b=[1000, 2000, 3000], dirs=[66, 66, 66], N=782, seed=1, K=2)
    raise KeyError(name)


def make_model(name, N=None):
    c = config(name)
    rng = np.random.default_rng(c["seed"])
    sch = make_scheme(rng, c["n_b0"], c["shells_b"], c["dirs"])
    dic = make_dictionary(rng, sch, N or c["N"])
    return sch, dic, rng


def make_voxels(rng, V, K, rotate, N, M0=500.0, snr=30.0):
    """peaks [V,3K], Y [V,M]: y = M0 * sum_k nu_k D_k[:, a_k] + N(0, M0/snr).
    `rotate(dirs[B,3]) -> [B,M,N]` supplies rotated dictionaries (device-backed in the product)."""
    peaks = np.concatenate([unit_vectors(rng, V) for _ in range(K)], axis=1)
    atoms = rng.integers(0, N, (V, K))
    nu = rng.dirichlet(np.ones(K), V)
    Y = None
    for k in range(K):
        Dk = rotate(peaks[:, 3 * k:3 * k + 3])             # [V, M, N]
        col = np.take_along_axis(Dk, atoms[:, k][:, None, None], axis=2)[:, :, 0]
        Y = (0 if Y is None else Y) + M0 * nu[:, k:k + 1] * col
    Y = Y + rng.normal(0, M0 / snr, Y.shape)
    return peaks, Y, atoms, nu
