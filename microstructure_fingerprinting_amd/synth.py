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


def make_phantom(model, grid, rng, k_frac=(0.1, 0.3, 0.6), p_csf=0.3, p_ear=0.1, M0=500.0, snr=30.0, device=0):
    """A brain-sized test volume for MFModel.fit: ellipsoidal mask inside ``grid``; per ROI voxel a number of fascicles
    drawn with probabilities ``k_frac`` (0, 1, 2), CSF / EAR flags with probabilities p_csf / p_ear, random directions,
    y = M0 * sum nu_c * component_c + noise.  Signals are generated on the device (rotated atoms through the model's
    plan) and returned as float32 in file order (Fortran-contiguous [grid x M], the layout of a NIfTI file).
    Returns dict(data, mask, numfasc, peaks, csf_mask, ear_mask) of arrays over ``grid``."""
    import torch
    from . import engine
    from . import mf_utils as mfu
    sch = np.ascontiguousarray(model.dic['sch_mat'], dtype=np.float64)
    plan = model.ms_interpolator.plan_for(sch)
    M, N = sch.shape[0], int(model.dic['num_atom'])
    ax = [(np.arange(n) + 0.5) / n * 2 - 1 for n in grid]
    r2 = ax[0][:, None, None] ** 2 + ax[1][None, :, None] ** 2 + ax[2][None, None, :] ** 2
    mask = (r2 <= 1.0).astype(np.float64)
    roi = np.flatnonzero(mask.reshape(-1))
    V = roi.shape[0]
    K = rng.choice(3, V, p=np.asarray(k_frac) / np.sum(k_frac))
    csf = rng.random(V) < p_csf
    ear = rng.random(V) < p_ear
    pk = np.concatenate([unit_vectors(rng, V), unit_vectors(rng, V)], axis=1)
    pk[K < 2, 3:] = 0
    pk[K < 1, :3] = 0
    nu = rng.dirichlet(np.ones(4), V) * np.stack([K >= 1, K >= 2, csf, ear], axis=1)
    nu /= np.maximum(nu.sum(axis=1, keepdims=True), 1e-300)
    gam = mfu.get_gyromagnetic_ratio('H')
    b = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / model.dic['T2_csf']) * np.exp(-b * model.dic['DIFF_csf'])
    D_ear = np.atleast_1d(model.dic['DIFF_ear'])
    sig_ear = np.exp(-sch[:, 6] / model.dic['T2_ear'])[:, None] * np.exp(-b[:, None] * D_ear[None, :])
    dev = torch.device("cuda", device)
    Y = torch.zeros((V, M), dtype=torch.float64, device=dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    safe = np.where(np.any(pk[:, :3], axis=1, keepdims=True), pk[:, :3], [0, 0, 1.0])
    safe2 = np.where(np.any(pk[:, 3:], axis=1, keepdims=True), pk[:, 3:], [0, 0, 1.0])
    for k, d in enumerate((safe, safe2)):
        col = engine.rotate_columns_dev(plan, t(d), t(rng.integers(0, N, V).astype(np.int32)))
        Y += t(nu[:, k:k + 1]) * col
    Y += t(nu[:, 2:3]) * t(sig_csf)[None, :]
    Y += t(nu[:, 3:4]) * t(sig_ear.T.copy())[t(rng.integers(0, D_ear.shape[0], V))]
    Y = M0 * Y + (M0 / snr) * torch.randn((V, M), dtype=torch.float64, device=dev,
                                          generator=torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 31))))
    nvox = int(np.prod(grid))
    fidx = np.ravel_multi_index(np.unravel_index(roi, grid), grid, order='F')
    planes = torch.zeros((M, nvox), dtype=torch.float32, device=dev)
    planes[:, t(fidx)] = Y.T.to(torch.float32)
    data = planes.cpu().numpy().reshape((M,) + tuple(grid)[::-1]).T          # (grid x M), Fortran-contiguous
    del planes, Y
    vol = lambda vals, extra=(): _scatter(vals, roi, grid, extra)
    return {"data": data, "mask": mask, "numfasc": vol(K.astype(np.float64)), "peaks": vol(pk, (6,)),
            "csf_mask": vol(csf.astype(np.float64)), "ear_mask": vol(ear.astype(np.float64))}


def _scatter(vals, roi, grid, extra=()):
    out = np.zeros((int(np.prod(grid)),) + tuple(extra))
    out[roi] = vals
    return out.reshape(tuple(grid) + tuple(extra))
