"""End-to-end timing of the reference-style API: MFModel(dict).fit(ndarrays) on host arrays (PCIe inclusive)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import microstructure_fingerprinting_amd as mf
from microstructure_fingerprinting_amd import synth, engine, mf_utils as mfu
import torch

sch, dic, rng = synth.make_model("C2")
N = dic.shape[1]
md = {"dictionary": dic, "sch_mat": sch, "orientation": np.array([0, 0, 1.0]), "num_atom": N, "num_ear": 10,
      "T2_csf": 2.0, "DIFF_csf": 3e-9, "T2_ear": 0.08, "DIFF_ear": np.linspace(0.2e-9, 1.2e-9, 10),
      "fasc_propnames": ["rad", "fin"], "rad": rng.uniform(0.2e-6, 2e-6, N), "fin": rng.uniform(0.2, 0.9, N)}
model = mf.MFModel(md)
shape = (50, 50, 40)                       # 1e5 voxels, all inside the mask, two fascicles everywhere
V = int(np.prod(shape))
peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
plan = model.ms_interpolator.plan_for(sch)
dev = torch.device("cuda", 0)
d_pk = torch.from_numpy(peaks).to(dev)
Y = torch.zeros((V, sch.shape[0]), dtype=torch.float64, device=dev)
nu = rng.dirichlet(np.ones(2), V)
for k in range(2):
    col = engine.rotate_columns_dev(plan, d_pk[:, 3 * k:3 * k + 3].contiguous(), torch.from_numpy(rng.integers(0, N, V).astype(np.int32)).to(dev))
    Y += 500.0 * torch.from_numpy(nu[:, k:k + 1].copy()).to(dev) * col
Y += torch.from_numpy(rng.normal(0, 500 / 30, (V, sch.shape[0]))).to(dev)
data = Y.cpu().numpy().reshape(shape + (-1,))
pk = peaks.reshape(shape + (6,))
mask = np.ones(shape)
for it in range(3):
    t0 = time.time()
    fit = model.fit(data, mask, 2, peaks=pk, pgse_scheme=sch, verbose=0)
    t1 = time.time()
    print("MFModel.fit on %d voxels (host arrays in, maps out): %.1f ms -> %.0f voxels/s" % (V, (t1 - t0) * 1e3, V / (t1 - t0)), flush=True)
if os.environ.get("MFX_DEV_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    fit = model.fit(data, mask, 2, peaks=pk, pgse_scheme=sch, verbose=0)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
Yh = data.reshape(V, -1); Kh = np.full(V, 2, dtype=np.int32)
for it in range(3):
    t0 = time.time()
    engine.fit_batch(plan, Yh, Kh, None, None, peaks, 2, False, False)
    t1 = time.time()
    print("engine.fit_batch (mfx_fit_batch_rows) on %d voxels: %.1f ms" % (V, (t1 - t0) * 1e3), flush=True)
print("mean MSE %.1f (noise variance %.1f)" % (fit.MSE.mean(), (500 / 30) ** 2))
# a mixed ROI: 40 % of the voxels with one fascicle, 60 % with two; 40 % carry the CSF flag (five voxel classes per chunk)
Km = np.where(rng.random(V) < 0.4, 1, 2).astype(np.int32)
csfm = rng.random(V) < 0.4
gam = mfu.get_gyromagnetic_ratio('H')
bval = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-bval * 3e-9)
for it in range(3):
    t0 = time.time()
    engine.fit_batch(plan, Yh, Km, csfm, None, peaks, 2, True, False, sig_csf)
    t1 = time.time()
    print("mixed ROI (K = 1 / 2, with / without CSF) on %d voxels: %.1f ms -> %.0f voxels/s; counters %s" % (V, (t1 - t0) * 1e3, V / (t1 - t0), [mf._lib.lib().mfx_debug_last_counter(q) for q in range(6)]), flush=True)
