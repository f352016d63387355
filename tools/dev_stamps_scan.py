"""Diagnostic (make stamps_scan -> libmfx_stamps_scan.so): wave 0's first pair screen of every voxel in detail."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microstructure_fingerprinting_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "microstructure_fingerprinting_amd", "libmfx_stamps_scan.so")
from microstructure_fingerprinting_amd import engine, synth
import bench
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sch, dic, ms = bench.build_model(782)
dev = torch.device("cuda", 0)
ms.device = 0
plan = engine.Plan(ms.device_tables(), scheme=sch)
M, N = sch.shape[0], ms.num_subs
rng = np.random.default_rng(1000)
peaks_h = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
atoms_h = rng.integers(0, N, (V, 2)).astype(np.int32)
nu_h = rng.dirichlet(np.ones(2), V)
d_peaks = torch.from_numpy(peaks_h).to(dev)
d_Y = torch.zeros((V, M), dtype=torch.float64, device=dev)
for k in range(2):
    col = engine.rotate_columns_dev(plan, d_peaks[:, 3 * k:3 * k + 3].contiguous(), torch.from_numpy(atoms_h[:, k].copy()).to(dev))
    d_Y += 500.0 * torch.from_numpy(nu_h[:, k:k + 1].copy()).to(dev) * col
gen = torch.Generator(device=dev); gen.manual_seed(1234)
d_Y += torch.randn((V, M), dtype=torch.float64, device=dev, generator=gen) * (500.0 / 30.0)
out = torch.zeros((V, 7), dtype=torch.float64, device=dev)
st = torch.zeros((V, 16), dtype=torch.int64, device=dev)
lib = L.lib()
lib.mfx_debug_set_stamps(st.data_ptr())
for _ in range(2):
    st.zero_()
    L.check(lib.mfx_fit_batch_dev(plan.handle(), d_Y.data_ptr(), d_peaks.data_ptr(), 2, 0, 0, None, None, 0, V, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
raw = st.cpu().numpy()[V // 4: 3 * V // 4].astype(np.float64)
ok = (raw[:, 0] > 0) & (raw[:, 1] > 0)
print("voxels with stamps: %d; first screen flagged in %.1f %%" % (ok.sum(), 100 * np.mean(raw[ok, 2] > 0)))
f = ok & (raw[:, 2] > 0) & (raw[:, 3] > 0)
x = raw[f]
print("fast test (entry -> mm[] known):        median %6.0f cycles" % np.median(x[:, 1] - x[:, 0]))
print("slow path (flag -> end):                median %6.0f  mean %6.0f cycles, groups evaluated: median %d mean %.1f" % (np.median(x[:, 3] - x[:, 2]), np.mean(x[:, 3] - x[:, 2]), np.median(x[:, 15]), x[:, 15].mean()))
g = x[x[:, 15] >= 8]
if len(g):
    d = np.diff(np.concatenate([g[:, 2:3], g[:, 4:12]], axis=1), axis=1)
    print("flag -> end of group 1, then group by group (voxels with >= 8 groups): " + " ".join("%5.0f" % v for v in np.median(d, axis=0)))
