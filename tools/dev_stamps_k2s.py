"""Diagnostic: phase timing + short-list statistics of the screening kernel from in-kernel stamps (needs a
-DMFX_STAMPS build of the library at microstructure_fingerprinting_amd/libmfx_stamps.so)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microstructure_fingerprinting_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "microstructure_fingerprinting_amd", "libmfx_stamps.so")
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
print("fallback voxels", lib.mfx_debug_last_fallback_count(), "of", V)
raw = st.cpu().numpy()
s = raw.astype(np.float64)[V // 4: 3 * V // 4]
def d(a, b): return np.median(s[:, b] - s[:, a])
tot = d(0, 6) + d(6, 8)
print("total cycles per voxel-WG (median, non-fallback path): %.0f" % tot)
for nm, a_, b_ in [("phase0 y+descriptors", 0, 1), ("phase1 column stats", 1, 2), ("round0: A operand", 2, 3), ("round0: gen chunk0+barrier", 3, 4),
                   ("round0: chunk loop", 4, 5), ("all rounds (2->6)", 2, 6), ("exact stage", 6, 7), ("outputs", 7, 8)]:
    print("  %-28s %10.0f cycles  %5.1f %%" % (nm, d(a_, b_), 100 * d(a_, b_) / tot))
print("exact stage split: candidates %d | argmin %d | family passes %d" % (d(6, 13), d(13, 14), d(14, 7)))
print("candidates split: list compaction %d | first candidate's table rows staged %d | sums + rest %d" % (d(6, 9), d(9, 15), d(15, 13)))
fam = (s[:, 7] - s[:, 14])
print("family passes: p10 %d p50 %d p90 %d p99 %d" % tuple(np.percentile(fam, [10, 50, 90, 99])))
can = (s[:, 13] - s[:, 6])
print("candidates:    p10 %d p50 %d p90 %d p99 %d" % tuple(np.percentile(can, [10, 50, 90, 99])))
napp = raw[:, 12]
nev = raw[:, 11]
err = raw[:, 10].view(np.float64) if raw.dtype == np.int64 else None
print("appends per voxel: median %d, mean %.0f, p90 %d, p99 %d, max %d; > 1024: %.1f %%" % (np.median(napp), napp.mean(), np.percentile(napp, 90), np.percentile(napp, 99), napp.max(), 100 * np.mean(napp > 1024)))
ok = napp <= 1024
print("exactly evaluated candidates (non-overflow voxels): median %d, p99 %d, max %d" % (np.median(nev[ok]), np.percentile(nev[ok], 99), nev[ok].max()))
e = np.ascontiguousarray(raw[:, 10]).view(np.float64)
print("max |S(c~) - S_exact| / |y|^2 over evaluated two-weight candidates: %.3e (bound used: 1e-5)" % e.max())
