"""Diagnostic: duration of the half-steps of the screening kernel's ping-pong sweep (round 1, half-steps 20..34, wave 0).
Needs a -DMFX_STAMPS_HS build of the library at microstructure_fingerprinting_amd/libmfx_stamps_hs.so (MFX_STAMPS must
NOT be defined: the slots overlap)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microstructure_fingerprinting_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "microstructure_fingerprinting_amd", "libmfx_stamps_hs.so")
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
raw = st.cpu().numpy().astype(np.float64)[V // 4: 3 * V // 4]
x = raw[(raw > 0).all(axis=1)]
d = np.diff(x, axis=1)
print("half-step periods, round 1, half-steps 20..34 (wave 0: even = its MFMA half-step), cycles")
print("  median " + " ".join("%5.0f" % v for v in np.median(d, axis=0)))
print("  mean   " + " ".join("%5.0f" % v for v in np.mean(d, axis=0)))
