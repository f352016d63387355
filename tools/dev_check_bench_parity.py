"""Developer check: bench.py's workload, screening kernel vs FP64 kernel on ALL voxels (GPU vs GPU), plus the oracle
on the voxels that differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from microstructure_fingerprinting_amd import _lib as L, engine, synth
from oracle import oracle as orc

V = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
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
lib = L.lib()
st = torch.cuda.current_stream(dev)
outs = []
for screen in (1, 0):
    lib.mfx_debug_set_k2_screen(screen)
    d_out = torch.zeros((V, 7), dtype=torch.float64, device=dev)
    L.check(lib.mfx_fit_batch_dev(plan.handle(), d_Y.data_ptr(), d_peaks.data_ptr(), 2, 0, 0, None, None, 0, V, d_out.data_ptr(), st.cuda_stream))
    torch.cuda.synchronize()
    if screen:
        print("fallback voxels:", lib.mfx_debug_last_fallback_count(), "of", V)
    outs.append(d_out.cpu().numpy())
a, b = outs
bad = np.where(np.any(a != b, axis=1))[0]
print("voxels where screen != fp64 kernel:", bad.size, bad[:20])
if bad.size:
    T = {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat}
    sel = bad[:8]
    Ys = d_Y[torch.from_numpy(sel).to(dev)].cpu().numpy()
    ref = orc.fit_batch(T, sch, Ys, np.full(sel.size, 2, dtype=np.int32), np.zeros(sel.size, np.uint8), np.zeros(sel.size, np.uint8),
                        np.ascontiguousarray(peaks_h[sel]), 2, False, False, None, None, 0, nthreads=8)
    for q, v in enumerate(sel):
        print(v, "\n  screen", a[v], "\n  fp64  ", b[v], "\n  oracle", ref[q])
sys.exit(1 if bad.size else 0)
