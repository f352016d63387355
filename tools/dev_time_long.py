"""Developer check: K=2 screening kernel on a 253-measurement protocol (the KS = 16 instantiations), three-image
against two-image schedule, against the FP64 kernel."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microstructure_fingerprinting_amd import _lib as L
from microstructure_fingerprinting_amd import engine, synth
from microstructure_fingerprinting_amd import mf_utils as mfu
V, N = 40000, 782
rng = np.random.default_rng(3)
sch = synth.make_scheme(rng, 2, [1000, 2000, 3000], [84, 84, 84])
dic = synth.make_dictionary(rng, sch, N)
ms = mfu.init_PGSE_multishell_interp(dic, sch, np.array([0, 0, 1.0]))
plan = ms.plan_for(sch)
M = sch.shape[0]
dev = torch.device("cuda", 0)
peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
atoms = rng.integers(0, N, (V, 2)).astype(np.int32)
nu = rng.dirichlet(np.ones(2), V)
d_pk = torch.from_numpy(peaks).to(dev)
d_Y = torch.zeros((V, M), dtype=torch.float64, device=dev)
for k in range(2):
    col = engine.rotate_columns_dev(plan, d_pk[:, 3 * k:3 * k + 3].contiguous(), torch.from_numpy(atoms[:, k].copy()).to(dev))
    d_Y += 500.0 * torch.from_numpy(nu[:, k:k + 1].copy()).to(dev) * col
d_Y += torch.from_numpy(rng.normal(0, 500 / 30.0, (V, M))).to(dev)
lib = L.lib()
res = {}
for name, screen, images in (("screening, 3 images", 1, 0), ("screening, 2 images", 1, 2), ("FP64 kernel", 0, 0)):
    lib.mfx_debug_set_k2_screen(screen); lib.mfx_debug_set_k2s_images(images)
    out = torch.zeros((V, 7), dtype=torch.float64, device=dev)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        L.check(lib.mfx_fit_batch_dev(plan.handle(), d_Y.data_ptr(), d_pk.data_ptr(), 2, 0, 0, None, None, 0, V, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[name] = out.cpu().numpy()
    print("M=%d N=%d %-22s %8.0f voxels/s" % (M, N, name, V / dt), flush=True)
lib.mfx_debug_set_k2_screen(1); lib.mfx_debug_set_k2s_images(0)
ref = res["FP64 kernel"]
for k, v in res.items():
    print(k, "differing voxels vs FP64:", int((np.abs(v - ref).max(axis=1) > 0).sum()))
