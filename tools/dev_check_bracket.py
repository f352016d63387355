"""Developer check: bracketed protocol, screening vs FP64 kernel vs oracle on differing voxels."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from microstructure_fingerprinting_amd import _lib as L, engine, synth, mf_utils as mfu
from oracle import oracle as orc
Z = np.array([0, 0, 1.0])
sch_ms, dic, rng = synth.make_model("C2")
ms = mfu.init_PGSE_multishell_interp(dic, sch_ms, Z)
Gs = ms["Gms_un"]
sch = sch_ms[rng.permutation(sch_ms.shape[0])[:150]].copy()
nz = np.where(sch[:, 3] > 0)[0]
between = [0.3 * Gs[1] + 0.7 * Gs[2], 0.55 * Gs[2] + 0.45 * Gs[3], 0.9 * Gs[2] + 0.1 * Gs[3], 0.5 * (Gs[1] + Gs[2])]
sch[nz[::2], 3] = rng.choice(between, size=nz[::2].size)
plan = ms.plan_for(sch)
V, N, M = 8000, ms.num_subs, sch.shape[0]
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
res = []
for screen in (1, 0):
    lib.mfx_debug_set_k2_screen(screen)
    out = torch.zeros((V, 7), dtype=torch.float64, device=dev)
    L.check(lib.mfx_fit_batch_dev(plan.handle(), d_Y.data_ptr(), d_pk.data_ptr(), 2, 0, 0, None, None, 0, V, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize(dev)
    res.append(out.cpu().numpy())
bad = np.where(np.any(res[0] != res[1], axis=1))[0]
print("differing voxels:", bad)
T = {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat, "scheme_DeldelTE": ms["scheme_DeldelTE"]}
for v in bad[:4]:
    ref = orc.fit_batch(T, sch, d_Y[v:v + 1].cpu().numpy(), np.full(1, 2), np.zeros(1, bool), np.zeros(1, bool), peaks[v:v + 1], 2, False, False, None, None, 0)
    print(v, "\n screen", res[0][v], "\n fp64  ", res[1][v], "\n oracle", ref[0])
