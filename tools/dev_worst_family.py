"""Diagnostic for tests/test_parity_stress_gpu.py::test_k2_worst_operand_family_as_dictionary_vs_oracle: which kernel decides
the voxels that differ from the oracle (screening kernel / FP64 kernel), hand-back counts."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from microstructure_fingerprinting_amd import _lib as L
from microstructure_fingerprinting_amd import engine, synth
from microstructure_fingerprinting_amd import mf_utils as mfu
from oracle import oracle as orc
Z = np.array([0, 0, 1.0])
rng = np.random.default_rng(2024)
S, N = 67, 782
sch = synth.make_scheme(rng, 2, list(np.linspace(200.0, 10000.0, S - 1)), [3] * (S - 1))
az = rng.uniform(0, 2 * np.pi, sch.shape[0])
uz = np.tile([0.15, 0.5, 0.85], S)[1:]
sch[2:, 0], sch[2:, 1], sch[2:, 2] = (np.sqrt(1 - uz ** 2) * np.cos(az))[2:], (np.sqrt(1 - uz ** 2) * np.sin(az))[2:], uz[2:]
M = sch.shape[0]
shell = np.concatenate([[0, 0], np.repeat(np.arange(1, S), 3)])
amp, rate = 0.3 + 0.7 * rng.random(N), rng.random(N)
dic = amp[None, :] * np.exp(-3.0 * rate[None, :] * shell[:, None] / S)
ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
plan = ms.plan_for(sch)
V = 384
peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
atoms = rng.integers(0, N, (V, 2))
nu = rng.dirichlet(np.ones(2), V)
nu[V // 2: V // 2 + V // 8] = [1.0, 0.0]
Y = 500.0 * (nu[:, :1] * dic[:, atoms[:, 0]].T + nu[:, 1:] * dic[:, atoms[:, 1]].T)
q = V // 4
Y[:q] += rng.normal(0, 500.0 / 30.0, (q, M))
Y[q:2 * q] += rng.normal(0, 500.0 / 100.0, (q, M))
Y[3 * q:] += rng.normal(0, 500.0 / 30.0, (V - 3 * q, M))
dev = torch.device("cuda", 0)
lib = L.lib()
T = {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat, "scheme_DeldelTE": ms["scheme_DeldelTE"]}
z = np.zeros(V, bool)
ref = orc.fit_batch(T, sch, Y, np.full(V, 2), z, z, peaks, 2, False, False, None, None, 0, nthreads=16)
dY, dP = torch.from_numpy(Y).to(dev), torch.from_numpy(peaks).to(dev)
for scr in (1, 0):
    lib.mfx_debug_set_k2_screen(scr)
    got = engine.fit_batch_dev(plan, dY, dP, 2).cpu().numpy()
    bad = np.where(np.any(got[:, 3:5] != ref[:, 3:5], axis=1))[0]
    print("screen", scr, "handed back", lib.mfx_debug_last_fallback_count(), "guard", lib.mfx_debug_last_guard_count(), "differ", bad,
          [(got[b, 3:5], ref[b, 3:5], got[b, -2] - ref[b, -2]) for b in bad[:6]])
    # one voxel at a time (is the result launch-size dependent?)
    for b in bad[:3]:
        g1 = engine.fit_batch_dev(plan, dY[b:b + 1].contiguous(), dP[b:b + 1].contiguous(), 2).cpu().numpy()
        print("   alone:", b, g1[0, 3:5], "fallback", lib.mfx_debug_last_fallback_count())
lib.mfx_debug_set_k2_screen(1)
