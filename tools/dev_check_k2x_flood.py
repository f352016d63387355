"""Developer check: K=2 + CSF at N=782 on voxels whose optimum has an inactive fascicle atom (short-list flood scenario)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from microstructure_fingerprinting_amd import engine, synth, mf_utils as mfu
from oracle import oracle as orc
Z = np.array([0, 0, 1.0])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 782
sch, dic, rng = synth.make_model("C2", N=N)
ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
T = {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat, "scheme_DeldelTE": ms["scheme_DeldelTE"]}
b = (orc.GAMMA_H * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3e-9)
V = 6
peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
Y = rng.normal(0, 500 / 30.0, (V, sch.shape[0]))
for v in range(V):
    a1 = orc.interp(sch, peaks[v, :3], T)[:, rng.integers(0, N)]
    a2 = orc.interp(sch, peaks[v, 3:], T)[:, rng.integers(0, N)]
    w = [(0.7, 0.0, 0.3), (0.0, 0.6, 0.4), (1.0, 0.0, 0.0), (0.5, 0.3, 0.2), (0.0, 0.0, 1.0), (0.9, 0.1, 0.0)][v]
    Y[v] += 500 * (w[0] * a1 + w[1] * a2 + w[2] * sig_csf)
Kv = np.full(V, 2); cm = np.ones(V, bool); em = np.zeros(V, bool)
t0 = time.time()
ref = orc.fit_batch(T, sch, Y, Kv, cm, em, peaks, 2, True, False, sig_csf, None, 0, nthreads=16)
print("oracle %.1f s" % (time.time() - t0))
got = engine.fit_batch(ms.plan_for(sch), Y, Kv, cm, em, peaks, 2, True, False, sig_csf, None, 0)
for v in range(V):
    ok = np.array_equal(got[v, 3:5], ref[v, 3:5]) and np.allclose(got[v], ref[v], rtol=1e-8, atol=1e-9)
    print(v, "OK " if ok else "BAD", "\n  gpu", got[v], "\n  ref", ref[v])
