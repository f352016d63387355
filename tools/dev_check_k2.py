"""Developer check: K=2 fused kernel vs the CPU oracle (small + C2-shaped)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from microstructure_fingerprinting_amd import engine, synth
from oracle import oracle as orc

def run(N, V, seed=1, name="C2"):
    c = synth.config(name)
    rng = np.random.default_rng(seed)
    sch = synth.make_scheme(rng, c["n_b0"], c["shells_b"], c["dirs"])
    dic = synth.make_dictionary(rng, sch, N)
    T = orc.init_tables(dic, sch, np.array([0, 0, 1.0]))
    def rot(dirs):
        return np.stack([orc.interp(sch, d, T) for d in dirs])
    peaks, Y, atoms, nu = synth.make_voxels(rng, V, 2, rot, N)
    t0 = time.time()
    Pref = orc.fit_batch(T, sch, Y, np.full(V, 2), np.zeros(V, bool), np.zeros(V, bool), peaks, 2, False, False, None, None, 0, nthreads=8)
    t1 = time.time()
    tabs = engine.DeviceTables(T["xs"], T["Ys"], T["G_un"])
    plan = engine.Plan(tabs, scheme=sch)
    t2 = time.time()
    P = engine.fit_batch(plan, Y, np.full(V, 2), None, None, peaks, 2, False, False)
    t3 = time.time()
    ids_ok = np.array_equal(P[:, 3:5], Pref[:, 3:5])
    rel = np.max(np.abs(P - Pref) / (np.abs(Pref) + 1e-300))
    print("N=%d V=%d: ids equal=%s  max rel diff=%.3e  oracle %.2fs gpu(first call) %.3fs" % (N, V, ids_ok, rel, t1 - t0, t3 - t2))
    if not ids_ok or rel > 1e-9:
        bad = np.where(np.any(P[:, 3:5] != Pref[:, 3:5], axis=1))[0]
        print("bad voxels", bad[:10]); print(P[bad[:3]]); print(Pref[bad[:3]])
        print("first rows gpu", P[:2]); print("first rows ref", Pref[:2])
    return ids_ok and rel < 1e-9

ok = run(48, 16) and run(100, 8) and run(782, 6)
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
