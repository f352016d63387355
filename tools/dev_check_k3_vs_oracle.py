"""One-off referee run: three fascicles at 400 atoms x 300 measurements (6.4e7 triples per voxel, ~100 s per voxel for the
oracle's solve_exhaustive_posweights_3) - the recipe of tests/test_parity_stress_gpu.py::test_three_fascicles_oracle_refereed_at_n400
(generic mixtures, two-atom and one-atom signals, 3 degree crossings, identical peaks, noise-free) with R different seeds.
    python tools/dev_check_k3_vs_oracle.py [R=8] [threads=16]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from concurrent.futures import ThreadPoolExecutor
import test_parity_stress_gpu as tp
from microstructure_fingerprinting_amd import engine, synth, mf_utils as mfu
from oracle import oracle as orc
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N, Z = 400, np.array([0.0, 0.0, 1.0])
bad, worst, total, t0 = 0, 0.0, 0, time.time()
for rep in range(R):
    rng = np.random.default_rng(7700 + rep)
    sch = synth.make_scheme(rng, 1, [1000, 2000, 3000, 4000], [75, 75, 75, 74])
    M = sch.shape[0]
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, Z)
    plan = ms.plan_for(sch)
    T = tp._tables(ms)
    kinds = ["generic"] * 5 + ["two_atoms", "two_atoms", "one_atom", "cross3", "cross3", "identical", "noise_free"]
    V = len(kinds)
    p = [synth.unit_vectors(rng, V) for _ in range(3)]
    nu = rng.dirichlet(np.ones(3) * 2, V)
    for v, kd in enumerate(kinds):
        if kd == "two_atoms": nu[v] = [0.55, 0.45, 0.0] if v % 2 else [0.0, 0.3, 0.7]
        if kd == "one_atom": nu[v] = [0.0, 1.0, 0.0]
        if kd == "cross3": p[1][v] = tp._second_peak(rng, p[0][v:v + 1], 3.0)[0]
        if kd == "identical": p[2][v] = p[0][v]
    peaks = np.concatenate(p, axis=1)
    atoms = rng.integers(0, N, (V, 3))
    Y = np.zeros((V, M))
    for k in range(3):
        Y += 500.0 * nu[:, k:k + 1] * tp._rotate_cols(plan, peaks[:, 3 * k:3 * k + 3], atoms[:, k])
    noisy = np.array([kd != "noise_free" for kd in kinds])
    Y[noisy] += rng.normal(0, 500.0 / 30.0, (int(noisy.sum()), M))
    got = engine.fit_batch(plan, Y, np.full(V, 3), None, None, peaks, 3, False, False)
    sizes = np.array([N, N, N])

    def referee(v):
        A = np.ascontiguousarray(np.concatenate([orc.interp(sch, peaks[v, 3 * k:3 * k + 3], T) for k in range(3)], axis=1))
        return orc.solve_exhaustive_posweights(A, Y[v], sizes)
    with ThreadPoolExecutor(max_workers=min(V, NT)) as ex:
        refs = list(ex.map(referee, range(V)))
    for v, (w, sub, tot, mo, yrec) in enumerate(refs):
        total += 1
        if not np.array_equal(got[v, 4:7], sub.astype(float)):
            bad += 1
            print("   DIFFERENT: rep %d voxel %d (%s): got %s, oracle %s" % (rep, v, kinds[v], got[v].tolist(), sub.tolist()), flush=True)
        ws = w.sum()
        worst = max(worst, abs(got[v, 0] - ws) / max(ws, 1e-300), float(np.max(np.abs(got[v, 1:4] - w / max(ws, 1e-300)))), abs(got[v, -2] * M - mo) / max(mo, 1e-9 * float(Y[v] @ Y[v])))
    print("rep %d: %d voxels checked, %d with different indices, worst difference %.3e, %.0f s" % (rep, total, bad, worst, time.time() - t0), flush=True)
print(json.dumps({"workload": "three fascicles, 400 atoms x 300 measurements, twelve voxel kinds x %d seeds" % R, "voxels": total,
                  "voxels_with_different_indices": bad, "worst_difference_M0_fractions_objective": worst, "oracle_seconds": round(time.time() - t0, 1)}))
