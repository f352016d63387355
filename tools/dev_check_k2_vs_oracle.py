"""One-off referee run: two fascicles on protocols of other lengths (the wide screening kernel of DESIGN 4.1b: 257..560
measurements; short protocols: KS = 4 / 8) against the CPU oracle on every voxel.
    python tools/dev_check_k2_vs_oracle.py V threads dirs_per_shell [atoms=782]"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from microstructure_fingerprinting_amd import _lib as L, engine, synth, mf_utils as mfu
from oracle import oracle as orc
V, NT, nd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 782
BRACKET = len(sys.argv) > 5 and sys.argv[5] == "bracket"   # subject protocol of 105 rows, half of them at gradient strengths BETWEEN the table's shells
rng = np.random.default_rng(nd)
sch = synth.make_scheme(rng, 2, [1000, 2000, 3000], [nd, nd, nd])
dic = synth.make_dictionary(rng, sch, N)
ms = mfu.init_PGSE_multishell_interp(dic, sch, np.array([0, 0, 1.0]))
if BRACKET:
    Gs = ms["Gms_un"]
    sch = sch[rng.permutation(sch.shape[0])[:105]].copy()
    nz = np.where(sch[:, 3] > 0)[0]
    between = [0.3 * Gs[1] + 0.7 * Gs[2], 0.55 * Gs[2] + 0.45 * Gs[3], 0.9 * Gs[2] + 0.1 * Gs[3], 0.5 * (Gs[1] + Gs[2])]
    sch[nz[::2], 3] = rng.choice(between, size=nz[::2].size)
M = sch.shape[0]
plan = ms.plan_for(sch)
dev = torch.device("cuda", 0)
peaks_h, d_peaks, d_Y = bench.synth_voxels(plan, V, N, M, dev, 100 + nd)
d_out = engine.fit_batch_dev(plan, d_Y, d_peaks, 2)
torch.cuda.synchronize()
lib = L.lib()
print("M = %d, N = %d: GPU done, handed back %d, counters %s" % (M, N, lib.mfx_debug_last_fallback_count(), [lib.mfx_debug_last_counter(q) for q in (8, 9, 10)]), flush=True)
got, Y = d_out.cpu().numpy(), d_Y.cpu().numpy()
T = {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat, "scheme_DeldelTE": ms["scheme_DeldelTE"]}
blk, bad, worst, t0 = 500, 0, 0.0, time.time()
for v0 in range(0, V, blk):
    n = min(blk, V - v0)
    z = np.zeros(n, dtype=np.uint8)
    ref = orc.fit_batch(T, sch, Y[v0:v0 + n], np.full(n, 2, dtype=np.int32), z, z, np.ascontiguousarray(peaks_h[v0:v0 + n]), 2, False, False, None, None, 0, nthreads=NT)
    g = got[v0:v0 + n]
    bad += int(np.sum(np.any(g[:, 3:5] != ref[:, 3:5], axis=1)))
    worst = max(worst, float(np.max(np.abs(g - ref) / np.maximum(np.abs(ref), 1e-300))))
    print("voxels %6d..%6d: total %d with different atom ids, worst relative difference %.3e, %.0f s" % (v0, v0 + n, bad, worst, time.time() - t0), flush=True)
print(json.dumps({"workload": "two fascicles, %d atoms x %d measurements%s" % (N, M, " (half of the rows bracketed between table shells)" if BRACKET else ""), "voxels": V, "voxels_with_different_atom_ids": bad,
                  "worst_relative_difference_of_any_output": worst, "oracle_seconds": round(time.time() - t0, 1)}))
