"""One-off referee run: bench.py's workload (config 2: 782 atoms x 200 measurements, two fascicles), the product path
against the CPU oracle on ALL voxels (default 1e5: about six minutes on the GPU box's 16 CPUs).  Prints a progress line
per block; the summary goes to profiles/ by hand."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from microstructure_fingerprinting_amd import _lib as L, engine, synth
from oracle import oracle as orc

V = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sch, dic, ms = bench.build_model(782)
dev = torch.device("cuda", 0)
ms.device = 0
plan = engine.Plan(ms.device_tables(), scheme=sch)
M, N = sch.shape[0], ms.num_subs
peaks_h, d_peaks, d_Y = bench.synth_voxels(plan, V, N, M, dev, 1000)     # rank 0 of bench.py
d_out = engine.fit_batch_dev(plan, d_Y, d_peaks, 2)
torch.cuda.synchronize()
lib = L.lib()
aud = [lib.mfx_debug_last_counter(q) for q in (8, 9, 10)]
print("GPU done: handed back %d, audit: %d pairs, max err %.3e, beyond quarter margin %d" % (lib.mfx_debug_last_fallback_count(), aud[2], aud[1] * 1e-11, aud[0]), flush=True)
got = d_out.cpu().numpy()
Y = d_Y.cpu().numpy()
T = {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat}
blk = 4000
bad_ids, worst, t0 = 0, 0.0, time.time()
for v0 in range(0, V, blk):
    n = min(blk, V - v0)
    z = np.zeros(n, dtype=np.uint8)
    ref = orc.fit_batch(T, sch, Y[v0:v0 + n], np.full(n, 2, dtype=np.int32), z, z, np.ascontiguousarray(peaks_h[v0:v0 + n]), 2, False, False,
                        None, None, 0, nthreads=NT)
    g = got[v0:v0 + n]
    bad = np.where(np.any(g[:, 3:5] != ref[:, 3:5], axis=1))[0]
    bad_ids += bad.size
    worst = max(worst, float(np.max(np.abs(g - ref) / np.maximum(np.abs(ref), 1e-300))))
    print("voxels %6d..%6d: %d with different atom ids (total %d), worst relative difference of any output so far %.3e, %.0f s" % (v0, v0 + n, bad.size, bad_ids, worst, time.time() - t0), flush=True)
    for i in bad[:5]:
        print("   voxel %d: got %s, oracle %s" % (v0 + i, g[i].tolist(), ref[i].tolist()), flush=True)
res = {"workload": "bench.py config 2, rank 0 (seed 1000)", "voxels": V, "oracle_threads": NT, "voxels_with_different_atom_ids": int(bad_ids),
       "worst_relative_difference_of_any_output": worst, "audit": {"pairs": aud[2], "max_abs_err": aud[1] * 1e-11, "beyond_quarter_margin": aud[0]},
       "oracle_seconds": round(time.time() - t0, 1)}
print(json.dumps(res))
