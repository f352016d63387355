"""One-off referee run: two fascicles + CSF ([782, 782, 1], 200 measurements; the screening pipeline of DESIGN 4.3b) against
the CPU oracle on every voxel of a bench-style batch (default 3 000 voxels: a few minutes on the GPU box's 16 CPUs)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from microstructure_fingerprinting_amd import _lib as L, engine, mf_utils as mfu
from oracle import oracle as orc

V = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sch, dic, ms = bench.build_model(782)
dev = torch.device("cuda", 0)
ms.device = 0
plan = engine.Plan(ms.device_tables(), scheme=sch)
M, N = sch.shape[0], ms.num_subs
gam = mfu.get_gyromagnetic_ratio('H')
b = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3.0e-9)
peaks_h, d_peaks, d_Y = bench.synth_voxels(plan, V, N, M, dev, 31, K=2, extra_cols=sig_csf[:, None])
d_csf = torch.from_numpy(sig_csf).to(dev)
d_out = engine.fit_batch_dev(plan, d_Y, d_peaks, 2, csf_on=True, d_sig_csf=d_csf)
torch.cuda.synchronize()
lib = L.lib()
print("GPU done; counters", [lib.mfx_debug_last_counter(q) for q in range(6)], flush=True)
got = d_out.cpu().numpy()
Y = d_Y.cpu().numpy()
T = {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat}
blk, bad_ids, worst, t0 = 250, 0, 0.0, time.time()
for v0 in range(0, V, blk):
    n = min(blk, V - v0)
    one = np.ones(n, dtype=np.uint8); z = np.zeros(n, dtype=np.uint8)
    ref = orc.fit_batch(T, sch, Y[v0:v0 + n], np.full(n, 2, dtype=np.int32), one, z, np.ascontiguousarray(peaks_h[v0:v0 + n]), 2, True, False,
                        sig_csf, None, 0, nthreads=NT)
    g = got[v0:v0 + n]
    bad = np.where(np.any(g[:, 3:5] != ref[:, 3:5], axis=1))[0]
    bad_ids += bad.size
    worst = max(worst, float(np.max(np.abs(g - ref) / np.maximum(np.abs(ref), 1e-300))))
    print("voxels %5d..%5d: %d with different atom ids (total %d), worst relative difference so far %.3e, %.0f s" % (v0, v0 + n, bad.size, bad_ids, worst, time.time() - t0), flush=True)
print(json.dumps({"workload": "[782, 782, 1] x 200 measurements, bench-style voxels with CSF signal", "voxels": V, "voxels_with_different_atom_ids": int(bad_ids),
                  "worst_relative_difference_of_any_output": worst, "oracle_seconds": round(time.time() - t0, 1)}))
