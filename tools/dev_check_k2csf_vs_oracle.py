"""One-off referee run: two fascicles + CSF and / or EAR ([782, 782, 1]: the screening pipeline of DESIGN 4.3b; [782, 782, 10],
[782, 782, 1, 10]: fit_k2x) against the CPU oracle on every voxel of a bench-style batch.
    python tools/dev_check_k2csf_vs_oracle.py V threads [csf 0|1] [ear 0|1]"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from microstructure_fingerprinting_amd import _lib as L, engine, mf_utils as mfu
from oracle import oracle as orc

V = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 16
CSF = int(sys.argv[3]) if len(sys.argv) > 3 else 1
EAR = int(sys.argv[4]) if len(sys.argv) > 4 else 0
KF = int(sys.argv[5]) if len(sys.argv) > 5 else 2          # fascicles per voxel (1: the one-fascicle kernel, fit_small.hip)
E = 10
sch, dic, ms = bench.build_model(782)
dev = torch.device("cuda", 0)
ms.device = 0
plan = engine.Plan(ms.device_tables(), scheme=sch)
M, N = sch.shape[0], ms.num_subs
gam = mfu.get_gyromagnetic_ratio('H')
b = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3.0e-9)
sig_ear = np.ascontiguousarray(np.stack([np.exp(-sch[:, 6] / 0.08) * np.exp(-b * D) for D in np.linspace(0.2e-9, 1.2e-9, E)], axis=1))
xc = np.concatenate(([sig_csf[:, None]] if CSF else []) + ([sig_ear[:, 3:4]] if EAR else []), axis=1) if (CSF or EAR) else None
peaks_h, d_peaks, d_Y = bench.synth_voxels(plan, V, N, M, dev, 31, K=KF, extra_cols=xc)
d_csf, d_ear = torch.from_numpy(sig_csf).to(dev), torch.from_numpy(sig_ear).to(dev)
d_out = engine.fit_batch_dev(plan, d_Y, d_peaks, KF, csf_on=bool(CSF), ear_on=bool(EAR), d_sig_csf=d_csf if CSF else None,
                             d_sig_ear=d_ear if EAR else None, E=E if EAR else 0)
torch.cuda.synchronize()
lib = L.lib()
print("GPU done; counters", [lib.mfx_debug_last_counter(q) for q in range(6)], "audit (beyond DC/4, max err 1e-11, pairs)", [lib.mfx_debug_last_counter(q) for q in (8, 9, 10)], flush=True)
got = d_out.cpu().numpy()
Y = d_Y.cpu().numpy()
T = {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat}
blk, bad_ids, worst, t0 = (250 if not EAR else (100 if not CSF else 32)), 0, 0.0, time.time()
for v0 in range(0, V, blk):
    n = min(blk, V - v0)
    cf = np.full(n, CSF, dtype=np.uint8); ef = np.full(n, EAR, dtype=np.uint8)
    ref = orc.fit_batch(T, sch, Y[v0:v0 + n], np.full(n, KF, dtype=np.int32), cf, ef, np.ascontiguousarray(peaks_h[v0:v0 + n]), KF, bool(CSF), bool(EAR),
                        sig_csf if CSF else None, sig_ear if EAR else None, E if EAR else 0, nthreads=NT)
    g = got[v0:v0 + n].copy()
    if CSF and EAR and KF == 2:      # _4up: the index of a compartment with zero weight is not reproducible (exact ties, see tests)
        for col_nu, col_id in ((1, 3), (2, 4), (6, 7)):
            off = ref[:, col_nu] <= 1e-9
            g[off, col_id] = 0; ref[off, col_id] = 0
    idc = [1 + KF + k for k in range(KF)] + ([2 * KF + CSF + 2] if EAR else [])
    bad = np.where(np.any(g[:, idc] != ref[:, idc], axis=1))[0]
    bad_ids += bad.size
    worst = max(worst, float(np.max(np.abs(g - ref) / np.maximum(np.abs(ref), 1e-300))))
    print("voxels %5d..%5d: %d with different atom ids (total %d), worst relative difference so far %.3e, %.0f s" % (v0, v0 + n, bad.size, bad_ids, worst, time.time() - t0), flush=True)
print(json.dumps({"workload": ("two fascicles" if KF == 2 else "one fascicle") + "%s%s, 782 atoms x 200 measurements, bench-style voxels with every compartment present" % (" + CSF" if CSF else "", " + EAR (10 columns)" if EAR else ""), "voxels": V, "voxels_with_different_atom_ids": int(bad_ids),
                  "worst_relative_difference_of_any_output": worst, "oracle_seconds": round(time.time() - t0, 1)}))
