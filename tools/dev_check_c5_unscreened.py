"""One-off referee run: config-5-sized three-fascicle voxels (1 500 atoms x 300 measurements) through the batched screened
path and through the unscreened scan of all 3.4e9 triples (mfx_debug_set_k3_screen(0)); rows must be identical."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from microstructure_fingerprinting_amd import _lib as L, engine, synth, mf_utils as mfu
import bench
V = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(3)
sch = synth.make_scheme(rng, 1, [1000, 2000, 3000, 4000], [75, 75, 75, 74])
N, M = 1500, sch.shape[0]
dic = synth.make_dictionary(rng, sch, N)
ms = mfu.init_PGSE_multishell_interp(dic, sch, np.array([0, 0, 1.0]))
plan = ms.plan_for(sch)
dev = torch.device("cuda", 0)
lib = L.lib()
bad, t0 = 0, time.time()
for snr, seed in ((30.0, 21), (10.0, 22), (100.0, 23)):
    _, dpk, dY = bench.synth_voxels(plan, V, N, M, dev, seed, K=3, snr=snr)
    got = engine.fit_batch_dev(plan, dY, dpk, 3).cpu().numpy()
    ref = np.zeros_like(got)
    lib.mfx_debug_set_k3_screen(0)
    try:
        for v0 in range(0, V, 200):      # progress lines: 200 voxels = 7 s
            ref[v0:v0 + 200] = engine.fit_batch_dev(plan, dY[v0:v0 + 200].contiguous(), dpk[v0:v0 + 200].contiguous(), 3).cpu().numpy()
            print("snr %g: %d voxels scanned, %.0f s" % (snr, min(v0 + 200, V), time.time() - t0), flush=True)
    finally:
        lib.mfx_debug_set_k3_screen(1)
    d = np.where(np.any(got != ref, axis=1))[0]
    bad += d.size
    print("snr %g: %d of %d voxels differ %s" % (snr, d.size, V, d[:8].tolist()), flush=True)
print(json.dumps({"workload": "1500^3 triples x 300 measurements, bench-style voxels at SNR 30 / 10 / 100", "voxels": 3 * V, "voxels_that_differ": int(bad)}))
