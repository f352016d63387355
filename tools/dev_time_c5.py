"""Developer timing of the three-fascicle class (BASELINE config 5: N atoms x 300 measurements, explicit-dictionary path,
one voxel at a time): seconds per voxel for growing dictionaries (the cost grows like N^3)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microstructure_fingerprinting_amd import _lib as L
if os.environ.get("MFX_DEV_LIB"):    # another build of the library (timing experiments)
    L.LIB_PATH = os.path.abspath(os.environ["MFX_DEV_LIB"])
from microstructure_fingerprinting_amd import engine, synth, mf_utils as mfu
import bench
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
sch = synth.make_scheme(rng, 1, [1000, 2000, 3000, 4000], [75, 75, 75, 74])
for N in [int(x) for x in (sys.argv[1:] or ["200", "400", "800"])]:
    dic = synth.make_dictionary(rng, sch, N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, np.array([0, 0, 1.0]))
    plan = ms.plan_for(sch)
    V = int(os.environ.get("MFX_DEV_V", "2"))
    _, dpk, dY = bench.synth_voxels(plan, V, N, sch.shape[0], dev, 7, K=3)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = engine.fit_batch_dev(plan, dY, dpk, 3)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("N = %4d (%.2e triples per voxel), M = %d: %.3f s per voxel -> %.3g voxels/s; ids %s" % (N, float(N) ** 3, sch.shape[0], dt / V, V / dt, out[0, 4:7].tolist()), flush=True)
