"""Developer timing: the wide screening kernel (fit_k2w.hip) against the default one on the bench workload, and the
long-protocol instantiations against the FP64 kernel."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microstructure_fingerprinting_amd import _lib as L, engine, synth, mf_utils as mfu
import bench
lib = L.lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream

def run(name, sch, dic, V, settings):
    ms = mfu.init_PGSE_multishell_interp(dic, sch, np.array([0, 0, 1.0]))
    plan = ms.plan_for(sch)
    M, N = sch.shape[0], ms.num_subs
    _, dpk, dY = bench.synth_voxels(plan, V, N, M, dev, 11)
    outs = []
    for label, wide, screen in settings:
        lib.mfx_debug_set_k2_wide(wide); lib.mfx_debug_set_k2_screen(screen)
        out = torch.zeros((V, 7), dtype=torch.float64, device=dev)
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            L.check(lib.mfx_fit_batch_dev(plan.handle(), dY.data_ptr(), dpk.data_ptr(), 2, 0, 0, None, None, 0, V, out.data_ptr(), st))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        outs.append(out.cpu().numpy())
        print("%-40s %-28s M=%3d N=%4d V=%6d: %8.2f ms -> %9.0f voxels/s (handed back %d)%s"
              % (name, label, M, N, V, dt * 1e3, V / dt, lib.mfx_debug_last_fallback_count(),
                 "" if len(outs) == 1 else "  identical to first: %s" % bool(np.array_equal(outs[0], outs[-1]))), flush=True)
    lib.mfx_debug_set_k2_wide(-1); lib.mfx_debug_set_k2_screen(1)

sch, dic, _ = synth.make_model("C2")
run("C2 (782 x 200)", sch, dic, 100000, [("default (k2s<13,3>)", 0, 1), ("wide (k2w<13,2,2>)", 1, 1)])
rng = np.random.default_rng(3)
sch = synth.make_scheme(rng, 2, [1000, 2000, 3000], [84, 84, 84])
run("254 rows", sch, synth.make_dictionary(rng, sch, 782), 50000, [("default (k2s<16>)", 0, 1), ("wide (k2w<16,2,2>)", 1, 1)])
sch = synth.make_scheme(rng, 2, [1000, 2000, 3000], [100, 100, 100])
run("302 rows", sch, synth.make_dictionary(rng, sch, 782), 20000, [("wide (k2w<24,1,1>)", -1, 1), ("FP64 kernel", -1, 0)])
sch = synth.make_scheme(rng, 2, [1000, 2000, 3000, 5000], [137, 137, 137, 139])
run("552 rows (HCP-MGH length)", sch, synth.make_dictionary(rng, sch, 782), 20000, [("wide (k2w<35,1,1>)", -1, 1), ("FP64 kernel", -1, 0)])
