"""Diagnostic: phase timing of the two-fascicle + CSF/EAR kernel from in-kernel stamps (needs the -DMFX_STAMPS build
microstructure_fingerprinting_amd/libmfx_stamps.so: `make -C microstructure_fingerprinting_amd/csrc stamps`).
    python tools/dev_stamps_k2x.py [csf] [ear] [V]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microstructure_fingerprinting_amd import _lib as L
WMODE = len(sys.argv) > 4 and sys.argv[4] == "w"   # chunk-level stamps (libmfx_stamps_w.so: make ... stamps_w)
L.LIB_PATH = os.path.join(ROOT, "microstructure_fingerprinting_amd", "libmfx_stamps_w.so" if WMODE else "libmfx_stamps.so")
from microstructure_fingerprinting_amd import engine, mf_utils as mfu
import bench
c = int(sys.argv[1]) if len(sys.argv) > 1 else 1
e = int(sys.argv[2]) if len(sys.argv) > 2 else 1
V = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
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
xc = np.concatenate(([sig_csf[:, None]] if c else []) + ([sig_ear[:, 3:4]] if e else []), axis=1)
_, dpk, dY = bench.synth_voxels(plan, V, N, M, dev, 5, K=2, extra_cols=xc)
dcsf, dear = torch.from_numpy(sig_csf).to(dev), torch.from_numpy(sig_ear).to(dev)
out = torch.zeros((V, engine.num_params(2, c, e)), dtype=torch.float64, device=dev)
st = torch.zeros((2048, 16), dtype=torch.int64, device=dev)   # the kernel is launched in chunks of 2048 workgroups
lib = L.lib()
lib.mfx_debug_set_stamps(st.data_ptr())
for _ in range(2):
    st.zero_()
    L.check(lib.mfx_fit_batch_dev(plan.handle(), dY.data_ptr(), dpk.data_ptr(), 2, c, e, dcsf.data_ptr() if c else None,
                                  dear.data_ptr() if e else None, E if e else 0, V, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)[:min(V, 2048)]
def d(a, b): return np.median(s[:, b] - s[:, a])
if WMODE:
    for w, off in (("wave 0", 0), ("last wave", 8)):
        print("chunk 10 of round 1, %s (median cycles):" % w)
        for k, nm in enumerate(["gen_load (issue)", "MFMA loop", "gen_store", "filter + scoring", "barrier wait"]):
            print("   %-20s %8.0f   (mean %8.0f)" % (nm, d(off + k, off + k + 1), np.mean(s[:, off + k + 1] - s[:, off + k])))
        ok = s[:, off + 6] > 0
        print("   of it the filter    %8.0f   (mean %8.0f)" % (np.median(s[ok, off + 6] - s[ok, off + 3]), np.mean(s[ok, off + 6] - s[ok, off + 3])))
        print("   queue + scoring     %8.0f   (mean %8.0f)" % (np.median(s[ok, off + 4] - s[ok, off + 6]), np.mean(s[ok, off + 4] - s[ok, off + 6])))
        print("   tuples in the wave's queue: median %d, mean %.1f, p90 %d, max %d" % (np.median(s[:, off + 7]), s[:, off + 7].mean(), np.percentile(s[:, off + 7], 90), s[:, off + 7].max()))
    sys.exit(0)
tot = d(0, 9)
print("class csf=%d ear=%d: total cycles per voxel (median) %.0f" % (c, e, tot))
for nm, a_, b_ in [("phase 0", 0, 1), ("phase 1 statistics", 1, 2), ("round 0: A operand + row constants", 2, 3), ("round 0: first chunk", 3, 4),
                   ("round 0: chunk loop", 4, 5), ("all rounds", 2, 6), ("family detection", 6, 7), ("exact stage", 7, 8), ("outputs", 8, 9)]:
    print("  %-36s %10.0f  %5.1f %%" % (nm, d(a_, b_), 100 * d(a_, b_) / tot))
if s[:, 12].any():   # list mode (the class runs on the screening pipeline): the exact stage in detail
    for nm, a_, b_ in [("  list: pairs + single atoms exact", 7, 12), ("  list: reduction", 12, 13), ("  list: family detection", 13, 14), ("  list: families exact + argmin", 14, 8)]:
        print("  %-36s %10.0f  %5.1f %%   (mean %.0f)" % (nm, d(a_, b_), 100 * d(a_, b_) / tot, np.mean(s[:, b_] - s[:, a_])))
print("tuples passing the filter per voxel: median %d, mean %.0f, p90 %d, max %d (of %d)" % (np.median(s[:, 10]), s[:, 10].mean(), np.percentile(s[:, 10], 90), s[:, 10].max(), N * N * (E if e else 1)))
print("(wave, row group) scoring passes per voxel: median %d, mean %.0f (of %d wave tiles x 4)" % (np.median(s[:, 11]), s[:, 11].mean(), 49 * 49))
