"""Diagnostic: phase timing of the K=2 kernel from in-kernel s_memtime stamps (needs a -DMFX_STAMPS build
of the library at microstructure_fingerprinting_amd/libmfx_stamps.so)."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microstructure_fingerprinting_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "microstructure_fingerprinting_amd", "libmfx_stamps.so")
from microstructure_fingerprinting_amd import engine, synth, mf_utils as mfu
V = 4096
sch, dic, rng = synth.make_model("C2")
ms = mfu.init_PGSE_multishell_interp(dic, sch, np.array([0, 0, 1.0]))
plan = ms.plan_for(sch)
M, N = sch.shape[0], ms.num_subs
peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
Y = 500 * dic[:, rng.integers(0, N, V)].T * rng.uniform(0.5, 1, (V, 1)) + rng.normal(0, 500 / 30, (V, M))
dY = torch.from_numpy(Y).cuda(); dpk = torch.from_numpy(peaks).cuda()
out = torch.zeros((V, 7), dtype=torch.float64, device="cuda")
st = torch.zeros((V, 16), dtype=torch.int64, device="cuda")
lib = L.lib()
lib.mfx_debug_set_stamps(st.data_ptr())
for _ in range(2):
    L.check(lib.mfx_fit_batch_dev(plan.handle(), dY.data_ptr(), dpk.data_ptr(), 2, 0, 0, None, None, 0, V, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
s = s[1024:3072]   # steady-state workgroups
def d(a, b): return np.median(s[:, b] - s[:, a])
tot = d(0, 8)
names = [("phase0 y+descriptors", 0, 1), ("phase1 column stats", 1, 2), ("round0: A-frag load", 2, 3), ("round0: gen chunk0+barrier", 3, 4),
         ("round0: chunk loop", 4, 5), ("round0: round end", 5, 9), ("all rounds (2->6)", 2, 6), ("phase3 exact", 6, 7), ("outputs", 7, 8)]
print("total cycles per voxel-WG: %.0f" % tot)
for nm, a_, b_ in names:
    print("  %-28s %10.0f cycles  %5.1f %%" % (nm, d(a_, b_), 100 * d(a_, b_) / tot))
