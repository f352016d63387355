"""Diagnostic: where a chunk's time goes in the wide screening kernel (needs the -DMFX_STAMPS_W build
microstructure_fingerprinting_amd/libmfx_stamps_w.so: `make -C microstructure_fingerprinting_amd/csrc stamps_w`)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microstructure_fingerprinting_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "microstructure_fingerprinting_amd", "libmfx_stamps_w.so")
from microstructure_fingerprinting_amd import engine
import bench
V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sch, dic, ms = bench.build_model(782)
dev = torch.device("cuda", 0)
ms.device = 0
plan = engine.Plan(ms.device_tables(), scheme=sch)
M, N = sch.shape[0], ms.num_subs
_, d_peaks, d_Y = bench.synth_voxels(plan, V, N, M, dev, 1000)
out = torch.zeros((V, 7), dtype=torch.float64, device=dev)
st = torch.zeros((V, 16), dtype=torch.int64, device=dev)
lib = L.lib()
lib.mfx_debug_set_k2_wide(1)
lib.mfx_debug_set_stamps(st.data_ptr())
for _ in range(2):
    st.zero_()
    L.check(lib.mfx_fit_batch_dev(plan.handle(), d_Y.data_ptr(), d_peaks.data_ptr(), 2, 0, 0, None, None, 0, V, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)[V // 4: 3 * V // 4]
names = ["screen_begin (LDS consts)", "MFMA loop + slices", "screen_finish (FP64 pass)", "gen_load issue", "barrier wait"]
for c in range(2):
    print("chunk %d of round 1, wave 0 (median cycles):" % (10 + c))
    for k, nm in enumerate(names):
        print("   %-28s %8.0f" % (nm, np.median(s[:, 8 * c + k + 1] - s[:, 8 * c + k])))
print("chunk period (start 10 -> start 11): %.0f" % np.median(s[:, 8] - s[:, 0]))
