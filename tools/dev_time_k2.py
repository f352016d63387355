"""Developer timing: K=2 fused kernel at C2 shape with device-resident inputs."""
import sys, time, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from microstructure_fingerprinting_amd import engine, synth, _lib as L
from oracle import oracle as orc

V = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 782
c = synth.config("C2")
rng = np.random.default_rng(1)
sch = synth.make_scheme(rng, c["n_b0"], c["shells_b"], c["dirs"])
dic = synth.make_dictionary(rng, sch, N)
T = orc.init_tables(dic, sch, np.array([0, 0, 1.0]))
tabs = engine.DeviceTables(T["xs"], T["Ys"], T["G_un"])
plan = engine.Plan(tabs, scheme=sch)
M = sch.shape[0]
peaks = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
Y = 500 * dic[:, rng.integers(0, N, V)].T * rng.uniform(0.5, 1, (V, 1)) + rng.normal(0, 500 / 30, (V, M))
dY = torch.from_numpy(Y).cuda(); dpk = torch.from_numpy(peaks).cuda()
out = torch.zeros((V, 7), dtype=torch.float64, device="cuda")
lib = L.lib(); lib.mfx_set_profiling(1)
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    L.check(lib.mfx_fit_batch_dev(plan.handle(), dY.data_ptr(), dpk.data_ptr(), 2, 0, 0, None, None, 0, V, out.data_ptr(), st))
    torch.cuda.synchronize(); t1 = time.time()
    ms = lib.mfx_last_kernel_ms()
    print("V=%d N=%d: wall %.4f s, kernel %.3f ms -> %.0f voxels/s, %.2f TFLOP/s (Gram only %.2f)" % (
        V, N, t1 - t0, ms, V / (ms * 1e-3), V * (2.0 * N * N * M + 23.0 * N * N + 7 * N * M) / (ms * 1e-3) / 1e12, V * 2.0 * N * N * M / (ms * 1e-3) / 1e12))
print(out[:2].cpu().numpy())
