"""Developer check: throughput and hand-back rate of the K=2 screening kernel across data regimes (SNR, crossing
angle, one dominant fascicle, noise-free), each compared with the FP64 kernel on the same voxels."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microstructure_fingerprinting_amd import _lib as L
from microstructure_fingerprinting_amd import engine, synth
import bench
V = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
sch, dic, ms = bench.build_model(782)
dev = torch.device("cuda", 0)
ms.device = 0
plan = engine.Plan(ms.device_tables(), scheme=sch)
M, N = sch.shape[0], ms.num_subs
lib = L.lib()
stream = torch.cuda.current_stream(dev)

def run(name, snr, angle_deg=None, nu0=None):
    rng = np.random.default_rng(7)
    p1 = synth.unit_vectors(rng, V)
    if angle_deg is None:
        p2 = synth.unit_vectors(rng, V)
    else:   # second peak at a fixed angle from the first
        r = synth.unit_vectors(rng, V)
        r -= (r * p1).sum(1, keepdims=True) * p1
        r /= np.linalg.norm(r, axis=1, keepdims=True)
        a = np.deg2rad(angle_deg)
        p2 = np.cos(a) * p1 + np.sin(a) * r
        p2 /= np.linalg.norm(p2, axis=1, keepdims=True)
    peaks_h = np.concatenate([p1, p2], axis=1)
    atoms_h = rng.integers(0, N, (V, 2)).astype(np.int32)
    nu_h = rng.dirichlet(np.ones(2), V) if nu0 is None else np.tile([nu0, 1 - nu0], (V, 1))
    d_peaks = torch.from_numpy(peaks_h).to(dev)
    d_Y = torch.zeros((V, M), dtype=torch.float64, device=dev)
    for k in range(2):
        col = engine.rotate_columns_dev(plan, d_peaks[:, 3 * k:3 * k + 3].contiguous(), torch.from_numpy(atoms_h[:, k].copy()).to(dev))
        d_Y += 500.0 * torch.from_numpy(nu_h[:, k:k + 1].copy()).to(dev) * col
    if snr:
        gen = torch.Generator(device=dev); gen.manual_seed(99)
        d_Y += torch.randn((V, M), dtype=torch.float64, device=dev, generator=gen) * (500.0 / snr)
    outs = []
    for screen in (1, 0):
        lib.mfx_debug_set_k2_screen(screen)
        out = torch.zeros((V, 7), dtype=torch.float64, device=dev)
        for it in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            L.check(lib.mfx_fit_batch_dev(plan.handle(), d_Y.data_ptr(), d_peaks.data_ptr(), 2, 0, 0, None, None, 0, V, out.data_ptr(), stream.cuda_stream))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        outs.append((out.cpu().numpy(), dt, lib.mfx_debug_last_fallback_count() if screen else 0))
    lib.mfx_debug_set_k2_screen(1)
    ndiff = int((np.abs(outs[0][0] - outs[1][0]).max(axis=1) > 0).sum())
    print("%-44s screening %8.0f voxels/s (handed back %5d of %d) | FP64 kernel %7.0f voxels/s | differing voxels %d"
          % (name, V / outs[0][1], outs[0][2], V, V / outs[1][1], ndiff), flush=True)

run("SNR 30, random crossing (bench)", 30)
run("SNR 10", 10)
run("SNR 100", 100)
run("noise-free", 0)
run("SNR 30, crossing angle 15 deg", 30, 15)
run("SNR 30, crossing angle 3 deg", 30, 3)
run("SNR 30, nu = (0.95, 0.05)", 30, None, 0.95)
run("SNR 30, nu = (1, 0): one fascicle only", 30, None, 1.0)
