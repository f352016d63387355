"""Developer timing of the secondary BASELINE configs through the device-pointer entry."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microstructure_fingerprinting_amd import _lib as L, engine, synth, mf_utils as mfu
if os.environ.get("MFX_DEV_LIB"):   # a diagnostic build of the library (file name inside the package directory)
    L.LIB_PATH = os.path.join(ROOT, "microstructure_fingerprinting_amd", os.environ["MFX_DEV_LIB"])

def run(name, cfg, V, K, c, e, E=10, N=None, bracket=False):
    sch, dic, rng = synth.make_model(cfg, N=N)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, np.array([0, 0, 1.0]))
    if bracket:   # subject protocol with 105 directions, half of them at 4 gradient strengths between the table's shells
        Gs = ms["Gms_un"]
        sch = sch[rng.permutation(sch.shape[0])[:105]].copy()
        nz = np.where(sch[:, 3] > 0)[0]
        between = [0.3 * Gs[1] + 0.7 * Gs[2], 0.55 * Gs[2] + 0.45 * Gs[3], 0.9 * Gs[2] + 0.1 * Gs[3], 0.5 * (Gs[1] + Gs[2])]
        sch[nz[::2], 3] = rng.choice(between, size=nz[::2].size)
        dic = np.stack([mfu.interp_PGSE_from_multishell(sch, np.array([0, 0, 1.0]), msinterp=ms)[:, n] for n in range(0, ms.num_subs, 97)], axis=1)
    plan = ms.plan_for(sch)
    M, Na = sch.shape[0], ms.num_subs
    gam = mfu.get_gyromagnetic_ratio('H')
    b = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3e-9)
    sig_ear = np.stack([np.exp(-sch[:, 6] / 0.08) * np.exp(-b * x) for x in np.linspace(0.2e-9, 1.2e-9, E)], axis=1)
    peaks = np.concatenate([synth.unit_vectors(rng, V) for _ in range(max(K, 1))], axis=1)
    if os.environ.get("MFX_DEV_MIX"):   # every compartment of the class present (Dirichlet fractions), atoms rotated to the peaks: bench.py's recipe
        import bench
        xc = np.concatenate(([sig_csf[:, None]] if c else []) + ([sig_ear[:, 3:4]] if e else []), axis=1) if (c or e) else None
        _, dpk_, dY_ = bench.synth_voxels(plan, V, Na, M, torch.device("cuda", 0), 5, K=max(K, 1), extra_cols=xc)
        Y, peaks = dY_.cpu().numpy(), dpk_.cpu().numpy()
    else:
        Y = 500 * dic[:, rng.integers(0, dic.shape[1], V)].T * rng.uniform(0.5, 1, (V, 1)) + rng.normal(0, 500 / 30, (V, M))
    dY = torch.from_numpy(Y).cuda(); dpk = torch.from_numpy(peaks).cuda()
    dcsf = torch.from_numpy(sig_csf).cuda(); dear = torch.from_numpy(np.ascontiguousarray(sig_ear)).cuda()
    npar = engine.num_params(K, c, e)
    out = torch.zeros((V, npar), dtype=torch.float64, device="cuda")
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        L.check(lib.mfx_fit_batch_dev(plan.handle(), dY.data_ptr(), dpk.data_ptr(), K, c, e, dcsf.data_ptr() if c else None,
                                      dear.data_ptr() if e else None, E if e else 0, V, out.data_ptr(), st))
        torch.cuda.synchronize(); t1 = time.time()
    cn = [lib.mfx_debug_last_counter(q) for q in range(6)]
    print("%-44s V=%6d N=%4d M=%3d: %8.2f ms -> %10.0f voxels/s   counters %s" % (name, V, Na, M, (t1 - t0) * 1e3, V / (t1 - t0), cn), flush=True)

if not os.environ.get("MFX_DEV_K2X_ONLY"):
    run("C1: K=1 [100]", "C1", 100000, 1, 0, 0)
    run("K=1 [782]", "C2", 100000, 1, 0, 0)
    run("K=1 [782,1,10]", "C2", 100000, 1, 1, 1)
    run("C2: K=2 [782,782]", "C2", 100000, 2, 0, 0)
    run("K=2 [782,782], 105-row bracketed protocol", "C2", 100000, 2, 0, 0, bracket=True)
run("K=2+CSF [782,782,1]", "C2", 10000, 2, 1, 0)
run("K=2+EAR [782,782,10]", "C2", 4000, 2, 0, 1)
run("C4: K=2+CSF+EAR [782,782,1,10]", "C2", 4000, 2, 1, 1)
