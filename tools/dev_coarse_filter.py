"""Developer study (GPU box, torch): how selective would a tuple-independent FIRST test of the CSF/EAR kernel's pair filter be?

The filter of fit_k2x.hip tests every extra tuple t of a pair: a1'.a2' <= (P1 P2 - Q1 Q2) |d1'| |d2'| in the complement of
R_t = {f, x_t}.  A first test with the per-atom maximum / minimum of the constants over t (and max |u_t|) bounds all of
them; this script counts, for config-4 bench voxels at the FINAL threshold, the (16-row tile, 16-column chunk) cells of
the scan in which some pair passes the first test against the cells in which some tuple really passes.
    python tools/dev_coarse_filter.py [voxels]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microstructure_fingerprinting_amd import engine, mf_utils as mfu
import bench

V = int(sys.argv[1]) if len(sys.argv) > 1 else 24
E = 10
sch, dic, ms = bench.build_model(782)
dev = torch.device("cuda", 0)
ms.device = 0
plan = engine.Plan(ms.device_tables(), scheme=sch)
M, N = sch.shape[0], ms.num_subs
gam = mfu.get_gyromagnetic_ratio('H')
b = (gam * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3.0e-9)
sig_ear = np.stack([np.exp(-sch[:, 6] / 0.08) * np.exp(-b * D) for D in np.linspace(0.2e-9, 1.2e-9, E)], axis=1)
xc = np.concatenate([sig_csf[:, None], sig_ear[:, 3:4]], axis=1)
_, dpk, dY = bench.synth_voxels(plan, V, N, M, dev, 5, K=2, extra_cols=xc)
dcsf, dear = torch.from_numpy(sig_csf).to(dev), torch.from_numpy(np.ascontiguousarray(sig_ear)).to(dev)
out = engine.fit_batch_dev(plan, dY, dpk, 2, True, True, dcsf, dear, E)
mse = out[:, -2].cpu().numpy()
cols = torch.arange(N, device=dev, dtype=torch.int32)
f = dcsf / dcsf.norm()
rows = []
for v in range(V):
    y = dY[v]
    T = float((y * y).sum() - mse[v] * M)              # the final best score
    D = [engine.rotate_columns_dev(plan, dpk[v, 3 * k:3 * k + 3].repeat(N, 1).contiguous(), cols).T.contiguous() for k in range(2)]   # [M, N]
    a12 = D[0].T @ D[1]
    uf = [f @ D[k] for k in range(2)]
    accf = a12 - uf[0][:, None] * uf[1][None, :]
    Pn, Qn, U, alw = [[], []], [[], []], [[], []], [[], []]
    for t in range(E):
        x = dear[:, t] - (dear[:, t] @ f) * f
        et = x / x.norm()
        q0 = float((y @ f) ** 2 + (y @ et) ** 2)
        yp = y - (y @ f) * f - (y @ et) * et
        for k in range(2):
            ut = et @ D[k]
            dp = D[k] - f[:, None] * uf[k][None, :] - et[:, None] * ut[None, :]
            n = dp.norm(dim=0)
            z = (yp @ dp) / n
            Tp = T - q0
            always = (z > 0) & (z * z >= Tp) if Tp > 0 else torch.ones_like(z, dtype=torch.bool)
            P = torch.clamp(torch.clamp(z, min=0) / np.sqrt(max(Tp, 1e-30)), max=1.0)
            Q = torch.sqrt(torch.clamp(1 - P * P, min=0))
            Pn[k].append(torch.where(always, torch.full_like(n, 1e18), P * n)); Qn[k].append(torch.where(always, torch.zeros_like(n), Q * n)); U[k].append(ut)
    Pn = [torch.stack(p) for p in Pn]; Qn = [torch.stack(q) for q in Qn]; U = [torch.stack(u) for u in U]      # [E, N]
    passt = torch.zeros((N, N), dtype=torch.bool, device=dev)
    npass = 0
    for t in range(E):
        bt = Pn[0][t][:, None] * Pn[1][t][None, :] - Qn[0][t][:, None] * Qn[1][t][None, :] - (accf - U[0][t][:, None] * U[1][t][None, :])
        passt |= bt >= 0
        npass += int((bt >= 0).sum())
    Pm = [p.max(0).values for p in Pn]; Qm = [q.min(0).values for q in Qn]; Um = [u.abs().max(0).values for u in U]
    coarse = (Pm[0][:, None] * Pm[1][None, :] - Qm[0][:, None] * Qm[1][None, :] - (accf - Um[0][:, None] * Um[1][None, :])) >= 0
    assert bool((coarse | ~passt).all())
    nt = (N + 15) // 16
    grp = {}
    for G in (2, 5):   # the same first test per GROUP of G adjacent tuples: share of (cell, group) combinations that pass
        tot = 0
        for g0 in range(0, E, G):
            sl = slice(g0, g0 + G)
            Pg = [p[sl].max(0).values for p in Pn]; Qg = [q[sl].min(0).values for q in Qn]; Ug = [u[sl].abs().max(0).values for u in U]
            cg = (Pg[0][:, None] * Pg[1][None, :] - Qg[0][:, None] * Qg[1][None, :] - (accf - Ug[0][:, None] * Ug[1][None, :])) >= 0
            pp = torch.zeros((nt * 16, nt * 16), dtype=torch.bool, device=dev); pp[:N, :N] = cg
            tot += int(pp.view(nt, 16, nt, 16).any(3).any(1).sum())
        grp[G] = tot / (nt ** 2 * (E // G))
    def cells(m):
        p = torch.zeros((nt * 16, nt * 16), dtype=torch.bool, device=dev); p[:N, :N] = m
        return int(p.view(nt, 16, nt, 16).any(3).any(1).sum())
    rows.append((npass / (N * N * E), float(passt.float().mean()), float(coarse.float().mean()), cells(passt) / nt ** 2, cells(coarse) / nt ** 2))
    rows[-1] = rows[-1] + (grp[2], grp[5])
    print("voxel %2d: tuples passing %.4f, pairs with a passing tuple %.4f, pairs passing the first test %.4f | cells: real %.3f, first test %.3f | (cell, group) passing: groups of 2 %.3f, of 5 %.3f" % ((v,) + rows[-1]), flush=True)
r = np.array(rows)
print("median / mean over %d voxels: cells that need the tuple loop: really %.3f / %.3f, by the first test %.3f / %.3f; (cell, group) combinations passing a per-group first test: groups of 2 %.3f / %.3f, of 5 %.3f / %.3f" % (V, np.median(r[:, 3]), r[:, 3].mean(), np.median(r[:, 4]), r[:, 4].mean(), np.median(r[:, 5]), r[:, 5].mean(), np.median(r[:, 6]), r[:, 6].mean()))
