"""One-off referee run of the volume path: a phantom volume written as NIfTI files (mixed K in {0, 1, 2}, CSF / EAR masks,
782 atoms, 10 EAR columns), MFModel.fit from the files, and the CPU oracle on a random sample of the ROI voxels (all twelve
voxel classes in the phantom's proportions).    python tools/dev_volume_vs_oracle.py [sample=10000] [threads=16]"""
import os, sys, time, json, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import microstructure_fingerprinting_amd as mf
from microstructure_fingerprinting_amd import synth, nifti
from oracle import oracle as orc

NS = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 16
grid = (64, 64, 58)
sch, dic, rng = synth.make_model("C2")
N, E = dic.shape[1], 10
md = {"dictionary": dic, "sch_mat": sch, "orientation": np.array([0, 0, 1.0]), "num_atom": N, "num_ear": E,
      "T2_csf": 2.0, "DIFF_csf": 3e-9, "T2_ear": 0.08, "DIFF_ear": np.linspace(0.2e-9, 1.2e-9, E),
      "fasc_propnames": ["rad", "fin"], "rad": rng.uniform(0.2e-6, 2e-6, N), "fin": rng.uniform(0.2, 0.9, N)}
model = mf.MFModel(md)
ph = synth.make_phantom(model, grid, rng)
tmp = tempfile.mkdtemp(prefix="mfx_vo_")
files = {}
for k, a in ph.items():
    files[k] = os.path.join(tmp, k + ".nii")
    nifti.save(a, np.eye(4), files[k])
fit = model.fit(files["data"], files["mask"], files["numfasc"], peaks=files["peaks"], pgse_scheme=sch, csf_mask=files["csf_mask"],
                ear_mask=files["ear_mask"], verbose=0)
shutil.rmtree(tmp, ignore_errors=True)
roi = ph["mask"] > 0
V = int(roi.sum())
Kv = ph["numfasc"][roi].astype(np.int32); cm = ph["csf_mask"][roi] > 0; em = ph["ear_mask"][roi] > 0
Yr = np.ascontiguousarray(ph["data"][roi], dtype=np.float64)
pk = np.ascontiguousarray(ph["peaks"][roi])
pick = np.sort(rng.choice(V, size=min(NS, V), replace=False))
cls = Kv * 4 + cm * 2 + em
print("ROI %d voxels; sample %d: per class %s" % (V, pick.size, {int(q): int(np.sum(cls[pick] == q)) for q in range(12)}), flush=True)
ms = model.ms_interpolator
T = {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat}
b = (orc.GAMMA_H * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
sig_csf = np.exp(-sch[:, 6] / 2.0) * np.exp(-b * 3e-9)
sig_ear = np.ascontiguousarray(np.stack([np.exp(-sch[:, 6] / 0.08) * np.exp(-b * x) for x in md["DIFF_ear"]], axis=1))
bad, worst, t0 = 0, 0.0, time.time()
# slow classes last, in small blocks (progress lines)
order = np.argsort((cls[pick] == 11) * 2 + ((cls[pick] == 9)), kind="stable")
pick = pick[order]
v0 = 0
while v0 < pick.size:
    n = 16 if cls[pick[v0]] == 11 else (200 if cls[pick[v0]] == 9 else 1000)
    sel = pick[v0:v0 + n]
    ref = orc.fit_batch(T, sch, Yr[sel], Kv[sel], cm[sel].astype(np.uint8), em[sel].astype(np.uint8), pk[sel], 2, True, True, sig_csf, sig_ear, E, nthreads=NT)
    g = fit.params_in_mask[sel].copy()
    for col_nu, col_id in ((1, 3), (2, 4), (6, 7)):
        off = ref[:, col_nu] <= 1e-9
        g[off, col_id] = 0; ref[off, col_id] = 0
    d = np.where(np.any(g[:, [3, 4, 7]] != ref[:, [3, 4, 7]], axis=1))[0]
    bad += d.size
    worst = max(worst, float(np.max(np.abs(g - ref) / np.maximum(np.abs(ref), 1e-12))))
    v0 += sel.size
    print("%6d of %d sampled voxels checked: %d with different indices, worst relative difference %.3e, %.0f s" % (v0, pick.size, bad, worst, time.time() - t0), flush=True)
print(json.dumps({"workload": "phantom volume %s, NIfTI files -> MFModel.fit; oracle on a random sample of the ROI" % (grid,), "roi_voxels": V,
                  "sampled": int(pick.size), "voxels_with_different_indices": int(bad), "worst_relative_difference": worst,
                  "oracle_seconds": round(time.time() - t0, 1)}))
