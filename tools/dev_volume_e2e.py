"""Row N1 at size: NIfTI files -> MFModel.fit (mixed K in {0, 1, 2}, CSF / EAR masks, ~1e6 ROI voxels at the default
grid) -> write_nifti, with stage times.  MFX_E2E_GRID=128,128,116 (default), MFX_E2E_OUT=<dir> (default: a temp dir)."""
import os, sys, time, json, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import microstructure_fingerprinting_amd as mf
from microstructure_fingerprinting_amd import synth, engine, nifti, _lib as L
from microstructure_fingerprinting_amd import mf as mfmod

grid = tuple(int(x) for x in os.environ.get("MFX_E2E_GRID", "128,128,116").split(","))
sch, dic, rng = synth.make_model("C2")
N, E = dic.shape[1], 10
md = {"dictionary": dic, "sch_mat": sch, "orientation": np.array([0, 0, 1.0]), "num_atom": N, "num_ear": E,
      "T2_csf": 2.0, "DIFF_csf": 3e-9, "T2_ear": 0.08, "DIFF_ear": np.linspace(0.2e-9, 1.2e-9, E),
      "fasc_propnames": ["rad", "fin"], "rad": rng.uniform(0.2e-6, 2e-6, N), "fin": rng.uniform(0.2, 0.9, N)}
model = mf.MFModel(md)
t0 = time.time()
ph = synth.make_phantom(model, grid, rng)
print("phantom %s: %d ROI voxels, data %.2f GB float32, generated in %.1f s" % (grid, int(ph["mask"].sum()), ph["data"].nbytes / 1e9, time.time() - t0), flush=True)
tmp = os.environ.get("MFX_E2E_OUT") or tempfile.mkdtemp(prefix="mfx_e2e_")
os.makedirs(tmp, exist_ok=True)
files = {}
t0 = time.time()
for k, a in ph.items():
    files[k] = os.path.join(tmp, k + ".nii")
    nifti.save(a, np.eye(4), files[k])
np.savetxt(os.path.join(tmp, "scheme.txt"), sch, header="VERSION: 1", comments="")
print("files written in %.1f s" % (time.time() - t0), flush=True)
V = int(ph["mask"].sum())
roi = ph["mask"] > 0
cls = (ph["numfasc"][roi].astype(int) * 4 + (ph["csf_mask"][roi] > 0) * 2 + (ph["ear_mask"][roi] > 0))
mix = {"K%d%s%s" % (q >> 2, "+csf" if q & 2 else "", "+ear" if q & 1 else ""): int(np.sum(cls == q)) for q in range(12)}
print("voxel classes:", mix, flush=True)

stages = {}
def timed(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        t = time.time()
        r = fn(*a, **k)
        stages[label] = stages.get(label, 0.0) + time.time() - t
        return r
    setattr(obj, name, w)
timed(nifti, "load_raw", "open data file (memory map)")
timed(nifti, "load", "read mask / numfasc / peaks / csf / ear files")
timed(engine, "fit_batch_volume", "mfx_fit_batch_volume (upload, gather, all classes, params back)")
timed(engine, "fit_batch", "mfx_fit_batch_rows")
timed(mfmod, "MFModelFit", "maps from parameter rows")

kw = dict(peaks=files["peaks"], pgse_scheme=os.path.join(tmp, "scheme.txt"), csf_mask=files["csf_mask"], ear_mask=files["ear_mask"], verbose=0)
res = {}
for it in range(3):
    stages.clear()
    t0 = time.time()
    fit = model.fit(files["data"], files["mask"], files["numfasc"], **kw)
    t1 = time.time()
    out = fit.write_nifti(os.path.join(tmp, "out.nii"))
    t2 = time.time()
    st = dict(stages)
    st["other host work in fit (ROI indices, peaks gather, checks)"] = (t1 - t0) - sum(stages.values())
    st["write_nifti (%d maps)" % len(out)] = t2 - t1
    print("run %d: files -> fit %.2f s, -> maps on disk %.2f s  (%.0f voxels/s end to end)" % (it, t1 - t0, t2 - t0, V / (t2 - t0)), flush=True)
    for k, v in st.items():
        print("    %-70s %8.3f s" % (k, v), flush=True)
    res = {"grid": grid, "roi_voxels": V, "classes": mix, "fit_s": t1 - t0, "total_s": t2 - t0, "stages_s": st, "voxels_per_s": V / (t2 - t0)}
cnt = [L.lib().mfx_debug_last_counter(q) for q in range(12)]
print("counters of the last call:", cnt)
if os.environ.get("MFX_E2E_SHARD"):
    n = int(os.environ["MFX_E2E_SHARD"])
    model.SHARD_DEVICES = [0] * n if L.lib().mfx_device_count() < n else list(range(n))
    t0 = time.time()
    fit_p = model.fit(files["data"], files["mask"], files["numfasc"], parallel=True, **kw)
    t1 = time.time()
    print("parallel=True over devices %s: %.2f s, identical rows: %s" % (model.SHARD_DEVICES, t1 - t0, np.array_equal(fit_p.params_in_mask, fit.params_in_mask)), flush=True)
    res["sharded"] = {"devices": model.SHARD_DEVICES, "fit_s": t1 - t0}
    model.SHARD_DEVICES = None
# the same volume as host arrays the way the reference gets them: get_fdata() (float64, Fortran order) then data[mask > 0]
t0 = time.time()
full, _ = nifti.load(files["data"])
t1 = time.time()
Yr = np.ascontiguousarray(full[roi])
t2 = time.time()
print("reference-style host preparation for comparison: get_fdata %.2f s, data[mask > 0] %.2f s" % (t1 - t0, t2 - t1), flush=True)
res["host_style_prep_s"] = {"get_fdata": t1 - t0, "roi_gather": t2 - t1}
Kv = ph["numfasc"][roi].astype(np.int32)
t0 = time.time()
stages.clear()
fit_a = model.fit(Yr, np.ones(V), Kv, peaks=ph["peaks"][roi], pgse_scheme=sch, csf_mask=(ph["csf_mask"][roi]), ear_mask=ph["ear_mask"][roi], verbose=0)
t1 = time.time()
print("fit on the gathered [V x M] float64 rows: %.2f s; identical rows: %s" % (t1 - t0, np.array_equal(fit_a.params_in_mask, fit.params_in_mask)), flush=True)
res["rows_path_fit_s"] = t1 - t0
print(json.dumps(res))
if not os.environ.get("MFX_E2E_OUT"):
    shutil.rmtree(tmp, ignore_errors=True)
