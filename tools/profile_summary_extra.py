#!/usr/bin/env python3
"""Condense tools/profile_run_extra.sh's rocprofv3 outputs (gpurun_out/px_*) into profiles/r03_*: kernel statistics of the
two-fascicle + CSF/EAR classes and of the wide screening kernel, PMC counters per k2x launch class."""
import collections, csv, glob, json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def newest(pat):
    fs = sorted(glob.glob(pat), key=os.path.getmtime)
    return fs[-1] if fs else None
for tag in ("k2x", "wide"):
    f = newest(R + '/gpurun_out/px_stats_%s/runc/*_kernel_stats.csv' % tag)
    if f:
        rows = list(csv.reader(open(f)))
        with open(R + '/profiles/r03_kernel_stats_%s.csv' % tag, 'w') as o:
            w = csv.writer(o)
            for r in rows[:9]:
                r = list(r); r[0] = r[0][:110]
                w.writerow(r)
# PMC: the tool launches the k2x kernel for three classes, 3 timed calls each, in chunks of 2048 workgroups; sum per class
# by dispatch order: class boundaries from the kernel trace are not needed - the tool prints voxel counts (10000, 4000, 4000)
res = collections.defaultdict(lambda: collections.defaultdict(float))
for d in glob.glob(R + '/gpurun_out/px_pmc_k2x_*/runc'):
    f = newest(d + '/*_counter_collection.csv')
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f)) if 'mfx_fit_k2x_kernel' in r['Kernel_Name']]
    # group dispatches: 3 calls x ceil(10000/2048)=5 launches, then 3 x 2, then 3 x 2 (dispatch ids ascending)
    ids = sorted(set(int(r['Dispatch_Id']) for r in rows))
    cls_of = {}
    # the two classes with EAR columns run 3 calls x 2 launches of 2 048 voxels each, at the end; whatever comes before them is
    # the CSF class (screening pipeline: list-mode launches per 32 768 voxels + the hand-back launches)
    for i in ids[-6:]:
        cls_of[i] = "csf_ear"
    for i in ids[-12:-6]:
        cls_of[i] = "ear"
    for i in ids[:-12]:
        cls_of[i] = "csf"
    for r in rows:
        c = cls_of.get(int(r['Dispatch_Id']))
        if c:
            res[c][r['Counter_Name']] += float(r['Counter_Value'])
out = {"round": 3, "command": "rocprofv3 --pmc <C> --kernel-trace -- python3 tools/dev_time_configs.py (MFX_DEV_K2X_ONLY=1 MFX_DEV_MIX=1), one pass per counter group",
       "kernel": "mfx_fit_k2x_kernel<50,false,8,2>", "classes": {}}
vox = {"csf": 3 * 10000, "ear": 3 * 4000, "csf_ear": 3 * 4000}
for c, cnt in res.items():
    V = vox[c]
    cyc = cnt.get('GRBM_GUI_ACTIVE', 0) / 8
    out["classes"][c] = {"voxels_counted": V, "counters": dict(cnt),
                         "valu_insts_per_voxel": cnt.get('SQ_INSTS_VALU', 0) / V, "mfma_insts_per_voxel": cnt.get('SQ_INSTS_MFMA', 0) / V,
                         "lds_insts_per_voxel": cnt.get('SQ_INSTS_LDS', 0) / V, "vmem_read_insts_per_voxel": cnt.get('SQ_INSTS_VMEM_RD', 0) / V,
                         "salu_insts_per_voxel": cnt.get('SQ_INSTS_SALU', 0) / V, "lds_bank_conflict_cycles_per_voxel": cnt.get('SQ_LDS_BANK_CONFLICT', 0) / V,
                         "cu_cycles_per_voxel": cyc * 256 / V if V else None,
                         "mfma_busy_cycles_per_inst": cnt.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(cnt.get('SQ_INSTS_MFMA', 1), 1),
                         "mfma_pipe_utilisation": (cnt.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / cyc) if cyc else None,
                         "hbm_bytes_per_voxel_FETCHx2_plus_WRITE": (2 * cnt.get('FETCH_SIZE', 0) + cnt.get('WRITE_SIZE', 0)) * 1024.0 / V,
                         "issue_utilisation": ((4.0 * cnt.get('SQ_INSTS_VALU', 0) + 8.0 * cnt.get('SQ_INSTS_MFMA', 0)) / (4.0 * cyc * 256)) if cyc else None}
# what bench.py quotes for config 4 (the csf_ear class), at the top level; issue_utilisation = (4 VALU + 8 MFMA wave-instructions)
# / (4 SIMDs x CU-cycles): the share of vector-issue slots in use, a lower bound (FP64 instructions take more than 4 cycles)
c4 = out["classes"].get("csf_ear", {})
out["issue_utilisation"] = c4.get("issue_utilisation")
out["mfma_pipe_utilisation"] = c4.get("mfma_pipe_utilisation")
out["hbm_bytes_per_voxel"] = c4.get("hbm_bytes_per_voxel_FETCHx2_plus_WRITE")
json.dump(out, open(R + '/profiles/r03_pmc_k2x.json', 'w'), indent=1)
print(json.dumps({c: {k: v for k, v in d.items() if k != "counters"} for c, d in out["classes"].items()}, indent=1))
for tag in ("k2x", "wide"):
    src = R + '/gpurun_out/px_stats_%s.txt' % tag
    if os.path.exists(src):
        open(R + '/profiles/r03_timing_%s.txt' % tag, 'w').write("".join(l for l in open(src) if 'amdgpu.ids' not in l))
