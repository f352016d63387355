#!/usr/bin/env python3
"""Condense tools/profile_run_k3.sh's rocprofv3 outputs (gpurun_out/p3_*) into profiles/r03_kernel_stats_k3.csv and
profiles/r03_pmc_k3_screen.json (the triple screen of fit_k3.hip: 2 calls x 1 batch of 32 voxels at 1500 atoms x 300 rows)."""
import collections, csv, glob, json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def newest(pat):
    fs = sorted(glob.glob(pat), key=os.path.getmtime)
    return fs[-1] if fs else None
f = newest(R + '/gpurun_out/p3_stats/runc/*_kernel_stats.csv')
if f:
    rows = list(csv.reader(open(f)))
    with open(R + '/profiles/r03_kernel_stats_k3.csv', 'w') as o:
        w = csv.writer(o)
        for r in rows[:12]:
            r = list(r); r[0] = r[0][:110]
            w.writerow(r)
res = collections.defaultdict(float)
for d in glob.glob(R + '/gpurun_out/p3_pmc_*/runc'):
    f = newest(d + '/*_counter_collection.csv')
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if 'mfx_k3b_screen_kernel' in r['Kernel_Name']:
            res[r['Counter_Name']] += float(r['Counter_Value'])
V = 2 * 32      # two timed calls of 32 voxels (one batch each)
cyc = res.get('GRBM_GUI_ACTIVE', 0) / 8
out = {"round": 3, "command": "rocprofv3 --pmc <C> --kernel-trace -- python3 tools/dev_time_c5.py 1500 (MFX_DEV_V=32), one pass per counter group",
       "kernel": "mfx_k3b_screen_kernel", "voxels_counted": V, "triples_per_voxel": 1500.0 ** 3, "counters": dict(res),
       "valu_insts_per_voxel": res.get('SQ_INSTS_VALU', 0) / V, "mfma_insts_per_voxel": res.get('SQ_INSTS_MFMA', 0) / V,
       "lds_insts_per_voxel": res.get('SQ_INSTS_LDS', 0) / V, "vmem_read_insts_per_voxel": res.get('SQ_INSTS_VMEM_RD', 0) / V,
       "salu_insts_per_voxel": res.get('SQ_INSTS_SALU', 0) / V,
       "cu_cycles_per_voxel": cyc * 256 / V,
       "mfma_busy_cycles_per_inst": res.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(res.get('SQ_INSTS_MFMA', 1), 1),
       "mfma_pipe_utilisation": (res.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / cyc) if cyc else None,
       "issue_utilisation": ((4.0 * res.get('SQ_INSTS_VALU', 0) + 8.0 * res.get('SQ_INSTS_MFMA', 0)) / (4.0 * cyc * 256)) if cyc else None,
       "issue_utilisation_is": "(4 SQ_INSTS_VALU + 8 SQ_INSTS_MFMA) / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8): share of the vector-issue slots in use (lower bound)",
       "valu_insts_per_1024_triples": res.get('SQ_INSTS_VALU', 0) / max(res.get('SQ_INSTS_MFMA', 1), 1),
       "hbm_bytes_per_voxel_FETCHx2_plus_WRITE": (2 * res.get('FETCH_SIZE', 0) + res.get('WRITE_SIZE', 0)) * 1024.0 / V}
json.dump(out, open(R + '/profiles/r03_pmc_k3_screen.json', 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("command", "counters")}, indent=1))
src = R + '/gpurun_out/p3_stats.txt'
if os.path.exists(src):
    open(R + '/profiles/r03_timing_k3.txt', 'w').write("".join(l for l in open(src) if 'amdgpu.ids' not in l))
