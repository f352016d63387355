#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of a profiling session (gpurun_out/prof_stats, gpurun_out/prof_pmc_*) into the
files kept under profiles/ (kernel stats CSV head + PMC summary JSON for the dominant kernel).
    python tools/profile_summary.py [kernel-name-substring] [tag] [round]"""
import collections, csv, glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kern = sys.argv[1] if len(sys.argv) > 1 else "mfx_fit_k2s_kernel"
tag = sys.argv[2] if len(sys.argv) > 2 else "k2s"
rnd = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = collections.defaultdict(float)
# gpurun merges new outputs into gpurun_out/ without removing older ones: take the newest run of every counter group
for d in glob.glob(R + '/gpurun_out/prof_pmc_*/runc'):
    fs = sorted(glob.glob(d + '/*_counter_collection.csv'), key=os.path.getmtime)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[-1])):
        if kern in r['Kernel_Name']:
            res[r['Counter_Name']] += float(r['Counter_Value'])
V = 100000
fetch_raw = res['FETCH_SIZE'] * 1024.0
write = res['WRITE_SIZE'] * 1024.0
cyc_xcd = res['GRBM_GUI_ACTIVE'] / 8
out = {"round": rnd,
       "command": "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
                  "--no-cpu-baseline (one pass per counter group; FETCH_SIZE and WRITE_SIZE in separate passes)",
       "kernel": kern, "voxels_per_launch": V, "counters": dict(res),
       "FETCH_SIZE_bytes_raw": fetch_raw, "WRITE_SIZE_bytes": write,
       "hbm_bytes_per_launch": 2 * fetch_raw + write, "hbm_bytes_per_launch_uncorrected": fetch_raw + write,
       "algorithmic_bytes_per_launch": 1710.0 * V,
       "note_traffic": "FETCH x2 per the gfx950 correction of MI355X_MICROARCH.md",
       "mfma_insts_per_voxel": res['SQ_INSTS_MFMA'] / V,
       "mfma_busy_cycles_per_inst": res['SQ_VALU_MFMA_BUSY_CYCLES'] / max(res['SQ_INSTS_MFMA'], 1),
       "valu_insts_per_voxel": res['SQ_INSTS_VALU'] / V, "lds_insts_per_voxel": res['SQ_INSTS_LDS'] / V,
       "vmem_read_insts_per_voxel": res.get('SQ_INSTS_VMEM_RD', 0) / V, "salu_insts_per_voxel": res.get('SQ_INSTS_SALU', 0) / V,
       "kernel_cycles_per_xcd": cyc_xcd, "cu_cycles_per_voxel": cyc_xcd * 256 / V,
       "mfma_pipe_utilisation": res['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc_xcd,
       "valu_insts_per_mfma": res['SQ_INSTS_VALU'] / max(res['SQ_INSTS_MFMA'], 1),
       # vector-issue slots in use: a wave64 vector instruction takes 4 issue cycles of its SIMD (MI355X_MICROARCH.md, 'vector-
       # instruction ISSUE cost'; FP64 and transcendental ones more - a lower bound), an MFMA 8; 4 SIMDs x CU-cycles are available
       "issue_utilisation": (4.0 * res['SQ_INSTS_VALU'] + 8.0 * res['SQ_INSTS_MFMA']) / max(4.0 * cyc_xcd * 256, 1),
       "issue_utilisation_is": "(4 SQ_INSTS_VALU + 8 SQ_INSTS_MFMA) / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8)",
       "lds_bank_conflict_cycles_per_voxel": res.get('SQ_LDS_BANK_CONFLICT', 0) / V,
       "lds_bank_conflict_frac_of_lds_active": res.get('SQ_LDS_BANK_CONFLICT', 0) / max(res.get('SQ_LDS_IDX_ACTIVE', 0), 1)}
json.dump(out, open(R + '/profiles/r%02d_pmc_traffic_%s.json' % (rnd, tag), 'w'), indent=1)
ks = sorted(glob.glob(R + '/gpurun_out/prof_stats/runc/*_kernel_stats.csv'), key=os.path.getmtime)
if ks:
    rows = list(csv.reader(open(ks[-1])))
    with open(R + '/profiles/r%02d_kernel_stats_%s.csv' % (rnd, tag), 'w') as f:
        w = csv.writer(f)
        for r in rows[:6]:
            r = list(r); r[0] = r[0][:120]
            w.writerow(r)
for src, dst in (("bench_final.json", "r%02d_bench_%s.json" % (rnd, tag)), ("prof_stats_bench.json", "r%02d_bench_%s_under_rocprof.json" % (rnd, tag))):
    if os.path.exists(R + '/gpurun_out/' + src):
        shutil.copy(R + '/gpurun_out/' + src, R + '/profiles/' + dst)
print(json.dumps({k: v for k, v in out.items() if k not in ("command", "counters")}, indent=1))
