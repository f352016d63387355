#!/bin/bash
# round 3: the whole GPU suite, the smoke entry, then the default bench (what the driver runs at round end)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_full_t.log 2>&1; rc=$?; tail -4 gpurun_out/r3_full_t.log; [ $rc -eq 0 ] || { tail -40 gpurun_out/r3_full_t.log; exit $rc; }
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 2
python bench.py > gpurun_out/r3_full_b.json 2> gpurun_out/r3_full_b.err || { tail -20 gpurun_out/r3_full_b.err; exit 3; }
python - <<PY
import json; r=json.load(open("gpurun_out/r3_full_b.json"))
print("headline", r["value"], r["unit"], "kernel_ms", r["roofline"]["kernel_ms"], "frac", r["roofline"]["frac"], "audit", r["roofline"].get("screen_audit"))
for k in ("fp64_kernel", "k2_csf", "c4", "c5", "host_api", "cpu_baseline", "wide", "k2x_ear"):
    if k in r: print(k, {a: b for a, b in r[k].items() if a in ("value", "unit", "ms_per_voxel", "outputs_identical_to_default_path", "cores", "sample")})
print([k for k in r.keys()])
PY
