#!/bin/bash
# round 3, call 2: MFMA summation model, full GPU suite after the shared-source refactor + audit, bench
set -o pipefail
mkdir -p gpurun_out
tools/micro/bin/mfma_sum_model > gpurun_out/r3_mfma_sum_model.txt 2>&1 || { tail -5 gpurun_out/r3_mfma_sum_model.txt; exit 1; }
cat gpurun_out/r3_mfma_sum_model.txt
python -m pytest tests -m gpu -x -q -s > gpurun_out/r3_t1.log 2>&1; rc=$?; grep -E "audit|passed|failed|Error|error" gpurun_out/r3_t1.log | tail -15; [ $rc -eq 0 ] || { tail -30 gpurun_out/r3_t1.log; exit $rc; }
python bench.py --steps 5 --warmup 2 > gpurun_out/r3_b1.json 2> gpurun_out/r3_b1.err || exit 3
python - <<PY
import json; r=json.load(open("gpurun_out/r3_b1.json")); print(r["value"], r["roofline"]["kernel_ms"], r["roofline"]["screen_audit"], r["fp64_kernel"]["outputs_identical_to_default_path"])
PY
