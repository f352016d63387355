#!/bin/bash
# round 3, first GPU call: bench of the starting state, margin / generation / tail experiments, 2-rank launcher rehearsal, full GPU suite
set -o pipefail
mkdir -p gpurun_out
python bench.py --steps 5 --warmup 2 > gpurun_out/r3_b0.json 2> gpurun_out/r3_b0.err || exit 1
echo "bench done"; tail -c 600 gpurun_out/r3_b0.json
for v in dc2 dc4 nogen notail nogentail; do
  python tools/dev_bench_lib.py microstructure_fingerprinting_amd/libmfx_exp_$v.so --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r3_exp_$v.json 2> gpurun_out/r3_exp_$v.err || exit 2
  python - <<PY
import json; r=json.load(open("gpurun_out/r3_exp_$v.json")); print("$v", r["value"], r["roofline"]["kernel_ms"], r["roofline"]["handed_back_to_fp64_kernel"])
PY
done
MFX_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r3_2rank_gloo.json 2> gpurun_out/r3_2rank_gloo.err || exit 3
echo "2-rank rehearsal done"; cat gpurun_out/r3_2rank_gloo.json | cut -c1-400
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t0.log 2>&1; rc=$?; tail -15 gpurun_out/r3_t0.log; exit $rc
