#!/bin/bash
# round 3: slab / LDS-DMA form of the screening kernel: correctness first (bounded by timeouts: the kernel takes locks), then A/B timing
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python tools/dev_check_bench_parity.py > gpurun_out/r3_slab_parity.txt 2>&1; rc=$?; tail -5 gpurun_out/r3_slab_parity.txt; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python -m pytest tests/test_fit_gpu.py tests/test_parity_stress_gpu.py -m gpu -x -q > gpurun_out/r3_t2.log 2>&1; rc=$?; tail -5 gpurun_out/r3_t2.log; [ $rc -eq 0 ] || exit 2
for s in 1 0 1 0; do
  MFX_K2S_SLAB=$s timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r3_slab$s.json 2> gpurun_out/r3_slab$s.err || exit 3
  python - <<PY
import json; r=json.load(open("gpurun_out/r3_slab$s.json")); print("slab=$s", r["value"], r["roofline"]["kernel_ms"], r["roofline"]["handed_back_to_fp64_kernel"], r["roofline"]["screen_audit"])
PY
done
