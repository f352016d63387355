#!/bin/bash
# round 3: batched three-fascicle path (fit_k3.hip): parity tests, then timing
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_parity_stress_gpu.py -m gpu -x -q -k "three or c5 or rotate_atom_plan" > gpurun_out/r3_k3_t.log 2>&1; rc=$?; tail -6 gpurun_out/r3_k3_t.log; [ $rc -eq 0 ] || exit 1
MFX_DEV_V=${MFX_DEV_V:-64} timeout -k 10 300 python tools/dev_time_c5.py 400 1500 2>&1 | grep -v amdgpu.ids || exit 2
