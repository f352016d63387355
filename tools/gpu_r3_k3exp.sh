#!/bin/bash
# timing experiments on the three-fascicle screen (wrong results by design): which part of a block costs what
set -o pipefail
for e in "" NOBUILD NOMFMA NOHIT NOSCORE; do
  lib=microstructure_fingerprinting_amd/libmfx.so
  [ -n "$e" ] && lib=microstructure_fingerprinting_amd/libmfx_exp_k3_$e.so
  echo "== ${e:-product}"
  MFX_DEV_LIB=$lib MFX_DEV_V=32 timeout -k 10 200 python tools/dev_time_c5.py 1500 2>&1 | grep -v amdgpu.ids || exit 2
done
