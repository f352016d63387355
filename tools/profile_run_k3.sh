#!/bin/bash
# Profiling session of the batched three-fascicle path (fit_k3.hip) at config 5's size: kernel statistics + PMC counters of
# tools/dev_time_c5.py 1500 (32 voxels = one batch, two timed calls); tools/profile_summary_k3.py condenses them into profiles/r03_*.
set -o pipefail
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/p3_*
export MFX_DEV_V=32
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p3_stats -- python3 $R/tools/dev_time_c5.py 1500 > $R/gpurun_out/p3_stats.txt 2> $R/gpurun_out/p3_stats.err < /dev/null || { echo FAILED stats k3; exit 1; }
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=${grp%% *}
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/p3_pmc_$tag -- python3 $R/tools/dev_time_c5.py 1500 > /dev/null 2> $R/gpurun_out/p3_pmc_$tag.err < /dev/null || { echo FAILED $tag; exit 1; }
  echo done k3 $tag
done
