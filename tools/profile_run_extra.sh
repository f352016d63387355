#!/bin/bash
# Profiling session for the kernels bench.py's headline line does not show (run through gpurun from the repo root):
# kernel-trace statistics + PMC counters of the two-fascicle + CSF/EAR classes (tools/dev_time_configs.py) and of the
# wide screening kernel on long protocols (tools/dev_time_wide.py).  One rocprofv3 --pmc pass per counter group.
set -o pipefail
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/px_*
export MFX_DEV_K2X_ONLY=1 MFX_DEV_MIX=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/px_stats_k2x -- python3 $R/tools/dev_time_configs.py > $R/gpurun_out/px_stats_k2x.txt 2> $R/gpurun_out/px_stats_k2x.err || { echo FAILED stats k2x; exit 1; }
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=${grp%% *}
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/px_pmc_k2x_$tag -- python3 $R/tools/dev_time_configs.py > /dev/null 2> $R/gpurun_out/px_pmc_k2x_$tag.err || { echo FAILED $tag; exit 1; }
  echo done k2x $tag
done
unset MFX_DEV_K2X_ONLY MFX_DEV_MIX
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/px_stats_wide -- python3 $R/tools/dev_time_wide.py > $R/gpurun_out/px_stats_wide.txt 2> $R/gpurun_out/px_stats_wide.err || { echo FAILED stats wide; exit 1; }
echo done wide
