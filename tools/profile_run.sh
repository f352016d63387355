#!/bin/bash
# Profiling session on the GPU box (run through gpurun from the repo root): kernel-trace statistics of the bench
# command, then one rocprofv3 --pmc pass per counter group (FETCH_SIZE and WRITE_SIZE never together), each under
# its own timeout.  tools/profile_summary.py condenses the outputs into profiles/.
set -o pipefail
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_pmc_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_stats_bench.json 2> $R/gpurun_out/prof_stats.err || { echo FAILED stats; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  tag=${grp%% *}
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/prof_pmc_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/prof_pmc_$tag.err || { echo FAILED $tag; exit 1; }
  echo done $tag
done
