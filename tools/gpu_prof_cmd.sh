#!/bin/bash
# usage: tools/gpu_prof_cmd.sh <tag> <python script and args...>: rocprofv3 kernel statistics of one command -> gpurun_out/prof_<tag>/, top kernels printed
set -o pipefail
R=$(pwd); tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/"$1" "${@:2}" > $R/gpurun_out/prof_$tag.log 2>&1 < /dev/null || { echo "profile run failed"; tail -5 $R/gpurun_out/prof_$tag.log; exit 1; }
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv"; exit 2; }
cp "$f" $R/gpurun_out/prof_${tag}_kernel_stats.csv
head -16 "$f" | cut -c1-160
grep -v "^W2026\|amdgpu.ids" $R/gpurun_out/prof_$tag.log | tail -4
