#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats only of a global-BA workload (8 clients x N keyframes) -> gpurun_out/<tag>/kernel_stats.txt
set -u
TAG=$1; KF=${2:-1250}; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; RAW=/tmp/corb_prof_$TAG
mkdir -p $OUT $RAW
CMD="python tools/ba_scale.py $* $KF"
rocprofv3 --kernel-trace --stats -d $RAW -o stats -- $CMD > $OUT/ba_scale_under_rocprof.txt 2> $RAW/stats.log
python tools/rocprof_summary.py $RAW/stats_results.db $OUT/kernel_stats.txt > /dev/null
head -30 $OUT/kernel_stats.txt; cat $OUT/ba_scale_under_rocprof.txt
