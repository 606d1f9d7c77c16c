#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats of an arbitrary command -> gpurun_out/<tag>/kernel_stats.txt.  usage: gpu_profile_cmd.sh TAG cmd...
set -u
TAG=$1; shift 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; RAW=/tmp/corb_prof_$TAG
mkdir -p $OUT $RAW
timeout 600 rocprofv3 --kernel-trace --stats -d $RAW -o stats -- "$@" > $OUT/cmd_under_rocprof.txt 2> $RAW/stats.log
python tools/rocprof_summary.py $RAW/stats_results.db $OUT/kernel_stats.txt > /dev/null
head -45 $OUT/kernel_stats.txt | cut -c1-140
