#!/bin/bash
# Run on the GPU box (through gpurun): HIP API call statistics (host side) of a command -> gpurun_out/<tag>/hip_api_stats.txt.  usage: gpu_hiptrace.sh TAG cmd...
set -u
TAG=$1; shift 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; RAW=/tmp/corb_hip_$TAG
mkdir -p $OUT $RAW
timeout 600 rocprofv3 --hip-trace --stats -f csv -d $RAW -o hip -- "$@" > $OUT/cmd_under_rocprof.txt 2> $RAW/hip.log
ls $RAW | head
f=$(ls $RAW/*hip_api_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp $f $OUT/hip_api_stats.csv; head -30 $f | cut -c1-160; else tail -5 $RAW/hip.log; fi
