#!/bin/bash
# GPU box: kernel timeline of the steady state of the timed region -> gpurun_out/<tag>/timeline.txt
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; RAW=/tmp/corb_prof_$TAG
mkdir -p $OUT $RAW
timeout 240 rocprofv3 --kernel-trace -d $RAW -o tl -- python bench.py --no-extras --no-profile --steps 32 --warmup 4 > /dev/null 2> $RAW/tl.log
python tools/rocprof_timeline.py $RAW/tl_results.db $OUT/timeline.txt > /dev/null || tail -5 $RAW/tl.log
