#!/bin/bash
# GPU box: BA parity tests, then the config-5-size global BA under rocprofv3 --kernel-trace --stats (kernel durations only) and once plain
# usage: tools/gpu_ba_quick.sh <tag> [KF per client = 6250] [pytest selection]
set -u
TAG=$1; KF=${2:-6250}; SEL=${3:-tests/test_gpu_ba.py}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; RAW=/tmp/corb_prof_$TAG
mkdir -p $OUT $RAW
timeout 1500 python -m pytest $SEL -x -q 2>&1 | tail -15 | tee $OUT/pytest.txt
python tools/ba_store_scale.py $KF > $OUT/cmd_plain.txt 2>&1; tail -2 $OUT/cmd_plain.txt
export CORB_BA_NO_GRAPH=1
timeout 500 rocprofv3 --kernel-trace --stats -d $RAW -o stats -- python tools/ba_store_scale.py $KF > $OUT/cmd_under_rocprof.txt 2> $RAW/stats.log
python tools/rocprof_summary.py $RAW/stats_results.db $OUT/kernel_stats.txt > /dev/null
head -24 $OUT/kernel_stats.txt
