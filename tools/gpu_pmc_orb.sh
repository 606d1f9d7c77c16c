#!/bin/bash
# Run on the GPU box (through gpurun): one rocprofv3 --pmc pass per counter group of the timed region of bench.py, summarised per kernel
# into gpurun_out/<tag>/pmc_<i>.txt.  usage: gpu_pmc_orb.sh TAG "CTR1 CTR2 ..." ["CTR3 ..."]
set -u
TAG=$1; shift 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; RAW=/tmp/corb_prof_$TAG
mkdir -p $OUT $RAW
CMD="python bench.py --no-extras --no-profile --steps 32 --warmup 1"
i=0
for grp in "$@"; do
  timeout 150 rocprofv3 --kernel-trace --pmc $grp -d $RAW -o p$i -- $CMD > /dev/null 2> $RAW/p$i.log
  python tools/rocprof_summary.py $RAW/p${i}_results.db $OUT/pmc_$i.txt > /dev/null || tail -5 $RAW/p$i.log
  i=$((i+1))
done
