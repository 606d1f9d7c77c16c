#!/bin/bash
# one SQ PMC pass of the bench workload (run on the GPU box through gpurun): instruction mix and waits per kernel -> gpurun_out/pmc3_<tag>/sq.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc3_$1; RAW=/tmp/pmc3_$1
mkdir -p $OUT $RAW
CMD="python bench.py --cpu-frames 0 --ba-cpu-kf 0 --ba-kf 0 --replay-frames 0 --steps 4 --warmup 1 --no-profile --batch 64"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $RAW -o sq -- $CMD > $RAW/sq.log 2>&1
python tools/rocprof_summary.py $RAW/sq_results.db $OUT/sq.txt > /dev/null || tail -5 $RAW/sq.log
grep "orb_fast\|orb_describe\|orb_octree" $OUT/sq.txt
