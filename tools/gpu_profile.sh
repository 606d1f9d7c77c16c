#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + PMC passes of the bench workload, summarised
# on the box into small text/json files under gpurun_out/<tag>/ (the raw sqlite DBs stay on the box).
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; RAW=/tmp/corb_prof_$TAG
mkdir -p $OUT $RAW
CMD="python bench.py --no-extras --steps 32 --warmup 4"
timeout 240 rocprofv3 --kernel-trace --stats -d $RAW -o stats -- $CMD > $OUT/bench_under_rocprof.json 2> $RAW/stats.log
python tools/rocprof_summary.py $RAW/stats_results.db $OUT/kernel_stats.txt > /dev/null
CMD="python bench.py --no-extras --steps 32 --warmup 1 --no-profile"      # same frames per step (hence images per launch) as the default bench line
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $RAW -o fetch -- $CMD > /dev/null 2> $RAW/fetch.log
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $RAW -o write -- $CMD > /dev/null 2> $RAW/write.log
python tools/pmc_to_json.py $RAW $OUT/pmc_hbm.json > /dev/null
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $RAW -o sq -- $CMD > /dev/null 2> $RAW/sq.log
python tools/rocprof_summary.py $RAW/sq_results.db $OUT/pmc_sq.txt > /dev/null
ls -la $OUT
