#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + PMC passes of a global-BA workload (8 clients x N keyframes),
# summarised on the box into gpurun_out/<tag>/ (raw DBs stay on the box).
set -u
TAG=$1; KF=${2:-1250}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; RAW=/tmp/corb_prof_$TAG
mkdir -p $OUT $RAW
CMD="python tools/ba_scale.py $KF"
timeout 240 rocprofv3 --kernel-trace --stats -d $RAW -o stats -- $CMD > $OUT/ba_scale_under_rocprof.txt 2> $RAW/stats.log
python tools/rocprof_summary.py $RAW/stats_results.db $OUT/kernel_stats.txt > /dev/null
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $RAW -o fetch -- $CMD > /dev/null 2> $RAW/fetch.log
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $RAW -o write -- $CMD > /dev/null 2> $RAW/write.log
python tools/pmc_to_json.py $RAW $OUT/pmc_hbm.json > /dev/null
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY -d $RAW -o sq -- $CMD > /dev/null 2> $RAW/sq.log
python tools/rocprof_summary.py $RAW/sq_results.db $OUT/pmc_sq.txt > /dev/null || tail -5 $RAW/sq.log
ls -la $OUT
timeout 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr -d $RAW -o tcc -- $CMD > /dev/null 2> $RAW/tcc.log
python tools/rocprof_summary.py $RAW/tcc_results.db $OUT/pmc_tcc.txt > /dev/null || tail -5 $RAW/tcc.log
