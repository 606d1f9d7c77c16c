#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + PMC passes of the global BA at BASELINE configs[4] size, solved from device-resident store
# records (tools/ba_store_scale.py: 8 clients x KF keyframes, 100 points per keyframe, 3..8 observations -- the problem of bench.py's ba.config5),
# summarised on the box into gpurun_out/<tag>/ (raw DBs stay on the box).  Counters in their own passes (no trace domains next to --pmc).
set -u
TAG=$1; KF=${2:-6250}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; RAW=/tmp/corb_prof_$TAG
mkdir -p $OUT $RAW
export CORB_BA_NO_GRAPH=1      # rocprofv3 crashes inside hipGraphLaunch after a few hundred replays of the captured CG chunk: launch its kernels one by one
CMD="python tools/ba_store_scale.py $KF"
git rev-parse HEAD > $OUT/commit.txt 2>/dev/null || true
timeout 400 rocprofv3 --kernel-trace --stats -d $RAW -o stats -- $CMD > $OUT/cmd_under_rocprof.txt 2> $RAW/stats.log
python tools/rocprof_summary.py $RAW/stats_results.db $OUT/kernel_stats.txt > /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $RAW -o fetch -- $CMD > /dev/null 2> $RAW/fetch.log
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $RAW -o write -- $CMD > /dev/null 2> $RAW/write.log
python tools/pmc_to_json.py $RAW $OUT/pmc_hbm.json > /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY -d $RAW -o sq -- $CMD > /dev/null 2> $RAW/sq.log
python tools/rocprof_summary.py $RAW/sq_results.db $OUT/pmc_sq.txt > /dev/null || tail -5 $RAW/sq.log
env -u CORB_BA_NO_GRAPH $CMD > $OUT/cmd_plain.txt 2>&1       # the product form (captured graph), not profiled: wall / device times to quote
python tools/ba_profile_to_json.py $OUT $OUT/ba_latest.json > /dev/null
ls -la $OUT; head -28 $OUT/kernel_stats.txt; cat $OUT/cmd_plain.txt | tail -2
