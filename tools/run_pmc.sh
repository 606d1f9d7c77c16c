#!/bin/bash
# Collect rocprofv3 PMC counters for the bench workload (run on the GPU box through gpurun).
# Separate passes: SQ issue counters, FETCH_SIZE, WRITE_SIZE (TCC slot limits, MI355X_MICROARCH.md).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1
mkdir -p $OUT
CMD="python bench.py --cpu-frames 0 --steps 4 --warmup 1 --no-profile"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT -o sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT -o lds -- $CMD > $OUT/lds.log 2>&1
ls -la $OUT
