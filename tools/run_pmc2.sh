#!/bin/bash
# latency / occupancy / dispatch counters (second PMC recipe)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc2_$1
mkdir -p $OUT
CMD="python bench.py --cpu-frames 0 --steps 4 --warmup 1 --no-profile"
rocprofv3 --kernel-trace --pmc SQ_LEVEL_WAVES SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM -d $OUT -o lat -- $CMD > $OUT/lat.log 2>&1
rocprofv3 --kernel-trace --pmc SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN SPI_CSN_BUSY SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN -d $OUT -o spi -- $CMD > $OUT/spi.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH -d $OUT -o ins -- $CMD > $OUT/ins.log 2>&1
tail -2 $OUT/spi.log
