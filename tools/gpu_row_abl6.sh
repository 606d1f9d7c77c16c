#!/bin/bash
# GPU box (round 6): ba_schur_row_stream_kernel with parts of its work left out (variants/lib_abl<mask>.so, -DROW_ABL: 1 second operands from one block, 2 no first-operand
# staging, 4 no matrix instructions; results wrong, timing only): the kernel's average duration under rocprofv3
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r06_row_abl}.txt; : > $OUT
cp corb-slam_amd/libcorb_accel.so /tmp/lib_default.so
for f in /tmp/lib_default.so variants/lib_abl*.so; do
  cp $f corb-slam_amd/libcorb_accel.so
  RAW=/tmp/abl6_$(basename $f .so); rm -rf $RAW; mkdir -p $RAW
  CORB_BA_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $RAW -o s -- python tools/ba_scale.py --pts 100 --obs 3 8 --iters 2 6250 > /dev/null 2> $RAW/log
  python tools/rocprof_summary.py $RAW/s_results.db $RAW/ks.txt > /dev/null 2>&1
  echo "$(basename $f .so): $(grep ba_schur_row_stream $RAW/ks.txt | awk '{print $3, "calls", $5, "ns avg"}')" >> $OUT
done
cp /tmp/lib_default.so corb-slam_amd/libcorb_accel.so
cat $OUT
