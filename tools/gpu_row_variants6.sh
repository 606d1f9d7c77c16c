#!/bin/bash
# GPU box (round 6): builds of ba_schur_row_stream_kernel under variants/lib_*.so against the default build: the kernel's average duration under rocprofv3 and the device
# time of the 50 000-keyframe global BA (two runs each)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r06_row_variants}.txt; : > $OUT
cp corb-slam_amd/libcorb_accel.so /tmp/lib_default.so
for f in /tmp/lib_default.so variants/lib_*.so; do
  cp $f corb-slam_amd/libcorb_accel.so
  RAW=/tmp/rv6_$(basename $f .so); rm -rf $RAW; mkdir -p $RAW
  CORB_BA_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $RAW -o s -- python tools/ba_scale.py --pts 100 --obs 3 8 --iters 3 6250 > /dev/null 2> $RAW/log
  python tools/rocprof_summary.py $RAW/s_results.db $RAW/ks.txt > /dev/null 2>&1
  echo "$(basename $f .so): $(grep "${KPAT:-ba_schur_row_stream}" $RAW/ks.txt | awk '{print $1, $(NF-3), "ns avg;"}' | tr "\n" " ")  $(python tools/ba_scale.py --pts 100 --obs 3 8 --iters 10 6250 2>&1 | grep -o "device [0-9.]* ms\|'schur': [0-9.]*" | tr '\n' ' ')" >> $OUT
done
cp /tmp/lib_default.so corb-slam_amd/libcorb_accel.so
cat $OUT
