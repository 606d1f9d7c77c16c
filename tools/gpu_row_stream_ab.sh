#!/bin/bash
# GPU box (round 6): the row-owner Schur products as round 5's per-unit kernel (CORB_BA_ROW_UNITS=1) and as the stream kernel (default; variants/lib_*.so = other builds of it):
# parity tests of the default build first, then device time of the 50 000-keyframe global BA and the kernels' durations under rocprofv3, build by build.
# usage: tools/gpu_row_stream_ab.sh [tag]   -> gpurun_out/<tag>.txt
set -u
TAG=${1:-r06_row_stream_ab}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG.txt; : > $OUT
python -m pytest tests -m gpu -q 2>&1 | tail -6 >> $OUT
cp corb-slam_amd/libcorb_accel.so /tmp/lib_default.so
run() {   # name, env assignment
  echo "== $1" >> $OUT
  env $2 python tools/ba_store_scale.py 6250 2>&1 | grep '^poses' | cut -c1-330 >> $OUT
  RAW=/tmp/rowab_$1; rm -rf $RAW; mkdir -p $RAW
  env $2 CORB_BA_NO_GRAPH=1 timeout 500 rocprofv3 --kernel-trace --stats -d $RAW -o stats -- python tools/ba_store_scale.py 6250 > /dev/null 2> $RAW/log
  python tools/rocprof_summary.py $RAW/stats_results.db $RAW/ks.txt > /dev/null 2>&1
  grep "ba_schur_row\|ba_schur_combine\|ba_backsub\|ba_update_scale\|ba_v_lean\|ba_rr_" $RAW/ks.txt | cut -c1-140 >> $OUT
}
run units CORB_BA_ROW_UNITS=1
run stream X=1
for f in variants/lib_*.so; do
  [ -f "$f" ] || continue
  cp $f corb-slam_amd/libcorb_accel.so
  run $(basename $f .so) X=1
done
cp /tmp/lib_default.so corb-slam_amd/libcorb_accel.so
cat $OUT
