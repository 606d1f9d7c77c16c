#!/bin/bash
# GPU box: the row-owner Schur kernel's (wavefronts, range) variants built into variants/lib_w<W>_r<R>.so (make EXTRA="-DBA_ROW_WAVES=W -DBA_ROW_RANGE=R"):
# device time of the 50 000-keyframe global BA and the kernel's average duration under rocprofv3, variant by variant
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp corb-slam_amd/libcorb_accel.so /tmp/lib_default.so
for f in /tmp/lib_default.so variants/lib_*.so; do
  cp $f corb-slam_amd/libcorb_accel.so
  RAW=/tmp/rowvar_$(basename $f .so); mkdir -p $RAW
  python tools/ba_store_scale.py 6250 2>&1 | tail -1 | cut -c1-200
  CORB_BA_NO_GRAPH=1 timeout 500 rocprofv3 --kernel-trace --stats -d $RAW -o stats -- python tools/ba_store_scale.py 6250 > /dev/null 2> $RAW/log
  python tools/rocprof_summary.py $RAW/stats_results.db $RAW/ks.txt > /dev/null
  echo "== $f"; grep "ba_schur_row_kernel\|ba_schur_combine" $RAW/ks.txt | cut -c1-130
done
cp /tmp/lib_default.so corb-slam_amd/libcorb_accel.so
