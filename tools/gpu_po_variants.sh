#!/bin/bash
# GPU box: pose_opt_kernel built with other workgroup sizes / cached edges per thread (variants/lib_po*.so) against the default build (tools/po_probe.py: wall per call at
# 400 / 1 750 observations, three runs each)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r06_po_variants}.txt; : > $OUT
cp corb-slam_amd/libcorb_accel.so /tmp/lib_default.so
for f in /tmp/lib_default.so variants/lib_po*.so; do
  cp $f corb-slam_amd/libcorb_accel.so
  for r in 1 2 3; do echo "$(basename $f .so): $(python tools/po_probe.py 2>&1 | tr '\n' ' ')" >> $OUT; done
done
cp /tmp/lib_default.so corb-slam_amd/libcorb_accel.so
cat $OUT
