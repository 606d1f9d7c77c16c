#!/bin/bash
# GPU box: device / wall time of the 50 000-keyframe global BA for library variants in variants/*.so (make EXTRA=...; e.g. -DBA_PCG_CHUNK_BIG=n builds) and, with
# variants/lib_dev.so (corb_ba.cpp built with -DCORB_DEV), for the preconditioner refresh periods given as arguments (default 1 2 3 5)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp corb-slam_amd/libcorb_accel.so /tmp/lib_default.so
for f in /tmp/lib_default.so variants/lib_*.so; do
  [ -f $f ] || continue; [ $f = variants/lib_dev.so ] && continue
  cp $f corb-slam_amd/libcorb_accel.so
  echo "== $f"; python tools/ba_store_scale.py 6250 2>&1 | tail -1 | cut -c60-330
done
if [ -f variants/lib_dev.so ]; then
  cp variants/lib_dev.so corb-slam_amd/libcorb_accel.so
  for p in ${@:-1 2 3 5}; do echo "== pc_period $p"; CORB_BA_PC_PERIOD=$p python tools/ba_store_scale.py 6250 2>&1 | tail -1 | cut -c60-330; done
fi
cp /tmp/lib_default.so corb-slam_amd/libcorb_accel.so
