#!/bin/bash
# GPU box: the timed region of the headline step with every library build under variants/ (make EXTRA=-D...), twice each, alternating
cd "$GRAFT_REPO_ROOT"
cp corb-slam_amd/libcorb_accel.so /tmp/lib_keep.so
for rep in 1 2; do for f in variants/lib_*.so; do
  cp $f corb-slam_amd/libcorb_accel.so
  timeout 200 python bench.py --no-extras --steps 16 --warmup 3 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'])"
done; done
cp /tmp/lib_keep.so corb-slam_amd/libcorb_accel.so
