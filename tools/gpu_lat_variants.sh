#!/bin/bash
# GPU box: the latency leg (corb_stereo_frames, B = 1 / 2 / 8) with every library build under variants/, after the ORB byte-parity tests with each
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
cp corb-slam_amd/libcorb_accel.so /tmp/lib_keep.so
for f in variants/lib_*.so; do
  cp $f corb-slam_amd/libcorb_accel.so
  echo "== $f"
  timeout 600 python -m pytest tests/test_gpu_orb.py -x -q 2>&1 | tail -2
  timeout 300 python tools/latency_frames.py 120 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for b in ('B1','B2','B8'): print(b, d[b]['host_to_host_ms'], d[b]['stages_ms']['kernels'], d[b]['resident_ms'])
print(d['kernels_alone_us_B1'])"
done
cp /tmp/lib_keep.so corb-slam_amd/libcorb_accel.so
