#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in 0 8 16; do
  touch corb-slam_amd/csrc/orb_kernels.hip
  make -C corb-slam_amd EXTRA="-DOT_KREG=$v" > gpurun_out/make_oct$v.log 2>&1 || { echo "build failed $v"; tail -5 gpurun_out/make_oct$v.log; continue; }
  for b in 8 16 32 64; do
    echo "== KREG=$v batch=$b"
    timeout 300 python bench.py --cpu-frames 0 --ba-cpu-kf 0 --batch $b > gpurun_out/bench_o.json 2> gpurun_out/bench_o.err
    python tools/benchsum.py < gpurun_out/bench_o.json | grep -E "^[0-9]|octree"
  done
done
