#!/bin/bash
# GPU box, round 5: quadtree kernel variants (threads per (image, level) per level group) -- bit-exactness of each, then the timed region of the headline step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
cp corb-slam_amd/libcorb_accel.so /tmp/lib_keep.so
for f in variants/lib_*.so; do
  cp $f corb-slam_amd/libcorb_accel.so
  echo "== $f" >> gpurun_out/octree_r5_tests.txt
  timeout 600 python -m pytest tests/test_gpu_orb.py -x -q 2>&1 | tail -3 >> gpurun_out/octree_r5_tests.txt
done
cp /tmp/lib_keep.so corb-slam_amd/libcorb_accel.so
bash tools/gpu_variants.sh > gpurun_out/octree_r5_variants.txt 2>&1
# per-kernel times inside the pipeline for the tree's build
python bench.py --no-extras --steps 16 --warmup 3 > gpurun_out/octree_r5_bench.json 2>gpurun_out/octree_r5_bench.err
