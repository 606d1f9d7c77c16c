#!/bin/bash
# GPU box: deviating BA calls of tools/conc_probe.py (a dense global BA beside tracking calls on a second thread) and tools/conc_probe3.py with each library under variants/
# and with the tree's own
set -u
cd "$GRAFT_REPO_ROOT"
N=${1:-100}
cp corb-slam_amd/libcorb_accel.so /tmp/lib_tree.so
for f in /tmp/lib_tree.so variants/lib_*.so; do
  cp $f corb-slam_amd/libcorb_accel.so
  echo "$(basename $f): $(python tools/conc_probe.py $N 2>&1 | tail -1) | $(python tools/conc_probe3.py $((N / 4)) 2>&1 | tail -2 | tr '\n' ' ')"
done
cp /tmp/lib_tree.so corb-slam_amd/libcorb_accel.so
