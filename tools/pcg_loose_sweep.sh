#!/bin/bash
# GPU box: the cap of the default PCG policy's forcing sequence (CORB_BA_PCG_LOOSE) against the oracle goldens (4 800 and 12 000 keyframes), the default-policy tests and the
# 50 000-keyframe timing
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r06_pcg_loose}.txt; : > $OUT
for t in 1e-6 3e-6 1e-5 3e-5 1e-4; do
  echo "== loose cap $t" >> $OUT
  CORB_BA_PCG_LOOSE=$t python -m pytest tests/test_gpu_ba.py -q -k "golden or default_pcg or config4_size or config3_size" 2>&1 | grep -E "passed|failed|^FAILED" >> $OUT
  CORB_BA_PCG_LOOSE=$t python tools/ba_store_scale.py 6250 2>&1 | grep -E "^poses|^host" | cut -c1-420 >> $OUT
done
cat $OUT
