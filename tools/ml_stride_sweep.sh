#!/bin/bash
# development aid (GPU box): CG iterations / solve time of the global BA against the multilevel preconditioner's strides (CORB_BA_ML_STRIDE0 / _STRIDE1, read once per process)
cd "$GRAFT_REPO_ROOT"
for cfg in "8 4" "4 4" "6 4" "4 2" "8 2" "16 4" "4 8"; do
  set -- $cfg
  echo "== stride0 $1 stride1 $2"
  CORB_BA_ML_STRIDE0=$1 CORB_BA_ML_STRIDE1=$2 ML_WEIGHTS="1" python tools/ml_weight_sweep.py ${SIZES:-600 6250} 2>&1 | grep "robust 0"
done
