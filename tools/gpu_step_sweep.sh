#!/bin/bash
# GPU box: the headline step at other part counts / frames per step / handles in flight (timed region only); first sweep in profiles/r04_step_sweep.txt
cd "$GRAFT_REPO_ROOT"
SWEEP=${SWEEP:-256:2:1 384:2:1 512:2:1 512:3:1 512:4:1 768:2:1 1024:2:1 1024:4:1}
for cfg in $SWEEP; do
  IFS=: read b pp inf <<< "$cfg"
  if [ "$pp" = "-" ]; then unset CORB_PARTS; else export CORB_PARTS=$pp; fi
  timeout 200 python bench.py --no-extras --batch $b --inflight $inf --steps 16 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch $b parts $pp inflight $inf:', d['value'], d['ms_per_step'])"
done
