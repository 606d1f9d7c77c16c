#!/bin/bash
# GPU box: LocalBundleAdjustment A/B (host-side classifications vs device ones, chain lengths) on tools/lba_store_probe.py + the tests that cover the staged solve
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_lba
for v in "CORB_BA_HOST_STAGES=1 CORB_LBA_HOST_FLATTEN=1" "CORB_LBA_HOST_FLATTEN=1" "CORB_X=1"; do
  echo "== $v" ; env LBA_PROBE_CALLS=60 $v timeout 300 python tools/lba_store_probe.py 2>&1 | tail -3
done > gpurun_out/r05_lba/ab.txt 2>&1
cat gpurun_out/r05_lba/ab.txt
timeout 1500 python -m pytest tests/test_gpu_staged.py tests/test_gpu_local_ba_store.py tests/test_gpu_ba.py tests/test_gpu_replay.py tests/test_gpu_replay_records.py tests/test_gpu_host.py -x -q -m gpu 2>&1 | tail -15
for mode in "" "--records"; do
  for v in "CORB_BA_HOST_STAGES=1 CORB_LBA_HOST_FLATTEN=1" "CORB_LBA_HOST_FLATTEN=1" "CORB_X=1"; do
    echo "== replay $mode $v"; env $v timeout 300 python tools/replay_client.py --frames 400 $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fps %.1f' % d['client_fps'], {k: round(v,1) for k,v in d['stage_ms'].items() if 'Bundle' in k})"
  done
done 2>&1 | tee gpurun_out/r05_lba/replay_ab.txt
