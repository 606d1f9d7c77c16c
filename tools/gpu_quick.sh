#!/bin/bash
# quick GPU loop: parity tests of the ORB front end + the timed region of bench.py (per-kernel averages)
# usage (on the GPU box, via gpurun): tools/gpu_quick.sh [pytest selection...]
sel=${@:-tests/test_gpu_orb.py}
timeout 900 python -m pytest $sel -x -q 2>&1 | tail -4
timeout 400 python bench.py --cpu-frames 0 --ba-cpu-kf 0 --ba-kf 0 --replay-frames 0 > gpurun_out/quick.json 2> gpurun_out/quick.err || tail -5 gpurun_out/quick.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/quick.json').read())
r = d['roofline']
print('fps', d['value'], 'ms/step', d['ms_per_step'])
print({k: v['avg_us'] for k, v in r['kernels'].items()})
print('alone', r.get('alone_unsplit_avg_us'))
PY
