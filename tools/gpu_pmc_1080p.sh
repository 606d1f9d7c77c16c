#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE passes (own runs, kernel trace only) of the 1920x1080 extraction leg at 128 frames per step -> gpurun_out/<tag>/pmc_hbm.json
# (bench.py's orb_1080p.roofline.traffic reads profiles/pmc_1080p.json)
set -u
TAG=${1:-r06_pmc_1080p}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; RAW=/tmp/corb_prof_$TAG
mkdir -p $OUT $RAW
CMD="python tools/bench_1080p_sweep.py 128"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $RAW -o fetch -- $CMD > /dev/null 2> $RAW/fetch.log
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $RAW -o write -- $CMD > /dev/null 2> $RAW/write.log
python tools/pmc_to_json.py $RAW $OUT/pmc_hbm.json > /dev/null || tail -5 $RAW/fetch.log
python - "$OUT/pmc_hbm.json" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
for k, v in r.items():
    if k.startswith("orb_") or k.startswith("stereo_"): print(k, v["hbm_bytes_per_launch"], v.get("launches"))
PY
