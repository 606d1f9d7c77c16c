#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of rocprofv3 against known byte counts (tools/ubench/fetch_calib.hip) -> gpurun_out/<tag>.txt  (counter KB x 1024 / bytes moved, per kernel)
set -u
TAG=${1:-r06_fetch_calib}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=/tmp/fetch_calib; rm -rf $RAW; mkdir -p $RAW
tools/ubench/fetch_calib > $RAW/bytes.txt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $RAW -o fetch -- tools/ubench/fetch_calib > /dev/null 2> $RAW/f.log
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $RAW -o write -- tools/ubench/fetch_calib > /dev/null 2> $RAW/w.log
python - "$RAW" > gpurun_out/$TAG.txt <<'PY'
import sqlite3, sys, os
raw = sys.argv[1]
known = {}
for line in open(os.path.join(raw, "bytes.txt")):
    if line.startswith("bytes_per_launch"):
        p = line.split(); known[" ".join(p[1:-1])] = int(p[-1])
def avg(db, counter):
    cur = sqlite3.connect(db).cursor()
    return dict((r[0], (r[1], r[2], r[3])) for r in cur.execute("select k.name, avg(p.value), count(*), (select avg(duration) from kernels k2 where k2.name = k.name) from counters_collection p join kernels k on k.dispatch_id = p.dispatch_id where p.counter_name = ? group by k.name", (counter,)).fetchall())
f = avg(os.path.join(raw, "fetch_results.db"), "FETCH_SIZE"); w = avg(os.path.join(raw, "write_results.db"), "WRITE_SIZE")
print("%-58s %14s %14s %8s %14s %8s %10s" % ("kernel", "bytes moved", "FETCH_SIZE B", "ratio", "WRITE_SIZE B", "ratio", "GB/s"))
for name in sorted(set(f) | set(w)):
    key = name.split("(")[0].replace("void ", "").strip()
    b = known.get(key)
    if not b: continue
    fb = f.get(name, (0, 0, 0))[0] * 1024; wb = w.get(name, (0, 0, 0))[0] * 1024; dur = (f.get(name) or w.get(name))[2]
    print("%-58s %14d %14d %8.3f %14d %8.3f %10.1f" % (key, b, fb, fb / b, wb, wb / b, b / dur))
PY
cat gpurun_out/$TAG.txt
