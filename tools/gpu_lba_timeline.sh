#!/bin/bash
# GPU box: kernel timeline of LocalBundleAdjustment calls (tools/lba_probe.py) -> gpurun_out/<tag>/lba_timeline.txt (the last ~3 ms of the trace: the third call)
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; RAW=/tmp/corb_prof_$TAG
mkdir -p $OUT $RAW
timeout 240 rocprofv3 --kernel-trace -d $RAW -o tl -- python tools/lba_probe.py > $RAW/tl.out 2> $RAW/tl.log
python - "$RAW/tl_results.db" "$OUT/lba_timeline.txt" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
kt = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
rows = cur.execute("select name, start, end from %s order by start" % kt).fetchall()
t1 = rows[-1][2]; win = [r for r in rows if r[1] >= t1 - 2.6e6]
w0 = win[0][1]
lines = ["last %.3f ms of the trace, %d kernels (start us, duration us, gap to previous end us, kernel)" % ((t1 - w0) / 1e6, len(win))]
prev = None
for r in win:
    lines.append("%9.1f %7.1f %7.1f  %s" % ((r[1] - w0) / 1e3, (r[2] - r[1]) / 1e3, 0.0 if prev is None else (r[1] - prev) / 1e3, r[0][:60])); prev = r[2]
busy = sum(r[2] - r[1] for r in win)
lines.append("busy %.1f %% of the window" % (100.0 * busy / (t1 - w0)))
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
PY
tail -3 $RAW/tl.out
