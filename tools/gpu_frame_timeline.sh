#!/bin/bash
# GPU box: kernel + copy timeline of the last frames of the configs[2] replay on records (tools/replay_client.py --records) -> gpurun_out/<tag>/frame_timeline.txt
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; RAW=/tmp/corb_prof_$TAG
mkdir -p $OUT $RAW
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $RAW -o tl -- python tools/replay_client.py --frames 42 --records > $RAW/tl.out 2> $RAW/tl.log
python - "$RAW/tl_results.db" "$OUT/frame_timeline.txt" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
kt = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
rows = [(r[0], r[1], r[2]) for r in cur.execute("select name, start, end from %s" % kt).fetchall()]
mt = [t for t in tabs if "memory_cop" in t.lower() or t.lower() == "memory_copies"]
for t in mt[:1]:
    cols = [c[1] for c in cur.execute("pragma table_info(%s)" % t).fetchall()]
    nm = "name" if "name" in cols else cols[0]; sz = "size" if "size" in cols else None
    try:
        for r in cur.execute("select %s, start, end%s from %s" % (nm, (", " + sz) if sz else "", t)).fetchall():
            rows.append(("copy %s%s" % (r[0], (" %d B" % r[3]) if sz else ""), r[1], r[2]))
    except Exception as e:
        pass
rows.sort(key=lambda r: r[1])
t1 = rows[-1][2]; import os
win = [r for r in rows if r[1] >= t1 - float(os.environ.get("FRAME_WINDOW_US", "4500")) * 1e3]
w0 = win[0][1]
lines = ["last %.3f ms of the trace, %d entries (start us, duration us, gap to previous end us, kernel / copy)" % ((t1 - w0) / 1e6, len(win))]
prev = None
for r in win:
    lines.append("%9.1f %7.1f %7.1f  %s" % ((r[1] - w0) / 1e3, (r[2] - r[1]) / 1e3, 0.0 if prev is None else (r[1] - prev) / 1e3, r[0][:80])); prev = r[2]
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
PY
tail -2 $RAW/tl.log
