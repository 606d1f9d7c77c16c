#!/usr/bin/env python3
"""Timeline of a rocprofv3 kernel trace (rocpd sqlite): the kernels of a steady-state window in start order with their queue, the busy / overlap / idle
shares of the window, and per kernel name the mean duration split by how many kernels ran beside it."""
import sqlite3, sys
def main(db, out, skip=0.5, span_ms=4.0):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = cur.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
    if not rows: return
    big = sorted(r[1] for r in rows if "fast" in r[0]) or [r[1] for r in rows]
    w0 = big[int(len(big) * skip)]; w1 = w0 + span_ms * 1e6            # a window in the middle of the timed region (median FAST launch)
    win = [r for r in rows if r[1] >= w0 and r[2] <= w1]
    lines = ["window %.3f ms, %d kernels (columns: start us, duration us, queue, kernel)" % ((w1 - w0) / 1e6, len(win))]
    for r in win: lines.append("%10.1f %8.1f  q%-3s %s" % ((r[1] - w0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0][:48]))
    ev = sorted([(r[1], 1) for r in win] + [(r[2], -1) for r in win])
    depth = 0; last = w0; acc = {}
    for t, d in ev:
        acc[depth] = acc.get(depth, 0) + (t - last); last = t; depth += d
    tot = sum(acc.values()) or 1
    lines.append("")
    lines.append("concurrency shares of the window: " + ", ".join("%d kernels %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(acc.items())))
    txt = "\n".join(lines); print(txt); open(out, "w").write(txt + "\n")
if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
