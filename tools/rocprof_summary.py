#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace / pmc) into a small text table for profiles/."""
import sqlite3, sys, json
def main(db, out=None):
    cur = sqlite3.connect(db).cursor()
    lines = []
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines.append("%-64s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for r in rows:
        lines.append("%-64s %8d %14d %12.0f %12d %12d %6.2f%%" % (r[0][:64], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    try:
        pm = cur.execute("select k.name, p.counter_name, count(*), sum(p.value), avg(p.value) from counters_collection p join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name order by k.name").fetchall()
    except Exception as e:
        pm = []
    if pm:
        lines.append("")
        lines.append("%-48s %-28s %8s %18s %16s" % ("kernel", "counter", "n", "sum", "avg/dispatch"))
        for r in pm:
            lines.append("%-48s %-28s %8d %18.0f %16.1f" % (r[0][:48], r[1], r[2], r[3], r[4]))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")
if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
