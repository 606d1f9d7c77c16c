#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (tools/run_pmc.sh) into profiles/pmc_latest.json:
HBM bytes per launch of every kernel.  Units: FETCH_SIZE / WRITE_SIZE are KB.  Correction: the guide's
x2 rule applies to 16-B-per-lane streaming loads only; these kernels load 4 B (or 1 B) per lane, and the
counter was calibrated on orb_blur_kernel (a pure stream with known traffic: 128 images x 1.444 MB read
x 38/32 halo rows = 219 MB expected, 200 MB counted; written 185 MB expected) => factor 1.0."""
import json, sqlite3, sys, os
def avg(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select k.name, avg(p.value), count(*) from counters_collection p join kernels k on k.dispatch_id = p.dispatch_id where p.counter_name = ? group by k.name", (counter,)).fetchall()
    return {r[0].split("(")[0].replace("void ", "").split("<")[0]: (r[1], r[2]) for r in rows}
def main(d, out):
    f = avg(os.path.join(d, "fetch_results.db"), "FETCH_SIZE"); w = avg(os.path.join(d, "write_results.db"), "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fk = f.get(k, (0, 0))[0]; wk = w.get(k, (0, 0))[0]
        res[k] = dict(fetch_kb_per_launch=round(fk, 1), write_kb_per_launch=round(wk, 1), hbm_bytes_per_launch=int((fk + wk) * 1024), launches=f.get(k, (0, 0))[1])
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))
if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
