#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (tools/run_pmc.sh) into profiles/pmc_latest.json: L2-miss bytes per launch of every kernel (what the L2 asks the
fabric for: Infinity-Cache hits are counted, MI355X_MICROARCH.md s HBM).  Units: FETCH_SIZE / WRITE_SIZE are KB.
Calibration, round 6 (tools/ubench/fetch_calib.hip + tools/gpu_fetch_calib.sh, profiles/r06_fetch_calib.txt: 2 GiB streams, 8x the Infinity Cache): FETCH_SIZE reports
0.500 of the bytes read for EVERY access width -- 1, 4, 8 and 16 bytes per lane alike -- and 0.889 of the useful bytes of 144-byte blocks gathered in 16-byte pieces (two
128-byte lines per block = 1.78x, counted at one half); WRITE_SIZE reports 1.000 of the bytes written (1.065 for 1-byte stores).  So reads are DOUBLED here, whatever
the width; writes are taken as counted.  (Rounds 1-5 used a factor of 1.0 for reads, from a calibration on orb_blur_kernel at 128 images whose planes sat in the
Infinity Cache's shadow: every `traffic` figure of those rounds under-reports the read part by 2x -- VERDICT r5 weak 11.)"""
FETCH_FACTOR = 2.0
WRITE_FACTOR = 1.0
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
        res[k] = dict(fetch_kb_per_launch=round(fk, 1), write_kb_per_launch=round(wk, 1), hbm_bytes_per_launch=int((FETCH_FACTOR * fk + WRITE_FACTOR * wk) * 1024), fetch_factor=FETCH_FACTOR, launches=f.get(k, (0, 0))[1])
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))
if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
