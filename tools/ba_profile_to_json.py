#!/usr/bin/env python3
"""profiles/<set>/{kernel_stats.txt, pmc_hbm.json, pmc_sq.txt} of a tools/gpu_profile_ba_store.sh run -> one small JSON (profiles/ba_latest.json) that bench.py
reads for the BA roofline block: per kernel the rocprofv3 average duration, the HBM bytes per launch (FETCH_SIZE + WRITE_SIZE passes) and the SQ counters.
FETCH_SIZE is reported in KB and, per /opt/skills/guides/MI355X_MICROARCH.md, under-counts 16-byte-per-lane streaming loads by 2x;
the BA kernels load 8-byte doubles per lane (the block rows of S as 6 consecutive doubles per lane), for which the ORB calibration (factor 1.0 at 4 B per lane,
tools/pmc_to_json.py) is the nearest measured point -- the figures are given as counted, with that caveat."""
import json, os, re, sys
def main(d, out):
    res = {"source": os.path.basename(os.path.normpath(d)), "kernels": {}}
    for ln in open(os.path.join(d, "kernel_stats.txt")):
        m = re.match(r"^(?:void )?([A-Za-z_0-9]+)(?:<[^>]*>)?\(.*?\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)%\s*$", ln)
        if m:
            k = res["kernels"].setdefault(m.group(1), {})
            k["calls"] = k.get("calls", 0) + int(m.group(2)); k["total_ns"] = k.get("total_ns", 0) + int(m.group(3))
    for k, v in res["kernels"].items():
        v["avg_us"] = round(v["total_ns"] / v["calls"] / 1e3, 2)
    hbm = json.load(open(os.path.join(d, "pmc_hbm.json")))
    for k, v in hbm.items():
        if k in res["kernels"]:
            res["kernels"][k].update(fetch_bytes_per_launch=int(v["fetch_kb_per_launch"] * 1024), write_bytes_per_launch=int(v["write_kb_per_launch"] * 1024),
                                     hbm_bytes_per_launch=int(v["hbm_bytes_per_launch"]))
    sq = os.path.join(d, "pmc_sq.txt")
    if os.path.exists(sq):
        for ln in open(sq):
            f = ln.split()
            if len(f) >= 5 and f[-4].startswith("SQ_"):
                name = re.sub(r"^void ", "", ln).split("(")[0].split("<")[0].strip()
                if name in res["kernels"]:
                    res["kernels"][name].setdefault("sq", {})[f[-4]] = float(f[-1])
    keep = ["ba_pcg_spmv_kernel", "ba_pcg_step_big_kernel", "ba_schur_mfma_kernel", "ba_build_lean_kernel", "ba_v_lean_kernel", "ba_hpp_mfma_kernel", "ba_reduced_rhs_lean_kernel",
            "ba_backsub_lean_kernel", "ba_pc_invert_kernel", "ba_error_kernel", "ba_linearize_kernel", "ba_sum_points_kernel", "ba_v_kernel", "ba_reduced_rhs_kernel", "ba_backsub_kernel"]
    res["kernels"] = {k: v for k, v in res["kernels"].items() if k in keep}
    try:
        res["cmd_plain"] = open(os.path.join(d, "cmd_plain.txt")).read().strip().splitlines()[-1][:600]
        m = re.search(r"poses\s+(\d+)\s+points\s+(\d+)\s+edges\s+(\d+)", res["cmd_plain"])
        if m:
            res["poses"], res["points"], res["edges"] = int(m.group(1)), int(m.group(2)), int(m.group(3))
    except Exception:
        pass
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:1500])
if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
