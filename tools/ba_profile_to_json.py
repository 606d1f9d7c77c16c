#!/usr/bin/env python3
"""profiles/<set>/{kernel_stats.txt, pmc_hbm.json, pmc_sq.txt} of a tools/gpu_profile_ba_store.sh run -> one small JSON (profiles/ba_latest.json) that bench.py
reads for the BA roofline block: per kernel the rocprofv3 average duration, the HBM bytes per launch (FETCH_SIZE + WRITE_SIZE passes) and the SQ counters.
FETCH_SIZE is reported in KB and = TCC_EA0_RDREQ x 64 B; on gfx950 the L2's fabric read requests of a streaming kernel are 128 B (/opt/skills/guides/MI355X_MICROARCH.md,
HBM section: "double it before comparing with a byte count ... calibrate on a known byte count in your own access pattern").  The calibration for the BA kernels' 8-byte-per-lane
streaming reads is profiles/r03_ba50k_b/pmc_tcc.txt: ba_pcg_spmv_kernel TCC_MISS 3.62 M lines x 128 B = 463 MB and TCC_EA0_RDREQ 3.55 M per launch, against 430 MB of S
(1.49 M blocks x 288 B) that cannot be resident (L2 32 MB, Infinity Cache 256 MB) plus the vector gathers -- i.e. FETCH_SIZE (226 MB) under-counts by the guide's factor 2 here as well.
fetch_bytes_per_launch is therefore 2 x FETCH_SIZE for the two CG kernels (fetch_bytes_counted keeps the raw figure); for the gather kernels (Schur, reduced right-hand side, Hpp ...)
the factor is not verified -- 2 x would put two of them above the 6.3 TB/s a stream achieves -- so they carry the counted figure and hbm_bytes_upper = 2 x FETCH_SIZE + WRITE_SIZE.  WRITE_SIZE is left as counted."""
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
            fc = int(v["fetch_kb_per_launch"] * 1024); wr = int(v["write_kb_per_launch"] * 1024)
            cal = k in ("ba_pcg_spmv_kernel", "ba_pcg_step_big_kernel", "ba_pcg_step_restrict_kernel")      # the factor 2 is verified (L2 hit / miss split) for these two streaming kernels only
            res["kernels"][k].update(fetch_bytes_counted=fc, fetch_bytes_per_launch=2 * fc if cal else fc, write_bytes_per_launch=wr,
                                     hbm_bytes_per_launch=(2 * fc if cal else fc) + wr, hbm_bytes_upper=2 * fc + wr)
    sq = os.path.join(d, "pmc_sq.txt")
    if os.path.exists(sq):
        for ln in open(sq):
            f = ln.split()
            if len(f) >= 5 and f[-4].startswith("SQ_"):
                name = re.sub(r"^void ", "", ln).split("(")[0].split("<")[0].strip()
                if name in res["kernels"]:
                    res["kernels"][name].setdefault("sq", {})[f[-4]] = float(f[-1])
    keep = ["ba_pcg_spmv_kernel", "ba_pcg_step_big_kernel", "ba_pcg_step_restrict_kernel", "ba_hpp_scratch_kernel", "ba_pairs_row_kernel", "ba_schur_mfma_kernel", "ba_build_lean_kernel", "ba_v_lean_kernel", "ba_hpp_mfma_kernel", "ba_reduced_rhs_lean_kernel",
            "ba_backsub_lean_kernel", "ba_pc_invert_kernel", "ba_pc_invert_all_kernel", "ba_schur_row_kernel", "ba_schur_combine_kernel", "ml_restrict_kernel", "ml_apply_kernel", "ml_prolong_kernel", "ml_galerkin_kernel", "ba_error_kernel", "ba_linearize_kernel", "ba_sum_points_kernel", "ba_v_kernel", "ba_reduced_rhs_kernel", "ba_backsub_kernel"]
    res["kernels"] = {k: v for k, v in res["kernels"].items() if k in keep}
    try:
        lines = [ln for ln in open(os.path.join(d, "cmd_plain.txt")).read().strip().splitlines() if ln.startswith("poses ")]      # (the tool prints the certificate on further lines)
        res["cmd_plain"] = (lines[-1] if lines else "")[:600]
        m = re.search(r"poses\s+(\d+)\s+points\s+(\d+)\s+edges\s+(\d+)", res["cmd_plain"])
        if m:
            res["poses"], res["points"], res["edges"] = int(m.group(1)), int(m.group(2)), int(m.group(3))
    except Exception:
        pass
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:1500])
if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
