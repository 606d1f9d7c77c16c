#!/usr/bin/env python3
"""profiles/<set>/{kernel_stats.txt, pmc_hbm.json, pmc_sq.txt} of a tools/gpu_profile_ba_store.sh run -> one small JSON (profiles/ba_latest.json) that bench.py
reads for the BA roofline block: per kernel the rocprofv3 average duration, the HBM bytes per launch (FETCH_SIZE + WRITE_SIZE passes) and the SQ counters.
FETCH_SIZE is reported in KB and = TCC_EA0_RDREQ x 64 B; on gfx950 the L2's fabric read requests are 128 B, so the counter reports HALF the bytes read -- for every access width:
tools/ubench/fetch_calib.hip (round 6, profiles/r06_fetch_calib.txt) reads 2 GiB streams with 1 / 4 / 8 / 16 bytes per lane: 0.500 each; 144-byte blocks gathered in 16-byte
pieces: 0.889 of the useful bytes = one half of the two 128-byte lines a block touches.  fetch_bytes_per_launch = 2 x FETCH_SIZE for every kernel (fetch_bytes_counted keeps
the raw figure; rounds 3-5 doubled the two CG kernels only and carried the gather kernels at the counted figure).  WRITE_SIZE: 1.000 of the bytes written, left as counted.
These are L2-miss bytes: Infinity-Cache hits are counted (a kernel whose working set sits in the 256 MB cache can exceed the HBM rate)."""
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
            cal = True                                       # (round 6: the factor 2 holds for every access width, see the header)
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
            "ba_backsub_lean_kernel", "ba_pc_invert_kernel", "ba_pc_invert_all_kernel", "ba_schur_row_kernel", "ba_schur_row_stream_kernel", "ba_schur_combine_kernel", "ml_restrict_kernel", "ml_apply_kernel", "ml_prolong_kernel", "ml_galerkin_kernel", "ba_error_kernel", "ba_linearize_kernel", "ba_sum_points_kernel", "ba_v_kernel", "ba_reduced_rhs_kernel", "ba_backsub_kernel"]
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
