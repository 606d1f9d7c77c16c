import json,sys
d=json.loads((open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()).strip().split('\n')[-1]); print(d["value"], d["ms_per_step"])
for k,v in d["roofline"]["kernels"].items(): print(k, v["avg_us"], d["roofline"].get("alone_unsplit_avg_us", {}).get(k))
