import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d["value"], d["ms_per_step"])
for k,v in d["roofline"]["kernels"].items(): print(k, v["avg_us"], d["roofline"].get("alone_unsplit_avg_us", {}).get(k))
