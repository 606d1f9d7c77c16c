#!/bin/bash
# GPU box: host-side phase times of the LocalBundleAdjustment calls of the configs[2] replay (CORB_BA_TIMING=1: the chains are off, the phases are what is read), averaged per phase
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_lba
for mode in "--records" ""; do
CORB_BA_TIMING=1 timeout 300 python tools/replay_client.py --frames 200 $mode 2> /tmp/ph.txt > /dev/null
python - "$mode" <<'PY'
import re, sys, collections
acc = collections.OrderedDict(); n = collections.Counter()
for ln in open("/tmp/ph.txt"):
    m = re.match(r"\[(corb_\w+)\]\s+(.*?)\s+([\d.]+) ms", ln)
    if m: k = m.group(1) + " " + m.group(2); acc[k] = acc.get(k, 0.0) + float(m.group(3)); n[k] += 1
print("== replay", sys.argv[1] or "(host pointers)")
for k, v in acc.items(): print("%-60s calls %4d  mean %7.3f ms  total %8.1f ms" % (k, n[k], v / n[k], v))
PY
done | tee gpurun_out/r05_lba/replay_phases.txt
