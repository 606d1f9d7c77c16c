#!/bin/bash
# development aid (-DCORB_DEV build): time of ba_schur_row_kernel with parts of its round left out (CORB_BA_ROWABL bit mask; results are wrong, timing only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CORB_BA_NO_GRAPH=1
for a in 0 16 17 31; do
  rm -rf /tmp/abl_$a; CORB_BA_ROWABL=$a timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/abl_$a -o s -- python tools/ba_scale.py --pts 100 --obs 3 8 --iters 2 6250 > /dev/null 2>&1
  python tools/rocprof_summary.py /tmp/abl_$a/s_results.db /tmp/abl_$a.txt > /dev/null 2>&1
  echo "abl $a: $(grep ba_schur_row_kernel /tmp/abl_$a.txt | awk '{print $(NF-3)}') ns avg"
done
