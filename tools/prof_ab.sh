run() { echo "$1 $2: $(env $1 timeout 300 python bench.py --no-extras $2 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; }
for i in 1 2 3; do run X=1 --no-profile; done
for i in 1 2 3; do run CORB_PROF_EVERY=16; done
for i in 1 2; do run CORB_PROF_EVERY=4; done
