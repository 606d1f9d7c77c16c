run() { echo "$1 $2: $(env $1 timeout 300 python bench.py --no-extras --no-profile $2 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; }
run X=1
run CORB_SIDE_PRIO=1
run CORB_SIDE_PRIO=-1
run X=1
run CORB_SIDE_PRIO=1
run CORB_SIDE_PRIO=-1
