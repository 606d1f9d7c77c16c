run() { echo "$1: $(env $1 timeout 300 python bench.py --no-extras --no-profile $2 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; }
run CORB_STAGE=0
run CORB_STAGE=1
run CORB_STAGE=2
run CORB_STAGE=3
run CORB_STAGE=4
run CORB_STAGE=1
