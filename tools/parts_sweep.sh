run() { echo "$1: $(env $1 timeout 300 python bench.py --no-extras --no-profile $2 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; }
run CORB_OCTREE_FIRST=1
run X=1
run CORB_OCTREE_FIRST=1
run X=1
