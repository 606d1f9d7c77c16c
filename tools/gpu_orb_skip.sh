cd "$GRAFT_REPO_ROOT"
for k in none pyramid fast octree blur describe "octree,describe" "pyramid,blur"; do
  v=$(CORB_ORB_SKIP=$k timeout 200 python bench.py --no-extras --steps 20 --warmup 12 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$k: $v"
done
