"""GPU box: bench.py's orb_1080p leg (configs[4]'s extraction half) at other frames per step."""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
for B in [int(a) for a in sys.argv[1:]] or [32, 64, 128]:
    r = bench.bench_1080p(corb, synth, 0, B=B, steps=8)
    print("1080p frames per step", B, r["value"], r["ms_per_step"], flush=True)
