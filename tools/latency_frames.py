"""bench.py's `latency` leg on its own (corb_stereo_frames: B = 1, 2, 8 stereo frames per call, host buffers in and out)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
import bench
print(json.dumps(bench.latency_bench(corb, synth, 0, calls=int(sys.argv[1]) if len(sys.argv) > 1 else 240), indent=1))
