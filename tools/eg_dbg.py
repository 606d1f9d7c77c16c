import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
import pyorc
for seed, K, fs in [(7100, 60, False), (7101, 100, False), (7102, 80, True), (7103, 150, False), (7001, 90, False)]:
    g = synth.essential_graph(seed, K=K)
    G = corb.Optimizer.OptimizeEssentialGraph(g, 20, fs); R = pyorc.optimize_essential_graph(g, 20, fs)
    print(seed, K, "chi2 gpu", np.round(G["chi2"], 5), "cpu", np.round(R["chi2"], 5))
    print("   dS", np.abs(G["S"] - R["S"]).max(), "dT", np.abs(G["Tiw"] - R["Tiw"]).max(), "dP", np.abs(G["points"] - R["points"]).max(), "|S|", np.abs(R["S"]).max())
