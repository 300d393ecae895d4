import sys, numpy as np
sys.path.insert(0, '/root/repo')
from dvm_slam_amd import capi, synth
from oracle import pyoracle as po
for n, noise in ((500, 0.002), (500, 0.0), (300, 0.003), (120, 0.003)):
    pg = synth.pose_graph(n=n, noise=noise, seed=n)
    for it in (1, 2, 3):
        So, sto = po.pose_graph_optimize(pg["S0"], pg["fixed"], pg["edges_v"], pg["edges_meas"], iterations=it)
        Sg, stg = capi.pose_graph_optimize(pg["S0"], pg["fixed"], pg["edges_v"], pg["edges_meas"], iterations=it)
        print(n, noise, it, "chi", sto[3], stg["chi2_final"], "max|dq|", np.abs(Sg[:, :4]-So[:, :4]).max(), "max|dt|", np.abs(Sg[:, 4:7]-So[:, 4:7]).max(), "max|ds|", np.abs(Sg[:, 7]-So[:, 7]).max(),
              "step", np.abs(So[:, 4:7]-pg["S0"][:, 4:7]).max())
