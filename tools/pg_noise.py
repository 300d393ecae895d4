import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvm_slam_amd import capi, synth
from oracle import pyoracle as oracle
for noise, seed in ((0.002, 500), (0.0, 500), (0.002, 501), (0.002, 502), (0.001, 503)):
    pg = synth.pose_graph(n=500, noise=noise, seed=seed)
    Sg, stg = capi.pose_graph_optimize(pg["S0"], pg["fixed"], pg["edges_v"], pg["edges_meas"], iterations=20)
    So, sto = oracle.pose_graph_optimize(pg["S0"], pg["fixed"], pg["edges_v"], pg["edges_meas"], iterations=20)
    e0 = np.abs(pg["S0"][:, 4:7] - pg["S_gt"][:, 4:7]).max()
    print(f"noise {noise} seed {seed}: chi2_0 {stg['chi2_initial']:.4g}  GPU final {stg['chi2_final']:.4g} it {stg['iterations']} err {np.abs(Sg[:, 4:7] - pg['S_gt'][:, 4:7]).max()/e0:.3f} | oracle final {sto[3]:.4g} it {int(sto[0])} err {np.abs(So[:, 4:7] - pg['S_gt'][:, 4:7]).max()/e0:.3f}")
