import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from dvm_slam_amd import capi, synth
delta = float(np.sqrt(np.float32(5.991)))
wins = []
for a in range(32):
    pr = synth.ba_problem(n_kf=30, n_pts=3000, k_obs=5, seed=0x1BA + a, radius=12.0)
    pr["fixed"][:10] = 1
    wins.append(dict(poses=pr["poses"], fixed=pr["fixed"], points=pr["points"], edges=capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"]),
                     intrinsics=pr["intrinsics"], huber_delta=delta, iterations=10))
for K in (32, 32, 32, 8, 8, 1, 1):
    t0 = time.perf_counter(); r = capi.ba_optimize_windows(wins[:K], fast=True); dt = time.perf_counter() - t0
    print(K, round(dt * 1e3, 2), round(r[0]["stats"]["ms_optimize"], 2))
