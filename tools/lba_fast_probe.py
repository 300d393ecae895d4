"""tools/lba_fast_probe.py -- phase clocks of the window kernels on LBA-sized windows (DVM_BA_WINDOW_PROF=1): fast form K = 1, K = 32, sequential-order K = 1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dvm_slam_amd import capi, synth
delta = float(np.sqrt(np.float32(5.991)))
wins = []
for a in range(32):
    pr = synth.ba_problem(n_kf=30, n_pts=3000, k_obs=5, seed=0x1BA + a, radius=12.0)
    pr["fixed"][:10] = 1
    wins.append(dict(poses=pr["poses"], fixed=pr["fixed"], points=pr["points"], edges=capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"]),
                     intrinsics=pr["intrinsics"], huber_delta=delta, iterations=10))
capi.ba_optimize_windows(wins[:1], fast=True)
for K, fast in ((1, True), (32, True), (1, False)):
    sys.stderr.write(f"---- K={K} fast={fast}\n"); sys.stderr.flush()
    t0 = time.perf_counter()
    r = capi.ba_optimize_windows(wins[:K], fast=fast)
    dt = time.perf_counter() - t0
    sys.stderr.write(f"     call {dt * 1e3:.2f} ms, structure {r[0]['stats']['ms_structure']:.3f} ms/window, upload+launch+download {r[0]['stats']['ms_optimize']:.2f} ms, trials {sum(r[0]['stats']['trials'])}\n")
