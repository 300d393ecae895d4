import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dvm_slam_amd import capi, synth
n_kf, n_pts = int(sys.argv[1]), int(sys.argv[2])
pr = synth.ba_problem(n_kf=n_kf, n_pts=n_pts, seed=7)
e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
delta = float(np.sqrt(5.991))
ba = capi.BundleAdjuster()
for it in (1, 2, 5, 10, 10):
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    t1 = time.perf_counter(); st = ba.optimize(it); t2 = time.perf_counter()
    print(it, "iterations", st["iterations"], "trials", st["total_trials"], "ms", (t2 - t1) * 1e3, "ms_optimize", st.get("ms_optimize"))
ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
for k in range(4):
    t1 = time.perf_counter(); st = ba.optimize(1); t2 = time.perf_counter()
    print("repeat optimize(1) without set_problem:", (t2 - t1) * 1e3, "ms")
