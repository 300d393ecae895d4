import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dvm_slam_amd import capi, synth
pr = synth.ba_problem(n_kf=2000, n_pts=80000, seed=7)
e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
ba = capi.BundleAdjuster()
ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
ba.optimize(2)
print("--- second problem, sleep 0.2 s before optimize", file=sys.stderr)
ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
time.sleep(0.2)
ba.optimize(2)
print("--- third: optimize twice", file=sys.stderr)
ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
ba.optimize(1)
ba.optimize(1)
