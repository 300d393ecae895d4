# host-side marks of one optimize() (DVM_BA_DEBUG_SCHEDULE=1 prints them to stderr)
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from dvm_slam_amd import capi, synth
pr = synth.ba_problem()
e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
ba = capi.BundleAdjuster()
for _ in range(3):
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
    st = ba.optimize(10)
print(st["ms_optimize"], st["iterations"])
