"""Short BA run for the --pmc passes: two optimisations of the 500-keyframe problem (no CPU leg, no event profiling).  Prints the number of LM
trials it ran as its last line ("trials N"): the fold divides per-kernel totals by it (tools/pmc_traffic.py --ba-trials)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dvm_slam_amd import capi, synth
pr = synth.ba_problem()
e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
ba = capi.BundleAdjuster(0)
trials = 0
for _ in range(2):
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
    trials += ba.optimize(10)["total_trials"]
ba.close()
print("trials", trials)
