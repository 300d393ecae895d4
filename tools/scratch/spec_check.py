# device-side accept decision vs the host's: how many speculative trials are kept (needs a GPU)
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from dvm_slam_amd import capi, synth
for (n_kf, n_pts, seed) in [(500, 20000, 0), (100, 4000, 7), (40, 1500, 3), (12, 300, 1)]:
    pr = synth.ba_problem(n_kf=n_kf, n_pts=n_pts, seed=seed) if n_kf != 500 else synth.ba_problem()
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    ba = capi.BundleAdjuster()
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
    st = ba.optimize(10)
    print(n_kf, "iterations", st["iterations"], "trials", st["trials"], "speculated", st["spec_trials"], "kept", st["spec_kept"], "chi2", st["chi2_final"])
    ba.close()
