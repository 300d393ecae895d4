import sys, numpy as np
sys.path.insert(0, '/root/repo')
from dvm_slam_amd import capi, synth
pr = synth.ba_problem()
e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
ref = None; bad = 0
ba = capi.BundleAdjuster()
for it in range(60):
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
    st = ba.optimize(10)
    p, x = ba.result()
    key = (tuple(st["trials"]), st["chi2_final"])
    if ref is None: ref = (key, p, x)
    elif key != ref[0] or not np.array_equal(p, ref[1]) or not np.array_equal(x, ref[2]):
        bad += 1; print("run", it, "differs", key, ref[0], np.abs(p - ref[1]).max())
print("repeat: 60 runs,", bad, "differ")
