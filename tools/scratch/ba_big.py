import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dvm_slam_amd import capi, synth
from oracle import pyoracle as po
n_kf, n_pts = int(sys.argv[1]), int(sys.argv[2])
pr = synth.ba_problem(n_kf=n_kf, n_pts=n_pts, seed=7)
e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
delta = float(np.sqrt(5.991))
ba = capi.BundleAdjuster()
t0 = time.perf_counter(); ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta); t1 = time.perf_counter()
st = ba.optimize(10); t2 = time.perf_counter()
pg, xg = ba.result(); info = ba.schedule_info(); ba.close()
print("edges", len(e), "set_problem ms", (t1 - t0) * 1e3, "optimize ms", (t2 - t1) * 1e3, "it/s", st["iterations"] / (t2 - t1), "levels", info["levels"], "nz_tiles", info["nz_tiles"], "trials", st["trials"])
if len(sys.argv) > 3:
    eo = po.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    t0 = time.perf_counter(); p_o, x_o, so, _ = po.ba_optimize(pr["poses"], pr["fixed"], pr["points"], eo, pr["intrinsics"], delta, 10); t1 = time.perf_counter()
    print("oracle s", t1 - t0, "trials equal", so["trials"] == st["trials"], "dP", np.abs(pg - p_o).max(), "dX", np.abs(xg - x_o).max())
ba = capi.BundleAdjuster()
ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
ba.profile(1); ba.optimize(10); prof = ba.profile(0); ba.close()
print({k: (v / max(prof["trials"], 1) if k.startswith("ms") else v) for k, v in prof.items()})
