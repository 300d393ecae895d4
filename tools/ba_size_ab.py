"""A/B of the two forms of the reduced solve at other sizes: python tools/ba_size_ab.py n_kf n_pts [laps long_range_frac]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvm_slam_amd import capi, synth  # noqa: E402

n_kf, n_pts = int(sys.argv[1]), int(sys.argv[2])
kw = {}
if len(sys.argv) > 4:
    kw = dict(laps=int(sys.argv[3]), long_range_frac=float(sys.argv[4]))
pr = synth.ba_problem(n_kf=n_kf, n_pts=n_pts, seed=7, **kw)
e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
for mode in ("0", "1", None):
    if mode is None:
        os.environ.pop("DVM_BA_FLOW", None)
    else:
        os.environ["DVM_BA_FLOW"] = mode
    ba = capi.BundleAdjuster()
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
    ba.optimize(2)
    dt, its = 0.0, 0
    for _ in range(8):
        ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
        st = ba.optimize(10)
        dt += st["ms_optimize"] * 1e-3; its += st["iterations"]
    print(mode, ba.solve_info(), ba.schedule_info()["levels"], f"{its / dt:.1f} it/s", f"{dt / its * 1e3:.3f} ms", st["chi2_final"])
    ba.close()
