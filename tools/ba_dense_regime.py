#!/usr/bin/env python3
"""tools/ba_dense_regime.py -- the reduced camera system where it is (nearly) DENSE: 2 000 keyframes / 80 000 landmarks, two laps, 2 % of the
observations long-range (synth.ba_problem) -- 96 % of the 201 x 201 tile triangle is structurally non-zero, 193 elimination levels.  Prints the
schedule, the FLOPs one LM trial executes in the reduced solve (ba_bench.executed_flops_per_trial) and the solve's rate from the solver's own
phase clocks.  Under `rocprofv3 --kernel-trace --stats` the k_chol_update line gives the trailing update's own rate."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from dvm_slam_amd import capi, synth
    import ba_bench
    n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    pr = synth.ba_problem(n_kf=n_kf, n_pts=40 * n_kf, seed=7, laps=2, long_range_frac=0.02)
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    ba = capi.BundleAdjuster()
    args = (pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
    ba.set_problem(*args)
    info = ba.schedule_info()
    fl = ba_bench.executed_flops_per_trial(info)
    ba.optimize(2)
    ba.set_problem(*args)
    ba.profile(1)
    st = ba.optimize(5)
    p = ba.profile(0)
    T = info["tiles_per_side"]
    print(json.dumps({"keyframes": n_kf, "schedule": info, "solver": ba.solve_info(), "tile_fill": info["nz_tiles"] / (T * (T + 1) / 2),
                      "gflop_per_trial": fl / 1e9, "update_products_gflop_per_trial": (info["products"] - 0.25 * info["products_on_diagonal_targets"]) * 2 * 64 ** 3 / 1e9,
                      "trials": p["trials"], "ms_cholesky_solve_per_trial": p["ms_cholesky_solve"] / p["trials"],
                      "solve_tflops": fl * p["trials"] / (p["ms_cholesky_solve"] * 1e-3) / 1e12, "frac_of_fp64_mfma_peak": fl * p["trials"] / (p["ms_cholesky_solve"] * 1e-3) / 1e12 / 78.6,
                      "ms_per_iteration": st["ms_optimize"] / st["iterations"], "chi2_final": st["chi2_final"]}))


if __name__ == "__main__":
    main()
