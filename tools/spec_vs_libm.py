#!/usr/bin/env python3
"""tools/spec_vs_libm.py [out.json] -- how far can "bit-identical to the oracle" sit from a glibc-linked g2o?  

The oracle (and the device) evaluate SE3Quat::exp's sin / cos / pow(theta, 3) (Thirdparty/g2o/g2o/types/se3quat.h:212-240) and the LM damping
update's pow(2 rho - 1, 3) (g2o/core/optimization_algorithm_levenberg.cpp:131) with a shared double-precision spec (oracle/f64_spec.h) instead of
libm, because no two libms return the same bits.  `make -C oracle libm` builds the SAME oracle with std::sin / std::cos / std::pow in those places.
This script runs both builds on (a) the small-problem suite of tests/test_gpu_ba_window.py (2..6 keyframes), (b) a sample of
tools/ba_sensitivity.py's random tiny problems -- next to the oracle's distance to ITSELF under a permutation of the edge list, the spread the
reference's own heap-address edge order already has --, (c) the 87- and 500-keyframe problems, and prints / stores the distribution."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from dvm_slam_amd import synth   # noqa: E402
from oracle import pyoracle as po   # noqa: E402

DELTA = float(np.sqrt(5.991))


def libm_path():
    so = os.path.join(ROOT, "oracle", "liboracle_libm.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libm"])
    return so


def both(pr, fixed, delta, iters, lm):
    e = po.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    Ps, Xs, ss, _ = po.ba_optimize(pr["poses"], fixed, pr["points"], e, pr["intrinsics"], delta, iters)
    Pl, Xl, sl, _ = po.ba_optimize(pr["poses"], fixed, pr["points"], e, pr["intrinsics"], delta, iters, libpath=lm)
    return dict(trials_equal=ss["trials"] == sl["trials"], dP=float(np.abs(Ps - Pl).max()), dX=float(np.abs(Xs - Xl).max()),
                chi_rel=abs(ss["chi2_final"] - sl["chi2_final"]) / max(abs(ss["chi2_final"]), 1e-300), bit_identical=bool(np.array_equal(Ps, Pl) and np.array_equal(Xs, Xl)))


def summary(rows, key):
    v = np.array([r[key] for r in rows])
    return dict(n=len(v), zero=int((v == 0).sum()), median=float(np.median(v)), p90=float(np.percentile(v, 90)), max=float(v.max()))


def main():
    lm = libm_path()
    out = {}
    # (a) the small-problem suite (tests/test_gpu_ba_window.py): mono initialisation (2 KF) and 3..6-keyframe windows
    rows = []
    for seed in range(6):
        pr = synth.small_window_problem(2, 100 + 40 * seed, seed=100 + seed, noise_px=[0.5, 1.0, 2.0][seed % 3], outlier_frac=[0.0, 0.05][seed % 2])
        rows.append(dict(both(pr, pr["fixed"], DELTA, 20, lm), tag=f"mono-init seed {seed}"))
    for n_kf in (3, 4, 5, 6):
        for seed in range(6):
            pr = synth.small_window_problem(n_kf, 60 + 30 * seed, seed=200 + 10 * n_kf + seed)
            rows.append(dict(both(pr, pr["fixed"], DELTA if seed % 2 else 0.0, 10, lm), tag=f"window {n_kf} KF seed {seed}"))
    out["small_suite"] = dict(dP=summary(rows, "dP"), dX=summary(rows, "dX"), trials_equal=sum(r["trials_equal"] for r in rows), bit_identical=sum(r["bit_identical"] for r in rows), n=len(rows),
                              worst=sorted(rows, key=lambda r: -r["dP"])[:3])
    # (b) random tiny problems, next to the oracle's own sensitivity to the edge order
    import ba_sensitivity
    rows, perm = [], []
    for c in ba_sensitivity.problems(int(os.environ.get("SPEC_LIBM_CASES", "400")), 7):
        pr = c["pr"]
        rows.append(dict(both(pr, c["fixed"], c["delta"], c["iters"], lm), tag=c["tag"]))
        e = po.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
        P0, X0, s0, _ = po.ba_optimize(pr["poses"], c["fixed"], pr["points"], e, pr["intrinsics"], c["delta"], c["iters"])
        P1, X1, s1, _ = po.ba_optimize(pr["poses"], c["fixed"], pr["points"], e[c["perm"]], pr["intrinsics"], c["delta"], c["iters"])
        perm.append(dict(dP=float(np.abs(P0 - P1).max()), dX=float(np.abs(X0 - X1).max()), trials_equal=s0["trials"] == s1["trials"]))
    out["random_tiny"] = dict(spec_vs_libm=dict(dP=summary(rows, "dP"), dX=summary(rows, "dX"), trials_equal=sum(r["trials_equal"] for r in rows), bit_identical=sum(r["bit_identical"] for r in rows)),
                              oracle_vs_permuted_oracle=dict(dP=summary(perm, "dP"), dX=summary(perm, "dX"), trials_equal=sum(r["trials_equal"] for r in perm)), n=len(rows),
                              libm_beyond_10x_permutation=int(sum(1 for r, q in zip(rows, perm) if r["dP"] > 10 * max(q["dP"], 1e-15))),
                              worst=sorted(rows, key=lambda r: -r["dP"])[:3])
    # (c) the sizes the 1e-6 bound is stated for
    pr = synth.ba_problem(n_kf=87, n_pts=2500, seed=11)
    out["kf87"] = both(pr, pr["fixed"], DELTA, 10, lm)
    pr = synth.ba_problem()
    out["kf500"] = both(pr, pr["fixed"], DELTA, 4, lm)
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
