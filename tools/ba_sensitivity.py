#!/usr/bin/env python3
"""tools/ba_sensitivity.py [cases] [seed] [--gpu] -- how far does the bundle adjustment of a TINY problem (2..6 keyframes: the
2-keyframe GlobalBundleAdjustemnt after monocular initialisation, Tracking.cc:2330; 3..5-keyframe local windows) move when nothing
but the ORDER of its edges changes?  The reference adds edges in the iteration order of std::map<KeyFrame*, ...> / hash sets, i.e.
in heap-address order: its own result on such a problem is one sample of this spread.  The CPU oracle is run on the problem and on
the same problem with its edge list permuted; with --gpu the HIP solver is run too and its distance to the oracle is printed next
to the oracle's distance to its permuted self."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvm_slam_amd import synth   # noqa: E402
from oracle import pyoracle as po   # noqa: E402


def problems(n, seed):
    rng = np.random.default_rng(seed)
    done = 0
    while done < n:
        n_kf = int(rng.integers(2, 7)); n_pts = int(rng.integers(30, 600)); k = int(rng.integers(2, min(8, n_kf) + 1))
        delta = float(np.sqrt(5.991)) if rng.random() < 0.6 else 0.0
        iters = int(rng.choice([5, 10, 20]))
        s = int(rng.integers(1 << 30))
        try:
            pr = synth.ba_problem(n_kf, n_pts, k, seed=s, noise_px=float(rng.choice([0.0, 0.5, 1.0, 3.0])), outlier_frac=float(rng.choice([0.0, 0.05, 0.2])))
        except RuntimeError:
            continue
        fixed = pr["fixed"].copy()
        fixed[rng.random(n_kf) < rng.choice([0.0, 0.1, 0.5])] = 1
        if fixed.all():
            fixed[-1] = 0
        done += 1
        yield dict(pr=pr, fixed=fixed, delta=delta, iters=iters, perm=rng.permutation(len(pr["edge_pose"])),
                   tag=f"kf={n_kf} fixed={int(fixed.sum())} pts={n_pts} k={k} delta={delta:.2f} it={iters} seed={s}")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    gpu = "--gpu" in sys.argv
    if gpu:
        from dvm_slam_amd import capi
    rows = []
    for c in problems(n, seed):
        pr, fixed, delta, iters, perm = c["pr"], c["fixed"], c["delta"], c["iters"], c["perm"]
        e = po.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
        Po, Xo, so, _ = po.ba_optimize(pr["poses"], fixed, pr["points"], e, pr["intrinsics"], delta, iters)
        e2 = po.make_edges(pr["edge_pose"][perm], pr["edge_point"][perm], pr["obs"][perm], pr["inv_sigma2"][perm])
        P2, X2, s2, _ = po.ba_optimize(pr["poses"], fixed, pr["points"], e2, pr["intrinsics"], delta, iters)
        d_perm = max(np.abs(P2 - Po).max(), np.abs(X2 - Xo).max())
        d_gpu, bits = float("nan"), False
        if gpu:
            ba = capi.BundleAdjuster()
            ba.set_problem(pr["poses"], fixed, pr["points"], e, pr["intrinsics"], delta)
            sg = ba.optimize(iters)
            Pg, Xg = ba.result()
            ba.close()
            d_gpu = max(np.abs(Pg - Po).max(), np.abs(Xg - Xo).max())
            bits = bool(np.array_equal(Pg, Po) and np.array_equal(Xg, Xo) and sg["trials"] == so["trials"])
        rows.append((d_perm, d_gpu, bits, c["tag"], so["total_trials"] == s2["total_trials"]))
    rows.sort(key=lambda r: -max(r[0], 0 if np.isnan(r[1]) else r[1]))
    for r in rows[:15]:
        print("oracle vs edge-permuted oracle %.3g | gpu vs oracle %.3g bit-identical=%s | %s same_trials=%s" % r)
    dp = np.array([r[0] for r in rows]); dg = np.array([r[1] for r in rows])
    print(f"{len(rows)} problems: oracle vs permuted oracle > 1e-6 in {(dp > 1e-6).mean():.1%}, > 1e-4 in {(dp > 1e-4).mean():.1%}, max {dp.max():.3g}")
    if gpu:
        print(f"gpu vs oracle: bit-identical in {np.mean([r[2] for r in rows]):.1%}, > 1e-6 in {(dg > 1e-6).mean():.1%}, > 1e-4 in {(dg > 1e-4).mean():.1%}, max {np.nanmax(dg):.3g}")


if __name__ == "__main__":
    main()
