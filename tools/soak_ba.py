#!/usr/bin/env python3
"""tools/soak_ba.py [seconds] [seed] -- randomized GPU-vs-oracle soak of the bundle adjustment (dvm_ba_*), the pose-only
optimisation and the list matcher: random problem sizes, fixed-camera sets, observation counts, noise / outlier levels,
Huber on / off, iteration counts.  Problems with <= 6 free cameras (and batches through dvm_ba_optimize_windows): BIT-IDENTICAL to the
oracle.  Larger ones: identical LM trial sequences, poses / points within 1e-6 -- or, for a problem whose result moves further than
that when only the ORDER of the oracle's own edge list changes, within 10 x that movement (printed as "order-sensitive").  Not part of pytest."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvm_slam_amd import capi, synth  # noqa: E402
from oracle import pyoracle as po     # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tri_scene                      # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    cases = bad = tri_cases = window_cases = order_sensitive = 0
    while time.time() - t0 < budget:
        n_kf = int(rng.integers(2, 90)); n_pts = int(rng.integers(30, 1500)); k = int(rng.integers(2, min(8, n_kf) + 1))
        if rng.random() < float(os.environ.get("SOAK_BIG_FRAC", "0.12")):       # many tile columns: the nested-dissection schedule, short tiles of every fill, the level kernels
            n_kf = int(rng.integers(90, 420)); n_pts = int(rng.integers(20, 30) * n_kf); k = int(rng.integers(4, 9))
        delta = float(np.sqrt(5.991)) if rng.random() < 0.6 else 0.0
        iters = int(rng.integers(1, 12))
        extra_kw = {}
        if n_kf >= 90 and rng.random() < 0.5:      # maps after loop closures / with landmarks seen from far apart: deep elimination trees, kept landmarks, the flow form
            extra_kw = dict(laps=int(rng.integers(1, 3)), long_range_frac=float(rng.choice([0.001, 0.003, 0.01])), long_range_obs=int(rng.integers(2, 5)))
        pr = synth.ba_problem(n_kf, n_pts, k, seed=int(rng.integers(1 << 30)), noise_px=float(rng.choice([0.0, 0.5, 1.0, 3.0])),
                              outlier_frac=float(rng.choice([0.0, 0.05, 0.2])), **extra_kw)
        fixed = pr["fixed"].copy()
        extra = rng.random(n_kf) < rng.choice([0.0, 0.1, 0.5])
        fixed[extra] = 1
        if fixed.all():
            fixed[-1] = 0
        tag = f"kf={n_kf} pts={n_pts} k={k} delta={delta:.2f} it={iters} fixed={int(fixed.sum())} E={len(pr['edge_pose'])} {extra_kw if extra_kw else ''}"
        try:
            e = po.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
            Po, Xo, so, chio = po.ba_optimize(pr["poses"], fixed, pr["points"], e, pr["intrinsics"], delta, iters)
            ba = capi.BundleAdjuster()
            ba.set_problem(pr["poses"], fixed, pr["points"], e, pr["intrinsics"], delta)
            sg = ba.optimize(iters)
            Pg, Xg = ba.result()
            chig, _ = ba.edge_chi2()
            ba.close()
            nfree = int(((1 - fixed) & np.isin(np.arange(n_kf), pr["edge_pose"])).sum())
            same_lm = sg["iterations"] == so["iterations"] and sg["total_trials"] == so["total_trials"]
            if nfree <= 6:
                # window mode (csrc/ba_window.hip): g2o's summation order -> the oracle's bits, whatever the conditioning
                window_cases += 1
                ok = (same_lm and np.array_equal(Pg.view(np.int64), Po.view(np.int64)) and np.array_equal(Xg.view(np.int64), Xo.view(np.int64)) and
                      np.array_equal(chig.view(np.int64), chio.view(np.int64)) and sg["chi2_final"] == so["chi2_final"] and sg["lambda_final"] == so["lambda_final"])
            else:
                # tile solver: its own (parallel) summation order -> 1e-6, or, where the problem itself is order-sensitive beyond that, within
                # 10 x the distance the ORACLE moves when nothing but the order of its edge list changes (two permutations) -- the
                # reference adds its edges in heap-address order, so its own result is one sample of that spread
                d = max(np.abs(Pg - Po).max(), np.abs(Xg - Xo).max())
                ok = same_lm and d <= 1e-6 and abs(sg["chi2_final"] - so["chi2_final"]) <= 1e-5 * max(1.0, abs(so["chi2_final"]))
                if not ok:
                    sens = 0.0
                    for _ in range(2):
                        perm = rng.permutation(len(e))
                        P2, X2, s2, _c = po.ba_optimize(pr["poses"], fixed, pr["points"], e[perm], pr["intrinsics"], delta, iters)
                        sens = max(sens, np.abs(P2 - Po).max(), np.abs(X2 - Xo).max())
                        same_lm = same_lm or s2["total_trials"] != so["total_trials"]     # the oracle's own trial sequence flips under re-ordering
                    if same_lm and d <= 10 * sens:
                        ok = True
                        order_sensitive += 1
                        print(f"order-sensitive case {cases}: {tag}: gpu-oracle {d:.3g}, oracle-permuted oracle {sens:.3g}", flush=True)
            if not ok:
                bad += 1
                print(f"BA MISMATCH case {cases}: {tag}: it {sg['iterations']} vs {so['iterations']} trials {sg['total_trials']} vs "
                      f"{so['total_trials']} chi2 {sg['chi2_final']:.9g} vs {so['chi2_final']:.9g} dP {np.abs(Pg - Po).max():.3g} "
                      f"dX {np.abs(Xg - Xo).max():.3g}", flush=True)
        except Exception as ex:
            bad += 1
            print(f"BA EXCEPTION case {cases}: {tag}: {ex!r}", flush=True)
        # every eighth case: a batch of windows through dvm_ba_optimize_windows, each against its own oracle run, bit for bit
        if cases % 8 == 0:
            try:
                wins = []
                for _ in range(int(rng.integers(1, 7))):
                    nk = int(rng.integers(2, 25))
                    q = (synth.small_window_problem(nk, int(rng.integers(20, 400)), seed=int(rng.integers(1 << 30)), noise_px=float(rng.choice([0.0, 1.0, 3.0])))
                         if nk <= 7 else synth.ba_problem(nk, int(rng.integers(60, 800)), int(rng.integers(2, 7)), seed=int(rng.integers(1 << 30))))
                    wins.append(dict(poses=q["poses"], fixed=q["fixed"], points=q["points"], edges=po.make_edges(q["edge_pose"], q["edge_point"], q["obs"], q["inv_sigma2"]),
                                     intrinsics=q["intrinsics"], huber_delta=float(rng.choice([0.0, np.sqrt(5.991)])), iterations=int(rng.integers(1, 21))))
                for g, w in zip(capi.ba_optimize_windows(wins), wins):
                    Pw, Xw, sw_, cw = po.ba_optimize(w["poses"], w["fixed"], w["points"], w["edges"], w["intrinsics"], w["huber_delta"], w["iterations"])
                    window_cases += 1
                    if not (np.array_equal(g["poses"].view(np.int64), Pw.view(np.int64)) and np.array_equal(g["points"].view(np.int64), Xw.view(np.int64)) and
                            np.array_equal(g["edge_chi2"].view(np.int64), cw.view(np.int64)) and list(g["stats"]["trials"]) == list(sw_["trials"])):
                        bad += 1
                        print(f"WINDOW MISMATCH case {cases}: P={len(w['poses'])} E={len(w['edges'])} it={w['iterations']} dP {np.abs(g['poses'] - Pw).max():.3g}", flush=True)
            except Exception as ex:
                bad += 1
                print(f"WINDOW EXCEPTION case {cases}: {ex!r}", flush=True)
        cases += 1
        # LocalMapping::CreateNewMapPoints' geometry: kernel == oracle bit for bit on a random two-view scene
        try:
            S = tri_scene.scene(seed=int(rng.integers(1 << 30)), n=int(rng.integers(1, 4000)), baseline=float(rng.uniform(0.02, 2.0)),
                                noise_px=float(rng.choice([0.0, 0.5, 2.0])), wrong_frac=float(rng.choice([0.0, 0.2])))
            a = (S["K1"], S["K2"], S["T1w"], S["T2w"], S["Ow1"], S["Ow2"], S["kps1"], S["kps2"], S["pairs"], S["sigma2_1"], S["sigma2_2"],
                 S["sf1"], S["sf2"], S["ratio_factor"])
            kw = dict(far_points=bool(rng.random() < 0.5), th_far=float(rng.uniform(2, 40)), cos_parallax_max=float(rng.choice([0.9998, 0.9996])))
            Xo, so_ = po.triangulate_matches(*a, **kw)
            Xg, sg_ = capi.triangulate_matches(*a, **kw)
            tri_cases += 1
            if not (np.array_equal(so_, sg_) and np.array_equal(Xo.view(np.uint32), Xg.view(np.uint32))):
                bad += 1
                print(f"TRIANGULATION MISMATCH case {cases}: n={len(S['pairs'])} status diff {(so_ != sg_).sum()}", flush=True)
        except Exception as ex:
            bad += 1
            print(f"TRIANGULATION EXCEPTION case {cases}: {ex!r}", flush=True)
    print(f"soak_ba: {tri_cases} triangulation scenes;", end=" ")
    print(f"soak_ba: {cases} cases ({window_cases} solved by the sequential-order kernel: bit-identical; {order_sensitive} tile-solver cases beyond 1e-6 "
          f"but within 10 x the oracle's own order sensitivity), {bad} mismatches, {time.time() - t0:.0f} s, seed {seed}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
