"""Weakly constrained bundle adjustments through the tile solver: 30..100 keyframes with three-view
landmarks and a few hundred points (the class of tools/soak_ba.py's case 1876: 87 KF / 273 points / k = 3, gpu-oracle 4e-5), and
7..30 free cameras with two-view landmarks.  On these the result of g2o's recipe itself moves by more than 1e-6 when nothing but the
order of the edge list changes -- the reference adds its edges in heap-address order (Optimizer.cc:1108-1230), so its own result is
one sample of that spread.  The committed contract (include/dvmslam_hip.h, dvm_ba_optimize): identical LM trial sequence, and poses /
landmarks within 1e-6 OR within 10 x the distance between runs of the oracle on permuted edge lists.  Problems with <= 6 free cameras
are outside this file: they run in g2o's own summation order and are bit-identical (tests/test_gpu_ba_window.py)."""
import numpy as np
import pytest

from dvm_slam_amd import capi, synth

pytestmark = pytest.mark.gpu
DELTA = float(np.sqrt(5.991))


def _case(oracle, pr, fixed, delta, iters, rng):
    e = oracle.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    Po, Xo, so, _ = oracle.ba_optimize(pr["poses"], fixed, pr["points"], e, pr["intrinsics"], delta, iters)
    ba = capi.BundleAdjuster()
    ba.set_problem(pr["poses"], fixed, pr["points"], e, pr["intrinsics"], delta)
    sg = ba.optimize(iters)
    Pg, Xg = ba.result()
    form = ba.solve_info()["form"]
    ba.close()
    d = float(max(np.abs(Pg - Po).max(), np.abs(Xg - Xo).max()))
    sens, flips = 0.0, False
    for _ in range(3):
        perm = rng.permutation(len(e))
        P2, X2, s2, _c = oracle.ba_optimize(pr["poses"], fixed, pr["points"], e[perm], pr["intrinsics"], delta, iters)
        sens = max(sens, float(np.abs(P2 - Po).max()), float(np.abs(X2 - Xo).max()))
        flips = flips or s2["trials"] != so["trials"]
    same_lm = sg["trials"] == so["trials"] and sg["iterations"] == so["iterations"]
    return dict(d=d, sens=sens, same_lm=same_lm, oracle_flips=flips, form=form, chi_rel=abs(sg["chi2_final"] - so["chi2_final"]) / max(1.0, abs(so["chi2_final"])))


def _problems(kind):
    rng = np.random.default_rng({"three_view": 1876, "two_view": 42}[kind])
    for i in range(24):
        if kind == "three_view":       # 30..100 keyframes, k = 3, few hundred points
            n_kf = int(rng.integers(30, 101)); n_pts = int(rng.integers(150, 500)); k = 3
        else:                          # 7..30 free cameras, two-view landmarks
            n_kf = int(rng.integers(8, 32)); n_pts = int(rng.integers(60, 400)); k = 2
        delta = DELTA if i % 3 else 0.0
        iters = int(rng.integers(4, 12))
        pr = synth.ba_problem(n_kf, n_pts, k, seed=int(rng.integers(1 << 30)), noise_px=float(rng.choice([0.5, 1.0, 3.0])), outlier_frac=float(rng.choice([0.0, 0.05, 0.2])))
        fixed = pr["fixed"].copy()
        if kind == "three_view" and i % 4 == 0:
            fixed[rng.random(n_kf) < 0.1] = 1
        if fixed.all():
            fixed[-1] = 0
        yield i, pr, fixed, delta, iters, rng


@pytest.mark.parametrize("kind", ["three_view", "two_view"])
def test_weakly_constrained_problems_stay_inside_the_order_sensitivity_of_the_recipe(oracle, kind):
    rows = []
    for i, pr, fixed, delta, iters, rng in _problems(kind):
        nfree = int(((1 - fixed) & np.isin(np.arange(len(fixed)), pr["edge_pose"])).sum())
        if nfree <= 6:
            continue                   # the sequential-order kernel's class
        r = _case(oracle, pr, fixed, delta, iters, rng)
        rows.append(r)
        assert r["same_lm"] or r["oracle_flips"], (kind, i, r)          # the trial sequence may only differ where the oracle's own flips under re-ordering
        assert r["d"] <= max(1e-6, 10 * r["sens"]), (kind, i, r)
        assert r["chi_rel"] <= max(1e-9, 10 * r["sens"]), (kind, i, r)
    assert len(rows) >= 20
    # wherever the oracle itself is insensitive to its edge order, the device is inside 1e-6 (most of the class: the order-sensitive
    # instances are about one in three thousand -- the fixed one below is the soak's)
    assert all(r["d"] <= 1e-6 for r in rows if r["sens"] <= 1e-8)


@pytest.mark.parametrize("iters", [8, 11, 14])
def test_soak_case_1876(oracle, iters):
    """THE instance tools/soak_ba.py found (seed 93, case 1876, replayed from its generator): 87 keyframes, 273 landmarks with three views
    each, 20 % gross outliers, no robust kernel, 11 iterations -- the oracle moves by 1e-5 .. 7e-5 under a permutation of its own edge list
    (2e-5 in the soak's two permutations), the device sat 4e-5 from it.  At 8 iterations the same problem is still order-INsensitive
    (1e-9) and the plain 1e-6 bound holds; the spread grows with the iterations of a not yet converged, barely held gauge."""
    rng = np.random.default_rng(5)
    pr = synth.ba_problem(87, 273, 3, seed=254257914, noise_px=0.5, outlier_frac=0.2)
    assert len(pr["edge_pose"]) == 819
    r = _case(oracle, pr, pr["fixed"], 0.0, iters, rng)
    assert r["same_lm"] or r["oracle_flips"], r
    assert r["d"] <= max(1e-6, 10 * r["sens"]), r
    if iters == 8:
        assert r["sens"] < 1e-7 and r["d"] < 1e-6, r
    if iters == 11:
        assert r["sens"] > 1e-6, r          # the class is real: the recipe's own result is not defined to 1e-6 here
