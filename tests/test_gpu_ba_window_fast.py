"""dvm_ba_optimize_windows_fast (csrc/ba_window.hip, k_ba_window<true>): K LocalBundleAdjustment windows in ONE launch, a workgroup per
window, the LM control on the device, every sum a tree in a fixed order (reference: Optimizer.cc:1030-1387 per window; g2o
block_solver.hpp:381-483, optimization_algorithm_levenberg.cpp:59-165).  Contract: the general solver's -- the oracle's LM trial
sequence, poses / landmarks within 1e-6, chi2 within 1e-9 relative -- and determinism (the same bits run to run, and for a window
whatever its companions in the batch)."""
import numpy as np
import pytest

from dvm_slam_amd import capi, synth

pytestmark = pytest.mark.gpu
DELTA = float(np.sqrt(5.991))
TOL = 1e-6          # north_star: BA poses / landmarks within 1e-6 of the reference


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.int64)


def _window(pr, iters, n_fixed=None):
    fixed = pr["fixed"].copy()
    if n_fixed is not None:
        fixed[:] = 0; fixed[:n_fixed] = 1
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    return dict(poses=pr["poses"], fixed=fixed, points=pr["points"], edges=e, intrinsics=pr["intrinsics"], huber_delta=DELTA, iterations=iters)


def _check_vs_oracle(oracle, w, g, tag):
    P, X, st, chi = oracle.ba_optimize(w["poses"], w["fixed"], w["points"], w["edges"], w["intrinsics"], w["huber_delta"], w["iterations"])
    chi_at, depth = oracle.ba_edge_chi2(P, X, w["edges"], w["intrinsics"])
    if w["iterations"] == 0:
        chi = chi_at         # (the oracle's optimize() evaluates nothing without an iteration)
    s = g["stats"]
    assert s["iterations"] == st["iterations"] and list(s["trials"]) == list(st["trials"]) and s["stop_reason"] == st["stop_reason"], (tag, s, st)
    if w["iterations"] > 0:      # (a call without iterations reports the chi2 of the input state; the oracle's statistics stay empty)
        assert abs(s["chi2_final"] - st["chi2_final"]) <= 1e-9 * abs(st["chi2_final"]), tag
    assert np.abs(g["poses"] - P).max() < TOL and np.abs(g["points"] - X).max() < TOL, (tag, np.abs(g["poses"] - P).max(), np.abs(g["points"] - X).max())
    assert np.allclose(g["edge_chi2"], chi, rtol=1e-6, atol=1e-9), tag
    assert (g["depth_positive"] != depth).sum() == 0, tag


@pytest.fixture(scope="module")
def windows():
    wins = []
    for k in range(4):     # LocalBundleAdjustment-sized: 30 keyframes of which 10 fixed
        pr = synth.ba_problem(n_kf=30, n_pts=800 + 700 * k, k_obs=5, seed=0x3BA + k, radius=12.0)
        wins.append(_window(pr, 10, n_fixed=10))
    pr = synth.ba_problem(n_kf=32, n_pts=900, k_obs=6, seed=0x3C0, radius=12.0)
    wins.append(_window(pr, 10, n_fixed=2))            # 30 free cameras: the largest reduced system the kernel holds (130 KB of LDS)
    pr = synth.ba_problem(n_kf=12, n_pts=400, k_obs=4, seed=0x3C1, radius=10.0, outlier_frac=0.05)
    wins.append(_window(pr, 5, n_fixed=3))
    pr = synth.ba_problem(n_kf=20, n_pts=1500, k_obs=5, seed=0x3C2, radius=12.0)
    wins.append(_window(pr, 0, n_fixed=5))             # no iteration: the edges are evaluated at the input state
    return wins


def test_fast_windows_agree_with_oracle(oracle, windows):
    res = capi.ba_optimize_windows(windows, fast=True)
    for k, (w, g) in enumerate(zip(windows, res)):
        _check_vs_oracle(oracle, w, g, f"window {k}")
    assert res[4]["stats"]["iterations"] > 0


def test_fast_windows_are_deterministic_and_independent_of_the_batch(windows):
    a = capi.ba_optimize_windows(windows, fast=True)
    b = capi.ba_optimize_windows(windows, fast=True)
    solo = [capi.ba_optimize_windows([w], fast=True)[0] for w in windows[:3]]
    rev = capi.ba_optimize_windows(windows[::-1], fast=True)[::-1]
    for k in range(len(windows)):
        for other in (b[k], rev[k]) + ((solo[k],) if k < 3 else ()):
            assert np.array_equal(_bits(a[k]["poses"]), _bits(other["poses"])) and np.array_equal(_bits(a[k]["points"]), _bits(other["points"])), k
            assert np.array_equal(_bits(a[k]["edge_chi2"]), _bits(other["edge_chi2"])), k


def test_fast_windows_small_and_degenerate(oracle):
    """two-keyframe initialisation maps, a window whose cameras are all fixed, a window without edges"""
    wins = [_window(synth.small_window_problem(2 + k, 100 + 30 * k, seed=700 + k), 10) for k in range(3)]
    pr = synth.ba_problem(n_kf=6, n_pts=200, k_obs=3, seed=0x3D0, radius=8.0)
    wins.append(_window(pr, 5, n_fixed=6))
    res = capi.ba_optimize_windows(wins, fast=True)
    exact = capi.ba_optimize_windows(wins)
    for k, (g, x) in enumerate(zip(res, exact)):
        # small windows are order-sensitive (tools/ba_sensitivity.py): same trial sequence as the sequential-order kernel, results close
        assert list(g["stats"]["trials"]) == list(x["stats"]["trials"]), k
        assert np.abs(g["poses"] - x["poses"]).max() < 1e-5 and np.abs(g["points"] - x["points"]).max() < 1e-4, k
    e = capi.make_edges(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 2)), np.zeros(0))
    none = dict(poses=pr["poses"], fixed=pr["fixed"], points=pr["points"], edges=e, intrinsics=pr["intrinsics"], huber_delta=DELTA, iterations=3)
    g = capi.ba_optimize_windows([none], fast=True)[0]
    assert np.array_equal(g["points"], pr["points"])


def test_fast_windows_capacity():
    pr = synth.ba_problem(n_kf=40, n_pts=600, k_obs=6, seed=0x3E0, radius=12.0)
    with pytest.raises(capi.DvmError):
        capi.ba_optimize_windows([_window(pr, 3, n_fixed=2)], fast=True)     # 38 free cameras


def test_fast_windows_do_not_depend_on_the_cluster_size(windows, monkeypatch):
    """k_ba_window_cluster: G = 1, 2, 4, 8 workgroups per window -- element ranges are split in eight FIXED parts and every camera / block
    of the reduced system is summed by one wave, so the bits do not depend on G (nor, therefore, on how many windows share the launch)."""
    ref = None
    for G in (1, 2, 4, 8):
        monkeypatch.setenv("DVM_BA_CLUSTER", str(G))
        res = capi.ba_optimize_windows(windows[:5], fast=True)
        if ref is None:
            ref = res
            continue
        for k, (a, b) in enumerate(zip(ref, res)):
            assert np.array_equal(_bits(a["poses"]), _bits(b["poses"])) and np.array_equal(_bits(a["points"]), _bits(b["points"])), (G, k)
            assert np.array_equal(_bits(a["edge_chi2"]), _bits(b["edge_chi2"])) and list(a["stats"]["trials"]) == list(b["stats"]["trials"]), (G, k)
            assert np.array_equal(_bits(a["stats"]["chi2"]), _bits(b["stats"]["chi2"])), (G, k)


@pytest.mark.parametrize("split", [False, True])
def test_fast_windows_large_batch_in_one_launch_or_two_halves(split, monkeypatch):
    """70 windows in one launch, or (DVM_BA_SPLIT) as two concurrent half-batches -- a helper thread with its own staging buffers and stream,
    one cluster size for both launches: window k's bits are those of the same window in a small batch."""
    if split:
        monkeypatch.setenv("DVM_BA_SPLIT", "1")
    base = []
    for k in range(6):
        pr = synth.ba_problem(n_kf=10 + 2 * k, n_pts=250 + 60 * k, k_obs=4, seed=0x4A0 + k, radius=10.0)
        base.append(_window(pr, 6, n_fixed=3))
    ref = capi.ba_optimize_windows(base, fast=True)
    wins = [base[k % 6] for k in range(70)]
    for _ in range(2):
        res = capi.ba_optimize_windows(wins, fast=True)
        for k, g in enumerate(res):
            r = ref[k % 6]
            assert np.array_equal(_bits(g["poses"]), _bits(r["poses"])) and np.array_equal(_bits(g["points"]), _bits(r["points"])), k
            assert np.array_equal(_bits(g["edge_chi2"]), _bits(r["edge_chi2"])) and list(g["stats"]["trials"]) == list(r["stats"]["trials"]), k


def test_fast_windows_repeat_with_one_workgroup_after_a_barrier_timeout(windows, monkeypatch):
    """A cluster barrier that times out (a workgroup not resident: other work holds compute units) makes the call repeat the batch with
    G = 1; DVM_BA_TEST_TIMEOUT takes that path without a real time-out.  Same bits (the results do not depend on G)."""
    ref = capi.ba_optimize_windows(windows[:4], fast=True)
    monkeypatch.setenv("DVM_BA_TEST_TIMEOUT", "1")
    res = capi.ba_optimize_windows(windows[:4], fast=True)
    for k, (a, b) in enumerate(zip(ref, res)):
        assert np.array_equal(_bits(a["poses"]), _bits(b["poses"])) and np.array_equal(_bits(a["points"]), _bits(b["points"])), k
        assert list(a["stats"]["trials"]) == list(b["stats"]["trials"]), k


def test_ba_pool_batches_blocking_calls_of_many_threads(windows):
    """dvm_ba_pool_optimize: LocalMapping threads of several agents, each with a blocking one-window call; the service batches them into
    launches of the cluster form; every caller gets the bits of a solo call (a window's result does not depend on its batch)."""
    import threading
    base = windows[:4]
    ref = capi.ba_optimize_windows(base, fast=True)
    pool = capi.BaPool(max_batch=8, window_us=2000)
    T, rounds = 8, 3
    out = [[None] * rounds for _ in range(T)]
    sizes = []
    def agent(t):
        for r in range(rounds):
            b = capi.BaWindowBatch([base[(t + r) % 4]])
            res, n = pool.optimize(b)
            out[t][r] = dict(poses=res["poses"].copy(), points=res["points"].copy(), trials=list(res["stats"]["trials"]))
            sizes.append(n)
    th = [threading.Thread(target=agent, args=(t,)) for t in range(T)]
    for x in th: x.start()
    for x in th: x.join()
    for t in range(T):
        for r in range(rounds):
            g, a = out[t][r], ref[(t + r) % 4]
            assert np.array_equal(_bits(g["poses"]), _bits(a["poses"])) and np.array_equal(_bits(g["points"]), _bits(a["points"])), (t, r)
            assert g["trials"] == list(a["stats"]["trials"]), (t, r)
    assert max(sizes) > 1, sizes          # calls did ride together
    pool.close()


def test_fast_windows_do_not_depend_on_the_placement(windows, monkeypatch):
    """The cluster's workgroups normally sit on one XCD (block b runs on XCD b % 8: speed only).  DVM_BA_CLUSTER_SCATTER maps a window's
    eight workgroups to CONSECUTIVE blocks -- one per XCD, whose L2s are not coherent with each other --: the same bits, run after run
    (the phase barriers release and acquire at agent scope; nothing relies on co-location)."""
    ref = capi.ba_optimize_windows(windows[:6], fast=True)
    monkeypatch.setenv("DVM_BA_CLUSTER_SCATTER", "1")
    for rep in range(4):
        res = capi.ba_optimize_windows(windows[:6], fast=True)
        for k, (a, b) in enumerate(zip(ref, res)):
            assert np.array_equal(_bits(a["poses"]), _bits(b["poses"])) and np.array_equal(_bits(a["points"]), _bits(b["points"])), (rep, k)
            assert np.array_equal(_bits(a["edge_chi2"]), _bits(b["edge_chi2"])) and list(a["stats"]["trials"]) == list(b["stats"]["trials"]), (rep, k)


def test_fast_windows_with_tables_in_blocks_of_their_own(oracle, windows, monkeypatch):
    """The windows' tables travel in one page-locked block sized from the edge counts before they are built (csrc/ba_window.hip:
    shared_tables); a window that finds it full -- its (edge, edge) pair list far beyond the estimate -- packs into a block of its own,
    sent on its own.  DVM_BA_TEST_PRIVATE_TABLES makes every second window of a call take that path: the same bits as without it."""
    pr = synth.ba_problem(n_kf=32, n_pts=260, k_obs=32, seed=0x3D1, radius=12.0)
    dense = _window(pr, 6, n_fixed=2)          # every landmark seen by all 30 free cameras: 465 pairs a landmark
    assert len(dense["edges"]) == 32 * 260
    batch = [windows[0], dense, windows[5], windows[1]]
    ref = capi.ba_optimize_windows(batch, fast=True)
    _check_vs_oracle(oracle, dense, ref[1], "dense window")
    monkeypatch.setenv("DVM_BA_TEST_PRIVATE_TABLES", "1")
    res = capi.ba_optimize_windows(batch, fast=True)
    for a, b in zip(ref, res):
        for key in ("poses", "points", "edge_chi2"):
            assert np.array_equal(_bits(a[key]), _bits(b[key])), key
        assert np.array_equal(a["depth_positive"], b["depth_positive"])
