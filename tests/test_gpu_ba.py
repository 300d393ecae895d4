"""GPU parity: HIP bundle adjustment (through the C ABI) vs the CPU oracle.

Tolerance (BASELINE.json north_star): optimised camera poses and landmarks within 1e-6 of the
reference path, compared in FP64 BEFORE the reference's f64->f32 write-back (Optimizer.cc:1374);
the LM trial sequence (accept/reject per iteration) must be identical.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-6


def _run_both(capi, oracle, pr, delta, iters):
    eo = oracle.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    po_, pto, so, chio = oracle.ba_optimize(pr["poses"], pr["fixed"], pr["points"], eo, pr["intrinsics"], delta, iters)
    ba = capi.BundleAdjuster()
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], eo, pr["intrinsics"], delta)
    sg = ba.optimize(iters)
    pg, ptg = ba.result()
    chig, depth = ba.edge_chi2()
    ba.close()
    return (po_, pto, so, chio), (pg, ptg, sg, chig, depth)


@pytest.mark.parametrize("n_kf,n_pts,delta,iters", [
    (12, 300, np.sqrt(5.991), 10),   # LocalBundleAdjustment-sized window, Huber on
    (40, 1500, np.sqrt(5.991), 10),
    (40, 1500, 0.0, 10),             # GlobalBundleAdjustemnt(bRobust=false)
    (100, 4000, np.sqrt(5.991), 5),
    (33, 777, np.sqrt(5.991), 20),   # n+1 not a multiple of the 64-wide Cholesky block
])
def test_ba_matches_oracle(capi, oracle, n_kf, n_pts, delta, iters):
    from dvm_slam_amd import synth
    pr = synth.ba_problem(n_kf=n_kf, n_pts=n_pts, seed=n_kf * 31 + n_pts)
    (po_, pto, so, chio), (pg, ptg, sg, chig, depth) = _run_both(capi, oracle, pr, delta, iters)
    assert sg["iterations"] == so["iterations"]
    assert sg["trials"] == so["trials"], "LM accept/reject sequence differs"
    assert sg["stop_reason"] == so["stop_reason"]
    assert abs(sg["chi2_initial"] - so["chi2_initial"]) <= 1e-9 * so["chi2_initial"]
    assert np.allclose(sg["chi2"], so["chi2"], rtol=1e-9)
    assert np.allclose(sg["lam"], so["lam"], rtol=1e-6)
    assert np.abs(pg - po_).max() < POSE_TOL, np.abs(pg - po_).max()
    assert np.abs(ptg - pto).max() < POSE_TOL, np.abs(ptg - pto).max()
    assert np.allclose(chig, chio, rtol=1e-6, atol=1e-9)
    # fixed camera untouched, quaternions normalised with w >= 0
    assert np.array_equal(pg[0], po_[0])
    assert np.allclose(np.linalg.norm(pg[:, 3:], axis=1), 1.0, atol=1e-12) and (pg[:, 6] >= 0).all()
    assert depth.all()


def test_ba_fixed_cameras_and_unobserved(capi, oracle):
    """LBA shape: several fixed cameras observing the window's landmarks, a camera and a landmark without edges."""
    from dvm_slam_amd import synth
    pr = synth.ba_problem(n_kf=30, n_pts=900, seed=5)
    pr["fixed"][:8] = 1                       # "lFixedCameras"
    P, L = len(pr["poses"]), len(pr["points"])
    pr["poses"] = np.vstack([pr["poses"], pr["poses"][-1:]])      # camera P: no observation at all
    pr["fixed"] = np.append(pr["fixed"], 0).astype(np.uint8)
    pr["points"] = np.vstack([pr["points"], [[1.0, 2.0, 3.0]]])   # landmark L: never observed
    (po_, pto, so, chio), (pg, ptg, sg, chig, depth) = _run_both(capi, oracle, pr, np.sqrt(5.991), 10)
    assert sg["trials"] == so["trials"]
    assert np.abs(pg - po_).max() < POSE_TOL and np.abs(ptg - pto).max() < POSE_TOL
    assert np.array_equal(pg[:8], pr["poses"][:8] / 1.0) or np.allclose(pg[:8], po_[:8], atol=0)
    assert np.array_equal(ptg[L], [1.0, 2.0, 3.0]) and np.array_equal(pg[P], po_[P])


def test_ba_stop_flag(capi, oracle):
    from dvm_slam_amd import synth
    pr = synth.ba_problem(n_kf=12, n_pts=300, seed=2)
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    ba = capi.BundleAdjuster()
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], np.sqrt(5.991))
    flag = np.ones(1, np.uint8)       # *pbStopFlag already set: optimize() performs no iteration
    st = ba.optimize(10, stop_flag=flag)
    assert st["iterations"] == 0
    p, _ = ba.result()
    assert np.allclose(p, pr["poses"], atol=1e-15)
    ba.close()


def test_ba_full_size_properties(capi):
    """BASELINE size (500 KF / 20 000 landmarks / 160 000 edges): size-independent properties."""
    from dvm_slam_amd import synth
    pr = synth.ba_problem()
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    ba = capi.BundleAdjuster()
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], np.sqrt(5.991))
    st = ba.optimize(10)
    p, pts = ba.result()
    chi, depth = ba.edge_chi2()
    # monotone accepted chi2, fixed gauge camera untouched, unit quaternions, finite state
    assert all(b <= a * (1 + 1e-12) for a, b in zip([st["chi2_initial"]] + st["chi2"][:-1], st["chi2"]))
    assert np.array_equal(p[0], pr["poses"][0])
    assert np.isfinite(p).all() and np.isfinite(pts).all()
    assert np.allclose(np.linalg.norm(p[:, 3:], axis=1), 1.0, atol=1e-12)
    # determinism: a second identical run is bitwise identical
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], np.sqrt(5.991))
    st2 = ba.optimize(10)
    p2, pts2 = ba.result()
    assert st2["trials"] == st["trials"] and np.array_equal(p, p2) and np.array_equal(pts, pts2)
    # the optimum is better than the perturbed start for the inlier edges
    assert st["chi2_final"] < 0.9 * st["chi2_initial"]
    ba.close()
