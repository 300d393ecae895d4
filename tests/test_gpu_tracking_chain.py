"""BASELINE.json config 2 as a chain on one GPU: constant-velocity prediction -> SearchByProjection(Cur, Last) ->
PoseOptimization per frame, local BA every fifth frame (dvm_slam_amd/tracking.py), once over the HIP library and once over
the CPU oracle: identical assignments frame by frame, poses / landmarks within 1e-6, and the trajectory is the true one."""
import numpy as np
import pytest

from tracking_scene import make_sequence

pytestmark = pytest.mark.gpu


class OracleOps:
    def __init__(self, po):
        self.po = po

    def search_by_projection(self, cur, last, mps, Tcw, K, bounds, scale, th):
        return self.po.search_by_projection_frames(cur["kps"], cur["desc"], cur["mp"], Tcw, K, bounds, scale, last["kps"], last["mp"],
                                                   last.get("outlier"), mps, th, True)

    def pose_optimize(self, pose, Xw, obs, w, K):
        return self.po.pose_optimize(pose, Xw, obs, w, K)

    def frustum_frame(self):
        return self.po.FrustumFrame()

    def pose_matrices(self, Tcw):
        return self.po.pose_matrices(Tcw)

    def is_in_frustum(self, F, P, normal, dmin, dmax):
        return self.po.is_in_frustum(F, P, normal, dmin, dmax, 0.5)

    def tracked_dtype(self):
        return self.po.TRACKED_POINT_DTYPE

    def search_local_points(self, cur, claimed, bounds, scale, pts, th, nnratio):
        return self.po.search_by_projection_points(cur["kps"], cur["desc"], cur["mp"], claimed, bounds, scale, pts, th, nnratio, False, 0.0)

    def local_ba(self, poses, fixed, points, edges, K, delta, iters):
        p, x, st, _ = self.po.ba_optimize(poses, fixed, points, self.po.make_edges(*edges), K, delta, iters)
        return p, x, st["iterations"]


@pytest.mark.parametrize("seed", [0, 1])
def test_tracking_and_local_ba_chain(capi, oracle, seed):
    from dvm_slam_amd import tracking
    sc = make_sequence(oracle.KP_DTYPE, seed)
    pose0 = tracking.pose7(*sc["gt"][0])
    K32 = sc["K"].astype(np.float32)
    out = {}
    for name, ops, dt in (("gpu", tracking.GpuOps(), capi.MAP_POINT_DTYPE), ("cpu", OracleOps(oracle), oracle.MAP_POINT_DTYPE)):
        out[name] = tracking.track(ops, sc["frames"], sc["map_points"], dt, K32, sc["bounds"], sc["scale"], sc["inv_sigma2"], pose0, sc["mp0"])
    g, c = out["gpu"], out["cpu"]
    assert g["nmatch"] == c["nmatch"] and g["ninl"] == c["ninl"] and g["nlocal"] == c["nlocal"] and min(g["ninl"]) > 250
    for t, (a, b) in enumerate(zip(g["assign"], c["assign"])):
        assert np.array_equal(a, b), t
    assert np.abs(g["poses"] - c["poses"]).max() < 1e-6 and np.abs(g["X"] - c["X"]).max() < 1e-6
    assert len(g["lba"]) == len(c["lba"]) == 3
    for a, b in zip(g["lba"], c["lba"]):
        assert (a["n_points"], a["n_edges"], a["iterations"]) == (b["n_points"], b["n_edges"], b["iterations"]) and a["n_edges"] > 2000
        assert np.abs(a["poses"] - b["poses"]).max() < 1e-6
    # the chain tracks: final pose close to the truth, assignments are the true points
    Rt, tt = sc["gt"][-1]
    Rg, tg = tracking.rt_of(g["poses"][-1])
    assert np.linalg.norm(Rg - Rt) < 0.01 and np.linalg.norm(tg - tt) < 0.05
    last = g["assign"][-1]
    hit = last >= 0
    assert hit.sum() > 300 and np.mean(last[hit] == sc["frames"][-1]["pt"][hit]) > 0.98
