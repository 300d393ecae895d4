"""BASELINE.json config 2 end to end FROM PIXELS: a textured plane rendered to 640x480 frames (tests/pixel_scene.py) ->
ORB extraction of every frame -> constant-velocity prediction -> SearchByProjection(Cur, Last) -> PoseOptimization -> TrackLocalMap
(isInFrustum + SearchByProjection(F, points) + PoseOptimization) per frame, LocalBundleAdjustment every fifth frame
(dvm_slam_amd/tracking.py; reference Frame.cc:371-411, Tracking.cc:2584-2667, :3100-3230, Optimizer.cc:1030-1387) -- once over the HIP
library, once over the CPU oracle, each with ITS OWN extraction: identical keypoints and descriptors per frame, identical assignments
frame by frame, poses / landmarks within 1e-6, and the trajectory is the rendered one.  The one-chain entry point
(dvmh_track_with_motion_model) is run beside it on the same frames and must give the first stage's matches and pose."""
import numpy as np
import pytest

import pixel_scene as ps
from test_gpu_tracking_chain import OracleOps

pytestmark = pytest.mark.gpu
N_FRAMES = 11
BOUNDS = np.array([0.0, 640.0, 0.0, 480.0], np.float32)


def _map_from_first_frame(kps, desc, R, t, scale, rng):
    X = ps.backproject(kps, R, t) + rng.normal(0, 0.01, (len(kps), 3))          # an imperfect map: 1 cm
    Ow = -R.T @ t
    v = X - Ow
    d = np.linalg.norm(v, axis=1)
    return dict(pos=X, desc=desc.copy(), n_obs=np.full(len(kps), 3, np.int32), normal=(v / d[:, None]).astype(np.float32),
                max_dist=(d * scale[kps["octave"]]).astype(np.float32), min_dist=(d * scale[kps["octave"]] / scale[7]).astype(np.float32))


def test_config2_chain_from_pixels(capi, oracle):
    from dvm_slam_amd import tracking
    frames_px, gt = ps.render(N_FRAMES)
    ext = capi.OrbExtractor(max_batch=1)
    orc = oracle.OrbOracle()
    tab = ext.tables()
    scale, inv_s2 = tab["scale"], tab["inv_sigma2"]
    fr_g, fr_c = [], []
    for t in range(N_FRAMES):
        ng, kg, dg, _ = ext.extract(frames_px[t])
        nc, kc, dc, _ = orc.extract(frames_px[t])
        assert ng == nc and ng > 900, (t, ng, nc)
        for f in ("x", "y", "size", "angle", "response", "octave"):
            assert np.array_equal(kg[f], kc[f]), (t, f)
        assert np.array_equal(dg, dc), t
        fr_g.append(dict(kps=kg.copy(), desc=dg.copy())); fr_c.append(dict(kps=kc.copy(), desc=dc.copy()))
    rng = np.random.default_rng(11)
    mpts = _map_from_first_frame(fr_g[0]["kps"], fr_g[0]["desc"], *gt[0], scale, rng)
    pose0 = ps.pose7(*gt[0])
    mp0 = np.arange(len(fr_g[0]["kps"]), dtype=np.int32)
    K32 = ps.K.astype(np.float32)
    out = {}
    for name, ops, dt, frames in (("gpu", tracking.GpuOps(), capi.MAP_POINT_DTYPE, fr_g), ("cpu", OracleOps(oracle), oracle.MAP_POINT_DTYPE, fr_c)):
        out[name] = tracking.track(ops, frames, mpts, dt, K32, BOUNDS, scale, inv_s2, pose0, mp0, lba_every=5, window=6)
    g, c = out["gpu"], out["cpu"]
    assert g["nmatch"] == c["nmatch"] and g["ninl"] == c["ninl"] and g["nlocal"] == c["nlocal"], (g["nmatch"], c["nmatch"], g["ninl"], c["ninl"])
    assert min(g["ninl"]) > 150, g["ninl"]                              # a real tracking run: hundreds of inliers in every frame
    for t, (a, b) in enumerate(zip(g["assign"], c["assign"])):
        assert np.array_equal(a, b), t
    assert np.abs(g["poses"] - c["poses"]).max() < 1e-6 and np.abs(g["X"] - c["X"]).max() < 1e-6
    assert len(g["lba"]) == len(c["lba"]) == 2
    for a, b in zip(g["lba"], c["lba"]):
        assert (a["n_points"], a["n_edges"], a["iterations"]) == (b["n_points"], b["n_edges"], b["iterations"]) and a["n_edges"] > 1000
        assert np.abs(a["poses"] - b["poses"]).max() < 1e-6
    # the chain tracks the rendered camera
    for t in (5, N_FRAMES - 1):
        Rg, tg = tracking.rt_of(g["poses"][t])
        assert np.linalg.norm(Rg - gt[t][0]) < 0.01 and np.linalg.norm(tg - gt[t][1]) < 0.06, (t, tg, gt[t][1])
    # the one-chain entry point on frame 1: pixels in -> the matches of SearchByProjection(Cur, Last) + PoseOptimization, as the first
    # stage of the chain above computes them from the separately extracted frame
    trk = capi.Tracker(ext)
    mps = np.zeros(len(mpts["pos"]), capi.MAP_POINT_DTYPE)
    mps["pos"], mps["desc"], mps["n_obs"] = mpts["pos"].astype(np.float32), mpts["desc"], mpts["n_obs"]
    from dvm_slam_amd import synth
    Tcw = synth.se3_from_Rt(*gt[0])
    r = trk.track(frames_px[1], Tcw, K32, BOUNDS, scale, inv_s2, fr_g[0]["kps"], mp0, None, mps, th=15.0)
    assert r["tracked"] == 1 and r["nmatches_search"] == g["nmatch"][0] and r["replayed_on_host"] == 0
    for f in ("x", "y", "angle", "octave"):
        assert np.array_equal(r["kps"][f], fr_g[1]["kps"][f]), f
    trk.close(); ext.close()
