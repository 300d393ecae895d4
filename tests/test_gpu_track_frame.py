"""dvm_track_begin / dvm_track_finish (dvmh_track_with_motion_model): the tracking step of one frame as ONE device chain, from pixels.
Reference: Frame::Frame -> ExtractORB (src/Frame.cc:371-411), Tracking::TrackWithMotionModel (src/Tracking.cc:2584-2667).
Checked against (a) the three separate calls of this library (bit for bit) and (b) the CPU oracle's chain (keypoints, assignments and
outlier flags identical; pose within 1e-6, the tolerance of PoseOptimization's own parity test)."""
import numpy as np
import pytest

import pixel_scene as ps

pytestmark = pytest.mark.gpu

BOUNDS = np.array([0, 640, 0, 480], np.float32)


def _tcw7f(p):   # (t, q) doubles -> dvm_se3f (q, t) floats
    return np.concatenate([p[3:7], p[0:3]]).astype(np.float32)


def _map_from_frame(capi, kps, desc, R, t, rng, noise=0.01, p_obs0=0.0):
    X = ps.backproject(kps, R, t) + rng.normal(0, noise, (len(kps), 3))
    mps = np.zeros(len(kps), capi.MAP_POINT_DTYPE)
    mps["pos"] = X.astype(np.float32); mps["desc"] = desc
    mps["n_obs"] = np.where(rng.random(len(kps)) < p_obs0, 0, 1)
    return mps


def _separate_calls(ops_extract, sbp, pose_opt, img, Tcw_pred, scale, inv_s2, kps_l, mp_l, mps, th):
    """extract -> SearchByProjection (doubled window below 20 matches) -> PoseOptimization -> outlier drop, as Tracking does it"""
    n, kps, desc, mono = ops_extract(img)
    mp0 = np.full(n, -1, np.int32)
    nm, mp = sbp(kps, desc, mp0, Tcw_pred, kps_l, mp_l, mps, th)
    wide = 0
    if nm < 20:
        wide = 1
        nm, mp = sbp(kps, desc, mp0, Tcw_pred, kps_l, mp_l, mps, 2 * th)
    out = dict(n=n, kps=kps, desc=desc, mp=mp.copy(), dropped=np.full(n, -1, np.int32), nmatches_search=nm, wide_window=wide, tracked=int(nm >= 20))
    if nm < 20:
        out.update(nmatches=nm)
        return out
    sel = np.flatnonzero(mp >= 0)
    Xw = mps["pos"][mp[sel]].astype(np.float64)
    obs = np.column_stack([kps["x"][sel], kps["y"][sel]]).astype(np.float64)
    w = inv_s2[kps["octave"][sel]].astype(np.float64)
    pose_in = np.concatenate([Tcw_pred[4:7], Tcw_pred[0:4]]).astype(np.float64)
    pose, outl, ninl = pose_opt(pose_in, Xw, obs, w)
    rej = sel[np.asarray(outl[:len(sel)]) != 0]
    out["dropped"][rej] = mp[rej]; out["mp"][rej] = -1
    keep = sel[np.asarray(outl[:len(sel)]) == 0]
    out.update(pose=pose, n_inliers=int(ninl), nmatches=nm - len(rej), nmatches_map=int((mps["n_obs"][mp[keep]] > 0).sum()))
    return out


@pytest.fixture(scope="module")
def scene():
    frames, poses = ps.render(12)
    return frames, poses


def _check(a, b, exact_pose):
    for k in ("n", "nmatches", "nmatches_search", "wide_window", "tracked"):
        assert a[k] == b[k], (k, a[k], b[k])
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(a["kps"][f], b["kps"][f]), f
    assert np.array_equal(a["desc"], b["desc"])
    assert np.array_equal(a["mp"], b["mp"]) and np.array_equal(a["dropped"], b["dropped"])
    if a["tracked"]:
        assert a["n_inliers"] == b["n_inliers"] and a["nmatches_map"] == b["nmatches_map"]
        if exact_pose:
            assert np.array_equal(a["pose"], b["pose"])
        else:
            assert np.abs(a["pose"] - b["pose"]).max() < 1e-6


@pytest.mark.parametrize("p_obs0,th", [(0.0, 15.0), (0.3, 15.0), (0.0, 2.0)])
def test_track_frame_equals_separate_calls_and_oracle(scene, p_obs0, th):
    from dvm_slam_amd import capi
    from oracle import pyoracle as po
    frames, poses = scene
    ext = capi.OrbExtractor(max_batch=1)
    tab = ext.tables()
    scale, inv_s2 = tab["scale"], tab["inv_sigma2"]
    trk = capi.Tracker(ext)
    orc = po.OrbOracle()
    rng = np.random.default_rng(3)
    n0, k0, d0, _ = ext.extract(frames[0])
    mps = _map_from_frame(capi, k0, d0, *poses[0], rng, p_obs0=p_obs0)
    kps_l, mp_l = k0, np.arange(n0, dtype=np.int32)
    mp_l[rng.random(n0) < 0.1] = -1
    for t in (1, 2, 3):
        Tcw_pred = _tcw7f(ps.pose7(*poses[t - 1]))          # zero-velocity prediction: the last frame's pose
        fused = trk.track(frames[t], Tcw_pred, ps.K, BOUNDS, scale, inv_s2, kps_l, mp_l, None, mps, th=th)
        sep = _separate_calls(lambda im: ext.extract(im),
                              lambda k, d, m, T, kl, ml, mp_, th_: capi.search_by_projection_frames(k, d, m, T, ps.K, BOUNDS, scale, kl, ml, None, mp_, th_)[:2],
                              lambda p, X, o, w: [r[0] for r in capi.pose_optimize(p[None], X[None], o[None], w[None], [len(X)], ps.K)],
                              frames[t], Tcw_pred, scale, inv_s2, kps_l, mp_l, mps, th)
        _check(fused, sep, exact_pose=True)
        orc_out = _separate_calls(lambda im: orc.extract(im),
                                  lambda k, d, m, T, kl, ml, mp_, th_: po.search_by_projection_frames(k, d, m, T, ps.K, BOUNDS, scale, kl, ml, None, mp_, th_)[:2],
                                  lambda p, X, o, w: po.pose_optimize(p, X, o, w, ps.K),
                                  frames[t], Tcw_pred, scale, inv_s2, kps_l, mp_l, mps, th)
        _check(fused, orc_out, exact_pose=False)
        if th == 15.0 and p_obs0 == 0.0:
            assert fused["tracked"] and fused["nmatches"] > 200      # a real tracking step, not a degenerate one
            gt = ps.pose7(*poses[t])
            assert np.abs(fused["pose"][:3] - gt[:3]).max() < 0.05   # and it finds the camera (map noise 1 cm, pixel noise of ORB)
    trk.close(); ext.close()


def test_track_frame_dense_stream_requeries_on_device():
    """The bench stream is dense enough that some query of every frame finds all four ranked candidates taken by earlier queries: the
    device searches those windows again at the query's turn (k_track_claims), nothing is replayed on the host, and the results are
    those of the separate calls (whose host epilogue re-queries the same windows) and of the oracle chain."""
    from dvm_slam_amd import capi, synth
    from oracle import pyoracle as po
    frames = synth.frame_stream(4)
    ext = capi.OrbExtractor(max_batch=1)
    tab = ext.tables()
    scale, inv_s2 = tab["scale"], tab["inv_sigma2"]
    trk = capi.Tracker(ext)
    orc = po.OrbOracle()
    rng = np.random.default_rng(9)
    Kc = np.array([500.0, 500.0, 320.0, 240.0], np.float32)
    Tcw = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    total_rq = 0
    for t in (1, 2, 3):
        n0, k0, d0, _ = ext.extract(frames[t - 1])
        z = rng.uniform(3, 9, n0).astype(np.float32)
        mps = np.zeros(n0, capi.MAP_POINT_DTYPE)
        mps["pos"][:, 0] = (k0["x"] - Kc[2]) / Kc[0] * z; mps["pos"][:, 1] = (k0["y"] - Kc[3]) / Kc[1] * z; mps["pos"][:, 2] = z
        mps["desc"] = d0; mps["n_obs"] = 1
        mp_l = np.arange(n0, dtype=np.int32)
        fused = trk.track(frames[t], Tcw, Kc, BOUNDS, scale, inv_s2, k0, mp_l, None, mps, th=15.0)
        assert fused["replayed_on_host"] == 0
        total_rq += fused["n_requeried"]
        sep = _separate_calls(lambda im: ext.extract(im),
                              lambda k, d, m, T, kl, ml, mp_, th_: capi.search_by_projection_frames(k, d, m, T, Kc, BOUNDS, scale, kl, ml, None, mp_, th_)[:2],
                              lambda p, X, o, w: [r[0] for r in capi.pose_optimize(p[None], X[None], o[None], w[None], [len(X)], Kc)],
                              frames[t], Tcw, scale, inv_s2, k0, mp_l, mps, 15.0)
        _check(fused, sep, exact_pose=True)
        orc_out = _separate_calls(lambda im: orc.extract(im),
                                  lambda k, d, m, T, kl, ml, mp_, th_: po.search_by_projection_frames(k, d, m, T, Kc, BOUNDS, scale, kl, ml, None, mp_, th_)[:2],
                                  lambda p, X, o, w: po.pose_optimize(p, X, o, w, Kc),
                                  frames[t], Tcw, scale, inv_s2, k0, mp_l, mps, 15.0)
        _check(fused, orc_out, exact_pose=False)
    assert total_rq > 0       # the case this test is for did occur
    trk.close(); ext.close()


def test_track_batch_equals_single_frames():
    """dvmh_track_with_motion_model_batch: the frames of several agents (different streams, different maps, different numbers of queries, one
    of them with too few matches -> the doubled window) through ONE chain of batched launches: every frame's outputs equal the single call's."""
    from dvm_slam_amd import capi, synth
    B = 6
    ext1 = capi.OrbExtractor(max_batch=1)
    extB = capi.OrbExtractor(max_batch=B)
    tab = ext1.tables()
    scale, inv_s2 = tab["scale"], tab["inv_sigma2"]
    trk1 = capi.Tracker(ext1)
    trkB = capi.TrackerBatch(extB, B)
    dense = synth.frame_stream(8)
    low = synth.frame_stream(4, texture="low")
    rng = np.random.default_rng(21)
    Kc = np.array([500.0, 500.0, 320.0, 240.0], np.float32)
    imgs, Ts, lasts, ths = [], [], [], []
    for b in range(B):
        stream, t = (low, 1 + b % 3) if b == 2 else (dense, 1 + b)
        n0, k0, d0, _ = ext1.extract(stream[t - 1])
        z = rng.uniform(3, 9, n0).astype(np.float32)
        mps = np.zeros(n0, capi.MAP_POINT_DTYPE)
        mps["pos"][:, 0] = (k0["x"] - Kc[2]) / Kc[0] * z; mps["pos"][:, 1] = (k0["y"] - Kc[3]) / Kc[1] * z; mps["pos"][:, 2] = z
        mps["desc"] = d0; mps["n_obs"] = np.where(rng.random(n0) < 0.2, 0, 1)
        mp_l = np.arange(n0, dtype=np.int32)
        mp_l[rng.random(n0) < (0.992 if b == 4 else 0.1 * b)] = -1        # agent 4 carries very few map points: fewer than 20 matches, the wide window
        imgs.append(stream[t]); Ts.append(np.array([0, 0, 0, 1, 0.002 * b, 0, 0], np.float32)); lasts.append((k0.copy(), mp_l, None, mps))
    ins, keep = trkB.prepare(Ts, lasts)
    got = trkB.track(np.stack(imgs), ins, Kc, BOUNDS, scale, inv_s2, th=15.0)
    wide = 0
    for b in range(B):
        k0, mp_l, _, mps = lasts[b]
        one = trk1.track(imgs[b], Ts[b], Kc, BOUNDS, scale, inv_s2, k0, mp_l, None, mps, th=15.0)
        _check(got[b], one, exact_pose=True)
        assert got[b]["n_requeried"] == one["n_requeried"] and got[b]["replayed_on_host"] == 0
        wide += got[b]["wide_window"]
    assert wide >= 1, [(g["nmatches_search"], g["wide_window"], g["tracked"]) for g in got]
    trkB.close(); trk1.close(); extB.close(); ext1.close()
