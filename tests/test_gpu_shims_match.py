"""The reference-side binding of the front end, EXECUTED: ORB_SLAM3::ORBmatcher (dvm_slam_amd/host/ORBmatcher_shim.h),
ORB_SLAM3::ORBextractor::operator() (ORBextractor_shim.h) and Frame::isInFrustum (Frame_grid_shim.h) compiled against the
behaving mock Frame / KeyFrame / MapPoint classes of tests/stubs/, linked with libdvmslam_host.so + libdvmslam_hip.so, and run
the way Tracking / LocalMapping / LoopClosing call them -- on pointers.  What they leave in mvpMapPoints / vpMatches / the map
graph must equal the sequential CPU oracle's output on the same scene, entry for entry (reference: ORBmatcher.cc:44-1860,
ORBextractor.cc:876-955, Frame.cc:575-636)."""
import ctypes as C

import numpy as np
import pytest

import shim_world as sw
from matcher_scene import make_init_scene, make_kf_pair_scene, make_local_map_scene, make_scene

pytestmark = pytest.mark.gpu
f32 = C.c_float


def _tables(W, scale):
    s = np.ascontiguousarray(scale, np.float32)
    g = (s * s).astype(np.float32)
    W.tables = (s, g, (np.float32(1.0) / g).astype(np.float32))


def _tq(T):
    """scene poses are Sophus-style (qx qy qz qw tx ty tz); the world helpers take (t, q)."""
    T = np.asarray(T, np.float32)
    return np.concatenate([T[4:7], T[:4]])


def _fv_args(fv):
    n, o, f = sw._i32(fv["fv_nodes"]), sw._i32(fv["fv_off"]), sw._i32(fv["fv_feat"])
    return len(n), sw._p(n), sw._p(o), sw._p(f), (n, o, f)


# ----------------------------------------------------------------------------------------------------- frame <-> frame / points
@pytest.mark.parametrize("seed,th,ori", [(0, 15.0, True), (2, 15.0, False)])
def test_search_by_projection_last_frame(capi, oracle, seed, th, ori):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) as TrackWithMotionModel calls it (Tracking.cc:2610)."""
    sc = make_scene(oracle, seed)
    n_o, mp_o = oracle.search_by_projection_frames(th=th, check_ori=ori, **sc)
    W = sw.World(); W.add_map(0); _tables(W, sc["scale_factors"])
    for i, m in enumerate(sc["mps"]):
        W.add_mappoint(0, i, m["pos"], desc=m["desc"])
        W.L.sw_mp_set_obs_count(W.h, i, int(m["n_obs"]))
    cur = W.add_frame(_tq(sc["Tcw"]), sc["K"], sc["kps_c"], sc["desc_c"], bounds=sc["bounds"])
    last = W.add_frame(np.array([0, 0, 0, 0, 0, 0, 1], np.float32), sc["K"], sc["kps_l"], None, bounds=sc["bounds"])
    W.frame_set_matches(cur, sc["mp_c"])
    W.frame_set_matches(last, sc["mp_l"], outlier=sc["outlier_l"])
    n = W._chk(W.L.sw_search_by_projection_last(W.h, cur, last, f32(th), f32(0.9), int(ori)))
    assert n == n_o > 300
    assert np.array_equal(W.get_frame(cur)["mp"], mp_o)
    assert np.array_equal(W.get_frame(last)["mp"], sc["mp_l"])


@pytest.mark.parametrize("seed,th,ratio,far", [(1, 3.0, 0.8, False), (2, 5.0, 0.9, True)])
def test_search_by_projection_local_map(capi, oracle, seed, th, ratio, far):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints) as SearchLocalPoints calls it (Tracking.cc:3103):
    the mTrack* members live on the map points; keypoints that already hold a point with observations are taken."""
    sc = make_local_map_scene(oracle, seed)
    n_o, mp_o = oracle.search_by_projection_points(th=th, nnratio=ratio, far_points=far, th_far=9.0, **sc)
    pts = sc["pts"]
    W = sw.World(); W.add_map(0); _tables(W, sc["scale_factors"])
    for i, p in enumerate(pts):
        W.add_mappoint(0, i, [0, 0, 0], desc=p["desc"], bad=bool(p["bad"]))
        W.L.sw_mp_set_obs_count(W.h, i, int(p["n_obs"]))
        W.L.sw_mp_set_track(W.h, i, f32(p["proj_x"]), f32(p["proj_y"]), f32(p["depth"]), f32(p["view_cos"]), int(p["level"]), int(p["in_view"]))
    # keypoints that are associated at entry hold their own placeholder point with (claimed) or without observations
    held = np.flatnonzero(sc["mp"] >= 0)
    mp_in = np.full(len(sc["kps"]), -1, np.int32)
    for j in held:
        mp_in[j] = W.add_mappoint(0, 100000 + int(j), [0, 0, 0])
        W.L.sw_mp_set_obs_count(W.h, int(mp_in[j]), int(sc["claimed_obs"][j]))
    F = W.add_frame(np.array([0, 0, 0, 0, 0, 0, 1], np.float32), (500, 500, 320, 240), sc["kps"], sc["desc"], bounds=sc["bounds"])
    W.frame_set_matches(F, mp_in)
    n = W._chk(W.L.sw_search_by_projection_points(W.h, F, sw._p(sw._i32(np.arange(len(pts)))), len(pts), f32(th), int(far), f32(9.0), f32(ratio)))
    got = W.get_frame(F)["mp"].copy()
    keep = got >= len(pts)                                       # still the placeholder it held at entry
    got[keep] = sc["mp"][keep]
    assert n == n_o > 100 and np.array_equal(got, mp_o)


@pytest.mark.parametrize("seed,window", [(0, 100), (2, 30)])
def test_search_for_initialization(capi, oracle, seed, window):
    sc = make_init_scene(oracle, seed)
    n_o, m_o, pm_o = oracle.search_for_initialization(sc["k1"], sc["d1"], sc["k2"], sc["d2"], sc["bounds"], sc["prev_matched"], window, 0.9, True)
    W = sw.World(); W.add_map(0); _tables(W, sc["scale_factors"])
    ident = np.array([0, 0, 0, 0, 0, 0, 1], np.float32)
    a = W.add_frame(ident, (500, 500, 320, 240), sc["k1"], sc["d1"], bounds=sc["bounds"])
    b = W.add_frame(ident, (500, 500, 320, 240), sc["k2"], sc["d2"], bounds=sc["bounds"])
    pm = np.ascontiguousarray(sc["prev_matched"], np.float32).copy(); m = np.zeros(len(sc["k1"]), np.int32)
    n = W._chk(W.L.sw_search_for_initialization(W.h, a, b, sw._p(pm), sw._p(m), window, f32(0.9), 1))
    assert n == n_o and np.array_equal(m, m_o) and np.array_equal(pm, pm_o)


def test_is_in_frustum_point_by_point_and_batched(capi, oracle):
    """Frame::isInFrustum(pMP, 0.5) for every local map point (the reference's loop, Tracking.cc:3041-3103) and the batched form:
    the mTrack* members written on the map points equal the oracle's, field for field."""
    from dvm_slam_amd import synth
    rng = np.random.default_rng(8)
    n = 600
    ang = 0.3
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    tcw = np.array([0.2, -0.1, 0.5], np.float32)
    Tcw = synth.se3_from_Rt(Rcw, tcw)
    K = (149.0, 149.0, 320.0, 240.0)
    P = rng.uniform(-15, 15, (n, 3)).astype(np.float32)
    normal = rng.normal(size=(n, 3)).astype(np.float32); normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    maxd = rng.uniform(5, 30, n).astype(np.float32); mind = (maxd / np.float32(1.2) ** 7).astype(np.float32)
    Fo = oracle.make_frustum_frame(Tcw, K)
    ref = oracle.is_in_frustum(Fo, P, normal, mind, maxd, 0.5)
    assert 20 < ref["in_view"].sum() < n
    for batched in (0, 1):
        W = sw.World(); W.add_map(0)
        for i in range(n):
            W.add_mappoint(0, i, P[i], normal=normal[i], min_dist=float(mind[i]), max_dist=float(maxd[i]))
        F = W.add_frame(_tq(Tcw), K, np.zeros(0, sw.KEYPOINT_DTYPE))
        R9, t3, Ow3 = [np.ascontiguousarray(a, np.float32) for a in oracle.pose_matrices(Tcw)]
        W.L.sw_frame_set_pose_matrices(W.h, F, sw._p(R9), sw._p(t3), sw._p(Ow3))
        tf = np.zeros((n, 5), np.float32); ti = np.zeros((n, 2), np.int32)
        nin = W._chk(W.L.sw_is_in_frustum(W.h, F, sw._p(sw._i32(np.arange(n))), n, f32(0.5), batched, sw._p(tf), sw._p(ti)))
        assert nin == int(ref["in_view"].sum())
        assert np.array_equal(ti[:, 0], ref["in_view"]) and np.array_equal(tf[:, 0], ref["proj_x"]) and np.array_equal(tf[:, 1], ref["proj_y"])
        v = ref["in_view"] != 0
        assert np.array_equal(tf[v, 2], ref["depth"][v]) and np.array_equal(tf[v, 3], ref["view_cos"][v]) and np.array_equal(ti[v, 1], ref["level"][v])
        assert np.array_equal(tf[v, 4], ref["proj_xr"][v])


# ------------------------------------------------------------------------------------------------------------ keyframe pairs
def _consistent_pair(oracle, seed, **kw):
    """make_kf_pair_scene with ONE bad flag per 3-D point (the scene draws them per view; real map points carry one)."""
    sc = make_kf_pair_scene(oracle, seed, **kw)
    ptbad = sc["pts"]["bad"].astype(bool)
    for kf in sc["kf"]:
        has = kf["mp"] >= 0
        kf["bad"] = np.where(has, ptbad[np.where(has, kf["mp"] - 1000, 0)], False).astype(np.uint8)
    return sc


def _pair_world(capi, sc, observe=(1,)):
    """World with the scene's 3-D points as map points and its two views as keyframes; the views listed in `observe` also
    register their matches as observations (IsInKeyFrame / Observations() are then the real thing)."""
    W = sw.World(); W.add_map(0); _tables(W, sc["scale_factors"])
    pts = sc["pts"]
    for i in range(len(pts["pos"])):
        W.add_mappoint(0, int(pts["id"][i]), pts["pos"][i], normal=pts["normal"][i], min_dist=float(pts["min_dist"][i]), max_dist=float(pts["max_dist"][i]),
                       desc=pts["desc"][i], bad=bool(pts["bad"][i]))
    keep = []
    for v, kf in enumerate(sc["kf"]):
        k = W.add_keyframe(0, 10 + v, _tq(kf["Tcw"]), kf["K"], kf["kps"], kf["desc"], bounds=(0, 0, 640, 480), pose_inv_tq=_tq(capi.se3_inverse(kf["Tcw"])),
                           log_scale=kf["log_scale_factor"])
        n, a, b, c, hold = _fv_args(kf["fv"]); keep.append(hold)
        W.L.sw_kf_set_feature_vector(W.h, k, n, a, b, c)
        for j, m in enumerate(kf["mp"]):
            if m >= 0:
                if v in observe:
                    W.observe(k, int(m) - 1000, j)
                else:
                    W.L.sw_kf_set_match(W.h, k, j, int(m) - 1000); W.kf[k]["matches"][j] = int(m) - 1000
    return W


def _ids(idx):
    idx = np.asarray(idx)
    return np.where(idx >= 0, idx + 1000, -1).astype(np.int32)


@pytest.mark.parametrize("seed,dup,ratio,ori", [(0, 0.1, 0.7, True), (3, 0.3, 0.75, False)])
def test_search_by_bow(capi, oracle, seed, dup, ratio, ori):
    sc = _consistent_pair(oracle, seed, dup_frac=dup)
    a, b = sc["kf"]
    W = _pair_world(capi, sc, observe=())
    n_o, m_o = oracle.search_by_bow_kf_kf(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["mp"], b["bad"], b["fv"], ratio, ori)
    out = np.full(len(a["kps"]), -7, np.int32)
    n = W._chk(W.L.sw_search_by_bow_kf_kf(W.h, 0, 1, sw._p(out), f32(ratio), int(ori)))
    assert n == n_o > 100 and np.array_equal(_ids(out), m_o)
    # KF -> Frame (relocalisation / reference-keyframe tracking): vpMapPointMatches has one entry per FRAME keypoint
    n_o, m_o = oracle.search_by_bow_kf_frame(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["fv"], ratio, ori)
    F = W.add_frame(_tq(b["Tcw"]), b["K"], b["kps"], b["desc"], bounds=b["bounds"])
    nf, x, y, z, hold = _fv_args(b["fv"])
    W.L.sw_frame_set_feature_vector(W.h, F, nf, x, y, z)
    out = np.full(len(b["kps"]), -7, np.int32)
    n = W._chk(W.L.sw_search_by_bow_kf_frame(W.h, 0, F, sw._p(out), f32(ratio), int(ori)))
    assert n == n_o > 100 and np.array_equal(_ids(out), m_o)


@pytest.mark.parametrize("seed,coarse", [(0, False), (2, True)])
def test_search_for_triangulation(capi, oracle, seed, coarse):
    sc = _consistent_pair(oracle, seed, mapped_frac=0.4, dup_frac=0.2)
    a, b = sc["kf"]
    W = _pair_world(capi, sc, observe=())
    geo = oracle.triangulation_geometry(a["Tcw"], b["Tcw"], a["K"], b["K"])
    n_o, p_o = oracle.search_for_triangulation(a["kps"], a["desc"], a["mp"], a["fv"], b["kps"], b["desc"], b["mp"], b["fv"], geo[3], geo[2],
                                               b["scale_factors"], b["level_sigma2"], coarse, True)
    pairs = np.zeros((len(a["kps"]), 2), np.int32)
    n = W._chk(W.L.sw_search_for_triangulation(W.h, 0, 1, sw._p(pairs), len(pairs), int(coarse), 1))
    assert n == n_o > 100 and np.array_equal(pairs[:n], np.asarray(p_o).reshape(-1, 2)[:n])


@pytest.mark.parametrize("seed,far", [(0, False), (2, True)])
def test_create_new_map_points_geometry(capi, oracle, seed, far):
    """LocalMapping::CreateNewMapPoints for one neighbour, as the shims run it: ORBmatcher::SearchForTriangulation, then
    TriangulateMatches (host/LocalMapping_shim.h) on the pairs it returned.  Compared with the oracle given the same keyframes."""
    sc = _consistent_pair(oracle, seed, mapped_frac=0.4, dup_frac=0.2)
    a, b = sc["kf"]
    W = _pair_world(capi, sc, observe=())
    pairs = np.zeros((len(a["kps"]), 2), np.int32)
    n = W._chk(W.L.sw_search_for_triangulation(W.h, 0, 1, sw._p(pairs), len(pairs), 0, 1))
    assert n > 100
    pairs = np.ascontiguousarray(pairs[:n])
    X = np.zeros((n, 3), np.float32); st = np.full(n, -7, np.int32)
    T1 = np.zeros(12, np.float32); T2 = np.zeros(12, np.float32); O1 = np.zeros(3, np.float32); O2 = np.zeros(3, np.float32)
    args = lambda f, t: (W.h, 0, 1, sw._p(pairs), n, 0, int(f), f32(t), sw._p(X), sw._p(st), sw._p(T1), sw._p(T2), sw._p(O1), sw._p(O2))
    W._chk(W.L.sw_triangulate_matches(*args(False, 0.0)))
    assert (st == 0).sum() > 30
    th_far = float(np.median(np.linalg.norm(X[st == 0] - O1, axis=1)))      # mThFarPoints in the middle of the accepted points
    W._chk(W.L.sw_triangulate_matches(*args(far, th_far)))
    Xo, so = oracle.triangulate_matches(a["K"], b["K"], T1, T2, O1, O2, a["kps"], b["kps"], pairs, a["level_sigma2"], b["level_sigma2"],
                                        a["scale_factors"], b["scale_factors"], np.float32(1.5) * np.float32(a["scale_factors"][1]),
                                        far_points=far, th_far=th_far)
    assert np.array_equal(st, so) and np.array_equal(X.view(np.uint32), Xo.view(np.uint32))
    assert (so == 0).sum() > 10 and ((so == 8).sum() > 10) == far
    # the poses the mock keyframes handed over are the scene's (rotation matrix of the unit quaternion | translation)
    Rt = np.asarray(oracle.se3_matrix(a["Tcw"]), np.float32).reshape(3, 4) if hasattr(oracle, "se3_matrix") else None
    if Rt is not None:
        assert np.allclose(T1.reshape(3, 4), Rt, atol=1e-6)


@pytest.mark.parametrize("seed", [0, 2])
def test_fuse_replays_replace_and_add_observation(capi, oracle, seed):
    """ORBmatcher::Fuse(pKF, vpMapPoints, th) (LocalMapping::SearchInNeighbors): the search is one device call, the graph edits
    -- Replace of the point with fewer observations, AddObservation + AddMapPoint where the keypoint was free -- are replayed on
    the map in the reference's order; an earlier Replace can change what a later candidate finds."""
    sc = _consistent_pair(oracle, seed, dup_frac=0.25)
    kf, pts = sc["kf"][1], sc["pts"]
    W = _pair_world(capi, sc, observe=(1,))
    rng = np.random.default_rng(seed)
    nobs = rng.integers(1, 6, len(pts["pos"]))
    for i in range(len(nobs)):
        W.L.sw_mp_set_obs_count(W.h, i, int(nobs[i]))
    in_kf = np.isin(pts["id"], kf["mp"][kf["mp"] >= 0]).astype(np.uint8)
    valid = ((pts["bad"] == 0) & (in_kf == 0)).astype(np.uint8)
    p2 = dict(pts); p2["valid"] = valid
    bi_o, bd_o, _ = oracle.project_search(kf["kps"], kf["desc"], kf["bounds"], None, kf["Tcw"], oracle.se3_inverse(kf["Tcw"])[4:], kf["K"], p2, 3.0,
                                          kf["scale_factors"], kf["log_scale_factor"], kf["inv_level_sigma2"], 5.99)
    want = np.where((bi_o >= 0) & (bd_o <= 50), bi_o, -1)
    # replay on a mirror: matches of the keyframe, observations (only the keyframe matters), counts, bad flags
    match = [int(m) - 1000 if m >= 0 else -1 for m in kf["mp"]]
    obs = {i: {} for i in range(len(nobs))}
    for j, m in enumerate(match):
        if m >= 0:
            obs[m] = {1: j}
    cnt = [int(c) for c in nobs]
    bad = [bool(b) for b in pts["bad"]]
    fused = 0

    def replace(old, new):                                       # MapPoint::Replace(old -> new) restricted to what this map holds
        nonlocal cnt
        bad[old] = True
        o = obs[old]; obs[old] = {}
        for k, j in o.items():
            if k not in obs[new]:
                match[j] = new; obs[new][k] = j; cnt[new] += 1
            else:
                match[j] = -1
    for i in range(len(want)):
        if want[i] < 0 or bad[i] or 1 in obs[i]:
            continue
        j = int(want[i]); there = match[j]
        if there >= 0:
            if not bad[there]:
                if cnt[there] > cnt[i]:
                    replace(i, there)
                else:
                    replace(there, i)
        else:
            obs[i][1] = j; match[j] = i; cnt[i] += 1
        fused += 1
    n = W._chk(W.L.sw_fuse(W.h, 1, sw._p(sw._i32(np.arange(len(want)))), len(want), f32(3.0)))
    assert n == fused > 100
    assert list(W.kf_matches(1)) == match
    for i in range(len(want)):
        g = W.get_mp(i)
        assert g["bad"] == bad[i] and g["n_obs"] == cnt[i], i
        assert W.mp_observations(i).get(1, -1) == obs[i].get(1, -1), i


@pytest.mark.parametrize("seed", [1])
def test_sim3_searches(capi, oracle, seed):
    """Fuse(pKF, Scw, vpPoints, th, vpReplacePoint), SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming) and
    SearchBySim3 (LoopClosing / merge): Sim3 handed over as a Sophus::Sim3f."""
    from dvm_slam_amd import synth
    sc = _consistent_pair(oracle, seed, dup_frac=0.25, mapped_frac=0.8)
    a, kf = sc["kf"]; pts = sc["pts"]
    s = 1.7
    Scw = synth.sim3_from_sRt(s, kf["Rcw"].reshape(3, 3), kf["tcw"] * s)
    Sp = sw._p(np.ascontiguousarray(Scw, np.float32))
    allp = sw._i32(np.arange(len(pts["pos"])))
    # Fuse with a Sim3
    nf_o, mp_o, rep_o = oracle.fuse_sim3(kf["kps"], kf["desc"], kf["bounds"], kf["mp"], kf["bad"], Scw, kf["K"], pts, 4.0, kf["scale_factors"], kf["log_scale_factor"])
    W = _pair_world(capi, sc, observe=(1,))
    rep = np.full(len(allp), -7, np.int32)
    n = W._chk(W.L.sw_fuse_sim3(W.h, 1, Sp, sw._p(allp), len(allp), f32(4.0), sw._p(rep)))
    assert n == nf_o > 100 and np.array_equal(_ids(rep), rep_o) and np.array_equal(_ids(W.kf_matches(1)), mp_o)
    added = np.flatnonzero((mp_o != kf["mp"]) & (mp_o >= 0))
    assert len(added) > 20 and all(W.mp_observations(int(mp_o[j]) - 1000).get(1) == j for j in added)
    # SearchByProjection with a Sim3
    matched = np.where(np.random.default_rng(seed).random(len(kf["kps"])) < 0.3, kf["mp"], -1).astype(np.int32)
    nm_o, m_o = oracle.search_by_projection_sim3(kf["kps"], kf["desc"], kf["bounds"], matched, Scw, kf["K"], pts, 8, 1.0, kf["scale_factors"], kf["log_scale_factor"])
    W = _pair_world(capi, sc, observe=())
    m = np.where(matched >= 0, matched - 1000, -1).astype(np.int32)
    n = W._chk(W.L.sw_search_by_projection_sim3(W.h, 1, Sp, sw._p(allp), len(allp), sw._p(m), 8, f32(1.0)))
    assert n == nm_o > 50 and np.array_equal(_ids(m), m_o)
    # SearchBySim3: every keypoint's own map point is read from the keyframes, GetIndexInKeyFrame gives the KF2 index of a known match
    geo = oracle.triangulation_geometry(a["Tcw"], kf["Tcw"], a["K"], kf["K"])
    S12 = synth.sim3_from_sRt(1.03, geo[0].reshape(3, 3), geo[1])

    def per_kp(k):
        idx = np.where(k["pt_of_kp"] >= 0, k["pt_of_kp"], 0).astype(np.int64)
        return dict(pos=pts["pos"][idx], normal=pts["normal"][idx], min_dist=pts["min_dist"][idx], max_dist=pts["max_dist"][idx], desc=pts["desc"][idx])
    rng = np.random.default_rng(seed)
    m_in = np.full(len(a["kps"]), -1, np.int32); idx2 = np.full(len(a["kps"]), -1, np.int32)
    for i in rng.choice(len(a["kps"]), 40, replace=False):
        j = np.nonzero((kf["pt_of_kp"] == a["pt_of_kp"][i]) & (a["pt_of_kp"][i] >= 0) & (kf["mp"] >= 0))[0]
        if len(j) and a["mp"][i] >= 0:
            m_in[i] = a["mp"][i]; idx2[i] = j[0]
    n_o, m12_o = oracle.search_by_sim3(a, per_kp(a), kf, per_kp(kf), S12, 7.5, m_in, idx2)
    W = _pair_world(capi, sc, observe=(0, 1))                     # (both views observe: GetIndexInKeyFrame(pKF2) must answer)
    # a point matched at entry must be observed by KF2 at idx2: true by construction (same 3-D point, mapped in both views)
    m12 = np.where(m_in >= 0, m_in - 1000, -1).astype(np.int32)
    n = W._chk(W.L.sw_search_by_sim3(W.h, 0, 1, sw._p(m12), sw._p(np.ascontiguousarray(S12, np.float32)), f32(7.5)))
    assert n == n_o > 100 and np.array_equal(_ids(m12), m12_o)


# ------------------------------------------------------------------------------------------------------------------ extractor
def test_orb_extractor_call_operator(capi, oracle, frames):
    """(*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors, vLapping) through the shim class, on a cv::Mat with a row stride:
    keypoints, descriptors, return value, scale tables and one pyramid level against the oracle; an empty image returns -1."""
    img = frames[1]
    rows, cols = img.shape
    padded = np.zeros((rows, cols + 24), np.uint8); padded[:, :cols] = img
    orc = oracle.OrbOracle()
    n_o, k_o, d_o, mono_o = orc.extract(img)
    W = sw.World()
    cap = 1400
    kps = np.zeros(cap, sw.KEYPOINT_DTYPE); desc = np.zeros((cap, 32), np.uint8); n = np.zeros(1, np.int32)
    pyr = np.zeros(rows * cols, np.uint8); dims = np.zeros(2, np.int32); tables = np.zeros((4, 8), np.float32)
    mono = W._chk(W.L.sw_extract(W.h, sw._p(padded), rows, cols, cols + 24, 1000, f32(1.2), 8, 20, 7, 0, 1000, sw._p(kps), sw._p(desc), cap, sw._p(n), 3,
                                 sw._p(pyr), sw._p(dims), sw._p(tables)))
    assert (mono, int(n[0])) == (mono_o, n_o) and n_o > 900
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(kps[f][:n_o], k_o[f]), f
    assert np.array_equal(desc[:n_o], d_o)
    lvl = orc.level(3)
    assert tuple(dims) == lvl.shape and np.array_equal(pyr[:lvl.size].reshape(lvl.shape), lvl)
    t = orc.tables()
    assert np.array_equal(tables[0], t["scale"]) and np.array_equal(tables[3], t["inv_sigma2"])
    # lapping area {0, 100}: keypoints left of x = 100 fill the array from the back, the others from the front (ORBextractor.cc:931-950)
    n2, k2, d2, mono2 = orc.extract(img, lap=(0, 100))
    mono = W._chk(W.L.sw_extract(W.h, sw._p(padded), rows, cols, cols + 24, 1000, f32(1.2), 8, 20, 7, 0, 100, sw._p(kps), sw._p(desc), cap, sw._p(n), 0, None, None, None))
    assert (mono, int(n[0])) == (mono2, n2) and 0 < mono2 < n2 and np.array_equal(kps["x"][:n2], k2["x"]) and np.array_equal(desc[:n2], d2)
    assert W._chk(W.L.sw_extract(W.h, None, 0, 0, 0, 1000, f32(1.2), 8, 20, 7, 0, 1000, sw._p(kps), sw._p(desc), cap, sw._p(n), 0, None, None, None)) == -1


def test_orb_vocabulary_class(oracle, tmp_path):
    """ORB_SLAM3::ORBVocabulary as Frame::ComputeBoW / KeyFrame::ComputeBoW use it: loadFromTextFile, transform(vector<cv::Mat>, BowVector&,
    FeatureVector&, 4), score(), size() -- BoW vector (ids, normalised weights) and feature vector identical to the oracle's."""
    from dvm_slam_amd import synth
    from test_gpu_match import _write_dbow2_text
    voc = synth.vocabulary(k=8, L=4, seed=21)
    path = tmp_path / "voc.txt"
    _write_dbow2_text(voc, path, 8)
    rng = np.random.default_rng(4)
    desc = voc["desc"][rng.integers(1, voc["n_nodes"], 700)].copy()
    desc[rng.random(desc.shape) < 0.03] ^= 0x18
    W = sw.World()
    g = W.vocab_compute_bow(str(path), desc, 4)
    r = oracle.vocab_transform(voc, desc, 4)
    for key in ("bow_ids", "bow_vals", "fv_nodes", "fv_off", "fv_feat"):
        assert np.array_equal(g[key], r[key]), key
    assert g["size"] == int((voc["word_id"] >= 0).sum())
    assert g["self_score"] == oracle.bow_score(r["bow_ids"], r["bow_vals"], r["bow_ids"], r["bow_vals"]) and abs(g["self_score"] - 1.0) < 1e-12


@pytest.mark.parametrize("seed", [0, 1])
def test_keyframe_database_class(oracle, seed):
    """ORB_SLAM3::KeyFrameDatabase on KeyFrame* / Map* / Frame* (host/KeyFrameDatabase_shim.h): add / erase, DetectMergePossibility and
    CalculateMergeScore on a peer's BoW vector + uuid, DetectNBestCandidates of a stored keyframe, DetectRelocalizationCandidates of
    a frame -- a mixed sequence against the oracle's database.  Bad flags and covisibility are changed ON THE OBJECTS between the
    queries: the class reads them live, as the reference does.  Keyframes ERASED from the database stay in their neighbours' covisibility
    lists with the query state they last had -- the reference reads the object's members, inverted file or not (a 140-step sequence found
    the class skipping them)."""
    from kfdb_scene import fill, make_db_scene
    kfs = make_db_scene(seed + 20, kf_per_map=30)
    dbo = oracle.KeyFrameDatabase()
    fill(dbo, kfs)
    W = sw.World()
    for m in range(3):
        W.add_map(0)
    ident = np.array([0, 0, 0, 0, 0, 0, 1], np.float32)
    K = np.array([149.0, 149.0, 320.0, 240.0])
    nokp = np.zeros(0, sw.KEYPOINT_DTYPE)
    for j, k in enumerate(kfs):
        i = W.add_keyframe(k["map_id"], k["mn_id"], ident, K, nokp)
        assert i == j
        W.kf_set_bow(i, k["ids"], k["vals"]); W.kf_set_uuid(i, k["uuid"])
    for j, k in enumerate(kfs):
        W.set_covisible(j, list(k["neigh"]), [100 - t for t in range(len(k["neigh"]))])
        W.kf_set_connected(j, list(k["connected"]))
    W.kfdb_create()
    for j in range(len(kfs)):
        W.kfdb_add(j)
    rng = np.random.default_rng(seed)
    alive = set(range(len(kfs)))
    hits = moved = 0
    qmap = {}
    for step in range(140):
        op = rng.random()
        j = int(rng.choice(sorted(alive)))
        q = kfs[j]
        if op < 0.3:
            m = int(rng.integers(0, 3))
            ro = dbo.detect_merge_possibility(q["ids"], q["vals"], q["uuid"], m)
            rg = W.kfdb_detect_merge_possibility(q["ids"], q["vals"], q["uuid"], m)
            assert (ro[0], ro[1]) == rg, (step, ro, rg)
            hits += ro[1] >= 0
        elif op < 0.4:
            m = int(rng.integers(0, 3))
            assert dbo.merge_score(q["ids"], q["vals"], q["uuid"], m, 0.0) == W.kfdb_merge_score(q["ids"], q["vals"], q["uuid"], m, 0.0)
        elif op < 0.65:
            lo, mo = dbo.detect_n_best(j, 3)
            lg, mg = W.kfdb_detect_n_best(j, 3)
            assert np.array_equal(lo, lg) and np.array_equal(mo, mg), (step, lo, lg, mo, mg)
            hits += len(lo) + len(mo) > 0
        elif op < 0.85:
            m = int(rng.integers(0, 3)); fid = int(rng.integers(1, 40))
            co = dbo.detect_reloc(q["ids"], q["vals"], fid, m)
            cg = W.kfdb_detect_reloc(q["ids"], q["vals"], fid, m)
            assert np.array_equal(co, cg), (step, co, cg)
            hits += len(co) > 0
        elif op < 0.92 and len(alive) > 20:
            dbo.erase(j); W.kfdb_erase(j); alive.discard(j)
        elif op < 0.96:
            b = bool(rng.integers(0, 2))
            dbo.set_bad(j, b); W.kf_set_bad(j, b)
        else:
            # LoopClosing::MergeLocal moves keyframes into the merged map with KeyFrame::UpdateMap (LoopClosing.cc:1558,1767) AFTER they
            # were added: every later query must see them in their new map (loop vs merge candidates, the relocalisation filter, the
            # map whose query state CalculateMergeScore resets)
            m_new = int(rng.integers(0, 3))
            dbo.set_map(j, m_new); W.kf_update_map(j, m_new)
            moved += 1
        # the per-keyframe query state (mnPlaceRecognitionQuery / Words / Score) after every step; query ids through a bijection (the oracle is
        # handed the uuid integers, the class hashes the uuid bytes as the reference does)
        for jj in sorted(alive):
            so, sg = dbo.state(jj), W.kfdb_get_state(jj)
            assert so[1:] == sg[1:] and qmap.setdefault(sg[0], so[0]) == so[0], (step, jj, so, sg)
    assert hits > 15 and moved > 0


def test_mappoint_compute_distinctive_descriptors(oracle):
    """MapPoint::ComputeDistinctiveDescriptors, the member and the batched form: rows gathered from the observing keyframes in the
    point's own observation order (bad keyframes skipped), the least-median row cloned into mDescriptor; bad points, points
    without (good) observations keep what they had."""
    rng = np.random.default_rng(12)
    W = sw.World(); W.add_map(0)
    K = np.array([149.0, 149.0, 320.0, 240.0]); ident = np.array([0, 0, 0, 0, 0, 0, 1], np.float32)
    nkf, nkp = 14, 40
    descs = []
    base = rng.integers(0, 256, (nkp, 32), dtype=np.uint8)
    for k in range(nkf):
        d = base.copy()
        flip = rng.random(d.shape) < 0.06
        d[flip] ^= rng.integers(1, 256, int(flip.sum()), dtype=np.uint8)
        descs.append(d)
        kps = np.zeros(nkp, sw.KEYPOINT_DTYPE)
        W.add_keyframe(0, k + 1, ident, K, kps, desc=d, bad=(k == 5))
    mps = []
    for i in range(nkp):
        mps.append(W.add_mappoint(0, i, np.zeros(3, np.float32), bad=(i == 3)))
        if i == 7:
            continue                                    # no observation at all
        seen = rng.choice(nkf, size=int(rng.integers(1, 13)), replace=False) if i != 9 else np.array([5])   # point 9: only the bad keyframe
        for k in seen:
            W.observe(int(k), mps[-1], i)
    before = np.stack([np.zeros(32, np.uint8)] * nkp)
    want = before.copy()
    for i in range(nkp):
        if i in (3, 7):
            continue
        rows = [descs[k][j] for k, j in W.mp_observations(mps[i]).items() if k != 5]      # the point's own std::map order
        if not rows:
            continue
        r = np.stack(rows)
        bi, _ = oracle.distinctive_descriptors(r.reshape(-1, 32), np.array([0, len(r)], np.int32))
        want[i] = r[int(bi[0])]
    got = W.compute_distinctive(mps, batched=False)
    assert np.array_equal(got, want)
    W2_got = W.compute_distinctive(mps, batched=True)
    assert np.array_equal(W2_got, want)


def test_frame_hand_off_stays_on_the_device(capi, oracle):
    """The reference's flow Frame.cc:411 (ExtractORB) -> Frame.cc:481 (grid) -> ORBmatcher.cc:1553 (SearchByProjection(CurrentFrame, LastFrame))
    through the shims with the optional `Frame::mDvmDevice = mpORBextractorLeft->LastDeviceResult()` line: the current frame's feature grid is
    built from the keypoints + descriptors the extractor left in HBM -- no second upload -- and the search returns exactly what it returns
    from the host copies.  A reference that has gone stale (the extractor has produced another frame since) is recognised and ignored."""
    from dvm_slam_amd import synth
    frames = synth.frame_stream(3)
    H, Wd = frames.shape[1:]
    K = (149.0, 149.0, 320.0, 240.0)
    ident = np.array([0, 0, 0, 0, 0, 0, 1], np.float32)
    scale = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    results = {}
    for mode in ("host", "device", "stale"):
        W = sw.World(); W.add_map(0); _tables(W, scale)
        nokp = np.zeros(0, sw.KEYPOINT_DTYPE)
        last = W.add_frame(ident, K, nokp, None, bounds=(0, Wd, 0, H))
        cur = W.add_frame(ident, K, nokp, None, bounds=(0, Wd, 0, H))
        nl = W.frame_extract(last, frames[0], False)
        nc = W.frame_extract(cur, frames[1], mode != "host")
        if mode == "stale":
            extra = W.add_frame(ident, K, nokp, None, bounds=(0, Wd, 0, H))
            W.frame_extract(extra, frames[2], True)                       # the extractor moves on: cur's reference no longer names its result
        assert nl > 900 and nc > 900
        kl, dl = W.frame_keypoints(last)
        # the last frame's map points: its keypoints back-projected to depth 4 (the identity pose), carrying its descriptors
        mp_l = np.full(nl, -1, np.int32)
        for j in range(0, nl, 2):
            z = 4.0
            X = [(kl["x"][j] - K[2]) / K[0] * z, (kl["y"][j] - K[3]) / K[1] * z, z]
            mp_l[j] = W.add_mappoint(0, j, X, desc=dl[j])
            W.L.sw_mp_set_obs_count(W.h, int(mp_l[j]), 2)
        W.frame_set_matches(last, mp_l)
        n = W._chk(W.L.sw_search_by_projection_last(W.h, cur, last, f32(15.0), f32(0.9), 1))
        results[mode] = (n, W.get_frame(cur)["mp"].copy(), bool(W.L.sw_last_grid_from_device(W.h)))
    assert results["host"][2] is False and results["device"][2] is True and results["stale"][2] is False
    assert results["host"][0] == results["device"][0] == results["stale"][0] > 200
    assert np.array_equal(results["host"][1], results["device"][1]) and np.array_equal(results["host"][1], results["stale"][1])
