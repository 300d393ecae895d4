"""GPU parity of the remaining whole ORBmatcher functions: the host C++ mirrors of
dvm_slam_amd/host/orb_matcher.cpp (device Hamming table / candidate-list search / projection + window search /
triangulation search + the reference's sequential bookkeeping on the host) against the sequential oracle restatements.
Exact: match counts and every output entry."""
import numpy as np
import pytest

from dvm_slam_amd import synth
from matcher_scene import make_init_scene, make_kf_pair_scene

pytestmark = pytest.mark.gpu


def _kfviews(capi, sc, copy_mp=True):
    out = []
    for kf in sc["kf"]:
        d = dict(kf)
        if copy_mp:
            d["mp"] = kf["mp"].copy()
        out.append((capi.keyframe_view(d), d))
    return out


@pytest.mark.parametrize("seed,window,ori", [(0, 100, True), (1, 100, False), (2, 30, True), (3, 10, True)])
def test_search_for_initialization(capi, oracle, seed, window, ori):
    sc = make_init_scene(oracle, seed)
    n_o, m_o, pm_o = oracle.search_for_initialization(sc["k1"], sc["d1"], sc["k2"], sc["d2"], sc["bounds"], sc["prev_matched"], window, 0.9, ori)
    F1 = capi.frame_view(sc["k1"], sc["d1"], sc["bounds"], sc["scale_factors"])
    F2 = capi.frame_view(sc["k2"], sc["d2"], sc["bounds"], sc["scale_factors"])
    n_g, m_g, pm_g = capi.search_for_initialization(F1, F2, sc["prev_matched"], window, 0.9, ori)
    assert n_g == n_o and np.array_equal(m_g, m_o) and np.array_equal(pm_g, pm_o)
    if window >= 30:
        assert n_o > 200


def test_search_for_initialization_degenerate(capi, oracle):
    sc = make_init_scene(oracle, 4)
    k1 = sc["k1"].copy(); k1["octave"] = 3      # no level-0 keypoint in F1
    F1 = capi.frame_view(k1, sc["d1"], sc["bounds"], sc["scale_factors"])
    F2 = capi.frame_view(sc["k2"], sc["d2"], sc["bounds"], sc["scale_factors"])
    n_g, m_g, pm_g = capi.search_for_initialization(F1, F2, sc["prev_matched"])
    assert n_g == 0 and np.all(m_g == -1) and np.array_equal(pm_g, sc["prev_matched"])


@pytest.mark.parametrize("seed,dup,ratio,ori", [(0, 0.1, 0.7, True), (1, 0.3, 0.9, True), (2, 0.0, 0.6, False), (3, 0.3, 0.75, False)])
def test_search_by_bow(capi, oracle, seed, dup, ratio, ori):
    sc = make_kf_pair_scene(oracle, seed, dup_frac=dup)
    a, b = sc["kf"]
    (va, _), (vb, _) = _kfviews(capi, sc)
    # KF-KF
    n_o, m_o = oracle.search_by_bow_kf_kf(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["mp"], b["bad"], b["fv"], ratio, ori)
    n_g, m_g, req = capi.search_by_bow_kf_kf(va, vb, ratio, ori)
    assert n_g == n_o and np.array_equal(m_g, m_o) and n_o > 100
    # KF-Frame
    n_o, m_o = oracle.search_by_bow_kf_frame(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["fv"], ratio, ori)
    F = capi.frame_view(b["kps"], b["desc"], b["bounds"], b["scale_factors"])
    n_g, m_g, req2 = capi.search_by_bow_kf_frame(va, F, b["fv"], ratio, ori)
    assert n_g == n_o and np.array_equal(m_g, m_o) and n_o > 100
    if dup >= 0.3:
        assert req > 0 and req2 > 0, "scene must exercise the claimed-candidate re-evaluation"


def test_search_by_bow_degenerate(capi, oracle):
    sc = make_kf_pair_scene(oracle, 9)
    a, b = sc["kf"]
    a2 = dict(a); a2["mp"] = np.full_like(a["mp"], -1)     # no map points in KF1 -> nothing to match
    va = capi.keyframe_view(a2); vb = capi.keyframe_view(dict(b))
    n_g, m_g, _ = capi.search_by_bow_kf_kf(va, vb)
    assert n_g == 0 and np.all(m_g == -1)
    # disjoint vocabulary nodes
    b2 = dict(b); fv = dict(b["fv"]); fv["fv_nodes"] = (b["fv"]["fv_nodes"] + 1).astype(np.int32); b2["fv"] = fv
    va = capi.keyframe_view(dict(a)); vb = capi.keyframe_view(b2)
    n_g, m_g, _ = capi.search_by_bow_kf_kf(va, vb)
    n_o, m_o = oracle.search_by_bow_kf_kf(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["mp"], b["bad"], fv)
    assert n_g == n_o == 0 and np.array_equal(m_g, m_o)


@pytest.mark.parametrize("seed,coarse,ori", [(0, False, True), (1, False, False), (2, True, True), (3, False, True)])
def test_search_for_triangulation(capi, oracle, seed, coarse, ori):
    sc = make_kf_pair_scene(oracle, seed, mapped_frac=0.4, dup_frac=0.2)
    a, b = sc["kf"]
    (va, _), (vb, _) = _kfviews(capi, sc)
    geo_o = oracle.triangulation_geometry(a["Tcw"], b["Tcw"], a["K"], b["K"])
    geo_g = capi.triangulation_geometry(va, vb)
    for x, y in zip(geo_o, geo_g):
        assert np.array_equal(x, y)
    n_o, p_o = oracle.search_for_triangulation(a["kps"], a["desc"], a["mp"], a["fv"], b["kps"], b["desc"], b["mp"], b["fv"], geo_o[3], geo_o[2],
                                               b["scale_factors"], b["level_sigma2"], coarse, ori)
    n_g, p_g = capi.search_for_triangulation(va, vb, coarse, ori)
    assert n_g == n_o and np.array_equal(p_g, p_o) and n_o > 100


def test_search_for_triangulation_epipole_inside(capi, oracle):
    """Camera 2 moved along the optical axis: the epipole lies inside the image and the exclusion disc removes candidates."""
    sc = make_kf_pair_scene(oracle, 11, mapped_frac=0.3)
    a, b = sc["kf"]
    b = dict(b); b["Tcw"] = synth.se3_from_Rt(np.eye(3), [0.02, -0.01, -1.5])
    va = capi.keyframe_view(dict(a)); vb = capi.keyframe_view(b)
    geo = oracle.triangulation_geometry(a["Tcw"], b["Tcw"], a["K"], b["K"])
    assert 0 < geo[2][0] < 640 and 0 < geo[2][1] < 480
    for coarse in (True, False):
        n_o, p_o = oracle.search_for_triangulation(a["kps"], a["desc"], a["mp"], a["fv"], b["kps"], b["desc"], b["mp"], b["fv"], geo[3], geo[2],
                                                   b["scale_factors"], b["level_sigma2"], coarse, True)
        n_g, p_g = capi.search_for_triangulation(va, vb, coarse, True)
        assert n_g == n_o and np.array_equal(p_g, p_o)


@pytest.mark.parametrize("seed,th,gate", [(0, 3.0, True), (1, 4.0, False), (2, 8.0, True), (3, 2.0, False)])
def test_project_search_device(capi, oracle, seed, th, gate):
    """dvm_project_search against the oracle: matches AND the projection / level / radius it derived."""
    sc = make_kf_pair_scene(oracle, seed)
    kf, pts = sc["kf"][1], sc["pts"]
    rng = np.random.default_rng(seed)
    skip = (rng.random(len(kf["kps"])) < 0.2).astype(np.uint8)
    valid = (rng.random(len(pts["pos"])) < 0.9).astype(np.uint8)
    p2 = dict(pts); p2["valid"] = valid
    gi = kf["inv_level_sigma2"] if gate else None
    bi_o, bd_o, pr_o = oracle.project_search(kf["kps"], kf["desc"], kf["bounds"], skip, kf["Tcw"], oracle.se3_inverse(kf["Tcw"])[4:], kf["K"], p2, th,
                                             kf["scale_factors"], kf["log_scale_factor"], gi, 5.99)
    g = capi.FrameGrid(2048)
    g.build(kf["kps"], kf["desc"], tuple(float(x) for x in kf["bounds"]))
    cam = dict(Tcw=kf["Tcw"], Ow=capi.se3_inverse(kf["Tcw"])[4:], K=kf["K"], bounds=kf["bounds"], log_scale_factor=kf["log_scale_factor"])
    m, pr = capi.project_search(g, cam, pts, th, kf["scale_factors"], skip=skip, gate_inv_sigma2=gi, gate=5.99, valid=valid)
    g.close()
    assert np.array_equal(pr["level"], pr_o[:, 3].astype(np.int32))
    ok = pr["level"] >= 0
    assert ok.sum() > 400
    assert np.array_equal(pr["u"][ok], pr_o[ok, 0]) and np.array_equal(pr["v"][ok], pr_o[ok, 1]) and np.array_equal(pr["radius"][ok], pr_o[ok, 2])
    assert np.array_equal(m["best_idx"], bi_o) and np.array_equal(m["best_dist"], bd_o)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fuse_and_sim3_searches(capi, oracle, seed):
    sc = make_kf_pair_scene(oracle, seed, dup_frac=0.25)
    kf, pts = sc["kf"][1], sc["pts"]
    P = capi.map_points_view(pts)
    # Fuse(KF, vpMapPoints, th): search part == oracle project_search with the chi2 gate
    in_kf = np.isin(pts["id"], kf["mp"][kf["mp"] >= 0]).astype(np.uint8)
    valid = ((pts["bad"] == 0) & (in_kf == 0)).astype(np.uint8)
    p2 = dict(pts); p2["valid"] = valid
    bi_o, bd_o, _ = oracle.project_search(kf["kps"], kf["desc"], kf["bounds"], None, kf["Tcw"], oracle.se3_inverse(kf["Tcw"])[4:], kf["K"], p2, 3.0,
                                          kf["scale_factors"], kf["log_scale_factor"], kf["inv_level_sigma2"], 5.99)
    want = np.where((bi_o >= 0) & (bd_o <= 50), bi_o, -1)
    d = dict(kf); d["mp"] = kf["mp"].copy()
    n_g, bi_g = capi.fuse(capi.keyframe_view(d), P, in_kf, 3.0)
    assert np.array_equal(bi_g, want) and n_g == int((want >= 0).sum()) > 100
    # Sim3 variants with a genuine similarity: Scw = (s, R, s * tcw) as a Sophus::Sim3f; both sides decompose it themselves
    s = 1.7
    Scw = synth.sim3_from_sRt(s, kf["Rcw"].reshape(3, 3), kf["tcw"] * s)
    nf_o, mp_o, rep_o = oracle.fuse_sim3(kf["kps"], kf["desc"], kf["bounds"], kf["mp"], kf["bad"], Scw, kf["K"], pts, 4.0,
                                         kf["scale_factors"], kf["log_scale_factor"])
    d = dict(kf); d["mp"] = kf["mp"].copy()
    nf_g, rep_g = capi.fuse_sim3(capi.keyframe_view(d), Scw, P, 4.0)
    assert nf_g == nf_o > 100 and np.array_equal(rep_g, rep_o) and np.array_equal(d["mp"], mp_o)
    matched = np.where(np.random.default_rng(seed).random(len(kf["kps"])) < 0.3, kf["mp"], -1).astype(np.int32)
    for th, ratio in ((8, 1.0), (3, 0.8)):
        nm_o, m_o = oracle.search_by_projection_sim3(kf["kps"], kf["desc"], kf["bounds"], matched, Scw, kf["K"], pts, th, ratio,
                                                     kf["scale_factors"], kf["log_scale_factor"])
        d = dict(kf); d["mp"] = kf["mp"].copy()
        nm_g, m_g, req = capi.search_by_projection_sim3(capi.keyframe_view(d), Scw, P, matched, th, ratio)
        assert nm_g == nm_o and np.array_equal(m_g, m_o)
        if th == 8:
            assert nm_o > 50 and req > 0, "scene must exercise the claimed-keypoint re-query"


def _per_keypoint_points(kf, pts):
    """Map point data per keypoint (the point a keypoint observes; arbitrary where mvpMapPoints < 0)."""
    idx = np.where(kf["pt_of_kp"] >= 0, kf["pt_of_kp"], 0).astype(np.int64)
    return dict(pos=pts["pos"][idx], normal=pts["normal"][idx], min_dist=pts["min_dist"][idx], max_dist=pts["max_dist"][idx], desc=pts["desc"][idx])


@pytest.mark.parametrize("seed,s12,th", [(0, 1.0, 7.5), (1, 1.03, 7.5), (2, 0.97, 3.0)])
def test_search_by_sim3(capi, oracle, seed, s12, th):
    sc = make_kf_pair_scene(oracle, seed, mapped_frac=0.8)
    a, b = sc["kf"]
    geo = oracle.triangulation_geometry(a["Tcw"], b["Tcw"], a["K"], b["K"])
    S12 = synth.sim3_from_sRt(s12, geo[0].reshape(3, 3), geo[1])
    p1, p2 = _per_keypoint_points(a, sc["pts"]), _per_keypoint_points(b, sc["pts"])
    rng = np.random.default_rng(seed)
    # a few matches known at entry (their KF2 index marks vbAlreadyMatched2)
    m_in = np.full(len(a["kps"]), -1, np.int32); idx2 = np.full(len(a["kps"]), -1, np.int32)
    for i in rng.choice(len(a["kps"]), 40, replace=False):
        j = np.nonzero((b["pt_of_kp"] == a["pt_of_kp"][i]) & (a["pt_of_kp"][i] >= 0))[0]
        if len(j) and a["mp"][i] >= 0:
            m_in[i] = a["mp"][i]; idx2[i] = j[0]
    n_o, m_o = oracle.search_by_sim3(a, p1, b, p2, S12, th, m_in, idx2)
    va = capi.keyframe_view(dict(a, mp=a["mp"].copy())); vb = capi.keyframe_view(dict(b, mp=b["mp"].copy()))
    n_g, m_g = capi.search_by_sim3(va, vb, capi.map_points_view(p1), capi.map_points_view(p2), m_in, idx2, S12, th)
    assert n_g == n_o and np.array_equal(m_g, m_o)
    if th > 5:
        assert n_o > 150


def test_search_by_projection_sim3_records_keyframes(capi, oracle):
    sc = make_kf_pair_scene(oracle, 4, dup_frac=0.25)
    kf, pts = sc["kf"][1], sc["pts"]
    matched = np.full(len(kf["kps"]), -1, np.int32)
    point_kf = (np.arange(len(pts["pos"])) % 17 + 500).astype(np.int32)
    Scw = synth.sim3_from_sRt(1.0, kf["Rcw"].reshape(3, 3), kf["tcw"])
    nm_o, m_o = oracle.search_by_projection_sim3(kf["kps"], kf["desc"], kf["bounds"], matched, Scw, kf["K"], pts, 8, 1.0,
                                                 kf["scale_factors"], kf["log_scale_factor"])
    v = capi.keyframe_view(dict(kf, mp=kf["mp"].copy()))
    nm_g, m_g, _, mk = capi.search_by_projection_sim3(v, Scw, capi.map_points_view(pts), matched, 8, 1.0, point_kf=point_kf,
                                                      matched_kf=np.full(len(kf["kps"]), -1, np.int32))
    assert nm_g == nm_o and np.array_equal(m_g, m_o)
    hit = m_g >= 0
    assert np.array_equal(mk[hit], point_kf[m_g[hit] - 1000]) and np.all(mk[~hit] == -1)


@pytest.mark.parametrize("seed,th,orb_dist,ori", [(0, 10.0, 100, True), (1, 3.0, 64, True), (2, 10.0, 100, False)])
def test_search_by_projection_relocalisation(capi, oracle, seed, th, orb_dist, ori):
    """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist): keyframe a's map points into frame b."""
    sc = make_kf_pair_scene(oracle, seed, mapped_frac=0.8, dup_frac=0.25)
    a, b = sc["kf"]
    pa = _per_keypoint_points(a, sc["pts"])
    rng = np.random.default_rng(seed)
    cur_mp = np.where(rng.random(len(b["kps"])) < 0.2, b["mp"], -1).astype(np.int32)      # some keypoints already matched
    already = np.unique(cur_mp[cur_mp >= 0])
    n_o, m_o = oracle.search_by_projection_reloc(b["kps"], b["desc"], cur_mp, b["bounds"], b["Tcw"], b["K"], a, pa, already, th,
                                                 orb_dist, b["scale_factors"], b["log_scale_factor"], ori)
    m_g = cur_mp.copy()
    F = capi.frame_view(b["kps"], b["desc"], b["bounds"], b["scale_factors"], mp=m_g, K=b["K"], Tcw=b["Tcw"])
    n_g, req = capi.search_by_projection_reloc(F, capi.keyframe_view(dict(a, mp=a["mp"].copy())), capi.map_points_view(pa), already, th, orb_dist, ori)
    assert n_g == n_o and np.array_equal(m_g, m_o)
    if th >= 10:
        assert n_o > 150 and req > 0
