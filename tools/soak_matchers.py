#!/usr/bin/env python3
"""tools/soak_matchers.py [seconds] [seed] -- randomized GPU-vs-oracle soak of the whole ORBmatcher functions (SURVEY.md 8a
M4-M7) and the KeyFrameDatabase queries: random scene sizes, duplicate fractions, thresholds, ratios, poses.  Every output
array must be identical to the sequential oracle.  Not part of the pytest suite; exit code 1 on any mismatch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dvm_slam_amd import capi, synth     # noqa: E402
from oracle import pyoracle as po        # noqa: E402
from matcher_scene import make_init_scene, make_kf_pair_scene, make_scene   # noqa: E402


def one_case(rng):
    bad = []
    seed = int(rng.integers(1 << 30))
    sc = make_kf_pair_scene(po, seed, n_pts=int(rng.integers(50, 1500)), n_clutter=int(rng.integers(0, 500)),
                            n_nodes=int(rng.integers(5, 400)), mapped_frac=float(rng.uniform(0.1, 0.9)),
                            flip_bits=int(rng.integers(0, 40)), dup_frac=float(rng.uniform(0, 0.4)))
    a, b = sc["kf"]
    pts = sc["pts"]
    ratio = float(rng.choice([0.6, 0.7, 0.75, 0.8, 0.9]))
    ori = bool(rng.integers(0, 2))
    va = capi.keyframe_view(dict(a, mp=a["mp"].copy())); vb = capi.keyframe_view(dict(b, mp=b["mp"].copy()))
    n_o, m_o = po.search_by_bow_kf_kf(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["mp"], b["bad"], b["fv"], ratio, ori)
    n_g, m_g, _ = capi.search_by_bow_kf_kf(va, vb, ratio, ori)
    if n_o != n_g or not np.array_equal(m_o, m_g): bad.append("bow_kf_kf")
    n_o, m_o = po.search_by_bow_kf_frame(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["fv"], ratio, ori)
    F = capi.frame_view(b["kps"], b["desc"], b["bounds"], b["scale_factors"])
    n_g, m_g, _ = capi.search_by_bow_kf_frame(va, F, b["fv"], ratio, ori)
    if n_o != n_g or not np.array_equal(m_o, m_g): bad.append("bow_kf_frame")
    geo = po.triangulation_geometry(a["Tcw"], b["Tcw"], a["K"], b["K"])
    coarse = bool(rng.integers(0, 2))
    n_o, p_o = po.search_for_triangulation(a["kps"], a["desc"], a["mp"], a["fv"], b["kps"], b["desc"], b["mp"], b["fv"], geo[3], geo[2],
                                           b["scale_factors"], b["level_sigma2"], coarse, ori)
    n_g, p_g = capi.search_for_triangulation(va, vb, coarse, ori)
    if n_o != n_g or not np.array_equal(p_o, p_g): bad.append("triangulation")
    th = float(rng.choice([2.0, 3.0, 4.0, 8.0, 15.0]))
    P = capi.map_points_view(pts)
    in_kf = np.isin(pts["id"], b["mp"][b["mp"] >= 0]).astype(np.uint8)
    p2 = dict(pts); p2["valid"] = ((pts["bad"] == 0) & (in_kf == 0)).astype(np.uint8)
    bi_o, bd_o, _ = po.project_search(b["kps"], b["desc"], b["bounds"], None, b["Tcw"], po.se3_inverse(b["Tcw"])[4:], b["K"], p2, th, b["scale_factors"],
                                      b["log_scale_factor"], b["inv_level_sigma2"], 5.99)
    n_g, bi_g = capi.fuse(vb, P, in_kf, th)
    if not np.array_equal(bi_g, np.where((bi_o >= 0) & (bd_o <= 50), bi_o, -1)): bad.append("fuse")
    s = float(rng.uniform(0.5, 2.0))
    Scw = synth.sim3_from_sRt(s, b["Rcw"].reshape(3, 3), b["tcw"] * np.float32(s))   # a Sophus::Sim3f; both sides decompose it
    nf_o, mp_o, rep_o = po.fuse_sim3(b["kps"], b["desc"], b["bounds"], b["mp"], b["bad"], Scw, b["K"], pts, th, b["scale_factors"],
                                     b["log_scale_factor"])
    d = dict(b, mp=b["mp"].copy())
    nf_g, rep_g = capi.fuse_sim3(capi.keyframe_view(d), Scw, P, th)
    if nf_o != nf_g or not np.array_equal(rep_o, rep_g) or not np.array_equal(mp_o, d["mp"]): bad.append("fuse_sim3")
    matched = np.where(rng.random(len(b["kps"])) < rng.uniform(0, 0.6), b["mp"], -1).astype(np.int32)
    rh = float(rng.choice([0.8, 1.0, 1.5]))
    nm_o, mm_o = po.search_by_projection_sim3(b["kps"], b["desc"], b["bounds"], matched, Scw, b["K"], pts, int(th), rh,
                                              b["scale_factors"], b["log_scale_factor"])
    nm_g, mm_g, _ = capi.search_by_projection_sim3(vb, Scw, P, matched, int(th), rh)
    if nm_o != nm_g or not np.array_equal(mm_o, mm_g): bad.append("search_by_projection_sim3")
    idx = lambda kf: np.where(kf["pt_of_kp"] >= 0, kf["pt_of_kp"], 0).astype(np.int64)
    pk = lambda kf: dict(pos=pts["pos"][idx(kf)], normal=pts["normal"][idx(kf)], min_dist=pts["min_dist"][idx(kf)], max_dist=pts["max_dist"][idx(kf)],
                         desc=pts["desc"][idx(kf)])
    s12 = float(rng.uniform(0.8, 1.25))
    m_in = np.full(len(a["kps"]), -1, np.int32)
    S12 = synth.sim3_from_sRt(s12, geo[0].reshape(3, 3), geo[1])
    ns_o, ms_o = po.search_by_sim3(a, pk(a), b, pk(b), S12, th, m_in, None)
    ns_g, ms_g = capi.search_by_sim3(va, vb, capi.map_points_view(pk(a)), capi.map_points_view(pk(b)), m_in, None, S12, th)
    if ns_o != ns_g or not np.array_equal(ms_o, ms_g): bad.append("search_by_sim3")
    si = make_init_scene(po, seed, n=int(rng.integers(50, 3000)), shift=float(rng.uniform(0, 30)), flip_bits=int(rng.integers(0, 40)))
    win = int(rng.choice([10, 30, 100]))
    n_o, m_o, pm_o = po.search_for_initialization(si["k1"], si["d1"], si["k2"], si["d2"], si["bounds"], si["prev_matched"], win, 0.9, ori)
    F1 = capi.frame_view(si["k1"], si["d1"], si["bounds"], si["scale_factors"]); F2 = capi.frame_view(si["k2"], si["d2"], si["bounds"], si["scale_factors"])
    n_g, m_g, pm_g = capi.search_for_initialization(F1, F2, si["prev_matched"], win, 0.9, ori)
    if n_o != n_g or not np.array_equal(m_o, m_g) or not np.array_equal(pm_o, pm_g): bad.append("init")
    # SearchByProjection(CurrentFrame, LastFrame): from sparse to crowded frames (the device's ranked four candidates per query, the
    # host's claim replay and -- when all four are taken -- its fallback search)
    n_cur = int(rng.choice([120, 200, 400, 800, 1100, 2000]))
    sf = make_scene(po, seed, n_last=int(rng.integers(100, 1500)), n_cur=n_cur, dup_frac=float(rng.uniform(0, 0.4)),
                    zero_obs_frac=float(rng.uniform(0, 0.3)), flip_bits=int(rng.integers(0, 40)))
    thf = float(rng.choice([7.0, 15.0, 30.0, 60.0]))
    n_o, mp_o = po.search_by_projection_frames(th=thf, check_ori=ori, **sf)
    n_g, mp_g, _ = capi.search_by_projection_frames(th=thf, check_ori=ori, **sf)
    if n_o != n_g or not np.array_equal(mp_o, mp_g): bad.append("search_by_projection_frames")
    return seed, bad


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time()
    cases = nbad = 0
    while time.time() - t0 < budget:
        seed, bad = one_case(rng)
        cases += 1
        if bad:
            nbad += 1
            print("MISMATCH seed", seed, bad, flush=True)
    print(f"soak_matchers: {cases} scenes x 10 functions, {nbad} scenes with mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
