#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (run in the build container).

The reference holds no golden vector for this path and cannot be built or imported here (C++ needing
OpenCV/Eigen/Boost/ROS 2), so these fixtures are produced by the oracle -- which is itself pinned
against independent numpy restatements (tests/test_oracle_*.py).  They freeze today's behaviour:
both the oracle and the HIP path must keep reproducing them bit for bit (BA: within 1e-9).
Fixtures are DATA only (inputs + expected outputs).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dvm_slam_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def orb_case(name, img, params):
    o = po.OrbOracle(*params)
    n, k, d, mono = o.extract(img)
    lv = [o.level_dims(l) for l in range(params[2])]
    cand_counts = [len(o.candidates(l)[0]) for l in range(params[2])]
    np.savez_compressed(os.path.join(HERE, name), image=img, params=np.array(params, np.float64), n=n, mono=mono,
                        kp_x=k["x"], kp_y=k["y"], kp_size=k["size"], kp_angle=k["angle"], kp_response=k["response"],
                        kp_octave=k["octave"], desc=d, level_dims=np.array(lv, np.int32), cand_counts=np.array(cand_counts, np.int32))
    print(name, n, mono, cand_counts)


def main():
    orb_case("orb_160x120.npz", synth.small_image(1, 120, 160), (300, 1.2, 4, 20, 7))
    orb_case("orb_320x240.npz", synth.frame_stream(1, start=40)[0][100:340, 200:520].copy(), (500, 1.2, 8, 20, 7))
    # matching: two frames' keypoints -> windowed best/second best
    fr = synth.frame_stream(2, start=7)
    o = po.OrbOracle(500, 1.2, 8, 20, 7)
    _, k0, d0, _ = o.extract(fr[0][:240, :320].copy())
    _, k1, d1, _ = o.extract(fr[1][:240, :320].copy())
    sc = o.tables()["scale"]
    g = po.Grid(k1, 0.0, 320.0, 0.0, 240.0)
    qr = (np.float32(15) * sc[k0["octave"]]).astype(np.float32)
    m = g.match_window(d1, d0, k0["x"], k0["y"], qr, k0["octave"] - 1, k0["octave"] + 1)
    np.savez_compressed(os.path.join(HERE, "match_320x240.npz"), k0=k0, d0=d0, k1=k1, d1=d1, qr=qr, **m)
    print("match", len(k0), len(k1), int((m["best_dist"] <= 100).sum()))
    # bundle adjustment: small window
    pr = synth.ba_problem(n_kf=10, n_pts=200, seed=99)
    e = po.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    poses, pts, st, chi = po.ba_optimize(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], np.sqrt(5.991), 10)
    np.savez_compressed(os.path.join(HERE, "ba_10kf_200pt.npz"), poses0=pr["poses"], fixed=pr["fixed"], points0=pr["points"],
                        edges=e, intrinsics=pr["intrinsics"], delta=np.sqrt(5.991), iters=10, poses=poses, points=pts,
                        trials=np.array(st["trials"]), chi2=np.array(st["chi2"]), edge_chi2=chi)
    print("ba", st["trials"], st["chi2"][-1])


def more():
    """Fixtures for the entry points added after the first set (second call so the first files stay byte-identical)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from matcher_scene import make_local_map_scene, make_scene, make_sim3_scene
    # whole-function matchers
    sc = make_scene(po, 0, n_last=300, n_cur=340)
    n, mp = po.search_by_projection_frames(th=15.0, check_ori=True, **sc)
    sl = make_local_map_scene(po, 1, n_last=300, n_cur=340)
    n2, mp2 = po.search_by_projection_points(th=3.0, nnratio=0.8, far_points=True, th_far=9.0, **sl)
    np.savez_compressed(os.path.join(HERE, "matcher_functions.npz"), frames_n=n, frames_mp=mp, points_n=n2, points_mp=mp2,
                        **{"f_" + k: v for k, v in sc.items()}, **{"p_" + k: v for k, v in sl.items()})
    print("matchers", n, n2)
    # distinctive descriptors + vocabulary
    rng = np.random.default_rng(77)
    sizes = [1, 2, 5, 9, 16, 33, 70]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    base = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    desc = base[rng.integers(0, 40, off[-1])].copy()
    desc[rng.random(desc.shape) < 0.05] ^= 0x24
    bi, bm = po.distinctive_descriptors(desc, off)
    voc = synth.vocabulary(k=6, L=3, seed=11)
    feats = voc["desc"][rng.integers(1, voc["n_nodes"], 200)].copy()
    feats[rng.random(feats.shape) < 0.04] ^= 0x81
    r = po.vocab_transform(voc, feats, 2)
    np.savez_compressed(os.path.join(HERE, "bow_distinctive.npz"), dd_desc=desc, dd_off=off, dd_best=bi, dd_median=bm, feats=feats,
                        levelsup=2, **{"voc_" + k: np.asarray(v) for k, v in voc.items()}, **{"tr_" + k: v for k, v in r.items()})
    print("distinctive", list(bi), "bow", len(r["bow_ids"]))
    # Sim3Solver hypotheses + essential graph
    s3, _ = make_sim3_scene(5, n=80)
    tri = np.array([rng.choice(80, 3, replace=False) for _ in range(40)], np.int32)
    T, nin, mask = po.sim3_hypotheses(triples=tri, **s3)
    pg = synth.pose_graph(n=24, noise=0.002, seed=3)
    S, st = po.pose_graph_optimize(pg["S0"], pg["fixed"], pg["edges_v"], pg["edges_meas"], iterations=20)
    np.savez_compressed(os.path.join(HERE, "sim3_posegraph.npz"), triples=tri, T12=T, n_inliers=nin, mask=mask,
                        **{"s3_" + k: v for k, v in s3.items()}, pg_S0=pg["S0"], pg_fixed=pg["fixed"], pg_ev=pg["edges_v"],
                        pg_em=pg["edges_meas"], pg_S=S, pg_stats=st)
    print("sim3", int(nin.max()), "pose graph chi2", st[2], "->", st[3])


def kf_functions():
    """Third set: the remaining whole ORBmatcher functions on one small two-keyframe scene."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from matcher_scene import make_init_scene, make_kf_pair_scene
    sc = make_kf_pair_scene(po, 21, n_pts=260, n_clutter=80, n_nodes=40, dup_frac=0.25)
    a, b = sc["kf"]
    pts = sc["pts"]
    out = {}
    for tag, kf in (("a", a), ("b", b)):
        for k in ("kps", "desc", "mp", "bad", "Tcw"):
            out[f"{tag}_{k}"] = kf[k]
        for k, v in kf["fv"].items():
            out[f"{tag}_{k}"] = v
    for k in ("K", "bounds", "scale_factors", "level_sigma2", "inv_level_sigma2"):
        out[k] = a[k]
    out["log_scale_factor"] = np.float32(a["log_scale_factor"])
    for k, v in pts.items():
        out["pt_" + k] = v
    n1, m1 = po.search_by_bow_kf_kf(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["mp"], b["bad"], b["fv"], 0.8, True)
    n2, m2 = po.search_by_bow_kf_frame(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["fv"], 0.7, True)
    geo = po.triangulation_geometry(a["Tcw"], b["Tcw"], a["K"], b["K"])
    n3, pairs = po.search_for_triangulation(a["kps"], a["desc"], a["mp"], a["fv"], b["kps"], b["desc"], b["mp"], b["fv"], geo[3], geo[2],
                                            b["scale_factors"], b["level_sigma2"], False, True)
    bi, bd, pr = po.project_search(b["kps"], b["desc"], b["bounds"], None, b["Tcw"], po.se3_inverse(b["Tcw"])[4:], b["K"], pts, 3.0, b["scale_factors"],
                                   b["log_scale_factor"], b["inv_level_sigma2"], 5.99)
    matched = np.where(np.random.default_rng(2).random(len(b["kps"])) < 0.3, b["mp"], -1).astype(np.int32)
    # a genuine similarity (scale 1.3) whose SE3 part is keyframe b's pose: both Sim3 consumers decompose it themselves
    Scw = synth.sim3_from_sRt(1.3, b["Rcw"].reshape(3, 3), b["tcw"] * np.float32(1.3))
    out["b_Scw"] = Scw
    n4, m4 = po.search_by_projection_sim3(b["kps"], b["desc"], b["bounds"], matched, Scw, b["K"], pts, 8, 1.0,
                                          b["scale_factors"], b["log_scale_factor"])
    n5, mp5, rep5 = po.fuse_sim3(b["kps"], b["desc"], b["bounds"], b["mp"], b["bad"], Scw, b["K"], pts, 4.0,
                                 b["scale_factors"], b["log_scale_factor"])
    si = make_init_scene(po, 22, n=400)
    n6, m6, pm6 = po.search_for_initialization(si["k1"], si["d1"], si["k2"], si["d2"], si["bounds"], si["prev_matched"], 100, 0.9, True)
    np.savez_compressed(os.path.join(HERE, "kf_matcher_functions.npz"), bowkk_n=n1, bowkk_m=m1, bowkf_n=n2, bowkf_m=m2, tri_F12=geo[3],
                        tri_ep=geo[2], tri_n=n3, tri_pairs=pairs, ps_idx=bi, ps_dist=bd, ps_proj=pr, sim3_matched_in=matched, sim3_n=n4,
                        sim3_m=m4, fuse_n=n5, fuse_mp=mp5, fuse_rep=rep5, init_n=n6, init_m=m6, init_pm=pm6,
                        **{"i_" + k: v for k, v in si.items()}, **out)
    print("kf functions", n1, n2, n3, int((bi >= 0).sum()), n4, n5, n6)


def db_wire():
    """Fourth set: KeyFrameDatabase queries, the relocalisation search and one DVMW block (byte-exact format pin)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from kfdb_scene import fill, make_db_scene
    from matcher_scene import make_kf_pair_scene
    from wire_scene import make_delta
    from dvm_slam_amd import capi, wire
    kfs = make_db_scene(31, n_maps=2, kf_per_map=14, n_words=1500, words_per_kf=90)
    db = po.KeyFrameDatabase()
    fill(db, kfs)
    out = dict(db_n=len(kfs))
    for i, k in enumerate(kfs):
        out[f"db_ids_{i}"], out[f"db_vals_{i}"] = k["ids"], k["vals"]
        out[f"db_neigh_{i}"], out[f"db_conn_{i}"] = k["neigh"], k["connected"]
    out["db_meta"] = np.array([[k["map_id"], k["mn_id"]] for k in kfs], np.int64)
    out["db_uuid"] = np.array([k["uuid"] for k in kfs], np.uint64)
    q = [3, 9, 20]
    out["db_queries"] = np.array(q)
    out["db_merge"] = np.array([db.detect_merge_possibility(kfs[i]["ids"], kfs[i]["vals"], kfs[i]["uuid"], 1 - kfs[i]["map_id"]) for i in q], np.float64)
    nb = [db.detect_n_best(i, 3) for i in (5, 17)]
    for j, (lo, me) in enumerate(nb):
        out[f"db_loop_{j}"], out[f"db_mergecand_{j}"] = lo, me
    sc = make_kf_pair_scene(po, 33, n_pts=300, n_clutter=60, mapped_frac=0.8, dup_frac=0.2)
    a, b = sc["kf"]
    idx = np.where(a["pt_of_kp"] >= 0, a["pt_of_kp"], 0).astype(np.int64)
    pa = dict(pos=sc["pts"]["pos"][idx], min_dist=sc["pts"]["min_dist"][idx], max_dist=sc["pts"]["max_dist"][idx], desc=sc["pts"]["desc"][idx])
    cur_mp = np.where(np.random.default_rng(3).random(len(b["kps"])) < 0.2, b["mp"], -1).astype(np.int32)
    already = np.unique(cur_mp[cur_mp >= 0])
    n, m = po.search_by_projection_reloc(b["kps"], b["desc"], cur_mp, b["bounds"], b["Tcw"], b["K"], a, pa, already, 10.0, 100,
                                         b["scale_factors"], b["log_scale_factor"], True)
    for tag, kf in (("ra", a), ("rb", b)):
        for k in ("kps", "desc", "mp", "bad", "Tcw", "K", "bounds", "scale_factors"):
            out[f"{tag}_{k}"] = kf[k]
    out.update({"rp_" + k: v for k, v in pa.items()}, r_cur_mp=cur_mp, r_already=already, r_n=n, r_m=m, r_lsf=np.float32(b["log_scale_factor"]))
    dk, dm = make_delta(wire, capi, 7, 2, 10)
    out["wire_block"] = wire.build(dk, dm, sender_agent=3)
    np.savez_compressed(os.path.join(HERE, "db_wire.npz"), **out)
    print("db/wire", out["db_merge"][:, 0], n, out["wire_block"].size)


def triangulation():
    """LocalMapping::CreateNewMapPoints' per-match geometry (oracle.triangulate_matches) on a tests/tri_scene.py scene."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    import tri_scene
    S = tri_scene.scene(seed=77, n=240, baseline=0.45)
    X, st = po.triangulate_matches(S["K1"], S["K2"], S["T1w"], S["T2w"], S["Ow1"], S["Ow2"], S["kps1"], S["kps2"], S["pairs"], S["sigma2_1"],
                                   S["sigma2_2"], S["sf1"], S["sf2"], S["ratio_factor"], far_points=True, th_far=9.0)
    np.savez_compressed(os.path.join(HERE, "triangulation.npz"), x3D=X, status=st, th_far=np.float32(9.0), **S)
    print("triangulation", np.bincount(st, minlength=10))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "tri":
        triangulation()
    elif len(sys.argv) > 1 and sys.argv[1] == "db":
        db_wire()
    elif len(sys.argv) > 1 and sys.argv[1] == "more":
        more()
    elif len(sys.argv) > 1 and sys.argv[1] == "kf":
        kf_functions()
    else:
        main()
        more()
        kf_functions()
        db_wire()
        triangulation()
