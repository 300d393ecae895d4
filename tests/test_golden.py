"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU part: the oracle still reproduces them.  GPU part (-m gpu): the HIP path reproduces them through
the C ABI -- bit-exact for keypoints / descriptors / matches, 1e-6 for BA states."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ORB_CASES = ["orb_160x120.npz", "orb_320x240.npz"]


def _params(z):
    p = z["params"]
    return int(p[0]), float(p[1]), int(p[2]), int(p[3]), int(p[4])


def _check_orb(z, n, k, d, mono):
    assert (n, mono) == (int(z["n"]), int(z["mono"]))
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(k[f], z["kp_" + f]), f
    assert np.array_equal(d, z["desc"])


@pytest.mark.parametrize("case", ORB_CASES)
def test_oracle_reproduces_orb_golden(oracle, case):
    z = np.load(os.path.join(G, case))
    o = oracle.OrbOracle(*_params(z))
    _check_orb(z, *o.extract(z["image"]))
    assert [o.level_dims(l) for l in range(_params(z)[2])] == [tuple(r) for r in z["level_dims"]]


def test_oracle_reproduces_match_golden(oracle):
    z = np.load(os.path.join(G, "match_320x240.npz"))
    g = oracle.Grid(z["k1"], 0.0, 320.0, 0.0, 240.0)
    m = g.match_window(z["d1"], z["d0"], z["k0"]["x"], z["k0"]["y"], z["qr"], z["k0"]["octave"] - 1, z["k0"]["octave"] + 1)
    for f in ("best_idx", "best_dist", "second_dist", "best_level", "second_level"):
        assert np.array_equal(m[f], z[f]), f


def test_oracle_reproduces_ba_golden(oracle):
    z = np.load(os.path.join(G, "ba_10kf_200pt.npz"))
    poses, pts, st, chi = oracle.ba_optimize(z["poses0"], z["fixed"], z["points0"], z["edges"], z["intrinsics"], float(z["delta"]), int(z["iters"]))
    assert st["trials"] == list(z["trials"])
    assert np.allclose(poses, z["poses"], atol=1e-9) and np.allclose(pts, z["points"], atol=1e-9)
    assert np.allclose(st["chi2"], z["chi2"], rtol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ORB_CASES)
def test_hip_reproduces_orb_golden(capi, case):
    z = np.load(os.path.join(G, case))
    e = capi.OrbExtractor(*_params(z), max_batch=1)
    _check_orb(z, *e.extract(z["image"]))
    e.close()


@pytest.mark.gpu
def test_hip_reproduces_match_golden(capi):
    z = np.load(os.path.join(G, "match_320x240.npz"))
    g = capi.FrameGrid(capacity=1024)
    g.build(z["k1"], z["d1"], bounds=(0.0, 320.0, 0.0, 240.0))
    m = g.match_window(z["d0"], z["k0"]["x"], z["k0"]["y"], z["qr"], z["k0"]["octave"] - 1, z["k0"]["octave"] + 1)
    for f in ("best_idx", "best_dist", "second_dist", "best_level", "second_level"):
        assert np.array_equal(m[f].astype(np.int64), z[f].astype(np.int64)), f
    g.close()


@pytest.mark.gpu
def test_hip_reproduces_ba_golden(capi):
    z = np.load(os.path.join(G, "ba_10kf_200pt.npz"))
    ba = capi.BundleAdjuster()
    ba.set_problem(z["poses0"], z["fixed"], z["points0"], z["edges"], z["intrinsics"], float(z["delta"]))
    st = ba.optimize(int(z["iters"]))
    p, pts = ba.result()
    assert st["trials"] == list(z["trials"])
    assert np.abs(p - z["poses"]).max() < 1e-6 and np.abs(pts - z["points"]).max() < 1e-6
    assert np.allclose(st["chi2"], z["chi2"], rtol=1e-9)
    ba.close()


# ---- second set: whole-function matchers, ComputeDistinctiveDescriptors, DBoW2 transform, Sim3Solver, essential graph
def _sub(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def _voc(z):
    v = _sub(z, "voc_")
    return dict(n_nodes=int(v["n_nodes"]), child_off=v["child_off"], children=v["children"], desc=v["desc"], weight=v["weight"],
                word_id=v["word_id"], L=int(v["L"]))


def test_oracle_reproduces_second_set(oracle):
    z = np.load(os.path.join(G, "matcher_functions.npz"))
    n, mp = oracle.search_by_projection_frames(th=15.0, check_ori=True, **_sub(z, "f_"))
    assert n == int(z["frames_n"]) and np.array_equal(mp, z["frames_mp"])
    n, mp = oracle.search_by_projection_points(th=3.0, nnratio=0.8, far_points=True, th_far=9.0, **_sub(z, "p_"))
    assert n == int(z["points_n"]) and np.array_equal(mp, z["points_mp"])
    z = np.load(os.path.join(G, "bow_distinctive.npz"))
    bi, bm = oracle.distinctive_descriptors(z["dd_desc"], z["dd_off"])
    assert np.array_equal(bi, z["dd_best"]) and np.array_equal(bm, z["dd_median"])
    r = oracle.vocab_transform(_voc(z), z["feats"], int(z["levelsup"]))
    for k, v in r.items():
        assert np.array_equal(v, z["tr_" + k]), k
    z = np.load(os.path.join(G, "sim3_posegraph.npz"))
    T, nin, mask = oracle.sim3_hypotheses(triples=z["triples"], **_sub(z, "s3_"))
    assert np.array_equal(T, z["T12"]) and np.array_equal(nin, z["n_inliers"]) and np.array_equal(mask, z["mask"])
    S, st = oracle.pose_graph_optimize(z["pg_S0"], z["pg_fixed"], z["pg_ev"], z["pg_em"], iterations=20)
    assert np.allclose(S, z["pg_S"], atol=1e-12) and np.allclose(st[:6], z["pg_stats"][:6], rtol=1e-12)


@pytest.mark.gpu
def test_hip_reproduces_second_set(capi):
    z = np.load(os.path.join(G, "matcher_functions.npz"))
    n, mp, _ = capi.search_by_projection_frames(th=15.0, check_ori=True, **_sub(z, "f_"))
    assert n == int(z["frames_n"]) and np.array_equal(mp, z["frames_mp"])
    n, mp, _ = capi.search_by_projection_points(th=3.0, nnratio=0.8, far_points=True, th_far=9.0, **_sub(z, "p_"))
    assert n == int(z["points_n"]) and np.array_equal(mp, z["points_mp"])
    z = np.load(os.path.join(G, "bow_distinctive.npz"))
    bi, bm = capi.distinctive_descriptors(z["dd_desc"], z["dd_off"])
    assert np.array_equal(bi, z["dd_best"]) and np.array_equal(bm, z["dd_median"])
    voc = _voc(z)
    v = capi.Vocabulary(voc)
    w, nd, wt = v.transform(z["feats"], int(z["levelsup"]))
    assert np.array_equal(w, z["tr_word"]) and np.array_equal(nd, z["tr_node"]) and np.array_equal(wt, z["tr_weight"])
    v.close()
    h = capi.vocab_transform_host(voc, z["feats"], int(z["levelsup"]))
    for k in ("bow_ids", "bow_vals", "fv_nodes", "fv_off", "fv_feat"):
        assert np.array_equal(h[k], z["tr_" + k]), k
    z = np.load(os.path.join(G, "sim3_posegraph.npz"))
    T, nin, mask = capi.sim3_hypotheses(triples=z["triples"], **_sub(z, "s3_"))
    assert np.allclose(T, z["T12"], rtol=1e-5, atol=1e-5) and np.all(np.abs(nin - z["n_inliers"]) <= 1)
    S, st = capi.pose_graph_optimize(z["pg_S0"], z["pg_fixed"], z["pg_ev"], z["pg_em"], iterations=20)
    assert abs(st["chi2_per_iter"][0] - z["pg_stats"][6]) <= 5e-5 * z["pg_stats"][6]        # first LM step (see test_gpu_ba)
    assert st["chi2_final"] <= 1.5 * z["pg_stats"][3] and z["pg_stats"][3] <= 1.5 * st["chi2_final"]


def _kf_from_golden(z, tag):
    kf = {k: z[f"{tag}_{k}"] for k in ("kps", "desc", "mp", "bad", "Tcw")}
    kf["fv"] = {k: z[f"{tag}_{k}"] for k in ("fv_nodes", "fv_off", "fv_feat")}
    for k in ("K", "bounds", "scale_factors", "level_sigma2", "inv_level_sigma2"):
        kf[k] = z[k]
    kf["log_scale_factor"] = float(z["log_scale_factor"])
    return kf


def test_oracle_reproduces_kf_matcher_functions(oracle):
    z = np.load(os.path.join(G, "kf_matcher_functions.npz"))
    a, b = _kf_from_golden(z, "a"), _kf_from_golden(z, "b")
    pts = _sub(z, "pt_")
    n, m = oracle.search_by_bow_kf_kf(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["mp"], b["bad"], b["fv"], 0.8, True)
    assert n == int(z["bowkk_n"]) and np.array_equal(m, z["bowkk_m"])
    n, m = oracle.search_by_bow_kf_frame(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["fv"], 0.7, True)
    assert n == int(z["bowkf_n"]) and np.array_equal(m, z["bowkf_m"])
    geo = oracle.triangulation_geometry(a["Tcw"], b["Tcw"], a["K"], b["K"])
    assert np.array_equal(geo[3], z["tri_F12"]) and np.array_equal(geo[2], z["tri_ep"])
    n, pairs = oracle.search_for_triangulation(a["kps"], a["desc"], a["mp"], a["fv"], b["kps"], b["desc"], b["mp"], b["fv"], geo[3], geo[2],
                                               b["scale_factors"], b["level_sigma2"], False, True)
    assert n == int(z["tri_n"]) and np.array_equal(pairs, z["tri_pairs"])
    bi, bd, pr = oracle.project_search(b["kps"], b["desc"], b["bounds"], None, b["Tcw"], oracle.se3_inverse(b["Tcw"])[4:], b["K"], pts, 3.0, b["scale_factors"],
                                       b["log_scale_factor"], b["inv_level_sigma2"], 5.99)
    assert np.array_equal(bi, z["ps_idx"]) and np.array_equal(bd, z["ps_dist"]) and np.array_equal(pr, z["ps_proj"])
    n, m = oracle.search_by_projection_sim3(b["kps"], b["desc"], b["bounds"], z["sim3_matched_in"], z["b_Scw"], b["K"], pts, 8,
                                            1.0, b["scale_factors"], b["log_scale_factor"])
    assert n == int(z["sim3_n"]) and np.array_equal(m, z["sim3_m"])
    n, mp, rep = oracle.fuse_sim3(b["kps"], b["desc"], b["bounds"], b["mp"], b["bad"], z["b_Scw"], b["K"], pts, 4.0,
                                  b["scale_factors"], b["log_scale_factor"])
    assert n == int(z["fuse_n"]) and np.array_equal(mp, z["fuse_mp"]) and np.array_equal(rep, z["fuse_rep"])
    si = _sub(z, "i_")
    n, m, pm = oracle.search_for_initialization(si["k1"], si["d1"], si["k2"], si["d2"], si["bounds"], si["prev_matched"], 100, 0.9, True)
    assert n == int(z["init_n"]) and np.array_equal(m, z["init_m"]) and np.array_equal(pm, z["init_pm"])


@pytest.mark.gpu
def test_hip_reproduces_kf_matcher_functions(capi):
    z = np.load(os.path.join(G, "kf_matcher_functions.npz"))
    a, b = _kf_from_golden(z, "a"), _kf_from_golden(z, "b")
    pts = _sub(z, "pt_")
    a["mp"] = a["mp"].copy(); b["mp"] = b["mp"].copy()
    va, vb = capi.keyframe_view(a), capi.keyframe_view(b)
    n, m, _ = capi.search_by_bow_kf_kf(va, vb, 0.8, True)
    assert n == int(z["bowkk_n"]) and np.array_equal(m, z["bowkk_m"])
    F = capi.frame_view(b["kps"], b["desc"], b["bounds"], b["scale_factors"])
    n, m, _ = capi.search_by_bow_kf_frame(va, F, b["fv"], 0.7, True)
    assert n == int(z["bowkf_n"]) and np.array_equal(m, z["bowkf_m"])
    geo = capi.triangulation_geometry(va, vb)
    assert np.array_equal(geo[3], z["tri_F12"]) and np.array_equal(geo[2], z["tri_ep"])
    n, pairs = capi.search_for_triangulation(va, vb, False, True)
    assert n == int(z["tri_n"]) and np.array_equal(pairs, z["tri_pairs"])
    P = capi.map_points_view(pts)
    n, bi = capi.fuse(vb, capi.map_points_view({k: v for k, v in pts.items() if k != "bad"}), None, 3.0)
    assert np.array_equal(bi, np.where((z["ps_idx"] >= 0) & (z["ps_dist"] <= 50), z["ps_idx"], -1))
    n, m, _ = capi.search_by_projection_sim3(vb, z["b_Scw"], P, z["sim3_matched_in"], 8, 1.0)
    assert n == int(z["sim3_n"]) and np.array_equal(m, z["sim3_m"])
    n, rep = capi.fuse_sim3(vb, z["b_Scw"], P, 4.0)
    assert n == int(z["fuse_n"]) and np.array_equal(b["mp"], z["fuse_mp"]) and np.array_equal(rep, z["fuse_rep"])
    si = _sub(z, "i_")
    F1 = capi.frame_view(si["k1"], si["d1"], si["bounds"], si["scale_factors"])
    F2 = capi.frame_view(si["k2"], si["d2"], si["bounds"], si["scale_factors"])
    n, m, pm = capi.search_for_initialization(F1, F2, si["prev_matched"], 100, 0.9, True)
    assert n == int(z["init_n"]) and np.array_equal(m, z["init_m"]) and np.array_equal(pm, z["init_pm"])


def _db_from_golden(z, db):
    n = int(z["db_n"])
    for i in range(n):
        s = db.add(z[f"db_ids_{i}"], z[f"db_vals_{i}"], int(z["db_meta"][i, 0]), int(z["db_uuid"][i]), int(z["db_meta"][i, 1]))
        assert s == i
    for i in range(n):
        db.set_neighbours(i, z[f"db_neigh_{i}"]); db.set_connected(i, z[f"db_conn_{i}"])
    return n


def _check_db(z, db):
    _db_from_golden(z, db)
    for row, i in zip(z["db_merge"], z["db_queries"]):
        i = int(i)
        r = db.detect_merge_possibility(z[f"db_ids_{i}"], z[f"db_vals_{i}"], int(z["db_uuid"][i]), 1 - int(z["db_meta"][i, 0]))
        assert np.array_equal(np.array(r, np.float64), row), (i, r, row)
    for j, i in enumerate((5, 17)):
        lo, me = db.detect_n_best(i, 3)
        assert np.array_equal(lo, z[f"db_loop_{j}"]) and np.array_equal(me, z[f"db_mergecand_{j}"])


def test_oracle_reproduces_db_wire_set(oracle, capi):
    from dvm_slam_amd import wire
    from wire_scene import make_delta
    z = np.load(os.path.join(G, "db_wire.npz"))
    _check_db(z, oracle.KeyFrameDatabase())
    a = {k: z[f"ra_{k}"] for k in ("kps", "desc", "mp", "bad")}
    n, m = oracle.search_by_projection_reloc(z["rb_kps"], z["rb_desc"], z["r_cur_mp"], z["rb_bounds"], z["rb_Tcw"], z["rb_K"], a,
                                             _sub(z, "rp_"), z["r_already"], 10.0, 100, z["rb_scale_factors"], float(z["r_lsf"]), True)
    assert n == int(z["r_n"]) and np.array_equal(m, z["r_m"])
    # the DVMW block is a format pin: the builder must reproduce it byte for byte, and it must parse
    dk, dm = make_delta(wire, capi, 7, 2, 10)
    blk = wire.build(dk, dm, sender_agent=3)
    assert np.array_equal(blk, z["wire_block"])
    h, kfs, mps = wire.parse(z["wire_block"])
    assert int(h["n_keyframes"]) == 2 and int(h["n_mappoints"]) == 10 and int(h["sender_agent"]) == 3


@pytest.mark.gpu
def test_hip_reproduces_db_wire_set(capi):
    z = np.load(os.path.join(G, "db_wire.npz"))
    _check_db(z, capi.HostKeyFrameDatabase())
    a = {k: z[f"ra_{k}"] for k in ("kps", "desc", "mp", "bad")}
    a.update(bounds=z["ra_bounds"], scale_factors=z["ra_scale_factors"], log_scale_factor=float(z["r_lsf"]))
    a["mp"] = a["mp"].copy()
    m = z["r_cur_mp"].copy()
    F = capi.frame_view(z["rb_kps"], z["rb_desc"], z["rb_bounds"], z["rb_scale_factors"], mp=m, K=z["rb_K"], Tcw=z["rb_Tcw"])
    pts = _sub(z, "rp_"); pts["normal"] = pts["pos"]
    n, _ = capi.search_by_projection_reloc(F, capi.keyframe_view(a), capi.map_points_view(pts), z["r_already"], 10.0, 100, True)
    assert n == int(z["r_n"]) and np.array_equal(m, z["r_m"])


def _tri_args(z):
    return (z["K1"], z["K2"], z["T1w"], z["T2w"], z["Ow1"], z["Ow2"], z["kps1"], z["kps2"], z["pairs"], z["sigma2_1"], z["sigma2_2"], z["sf1"],
            z["sf2"], float(z["ratio_factor"]))


def test_oracle_reproduces_triangulation_set(oracle):
    z = np.load(os.path.join(G, "triangulation.npz"))
    X, st = oracle.triangulate_matches(*_tri_args(z), far_points=True, th_far=float(z["th_far"]))
    assert np.array_equal(st, z["status"]) and np.array_equal(X.view(np.uint32), z["x3D"].view(np.uint32))
    assert {0, 1, 3, 5, 8, 9} <= set(np.unique(st).tolist())


@pytest.mark.gpu
def test_hip_reproduces_triangulation_set(capi):
    z = np.load(os.path.join(G, "triangulation.npz"))
    X, st = capi.triangulate_matches(*_tri_args(z), far_points=True, th_far=float(z["th_far"]))
    assert np.array_equal(st, z["status"]) and np.array_equal(X.view(np.uint32), z["x3D"].view(np.uint32))
