"""Oracle restatements of the remaining whole ORBmatcher functions checked on the CPU against
independent brute-force numpy definitions of what each function must return on scenes WITHOUT sequential conflicts, and
for internal consistency (claims, mutual best) on scenes with them."""
import numpy as np

from dvm_slam_amd import synth
import pytest

from matcher_scene import make_init_scene, make_kf_pair_scene
from oracle import pyoracle as po


def _ham(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def test_search_for_initialization_properties():
    sc = make_init_scene(po, seed=3)
    n, m12, pm = po.search_for_initialization(sc["k1"], sc["d1"], sc["k2"], sc["d2"], sc["bounds"], sc["prev_matched"], window=100)
    assert n == int((m12 >= 0).sum()) and n > 300
    sel = np.nonzero(m12 >= 0)[0]
    assert len(np.unique(m12[sel])) == len(sel), "one F2 keypoint matched twice"
    for i1 in sel:
        i2 = m12[i1]
        assert sc["k1"]["octave"][i1] == 0 and sc["k2"]["octave"][i2] == 0
        assert abs(sc["k2"]["x"][i2] - sc["prev_matched"][i1, 0]) < 100 and abs(sc["k2"]["y"][i2] - sc["prev_matched"][i1, 1]) < 100
        assert _ham(sc["d1"][i1], sc["d2"][i2]) <= 50
        assert pm[i1, 0] == sc["k2"]["x"][i2] and pm[i1, 1] == sc["k2"]["y"][i2]
    untouched = m12 < 0
    assert np.array_equal(pm[untouched], sc["prev_matched"][untouched])
    # without the orientation filter there are at least as many matches
    n2, _, _ = po.search_for_initialization(sc["k1"], sc["d1"], sc["k2"], sc["d2"], sc["bounds"], sc["prev_matched"], window=100, check_ori=False)
    assert n2 >= n


def _brute_bow(kf1, kf2, nnratio, th_strict):
    """SearchByBoW(KF1, KF2) without the vbMatched2 rule and without the histogram: valid only where no KF2 feature is chosen twice."""
    out = {}
    fa, fb = kf1["fv"], kf2["fv"]
    nb = {int(n): k for k, n in enumerate(fb["fv_nodes"])}
    for ka, node in enumerate(fa["fv_nodes"]):
        if int(node) not in nb:
            continue
        kb = nb[int(node)]
        c2 = [int(j) for j in fb["fv_feat"][fb["fv_off"][kb]:fb["fv_off"][kb + 1]] if kf2["mp"][j] >= 0 and not kf2["bad"][j]]
        for i in fa["fv_feat"][fa["fv_off"][ka]:fa["fv_off"][ka + 1]]:
            i = int(i)
            if kf1["mp"][i] < 0 or kf1["bad"][i] or not c2:
                continue
            d = sorted((_ham(kf1["desc"][i], kf2["desc"][j]), k) for k, j in enumerate(c2))
            b1 = d[0][0]; b2 = d[1][0] if len(d) > 1 else 256
            if (b1 < 50 if th_strict else b1 <= 50) and np.float32(b1) < np.float32(nnratio) * np.float32(b2):
                out[i] = c2[d[0][1]]
    return out


def test_search_by_bow_kf_kf_matches_definition():
    sc = make_kf_pair_scene(po, seed=5, dup_frac=0.0)
    a, b = sc["kf"]
    n, m12 = po.search_by_bow_kf_kf(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["mp"], b["bad"], b["fv"],
                                    nnratio=0.8, check_ori=False)
    ref = _brute_bow(a, b, 0.8, True)
    assert len(set(ref.values())) == len(ref), "scene has conflicts; use dup_frac=0"
    got = {int(i): int(m12[i]) for i in np.nonzero(m12 >= 0)[0]}
    assert n == len(got) > 150
    assert got == {i: int(b["mp"][j]) for i, j in ref.items()}


def test_search_by_bow_with_conflicts_is_consistent():
    sc = make_kf_pair_scene(po, seed=6, dup_frac=0.2)
    a, b = sc["kf"]
    n, m12 = po.search_by_bow_kf_kf(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["mp"], b["bad"], b["fv"])
    assert n == int((m12 >= 0).sum()) > 100
    n2, mf = po.search_by_bow_kf_frame(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["fv"])
    assert n2 == int((mf >= 0).sum()) > 100
    got = mf[mf >= 0]
    assert set(got.tolist()) <= set(a["mp"][(a["mp"] >= 0) & (a["bad"] == 0)].tolist())
    # most matches are the true correspondence (same 3-D point)
    true = sum(1 for j in np.nonzero(mf >= 0)[0] if b["pt_of_kp"][j] >= 0 and b["pt_of_kp"][j] + 1000 == mf[j])
    assert true > 0.9 * n2


def test_search_for_triangulation_definition():
    sc = make_kf_pair_scene(po, seed=7)
    a, b = sc["kf"]
    R12, t12, ep, F12 = po.triangulation_geometry(a["Tcw"], b["Tcw"], a["K"], b["K"])
    # F12 against the double-precision definition
    R1, R2 = a["Rcw"].reshape(3, 3).astype(np.float64), b["Rcw"].reshape(3, 3).astype(np.float64)
    Rd = R1 @ R2.T; td = a["tcw"].astype(np.float64) - Rd @ b["tcw"].astype(np.float64)
    Kd = np.array([[500.0, 0, 320], [0, 500, 240], [0, 0, 1]])
    tx = np.array([[0, -td[2], td[1]], [td[2], 0, -td[0]], [-td[1], td[0], 0]])
    Fd = np.linalg.inv(Kd).T @ tx @ Rd @ np.linalg.inv(Kd)
    assert np.allclose(F12.reshape(3, 3), Fd, atol=1e-6 * max(1.0, np.abs(Fd).max()))
    n, pairs = po.search_for_triangulation(a["kps"], a["desc"], a["mp"], a["fv"], b["kps"], b["desc"], b["mp"], b["fv"], F12, ep,
                                           b["scale_factors"], b["level_sigma2"], coarse=False, check_ori=False)
    assert n == len(pairs) > 100
    assert np.all(a["mp"][pairs[:, 0]] < 0) and np.all(b["mp"][pairs[:, 1]] < 0)
    assert np.all(np.diff(pairs[:, 0]) > 0)
    good = np.mean(a["pt_of_kp"][pairs[:, 0]] == b["pt_of_kp"][pairs[:, 1]])
    assert good > 0.9
    # every pair satisfies the gates it was accepted under
    for i1, i2 in pairs:
        assert _ham(a["desc"][i1], b["desc"][i2]) <= 50
        x1 = np.array([a["kps"]["x"][i1], a["kps"]["y"][i1], 1.0]); x2 = np.array([b["kps"]["x"][i2], b["kps"]["y"][i2], 1.0])
        l = x1 @ Fd
        assert (l @ x2) ** 2 / (l[0] ** 2 + l[1] ** 2) < 3.84 * b["level_sigma2"][b["kps"]["octave"][i2]] * 1.001
    nc, _ = po.search_for_triangulation(a["kps"], a["desc"], a["mp"], a["fv"], b["kps"], b["desc"], b["mp"], b["fv"], F12, ep,
                                        b["scale_factors"], b["level_sigma2"], coarse=True, check_ori=False)
    assert nc >= n


def test_project_search_and_fuse():
    sc = make_kf_pair_scene(po, seed=8)
    kf, pts = sc["kf"][1], sc["pts"]
    bi, bd, pr = po.project_search(kf["kps"], kf["desc"], kf["bounds"], None, kf["Tcw"], po.se3_inverse(kf["Tcw"])[4:], kf["K"], pts, 3.0,
                                   kf["scale_factors"], kf["log_scale_factor"], gate_inv_sigma2=kf["inv_level_sigma2"], gate=5.99)
    hit = bi >= 0
    assert hit.sum() > 300
    # brute force per point
    R = kf["Rcw"].reshape(3, 3)
    for i in np.nonzero(pr[:, 3] >= 0)[0][:200]:
        u, v, r, l = pr[i]
        best = (256, -1)
        for j in range(len(kf["kps"])):
            kp = kf["kps"][j]
            if not (abs(kp["x"] - u) < r and abs(kp["y"] - v) < r) or kp["octave"] < l - 1 or kp["octave"] > l:
                continue
            e2 = np.float32(u - kp["x"]) ** 2 + np.float32(v - kp["y"]) ** 2
            if np.float32(e2) * kf["inv_level_sigma2"][kp["octave"]] > 5.99:
                continue
            d = _ham(pts["desc"][i], kf["desc"][j])
            if d < best[0]:
                best = (d, j)
        assert best[0] == bd[i]
        if best[1] >= 0:
            assert _ham(pts["desc"][i], kf["desc"][bi[i]]) == best[0]
    # the Sim3 variants: scale 1 similarity == the SE3 pose
    Scw = synth.sim3_from_sRt(1.0, kf["Rcw"].reshape(3, 3), kf["tcw"])
    nf, mp_new, rep = po.fuse_sim3(kf["kps"], kf["desc"], kf["bounds"], kf["mp"], kf["bad"], Scw, kf["K"], pts, 4.0,
                                   kf["scale_factors"], kf["log_scale_factor"])
    assert nf > 100 and nf == int((rep >= 0).sum()) + int((mp_new != kf["mp"]).sum())
    matched = np.where(np.random.default_rng(1).random(len(kf["kps"])) < 0.3, kf["mp"], -1).astype(np.int32)
    nm, m2 = po.search_by_projection_sim3(kf["kps"], kf["desc"], kf["bounds"], matched, Scw, kf["K"], pts, 8, 1.0,
                                          kf["scale_factors"], kf["log_scale_factor"])
    assert nm == int((m2 != matched).sum()) > 50
    new_ids = m2[m2 != matched]
    assert len(np.unique(new_ids)) == len(new_ids) and not set(new_ids.tolist()) & set(matched[matched >= 0].tolist())
