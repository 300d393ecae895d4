"""Two agents that mapped the same place in their own (monocular: arbitrary scale) world frames -- input of the
config-3 merge pipeline test."""
import numpy as np

from dvm_slam_amd import synth
from matcher_scene import make_kf_pair_scene


def make_two_agent_scene(oracle, seed=0, n_distract=12, s_w=1.6):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed + 500)
    sc = make_kf_pair_scene(oracle, seed, n_pts=1100, n_clutter=150, mapped_frac=0.85, flip_bits=10, dup_frac=0.05)
    a, b = sc["kf"]
    pts = sc["pts"]
    # agent B's world frame: X_B = s_w * R_w X + t_w; its keyframe pose in ITS units
    R_w = Rotation.from_rotvec(rng.normal(0, 0.4, 3)).as_matrix().astype(np.float32)
    t_w = rng.normal(0, 2.0, 3).astype(np.float32)
    R2 = b["Rcw"].reshape(3, 3)
    R2p = (R2 @ R_w.T).astype(np.float32)
    t2p = (np.float32(s_w) * b["tcw"] - R2p @ t_w).astype(np.float32)
    XB = (np.float32(s_w) * (pts["pos"] @ R_w.T) + t_w).astype(np.float32)

    def per_kp(kf, pos, scale):
        idx = np.where(kf["pt_of_kp"] >= 0, kf["pt_of_kp"], 0).astype(np.int64)
        return dict(pos=pos[idx], normal=pts["normal"][idx], min_dist=(pts["min_dist"][idx] * scale).astype(np.float32),
                    max_dist=(pts["max_dist"][idx] * scale).astype(np.float32), desc=pts["desc"][idx])
    a = dict(a, uuid=111, map_id=0, mn_id=1)
    b = dict(b, Tcw=synth.se3_from_Rt(R2p, t2p), Rcw=R2p.reshape(-1), tcw=t2p, Ow=(-(R2p.T @ t2p)).astype(np.float32), uuid=222, map_id=1, mn_id=1,
             mp=np.where(b["mp"] >= 0, b["mp"] + 100000, -1).astype(np.int32))     # B's own map point ids
    pa, pb = per_kp(a, pts["pos"], 1.0), per_kp(b, XB, s_w)
    wrong = rng.random(len(pb["pos"])) < 0.15          # badly triangulated points in B's map: geometric outliers for RANSAC
    pb["pos"] = pb["pos"].copy(); pb["pos"][wrong] += rng.normal(0, 1.5 * s_w, (int(wrong.sum()), 3)).astype(np.float32)
    # distractor keyframes of agent B: other places (random descriptors)
    peers, peer_pts = [b], [pb]
    for k in range(n_distract):
        n = int(rng.integers(600, 1100))
        d = dict(b, kps=b["kps"][:n].copy(), desc=rng.integers(0, 256, (n, 32), dtype=np.uint8), mp=np.full(n, -1, np.int32),
                 bad=np.zeros(n, np.uint8), uuid=300 + k, mn_id=2 + k, pt_of_kp=np.full(n, -1))
        peers.append(d); peer_pts.append(dict(pos=np.zeros((n, 3), np.float32), normal=np.zeros((n, 3), np.float32),
                                              min_dist=np.ones(n, np.float32), max_dist=np.ones(n, np.float32), desc=d["desc"]))
    order = rng.permutation(len(peers))
    peers = [peers[i] for i in order]; peer_pts = [peer_pts[i] for i in order]
    true_idx = int(np.flatnonzero(order == 0)[0])
    for i, p in enumerate(peers):     # the distractors are unrelated places: no covisibility between any of B's keyframes here
        p["neigh"] = np.zeros(0, np.int32)
    # ground truth S12: p_c1 = R12 p_c2_common + t12 with p_c2_common = p_c2(B units) / s_w
    R1 = a["Rcw"].reshape(3, 3).astype(np.float64)
    R12 = R1 @ R2.astype(np.float64).T
    t12 = a["tcw"].astype(np.float64) - R12 @ sc["kf"][1]["tcw"].astype(np.float64)
    return dict(a=a, pa=pa, peers=peers, peer_pts=peer_pts, true_idx=true_idx, gt=dict(s=1.0 / s_w, R=R12, t=t12))
