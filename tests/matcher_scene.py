"""Synthetic two-frame tracking scene for the whole-function SearchByProjection(CurrentFrame, LastFrame) tests."""
import numpy as np

from dvm_slam_amd import synth


def make_scene(oracle, seed=0, n_last=1000, n_cur=1100, dup_frac=0.15, zero_obs_frac=0.1, flip_bits=20):
    """LastFrame keypoints carry map points; CurrentFrame keypoints are noisy re-projections (plus clutter) so that
    several queries compete for the same current keypoint (exercises the sequential claim rule)."""
    rng = np.random.default_rng(seed)
    K = np.array([500.0, 500.0, 320.0, 240.0], np.float32)
    bounds = np.array([0.0, 640.0, 0.0, 480.0], np.float32)
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    # last frame at identity
    kl = np.zeros(n_last, oracle.KP_DTYPE)
    kl["x"] = rng.uniform(20, 620, n_last).astype(np.float32)
    kl["y"] = rng.uniform(20, 460, n_last).astype(np.float32)
    kl["octave"] = rng.integers(0, 8, n_last)
    kl["angle"] = rng.uniform(0, 360, n_last).astype(np.float32)
    z = rng.uniform(2, 12, n_last).astype(np.float32)
    mps = np.zeros(n_last, oracle.MAP_POINT_DTYPE)
    mps["pos"][:, 0] = (kl["x"] - K[2]) / K[0] * z
    mps["pos"][:, 1] = (kl["y"] - K[3]) / K[1] * z
    mps["pos"][:, 2] = z
    mps["desc"] = rng.integers(0, 256, (n_last, 32), dtype=np.uint8)
    mps["n_obs"] = np.where(rng.random(n_last) < zero_obs_frac, 0, rng.integers(1, 6, n_last))
    # duplicates: a second map point at (almost) the same place with the same descriptor -> competes for one keypoint
    ndup = int(dup_frac * n_last)
    src = rng.choice(n_last, ndup, replace=False)
    dst = rng.choice(np.setdiff1d(np.arange(n_last), src), ndup, replace=False)
    mps["pos"][dst] = mps["pos"][src] + rng.normal(0, 1e-3, (ndup, 3)).astype(np.float32)
    mps["desc"][dst] = mps["desc"][src]
    kl["octave"][dst] = kl["octave"][src]
    # a few behind the camera / far outside
    mps["pos"][rng.choice(n_last, 20, replace=False), 2] *= -1
    mp_l = np.arange(n_last, dtype=np.int32)
    mp_l[rng.random(n_last) < 0.1] = -1
    outl = (rng.random(n_last) < 0.05).astype(np.uint8)
    # current pose: small motion
    a = 0.01
    Rcw = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    tcw = np.array([0.03, -0.02, 0.05], np.float32)
    Xc = mps["pos"] @ Rcw.T + tcw
    u = K[0] * Xc[:, 0] / Xc[:, 2] + K[2]
    v = K[1] * Xc[:, 1] / Xc[:, 2] + K[3]
    kc = np.zeros(n_cur, oracle.KP_DTYPE)
    dc = rng.integers(0, 256, (n_cur, 32), dtype=np.uint8)
    pick = rng.choice(n_last, min(n_last, n_cur - 100), replace=False)
    m = len(pick)
    kc["x"][:m] = (u[pick] + rng.normal(0, 2.0, m)).astype(np.float32)
    kc["y"][:m] = (v[pick] + rng.normal(0, 2.0, m)).astype(np.float32)
    kc["octave"][:m] = np.clip(kl["octave"][pick] + rng.integers(-1, 2, m), 0, 7)
    # most matches share one rotation; a minority is rotated elsewhere -> removed by the histogram filter
    rot = np.where(rng.random(m) < 0.8, 10.0, rng.uniform(0, 360, m))
    kc["angle"][:m] = np.mod(kl["angle"][pick] - rot + 720.0, 360.0).astype(np.float32)
    d = mps["desc"][pick].copy()
    bits = np.unpackbits(d, axis=1)
    for r in range(m):
        bits[r, rng.choice(256, rng.integers(0, flip_bits + 1), replace=False)] ^= 1
    dc[:m] = np.packbits(bits, axis=1)
    kc["x"][m:] = rng.uniform(0, 640, n_cur - m).astype(np.float32)
    kc["y"][m:] = rng.uniform(0, 480, n_cur - m).astype(np.float32)
    kc["octave"][m:] = rng.integers(0, 8, n_cur - m)
    kc["angle"][m:] = rng.uniform(0, 360, n_cur - m).astype(np.float32)
    perm = rng.permutation(n_cur)
    kc, dc = kc[perm], dc[perm]
    mp_c = np.full(n_cur, -1, np.int32)
    pre = rng.choice(n_cur, 40, replace=False)          # already-associated keypoints (some with 0 observations)
    mp_c[pre] = rng.integers(0, n_last, 40)
    # Tcw: the pose as the reference holds it (Sophus::SE3f, 7 floats: unit quaternion x, y, z, w + translation)
    return dict(kps_c=kc, desc_c=dc, mp_c=mp_c, Tcw=synth.se3_from_Rt(Rcw, tcw), K=K, bounds=bounds, scale_factors=scale,
                kps_l=kl, mp_l=mp_l, outlier_l=outl, mps=mps)


def make_local_map_scene(oracle, seed=0, **kw):
    """TrackLocalMap-style input for SearchByProjection(F, vpMapPoints): tracked points (the mTrack* fields) built from
    the two-frame scene; points compete for keypoints (duplicates), some are bad / out of view / far."""
    sc = make_scene(oracle, seed, **kw)
    rng = np.random.default_rng(seed + 100)
    mps, K = sc["mps"], sc["K"]
    R, t = synth.Rt_from_se3(sc["Tcw"])
    Xc = (mps["pos"] @ R.T + t).astype(np.float32)
    n = len(mps)
    pts = np.zeros(n, oracle.TRACKED_POINT_DTYPE)
    pts["proj_x"] = K[0] * Xc[:, 0] / Xc[:, 2] + K[2]
    pts["proj_y"] = K[1] * Xc[:, 1] / Xc[:, 2] + K[3]
    pts["depth"] = np.linalg.norm(Xc, axis=1)
    pts["view_cos"] = np.where(rng.random(n) < 0.5, 0.9995, rng.uniform(0.5, 0.998, n)).astype(np.float32)
    pts["level"] = np.clip(sc["kps_l"]["octave"] + rng.integers(0, 2, n), 0, 7)
    inside = (Xc[:, 2] > 0) & (pts["proj_x"] >= 0) & (pts["proj_x"] < 640) & (pts["proj_y"] >= 0) & (pts["proj_y"] < 480)
    pts["in_view"] = (inside & (rng.random(n) < 0.9)).astype(np.uint8)
    pts["bad"] = (rng.random(n) < 0.03).astype(np.uint8)
    pts["desc"] = mps["desc"]
    pts["n_obs"] = mps["n_obs"]
    claimed_obs = (rng.random(len(sc["kps_c"])) < 0.5).astype(np.uint8)
    return dict(kps=sc["kps_c"], desc=sc["desc_c"], mp=sc["mp_c"], claimed_obs=claimed_obs, bounds=sc["bounds"],
                scale_factors=sc["scale_factors"], pts=pts)


def make_sim3_scene(seed=0, n=200, outlier_frac=0.3, scale=1.7, noise=0.002):
    """Two keyframes seeing the same map points up to a similarity (s, R, t) + noise + gross outliers.
    Returns P1c, P2c (float32 [n,3]), max errors, intrinsics and the ground truth."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    P2 = np.column_stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(3, 9, n)])
    R = Rotation.from_rotvec(rng.normal(0, 0.25, 3)).as_matrix()
    t = rng.normal(0, 0.3, 3)
    P1 = scale * (P2 @ R.T) + t + rng.normal(0, noise, (n, 3))
    bad = rng.random(n) < outlier_frac
    P1[bad] += rng.normal(0, 1.0, (int(bad.sum()), 3))
    P1[:, 2] = np.maximum(P1[:, 2], 0.5)
    K = np.array([500.0, 500.0, 320.0, 240.0], np.float32)
    sigma2 = (1.2 ** (2 * rng.integers(0, 8, n))).astype(np.float32)
    max_err = np.floor(9.210 * sigma2).astype(np.float32)      # vector<size_t> in the reference
    return dict(P1c=P1.astype(np.float32), P2c=P2.astype(np.float32), max_err1=max_err, max_err2=max_err.copy(), K1=K, K2=K.copy()), \
        dict(s=scale, R=R, t=t, bad=bad)


def _rot(rng, sigma):
    from scipy.spatial.transform import Rotation
    return Rotation.from_rotvec(rng.normal(0, sigma, 3)).as_matrix().astype(np.float32)


def _flip(rng, d, nbits):
    d = d.copy()
    for r in range(len(d)):
        for b in rng.choice(256, nbits, replace=False):
            d[r, b >> 3] ^= np.uint8(1 << (b & 7))
    return d


def make_kf_pair_scene(oracle, seed=0, n_pts=900, n_clutter=250, n_nodes=120, mapped_frac=0.5, flip_bits=12, dup_frac=0.1):
    """Two keyframes looking at the same 3-D points (+ clutter): the input of SearchByBoW x2, SearchForTriangulation, Fuse x2
    and SearchByProjection(KF, Scw).  Keypoints of a 3-D point share a vocabulary node in both views (with a few defects),
    descriptors differ by `flip_bits` bits, duplicates make several features compete for one candidate (claims)."""
    rng = np.random.default_rng(seed)
    K = np.array([500.0, 500.0, 320.0, 240.0], np.float32)
    bounds = np.array([0.0, 640.0, 0.0, 480.0], np.float32)
    L = 8
    scale = (np.float32(1.2) ** np.arange(L)).astype(np.float32)
    sigma2 = (scale * scale).astype(np.float32)
    inv_sigma2 = (np.float32(1.0) / sigma2).astype(np.float32)
    X = np.column_stack([rng.uniform(-6, 6, n_pts), rng.uniform(-4, 4, n_pts), rng.uniform(4, 14, n_pts)]).astype(np.float32)
    base = rng.integers(0, 256, (n_pts, 32), dtype=np.uint8)
    ndup = int(dup_frac * n_pts)                      # near-identical descriptors at nearby places
    src = rng.choice(n_pts, ndup, replace=False); dst = rng.choice(np.setdiff1d(np.arange(n_pts), src), ndup, replace=False)
    base[dst] = _flip(rng, base[src], 3)
    X[dst] = X[src] + rng.normal(0, 0.02, (ndup, 3)).astype(np.float32)
    node_of_pt = rng.integers(0, n_nodes, n_pts) * 7 + 3      # sparse, ascending-sortable node ids
    node_of_pt[dst] = node_of_pt[src]
    pt_id = np.arange(n_pts, dtype=np.int32) + 1000             # map point "pointers"
    mapped = rng.random(n_pts) < mapped_frac
    octv = rng.integers(0, L, n_pts)
    kfs = []
    for v in range(2):
        R = _rot(rng, 0.03 if v else 0.0)
        t = (rng.normal(0, 0.4, 3) if v else np.zeros(3)).astype(np.float32)
        Xc = X @ R.T + t
        u = K[0] * Xc[:, 0] / Xc[:, 2] + K[2] + rng.normal(0, 0.4, n_pts)
        w = K[1] * Xc[:, 1] / Xc[:, 2] + K[3] + rng.normal(0, 0.4, n_pts)
        vis = (Xc[:, 2] > 0) & (u > 5) & (u < 635) & (w > 5) & (w < 475) & (rng.random(n_pts) < 0.9)
        idx = np.nonzero(vis)[0]
        idx = idx[rng.permutation(len(idx))]
        n = len(idx) + n_clutter
        kps = np.zeros(n, oracle.KP_DTYPE)
        kps["x"][:len(idx)] = u[idx]; kps["y"][:len(idx)] = w[idx]
        kps["octave"][:len(idx)] = np.clip(octv[idx] + rng.integers(-1, 2, len(idx)) * (rng.random(len(idx)) < 0.2), 0, L - 1)
        kps["angle"][:len(idx)] = (37.0 * (idx % 9) + (15.0 if v else 0.0) + rng.normal(0, 3, len(idx))) % 360
        kps["x"][len(idx):] = rng.uniform(5, 635, n_clutter); kps["y"][len(idx):] = rng.uniform(5, 475, n_clutter)
        kps["octave"][len(idx):] = rng.integers(0, L, n_clutter); kps["angle"][len(idx):] = rng.uniform(0, 360, n_clutter)
        desc = np.concatenate([_flip(rng, base[idx], flip_bits), rng.integers(0, 256, (n_clutter, 32), dtype=np.uint8)])
        node = np.concatenate([np.where(rng.random(len(idx)) < 0.92, node_of_pt[idx], rng.integers(0, n_nodes, len(idx)) * 7 + 3),
                               rng.integers(0, n_nodes, n_clutter) * 7 + 3])
        mp = np.full(n, -1, np.int32)
        mp[:len(idx)] = np.where(mapped[idx], pt_id[idx], -1)
        bad = ((mp >= 0) & (rng.random(n) < 0.04)).astype(np.uint8)
        order = np.argsort(node, kind="stable")
        nodes, counts = np.unique(node, return_counts=True)
        fv = dict(fv_nodes=nodes.astype(np.int32), fv_off=np.concatenate([[0], np.cumsum(counts)]).astype(np.int32),
                  fv_feat=order.astype(np.int32))
        Ow = (-(R.T @ t)).astype(np.float32)
        kfs.append(dict(kps=kps, desc=desc, mp=mp, bad=bad, fv=fv, Tcw=synth.se3_from_Rt(R, t), Rcw=R.reshape(-1).copy(), tcw=t, Ow=Ow, K=K, bounds=bounds,
                        scale_factors=scale, level_sigma2=sigma2, inv_level_sigma2=inv_sigma2,
                        log_scale_factor=float(np.log(np.float32(1.2))), pt_of_kp=np.concatenate([idx, np.full(n_clutter, -1)])))
    # map points for the projection searches: the 3-D points + normals / scale ranges as MapPoint::UpdateNormalAndDepth leaves them
    d = np.linalg.norm(X, axis=1).astype(np.float32)
    lvl = octv
    pts = dict(id=pt_id.copy(), pos=X, normal=(-X / d[:, None] + rng.normal(0, 0.1, X.shape)).astype(np.float32),
               max_dist=(d * scale[lvl]).astype(np.float32), min_dist=(d * scale[lvl] / scale[L - 1]).astype(np.float32),
               desc=_flip(rng, base, flip_bits // 2), bad=(rng.random(n_pts) < 0.03).astype(np.uint8))
    pts["normal"] = (-pts["normal"]).astype(np.float32)   # PO . Pn must be positive for visible points: normal points away from camera 0
    pts["normal"] /= np.linalg.norm(pts["normal"], axis=1, keepdims=True)
    # a few degenerate ones: behind the camera, far outside the scale range, grazing normals
    pts["max_dist"][rng.choice(n_pts, 15, replace=False)] *= 0.2
    pts["normal"][rng.choice(n_pts, 15, replace=False)] *= -1
    return dict(kf=kfs, pts=pts, K=K, bounds=bounds, scale_factors=scale)


def make_init_scene(oracle, seed=0, n=1400, shift=6.0, flip_bits=14):
    """Two frames for SearchForInitialization: mostly level-0 keypoints, frame 2 = frame 1 moved by a few pixels; near-duplicate
    descriptors so that several F1 keypoints fight for one F2 keypoint (vnMatches21 / vMatchedDistance bookkeeping)."""
    rng = np.random.default_rng(seed)
    bounds = np.array([0.0, 640.0, 0.0, 480.0], np.float32)
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    k1 = np.zeros(n, oracle.KP_DTYPE)
    k1["x"] = rng.uniform(10, 630, n); k1["y"] = rng.uniform(10, 470, n)
    k1["octave"] = np.where(rng.random(n) < 0.75, 0, rng.integers(1, 8, n))
    k1["angle"] = rng.uniform(0, 360, n)
    d1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ndup = n // 6
    src = rng.choice(n, ndup, replace=False); dst = rng.choice(np.setdiff1d(np.arange(n), src), ndup, replace=False)
    d1[dst] = _flip(rng, d1[src], 4)
    k1["x"][dst] = k1["x"][src] + rng.normal(0, 8, ndup); k1["y"][dst] = k1["y"][src] + rng.normal(0, 8, ndup)
    keep = rng.random(n) < 0.85
    perm = rng.permutation(int(keep.sum()))
    k2 = k1[keep][perm].copy()
    k2["x"] += shift + rng.normal(0, 1.0, len(k2)); k2["y"] += -shift / 2 + rng.normal(0, 1.0, len(k2))
    k2["angle"] = (k2["angle"] + 10 + rng.normal(0, 4, len(k2))) % 360
    wild = rng.random(len(k2)) < 0.1
    k2["angle"][wild] = rng.uniform(0, 360, int(wild.sum()))
    d2 = _flip(rng, d1[keep][perm], flip_bits)
    prev = np.column_stack([k1["x"], k1["y"]]).astype(np.float32)
    return dict(k1=k1, d1=d1, k2=k2, d2=d2, bounds=bounds, scale_factors=scale, prev_matched=prev)
