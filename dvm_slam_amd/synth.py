"""Deterministic synthetic inputs for the hot path.

* `frame_stream`  -- 640x480 8-bit mono stream, seed 0x5EED: a 1280x960 base texture (3 octaves of
  value noise + 2400 random axis-aligned / rotated rectangles) viewed through a smoothly moving
  homography (<= 8 px/frame), so consecutive frames are matchable and every pyramid level holds
  far more than its quota of FAST corners.
* `ba_problem`    -- bundle-adjustment problem, seed 0xBA5E: cameras on a closed loop looking inward,
  landmarks in a shell, k observations each, pixel noise + gross outliers, perturbed initial state.

Only numpy; no reference code involved (the reference ships no synthetic generator: its inputs are
rosbags / Webots, src/webots_sim/worlds/webots.yaml for the 640x480 pinhole intrinsics).
"""
from __future__ import annotations

import numpy as np

SEED_FRAMES = 0x5EED
SEED_BA = 0xBA5E
# src/webots_sim/worlds/webots.yaml:23-34 (pinhole, zero distortion)
FX = FY = 149.0
CX, CY = 320.0, 240.0
WIDTH, HEIGHT = 640, 480


def _value_noise(rng: np.random.Generator, h: int, w: int, cell: int) -> np.ndarray:
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.random((gh, gw))
    ys = np.arange(h) / cell
    xs = np.arange(w) / cell
    y0 = ys.astype(np.int64)
    x0 = xs.astype(np.int64)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def base_texture(seed: int = SEED_FRAMES, h: int = 960, w: int = 1280, nrect: int = 2400) -> np.ndarray:
    rng = np.random.default_rng(seed)
    tex = 0.5 * _value_noise(rng, h, w, 64) + 0.3 * _value_noise(rng, h, w, 16) + 0.2 * _value_noise(rng, h, w, 4)
    tex = 40.0 + 150.0 * tex
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(nrect):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        hw, hh = rng.uniform(3, 40), rng.uniform(3, 40)
        ang = rng.uniform(0, np.pi) if rng.random() < 0.5 else 0.0
        val = rng.uniform(0, 255)
        r = int(np.ceil(np.hypot(hw, hh))) + 1
        x0, x1 = max(int(cx) - r, 0), min(int(cx) + r + 1, w)
        y0, y1 = max(int(cy) - r, 0), min(int(cy) + r + 1, h)
        if x0 >= x1 or y0 >= y1:
            continue
        dx = xx[y0:y1, x0:x1] - cx
        dy = yy[y0:y1, x0:x1] - cy
        u = dx * np.cos(ang) + dy * np.sin(ang)
        v = -dx * np.sin(ang) + dy * np.cos(ang)
        m = (np.abs(u) <= hw) & (np.abs(v) <= hh)
        tex[y0:y1, x0:x1][m] = val
    return np.clip(np.rint(tex), 0, 255).astype(np.uint8)


def _warp(tex: np.ndarray, H: np.ndarray, h: int, w: int) -> np.ndarray:
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    d = H[2, 0] * xs + H[2, 1] * ys + H[2, 2]
    sx = (H[0, 0] * xs + H[0, 1] * ys + H[0, 2]) / d
    sy = (H[1, 0] * xs + H[1, 1] * ys + H[1, 2]) / d
    sx = np.clip(sx, 0, tex.shape[1] - 1.001)
    sy = np.clip(sy, 0, tex.shape[0] - 1.001)
    x0 = sx.astype(np.int64)
    y0 = sy.astype(np.int64)
    fx = sx - x0
    fy = sy - y0
    t = tex.astype(np.float64)
    v = (t[y0, x0] * (1 - fx) + t[y0, x0 + 1] * fx) * (1 - fy) + (t[y0 + 1, x0] * (1 - fx) + t[y0 + 1, x0 + 1] * fx) * fy
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def frame_homography(t: int) -> np.ndarray:
    """Camera path over the base texture: slow pan + roll + zoom + slight perspective, <= 8 px/frame."""
    a = 0.004 * t
    s = 1.0 + 0.15 * np.sin(0.011 * t)
    c, si = np.cos(a) * s, np.sin(a) * s
    tx = 320.0 + 250.0 * np.sin(0.013 * t) + 1.5 * t * 0.3
    ty = 240.0 + 180.0 * np.sin(0.009 * t + 1.0)
    # rotate/scale about the frame centre, then translate into the texture
    T0 = np.array([[1, 0, -WIDTH / 2], [0, 1, -HEIGHT / 2], [0, 0, 1.0]])
    R = np.array([[c, -si, 0], [si, c, 0], [0, 0, 1.0]])
    T1 = np.array([[1, 0, tx + WIDTH / 2], [0, 1, ty + HEIGHT / 2], [0, 0, 1.0]])
    P = np.array([[1, 0, 0], [0, 1, 0], [2e-5 * np.sin(0.02 * t), 1e-5 * np.cos(0.017 * t), 1.0]])
    return T1 @ R @ P @ T0


def low_texture(seed: int = SEED_FRAMES, h: int = 960, w: int = 1280, nrect: int = 220) -> np.ndarray:
    """The other end of the data dependence of the extractor: a weakly textured scene (walls, floor, a few objects) -- smooth shading, faint
    surface noise, a couple of hundred low-contrast patches and only a few strong ones.  Few pixels are FAST corners at iniThFAST = 20 (~1 %
    against ~8 % of base_texture), many cells have none and are re-run at minThFAST = 7 (ORBextractor.cc:664-670), levels fall short of
    their keypoint quota."""
    rng = np.random.default_rng(seed + 77)
    tex = 90.0 + 70.0 * (0.7 * _value_noise(rng, h, w, 256) + 0.3 * _value_noise(rng, h, w, 64)) + 6.0 * _value_noise(rng, h, w, 3)
    yy, xx = np.mgrid[0:h, 0:w]
    for j in range(nrect):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        hw, hh = rng.uniform(8, 90), rng.uniform(8, 90)
        ang = rng.uniform(0, np.pi) if rng.random() < 0.5 else 0.0
        strong = j % 9 == 0
        r = int(np.ceil(np.hypot(hw, hh))) + 1
        x0, x1 = max(int(cx) - r, 0), min(int(cx) + r + 1, w)
        y0, y1 = max(int(cy) - r, 0), min(int(cy) + r + 1, h)
        if x0 >= x1 or y0 >= y1:
            continue
        dx = xx[y0:y1, x0:x1] - cx
        dy = yy[y0:y1, x0:x1] - cy
        u = dx * np.cos(ang) + dy * np.sin(ang)
        v = -dx * np.sin(ang) + dy * np.cos(ang)
        m = (np.abs(u) <= hw) & (np.abs(v) <= hh)
        tex[y0:y1, x0:x1][m] += rng.uniform(-60, 60) if strong else rng.uniform(-14, 14)
    return np.clip(np.rint(tex), 0, 255).astype(np.uint8)


def frame_stream(n: int = 256, seed: int = SEED_FRAMES, h: int = HEIGHT, w: int = WIDTH, start: int = 0, texture: str = "rich") -> np.ndarray:
    """Returns uint8 [n, h, w].  texture: "rich" (the BASELINE stream: corner-rich) or "low" (low_texture: few corners, minThFAST cells)."""
    tex = base_texture(seed) if texture == "rich" else low_texture(seed)
    return np.stack([_warp(tex, frame_homography(start + t), h, w) for t in range(n)])


def small_image(seed: int, h: int, w: int) -> np.ndarray:
    """Small textured test image (value noise + rectangles), any size >= 40x40."""
    rng = np.random.default_rng(seed)
    tex = base_texture(seed, h=max(h, 64), w=max(w, 64), nrect=max(8, (h * w) // 3000))
    img = tex[:h, :w].copy()
    # sprinkle salt so FAST has work on tiny images too
    m = rng.random((h, w)) < 0.01
    img[m] = rng.integers(0, 256, size=int(m.sum()), dtype=np.uint8)
    return img


# ------------------------------------------------------------------------------------------- BA
def _rot_from_axis_angle(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def _quat_from_rot(R: np.ndarray) -> np.ndarray:
    """(x, y, z, w), w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def se3_from_Rt(R, t) -> np.ndarray:
    """A pose as the reference holds it (Sophus::SE3f): 7 float32 = unit quaternion coeffs (x, y, z, w), translation."""
    q = _quat_from_rot(np.asarray(R, np.float64).reshape(3, 3))
    return np.concatenate([q, np.asarray(t, np.float64).reshape(3)]).astype(np.float32)


def Rt_from_se3(T):
    """(R[3,3], t[3]) float64 of a 7-float pose (plain numpy; NOT the reference's float arithmetic -- test geometry only)."""
    T = np.asarray(T, np.float64).reshape(7)
    x, y, z, w = T[:4] / np.linalg.norm(T[:4])
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return R, T[4:].copy()


def sim3_from_sRt(s, R, t) -> np.ndarray:
    """Sophus::Sim3f as stored: RxSO3 quaternion (x, y, z, w) with |q|^2 = scale, translation; 7 float32."""
    q = _quat_from_rot(np.asarray(R, np.float64).reshape(3, 3)) * np.sqrt(float(s))
    return np.concatenate([q, np.asarray(t, np.float64).reshape(3)]).astype(np.float32)


def ba_problem(n_kf: int = 500, n_pts: int = 20000, k_obs: int = 8, seed: int = SEED_BA, noise_px: float = 1.0,
               outlier_frac: float = 0.05, radius: float = 50.0, laps: int = 1, long_range_frac: float = 0.0, long_range_obs: int = 3):
    """Synthetic global-BA problem.  Returns a dict of numpy arrays:

    poses  [P,7] f64  (tx,ty,tz, qx,qy,qz,qw) world->camera (Tcw), perturbed initial estimate
    poses_gt [P,7], fixed [P] u8 (KF 0 fixed), points [L,3] f64 perturbed, points_gt [L,3],
    edge_pose [E] i32, edge_point [E] i32, obs [E,2] f64, inv_sigma2 [E] f64, intrinsics (fx,fy,cx,cy).
    Every landmark is observed by k_obs consecutive keyframes that see it in-frame.
    laps > 1: the camera goes round the loop `laps` times (keyframe i and i + n_kf / laps stand at the same place) and a landmark is seen by
    k_obs / laps consecutive keyframes of EVERY lap -- the co-visibility of a map after loop closures: bands far off the diagonal.
    long_range_frac: that share of the landmarks is also observed by long_range_obs keyframes anywhere on the loop that have it in view (the
    wide lens sees across the ring): scattered long-range couplings, as a merged map has them.
    """
    rng = np.random.default_rng(seed)
    ang = 2 * np.pi * laps * np.arange(n_kf) / n_kf
    centres = np.stack([radius * np.cos(ang), radius * np.sin(ang), np.zeros(n_kf)], axis=1)
    Rcw = np.zeros((n_kf, 3, 3))
    tcw = np.zeros((n_kf, 3))
    for i in range(n_kf):
        # camera z axis points to the loop centre (inward), y axis = world -z (down), x = y cross z
        z = -centres[i] / np.linalg.norm(centres[i])
        y = np.array([0.0, 0.0, -1.0])
        x = np.cross(y, z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        Rwc = np.stack([x, y, z], axis=1)
        Rcw[i] = Rwc.T
        tcw[i] = -Rwc.T @ centres[i]
    pts = np.zeros((n_pts, 3))
    edge_pose, edge_point, obs = [], [], []
    l = 0
    tries = 0
    while l < n_pts:
        tries += 1
        if tries > 200 * n_pts:
            raise RuntimeError("could not place landmarks")
        # shell of points 8..18 m in front of a random anchor keyframe
        a = int(rng.integers(0, n_kf))
        depth = rng.uniform(8.0, 18.0)
        u = rng.uniform(60, WIDTH - 60)
        v = rng.uniform(60, HEIGHT - 60)
        pc = np.array([(u - CX) / FX * depth, (v - CY) / FY * depth, depth])
        pw = Rcw[a].T @ (pc - tcw[a])
        if laps == 1:
            first = a - k_obs // 2
            kfs = [(first + j) % n_kf for j in range(k_obs)]
        else:
            per = max(1, k_obs // laps)
            lap_len = n_kf // laps
            first = a % lap_len - per // 2
            kfs = [((first + j) % lap_len + lp * lap_len) % n_kf for lp in range(laps) for j in range(per)]
        uv = []
        ok = True
        for kf in kfs:
            q = Rcw[kf] @ pw + tcw[kf]
            if q[2] <= 0.5:
                ok = False
                break
            uu, vv = FX * q[0] / q[2] + CX, FY * q[1] / q[2] + CY
            if not (0 <= uu < WIDTH and 0 <= vv < HEIGHT):
                ok = False
                break
            uv.append((uu, vv))
        if not ok or len(set(kfs)) != len(kfs):
            continue
        pts[l] = pw
        for kf, o in zip(kfs, uv):
            edge_pose.append(kf)
            edge_point.append(l)
            obs.append(o)
        l += 1
    if long_range_frac > 0:       # (its own generator: the default problem's random stream is untouched)
        rng2 = np.random.default_rng(seed + 0x10C)
        have = {}
        for kf, l2 in zip(edge_pose, edge_point):
            have.setdefault(l2, set()).add(kf)
        for l2 in np.flatnonzero(rng2.random(n_pts) < long_range_frac):
            added = 0
            for kf in rng2.permutation(n_kf):
                if added >= long_range_obs:
                    break
                if kf in have[l2]:
                    continue
                q = Rcw[kf] @ pts[l2] + tcw[kf]
                if q[2] <= 0.5:
                    continue
                uu, vv = FX * q[0] / q[2] + CX, FY * q[1] / q[2] + CY
                if 0 <= uu < WIDTH and 0 <= vv < HEIGHT:
                    edge_pose.append(int(kf)); edge_point.append(int(l2)); obs.append((uu, vv)); have[l2].add(int(kf)); added += 1
        order = np.lexsort((np.asarray(edge_pose), np.asarray(edge_point)))      # landmark-major, as the base problem
        edge_pose = list(np.asarray(edge_pose)[order]); edge_point = list(np.asarray(edge_point)[order]); obs = list(np.asarray(obs)[order])
    edge_pose = np.asarray(edge_pose, np.int32)
    edge_point = np.asarray(edge_point, np.int32)
    obs = np.asarray(obs, np.float64)
    E = len(edge_pose)
    obs += rng.normal(0, noise_px, size=obs.shape)
    out = rng.random(E) < outlier_frac
    obs[out] += rng.choice([-1.0, 1.0], size=(int(out.sum()), 2)) * 50.0
    octave = rng.integers(0, 8, size=E)
    inv_sigma2 = (1.2 ** (-2.0 * octave)).astype(np.float64)
    poses_gt = np.zeros((n_kf, 7))
    poses = np.zeros((n_kf, 7))
    for i in range(n_kf):
        poses_gt[i, :3] = tcw[i]
        poses_gt[i, 3:] = _quat_from_rot(Rcw[i])
        if i == 0:
            poses[i] = poses_gt[i]
            continue
        dR = _rot_from_axis_angle(rng.normal(0, np.deg2rad(0.5) / np.sqrt(3), 3))
        Rn = dR @ Rcw[i]
        tn = dR @ tcw[i] + rng.normal(0, 0.02 / np.sqrt(3), 3)
        poses[i, :3] = tn
        poses[i, 3:] = _quat_from_rot(Rn)
    points = pts + rng.normal(0, 0.05 / np.sqrt(3), pts.shape)
    fixed = np.zeros(n_kf, np.uint8)
    fixed[0] = 1
    return dict(poses=poses, poses_gt=poses_gt, fixed=fixed, points=points, points_gt=pts, edge_pose=edge_pose,
                edge_point=edge_point, obs=obs, inv_sigma2=inv_sigma2, intrinsics=np.array([FX, FY, CX, CY]))


def small_window_problem(n_kf: int = 2, n_pts: int = 150, seed: int = 1, noise_px: float = 1.0, outlier_frac: float = 0.05, baseline: float = 0.15,
                         min_obs: int = 2):
    """The shape of the bundle adjustments ORB-SLAM3 runs on a FRESH monocular map: n_kf = 2 is Tracking::CreateInitialMapMonocular's
    GlobalBundleAdjustemnt(map, 20) (Tracking.cc:2330) -- keyframe 0 fixed at the identity, keyframe 1 a small baseline away, the
    points triangulated from the two views at a median depth of ~1 (the map is rescaled so) --; n_kf = 3..6 are the local windows of the
    keyframes inserted right after (Optimizer.cc:1030-1387).  Cameras look down +z from a line along x; every point is observed by a
    random subset of >= min_obs cameras.  Same keys as ba_problem()."""
    rng = np.random.default_rng(seed)
    centres = np.stack([baseline * np.arange(n_kf) + rng.normal(0, 0.02 * baseline, n_kf), rng.normal(0, 0.1 * baseline, n_kf),
                        rng.normal(0, 0.1 * baseline, n_kf)], axis=1)
    centres[0] = 0
    Rcw, tcw = [], []
    for i in range(n_kf):
        R = np.eye(3) if i == 0 else _rot_from_axis_angle(rng.normal(0, np.deg2rad(2.0), 3))
        Rcw.append(R); tcw.append(-R @ centres[i])
    pts, edge_pose, edge_point, obs = [], [], [], []
    tries = 0
    while len(pts) < n_pts:
        tries += 1
        if tries > 400 * n_pts:
            raise RuntimeError("could not place landmarks")
        depth = rng.uniform(0.5, 2.0)
        pw = np.array([(rng.uniform(40, WIDTH - 40) - CX) / FX * depth, (rng.uniform(40, HEIGHT - 40) - CY) / FY * depth, depth])
        seen = []
        for i in range(n_kf):
            q = Rcw[i] @ pw + tcw[i]
            if q[2] > 0.1:
                u, v = FX * q[0] / q[2] + CX, FY * q[1] / q[2] + CY
                if 0 <= u < WIDTH and 0 <= v < HEIGHT:
                    seen.append((i, u, v))
        if len(seen) < min_obs:
            continue
        k = int(rng.integers(min_obs, len(seen) + 1))
        pick = sorted(rng.choice(len(seen), size=k, replace=False))
        for j in pick:
            edge_pose.append(seen[j][0]); edge_point.append(len(pts)); obs.append(seen[j][1:])
        pts.append(pw)
    pts = np.asarray(pts)
    edge_pose = np.asarray(edge_pose, np.int32); edge_point = np.asarray(edge_point, np.int32)
    obs = np.asarray(obs, np.float64)
    E = len(edge_pose)
    obs += rng.normal(0, noise_px, obs.shape)
    out = rng.random(E) < outlier_frac
    obs[out] += rng.choice([-1.0, 1.0], size=(int(out.sum()), 2)) * 30.0
    inv_sigma2 = (1.2 ** (-2.0 * rng.integers(0, 8, size=E))).astype(np.float64)
    poses_gt = np.zeros((n_kf, 7)); poses = np.zeros((n_kf, 7))
    for i in range(n_kf):
        poses_gt[i, :3] = tcw[i]; poses_gt[i, 3:] = _quat_from_rot(Rcw[i])
        if i == 0:
            poses[i] = poses_gt[i]
            continue
        dR = _rot_from_axis_angle(rng.normal(0, np.deg2rad(0.3), 3))
        poses[i, :3] = dR @ tcw[i] + rng.normal(0, 0.03 * baseline, 3); poses[i, 3:] = _quat_from_rot(dR @ Rcw[i])
    points = pts * (1.0 + rng.normal(0, 0.03, (len(pts), 1)))        # triangulation error is mostly along the ray
    fixed = np.zeros(n_kf, np.uint8); fixed[0] = 1
    return dict(poses=poses, poses_gt=poses_gt, fixed=fixed, points=points, points_gt=pts, edge_pose=edge_pose, edge_point=edge_point, obs=obs,
                inv_sigma2=inv_sigma2, intrinsics=np.array([FX, FY, CX, CY]))


def vocabulary(k: int = 10, L: int = 3, seed: int = 0x0B0C, ragged: bool = True, stop_frac: float = 0.02):
    """Synthetic DBoW2 vocabulary tree (ORBvoc.txt is not shipped with the reference: .MISSING_LARGE_BLOBS).
    Breadth-first node ids, root = 0, depth L, fan-out k (6..k when `ragged`), random 256-bit node descriptors,
    idf-like leaf weights (a few exactly 0 = stopped words).  Returns CSR children lists like dvm_vocab_create."""
    rng = np.random.default_rng(seed)
    child_off, children, level = [0], [], [0]
    frontier, n = [0], 1
    off_of = {}
    for d in range(L):
        nxt = []
        for node in frontier:
            kk = int(rng.integers(max(2, k - 4), k + 1)) if ragged else k
            off_of[node] = list(range(n, n + kk))
            nxt.extend(off_of[node]); level.extend([d + 1] * kk); n += kk
        frontier = nxt
    for node in range(n):
        ch = off_of.get(node, [])
        children.extend(ch)
        child_off.append(len(children))
    level = np.array(level)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    word_id = np.full(n, -1, np.int32)
    leaves = np.flatnonzero(level == L)
    word_id[leaves] = np.arange(len(leaves), dtype=np.int32)
    weight = np.zeros(n, np.float64)
    weight[leaves] = np.log(rng.uniform(1.5, 400.0, len(leaves)))
    weight[leaves[rng.random(len(leaves)) < stop_frac]] = 0.0
    return dict(n_nodes=n, child_off=np.array(child_off, np.int32), children=np.array(children, np.int32), desc=desc,
                weight=weight, word_id=word_id, L=L)


def _sim3_mul(a, b):
    from scipy.spatial.transform import Rotation as Rot
    ra, rb = Rot.from_quat(a[:4]), Rot.from_quat(b[:4])
    q = (ra * rb).as_quat()
    if q[3] < 0:
        q = -q
    return np.concatenate([q, a[7] * ra.apply(b[4:7]) + a[4:7], [a[7] * b[7]]])


def _sim3_inv(a):
    from scipy.spatial.transform import Rotation as Rot
    ri = Rot.from_quat(a[:4]).inv()
    q = ri.as_quat()
    if q[3] < 0:
        q = -q
    return np.concatenate([q, ri.apply(-a[4:7] / a[7]), [1.0 / a[7]]])


def pose_graph(n: int = 60, seed: int = 0xE55E, drift: float = 0.02, extra: int = 2, noise: float = 0.0):
    """Essential-graph style Sim3 pose graph: keyframes on a loop; spanning-tree edges (i, i-1), `extra` covisibility edges
    per keyframe, one loop-closure edge (n-1, 0).  Measurements Sji = Sjw * Swi come from the ground truth (optionally
    noisy); the initial estimates carry an accumulated drift in rotation, translation and scale.  Vertex 0 is fixed.
    Returns dict(S0 [n,8] (q_xyzw, t, s), S_gt, fixed, edges_v [E,2] (i, j), edges_meas [E,8])."""
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(seed)
    gt = np.zeros((n, 8))
    for i in range(n):
        a = 2 * np.pi * i / n
        Rwc = Rot.from_euler("zyx", [a + np.pi / 2, 0.1 * np.sin(3 * a), 0.05 * np.cos(2 * a)])
        twc = np.array([10 * np.cos(a), 10 * np.sin(a), 0.3 * np.sin(4 * a)])
        Rcw = Rwc.inv()
        q = Rcw.as_quat()
        if q[3] < 0:
            q = -q
        gt[i] = np.concatenate([q, -Rcw.apply(twc), [1.0]])
    S0 = gt.copy()
    acc = np.array([0, 0, 0, 1, 0, 0, 0, 1.0])
    for i in range(1, n):
        d = np.concatenate([Rot.from_rotvec(rng.normal(0, drift, 3)).as_quat(), rng.normal(0, drift * 5, 3), [np.exp(rng.normal(0, drift))]])
        acc = _sim3_mul(d, acc)
        S0[i] = _sim3_mul(acc, gt[i])
    ev, em = [], []

    def add(i, j):
        m = _sim3_mul(gt[j], _sim3_inv(gt[i]))
        if noise > 0:
            d = np.concatenate([Rot.from_rotvec(rng.normal(0, noise, 3)).as_quat(), rng.normal(0, noise, 3), [np.exp(rng.normal(0, noise))]])
            m = _sim3_mul(d, m)
        ev.append((i, j)); em.append(m)

    for i in range(1, n):
        add(i, i - 1)
        for k in range(extra):
            j = i - 2 - int(rng.integers(0, 4))
            if j >= 0:
                add(i, j)
    add(n - 1, 0)
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    return dict(S0=S0, S_gt=gt, fixed=fixed, edges_v=np.array(ev, np.int32), edges_meas=np.array(em, np.float64))
