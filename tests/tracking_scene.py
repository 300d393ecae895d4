"""A camera moving past a cloud of map points: per-frame keypoints / descriptors + an imperfect map (config-2 chain test)."""
import numpy as np
from scipy.spatial.transform import Rotation

from matcher_scene import _flip


def make_sequence(kp_dtype, seed=0, n_frames=16, n_pts=900, n_clutter=150):
    rng = np.random.default_rng(seed)
    K = np.array([500.0, 500.0, 320.0, 240.0])
    X = np.column_stack([rng.uniform(-8, 8, n_pts), rng.uniform(-5, 5, n_pts), rng.uniform(5, 16, n_pts)])
    base = rng.integers(0, 256, (n_pts, 32), dtype=np.uint8)
    octv = rng.integers(0, 8, n_pts)
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    frames, gt = [], []
    for t in range(n_frames):
        R = Rotation.from_rotvec([0.004 * t, -0.01 * t, 0.002 * t]).as_matrix()
        tt = np.array([-0.12 * t, 0.02 * t, 0.05 * t])
        Xc = X @ R.T + tt
        u = K[0] * Xc[:, 0] / Xc[:, 2] + K[2]; v = K[1] * Xc[:, 1] / Xc[:, 2] + K[3]
        vis = np.flatnonzero((Xc[:, 2] > 0) & (u > 8) & (u < 632) & (v > 8) & (v < 472) & (rng.random(n_pts) < 0.93))
        vis = vis[rng.permutation(len(vis))]
        n = len(vis) + n_clutter
        kps = np.zeros(n, kp_dtype)
        noise = 0.35 * scale[octv[vis]]
        kps["x"][:len(vis)] = u[vis] + rng.normal(0, 1, len(vis)) * noise; kps["y"][:len(vis)] = v[vis] + rng.normal(0, 1, len(vis)) * noise
        kps["octave"][:len(vis)] = octv[vis]; kps["angle"][:len(vis)] = (29.0 * (vis % 11) + 1.5 * t + rng.normal(0, 2, len(vis))) % 360
        kps["x"][len(vis):] = rng.uniform(8, 632, n_clutter); kps["y"][len(vis):] = rng.uniform(8, 472, n_clutter)
        kps["octave"][len(vis):] = rng.integers(0, 8, n_clutter); kps["angle"][len(vis):] = rng.uniform(0, 360, n_clutter)
        desc = np.concatenate([_flip(rng, base[vis], 9), rng.integers(0, 256, (n_clutter, 32), dtype=np.uint8)])
        frames.append(dict(kps=kps, desc=desc, pt=np.concatenate([vis, np.full(n_clutter, -1)])))
        gt.append((R, tt))
    d = np.linalg.norm(X, axis=1)
    map_points = dict(pos=X + rng.normal(0, 0.03, X.shape), desc=base, n_obs=np.full(n_pts, 3, np.int32),   # an imperfect map
                      normal=(X / d[:, None]).astype(np.float32), max_dist=(d * scale[octv]).astype(np.float32),
                      min_dist=(d * scale[octv] / scale[7]).astype(np.float32))
    mp0 = np.where(rng.random(len(frames[0]["pt"])) < 0.9, frames[0]["pt"], -1).astype(np.int32)
    return dict(frames=frames, map_points=map_points, K=K, bounds=np.array([0.0, 640.0, 0.0, 480.0], np.float32), scale=scale,
                inv_sigma2=(1.0 / (scale * scale)).astype(np.float32), mp0=mp0, gt=gt)
