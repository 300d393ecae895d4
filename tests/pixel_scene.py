"""A textured plane seen by a moving pinhole camera, rendered to 640x480 u8 frames: the config-2 chain FROM PIXELS
(ORB extraction -> SearchByProjection(Cur, Last) -> PoseOptimization -> local BA) needs images whose keypoints are consistent with
a 3-D scene.  The plane z = Z0 carries synth.base_texture; a frame is the plane-induced homography of the camera's pose applied to
the texture (synth._warp, bilinear)."""
import numpy as np
from scipy.spatial.transform import Rotation

from dvm_slam_amd import synth

K = np.array([500.0, 500.0, 320.0, 240.0])
Z0, M_PER_PX = 6.0, 0.012     # 1 texture pixel = 1 image pixel at the start pose
TEX_H, TEX_W = 960, 1280


def pose_at(t):
    """World -> camera (R, t) of frame t: a slow dolly past the plane with a little roll and yaw."""
    R = Rotation.from_rotvec([0.0015 * t, -0.0025 * t, 0.004 * t]).as_matrix()
    tt = np.array([-0.035 * t, 0.012 * t, -0.02 * t])
    return R, tt


def _tex_to_plane():
    return np.array([[M_PER_PX, 0, -M_PER_PX * TEX_W / 2], [0, M_PER_PX, -M_PER_PX * TEX_H / 2], [0, 0, 1.0]])


def homography_pix_to_tex(R, t):
    Kmat = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1.0]])
    G = np.column_stack([R[:, 0], R[:, 1], Z0 * R[:, 2] + t])        # plane coordinates (a, b, 1) -> camera
    return np.linalg.inv(Kmat @ G @ _tex_to_plane())


def render(n_frames, seed=synth.SEED_FRAMES):
    tex = synth.base_texture(seed, h=TEX_H, w=TEX_W)
    frames, poses = [], []
    for t in range(n_frames):
        R, tt = pose_at(t)
        frames.append(synth._warp(tex, homography_pix_to_tex(R, tt), 480, 640))
        poses.append((R, tt))
    return np.stack(frames), poses


def backproject(kps, R, t):
    """Keypoints of a frame with pose (R, t) -> their points on the plane, world coordinates [n, 3]."""
    rays = np.column_stack([(kps["x"] - K[2]) / K[0], (kps["y"] - K[3]) / K[1], np.ones(len(kps))])
    rw = rays @ R                       # R^T ray
    Ow = -R.T @ t
    s = (Z0 - Ow[2]) / rw[:, 2]
    return Ow[None, :] + s[:, None] * rw


def pose7(R, t):
    q = Rotation.from_matrix(R).as_quat()
    if q[3] < 0:
        q = -q
    return np.concatenate([t, q])
