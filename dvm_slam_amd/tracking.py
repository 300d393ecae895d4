"""Single-agent tracking + local BA chain (BASELINE.json config 2: "Single agent tracking + local BA, 640x480 8-level ORB,
1000 features/frame"), composed from the accelerated pieces in the order the reference's Tracking / LocalMapping threads
run them (reference src/Tracking.cc TrackWithMotionModel: UpdateLastFrame -> ORBmatcher::SearchByProjection(Cur, Last, th)
[th doubled when fewer than 20 matches] -> Optimizer::PoseOptimization -> outlier matches dropped;
src/LocalMapping.cc -> Optimizer::LocalBundleAdjustment over the recent keyframes, the oldest ones fixed):

  per frame   constant-velocity pose prediction -> SearchByProjection(CurrentFrame, LastFrame) -> PoseOptimization
              TrackLocalMap: Frame::isInFrustum over the map -> SearchByProjection(F, vpMapPoints) -> PoseOptimization
  every k-th  local BA over the last W frames (first two fixed), poses and landmarks written back

The ORB extraction in front of it is the bench's hot path and has its own parity tests; here the frames arrive as
keypoints + descriptors.  Steps are injected (`ops`) so that the same composition runs over the HIP library (GpuOps) and,
in the tests, over the CPU oracle."""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation

from . import synth


class GpuOps:
    def __init__(self, device=0):
        from . import capi
        self.capi, self.device = capi, device

    def search_by_projection(self, cur, last, mps, Tcw, K, bounds, scale, th):
        n, mp, _ = self.capi.search_by_projection_frames(cur["kps"], cur["desc"], cur["mp"], Tcw, K, bounds, scale, last["kps"], last["mp"],
                                                         last.get("outlier"), mps, th, True, self.device)
        return n, mp

    def pose_optimize(self, pose, Xw, obs, w, K):
        p, out, nin = self.capi.pose_optimize(pose[None], Xw[None], obs[None], w[None], [len(Xw)], K, self.device)
        return p[0], out[0], int(nin[0])

    def frustum_frame(self):
        return self.capi.FrustumFrame()

    def pose_matrices(self, Tcw):
        return self.capi.pose_matrices(Tcw)

    def is_in_frustum(self, F, P, normal, dmin, dmax):
        return self.capi.is_in_frustum(F, P, normal, dmin, dmax, 0.5)

    def tracked_dtype(self):
        return self.capi.TRACKED_POINT_DTYPE

    def search_local_points(self, cur, claimed, bounds, scale, pts, th, nnratio):
        n, mp, _ = self.capi.search_by_projection_points(cur["kps"], cur["desc"], cur["mp"], claimed, bounds, scale, pts, th, nnratio, False, 0.0, self.device)
        return n, mp

    def local_ba(self, poses, fixed, points, edges, K, delta, iters):
        e = self.capi.make_edges(*edges)
        ba = self.capi.BundleAdjuster(self.device)
        ba.set_problem(poses, fixed, points, e, K, delta)
        st = ba.optimize(iters)
        p, x = ba.result()
        ba.close()
        return p, x, st["iterations"]


def pose7(R, t):
    q = Rotation.from_matrix(R).as_quat()
    if q[3] < 0:
        q = -q
    return np.concatenate([t, q])


def rt_of(p):
    return Rotation.from_quat(p[3:7]).as_matrix(), p[:3]


def track(ops, frames, map_points, mp_dtype, K, bounds, scale, inv_sigma2, pose0, mp0, lba_every=5, window=6, th=15.0):
    """frames[t]: dict(kps, desc); map_points: dict(pos [M,3] float64 (the map, refined by the local BA), desc [M,32], n_obs [M]).
    pose0 / mp0: pose (7,) and keypoint -> map point assignment of frame 0.  Returns the per-frame poses, assignments,
    match counts and the local-BA results."""
    poses = [np.asarray(pose0, np.float64)]
    assign = [np.asarray(mp0, np.int32)]
    log = dict(nmatch=[], ninl=[], nlocal=[], lba=[])
    X = np.array(map_points["pos"], np.float64, copy=True)
    for t in range(1, len(frames)):
        # constant-velocity prediction (Tracking.cc: mVelocity * mLastFrame.GetPose())
        Rl, tl = rt_of(poses[-1])
        if t >= 2:
            Rp, tp = rt_of(poses[-2])
            Rv = Rl @ Rp.T; tv = tl - Rv @ tp
            Rc, tc = Rv @ Rl, Rv @ tl + tv
        else:
            Rc, tc = Rl, tl
        mps = np.zeros(len(X), mp_dtype)
        mps["pos"], mps["desc"], mps["n_obs"] = X.astype(np.float32), map_points["desc"], map_points["n_obs"]
        cur = dict(frames[t], mp=np.full(len(frames[t]["kps"]), -1, np.int32))
        last = dict(frames[t - 1], mp=assign[-1])
        Tcw = synth.se3_from_Rt(Rc, tc)   # mCurrentFrame.SetPose(mVelocity * mLastFrame.GetPose()): a Sophus::SE3f
        n, mp = ops.search_by_projection(cur, last, mps, Tcw, K, bounds, scale, th)
        if n < 20:   # Tracking.cc: retry with a wider window
            n, mp = ops.search_by_projection(cur, last, mps, Tcw, K, bounds, scale, 2 * th)
        sel = np.flatnonzero(mp >= 0)
        kp = frames[t]["kps"]
        obs = np.stack([kp["x"][sel], kp["y"][sel]], 1).astype(np.float64)
        pose, outl, nin = ops.pose_optimize(pose7(Rc, tc), X[mp[sel]], obs, inv_sigma2[kp["octave"][sel]].astype(np.float64), K)
        mp = mp.copy()
        mp[sel[outl[:len(sel)].astype(bool)]] = -1          # outlier matches are dropped (Tracking.cc: mvpMapPoints[i] = NULL)
        # ---- TrackLocalMap (Tracking.cc SearchLocalPoints + TrackLocalMap): project the map, match what is not matched yet
        Rn, tn = rt_of(pose)
        F = ops.frustum_frame()
        mRcw, mtcw, mOw = ops.pose_matrices(synth.se3_from_Rt(Rn, tn))   # Frame::SetPose -> UpdatePoseMatrices
        F.Rcw[:] = mRcw.reshape(-1).tolist(); F.tcw[:] = mtcw.tolist(); F.Ow[:] = mOw.tolist()
        F.fx, F.fy, F.cx, F.cy = (float(v) for v in K)
        F.min_x, F.max_x, F.min_y, F.max_y = (float(v) for v in bounds)
        F.bf, F.log_scale_factor, F.n_levels = 0.0, float(np.log(np.float32(1.2))), len(scale)
        tp = ops.is_in_frustum(F, X.astype(np.float32), map_points["normal"], map_points["min_dist"], map_points["max_dist"])
        pts = np.zeros(len(X), ops.tracked_dtype())
        for a, b in (("proj_x", "proj_x"), ("proj_y", "proj_y"), ("depth", "depth"), ("view_cos", "view_cos"), ("level", "level")):
            pts[a] = tp[b]
        already = np.zeros(len(X), bool); already[mp[mp >= 0]] = True
        pts["in_view"] = (tp["in_view"] != 0) & ~already          # points already in the frame: mbTrackInView = false
        pts["desc"], pts["n_obs"] = map_points["desc"], map_points["n_obs"]
        cur = dict(frames[t], mp=mp)
        claimed = ((mp >= 0) & (map_points["n_obs"][np.maximum(mp, 0)] > 0)).astype(np.uint8)
        n_local, mp = ops.search_local_points(cur, claimed, bounds, scale, pts, 1.0, 0.8)
        sel = np.flatnonzero(mp >= 0)
        obs = np.stack([kp["x"][sel], kp["y"][sel]], 1).astype(np.float64)
        pose, outl, nin = ops.pose_optimize(pose, X[mp[sel]], obs, inv_sigma2[kp["octave"][sel]].astype(np.float64), K)
        mp = mp.copy()
        mp[sel[outl[:len(sel)].astype(bool)]] = -1
        poses.append(pose); assign.append(mp)
        log["nmatch"].append(n); log["ninl"].append(nin); log["nlocal"].append(n_local)
        if t % lba_every == 0 and t + 1 >= window:
            ids = list(range(t + 1 - window, t + 1))
            ep, ept, eo, ew = [], [], [], []
            for j, f in enumerate(ids):
                s = np.flatnonzero(assign[f] >= 0)
                ep.append(np.full(len(s), j)); ept.append(assign[f][s])
                eo.append(np.stack([frames[f]["kps"]["x"][s], frames[f]["kps"]["y"][s]], 1)); ew.append(inv_sigma2[frames[f]["kps"]["octave"][s]])
            ep, ept, eo, ew = np.concatenate(ep), np.concatenate(ept), np.concatenate(eo).astype(np.float64), np.concatenate(ew).astype(np.float64)
            used, local = np.unique(ept, return_inverse=True)     # only the landmarks seen in the window
            fixed = np.zeros(window, np.uint8); fixed[:2] = 1
            P = np.stack([poses[f] for f in ids])
            Pn, Xn, iters = ops.local_ba(P, fixed, X[used], (ep.astype(np.int32), local.astype(np.int32), eo, ew), K, float(np.sqrt(5.991)), 10)
            for j, f in enumerate(ids):
                poses[f] = Pn[j]
            X[used] = Xn
            log["lba"].append(dict(frame=t, poses=Pn.copy(), n_points=len(used), n_edges=len(ep), iterations=iters))
    return dict(poses=np.stack(poses), assign=assign, X=X, **log)
