"""ctypes binding of oracle/liboracle.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package (dvm_slam_amd).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib(path: str | None = None):
    global _LIB
    if _LIB is None or path is not None:
        L = C.CDLL(path or build())
        vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
        L.orc_orb_create.restype = vp
        L.orc_orb_create.argtypes = [C.POINTER(OrbParams)]
        L.orc_orb_destroy.argtypes = [vp]
        L.orc_orb_tables.argtypes = [vp] * 7
        L.orc_orb_extract.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp]
        L.orc_orb_level_dims.argtypes = [vp, i32, vp, vp]
        for n in ("orc_orb_get_level", "orc_orb_get_level_bordered", "orc_orb_get_blurred"):
            getattr(L, n).argtypes = [vp, i32, vp]
        L.orc_orb_get_candidates.argtypes = [vp, i32, vp, vp, vp, i32]
        L.orc_orb_get_level_keypoints.argtypes = [vp, i32, vp, i32]
        L.orc_resize_linear_u8.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32]
        L.orc_gaussian_blur7_s2_u8.argtypes = [vp, i32, i32, i32, vp, i32]
        L.orc_gaussian_kernel7_s2_q8.argtypes = [vp]
        L.orc_fast9_16.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, i32]
        L.orc_fast_atan2.restype = f32
        L.orc_fast_atan2.argtypes = [f32, f32]
        L.orc_cv_round.argtypes = [f32]
        L.orc_sincos_deg.argtypes = [f32, vp, vp]
        L.orc_ic_angle.restype = f32
        L.orc_ic_angle.argtypes = [vp, i32, i32, i32]
        L.orc_brief_descriptor.argtypes = [vp, i32, i32, i32, f32, vp]
        L.orc_distribute_octree.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32]
        L.orc_descriptor_distance.argtypes = [vp, vp]
        L.orc_hamming_matrix.argtypes = [vp, i32, vp, i32, vp]
        L.orc_grid_create.restype = vp
        L.orc_grid_create.argtypes = [vp, i32, f32, f32, f32, f32]
        L.orc_grid_destroy.argtypes = [vp]
        L.orc_grid_features_in_area.argtypes = [vp, f32, f32, f32, i32, i32, vp, i32]
        L.orc_match_window.argtypes = [vp] * 9 + [i32] + [vp] * 5
        if path is not None:
            return L
        _LIB = L
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class OrbOracle:
    """CPU oracle of ORB_SLAM3::ORBextractor (reference ORBextractor.cc)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, libpath=None):
        self.L = lib(libpath) if libpath else lib()
        self.params = OrbParams(nfeatures, scale_factor, nlevels, ini_th, min_th)
        self.h = self.L.orc_orb_create(C.byref(self.params))
        self.nlevels = nlevels
        self.nfeatures = nfeatures

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_orb_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        nf = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        self.L.orc_orb_tables(self.h, _p(sc), _p(isc), _p(s2), _p(is2), _p(nf), _p(um))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, nfeat=nf, umax=um)

    def extract(self, img: np.ndarray, lap=(0, 1000), cap=None):
        img = np.ascontiguousarray(img, np.uint8)
        cap = cap or (self.nfeatures * 2 + 64)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        mono = C.c_int(0)
        n = self.L.orc_orb_extract(self.h, _p(img), img.shape[0], img.shape[1], img.strides[0], lap[0], lap[1],
                                   _p(kps), _p(desc), cap, C.byref(mono))
        if n < 0:
            return n, None, None, mono.value
        return n, kps[:n].copy(), desc[:n].copy(), mono.value

    def level_dims(self, level):
        r, c = C.c_int(0), C.c_int(0)
        self.L.orc_orb_level_dims(self.h, level, C.byref(r), C.byref(c))
        return r.value, c.value

    def level(self, level, bordered=False):
        r, c = self.level_dims(level)
        if bordered:
            out = np.zeros((r + 38, c + 38), np.uint8)
            self.L.orc_orb_get_level_bordered(self.h, level, _p(out))
        else:
            out = np.zeros((r, c), np.uint8)
            self.L.orc_orb_get_level(self.h, level, _p(out))
        return out

    def blurred(self, level):
        r, c = self.level_dims(level)
        out = np.zeros((r, c), np.uint8)
        rc = self.L.orc_orb_get_blurred(self.h, level, _p(out))
        return out if rc == 0 else None

    def candidates(self, level, cap=200000):
        xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
        n = self.L.orc_orb_get_candidates(self.h, level, _p(xs), _p(ys), _p(sc), cap)
        return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()

    def level_keypoints(self, level, cap=20000):
        k = np.zeros(cap, KP_DTYPE)
        n = self.L.orc_orb_get_level_keypoints(self.h, level, _p(k), cap)
        return k[:n].copy()


def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def gaussian_blur7(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().orc_gaussian_blur7_s2_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def gaussian_kernel7():
    k = np.zeros(7, np.int32)
    lib().orc_gaussian_kernel7_s2_q8(_p(k))
    return k


def fast9_16(roi: np.ndarray, threshold: int):
    roi = np.ascontiguousarray(roi, np.uint8)
    cap = roi.size
    xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().orc_fast9_16(_p(roi), roi.shape[1], roi.shape[0], roi.strides[0], threshold, _p(xs), _p(ys), _p(sc), cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(float(y), float(x)))


def sincos_deg(a):
    c, s = C.c_float(0), C.c_float(0)
    lib().orc_sincos_deg(float(a), C.byref(c), C.byref(s))
    return c.value, s.value


def ic_angle(img: np.ndarray, cx: int, cy: int) -> float:
    img = np.ascontiguousarray(img, np.uint8)
    return float(lib().orc_ic_angle(_p(img), img.strides[0], cx, cy))


def brief_descriptor(blurred: np.ndarray, cx: int, cy: int, angle_deg: float) -> np.ndarray:
    blurred = np.ascontiguousarray(blurred, np.uint8)
    out = np.zeros(32, np.uint8)
    lib().orc_brief_descriptor(_p(blurred), blurred.strides[0], cx, cy, float(angle_deg), _p(out))
    return out


def distribute_octree(xs, ys, scores, minX, maxX, minY, maxY, N):
    xs, ys, scores = (np.ascontiguousarray(a, np.int32) for a in (xs, ys, scores))
    cap = len(xs) + 8
    out = np.zeros(cap, np.int32)
    n = lib().orc_distribute_octree(_p(xs), _p(ys), _p(scores), len(xs), minX, maxX, minY, maxY, N, _p(out), cap)
    return out[:n].copy()


def hamming_matrix(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    A = np.ascontiguousarray(A, np.uint8)
    B = np.ascontiguousarray(B, np.uint8)
    D = np.zeros((len(A), len(B)), np.uint16)
    lib().orc_hamming_matrix(_p(A), len(A), _p(B), len(B), _p(D))
    return D


class Grid:
    """Frame's 64x48 feature grid (reference Frame.cc:443-506, 712-782)."""

    def __init__(self, kps: np.ndarray, minX=0.0, maxX=640.0, minY=0.0, maxY=480.0):
        self.L = lib()
        self.kps = np.ascontiguousarray(kps, KP_DTYPE)
        self.g = self.L.orc_grid_create(_p(self.kps), len(self.kps), minX, maxX, minY, maxY)

    def __del__(self):
        if getattr(self, "g", None):
            self.L.orc_grid_destroy(self.g)
            self.g = None

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        out = np.zeros(len(self.kps) + 1, np.int32)
        n = self.L.orc_grid_features_in_area(self.g, x, y, r, min_level, max_level, _p(out), len(out))
        return out[:n].copy()

    def match_window(self, tdesc, qdesc, qx, qy, qr, qmin, qmax, skip=None):
        nq = len(qdesc)
        tdesc = np.ascontiguousarray(tdesc, np.uint8)
        qdesc = np.ascontiguousarray(qdesc, np.uint8)
        qx, qy, qr = (np.ascontiguousarray(a, np.float32) for a in (qx, qy, qr))
        qmin, qmax = (np.ascontiguousarray(a, np.int32) for a in (qmin, qmax))
        outs = [np.zeros(nq, np.int32) for _ in range(5)]
        sk = _p(np.ascontiguousarray(skip, np.uint8)) if skip is not None else None
        self.L.orc_match_window(self.g, _p(tdesc), sk, _p(qdesc), _p(qx), _p(qy), _p(qr), _p(qmin), _p(qmax), nq,
                                *[_p(o) for o in outs])
        return dict(best_idx=outs[0], best_dist=outs[1], second_dist=outs[2], best_level=outs[3],
                    second_level=outs[4])


# ------------------------------------------------------------------------------------------- BA
BA_EDGE_DTYPE = np.dtype([("pose", "<i4"), ("point", "<i4"), ("u", "<f8"), ("v", "<f8"), ("inv_sigma2", "<f8")])


class BaCamera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("huber_delta", C.c_double)]


class BaStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("total_trials", C.c_int32), ("stop_reason", C.c_int32), ("pad", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
                ("trials_per_iter", C.c_int32 * 64), ("chi2_per_iter", C.c_double * 64),
                ("lambda_per_iter", C.c_double * 64)]


def make_edges(edge_pose, edge_point, obs, inv_sigma2) -> np.ndarray:
    e = np.zeros(len(edge_pose), BA_EDGE_DTYPE)
    e["pose"], e["point"] = edge_pose, edge_point
    e["u"], e["v"] = obs[:, 0], obs[:, 1]
    e["inv_sigma2"] = inv_sigma2
    return e


def ba_optimize(poses, fixed, points, edges, intrinsics, huber_delta, iterations, libpath=None, continue_graph=False):
    """Returns (poses, points, stats dict, edge_chi2).  continue_graph: a further optimize() on the same graph -- the estimates are
    taken as the previous call left them (no normalisation of the input quaternions)."""
    L = lib(libpath) if libpath else lib()
    fn = L.orc_ba_optimize_continue if continue_graph else L.orc_ba_optimize
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                   C.POINTER(BaCamera), C.c_int32, C.POINTER(BaStats), C.c_void_p]
    poses = np.array(poses, np.float64, copy=True, order="C")
    points = np.array(points, np.float64, copy=True, order="C")
    fixed = np.ascontiguousarray(fixed, np.uint8)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    cam = BaCamera(*[float(v) for v in intrinsics], float(huber_delta))
    st = BaStats()
    chi = np.zeros(len(edges), np.float64)
    it = fn(_p(poses), _p(fixed), len(poses), _p(points), len(points), _p(edges), len(edges),
            C.byref(cam), iterations, C.byref(st), _p(chi))
    n = st.iterations
    stats = dict(iterations=it, total_trials=st.total_trials, stop_reason=st.stop_reason, chi2_initial=st.chi2_initial,
                 chi2_final=st.chi2_final, lambda_final=st.lambda_final, trials=list(st.trials_per_iter[:n]),
                 chi2=list(st.chi2_per_iter[:n]), lam=list(st.lambda_per_iter[:n]))
    return poses, points, stats, chi


def f64_spec(x):
    """oracle/f64_spec.h: (sin, cos, cube) of the doubles in x."""
    L = lib()
    x = np.ascontiguousarray(x, np.float64)
    out = np.zeros(3 * len(x), np.float64)
    L.orc_f64_spec.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.orc_f64_spec(_p(x), len(x), _p(out))
    return out[:len(x)], out[len(x):2 * len(x)], out[2 * len(x):]


def ba_edge_chi2(poses, points, edges, intrinsics):
    L = lib()
    L.orc_ba_edge_chi2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(BaCamera), C.c_void_p, C.c_void_p]
    poses = np.ascontiguousarray(poses, np.float64)
    points = np.ascontiguousarray(points, np.float64)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    cam = BaCamera(*[float(v) for v in intrinsics], 0.0)
    chi = np.zeros(len(edges), np.float64)
    dp = np.zeros(len(edges), np.uint8)
    L.orc_ba_edge_chi2(_p(poses), _p(points), _p(edges), len(edges), C.byref(cam), _p(chi), _p(dp))
    return chi, dp


def pose_optimize(pose, Xw, obs, inv_sigma2, intrinsics):
    """Optimizer::PoseOptimization.  Returns (pose[7], outlier mask, n_inliers)."""
    L = lib()
    L.orc_pose_optimize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(BaCamera), C.c_void_p]
    pose = np.array(pose, np.float64, copy=True)
    Xw = np.ascontiguousarray(Xw, np.float64)
    obs = np.ascontiguousarray(obs, np.float64)
    inv_sigma2 = np.ascontiguousarray(inv_sigma2, np.float64)
    cam = BaCamera(*[float(v) for v in intrinsics], 0.0)
    out = np.zeros(len(Xw), np.uint8)
    n = L.orc_pose_optimize(_p(pose), _p(Xw), _p(obs), _p(inv_sigma2), len(Xw), C.byref(cam), _p(out))
    return pose, out, n


# ------------------------------------------------------------------------------------- frustum
class FrustumFrame(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float),
                ("max_y", C.c_float), ("bf", C.c_float), ("log_scale_factor", C.c_float), ("n_levels", C.c_int32)]


TRACK_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("depth", "<f4"), ("view_cos", "<f4"),
                        ("level", "<i4"), ("in_view", "<i4")])


def pose_matrices(Tcw):
    """Frame::UpdatePoseMatrices (Frame.cc:553-559) on a 7-float SE3f (qx, qy, qz, qw, t): (mRcw[3,3], mtcw, mOw) float32."""
    Rcw = np.zeros(9, np.float32); tcw = np.zeros(3, np.float32); Ow = np.zeros(3, np.float32)
    _call(lib().orc_pose_matrices, None, _f32(Tcw), Rcw, tcw, Ow)
    return Rcw.reshape(3, 3), tcw, Ow


def se3_inverse(T):
    out = np.zeros(7, np.float32)
    _call(lib().orc_se3_inverse, None, _f32(T), out)
    return out


def sim3_inverse(S):
    out = np.zeros(7, np.float32)
    _call(lib().orc_sim3_inverse, None, _f32(S), out)
    return out


def sim3_to_se3(S):
    """(Tcw7, Ow) = SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale()), Tcw.inverse().translation()."""
    T = np.zeros(7, np.float32); Ow = np.zeros(3, np.float32)
    _call(lib().orc_sim3_to_se3, None, _f32(S), T, Ow)
    return T, Ow


def se3_act(T, P):
    P = _f32(P).reshape(-1, 3); out = np.zeros_like(P)
    _call(lib().orc_se3_act, None, _f32(T), P, len(P), out)
    return out


def sim3_act(S, P):
    P = _f32(P).reshape(-1, 3); out = np.zeros_like(P)
    _call(lib().orc_sim3_act, None, _f32(S), P, len(P), out)
    return out


def logf(x):
    L = lib()
    L.orc_logf.restype = C.c_float; L.orc_logf.argtypes = [C.c_float]
    return np.array([L.orc_logf(float(v)) for v in np.asarray(x, np.float32).reshape(-1)], np.float32)


def make_frustum_frame(Tcw, K, bounds=(0.0, 640.0, 0.0, 480.0), bf=0.0, scale_factor=1.2, n_levels=8, cls=FrustumFrame, matrices=None):
    """Tcw: 7-float SE3f; mRcw / mtcw / mOw are derived as Frame::UpdatePoseMatrices does (`matrices` = a function doing that:
    the HIP tests pass the product's own, capi.pose_matrices)."""
    Rcw, tcw, Ow = (matrices or pose_matrices)(Tcw)
    F = cls()
    F.Rcw[:] = list(Rcw.reshape(-1)); F.tcw[:] = list(tcw); F.Ow[:] = list(Ow)
    F.fx, F.fy, F.cx, F.cy = [float(np.float32(v)) for v in K]
    F.min_x, F.max_x, F.min_y, F.max_y = [float(v) for v in bounds]
    F.bf = float(bf); F.log_scale_factor = float(np.float32(np.log(np.float32(scale_factor)))); F.n_levels = n_levels
    return F


def is_in_frustum(F, P, normal, min_dist, max_dist, viewing_cos_limit=0.5):
    L = lib()
    L.orc_is_in_frustum.argtypes = [C.POINTER(FrustumFrame), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p]
    P = np.ascontiguousarray(P, np.float32); normal = np.ascontiguousarray(normal, np.float32)
    min_dist = np.ascontiguousarray(min_dist, np.float32); max_dist = np.ascontiguousarray(max_dist, np.float32)
    out = np.zeros(len(P), TRACK_DTYPE)
    L.orc_is_in_frustum(C.byref(F), _p(P), _p(normal), _p(min_dist), _p(max_dist), len(P), float(viewing_cos_limit), _p(out))
    return out


def undistort_points(cam, xy):
    """Frame::UndistortKeyPoints' cv::undistortPoints(pts, K, D, noArray(), K); cam = (fx, fy, cx, cy, k1, k2, p1, p2, k3)."""
    L = lib()
    L.orc_undistort_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    cam = np.ascontiguousarray(cam, np.float32); xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    assert cam.shape == (9,)
    out = np.zeros_like(xy)
    L.orc_undistort_points(_p(cam), _p(xy), len(xy), _p(out))
    return out


def image_bounds(cam, cols, rows):
    """Frame::ComputeImageBounds -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    L = lib()
    L.orc_image_bounds.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    cam = np.ascontiguousarray(cam, np.float32)
    out = np.zeros(4, np.float32)
    L.orc_image_bounds(_p(cam), int(cols), int(rows), _p(out))
    return out


def optimize_sim3(S12, fix_scale, P1c, P2c, obs1, obs2, w1, w2, K1, K2, th2):
    """Optimizer::OptimizeSim3 numerics.  Returns (S12[8], inlier mask, nIn)."""
    L = lib()
    vp = C.c_void_p
    L.orc_optimize_sim3.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp, C.c_int32, vp, vp, C.c_double, vp]
    S = np.array(S12, np.float64, copy=True)
    arrs = [np.ascontiguousarray(a, np.float64) for a in (P1c, P2c, obs1, obs2, w1, w2, K1, K2)]
    inl = np.zeros(len(arrs[0]), np.uint8)
    n = L.orc_optimize_sim3(_p(S), int(fix_scale), *[_p(a) for a in arrs[:6]], len(arrs[0]), _p(arrs[6]), _p(arrs[7]), float(th2), _p(inl))
    return S, inl, n


MAP_POINT_DTYPE = np.dtype([("pos", "<f4", (3,)), ("desc", "u1", (32,)), ("n_obs", "<i4")])


def search_by_projection_frames(kps_c, desc_c, mp_c, Tcw, K, bounds, scale_factors, kps_l, mp_l, outlier_l, mps, th,
                                check_ori=True):
    """Whole ORBmatcher::SearchByProjection(CurrentFrame, LastFrame) (mono).  Returns (nmatches, mp_c updated copy)."""
    L = lib()
    vp = C.c_void_p
    L.orc_search_by_projection_frames.restype = C.c_int32
    L.orc_search_by_projection_frames.argtypes = [C.c_int32, vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp,
                                                  C.c_float, C.c_int32]
    kps_c = np.ascontiguousarray(kps_c, KP_DTYPE); kps_l = np.ascontiguousarray(kps_l, KP_DTYPE)
    desc_c = np.ascontiguousarray(desc_c, np.uint8)
    mp = np.array(mp_c, np.int32, copy=True)
    mp_l = np.ascontiguousarray(mp_l, np.int32)
    outl = None if outlier_l is None else np.ascontiguousarray(outlier_l, np.uint8)
    f = [np.ascontiguousarray(a, np.float32) for a in (Tcw, K, bounds, scale_factors)]
    mps = np.ascontiguousarray(mps, MAP_POINT_DTYPE)
    n = L.orc_search_by_projection_frames(len(kps_c), _p(kps_c), _p(desc_c), _p(mp), *[_p(a) for a in f], len(kps_l), _p(kps_l),
                                          _p(mp_l), None if outl is None else _p(outl), _p(mps), float(th), int(check_ori))
    return n, mp


TRACKED_POINT_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("depth", "<f4"), ("view_cos", "<f4"), ("level", "<i4"),
                                ("in_view", "u1"), ("bad", "u1"), ("pad", "u1", (2,)), ("desc", "u1", (32,)), ("n_obs", "<i4")])


def search_by_projection_points(kps, desc, mp, claimed_obs, bounds, scale_factors, pts, th, nnratio=0.8, far_points=False,
                                th_far=0.0):
    """Whole ORBmatcher::SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints) (mono).  Returns (nmatches, mp copy)."""
    L = lib()
    vp = C.c_void_p
    L.orc_search_by_projection_points.restype = C.c_int32
    L.orc_search_by_projection_points.argtypes = [C.c_int32, vp, vp, vp, vp, vp, vp, vp, C.c_int32, C.c_float, C.c_float,
                                                  C.c_int32, C.c_float]
    kps = np.ascontiguousarray(kps, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    mpc = np.array(mp, np.int32, copy=True)
    co = np.ascontiguousarray(claimed_obs, np.uint8)
    b = np.ascontiguousarray(bounds, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
    pts = np.ascontiguousarray(pts, TRACKED_POINT_DTYPE)
    n = L.orc_search_by_projection_points(len(kps), _p(kps), _p(desc), _p(mpc), _p(co), _p(b), _p(sf), _p(pts), len(pts),
                                          float(th), float(nnratio), int(far_points), float(th_far))
    return n, mpc


class F32(float):
    """Marks a Python float as a C float argument for _call."""


class F64(float):
    """Marks a Python float as a C double argument for _call."""


def _call(fn, restype, *args):
    """ctypes call without an argtypes table: arrays -> pointers, int -> int32, F32 / F64 -> float / double."""
    conv = []
    for a in args:
        if a is None:
            conv.append(None)
        elif isinstance(a, np.ndarray):
            conv.append(C.c_void_p(a.ctypes.data) if a.size else None)
        elif isinstance(a, F32):
            conv.append(C.c_float(a))
        elif isinstance(a, F64):
            conv.append(C.c_double(a))
        elif isinstance(a, (int, np.integer, bool)):
            conv.append(C.c_int32(int(a)))
        else:
            raise TypeError(type(a))
    fn.restype = restype
    fn.argtypes = None
    return fn(*conv)


def _kd(kps, desc):
    return np.ascontiguousarray(kps, KP_DTYPE), np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)


def _fv(fv):
    """FeatureVector dict (fv_nodes, fv_off, fv_feat) -> 3 int32 arrays + node count."""
    n, o, f = (np.ascontiguousarray(fv[k], np.int32) for k in ("fv_nodes", "fv_off", "fv_feat"))
    return n, o, f, len(n)


def _u8(x):
    return None if x is None else np.ascontiguousarray(x, np.uint8)


def _f32(x):
    return None if x is None else np.ascontiguousarray(x, np.float32)


def search_for_initialization(kps1, desc1, kps2, desc2, bounds, prev_matched, window=100, nnratio=0.9, check_ori=True):
    """ORBmatcher::SearchForInitialization.  Returns (nmatches, vnMatches12, vbPrevMatched updated)."""
    k1, d1 = _kd(kps1, desc1); k2, d2 = _kd(kps2, desc2)
    pm = np.array(prev_matched, np.float32, copy=True).reshape(-1, 2)
    m = np.zeros(len(k1), np.int32)
    n = _call(lib().orc_search_for_initialization, C.c_int32, len(k1), k1, d1, len(k2), k2, d2, _f32(bounds), pm, m, int(window),
              F32(nnratio), int(check_ori))
    return n, m, pm


def search_by_bow_kf_frame(kps_kf, desc_kf, mp_kf, bad_kf, fv_kf, kps_f, desc_f, fv_f, nnratio=0.7, check_ori=True):
    """ORBmatcher::SearchByBoW(KF, F, vpMapPointMatches) (mono).  Returns (nmatches, matches[F.N] = map point id or -1)."""
    kk, dk = _kd(kps_kf, desc_kf); kf, df = _kd(kps_f, desc_f)
    a = _fv(fv_kf); b = _fv(fv_f)
    m = np.zeros(len(kf), np.int32)
    n = _call(lib().orc_search_by_bow_kf_frame, C.c_int32, kk, dk, np.ascontiguousarray(mp_kf, np.int32), _u8(bad_kf), a[0], a[1], a[2],
              a[3], len(kf), kf, df, b[0], b[1], b[2], b[3], F32(nnratio), int(check_ori), m)
    return n, m


def search_by_bow_kf_kf(kps1, desc1, mp1, bad1, fv1, kps2, desc2, mp2, bad2, fv2, nnratio=0.8, check_ori=True):
    """ORBmatcher::SearchByBoW(KF1, KF2, vpMatches12).  Returns (nmatches, matches12[N1] = KF2 map point id or -1)."""
    k1, d1 = _kd(kps1, desc1); k2, d2 = _kd(kps2, desc2)
    a = _fv(fv1); b = _fv(fv2)
    m = np.zeros(len(k1), np.int32)
    n = _call(lib().orc_search_by_bow_kf_kf, C.c_int32, len(k1), k1, d1, np.ascontiguousarray(mp1, np.int32), _u8(bad1), a[0], a[1], a[2],
              a[3], len(k2), k2, d2, np.ascontiguousarray(mp2, np.int32), _u8(bad2), b[0], b[1], b[2], b[3], F32(nnratio),
              int(check_ori), m)
    return n, m


def triangulation_geometry(T1w, T2w, K1, K2):
    """R12, t12, epipole in image 2 and F12 (float32) as SearchForTriangulation / epipolarConstrain build them."""
    R12 = np.zeros(9, np.float32); t12 = np.zeros(3, np.float32); ep = np.zeros(2, np.float32); F12 = np.zeros(9, np.float32)
    _call(lib().orc_triangulation_geometry, None, _f32(T1w), _f32(T2w), _f32(K1), _f32(K2),
          R12, t12, ep, F12)
    return R12, t12, ep, F12


def search_for_triangulation(kps1, desc1, mp1, fv1, kps2, desc2, mp2, fv2, F12, ep, scale_factors2, level_sigma2_2, coarse=False,
                             check_ori=True):
    """ORBmatcher::SearchForTriangulation (mono).  Returns (nmatches, pairs[nmatches, 2])."""
    k1, d1 = _kd(kps1, desc1); k2, d2 = _kd(kps2, desc2)
    a = _fv(fv1); b = _fv(fv2)
    pairs = np.zeros((max(len(k1), 1), 2), np.int32)
    n = _call(lib().orc_search_for_triangulation, C.c_int32, len(k1), k1, d1, np.ascontiguousarray(mp1, np.int32), a[0], a[1], a[2], a[3],
              len(k2), k2, d2, np.ascontiguousarray(mp2, np.int32), b[0], b[1], b[2], b[3], _f32(F12), _f32(ep), _f32(scale_factors2),
              _f32(level_sigma2_2), int(coarse), int(check_ori), pairs)
    return n, pairs[:n]


def project_search(kps, desc, bounds, skip, Tcw, Ow, K, pts, th, scale_factors, log_scale_factor, gate_inv_sigma2=None,
                   gate=0.0):
    """Projection gates + window search of Fuse / SearchByProjection(KF, Scw, ...).  pts: dict(pos, normal, min_dist, max_dist,
    desc, valid).  Returns (best_idx, best_dist, proj[n, 4] = u, v, radius, level)."""
    k, d = _kd(kps, desc)
    n = len(pts["pos"])
    bi = np.zeros(n, np.int32); bd = np.zeros(n, np.int32); pr = np.zeros((n, 4), np.float32)
    sf = _f32(scale_factors)
    _call(lib().orc_project_search, None, len(k), k, d, _f32(bounds), _u8(skip), _f32(Tcw), _f32(Ow), _f32(K), n,
          _f32(pts["pos"]), _f32(pts["normal"]), _f32(pts["min_dist"]), _f32(pts["max_dist"]), _u8(pts["desc"]), _u8(pts.get("valid")),
          F32(th), sf, F32(log_scale_factor), len(sf), _f32(gate_inv_sigma2), F64(gate), bi, bd, pr)
    return bi, bd, pr


def fuse_sim3(kps, desc, bounds, kf_mp, kf_mp_bad, Scw, K, pts, th, scale_factors, log_scale_factor):
    """ORBmatcher::Fuse(KF, Scw, vpPoints, th, vpReplacePoint).  pts adds id, bad.  Returns (nFused, kf_mp updated, replace)."""
    k, d = _kd(kps, desc)
    n = len(pts["pos"])
    mp = np.array(kf_mp, np.int32, copy=True); rep = np.zeros(n, np.int32)
    sf = _f32(scale_factors)
    nf = _call(lib().orc_fuse_sim3, C.c_int32, len(k), k, d, _f32(bounds), mp, _u8(kf_mp_bad), _f32(Scw),
               _f32(K), n, np.ascontiguousarray(pts["id"], np.int32), _u8(pts.get("bad")), _f32(pts["pos"]), _f32(pts["normal"]),
               _f32(pts["min_dist"]), _f32(pts["max_dist"]), _u8(pts["desc"]), F32(th), sf, F32(log_scale_factor), len(sf), rep)
    return nf, mp, rep


def search_by_projection_sim3(kps, desc, bounds, matched, Scw, K, pts, th, ratio_hamming, scale_factors, log_scale_factor):
    """ORBmatcher::SearchByProjection(KF, Scw, vpPoints, vpMatched, th, ratioHamming).  Returns (nmatches, vpMatched updated)."""
    k, d = _kd(kps, desc)
    n = len(pts["pos"])
    m = np.array(matched, np.int32, copy=True)
    sf = _f32(scale_factors)
    nm = _call(lib().orc_search_by_projection_sim3, C.c_int32, len(k), k, d, _f32(bounds), m, _f32(Scw),
               _f32(K), n, np.ascontiguousarray(pts["id"], np.int32), _u8(pts.get("bad")), _f32(pts["pos"]), _f32(pts["normal"]),
               _f32(pts["min_dist"]), _f32(pts["max_dist"]), _u8(pts["desc"]), int(th), F32(ratio_hamming), sf, F32(log_scale_factor),
               len(sf))
    return nm, m


def search_by_sim3(kf1, mps1, kf2, mps2, S12, th, matches12, idx_in_kf2):
    """ORBmatcher::SearchBySim3.  S12: 7-float Sim3f.  kfN: keyframe dicts (kps, desc, mp, bad, Tcw, bounds, K, scale_factors, log_scale_factor);
    mpsN: per-keypoint map point data dict(pos, min_dist, max_dist, desc).  Returns (nFound, vpMatches12 updated)."""
    k1, d1 = _kd(kf1["kps"], kf1["desc"]); k2, d2 = _kd(kf2["kps"], kf2["desc"])
    m = np.array(matches12, np.int32, copy=True)
    sf = _f32(kf1["scale_factors"])
    n = _call(lib().orc_search_by_sim3, C.c_int32, len(k1), k1, d1, np.ascontiguousarray(kf1["mp"], np.int32), _u8(kf1.get("bad")),
              _f32(mps1["pos"]), _f32(mps1["min_dist"]), _f32(mps1["max_dist"]), _u8(mps1["desc"]), _f32(kf1["Tcw"]),
              len(k2), k2, d2, np.ascontiguousarray(kf2["mp"], np.int32), _u8(kf2.get("bad")), _f32(mps2["pos"]), _f32(mps2["min_dist"]),
              _f32(mps2["max_dist"]), _u8(mps2["desc"]), _f32(kf2["Tcw"]), _f32(kf1["bounds"]), _f32(kf1["K"]),
              _f32(S12), F32(th), sf, F32(kf1["log_scale_factor"]), len(sf), m,
              None if idx_in_kf2 is None else np.ascontiguousarray(idx_in_kf2, np.int32))
    return n, m


def search_by_projection_reloc(cur_kps, cur_desc, cur_mp, bounds, Tcw, K, kf, kf_pts, already, th, orb_dist, scale_factors,
                               log_scale_factor, check_ori=True):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist).  Returns (nmatches, mvpMapPoints updated)."""
    kc, dc = _kd(cur_kps, cur_desc); kk, _ = _kd(kf["kps"], kf["desc"])
    m = np.array(cur_mp, np.int32, copy=True)
    al = np.sort(np.ascontiguousarray(already, np.int32))
    sf = _f32(scale_factors)
    n = _call(lib().orc_search_by_projection_reloc, C.c_int32, len(kc), kc, dc, m, _f32(bounds), _f32(Tcw), _f32(K),
              len(kk), kk, np.ascontiguousarray(kf["mp"], np.int32), _u8(kf.get("bad")), _f32(kf_pts["pos"]), _f32(kf_pts["min_dist"]),
              _f32(kf_pts["max_dist"]), _u8(kf_pts["desc"]), al, len(al), F32(th), int(orb_dist), sf, F32(log_scale_factor), len(sf), int(check_ori))
    return n, m


def distinctive_descriptors(desc, off):
    """MapPoint::ComputeDistinctiveDescriptors for a batch.  Returns (best_idx, best_median)."""
    L = lib()
    L.orc_distinctive_descriptors.restype = None
    L.orc_distinctive_descriptors.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); off = np.ascontiguousarray(off, np.int32)
    n = len(off) - 1
    bi = np.zeros(max(n, 1), np.int32); bm = np.zeros(max(n, 1), np.int32)
    if n > 0:
        L.orc_distinctive_descriptors(_p(desc) if len(desc) else None, _p(off), n, _p(bi), _p(bm))
    return bi[:n], bm[:n]


def vocab_transform(voc, features, levelsup):
    """DBoW2 transform (TF_IDF / L1) on a synthetic vocabulary.  Returns dict(word, node, weight, bow_ids, bow_vals,
    fv_nodes, fv_off, fv_feat)."""
    L = lib()
    vp, i32 = C.c_void_p, C.c_int32
    L.orc_vocab_transform.restype = None
    L.orc_vocab_transform.argtypes = [i32, vp, vp, vp, vp, vp, i32, vp, i32, i32] + [vp] * 10
    f = np.ascontiguousarray(features, np.uint8).reshape(-1, 32)
    n = len(f)
    word = np.zeros(n, np.int32); node = np.zeros(n, np.int32); w = np.zeros(n, np.float64)
    bi = np.zeros(n + 1, np.int32); bv = np.zeros(n + 1, np.float64); fn = np.zeros(n + 1, np.int32)
    fo = np.zeros(n + 2, np.int32); ff = np.zeros(n + 1, np.int32)
    nb = np.zeros(1, np.int32); nf = np.zeros(1, np.int32)
    L.orc_vocab_transform(voc["n_nodes"], _p(voc["child_off"]), _p(voc["children"]), _p(voc["desc"]), _p(voc["weight"]),
                          _p(voc["word_id"]), voc["L"], _p(f), n, levelsup, _p(word), _p(node), _p(w), _p(bi), _p(bv), _p(nb),
                          _p(fn), _p(fo), _p(ff), _p(nf))
    return dict(word=word, node=node, weight=w, bow_ids=bi[:nb[0]], bow_vals=bv[:nb[0]], fv_nodes=fn[:nf[0]],
                fv_off=fo[:nf[0] + 1], fv_feat=ff[:fo[nf[0]]])


def bow_score(ids1, vals1, ids2, vals2):
    L = lib()
    L.orc_bow_score.restype = C.c_double
    L.orc_bow_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    a = np.ascontiguousarray(ids1, np.int32); b = np.ascontiguousarray(vals1, np.float64)
    c = np.ascontiguousarray(ids2, np.int32); d = np.ascontiguousarray(vals2, np.float64)
    return L.orc_bow_score(_p(a), _p(b), len(a), _p(c), _p(d), len(c))


def sim3_hypotheses(P1c, P2c, max_err1, max_err2, K1, K2, triples, fix_scale=False):
    """Sim3Solver::ComputeSim3 + CheckInliers for given minimal sets.  Returns (T12[H,13], n_inliers[H], mask[H,N])."""
    L = lib()
    vp, i32 = C.c_void_p, C.c_int32
    L.orc_sim3_hypotheses.restype = None
    L.orc_sim3_hypotheses.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp]
    a = [np.ascontiguousarray(x, np.float32) for x in (P1c, P2c, max_err1, max_err2, K1, K2)]
    tr = np.ascontiguousarray(triples, np.int32).reshape(-1, 3)
    N, H = len(a[0]), len(tr)
    T = np.zeros((H, 13), np.float32); nin = np.zeros(H, np.int32); mask = np.zeros((H, N), np.uint8)
    L.orc_sim3_hypotheses(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), N, _p(a[4]), _p(a[5]), _p(tr), H, int(fix_scale), _p(T), _p(nin), _p(mask))
    return T, nin, mask


def triangulate_matches(K1, K2, T1w, T2w, Ow1, Ow2, kps1, kps2, pairs, sigma2_1, sigma2_2, sf1, sf2, ratio_factor,
                        cos_parallax_max=0.9998, far_points=False, th_far=0.0):
    """LocalMapping::CreateNewMapPoints' per-match geometry (mono pinhole).  kps*: structured keypoint arrays (KP_DTYPE) or
    [N,7] float32 in cv::KeyPoint layout.  Returns (x3D[n,3] float32, status[n] int32)."""
    L = lib()
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    L.orc_triangulate_matches.restype = None
    L.orc_triangulate_matches.argtypes = [vp] * 9 + [i32] + [vp] * 4 + [f32, C.c_double, i32, f32, vp, vp]
    f = [np.ascontiguousarray(x, np.float32) for x in (K1, K2, T1w, T2w, Ow1, Ow2, sigma2_1, sigma2_2, sf1, sf2)]
    k1 = np.ascontiguousarray(kps1); k2 = np.ascontiguousarray(kps2)
    assert k1.dtype.itemsize * (k1.shape[1] if k1.ndim == 2 else 1) == 28 and k2.dtype.itemsize * (k2.shape[1] if k2.ndim == 2 else 1) == 28
    pr = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    n = len(pr)
    X = np.zeros((n, 3), np.float32); st = np.zeros(n, np.int32)
    L.orc_triangulate_matches(_p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]), _p(f[4]), _p(f[5]), _p(k1), _p(k2), _p(pr), n, _p(f[6]), _p(f[7]),
                              _p(f[8]), _p(f[9]), float(ratio_factor), float(cos_parallax_max), int(far_points), float(th_far), _p(X), _p(st))
    return X, st


def sim3_exp_log(u):
    L = lib()
    L.orc_sim3_exp_log.restype = None
    L.orc_sim3_exp_log.argtypes = [C.c_void_p] * 3
    u = np.ascontiguousarray(u, np.float64); S = np.zeros(8); lg = np.zeros(7)
    L.orc_sim3_exp_log(_p(u), _p(S), _p(lg))
    return S, lg


def pose_graph_optimize(S, fixed, edges_v, edges_meas, fix_scale=False, iterations=20):
    """OptimizeEssentialGraph numerics.  S[n,8] (q_xyzw,t,s); edges_v[E,2]; edges_meas[E,8].  Returns (S_opt, stats[6])."""
    L = lib()
    vp, i32 = C.c_void_p, C.c_int32
    L.orc_pose_graph_optimize.restype = i32
    L.orc_pose_graph_optimize.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, vp]
    So = np.array(S, np.float64, copy=True)
    fx = np.ascontiguousarray(fixed, np.uint8); ev = np.ascontiguousarray(edges_v, np.int32); em = np.ascontiguousarray(edges_meas, np.float64)
    st = np.zeros(70)
    L.orc_pose_graph_optimize(_p(So), _p(fx), len(So), _p(ev), _p(em), len(ev), int(fix_scale), int(iterations), _p(st))
    return So, st


class KeyFrameDatabase:
    """KeyFrameDatabase place-recognition queries (kfdb_oracle.cpp) on keyframe slots."""
    PREFIX = "orc_kfdb_"

    def _lib(self):
        return lib()

    def _create(self):
        f = getattr(self._lib(), self.PREFIX + "create"); f.restype = C.c_void_p; f.argtypes = []
        return C.c_void_p(f())

    def __init__(self):
        self.h = self._create()
        assert self.h.value

    def _f(self, name, restype=None):
        f = getattr(self._lib(), self.PREFIX + name); f.restype = restype; f.argtypes = None
        return f

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self._f("destroy")(self.h); self.h = C.c_void_p()

    __del__ = close

    @staticmethod
    def _bow(ids, vals):
        return np.ascontiguousarray(ids, np.int32), np.ascontiguousarray(vals, np.float64)

    def add(self, ids, vals, map_id, uuid, mn_id):
        i, v = self._bow(ids, vals)
        return self._f("add", C.c_int32)(self.h, _p(i), _p(v), C.c_int32(len(i)), C.c_int32(map_id), C.c_uint64(uuid), C.c_int64(mn_id))

    def erase(self, slot): self._f("erase")(self.h, C.c_int32(slot))
    def set_bad(self, slot, bad): self._f("set_bad")(self.h, C.c_int32(slot), C.c_int32(int(bad)))
    def set_map(self, slot, map_id): self._f("set_map")(self.h, C.c_int32(slot), C.c_int32(map_id))
    def set_map_bad(self, map_id, bad): self._f("set_map_bad")(self.h, C.c_int32(map_id), C.c_int32(int(bad)))

    def set_neighbours(self, slot, neigh):
        a = np.ascontiguousarray(neigh, np.int32)
        self._f("set_neighbours")(self.h, C.c_int32(slot), _p(a) if len(a) else None, C.c_int32(len(a)))

    def set_connected(self, slot, conn):
        a = np.ascontiguousarray(conn, np.int32)
        self._f("set_connected")(self.h, C.c_int32(slot), _p(a) if len(a) else None, C.c_int32(len(a)))

    def state(self, slot):
        q = C.c_uint64(0); w = C.c_int32(0); s = C.c_float(0)
        self._f("get_state")(self.h, C.c_int32(slot), C.byref(q), C.byref(w), C.byref(s))
        return q.value, w.value, s.value

    def merge_score(self, ids, vals, key_frame_id, map_id, score=0.0):
        i, v = self._bow(ids, vals)
        sc = C.c_float(score); best = C.c_int32(-1)
        self._f("merge_score", C.c_int32)(self.h, _p(i), _p(v), C.c_int32(len(i)), C.c_uint64(key_frame_id), C.c_int32(map_id), C.byref(sc), C.byref(best))
        return sc.value, best.value

    def detect_merge_possibility(self, ids, vals, uuid, map_id):
        i, v = self._bow(ids, vals)
        best = C.c_int32(-1); sc = C.c_float(0); base = C.c_float(0)
        r = self._f("detect_merge_possibility", C.c_int32)(self.h, _p(i), _p(v), C.c_int32(len(i)), C.c_uint64(uuid), C.c_int32(map_id), C.byref(best),
                                                           C.byref(sc), C.byref(base))
        return r, best.value, sc.value, base.value

    def detect_reloc(self, ids, vals, frame_id, map_id, cap=4096):
        """DetectRelocalizationCandidates(F, pMap): candidate slots in the reference's order."""
        i, v = self._bow(ids, vals)
        out = np.zeros(cap, np.int32); n = C.c_int32(0)
        self._f("detect_reloc", C.c_int32)(self.h, _p(i), _p(v), C.c_int32(len(i)), C.c_uint64(frame_id), C.c_int32(map_id), _p(out), C.byref(n))
        return out[:n.value].copy()

    def reloc_state(self, slot):
        q = C.c_uint64(0); w = C.c_int32(0); sc = C.c_float(0)
        self._f("get_reloc_state")(self.h, C.c_int32(slot), C.byref(q), C.byref(w), C.byref(sc))
        return q.value, w.value, sc.value

    def detect_n_best(self, slot, n_num):
        lo = np.zeros(max(n_num, 1), np.int32); me = np.zeros(max(n_num, 1), np.int32)
        nl = C.c_int32(0); nm = C.c_int32(0)
        self._f("detect_n_best", C.c_int32)(self.h, C.c_int32(slot), C.c_int32(n_num), _p(lo), C.byref(nl), _p(me), C.byref(nm))
        return lo[:nl.value].copy(), me[:nm.value].copy()
