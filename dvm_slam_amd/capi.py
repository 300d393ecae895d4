"""ctypes binding of libdvmslam_hip.so (include/dvmslam_hip.h).

This is the only way Python (tests, bench.py, smoke) reaches the product: through the C ABI a
reference maintainer would bind.  There is NO CPU fallback: if the shared library is missing the
import raises, and on a box without a gfx950 device every compute entry point returns
DVM_ERR_NO_DEVICE (raised here as DvmError).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVM_HIP_LIB") or os.path.join(_HERE, "lib", "libdvmslam_hip.so")   # (DVM_HIP_LIB: a measurement build, tools/build_flow_stamps.sh)

DVM_OK = 0
ERRORS = {-1: "DVM_ERR_INVALID", -2: "DVM_ERR_EMPTY", -3: "DVM_ERR_CAPACITY", -4: "DVM_ERR_HIP",
          -5: "DVM_ERR_NO_DEVICE", -6: "DVM_ERR_STATE"}

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
MATCH_DTYPE = np.dtype([("best_idx", "<i4"), ("best_dist", "<i4"), ("second_dist", "<i4"),
                        ("best_level", "<i2"), ("second_level", "<i2")])


class DvmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


BA_EDGE_DTYPE = np.dtype([("pose", "<i4"), ("point", "<i4"), ("u", "<f8"), ("v", "<f8"), ("inv_sigma2", "<f8")])


class BaCamera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("huber_delta", C.c_double)]


class BaStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("total_trials", C.c_int32), ("stop_reason", C.c_int32), ("kernel_us", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
                ("trials_per_iter", C.c_int32 * 64), ("chi2_per_iter", C.c_double * 64),
                ("lambda_per_iter", C.c_double * 64), ("ms_structure", C.c_double), ("ms_optimize", C.c_double),
                ("spec_trials", C.c_int32), ("spec_kept", C.c_int32)]


class FrustumFrame(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float),
                ("max_y", C.c_float), ("bf", C.c_float), ("log_scale_factor", C.c_float), ("n_levels", C.c_int32)]


class TriPair(C.Structure):   # == dvm_tri_pair
    _fields_ = [("cos_parallax_max", C.c_double), ("K1", C.c_float * 4), ("K2", C.c_float * 4), ("T1w", C.c_float * 12), ("T2w", C.c_float * 12),
                ("Ow1", C.c_float * 3), ("Ow2", C.c_float * 3), ("ratio_factor", C.c_float), ("th_far", C.c_float), ("far_points", C.c_int32),
                ("n_levels", C.c_int32)]


TRACK_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("depth", "<f4"), ("view_cos", "<f4"),
                        ("level", "<i4"), ("in_view", "<i4")])

_LIB = None


def lib():
    """Loads the HIP library; raises if it has not been built (no silent fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); dvm_slam_amd has no CPU implementation")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same soname as
    # /opt/rocm's).  Whichever is loaded first serves both; loading torch AFTER us would pull a second
    # runtime in and neither would see the GPU.  So when torch is importable, import it first.
    if os.environ.get("DVM_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.dvm_last_error.restype = C.c_char_p
    L.dvm_version.restype = C.c_char_p
    L.dvm_device_count.restype = i32
    L.dvm_orb_create.argtypes = [C.POINTER(OrbParams), i32, i32, C.POINTER(vp)]
    L.dvm_orb_destroy.argtypes = [vp]
    L.dvm_orb_destroy.restype = None
    L.dvm_orb_tables.argtypes = [vp] * 6
    L.dvm_orb_extract.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp]
    L.dvm_orb_extract_batch_device.argtypes = [vp, vp, i32, i32, i32, i32, i64, i32, i32]
    L.dvm_orb_extract_batch_host.argtypes = [vp, vp, i32, i32, i32, i32, i64, i32, i32]
    L.dvm_orb_sync.argtypes = [vp]
    L.dvm_orb_result_device.argtypes = [vp, i32, vp, vp, vp, vp]
    L.dvm_orb_copy_result.argtypes = [vp, i32, vp, vp, vp]
    L.dvm_orb_scale_factors_device.argtypes = [vp]
    L.dvm_orb_scale_factors_device.restype = vp
    L.dvm_orb_download.argtypes = [vp, i32, vp, vp, i32, vp, vp]
    L.dvm_orb_pyramid.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    L.dvm_orb_debug_level.argtypes = [vp, i32, i32, i32, vp]
    L.dvm_orb_debug_blurred.argtypes = [vp, i32, i32, vp]
    L.dvm_orb_debug_candidates.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp]
    L.dvm_orb_debug_level_keypoints.argtypes = [vp, i32, i32, vp, i32, vp]
    L.dvm_orb_profiling.argtypes = [vp, i32]
    L.dvm_orb_profile_get.argtypes = [vp, C.c_char_p, vp, vp]
    L.dvm_orb_profile_reset.argtypes = [vp]
    L.dvm_orb_stream.argtypes = [vp]
    L.dvm_orb_stream.restype = vp
    L.dvm_hamming_matrix.argtypes = [vp, i32, vp, i32, vp, i32, vp]
    L.dvm_frame_create.argtypes = [i32, i32, i32, C.POINTER(vp)]
    L.dvm_frame_destroy.argtypes = [vp]
    L.dvm_frame_destroy.restype = None
    L.dvm_frame_build.argtypes = [vp, i32, vp, vp, i32, vp, f32, f32, f32, f32, i32, vp]
    L.dvm_frame_build_batch.argtypes = [vp, i32, i32, vp, i64, vp, i64, vp, f32, f32, f32, f32, vp]
    L.dvm_match_window.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp]
    L.dvm_is_in_frustum.argtypes = [C.POINTER(FrustumFrame), vp, vp, vp, vp, i32, f32, vp, i32, vp]
    L.dvm_undistort_keypoints.argtypes = [vp, vp, vp, i32, i32, vp]
    L.dvm_image_bounds.argtypes = [vp, i32, i32, vp]
    L.dvm_match_lists.argtypes = [vp, i32, vp, i32, vp, vp, vp, i32, vp]
    L.dvm_match_frames_batch.argtypes = [vp, i32, i32, vp, i64, vp, i64, vp, vp, vp, vp, i32, f32, vp, i32, vp, i64,
                                         vp, vp]
    L.dvm_ba_create.argtypes = [i32, C.POINTER(vp)]
    L.dvm_ba_destroy.argtypes = [vp]
    L.dvm_ba_destroy.restype = None
    L.dvm_ba_set_problem.argtypes = [vp, vp, vp, i32, vp, i32, vp, i32, C.POINTER(BaCamera)]
    L.dvm_ba_optimize.argtypes = [vp, i32, vp, C.POINTER(BaStats)]
    L.dvm_ba_get_result.argtypes = [vp, vp, vp]
    L.dvm_ba_edge_chi2.argtypes = [vp, vp, vp]
    L.dvm_ba_stream.argtypes = [vp]
    L.dvm_ba_stream.restype = vp
    L.dvm_optimize_sim3.argtypes = [i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, C.c_double, vp, vp]
    L.dvm_pose_optimize.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, C.POINTER(BaCamera), vp, vp, vp]
    _LIB = L
    return L


def check(rc):
    if rc != DVM_OK:
        raise DvmError(rc, lib().dvm_last_error().decode(errors="replace"))


def device_count() -> int:
    return int(lib().dvm_device_count())


def _p(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))


class OrbExtractor:
    """Mirror of ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:47-91) over the C ABI."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, device=0, max_batch=1):
        self.L = lib()
        self.h = C.c_void_p()
        self.params = OrbParams(nfeatures, scale_factor, nlevels, ini_th, min_th)
        self.nlevels, self.nfeatures, self.max_batch = nlevels, nfeatures, max_batch
        check(self.L.dvm_orb_create(C.byref(self.params), device, max_batch, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.dvm_orb_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        nf = np.zeros(n, np.int32)
        check(self.L.dvm_orb_tables(self.h, _p(sc), _p(isc), _p(s2), _p(is2), _p(nf)))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, nfeat=nf)

    @property
    def cap(self):
        return self.nfeatures + 5 * self.nlevels + 8

    def last_result(self):
        """dvm_orb_last_result: a DeviceFrame naming the keypoints + descriptors of the last extract(), still in HBM."""
        r = DeviceFrame()
        f = self.L.dvm_orb_last_result
        f.restype = C.c_int32; f.argtypes = None
        check(f(self.h, C.byref(r)))
        return r

    def extract(self, img: np.ndarray, lap=(0, 1000)):
        """operator(): returns (n, keypoints, descriptors, monoIndex); n == -1 for an empty image."""
        if img is None or img.size == 0:
            rc = self.L.dvm_orb_extract(self.h, None, 0, 0, 0, lap[0], lap[1], None, None, 0, None, None)
            assert rc == -2, rc
            return -1, None, None, -1
        if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1:
            img = np.ascontiguousarray(img, np.uint8)  # row-strided views (stride > cols) pass through as they are
        kps = np.empty(self.cap, KP_DTYPE)          # (the call fills [0, n); the rest is never looked at)
        desc = np.empty((self.cap, 32), np.uint8)
        n, mono = C.c_int(0), C.c_int(0)
        rc = self.L.dvm_orb_extract(self.h, _p(img), img.shape[0], img.shape[1], img.strides[0], lap[0], lap[1],
                                    _p(kps), _p(desc), self.cap, C.byref(n), C.byref(mono))
        if rc == -3 and n.value > self.cap:   # DVM_ERR_CAPACITY: *n holds the size needed (small quotas on wide images)
            return self.download(0, cap=n.value)
        check(rc)
        return n.value, kps[:n.value], desc[:n.value], mono.value

    def extract_batch_host(self, imgs: np.ndarray, lap=(0, 1000)):
        imgs = np.ascontiguousarray(imgs, np.uint8)
        b, r, c = imgs.shape
        check(self.L.dvm_orb_extract_batch_host(self.h, _p(imgs), b, r, c, imgs.strides[1], imgs.strides[0], lap[0], lap[1]))

    def extract_batch_device(self, d_ptr: int, batch, rows, cols, stride=None, frame_stride=None, lap=(0, 1000)):
        stride = stride or cols
        frame_stride = frame_stride or rows * stride
        check(self.L.dvm_orb_extract_batch_device(self.h, C.c_void_p(d_ptr), batch, rows, cols, stride, frame_stride,
                                                  lap[0], lap[1]))

    def staging(self, batch, rows, cols) -> np.ndarray:
        """The handle's pinned input buffer as a (batch, rows, cols) uint8 array (dvm_orb_staging); waits for the previous copy."""
        p = C.c_void_p()
        f = self.L.dvm_orb_staging
        f.restype = C.c_int32; f.argtypes = None
        check(f(self.h, C.c_int32(batch), C.c_int32(rows), C.c_int32(cols), C.byref(p)))
        buf = (C.c_uint8 * (batch * rows * cols)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(batch, rows, cols)

    def extract_staged(self, batch, rows, cols, lap=(0, 1000)):
        f = self.L.dvm_orb_extract_staged
        f.restype = C.c_int32; f.argtypes = None
        check(f(self.h, C.c_int32(batch), C.c_int32(rows), C.c_int32(cols), C.c_int32(lap[0]), C.c_int32(lap[1])))

    def sync(self):
        check(self.L.dvm_orb_sync(self.h))

    def download(self, frame=0, cap=None):
        cap = cap or self.cap
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n, mono = C.c_int(0), C.c_int(0)
        rc = self.L.dvm_orb_download(self.h, frame, _p(kps), _p(desc), cap, C.byref(n), C.byref(mono))
        if rc == -3 and n.value > cap:
            return self.download(frame, cap=n.value)
        check(rc)
        return n.value, kps[:n.value].copy(), desc[:n.value].copy(), mono.value

    def result_device(self, frame=0):
        k, d, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
        cap = C.c_int(0)
        check(self.L.dvm_orb_result_device(self.h, frame, C.byref(k), C.byref(d), C.byref(n), C.byref(cap)))
        return k.value, d.value, n.value, cap.value

    def copy_result(self, frame, d_kps, d_desc, d_n):
        check(self.L.dvm_orb_copy_result(self.h, frame, C.c_void_p(d_kps), C.c_void_p(d_desc), C.c_void_p(d_n)))

    def scale_factors_device(self):
        return self.L.dvm_orb_scale_factors_device(self.h)

    def pyramid(self, frame, level):
        ptr = C.c_void_p()
        r, c, s = C.c_int(0), C.c_int(0), C.c_int(0)
        check(self.L.dvm_orb_pyramid(self.h, frame, level, C.byref(ptr), C.byref(r), C.byref(c), C.byref(s)))
        return ptr.value, r.value, c.value, s.value

    def debug_level(self, frame, level, bordered=False):
        _, r, c, _ = self.pyramid(frame, level)
        out = np.zeros((r + 38, c + 38) if bordered else (r, c), np.uint8)
        check(self.L.dvm_orb_debug_level(self.h, frame, level, int(bordered), _p(out)))
        return out

    def debug_blurred(self, frame, level):
        _, r, c, _ = self.pyramid(frame, level)
        out = np.zeros((r, c), np.uint8)
        check(self.L.dvm_orb_debug_blurred(self.h, frame, level, _p(out)))
        return out

    def debug_candidates(self, frame, level, cap=300000):
        xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
        n = C.c_int(0)
        check(self.L.dvm_orb_debug_candidates(self.h, frame, level, _p(xs), _p(ys), _p(sc), cap, C.byref(n)))
        return xs[:n.value].copy(), ys[:n.value].copy(), sc[:n.value].copy()

    def debug_level_keypoints(self, frame, level, cap=20000):
        k = np.zeros(cap, KP_DTYPE)
        n = C.c_int(0)
        check(self.L.dvm_orb_debug_level_keypoints(self.h, frame, level, _p(k), cap, C.byref(n)))
        return k[:n.value].copy()

    def profiling(self, on=True):
        """True / 1: HIP events around every stage; 2: around the dominant kernel (k_fast_cells) only -- an event record is a marker
        packet between two kernels of the batch, a few microseconds of idle device each; False / 0: none."""
        check(self.L.dvm_orb_profiling(self.h, int(on)))

    def profile_reset(self):
        check(self.L.dvm_orb_profile_reset(self.h))

    def profile_get(self, name):
        ms, cnt = C.c_double(0), C.c_int64(0)
        rc = self.L.dvm_orb_profile_get(self.h, name.encode(), C.byref(ms), C.byref(cnt))
        return (ms.value, cnt.value) if rc == 0 else (0.0, 0)

    def stream(self):
        return self.L.dvm_orb_stream(self.h)


def hamming_matrix(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """All-pairs ORBmatcher::DescriptorDistance (host arrays in, host matrix out)."""
    A = np.ascontiguousarray(A, np.uint8).reshape(-1, 32)
    B = np.ascontiguousarray(B, np.uint8).reshape(-1, 32)
    D = np.zeros((len(A), len(B)), np.uint16)
    check(lib().dvm_hamming_matrix(_p(A), len(A), _p(B), len(B), _p(D), 0, None))
    return D


def match_lists(tdesc: np.ndarray, qdesc: np.ndarray, offsets: np.ndarray, cand: np.ndarray) -> np.ndarray:
    """Best / second best of query q over train indices cand[offsets[q]:offsets[q+1]] (list order = tie order)."""
    tdesc = np.ascontiguousarray(tdesc, np.uint8).reshape(-1, 32)
    qdesc = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    cand = np.ascontiguousarray(cand, np.int32)
    out = np.zeros(len(qdesc), MATCH_DTYPE)
    check(lib().dvm_match_lists(_p(tdesc), len(tdesc), _p(qdesc), len(qdesc), _p(offsets), _p(cand), _p(out), 0, None))
    return out


class OrbPool:
    """dvm_orb_pool_*: one extractor shared by several agents' threads; frames arriving together are extracted as one batch."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, max_batch=32, window_us=-1, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        p = OrbParams(nfeatures, scale_factor, nlevels, ini_th, min_th)
        f = self.L.dvm_orb_pool_create
        f.restype = C.c_int32; f.argtypes = None
        check(f(C.byref(p), C.c_int32(device), C.c_int32(max_batch), C.c_int32(window_us), C.byref(self.h)))
        self.cap = 4 * max(nfeatures, 1) + 256   # (small quotas on wide images keep up to 4 * nIni keypoints per level)
        self._x = self.L.dvm_orb_pool_extract
        self._x.restype = C.c_int32; self._x.argtypes = None

    def extract(self, img, lap=(0, 1000)):
        """One frame (blocking, from any thread): (n, keypoints, descriptors, monoIndex, frames in the batch this call ran in)."""
        if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1:
            img = np.ascontiguousarray(img, np.uint8)
        kps = np.empty(self.cap, KP_DTYPE)
        desc = np.empty((self.cap, 32), np.uint8)
        n, mono, bs = C.c_int(0), C.c_int(0), C.c_int(0)
        check(self._x(self.h, _p(img), C.c_int32(img.shape[0]), C.c_int32(img.shape[1]), C.c_int32(img.strides[0]), C.c_int32(lap[0]), C.c_int32(lap[1]),
                      _p(kps), _p(desc), C.c_int32(self.cap), C.byref(n), C.byref(mono), C.byref(bs)))
        return n.value, kps[:n.value], desc[:n.value], mono.value, bs.value

    def close(self):
        if self.h:
            f = self.L.dvm_orb_pool_destroy
            f.restype = None; f.argtypes = None
            f(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PosePool:
    """dvm_pose_pool_*: PoseOptimization of one frame per call, from any thread; calls arriving together run as one launch."""

    def __init__(self, max_batch=32, window_us=-1, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        f = self.L.dvm_pose_pool_create
        f.restype = C.c_int32; f.argtypes = None
        check(f(C.c_int32(device), C.c_int32(max_batch), C.c_int32(window_us), C.byref(self.h)))
        self._x = self.L.dvm_pose_pool_optimize
        self._x.restype = C.c_int32; self._x.argtypes = None

    def optimize(self, pose, Xw, obs, inv_sigma2, intrinsics):
        """One frame: pose [7], Xw [n,3], obs [n,2], inv_sigma2 [n].  Returns (pose [7], outlier [n] uint8, n_inliers, frames in the launch)."""
        pose = np.ascontiguousarray(pose, np.float64).reshape(7)
        Xw = np.ascontiguousarray(Xw, np.float64).reshape(-1, 3)
        n = len(Xw)
        obs = np.ascontiguousarray(obs, np.float64).reshape(n, 2)
        w = np.ascontiguousarray(inv_sigma2, np.float64).reshape(n)
        cam = BaCamera(*[float(v) for v in intrinsics], 0.0)
        out = np.zeros(7, np.float64); outl = np.zeros(max(n, 1), np.uint8)
        ninl, bs = C.c_int32(0), C.c_int(0)
        check(self._x(self.h, _p(pose), _p(Xw), _p(obs), _p(w), C.c_int32(n), C.byref(cam), _p(out), _p(outl), C.byref(ninl), C.byref(bs)))
        return out, outl[:n], ninl.value, bs.value

    def close(self):
        if self.h:
            f = self.L.dvm_pose_pool_destroy
            f.restype = None; f.argtypes = None
            f(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MatchPool:
    """dvm_match_pool_*: the grid build + ranked window search of SearchByProjection(Cur, Last) for several threads at once.
    use(): route this process's search_by_projection_frames calls through it (dvmh_set_match_pool); release() / close() undo that."""

    def __init__(self, max_batch=32, kp_cap=2048, q_cap=2048, window_us=-1, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        f = self.L.dvm_match_pool_create
        f.restype = C.c_int32; f.argtypes = None
        check(f(C.c_int32(device), C.c_int32(max_batch), C.c_int32(kp_cap), C.c_int32(q_cap), C.c_int32(window_us), C.byref(self.h)))

    def use(self):
        f = host_lib().dvmh_set_match_pool
        f.restype = None; f.argtypes = [C.c_void_p]
        f(self.h)

    def release(self):
        f = host_lib().dvmh_set_match_pool
        f.restype = None; f.argtypes = [C.c_void_p]
        f(None)

    def close(self):
        if self.h:
            self.release()
            f = self.L.dvm_match_pool_destroy
            f.restype = None; f.argtypes = None
            f(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FrameGrid:
    """Frame's feature grid + windowed search (reference Frame.cc:443-506,712-782; ORBmatcher.cc:70-115)."""

    def __init__(self, capacity=2048, slots=1, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        self.capacity, self.slots = capacity, slots
        check(self.L.dvm_frame_create(device, capacity, slots, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.dvm_frame_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def build(self, kps: np.ndarray, desc: np.ndarray, bounds=(0.0, 640.0, 0.0, 480.0), slot=0):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        check(self.L.dvm_frame_build(self.h, slot, _p(kps), _p(desc), len(kps), None, *bounds, 0, None))

    def overflows(self):
        """Sticky count of device-side keypoint counts that exceeded the capacity (raises DvmError if non-zero)."""
        n = C.c_int32(0)
        self.L.dvm_frame_overflows.restype = C.c_int32; self.L.dvm_frame_overflows.argtypes = [C.c_void_p, C.c_void_p]
        check(self.L.dvm_frame_overflows(self.h, C.byref(n)))
        return n.value

    def build_batch_device(self, first_slot, count, d_kps, kps_stride, d_desc, desc_stride, d_n, bounds, stream=None):
        check(self.L.dvm_frame_build_batch(self.h, first_slot, count, C.c_void_p(d_kps), kps_stride, C.c_void_p(d_desc),
                                           desc_stride, C.c_void_p(d_n), *bounds, C.c_void_p(stream or 0)))

    def match_window(self, qdesc, qx, qy, qr, qmin, qmax, skip=None, slot=0, top2=False):
        qdesc = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
        nq = len(qdesc)
        qx, qy, qr = (np.ascontiguousarray(a, np.float32) for a in (qx, qy, qr))
        qmin, qmax = (np.ascontiguousarray(a, np.int32) for a in (qmin, qmax))
        out = np.zeros(nq, MATCH_DTYPE)
        sk = None
        if skip is not None:
            sk = np.zeros(self.capacity, np.uint8)
            sk[:len(skip)] = skip
        if not top2:
            check(self.L.dvm_match_window(self.h, slot, _p(sk), _p(qdesc), _p(qx), _p(qy), _p(qr), _p(qmin), _p(qmax), nq,
                                          None, _p(out), 0, None))
            return out
        second = np.full(nq, -7, np.int32)
        self.L.dvm_match_window_top2.restype = C.c_int
        self.L.dvm_match_window_top2.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 7 + [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p]
        check(self.L.dvm_match_window_top2(self.h, slot, _p(sk), _p(qdesc), _p(qx), _p(qy), _p(qr), _p(qmin), _p(qmax), nq,
                                           None, _p(out), _p(second), 0, None))
        return out, second

    def match_window_ranked(self, qdesc, qx, qy, qr, qmin, qmax, skip=None, slot=0):
        """dvm_match_window_ranked: the four best candidates per query, best first.  Returns (idx [nq,4] int32, -1 past the end; dist [nq,4])."""
        qdesc = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
        nq = len(qdesc)
        qx, qy, qr = (np.ascontiguousarray(a, np.float32) for a in (qx, qy, qr))
        qmin, qmax = (np.ascontiguousarray(a, np.int32) for a in (qmin, qmax))
        ranked = np.zeros((nq, 4), np.uint32)
        sk = None
        if skip is not None:
            sk = np.zeros(self.capacity, np.uint8)
            sk[:len(skip)] = skip
        f = self.L.dvm_match_window_ranked
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 7 + [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        check(f(self.h, slot, _p(sk), _p(qdesc), _p(qx), _p(qy), _p(qr), _p(qmin), _p(qmax), nq, _p(ranked), 0, None))
        dist = (ranked >> 16).astype(np.int32)
        idx = np.where(dist < 256, (ranked & 0xFFFF).astype(np.int32), -1)
        return idx, dist

    def match_frames_batch(self, first_slot, count, d_kps, kps_stride, d_desc, desc_stride, d_n, carry, cap, th,
                           d_scale, nlevels, d_out, out_stride, d_nq_out=None, stream=None):
        ck, cd, cn = carry if carry else (0, 0, 0)
        check(self.L.dvm_match_frames_batch(self.h, first_slot, count, C.c_void_p(d_kps), kps_stride, C.c_void_p(d_desc),
                                            desc_stride, C.c_void_p(d_n), C.c_void_p(ck), C.c_void_p(cd), C.c_void_p(cn),
                                            cap, th, C.c_void_p(d_scale), nlevels, C.c_void_p(d_out), out_stride,
                                            C.c_void_p(d_nq_out or 0), C.c_void_p(stream or 0)))


def make_edges(edge_pose, edge_point, obs, inv_sigma2) -> np.ndarray:
    e = np.zeros(len(edge_pose), BA_EDGE_DTYPE)
    e["pose"], e["point"] = edge_pose, edge_point
    e["u"], e["v"] = obs[:, 0], obs[:, 1]
    e["inv_sigma2"] = inv_sigma2
    return e


BA_EDGE_ACTIVE, BA_EDGE_ROBUST = 1, 2     # DVM_BA_EDGE_ACTIVE / DVM_BA_EDGE_ROBUST


class BundleAdjuster:
    """Optimizer::BundleAdjustment / LocalBundleAdjustment numerics (reference Optimizer.cc:55-356,1030-1387)."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        check(self.L.dvm_ba_create(device, C.byref(self.h)))
        self.P = self.Lm = self.E = 0

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.dvm_ba_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def set_problem(self, poses, fixed, points, edges, intrinsics, huber_delta):
        poses = np.ascontiguousarray(poses, np.float64)
        points = np.ascontiguousarray(points, np.float64)
        fixed = np.ascontiguousarray(fixed, np.uint8)
        edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
        cam = BaCamera(*[float(v) for v in intrinsics], float(huber_delta))
        self.P, self.Lm, self.E = len(poses), len(points), len(edges)
        check(self.L.dvm_ba_set_problem(self.h, _p(poses), _p(fixed), self.P, _p(points), self.Lm, _p(edges), self.E,
                                        C.byref(cam)))

    def set_problem_sharded(self, poses, fixed, points, edges, intrinsics, huber_delta, rank, world):
        """BASELINE config 5: the whole problem on every rank, the observations of the landmarks `point % world == rank`
        evaluated here (dvm_ba_set_problem_sharded).  Needs set_allreduce() before optimize()."""
        poses = np.ascontiguousarray(poses, np.float64)
        points = np.ascontiguousarray(points, np.float64)
        fixed = np.ascontiguousarray(fixed, np.uint8)
        edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
        cam = BaCamera(*[float(v) for v in intrinsics], float(huber_delta))
        self.P, self.Lm = len(poses), len(points)
        self.E = int((edges["point"] % world == rank).sum())
        f = self.L.dvm_ba_set_problem_sharded
        f.restype = C.c_int32; f.argtypes = None
        check(f(self.h, _p(poses), _p(fixed), C.c_int32(self.P), _p(points), C.c_int32(self.Lm), _p(edges), C.c_int32(len(edges)),
                C.byref(cam), C.c_int32(rank), C.c_int32(world)))

    def schedule_info(self):
        out = (C.c_int64 * 12)()
        f = self.L.dvm_ba_schedule_info
        f.restype = C.c_int32; f.argtypes = None
        check(f(self.h, out))
        keys = ("levels", "columns", "strips", "targets", "products", "products_on_diagonal_targets", "nz_tiles", "ldS", "free_cameras",
                "nz_blocks", "edges", "tiles_per_side")
        return dict(zip(keys, [int(v) for v in out]))

    def solve_info(self):
        out = (C.c_int64 * 6)()
        f = self.L.dvm_ba_solve_info
        f.restype = C.c_int32; f.argtypes = None
        check(f(self.h, out))
        return dict(form="flow" if out[0] else "levels", flow_tasks=int(out[1]), chains=int(out[2]), workgroups=int(out[3]), kept_landmarks=int(out[4]),
                    camera_tiles=int(out[5]))

    def profile(self, enable=-1):
        ms = (C.c_double * 4)(); tr = C.c_int32(0); it = C.c_int32(0)
        f = self.L.dvm_ba_profile
        f.restype = C.c_int32; f.argtypes = None
        check(f(self.h, C.c_int32(enable), ms, C.byref(tr), C.byref(it)))
        return dict(ms_linearise=ms[0], ms_schur=ms[1], ms_cholesky_solve=ms[2], ms_update_chi2=ms[3], trials=tr.value, iterations=it.value)

    def allreduce_doubles(self):
        f = self.L.dvm_ba_allreduce_doubles
        f.restype = C.c_int64; f.argtypes = [C.c_void_p]
        return int(f(self.h))

    ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p)

    def set_allreduce(self, fn, d_buf, cap_doubles):
        """fn(buf_address, n, on_host, op, stream) -> 0 on success: in-place all-reduce of n doubles (device buffer = d_buf, or a
        host address); op 0 sum, 1 max."""
        self._ar_cb = self.ALLREDUCE_FN(lambda ctx, buf, n, on_host, op, stream: int(fn(buf, n, on_host, op, stream)))
        f = self.L.dvm_ba_set_allreduce
        f.restype = C.c_int32; f.argtypes = None
        check(f(self.h, self._ar_cb, None, C.c_void_p(d_buf), C.c_int64(cap_doubles)))

    def optimize(self, iterations, stop_flag=None):
        st = BaStats()
        check(self.L.dvm_ba_optimize(self.h, iterations, _p(stop_flag) if stop_flag is not None else None, C.byref(st)))
        n = min(st.iterations, 64)
        return dict(iterations=st.iterations, total_trials=st.total_trials, stop_reason=st.stop_reason,
                    chi2_initial=st.chi2_initial, chi2_final=st.chi2_final, lambda_final=st.lambda_final,
                    trials=st.trials_per_iter[:n], chi2=st.chi2_per_iter[:n], lam=st.lambda_per_iter[:n],   # (a ctypes array slice is a list)
                    ms_structure=st.ms_structure, ms_optimize=st.ms_optimize, spec_trials=st.spec_trials, spec_kept=st.spec_kept)

    def set_edge_flags(self, flags):
        """dvm_ba_set_edge_flags: per edge BA_EDGE_ACTIVE | BA_EDGE_ROBUST (None: every edge active and robust again)."""
        f = self.L.dvm_ba_set_edge_flags
        f.restype = C.c_int32; f.argtypes = None
        if flags is None:
            check(f(self.h, None))
        else:
            fl = np.ascontiguousarray(flags, np.uint8)
            assert len(fl) == self.E
            check(f(self.h, _p(fl)))

    def result(self):
        poses = np.zeros((self.P, 7), np.float64)
        points = np.zeros((self.Lm, 3), np.float64)
        check(self.L.dvm_ba_get_result(self.h, _p(poses), _p(points)))
        return poses, points

    def edge_chi2(self):
        chi = np.zeros(self.E, np.float64)
        dp = np.zeros(self.E, np.uint8)
        check(self.L.dvm_ba_edge_chi2(self.h, _p(chi), _p(dp)))
        return chi, dp


class BaWindow(C.Structure):
    """dvm_ba_window (include/dvmslam_hip.h)"""
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32), ("iterations", C.c_int32),
                ("poses", C.c_void_p), ("fixed", C.c_void_p), ("points", C.c_void_p), ("edges", C.c_void_p), ("cam", BaCamera),
                ("poses_out", C.c_void_p), ("points_out", C.c_void_p), ("edge_chi2_out", C.c_void_p), ("depth_positive_out", C.c_void_p)]


def ba_optimize_batch(problems, device=0, threads=0, stop_flag=None):
    """dvm_ba_optimize_batch: the same K problems, each on the general solver, up to `threads` of them concurrently (0: the library's default).
    Same arguments and return value as ba_optimize_windows."""
    return ba_optimize_windows(problems, device, stop_flag, _threads=int(threads), _batch=True)


class BaWindowBatch:
    """K bundle-adjustment windows marshalled ONCE into the dvm_ba_window array of the C ABI (input arrays referenced, output arrays allocated):
    what a C++ agent node holds anyway.  run() is then the bare library call -- the per-call Python work of ba_optimize_windows (~35 us per
    window) stays out of a timed loop."""

    def __init__(self, problems):
        K = len(problems)
        self.K = K
        self.wins = (BaWindow * max(K, 1))()
        self.stats = (BaStats * max(K, 1))()
        self.keep, self.outs = [], []
        for k, pr in enumerate(problems):
            poses = np.ascontiguousarray(pr["poses"], np.float64); points = np.ascontiguousarray(pr["points"], np.float64)
            fixed = np.ascontiguousarray(pr["fixed"], np.uint8); edges = np.ascontiguousarray(pr["edges"], BA_EDGE_DTYPE)
            o = dict(poses=np.zeros_like(poses), points=np.zeros_like(points), edge_chi2=np.zeros(len(edges), np.float64),
                     depth_positive=np.zeros(len(edges), np.uint8))
            self.keep.append((poses, points, fixed, edges)); self.outs.append(o)
            w = self.wins[k]
            w.n_poses, w.n_points, w.n_edges, w.iterations = len(poses), len(points), len(edges), int(pr["iterations"])
            w.poses, w.fixed, w.points, w.edges = poses.ctypes.data, fixed.ctypes.data, points.ctypes.data, edges.ctypes.data
            w.cam = BaCamera(*[float(v) for v in pr["intrinsics"]], float(pr["huber_delta"]))
            w.poses_out, w.points_out, w.edge_chi2_out, w.depth_positive_out = (o["poses"].ctypes.data, o["points"].ctypes.data, o["edge_chi2"].ctypes.data,
                                                                                o["depth_positive"].ctypes.data)

    def run(self, device=0, stop_flag=None, fast=False, _threads=0, _batch=False, collect=True):
        """One library call over the K windows; returns the per-window dicts (the output arrays are this object's: copy what must outlive the next run).
        collect=False: the bare call -- results() builds the dicts (ctypes slices of the statistics: ~5 us a window) when they are wanted."""
        sf = _p(stop_flag) if stop_flag is not None else None
        if _batch:
            f = lib().dvm_ba_optimize_batch
            f.restype = C.c_int32; f.argtypes = None
            check(f(C.c_int32(device), self.wins, C.c_int32(self.K), C.c_int32(_threads), sf, self.stats))
        else:
            f = lib().dvm_ba_optimize_windows_fast if fast else lib().dvm_ba_optimize_windows
            f.restype = C.c_int32; f.argtypes = None
            check(f(C.c_int32(device), self.wins, C.c_int32(self.K), sf, self.stats))
        return self.results() if collect else None

    def results(self):
        for k, o in enumerate(self.outs):
            st = self.stats[k]
            n = min(st.iterations, 64)
            o["stats"] = dict(iterations=st.iterations, total_trials=st.total_trials, stop_reason=st.stop_reason, chi2_initial=st.chi2_initial,
                              chi2_final=st.chi2_final, lambda_final=st.lambda_final, trials=st.trials_per_iter[:n], chi2=st.chi2_per_iter[:n],
                              lam=st.lambda_per_iter[:n], ms_structure=st.ms_structure, ms_optimize=st.ms_optimize, kernel_us=st.kernel_us)
        return self.outs


class BaPool:
    """dvm_ba_pool_*: blocking one-window calls from any number of threads, batched behind the boundary into launches of the cluster form."""

    def __init__(self, device=0, max_batch=32, window_us=-1):
        self.p = C.c_void_p()
        f = lib().dvm_ba_pool_create
        f.restype = C.c_int32; f.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        check(f(device, max_batch, window_us, C.byref(self.p)))

    def close(self):
        if getattr(self, "p", None) and self.p.value:
            f = lib().dvm_ba_pool_destroy
            f.restype = None; f.argtypes = [C.c_void_p]
            f(self.p)
            self.p = C.c_void_p()

    __del__ = close

    def optimize(self, batch: "BaWindowBatch"):
        """batch: a BaWindowBatch of ONE window (marshalled by the caller's thread).  Returns (result dict, windows in the launch)."""
        n = C.c_int32(0)
        f = lib().dvm_ba_pool_optimize
        f.restype = C.c_int32; f.argtypes = None
        check(f(self.p, batch.wins, batch.stats, C.byref(n)))
        return batch.results()[0], n.value


def ba_optimize_windows(problems, device=0, stop_flag=None, _threads=0, _batch=False, fast=False):
    """dvm_ba_optimize_windows: K independent bundle adjustments in one launch.  problems: dicts with poses [P,7], fixed [P], points [L,3],
    edges (BA_EDGE_DTYPE), intrinsics (fx, fy, cx, cy), huber_delta, iterations.  Returns one dict per window: poses, points, edge_chi2,
    depth_positive, stats (the keys of BundleAdjuster.optimize).  fast=True: dvm_ba_optimize_windows_fast (tree sums in a fixed order)."""
    return BaWindowBatch(problems).run(device, stop_flag, fast, _threads, _batch)


def f64_spec_eval(x, device=0):
    """csrc/f64_spec.h on the device: returns (sin, cos, cube) of the doubles in x."""
    x = np.ascontiguousarray(x, np.float64)
    out = np.zeros(3 * len(x), np.float64)
    f = lib().dvm_f64_spec_eval
    f.restype = C.c_int32; f.argtypes = None
    check(f(C.c_int32(device), _p(x), C.c_int32(len(x)), _p(out)))
    return out[:len(x)], out[len(x):2 * len(x)], out[2 * len(x):]


def pose_optimize(poses, Xw, obs, inv_sigma2, n, intrinsics, device=0):
    """Optimizer::PoseOptimization for a batch of frames.  poses [B,7]; Xw [B,S,3]; obs [B,S,2]; inv_sigma2 [B,S];
    n [B].  Returns (poses [B,7], outlier [B,S] uint8, n_inliers [B])."""
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
    B = len(poses)
    Xw = np.ascontiguousarray(Xw, np.float64).reshape(B, -1, 3)
    S = Xw.shape[1]
    obs = np.ascontiguousarray(obs, np.float64).reshape(B, S, 2)
    inv_sigma2 = np.ascontiguousarray(inv_sigma2, np.float64).reshape(B, S)
    n = np.ascontiguousarray(n, np.int32).reshape(B)
    cam = BaCamera(*[float(v) for v in intrinsics], 0.0)
    out = np.zeros((B, 7), np.float64)
    outl = np.zeros((B, S), np.uint8)
    nin = np.zeros(B, np.int32)
    check(lib().dvm_pose_optimize(device, _p(poses), _p(Xw), _p(obs), _p(inv_sigma2), _p(n), S, B, C.byref(cam), _p(out),
                                  _p(outl), _p(nin)))
    return out, outl, nin


def is_in_frustum(F: "FrustumFrame", P, normal, min_dist, max_dist, viewing_cos_limit=0.5):
    """Frame::isInFrustum for an array of map points; returns a TRACK_DTYPE array."""
    P = np.ascontiguousarray(P, np.float32); normal = np.ascontiguousarray(normal, np.float32)
    min_dist = np.ascontiguousarray(min_dist, np.float32); max_dist = np.ascontiguousarray(max_dist, np.float32)
    out = np.zeros(len(P), TRACK_DTYPE)
    check(lib().dvm_is_in_frustum(C.byref(F), _p(P), _p(normal), _p(min_dist), _p(max_dist), len(P), float(viewing_cos_limit),
                                  _p(out), 0, None))
    return out


def triangulate_matches(K1, K2, T1w, T2w, Ow1, Ow2, kps1, kps2, pairs, sigma2_1, sigma2_2, sf1, sf2, ratio_factor,
                        cos_parallax_max=0.9998, far_points=False, th_far=0.0):
    """LocalMapping::CreateNewMapPoints' per-match geometry for one neighbour keyframe (dvm_triangulate_matches).
    Returns (x3D[n,3] float32, status[n] int32)."""
    P = TriPair()
    P.cos_parallax_max = float(cos_parallax_max)
    for name, v, k in (("K1", K1, 4), ("K2", K2, 4), ("T1w", T1w, 12), ("T2w", T2w, 12), ("Ow1", Ow1, 3), ("Ow2", Ow2, 3)):
        setattr(P, name, (C.c_float * k)(*np.asarray(v, np.float32).reshape(-1)))
    f = [np.ascontiguousarray(x, np.float32) for x in (sigma2_1, sigma2_2, sf1, sf2)]
    P.ratio_factor = float(ratio_factor); P.th_far = float(th_far); P.far_points = int(far_points); P.n_levels = len(f[0])
    k1 = np.ascontiguousarray(kps1); k2 = np.ascontiguousarray(kps2)
    pr = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    n = len(pr)
    X = np.zeros((n, 3), np.float32); st = np.zeros(n, np.int32)
    L = lib()
    L.dvm_triangulate_matches.restype = C.c_int
    L.dvm_triangulate_matches.argtypes = [C.POINTER(TriPair), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p]
    check(L.dvm_triangulate_matches(C.byref(P), _p(k1), len(k1), _p(k2), len(k2), _p(pr), n, _p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]), _p(X), _p(st), 0, None))
    return X, st


def undistort_keypoints(cam, kps, d_in=None, d_out=None, n=None, stream=None):
    """Frame::UndistortKeyPoints: cam = (fx, fy, cx, cy, k1, k2, p1, p2, k3) float32; kps a KP_DTYPE array (returns the
    undistorted copy), or device pointers d_in / d_out of n keypoints (asynchronous on `stream`)."""
    cam = np.ascontiguousarray(cam, np.float32)
    assert cam.shape == (9,)
    if d_in is not None:
        check(lib().dvm_undistort_keypoints(_p(cam), d_in, d_out, int(n), 1, stream))
        return None
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    out = np.zeros_like(kps)
    check(lib().dvm_undistort_keypoints(_p(cam), _p(kps), _p(out), len(kps), 0, None))
    return out


def image_bounds(cam, cols, rows):
    """Frame::ComputeImageBounds -> float32 (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    cam = np.ascontiguousarray(cam, np.float32)
    out = np.zeros(4, np.float32)
    check(lib().dvm_image_bounds(_p(cam), int(cols), int(rows), _p(out)))
    return out


def optimize_sim3(S12, fix_scale, P1c, P2c, obs1, obs2, w1, w2, K1, K2, th2, device=0):
    """Optimizer::OptimizeSim3 numerics on the device.  Returns (S12[8], inlier mask, nIn)."""
    S = np.array(S12, np.float64, copy=True)
    arrs = [np.ascontiguousarray(a, np.float64) for a in (P1c, P2c, obs1, obs2, w1, w2, K1, K2)]
    n = len(arrs[0])
    inl = np.zeros(n, np.uint8)
    nin = C.c_int32(0)
    check(lib().dvm_optimize_sim3(device, _p(S), int(fix_scale), *[_p(a) for a in arrs[:6]], n, _p(arrs[6]), _p(arrs[7]),
                                  float(th2), _p(inl), C.byref(nin)))
    return S, inl, nin.value


# ---- host C++ mirror (dvm_slam_amd/host, libdvmslam_host.so): ORBmatcher on plain structs over the C ABI
HOST_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libdvmslam_host.so")
MAP_POINT_DTYPE = np.dtype([("pos", "<f4", (3,)), ("desc", "u1", (32,)), ("n_obs", "<i4")])
_HOST = None


def host_lib():
    global _HOST
    if _HOST is None:
        lib()  # HIP library (and torch's runtime) first
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} not built (make -C dvm_slam_amd/host)")
        _HOST = C.CDLL(HOST_LIB_PATH)
        vp = C.c_void_p
        _HOST.dvmh_search_by_projection_frames.restype = C.c_int32
        _HOST.dvmh_search_by_projection_frames.argtypes = [C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, C.c_int32,
                                                           C.c_int32, vp, vp, vp, vp, C.c_float, C.c_int32, vp]
    return _HOST


class DeviceFrame(C.Structure):
    """dvm_device_frame (include/dvmslam_hip.h): the extractor's last single-frame result, still in HBM"""
    _fields_ = [("d_kps", C.c_void_p), ("d_desc", C.c_void_p), ("n", C.c_int32), ("device", C.c_int32), ("handle_id", C.c_uint64), ("serial", C.c_uint64)]


def search_by_projection_frames(kps_c, desc_c, mp_c, Tcw, K, bounds, scale_factors, kps_l, mp_l, outlier_l, mps, th,
                                check_ori=True, device=0, dev_c=None):
    """dvm_host::ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono=true) -- reference
    ORBmatcher.cc:1553-1748.  Tcw: CurrentFrame.GetPose() as 7 floats (qx, qy, qz, qw, t).  Returns (nmatches, updated mvpMapPoints of the current frame, #host re-queries)."""
    H = host_lib()
    kps_c = np.ascontiguousarray(kps_c, KP_DTYPE); kps_l = np.ascontiguousarray(kps_l, KP_DTYPE)
    desc_c = np.ascontiguousarray(desc_c, np.uint8)
    mp = np.array(mp_c, np.int32, copy=True)
    mp_l = np.ascontiguousarray(mp_l, np.int32)
    outl = None if outlier_l is None else np.ascontiguousarray(outlier_l, np.uint8)
    f = [np.ascontiguousarray(a, np.float32) for a in (Tcw, K, bounds, scale_factors)]
    assert f[0].shape == (7,)
    mps = np.ascontiguousarray(mps, MAP_POINT_DTYPE)
    req = C.c_int32(0)
    if dev_c is not None:       # the current frame's data is still in HBM (OrbExtractor.last_result()): returns a 4th value, 1 if the grid was built from it
        fn = H.dvmh_search_by_projection_frames_dev
        fn.restype = C.c_int32; fn.argtypes = None
        used = C.c_int32(0)
        n = fn(C.c_int32(device), C.c_int32(len(kps_c)), _p(kps_c), _p(desc_c), _p(mp), *[_p(a) for a in f], C.c_int32(len(f[3])), C.c_int32(len(kps_l)), _p(kps_l),
               _p(mp_l), None if outl is None else _p(outl), _p(mps), C.c_float(float(th)), C.c_int32(int(check_ori)), C.byref(req), C.byref(dev_c), C.byref(used))
        if n < 0:
            check(n)
        return n, mp, req.value, used.value
    n = H.dvmh_search_by_projection_frames(device, len(kps_c), _p(kps_c), _p(desc_c), _p(mp), *[_p(a) for a in f],
                                           len(f[3]), len(kps_l), _p(kps_l), _p(mp_l), None if outl is None else _p(outl),
                                           _p(mps), float(th), int(check_ori), C.byref(req))
    if n < 0:
        check(n)
    return n, mp, req.value


class TrackResult(C.Structure):
    """dvmh_track_result (include/dvmslam_host.h)"""
    _fields_ = [("n", C.c_int32), ("mono_index", C.c_int32), ("nmatches", C.c_int32), ("nmatches_search", C.c_int32), ("nmatches_map", C.c_int32),
                ("n_inliers", C.c_int32), ("wide_window", C.c_int32), ("replayed_on_host", C.c_int32), ("tracked", C.c_int32),
                ("Tcw", C.c_float * 7), ("pose", C.c_double * 7), ("n_requeried", C.c_int32), ("pad_", C.c_int32)]


class Distortion(C.Structure):
    """dvm_distortion (include/dvmslam_hip.h)"""
    _fields_ = [(k, C.c_float) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]


class Tracker:
    """dvm_tracker + dvmh_track_with_motion_model: Frame::Frame -> ExtractORB and Tracking::TrackWithMotionModel of one frame as ONE
    device chain behind one synchronisation (reference src/Frame.cc:371-411, src/Tracking.cc:2584-2667)."""

    def __init__(self, ext: "OrbExtractor", device=0, max_queries=None):
        self.L, self.H, self.ext, self.device = lib(), host_lib(), ext, device
        self.t = C.c_void_p()
        f = self.L.dvm_tracker_create
        f.restype = C.c_int32; f.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        check(f(device, ext.cap, max_queries or ext.cap, C.byref(self.t)))

    def close(self):
        if getattr(self, "t", None) and self.t.value:
            f = self.L.dvm_tracker_destroy
            f.restype = None; f.argtypes = [C.c_void_p]
            f(self.t)
            self.t = C.c_void_p()

    __del__ = close

    def track(self, img, Tcw_pred, K, bounds, scale_factors, inv_sigma2, kps_l, mp_l, outlier_l, mps, th=15.0, check_ori=True, dist=None,
              lap=(0, 1000)):
        """Tcw_pred: 7 floats (qx, qy, qz, qw, t).  Returns a dict: n, kps, desc, kps_un, mp (mvpMapPoints after the outlier drop), dropped,
        pose (7 doubles: t, q), and the counters of dvmh_track_result."""
        img = np.ascontiguousarray(img, np.uint8)
        cap = self.ext.cap
        kps = np.empty(cap, KP_DTYPE); kun = np.empty(cap, KP_DTYPE); desc = np.empty((cap, 32), np.uint8)
        mp = np.empty(cap, np.int32); dropped = np.empty(cap, np.int32)
        kps_l = np.ascontiguousarray(kps_l, KP_DTYPE); mp_l = np.ascontiguousarray(mp_l, np.int32)
        outl = None if outlier_l is None else np.ascontiguousarray(outlier_l, np.uint8)
        f4 = [np.ascontiguousarray(a, np.float32) for a in (Tcw_pred, K, bounds, scale_factors, inv_sigma2)]
        mps = np.ascontiguousarray(mps, MAP_POINT_DTYPE)
        res = TrackResult()
        fn = self.H.dvmh_track_with_motion_model
        fn.restype = C.c_int32; fn.argtypes = None
        vp = C.c_void_p
        rc = fn(self.t, self.ext.h, C.c_int32(self.device), vp(img.ctypes.data), C.c_int32(img.shape[0]), C.c_int32(img.shape[1]), C.c_int32(img.strides[0]),
                C.c_int32(lap[0]), C.c_int32(lap[1]), _p(f4[0]), _p(f4[1]), _p(f4[2]), None if dist is None else C.byref(dist), _p(f4[3]), _p(f4[4]),
                C.c_int32(len(f4[3])), C.c_int32(len(kps_l)), _p(kps_l), _p(mp_l), None if outl is None else _p(outl), _p(mps), C.c_float(float(th)),
                C.c_int32(int(check_ori)), _p(kps), _p(desc), C.c_int32(cap), _p(kun), _p(mp), _p(dropped), C.byref(res))
        check(rc)
        n = res.n
        out = {k: getattr(res, k) for k in ("n", "mono_index", "nmatches", "nmatches_search", "nmatches_map", "n_inliers", "wide_window",
                                            "replayed_on_host", "tracked", "n_requeried")}
        out.update(kps=kps[:n], kps_un=kun[:n], desc=desc[:n], mp=mp[:n], dropped=dropped[:n], pose=np.array(res.pose[:], np.float64),
                   Tcw=np.array(res.Tcw[:], np.float32))
        return out


class TrackIn(C.Structure):
    """dvmh_track_in (include/dvmslam_host.h)"""
    _fields_ = [("Tcw_pred", C.c_void_p), ("Nl", C.c_int32), ("kps_l", C.c_void_p), ("mp_l", C.c_void_p), ("outlier_l", C.c_void_p), ("mps", C.c_void_p)]


class TrackOut(C.Structure):
    """dvmh_track_out (include/dvmslam_host.h)"""
    _fields_ = [("kps", C.c_void_p), ("desc", C.c_void_p), ("cap", C.c_int32), ("kps_un", C.c_void_p), ("mp_c", C.c_void_p), ("dropped", C.c_void_p)]


class TrackerBatch:
    """dvm_tracker_create_batch + dvmh_track_with_motion_model_batch: the tracked frames of up to max_frames agents as ONE chain of batched
    launches.  ext: an OrbExtractor with max_batch >= max_frames."""

    def __init__(self, ext: "OrbExtractor", max_frames, device=0):
        self.L, self.H, self.ext, self.device, self.max_frames = lib(), host_lib(), ext, device, int(max_frames)
        self.t = C.c_void_p()
        f = self.L.dvm_tracker_create_batch
        f.restype = C.c_int32; f.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        check(f(device, self.max_frames, ext.cap, ext.cap, C.byref(self.t)))
        cap, B = ext.cap, self.max_frames
        self.kps = np.empty((B, cap), KP_DTYPE); self.kun = np.empty((B, cap), KP_DTYPE); self.desc = np.empty((B, cap, 32), np.uint8)
        self.mp = np.empty((B, cap), np.int32); self.dropped = np.empty((B, cap), np.int32)
        self.res = (TrackResult * B)()
        self.outs = (TrackOut * B)()
        for b in range(B):
            o = self.outs[b]
            o.kps, o.desc, o.cap, o.kps_un = self.kps[b].ctypes.data, self.desc[b].ctypes.data, cap, self.kun[b].ctypes.data
            o.mp_c, o.dropped = self.mp[b].ctypes.data, self.dropped[b].ctypes.data

    def close(self):
        if getattr(self, "t", None) and self.t.value:
            f = self.L.dvm_tracker_destroy
            f.restype = None; f.argtypes = [C.c_void_p]
            f(self.t)
            self.t = C.c_void_p()

    __del__ = close

    def prepare(self, Tcw_preds, lasts):
        """lasts: per agent (kps_l, mp_l, outlier_l or None, mps).  Returns the marshalled dvmh_track_in array (keep it alive with its arrays)."""
        n = len(lasts)
        ins = (TrackIn * n)()
        keep = []
        for b, (T, (kl, ml, ol, mps)) in enumerate(zip(Tcw_preds, lasts)):
            T = np.ascontiguousarray(T, np.float32); kl = np.ascontiguousarray(kl, KP_DTYPE); ml = np.ascontiguousarray(ml, np.int32)
            ol = None if ol is None else np.ascontiguousarray(ol, np.uint8); mps = np.ascontiguousarray(mps, MAP_POINT_DTYPE)
            keep.append((T, kl, ml, ol, mps))
            i = ins[b]
            i.Tcw_pred, i.Nl, i.kps_l, i.mp_l, i.outlier_l, i.mps = T.ctypes.data, len(kl), kl.ctypes.data, ml.ctypes.data, None if ol is None else ol.ctypes.data, mps.ctypes.data
        return ins, keep

    def track(self, imgs, ins, K, bounds, scale_factors, inv_sigma2, th=15.0, check_ori=True, lap=(0, 1000)):
        """imgs [count, rows, cols] u8; ins: prepare()'s array.  Returns one dict per frame (views into this object's arrays)."""
        imgs = np.ascontiguousarray(imgs, np.uint8)
        count = imgs.shape[0]
        f4 = [np.ascontiguousarray(a, np.float32) for a in (K, bounds, scale_factors, inv_sigma2)]
        fn = self.H.dvmh_track_with_motion_model_batch
        fn.restype = C.c_int32; fn.argtypes = None
        vp = C.c_void_p
        rc = fn(self.t, self.ext.h, C.c_int32(self.device), C.c_int32(count), vp(imgs.ctypes.data), C.c_int32(imgs.shape[1]), C.c_int32(imgs.shape[2]),
                C.c_int32(imgs.strides[1]), C.c_int64(imgs.strides[0]), C.c_int32(lap[0]), C.c_int32(lap[1]), _p(f4[0]), _p(f4[1]), _p(f4[2]), _p(f4[3]),
                C.c_int32(len(f4[2])), C.c_float(float(th)), C.c_int32(int(check_ori)), ins[0] if isinstance(ins, tuple) else ins, self.outs, self.res)
        check(rc)
        out = []
        for b in range(count):
            r = self.res[b]
            n = r.n
            d = {k: getattr(r, k) for k in ("n", "mono_index", "nmatches", "nmatches_search", "nmatches_map", "n_inliers", "wide_window", "replayed_on_host",
                                            "tracked", "n_requeried")}
            d.update(kps=self.kps[b, :n], kps_un=self.kun[b, :n], desc=self.desc[b, :n], mp=self.mp[b, :n], dropped=self.dropped[b, :n],
                     pose=np.array(r.pose[:], np.float64), Tcw=np.array(r.Tcw[:], np.float32))
            out.append(d)
        return out


TRACKED_POINT_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("depth", "<f4"), ("view_cos", "<f4"), ("level", "<i4"),
                                ("in_view", "u1"), ("bad", "u1"), ("pad", "u1", (2,)), ("desc", "u1", (32,)), ("n_obs", "<i4")])


def search_by_projection_points(kps, desc, mp, claimed_obs, bounds, scale_factors, pts, th, nnratio=0.8, far_points=False,
                                th_far=0.0, device=0):
    """dvm_host::ORBmatcher::SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints) -- reference
    ORBmatcher.cc:44-205.  Returns (nmatches, updated mvpMapPoints, #host re-queries)."""
    H = host_lib()
    vp = C.c_void_p
    H.dvmh_search_by_projection_points.restype = C.c_int32
    H.dvmh_search_by_projection_points.argtypes = [C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, C.c_int32, vp, C.c_int32,
                                                   C.c_float, C.c_float, C.c_int32, C.c_float, vp]
    kps = np.ascontiguousarray(kps, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    mpc = np.array(mp, np.int32, copy=True)
    co = np.ascontiguousarray(claimed_obs, np.uint8)
    b = np.ascontiguousarray(bounds, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
    pts = np.ascontiguousarray(pts, TRACKED_POINT_DTYPE)
    req = C.c_int32(0)
    n = H.dvmh_search_by_projection_points(device, len(kps), _p(kps), _p(desc), _p(mpc), _p(co), _p(b), _p(sf), len(sf), _p(pts),
                                           len(pts), float(th), float(nnratio), int(far_points), float(th_far), C.byref(req))
    if n < 0:
        check(n)
    return n, mpc, req.value


def distinctive_descriptors(desc, off):
    """dvm_distinctive_descriptors: MapPoint::ComputeDistinctiveDescriptors for a batch of map points (CSR offsets).
    Returns (best_idx, best_median)."""
    L = lib()
    L.dvm_distinctive_descriptors.restype = C.c_int32
    L.dvm_distinctive_descriptors.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); off = np.ascontiguousarray(off, np.int32)
    n = len(off) - 1
    bi = np.zeros(max(n, 1), np.int32); bm = np.zeros(max(n, 1), np.int32)
    dummy = np.zeros((1, 32), np.uint8)
    check(L.dvm_distinctive_descriptors(_p(desc if len(desc) else dummy), _p(off), n, _p(bi), _p(bm), 0, None))
    return bi[:n], bm[:n]


class Vocabulary:
    """dvm_vocab_*: DBoW2 vocabulary tree resident on the device, per-feature transform."""

    def __init__(self, voc, device=0):
        L = lib()
        vp, i32 = C.c_void_p, C.c_int32
        L.dvm_vocab_create.argtypes = [i32, i32, vp, vp, vp, vp, vp, i32, C.POINTER(vp)]
        L.dvm_vocab_destroy.argtypes = [vp]; L.dvm_vocab_destroy.restype = None
        L.dvm_vocab_transform.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, vp]
        self.L, self.h = L, vp()
        self.keep = [np.ascontiguousarray(voc[k]) for k in ("child_off", "children", "desc", "weight", "word_id")]
        check(L.dvm_vocab_create(device, voc["n_nodes"], *[_p(a) for a in self.keep], voc["L"], C.byref(self.h)))

    def transform(self, features, levelsup):
        f = np.ascontiguousarray(features, np.uint8).reshape(-1, 32)
        n = len(f)
        word = np.zeros(max(n, 1), np.int32); node = np.zeros(max(n, 1), np.int32); w = np.zeros(max(n, 1), np.float64)
        check(self.L.dvm_vocab_transform(self.h, _p(f) if n else None, n, levelsup, _p(word), _p(node), _p(w), 0, None))
        return word[:n], node[:n], w[:n]

    def close(self):
        if self.h:
            self.L.dvm_vocab_destroy(self.h)
            self.h = None


def vocab_transform_host(voc, features, levelsup, device=0):
    """dvm_host::ORBVocabulary::transform (host C++ mirror): returns dict(bow_ids, bow_vals, fv_nodes, fv_off, fv_feat)."""
    H = host_lib()
    vp, i32 = C.c_void_p, C.c_int32
    H.dvmh_vocab_transform.restype = i32
    H.dvmh_vocab_transform.argtypes = [i32, i32, vp, vp, vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    f = np.ascontiguousarray(features, np.uint8).reshape(-1, 32)
    n = len(f)
    bi = np.zeros(n + 1, np.int32); bv = np.zeros(n + 1, np.float64); fn = np.zeros(n + 1, np.int32)
    fo = np.zeros(n + 2, np.int32); ff = np.zeros(n + 1, np.int32)
    nb = C.c_int32(0); nf = C.c_int32(0)
    check(H.dvmh_vocab_transform(device, voc["n_nodes"], _p(voc["child_off"]), _p(voc["children"]), _p(voc["desc"]),
                                 _p(voc["weight"]), _p(voc["word_id"]), voc["L"], _p(f), n, levelsup, _p(bi), _p(bv),
                                 C.byref(nb), _p(fn), _p(fo), _p(ff), C.byref(nf)))
    return dict(bow_ids=bi[:nb.value], bow_vals=bv[:nb.value], fv_nodes=fn[:nf.value], fv_off=fo[:nf.value + 1],
                fv_feat=ff[:fo[nf.value]])


class HostVocabulary:
    """dvm_host::ORBVocabulary loaded from DBoW2's text format (loadFromTextFile, TemplatedVocabulary.h:1211-1286) and kept on the device."""

    def __init__(self, filename, device=0):
        H = host_lib()
        H.dvmh_vocab_load_text.restype = C.c_void_p
        H.dvmh_vocab_load_text.argtypes = [C.c_int32, C.c_char_p, C.c_void_p]
        info = np.zeros(4, np.int32)
        self.h = C.c_void_p(H.dvmh_vocab_load_text(device, os.fsencode(filename), _p(info)))
        if not self.h.value:
            raise DvmError(-1, f"{filename}: not a DBoW2 text vocabulary (or the device refused it)")
        self.k, self.L, self.nodes, self.words = (int(x) for x in info)

    def transform(self, features, levelsup):
        H = host_lib()
        H.dvmh_vocab_transform_loaded.restype = C.c_int32; H.dvmh_vocab_transform_loaded.argtypes = None
        f = np.ascontiguousarray(features, np.uint8).reshape(-1, 32)
        n = len(f)
        bi = np.zeros(n + 1, np.int32); bv = np.zeros(n + 1, np.float64); fn = np.zeros(n + 1, np.int32)
        fo = np.zeros(n + 2, np.int32); ff = np.zeros(n + 1, np.int32)
        nb = C.c_int32(0); nf = C.c_int32(0)
        check(H.dvmh_vocab_transform_loaded(self.h, _p(f), C.c_int32(n), C.c_int32(levelsup), _p(bi), _p(bv), C.byref(nb), _p(fn), _p(fo), _p(ff), C.byref(nf)))
        return dict(bow_ids=bi[:nb.value], bow_vals=bv[:nb.value], fv_nodes=fn[:nf.value], fv_off=fo[:nf.value + 1], fv_feat=ff[:fo[nf.value]])

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            f = host_lib().dvmh_vocab_destroy; f.restype = None; f.argtypes = [C.c_void_p]
            f(self.h); self.h = C.c_void_p()

    __del__ = close


def bow_score_host(ids1, vals1, ids2, vals2):
    H = host_lib()
    H.dvmh_bow_score.restype = C.c_double
    H.dvmh_bow_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    a = np.ascontiguousarray(ids1, np.int32); b = np.ascontiguousarray(vals1, np.float64)
    c = np.ascontiguousarray(ids2, np.int32); d = np.ascontiguousarray(vals2, np.float64)
    return H.dvmh_bow_score(_p(a), _p(b), len(a), _p(c), _p(d), len(c))


def sim3_hypotheses(P1c, P2c, max_err1, max_err2, K1, K2, triples, fix_scale=False, device=0):
    """dvm_sim3_hypotheses: Sim3Solver::ComputeSim3 + CheckInliers for H minimal sets in one launch.
    Returns (T12[H,13] = s, R row-major, t; n_inliers[H]; mask[H,N])."""
    L = lib()
    vp, i32 = C.c_void_p, C.c_int32
    L.dvm_sim3_hypotheses.restype = i32
    L.dvm_sim3_hypotheses.argtypes = [i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp]
    a = [np.ascontiguousarray(x, np.float32) for x in (P1c, P2c, max_err1, max_err2, K1, K2)]
    tr = np.ascontiguousarray(triples, np.int32).reshape(-1, 3)
    N, H = len(a[0]), len(tr)
    T = np.zeros((H, 13), np.float32); nin = np.zeros(H, np.int32); mask = np.zeros((H, N), np.uint8)
    check(L.dvm_sim3_hypotheses(device, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), N, _p(a[4]), _p(a[5]), _p(tr), H, int(fix_scale),
                                _p(T), _p(nin), _p(mask)))
    return T, nin, mask


PG_EDGE_DTYPE = np.dtype([("vi", "<i4"), ("vj", "<i4"), ("Sji", "<f8", (8,))])


class PgStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("total_trials", C.c_int32), ("stop_reason", C.c_int32), ("levels", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double), ("tile_fill", C.c_double),
                ("ms_structure", C.c_double), ("ms_optimize", C.c_double), ("chi2_per_iter", C.c_double * 32),
                ("trials_per_iter", C.c_int32 * 32)]


def pose_graph_optimize(S, fixed, edges_v, edges_meas, fix_scale=False, iterations=20, device=0):
    """dvm_pose_graph_optimize (Optimizer::OptimizeEssentialGraph numerics).  Returns (S_opt[n,8], stats dict)."""
    L = lib()
    L.dvm_pose_graph_optimize.restype = C.c_int32
    L.dvm_pose_graph_optimize.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.POINTER(PgStats)]
    So = np.array(S, np.float64, copy=True)
    fx = np.ascontiguousarray(fixed, np.uint8)
    e = np.zeros(len(edges_v), PG_EDGE_DTYPE)
    ev = np.asarray(edges_v)
    e["vi"] = ev[:, 0]; e["vj"] = ev[:, 1]; e["Sji"] = edges_meas
    st = PgStats()
    check(L.dvm_pose_graph_optimize(device, _p(So), _p(fx), len(So), _p(e), len(e), int(fix_scale), int(iterations), C.byref(st)))
    d = {k: getattr(st, k) for k, _ in PgStats._fields_}
    d["chi2_per_iter"] = np.array(st.chi2_per_iter[:]); d["trials_per_iter"] = np.array(st.trials_per_iter[:])
    return So, d


# ---------------------------------------------------------------------------------------------------------------------
# the remaining whole ORBmatcher functions (host mirrors in dvm_slam_amd/host/orb_matcher.cpp over
# dvm_hamming_matrix / dvm_match_lists / dvm_project_search / dvm_match_triangulation).  The view structs of
# host/orb_matcher.h are mirrored as ctypes.Structure; numpy arrays referenced by a view are kept alive on the object.
class _FeatureVectorView(C.Structure):
    _fields_ = [("n", C.c_int32), ("node", C.c_void_p), ("off", C.c_void_p), ("feat", C.c_void_p)]


class _FrameView(C.Structure):
    _fields_ = [("N", C.c_int32), ("mvKeysUn", C.c_void_p), ("mDescriptors", C.c_void_p), ("mvpMapPoints", C.c_void_p),
                ("mvbOutlier", C.c_void_p), ("Tcw", C.c_float * 7), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("mnMinX", C.c_float), ("mnMaxX", C.c_float), ("mnMinY", C.c_float),
                ("mnMaxY", C.c_float), ("mvScaleFactors", C.c_void_p), ("nLevels", C.c_int32), ("dev", C.c_void_p)]


class _KeyFrameView(C.Structure):
    _fields_ = [("N", C.c_int32), ("mvKeysUn", C.c_void_p), ("mDescriptors", C.c_void_p), ("mvpMapPoints", C.c_void_p),
                ("mpBad", C.c_void_p), ("mFeatVec", _FeatureVectorView), ("Tcw", C.c_float * 7), ("Twc", C.c_float * 7),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("mnMinX", C.c_float), ("mnMaxX", C.c_float), ("mnMinY", C.c_float), ("mnMaxY", C.c_float),
                ("mvScaleFactors", C.c_void_p), ("mvLevelSigma2", C.c_void_p), ("mvInvLevelSigma2", C.c_void_p),
                ("mfLogScaleFactor", C.c_float), ("nLevels", C.c_int32)]


class _MapPointsView(C.Structure):
    _fields_ = [("n", C.c_int32), ("id", C.c_void_p), ("bad", C.c_void_p), ("pos", C.c_void_p), ("normal", C.c_void_p),
                ("min_dist", C.c_void_p), ("max_dist", C.c_void_p), ("desc", C.c_void_p)]


class _Sim3View(C.Structure):   # dvm_sim3f: Sophus::Sim3f as stored (RxSO3 quaternion x, y, z, w with |q|^2 = scale; translation)
    _fields_ = [("q", C.c_float * 4), ("t", C.c_float * 3)]


def _ptr(a):
    return None if a is None or a.size == 0 else a.ctypes.data


def _fv_view(fv, keep):
    v = _FeatureVectorView()
    if fv is not None:
        arrs = [np.ascontiguousarray(fv[k], np.int32) for k in ("fv_nodes", "fv_off", "fv_feat")]
        keep.extend(arrs)
        v.n, v.node, v.off, v.feat = len(arrs[0]), _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2])
    return v


def frame_view(kps, desc, bounds, scale_factors, mp=None, K=(0, 0, 0, 0), Tcw=None):
    """FrameView for the matcher mirrors; returns (struct, keepalive list).  Tcw: 7-float SE3f (qx, qy, qz, qw, t)."""
    keep = [np.ascontiguousarray(kps, KP_DTYPE), np.ascontiguousarray(desc, np.uint8), np.ascontiguousarray(scale_factors, np.float32)]
    v = _FrameView()
    v.N, v.mvKeysUn, v.mDescriptors, v.mvScaleFactors, v.nLevels = len(keep[0]), _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), len(keep[2])
    if mp is not None:
        keep.append(mp)
        v.mvpMapPoints = _ptr(mp)
    v.fx, v.fy, v.cx, v.cy = (float(x) for x in K)
    v.mnMinX, v.mnMaxX, v.mnMinY, v.mnMaxY = (float(x) for x in bounds)
    if Tcw is not None:
        v.Tcw = (C.c_float * 7)(*np.asarray(Tcw, np.float32).reshape(7))
    return v, keep


def _pose_call(name, T, *outs):
    a = np.ascontiguousarray(T, np.float32).reshape(7)
    _hcall(name, None, C.c_void_p(a.ctypes.data), *(C.c_void_p(o.ctypes.data) for o in outs))


def se3_inverse(T):
    """Sophus::SE3f::inverse() in the product's float arithmetic (csrc/pose_f32.h); 7-float poses."""
    out = np.zeros(7, np.float32)
    _pose_call("dvmh_se3_inverse", T, out)
    return out


def pose_matrices(Tcw):
    """Frame::UpdatePoseMatrices (Frame.cc:553-559): (mRcw[3,3], mtcw, mOw) of a 7-float SE3f."""
    R = np.zeros(9, np.float32); t = np.zeros(3, np.float32); Ow = np.zeros(3, np.float32)
    _pose_call("dvmh_pose_matrices", Tcw, R, t, Ow)
    return R.reshape(3, 3), t, Ow


def sim3_to_se3(S):
    """Tcw = SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale()), Ow = Tcw.inverse().translation() (ORBmatcher.cc:403-404)."""
    T = np.zeros(7, np.float32); Ow = np.zeros(3, np.float32)
    _pose_call("dvmh_sim3_to_se3", S, T, Ow)
    return T, Ow


def sim3_inverse(S):
    out = np.zeros(7, np.float32)
    _pose_call("dvmh_sim3_inverse", S, out)
    return out


def se3_act(T, P):
    P = np.ascontiguousarray(P, np.float32).reshape(-1, 3); out = np.zeros_like(P)
    a = np.ascontiguousarray(T, np.float32).reshape(7)
    _hcall("dvmh_se3_apply", None, C.c_void_p(a.ctypes.data), C.c_void_p(P.ctypes.data), C.c_int32(len(P)), C.c_void_p(out.ctypes.data))
    return out


def sim3_act(S, P):
    P = np.ascontiguousarray(P, np.float32).reshape(-1, 3); out = np.zeros_like(P)
    a = np.ascontiguousarray(S, np.float32).reshape(7)
    _hcall("dvmh_sim3_apply", None, C.c_void_p(a.ctypes.data), C.c_void_p(P.ctypes.data), C.c_int32(len(P)), C.c_void_p(out.ctypes.data))
    return out


def logf_shared(x):
    """The shared logf of MapPoint::PredictScale (csrc/pose_f32.h), host build."""
    return np.array([_hcall("dvmh_logf", C.c_float, C.c_float(float(v))) for v in np.asarray(x, np.float32).reshape(-1)], np.float32)


def keyframe_view(kf):
    """kf: dict(kps, desc, mp (int32, modified in place by Fuse), bad, fv, Tcw (7-float SE3f; Twc is derived as KeyFrame::SetPose
    does), K, bounds, scale_factors, level_sigma2, inv_level_sigma2, log_scale_factor); missing optional keys are NULL.
    Returns (struct, keepalive)."""
    keep = [np.ascontiguousarray(kf["kps"], KP_DTYPE), np.ascontiguousarray(kf["desc"], np.uint8)]
    v = _KeyFrameView()
    v.N, v.mvKeysUn, v.mDescriptors = len(keep[0]), _ptr(keep[0]), _ptr(keep[1])
    mp = kf.get("mp")
    if mp is not None:
        assert mp.dtype == np.int32 and mp.flags.c_contiguous
        keep.append(mp); v.mvpMapPoints = _ptr(mp)
    if kf.get("bad") is not None:
        b = np.ascontiguousarray(kf["bad"], np.uint8); keep.append(b); v.mpBad = _ptr(b)
    v.mFeatVec = _fv_view(kf.get("fv"), keep)
    if kf.get("Tcw") is not None:
        v.Tcw = (C.c_float * 7)(*np.asarray(kf["Tcw"], np.float32).reshape(7))
        v.Twc = (C.c_float * 7)(*se3_inverse(kf["Tcw"]))
    v.fx, v.fy, v.cx, v.cy = (float(x) for x in kf.get("K", (0, 0, 0, 0)))
    v.mnMinX, v.mnMaxX, v.mnMinY, v.mnMaxY = (float(x) for x in kf["bounds"])
    for name, key in (("mvScaleFactors", "scale_factors"), ("mvLevelSigma2", "level_sigma2"), ("mvInvLevelSigma2", "inv_level_sigma2")):
        if kf.get(key) is not None:
            a = np.ascontiguousarray(kf[key], np.float32); keep.append(a); setattr(v, name, _ptr(a)); v.nLevels = len(a)
    v.mfLogScaleFactor = float(kf.get("log_scale_factor", 0.0))
    return v, keep


def map_points_view(pts):
    """pts: dict(pos, normal, min_dist, max_dist, desc[, id, bad])."""
    keep = [np.ascontiguousarray(pts[k], np.float32) for k in ("pos", "normal", "min_dist", "max_dist")]
    keep.append(np.ascontiguousarray(pts["desc"], np.uint8))
    v = _MapPointsView()
    v.n = len(keep[2])
    v.pos, v.normal, v.min_dist, v.max_dist, v.desc = (_ptr(a) for a in keep)
    if pts.get("id") is not None:
        a = np.ascontiguousarray(pts["id"], np.int32); keep.append(a); v.id = _ptr(a)
    if pts.get("bad") is not None:
        a = np.ascontiguousarray(pts["bad"], np.uint8); keep.append(a); v.bad = _ptr(a)
    return v, keep


def _sim3_view(S):
    S = np.asarray(S, np.float32).reshape(7)
    v = _Sim3View()
    v.q = (C.c_float * 4)(*S[:4]); v.t = (C.c_float * 3)(*S[4:])
    return v


def _hcall(name, restype, *args):
    fn = getattr(host_lib(), name)
    fn.restype = restype
    fn.argtypes = None
    return fn(*args)


def search_for_initialization(F1, F2, prev_matched, window=100, nnratio=0.9, check_ori=True, device=0):
    """dvm_host::ORBmatcher::SearchForInitialization (ORBmatcher.cc:605-707).  F1 / F2 = frame_view() results.
    Returns (nmatches, vnMatches12, vbPrevMatched updated)."""
    pm = np.array(prev_matched, np.float32, copy=True).reshape(-1, 2)
    m = np.zeros(max(F1[0].N, 1), np.int32)
    n = _hcall("dvmh_search_for_initialization", C.c_int32, C.c_int32(device), C.byref(F1[0]), C.byref(F2[0]), C.c_void_p(pm.ctypes.data),
               C.c_void_p(m.ctypes.data), C.c_int32(int(window)), C.c_float(nnratio), C.c_int32(int(check_ori)))
    check(min(n, 0))
    return n, m[:F1[0].N], pm


def search_by_bow_kf_frame(KF, F, fv_f, nnratio=0.7, check_ori=True, device=0):
    """SearchByBoW(KF, F, vpMapPointMatches) (:214-393).  Returns (nmatches, matches[F.N], #host re-queries)."""
    keep = []
    fv = _fv_view(fv_f, keep)
    m = np.zeros(max(F[0].N, 1), np.int32); rq = C.c_int32(0)
    n = _hcall("dvmh_search_by_bow_kf_frame", C.c_int32, C.c_int32(device), C.byref(KF[0]), C.byref(F[0]), C.byref(fv), C.c_float(nnratio),
               C.c_int32(int(check_ori)), C.c_void_p(m.ctypes.data), C.byref(rq))
    check(min(n, 0))
    return n, m[:F[0].N], rq.value


def search_by_bow_kf_kf(KF1, KF2, nnratio=0.8, check_ori=True, device=0):
    """SearchByBoW(KF1, KF2, vpMatches12) (:709-834).  Returns (nmatches, matches12[KF1.N], #host re-queries)."""
    m = np.zeros(max(KF1[0].N, 1), np.int32); rq = C.c_int32(0)
    n = _hcall("dvmh_search_by_bow_kf_kf", C.c_int32, C.c_int32(device), C.byref(KF1[0]), C.byref(KF2[0]), C.c_float(nnratio),
               C.c_int32(int(check_ori)), C.c_void_p(m.ctypes.data), C.byref(rq))
    check(min(n, 0))
    return n, m[:KF1[0].N], rq.value


def triangulation_geometry(KF1, KF2):
    R12 = np.zeros(9, np.float32); t12 = np.zeros(3, np.float32); ep = np.zeros(2, np.float32); F12 = np.zeros(9, np.float32)
    _hcall("dvmh_triangulation_geometry", None, C.byref(KF1[0]), C.byref(KF2[0]), *(C.c_void_p(a.ctypes.data) for a in (R12, t12, ep, F12)))
    return R12, t12, ep, F12


def search_for_triangulation(KF1, KF2, coarse=False, check_ori=True, device=0):
    """SearchForTriangulation (:836-1058, mono).  Returns (nmatches, pairs[nmatches, 2])."""
    pairs = np.zeros((max(KF1[0].N, 1), 2), np.int32)
    n = _hcall("dvmh_search_for_triangulation", C.c_int32, C.c_int32(device), C.byref(KF1[0]), C.byref(KF2[0]), C.c_int32(int(coarse)),
               C.c_int32(int(check_ori)), C.c_void_p(pairs.ctypes.data))
    check(min(n, 0))
    return n, pairs[:n]


def fuse(KF, P, in_kf, th, device=0):
    """Fuse(KF, vpMapPoints, th) search part (:1060-1213).  Returns (count, vBestIdx)."""
    bi = np.zeros(max(P[0].n, 1), np.int32)
    ik = None if in_kf is None else np.ascontiguousarray(in_kf, np.uint8)
    n = _hcall("dvmh_fuse", C.c_int32, C.c_int32(device), C.byref(KF[0]), C.byref(P[0]), None if ik is None else C.c_void_p(ik.ctypes.data),
               C.c_float(th), C.c_void_p(bi.ctypes.data))
    check(min(n, 0))
    return n, bi[:P[0].n]


def fuse_sim3(KF, Scw, P, th, device=0):
    """Fuse(KF, Scw, vpPoints, th, vpReplacePoint) (:1236-1345); Scw: 7-float Sim3f; KF's mp array is updated in place.
    Returns (nFused, replace)."""
    S = _sim3_view(Scw)
    rep = np.zeros(max(P[0].n, 1), np.int32)
    n = _hcall("dvmh_fuse_sim3", C.c_int32, C.c_int32(device), C.byref(KF[0]), C.byref(S), C.byref(P[0]), C.c_float(th), C.c_void_p(rep.ctypes.data))
    check(min(n, 0))
    return n, rep[:P[0].n]


def search_by_projection_sim3(KF, Scw, P, matched, th, ratio_hamming=1.0, device=0, point_kf=None, matched_kf=None):
    """SearchByProjection(KF, Scw, vpPoints, vpMatched, th, ratioHamming) (:395-496); with point_kf / matched_kf the overload
    that also fills vpMatchedKF (:498-603).  Scw: 7-float Sim3f.  Returns (nmatches, vpMatched, #re-queries[, vpMatchedKF])."""
    S = _sim3_view(Scw)
    m = np.array(matched, np.int32, copy=True); rq = C.c_int32(0)
    pk = None if point_kf is None else np.ascontiguousarray(point_kf, np.int32)
    mk = None if matched_kf is None else np.array(matched_kf, np.int32, copy=True)
    n = _hcall("dvmh_search_by_projection_sim3", C.c_int32, C.c_int32(device), C.byref(KF[0]), C.byref(S), C.byref(P[0]),
               None if pk is None else C.c_void_p(pk.ctypes.data), C.c_void_p(m.ctypes.data),
               None if mk is None else C.c_void_p(mk.ctypes.data), C.c_int32(int(th)), C.c_float(ratio_hamming), C.byref(rq))
    check(min(n, 0))
    return (n, m, rq.value) if mk is None else (n, m, rq.value, mk)


def search_by_projection_reloc(Cur, KF, P, already, th, orb_dist, check_ori=True, device=0):
    """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (:1750-1860).  Cur = frame_view(...) built with its mp
    array (updated in place) and pose set on the struct.  Returns (nmatches, #host re-queries)."""
    al = np.sort(np.ascontiguousarray(already, np.int32)); rq = C.c_int32(0)
    n = _hcall("dvmh_search_by_projection_reloc", C.c_int32, C.c_int32(device), C.byref(Cur[0]), C.byref(KF[0]), C.byref(P[0]),
               C.c_void_p(al.ctypes.data) if len(al) else None, C.c_int32(len(al)), C.c_float(th), C.c_int32(int(orb_dist)),
               C.c_int32(int(check_ori)), C.byref(rq))
    check(min(n, 0))
    return n, rq.value


def search_by_sim3(KF1, KF2, P1, P2, matches12, idx_in_kf2, S12, th, device=0):
    """SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (:1347-1551).  S12: 7-float Sim3f.  Returns (nFound, vpMatches12 updated)."""
    S = _sim3_view(S12)
    m = np.array(matches12, np.int32, copy=True)
    ix = None if idx_in_kf2 is None else np.ascontiguousarray(idx_in_kf2, np.int32)
    n = _hcall("dvmh_search_by_sim3", C.c_int32, C.c_int32(device), C.byref(KF1[0]), C.byref(KF2[0]), C.byref(P1[0]), C.byref(P2[0]),
               C.c_void_p(m.ctypes.data), None if ix is None else C.c_void_p(ix.ctypes.data), C.byref(S), C.c_float(th))
    check(min(n, 0))
    return n, m


def project_search(grid, cam, pts, th, scale_factors, skip=None, gate_inv_sigma2=None, gate=5.99, valid=None):
    """dvm_project_search on a FrameGrid slot 0.  cam: dict(Tcw (7-float SE3f), Ow, K, bounds, log_scale_factor[, sim3_pair, S2]).
    Returns (matches, proj)."""
    class _Cam(C.Structure):
        _fields_ = [("Tcw", C.c_float * 7), ("Ow", C.c_float * 3), ("K", C.c_float * 4), ("b", C.c_float * 4),
                    ("lsf", C.c_float), ("nl", C.c_int32), ("sim3_pair", C.c_int32), ("S2", C.c_float * 7)]
    sf = np.ascontiguousarray(scale_factors, np.float32)
    c = _Cam()
    c.Tcw = (C.c_float * 7)(*np.asarray(cam["Tcw"], np.float32).reshape(7))
    c.sim3_pair = int(cam.get("sim3_pair", 0))
    if cam.get("S2") is not None:
        c.S2 = (C.c_float * 7)(*np.asarray(cam["S2"], np.float32).reshape(7))
    c.Ow = (C.c_float * 3)(*np.asarray(cam["Ow"], np.float32)); c.K = (C.c_float * 4)(*np.asarray(cam["K"], np.float32))
    c.b = (C.c_float * 4)(*np.asarray(cam["bounds"], np.float32)); c.lsf = float(cam["log_scale_factor"]); c.nl = len(sf)
    P, keep = map_points_view(pts)
    n = P.n
    out = np.zeros(max(n, 1), MATCH_DTYPE); proj = np.zeros(max(n, 1), np.dtype([("u", "<f4"), ("v", "<f4"), ("radius", "<f4"), ("level", "<i4")]))
    sk = None
    if skip is not None:
        sk = np.zeros(grid.capacity, np.uint8); sk[:len(skip)] = skip
    gi = None if gate_inv_sigma2 is None else np.ascontiguousarray(gate_inv_sigma2, np.float32)
    va = None if valid is None else np.ascontiguousarray(valid, np.uint8)
    fn = lib().dvm_project_search
    fn.restype = C.c_int32; fn.argtypes = None
    vp = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
    check(fn(grid.h, C.c_int32(0), vp(sk), C.byref(c), C.c_void_p(P.pos), C.c_void_p(P.normal), C.c_void_p(P.min_dist), C.c_void_p(P.max_dist),
             C.c_void_p(P.desc), vp(va), C.c_int32(n), C.c_float(th), vp(sf), vp(gi), C.c_double(gate), vp(out), vp(proj), C.c_int32(0), None))
    return out[:n], proj[:n]


def bowdb_query_raw(bows, q_ids, q_vals, erase=(), device=0):
    """dvm_bowdb_* directly: store `bows` (list of (ids, vals)), erase some slots, query.  Returns (common, first_word, score)."""
    L = lib()
    h = C.c_void_p()
    L.dvm_bowdb_create.restype = C.c_int32; L.dvm_bowdb_create.argtypes = [C.c_int32, C.c_void_p]
    check(L.dvm_bowdb_create(device, C.byref(h)))
    try:
        for ids, vals in bows:
            i = np.ascontiguousarray(ids, np.int32); v = np.ascontiguousarray(vals, np.float64)
            slot = C.c_int32(-1)
            L.dvm_bowdb_add.restype = C.c_int32; L.dvm_bowdb_add.argtypes = None
            check(L.dvm_bowdb_add(h, C.c_void_p(i.ctypes.data) if len(i) else None, C.c_void_p(v.ctypes.data) if len(v) else None,
                                  C.c_int32(len(i)), C.byref(slot)))
        for s in erase:
            L.dvm_bowdb_erase.restype = C.c_int32; L.dvm_bowdb_erase.argtypes = None
            check(L.dvm_bowdb_erase(h, C.c_int32(int(s))))
        n = len(bows)
        qi = np.ascontiguousarray(q_ids, np.int32); qv = np.ascontiguousarray(q_vals, np.float64)
        common = np.zeros(n, np.int32); first = np.zeros(n, np.int32); score = np.zeros(n, np.float32)
        L.dvm_bowdb_query.restype = C.c_int32; L.dvm_bowdb_query.argtypes = None
        check(L.dvm_bowdb_query(h, C.c_void_p(qi.ctypes.data) if len(qi) else None, C.c_void_p(qv.ctypes.data) if len(qv) else None,
                                C.c_int32(len(qi)), C.c_void_p(common.ctypes.data), C.c_void_p(first.ctypes.data), C.c_void_p(score.ctypes.data)))
        return common, first, score
    finally:
        L.dvm_bowdb_destroy.restype = None; L.dvm_bowdb_destroy.argtypes = [C.c_void_p]
        L.dvm_bowdb_destroy(h)


class BowDb:
    """dvm_bowdb_* handle (tests): add / erase / query / stats on one store."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        self.L.dvm_bowdb_create.restype = C.c_int32; self.L.dvm_bowdb_create.argtypes = [C.c_int32, C.c_void_p]
        check(self.L.dvm_bowdb_create(device, C.byref(self.h)))
        self.n = 0

    def add(self, ids, vals):
        i = np.ascontiguousarray(ids, np.int32); v = np.ascontiguousarray(vals, np.float64)
        slot = C.c_int32(-1)
        self.L.dvm_bowdb_add.restype = C.c_int32; self.L.dvm_bowdb_add.argtypes = None
        check(self.L.dvm_bowdb_add(self.h, C.c_void_p(i.ctypes.data) if len(i) else None, C.c_void_p(v.ctypes.data) if len(v) else None,
                                   C.c_int32(len(i)), C.byref(slot)))
        self.n += 1
        return slot.value

    def erase(self, slot):
        self.L.dvm_bowdb_erase.restype = C.c_int32; self.L.dvm_bowdb_erase.argtypes = None
        check(self.L.dvm_bowdb_erase(self.h, C.c_int32(int(slot))))

    def query(self, q_ids, q_vals):
        qi = np.ascontiguousarray(q_ids, np.int32); qv = np.ascontiguousarray(q_vals, np.float64)
        common = np.zeros(self.n, np.int32); first = np.zeros(self.n, np.int32); score = np.zeros(self.n, np.float32)
        self.L.dvm_bowdb_query.restype = C.c_int32; self.L.dvm_bowdb_query.argtypes = None
        check(self.L.dvm_bowdb_query(self.h, C.c_void_p(qi.ctypes.data) if len(qi) else None, C.c_void_p(qv.ctypes.data) if len(qv) else None,
                                     C.c_int32(len(qi)), C.c_void_p(common.ctypes.data), C.c_void_p(first.ctypes.data), C.c_void_p(score.ctypes.data)))
        return common, first, score

    def stats(self):
        out = np.zeros(4, np.int64)
        self.L.dvm_bowdb_stats.restype = C.c_int32; self.L.dvm_bowdb_stats.argtypes = [C.c_void_p, C.c_void_p]
        check(self.L.dvm_bowdb_stats(self.h, _p(out)))
        return dict(slots=int(out[0]), live=int(out[1]), words=int(out[2]), capacity=int(out[3]))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.dvm_bowdb_destroy.restype = None; self.L.dvm_bowdb_destroy.argtypes = [C.c_void_p]
            self.L.dvm_bowdb_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close


class HostKeyFrameDatabase:
    """dvm_host::KeyFrameDatabase (host/keyframe_database.cpp over dvm_bowdb_*): add / erase / CalculateMergeScore /
    DetectMergePossibility / DetectNBestCandidates on keyframe slots (reference KeyFrameDatabase.cc:43-70,555-808)."""
    PREFIX = "dvmh_kfdb_"

    def _lib(self):
        return host_lib()

    def __init__(self, device=0):
        f = host_lib().dvmh_kfdb_create; f.restype = C.c_void_p; f.argtypes = [C.c_int32]
        self.h = C.c_void_p(f(device))
        if not self.h.value:
            raise DvmError(-5, lib().dvm_last_error().decode(errors="replace"))

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
        if r < 0:
            check(r)
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
