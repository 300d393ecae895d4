"""TEST INFRASTRUCTURE: Python handle on tests/shim_driver/libshimdriver.so (the reference-side shims of dvm_slam_amd/host/
compiled against the behaving mock ORB_SLAM3 classes of tests/stubs/) plus a plain-Python MIRROR of the map that is pushed into
it, so that a test can (1) run the shim exactly as the reference would call it and (2) work out independently -- from the
reference's gathering / write-back rules restated here and the CPU oracle for the numerics -- what the map must look like
afterwards."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV_DIR = os.path.join(ROOT, "tests", "shim_driver")
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"),
                           ("class_id", "<i4")])
_lib = None


def build():
    subprocess.check_call(["make", "-C", DRV_DIR], stdout=subprocess.DEVNULL)
    return os.path.join(DRV_DIR, "libshimdriver.so")


def lib():
    global _lib
    if _lib is None:
        from dvm_slam_amd import capi
        capi.lib(); capi.host_lib()           # the product libraries first (RTLD_GLOBAL handles are not needed: rpath resolves them)
        path = os.path.join(DRV_DIR, "libshimdriver.so")
        if not os.path.exists(path):
            build()
        try:
            _lib = C.CDLL(path)
        except OSError:
            # a prebuilt driver whose oracle/_ref/libdutils_ref.so did not travel with it: rebuild (its Makefile links the
            # reference's DUtils::Random only when that file is there, the stub's inline form otherwise)
            subprocess.check_call(["make", "-B", "-C", DRV_DIR], stdout=subprocess.DEVNULL)
            _lib = C.CDLL(path)
        _lib.sw_create.restype = C.c_void_p
        _lib.sw_error.restype = C.c_char_p
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i32(v):
    return np.ascontiguousarray(v, np.int32)


def scale_tables(nlevels=8, factor=1.2):
    """mvScaleFactors / mvLevelSigma2 / mvInvLevelSigma2 as ORBextractor's constructor builds them (float arithmetic, ORBextractor.cc:288-303)."""
    s = np.ones(nlevels, np.float32); g = np.ones(nlevels, np.float32)
    for i in range(1, nlevels):
        s[i] = np.float32(s[i - 1] * np.float32(factor))
        g[i] = np.float32(s[i] * s[i])
    return s, g, (np.float32(1.0) / g).astype(np.float32)


def qt(pose_tq):
    """(t, q) -> the driver's (q, t) float32 layout."""
    p = np.asarray(pose_tq, np.float32)
    return np.ascontiguousarray(np.concatenate([p[3:7], p[0:3]]), np.float32)


class ShimError(RuntimeError):
    pass


class World:
    """Mock ORB_SLAM3 map on the C++ side + its mirror here.  Indices are positions in the keyframe / map point / frame tables."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.sw_create())
        self.kf = []      # dict(id, map, pose (t,q) f32, K, kps, bad, covis [(kf, w)], parent, loops set, matches [mp or -1])
        self.mp = []      # dict(id, map, pos f32[3], bad, obs {kf: idx}, ref)
        self.maps = []    # dict(init_id)
        self.tables = scale_tables()

    def close(self):
        if self.h:
            self.L.sw_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc == -1000:
            raise ShimError(self.L.sw_error(self.h).decode())
        return rc

    # ---- building
    def add_map(self, init_kf_id):
        self.maps.append(dict(init_id=init_kf_id))
        return self.L.sw_add_map(self.h, C.c_ulong(init_kf_id))

    def add_keyframe(self, m, kf_id, pose_tq, K, kps, desc=None, bad=False, bounds=(0, 0, 640, 480), pose_inv_tq=None, log_scale=None):
        kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
        s, g, ig = self.tables
        K = np.ascontiguousarray(K, np.float32)
        b = _i32(bounds)
        pq = qt(pose_tq)
        pinv = qt(pose_inv_tq) if pose_inv_tq is not None else None
        d = np.ascontiguousarray(desc, np.uint8) if desc is not None else None
        ls = float(np.log(np.float32(1.2))) if log_scale is None else log_scale
        i = self.L.sw_add_keyframe(self.h, m, C.c_ulong(kf_id), _p(pq), _p(pinv), _p(K), len(kps), _p(kps), _p(d), _p(s), _p(g), _p(ig), len(s),
                                   C.c_float(ls), _p(b), int(bad))
        self.kf.append(dict(id=kf_id, map=m, pose=np.asarray(pose_tq, np.float32).copy(), K=K, kps=kps, bad=bool(bad), covis=[], parent=None,
                            loops=set(), matches=[-1] * len(kps)))
        return i

    def add_mappoint(self, m, mp_id, xyz, normal=None, min_dist=0.0, max_dist=0.0, desc=None, bad=False):
        x = np.ascontiguousarray(xyz, np.float32)
        n = np.ascontiguousarray(normal, np.float32) if normal is not None else None
        d = np.ascontiguousarray(desc, np.uint8) if desc is not None else None
        i = self.L.sw_add_mappoint(self.h, m, C.c_ulong(mp_id), _p(x), _p(n), C.c_float(min_dist), C.c_float(max_dist), _p(d), int(bad))
        self.mp.append(dict(id=mp_id, map=m, pos=x.copy(), bad=bool(bad), obs={}, ref=None))
        return i

    def observe(self, kf, mp, idx):
        self.L.sw_observe(self.h, kf, mp, idx)
        self.mp[mp]["obs"][kf] = idx
        self.kf[kf]["matches"][idx] = mp
        if self.mp[mp]["ref"] is None:
            self.mp[mp]["ref"] = kf

    def set_covisible(self, kf, others, weights):
        self.L.sw_set_covisible(self.h, kf, _p(_i32(others)), _p(_i32(weights)), len(others))
        self.kf[kf]["covis"] = list(zip([int(o) for o in others], [int(w) for w in weights]))

    def set_parent(self, kf, parent):
        self.L.sw_set_parent(self.h, kf, parent)
        self.kf[kf]["parent"] = parent

    def add_loop_edge(self, a, b):
        self.L.sw_add_loop_edge(self.h, a, b)
        self.kf[a]["loops"].add(b); self.kf[b]["loops"].add(a)

    def set_origin(self, m, kf):
        self.L.sw_map_set_origin(self.h, m, kf)
        self.maps[m]["origin"] = kf

    def set_bef_merge(self, kf, Tcw_tq, Twc_tq):
        self.L.sw_kf_set_bef_merge(self.h, kf, _p(qt(Tcw_tq)), _p(qt(Twc_tq)))

    # ---- reading back
    def get_kf(self, kf):
        pose = np.zeros(7, np.float32); gba = np.zeros(7, np.float32); bef = np.zeros(7, np.float32); info = np.zeros(2, np.int32)
        self.L.sw_get_kf(self.h, kf, _p(pose), _p(gba), _p(bef), _p(info))
        tq = lambda p: np.concatenate([p[4:7], p[0:4]])
        return dict(pose=tq(pose), gba=tq(gba), bef_merge=tq(bef), set_pose=int(info[0]), gba_for=int(info[1]))

    def get_mp(self, mp):
        x = np.zeros(3, np.float32); g = np.zeros(3, np.float32); info = np.zeros(6, np.int32)
        self.L.sw_get_mp(self.h, mp, _p(x), _p(g), _p(info))
        return dict(pos=x, gba=g, bad=bool(info[0]), set_pos=int(info[1]), update_normal=int(info[2]), n_obs=int(info[3]), gba_for=int(info[4]),
                    replaced=int(info[5]))

    def kf_matches(self, kf):
        out = np.zeros(max(len(self.kf[kf]["kps"]), 1), np.int32)
        n = self.L.sw_get_kf_matches(self.h, kf, _p(out))
        return out[:n]

    def mp_observations(self, mp):
        a = np.zeros(256, np.int32); b = np.zeros(256, np.int32)
        n = self.L.sw_get_mp_observations(self.h, mp, _p(a), _p(b), 256)
        return {int(k): int(i) for k, i in zip(a[:n], b[:n])}

    # ---- Optimizer statics
    def local_ba(self, kf, m, stop=None):
        c = np.full(4, -1, np.int32)
        self._chk(self.L.sw_local_ba(self.h, kf, m, _p(stop), _p(c)))
        return dict(num_fixedKF=int(c[0]), num_OptKF=int(c[1]), num_MPs=int(c[2]), num_edges=int(c[3]))

    def global_ba(self, m, iterations, loop_kf, robust):
        self._chk(self.L.sw_global_ba(self.h, m, iterations, C.c_ulong(loop_kf), int(robust)))

    def welding_ba(self, main_kf, adjust, fixed, stop=None):
        self._chk(self.L.sw_welding_ba(self.h, main_kf, _p(_i32(adjust)), len(adjust), _p(_i32(fixed)), len(fixed), _p(stop)))

    def essential_graph_merge(self, cur_kf, fixed, fixed_corrected, non_fixed, mps):
        self._chk(self.L.sw_essential_graph_merge(self.h, cur_kf, _p(_i32(fixed)), len(fixed), _p(_i32(fixed_corrected)), len(fixed_corrected),
                                                  _p(_i32(non_fixed)), len(non_fixed), _p(_i32(mps)), len(mps)))

    def essential_graph_loop(self, m, loop_kf, cur_kf, non_corrected, corrected, connections, fix_scale):
        """non_corrected / corrected: {kf: sim3[8] (qx qy qz qw tx ty tz s)}; connections: [(kf, kf)]."""
        nk = _i32(list(non_corrected.keys())); ns = np.ascontiguousarray([non_corrected[k] for k in non_corrected], np.float64).reshape(-1, 8)
        ck = _i32(list(corrected.keys())); cs = np.ascontiguousarray([corrected[k] for k in corrected], np.float64).reshape(-1, 8)
        cp = _i32(connections).reshape(-1, 2)
        self._chk(self.L.sw_essential_graph_loop(self.h, m, loop_kf, cur_kf, _p(nk), _p(ns), len(nk), _p(ck), _p(cs), len(ck), _p(cp), len(cp), int(fix_scale)))

    # ---- KeyFrameDatabase (the class)
    @staticmethod
    def _uuid16(u):
        return np.frombuffer(int(u).to_bytes(16, "little"), np.uint8).copy()

    def kf_set_bow(self, kf, ids, vals):
        i = _i32(ids); v = np.ascontiguousarray(vals, np.float64)
        self.L.sw_kf_set_bow(self.h, kf, _p(i), _p(v), len(i))

    def kf_set_uuid(self, kf, u): self.L.sw_kf_set_uuid(self.h, kf, _p(self._uuid16(u)))

    def kf_set_connected(self, kf, others):
        o = _i32(others); self.L.sw_kf_set_connected(self.h, kf, _p(o) if len(o) else None, len(o))

    def kf_update_map(self, kf, m): self.L.sw_kf_update_map(self.h, kf, int(m))
    def kf_set_bad(self, kf, bad): self.L.sw_kf_set_bad(self.h, kf, int(bad))
    def map_set_bad(self, m, bad): self.L.sw_map_set_bad(self.h, m, int(bad))
    def kfdb_create(self): self._chk(self.L.sw_kfdb_create(self.h))
    def kfdb_get_state(self, kf, reloc=False):
        q = C.c_uint64(0); w = C.c_int32(0); s = C.c_float(0)
        if not self.L.sw_kfdb_get_state(self.h, kf, int(reloc), C.byref(q), C.byref(w), C.byref(s)):
            return None
        return q.value, w.value, s.value

    def kfdb_add(self, kf): self._chk(self.L.sw_kfdb_add(self.h, kf))
    def kfdb_erase(self, kf): self._chk(self.L.sw_kfdb_erase(self.h, kf))

    def kfdb_detect_merge_possibility(self, ids, vals, uuid, m):
        i = _i32(ids); v = np.ascontiguousarray(vals, np.float64); best = C.c_int32(-1)
        r = self._chk(self.L.sw_kfdb_detect_merge_possibility(self.h, _p(i), _p(v), len(i), _p(self._uuid16(uuid)), m, C.byref(best)))
        return r, best.value

    def kfdb_merge_score(self, ids, vals, uuid, m, score=0.0):
        i = _i32(ids); v = np.ascontiguousarray(vals, np.float64); best = C.c_int32(-1); sc = C.c_float(score)
        self._chk(self.L.sw_kfdb_merge_score(self.h, _p(i), _p(v), len(i), _p(self._uuid16(uuid)), m, C.byref(sc), C.byref(best)))
        return sc.value, best.value

    def kfdb_detect_n_best(self, kf, n_num):
        lo = np.zeros(max(n_num, 1), np.int32); me = np.zeros(max(n_num, 1), np.int32); nl = C.c_int32(0); nm = C.c_int32(0)
        self._chk(self.L.sw_kfdb_detect_n_best(self.h, kf, n_num, _p(lo), C.byref(nl), _p(me), C.byref(nm)))
        return lo[:nl.value].copy(), me[:nm.value].copy()

    def kfdb_detect_reloc(self, ids, vals, frame_id, m, cap=4096):
        i = _i32(ids); v = np.ascontiguousarray(vals, np.float64); out = np.zeros(cap, np.int32); n = C.c_int32(0)
        self._chk(self.L.sw_kfdb_detect_reloc(self.h, _p(i), _p(v), len(i), C.c_ulong(frame_id), m, _p(out), C.byref(n)))
        return out[:n.value].copy()

    def compute_distinctive(self, mps, batched):
        m = _i32(mps); out = np.zeros((len(m), 32), np.uint8)
        self._chk(self.L.sw_compute_distinctive(self.h, _p(m), len(m), int(batched), _p(out)))
        return out

    def vocab_compute_bow(self, path, desc, levelsup):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(d)
        bi = np.zeros(n + 1, np.int32); bv = np.zeros(n + 1, np.float64); fn = np.zeros(n + 1, np.int32); fo = np.zeros(n + 2, np.int32); ff = np.zeros(n + 1, np.int32)
        nb = C.c_int32(0); nf = C.c_int32(0); vs = C.c_int32(0); sc = C.c_double(0)
        self._chk(self.L.sw_vocab_compute_bow(self.h, os.fsencode(path), _p(d), n, levelsup, _p(bi), _p(bv), C.byref(nb), _p(fn), _p(fo), _p(ff), C.byref(nf),
                                              C.byref(vs), C.byref(sc)))
        return dict(bow_ids=bi[:nb.value], bow_vals=bv[:nb.value], fv_nodes=fn[:nf.value], fv_off=fo[:nf.value + 1], fv_feat=ff[:fo[nf.value]],
                    size=vs.value, self_score=sc.value)

    def sim3_solver(self, kf1, kf2, matches12, fix_scale, min_inliers, max_its, per_call, seed, n1):
        m = _i32(matches12)
        T = np.zeros(16, np.float32); est = np.zeros(13, np.float32); inl = np.zeros(n1, np.uint8); info = np.zeros(4, np.int32)
        self._chk(self.L.sw_sim3_solver(self.h, kf1, kf2, _p(m), int(fix_scale), min_inliers, max_its, per_call, C.c_uint(seed), _p(T), _p(est), _p(inl), _p(info)))
        return T.reshape(4, 4), est, inl.astype(bool), dict(calls=int(info[0]), converged=bool(info[1]), no_more=bool(info[2]), n_inliers=int(info[3]))

    def optimize_sim3(self, kf1, kf2, matches1, S12, th2, fix_scale, all_points=False):
        m = _i32(matches1).copy(); S = np.ascontiguousarray(S12, np.float64).copy()
        n = self._chk(self.L.sw_optimize_sim3(self.h, kf1, kf2, _p(m), _p(S), C.c_float(th2), int(fix_scale), int(all_points)))
        return n, m, S


# --------------------------------------------------------------------------------------------------------------------------
# A synthetic map from dvm_slam_amd.synth.ba_problem: keyframe k's keypoints are its observations (in edge order), octave from
# the edge weight, poses / positions rounded to float (the reference stores Sophus::SE3f / Vector3f).
def world_from_problem(pr, kf_ids=None, mp_ids=None, map_of_kf=None, init_kf_id=None, bad_kf=(), bad_mp=()):
    W = World()
    P, Lm = len(pr["poses"]), len(pr["points"])
    kf_ids = list(range(P)) if kf_ids is None else list(kf_ids)
    mp_ids = list(range(Lm)) if mp_ids is None else list(mp_ids)
    map_of_kf = [0] * P if map_of_kf is None else list(map_of_kf)
    for m in range(max(map_of_kf) + 1):
        W.add_map(kf_ids[0] if init_kf_id is None else init_kf_id)
    K = np.asarray(pr["intrinsics"], np.float32)
    octave = np.rint(-np.log(pr["inv_sigma2"]) / (2 * np.log(1.2))).astype(np.int32)
    per_kf = [[] for _ in range(P)]
    for e in range(len(pr["edge_pose"])):
        per_kf[pr["edge_pose"][e]].append(e)
    W.edge_slot = {}                    # edge -> (kf, keypoint index)
    for k in range(P):
        kps = np.zeros(len(per_kf[k]), KEYPOINT_DTYPE)
        for j, e in enumerate(per_kf[k]):
            kps[j]["x"], kps[j]["y"], kps[j]["octave"] = pr["obs"][e, 0], pr["obs"][e, 1], octave[e]
            W.edge_slot[e] = (k, j)
        W.add_keyframe(map_of_kf[k], kf_ids[k], pr["poses"][k], K, kps, bad=k in bad_kf)
    for l in range(Lm):
        first = int(pr["edge_pose"][np.flatnonzero(pr["edge_point"] == l)[0]]) if np.any(pr["edge_point"] == l) else 0
        W.add_mappoint(map_of_kf[first], mp_ids[l], pr["points"][l], bad=l in bad_mp)
    for e in range(len(pr["edge_pose"])):
        k, j = W.edge_slot[e]
        W.observe(k, int(pr["edge_point"][e]), j)
    # covisibility: shared map points, strongest first (KeyFrame::UpdateConnections order: weight descending)
    share = np.zeros((P, P), np.int32)
    for l in range(Lm):
        ks = list(W.mp[l]["obs"].keys())
        for a in ks:
            for b in ks:
                if a != b:
                    share[a, b] += 1
    for k in range(P):
        o = [int(j) for j in np.argsort(-share[k], kind="stable") if share[k, j] > 0]
        W.set_covisible(k, o, [int(share[k, j]) for j in o])
        if k > 0:
            W.set_parent(k, k - 1)
    return W


# --------------------------------------------------------------------------------------------------------------------------
# Mirror-side restatement of the map mutations the optimisers perform (MapPoint::EraseObservation / SetBadFlag,
# KeyFrame::EraseMapPointMatch): tests replay the expected erasures on the mirror and compare tables.
def mirror_erase(W, kf, mp, min_obs=3):
    P, K = W.mp[mp], W.kf[kf]
    idx = P["obs"].get(kf, -1)
    if idx != -1:
        K["matches"][idx] = -1                      # KeyFrame::EraseMapPointMatch(pMP)
        del P["obs"][kf]                            # MapPoint::EraseObservation
        if P["ref"] == kf:
            P["ref"] = None if not P["obs"] else "any"
        if P["ref"] is None or len(P["obs"]) < min_obs:
            P["bad"] = True                          # SetBadFlag: every keyframe forgets the point
            for k2, i2 in P["obs"].items():
                W.kf[k2]["matches"][i2] = -1
            P["obs"] = {}


# --------------------------------------------------------------------------------------------------------------------------
# Frames (Tracking's per-image object)
def _add_frame(W, pose_tq, K, kps, desc=None, bounds=(0.0, 640.0, 0.0, 480.0)):
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
    s, g, ig = W.tables
    K = np.ascontiguousarray(K, np.float32)
    b = np.ascontiguousarray(bounds, np.float32)
    d = np.ascontiguousarray(desc, np.uint8) if desc is not None else None
    f = W.L.sw_add_frame(W.h, _p(qt(pose_tq)), _p(K), len(kps), _p(kps), _p(d), _p(s), _p(g), _p(ig), len(s), C.c_float(float(np.log(np.float32(1.2)))), _p(b))
    W.frame_n = getattr(W, "frame_n", {})
    W.frame_n[f] = len(kps)
    return f


def _frame_set_matches(W, f, mp, outlier=None):
    o = np.ascontiguousarray(outlier, np.uint8) if outlier is not None else None
    W.L.sw_frame_set_matches(W.h, f, _p(_i32(mp)), _p(o))


def _get_frame(W, f):
    n = W.frame_n[f]
    pose = np.zeros(7, np.float32); mp = np.zeros(max(n, 1), np.int32); out = np.zeros(max(n, 1), np.uint8)
    calls = W.L.sw_get_frame(W.h, f, _p(pose), _p(mp), _p(out))
    return dict(pose=np.concatenate([pose[4:7], pose[0:4]]), mp=mp[:n], outlier=out[:n], set_pose=calls)


def _pose_optimization(W, f):
    return W._chk(W.L.sw_pose_optimization(W.h, f))


def _frame_extract(W, f, img, keep_device):
    """Frame::ExtractORB on the world's persistent extractor (+ the optional mDvmDevice line): returns N."""
    img = np.ascontiguousarray(img, np.uint8)
    n = W._chk(W.L.sw_frame_extract(W.h, f, _p(img), img.shape[0], img.shape[1], img.shape[1], int(keep_device)))
    W.frame_n[f] = n
    return n


def _frame_keypoints(W, f):
    n = W.frame_n[f]
    k = np.zeros(max(n, 1), KEYPOINT_DTYPE); d = np.zeros((max(n, 1), 32), np.uint8)
    W.L.sw_frame_keypoints(W.h, f, _p(k), _p(d))
    return k[:n], d[:n]


World.add_frame = _add_frame
World.frame_extract = _frame_extract
World.frame_keypoints = _frame_keypoints
World.frame_set_matches = _frame_set_matches
World.get_frame = _get_frame
World.pose_optimization = _pose_optimization


# --------------------------------------------------------------------------------------------------------------------------
# g2o::Sim3 in plain Python floats with the operation order of tests/stubs (Eigen's quaternion product and point action):
# the measurements a test derives are then bit-identical to the ones the shim derives from the same poses.
def _cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def q_rot(q, p):
    v = q[:3]; w = q[3]
    uv = _cross(v, p); uv = [u + u for u in uv]
    c = _cross(v, uv)
    return [(p[i] + w * uv[i]) + c[i] for i in range(3)]


def q_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return [aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx,
            aw * bw - ax * bx - ay * by - az * bz]


def s_inv(S):
    q, t, s = list(S[:4]), list(S[4:7]), S[7]
    qc = [-q[0], -q[1], -q[2], q[3]]
    k = -1. / s
    return qc + q_rot(qc, [t[i] * k for i in range(3)]) + [1. / s]


def s_mul(A, B):
    r = q_rot(list(A[:4]), list(B[4:7]))
    return q_mul(list(A[:4]), list(B[:4])) + [r[i] * A[7] + A[4 + i] for i in range(3)] + [A[7] * B[7]]


def s_map(S, x):
    r = q_rot(list(S[:4]), list(x))
    return [r[i] * S[7] + S[4 + i] for i in range(3)]


def sim3_of_pose(pose_tq_f32):
    """g2o::Sim3(Tcw.unit_quaternion().cast<double>(), Tcw.translation().cast<double>(), 1.0) of a float (t, q) pose."""
    p = np.asarray(pose_tq_f32, np.float32).astype(np.float64)
    x, y, z, w = [float(v) for v in p[3:7]]
    n = ((x * x + z * z) + (y * y + w * w)) ** 0.5        # Sophus: SE3f::cast<double>() re-normalises the quaternion (SO3's constructor)
    return [x / n, y / n, z / n, w / n] + [float(v) for v in p[0:3]] + [1.0]
