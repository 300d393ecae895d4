"""DVMW map wire format (include/dvmslam_wire.h): numpy views of the records / sections, block assembly and parsing
through the C ABI (dvm_wire_build / dvm_wire_validate / dvm_wire_layout), and the device-side sender path
(dvm_wire_gather_keypoints).  Replaces the Boost archive of KeyFrame / MapPoint the reference ships between agents
(reference include/KeyFrame.h:57-194, include/MapPoint.h:50-103, src/slam_system/src/orb_slam3_wrapper.cpp:212-455)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

UUID = np.dtype(("u1", (16,)))
HEADER = np.dtype([("magic", "<u4"), ("version", "<u4"), ("total_bytes", "<u8"), ("n_keyframes", "<u4"), ("n_mappoints", "<u4"),
                   ("n_keypoints", "<u4"), ("n_bow", "<u4"), ("n_fv_nodes", "<u4"), ("n_fv_feats", "<u4"), ("n_links", "<u4"),
                   ("n_obs", "<u4"), ("sender_agent", "<u4"), ("flags", "<u4"), ("reserved", "<u4", (2,))])
KEYFRAME = np.dtype([("uuid", "u1", (16,)), ("parent_uuid", "u1", (16,)), ("mn_id", "<u8"), ("frame_id", "<u8"), ("timestamp", "<f8"),
                     ("tcw", "<f4", (3,)), ("qcw", "<f4", (4,)), ("fx", "<f4"), ("fy", "<f4"), ("cx", "<f4"), ("cy", "<f4"),
                     ("min_x", "<f4"), ("max_x", "<f4"), ("min_y", "<f4"), ("max_y", "<f4"), ("scale_factor", "<f4"),
                     ("log_scale_factor", "<f4"), ("n_levels", "<i4"), ("creator_agent", "<i4"), ("origin_map_id", "<i4"),
                     ("flags", "<u4"), ("n_kp", "<u4"), ("kp_off", "<u4"), ("n_bow", "<u4"), ("bow_off", "<u4"), ("n_fv_nodes", "<u4"),
                     ("fv_node_off", "<u4"), ("fv_feat_off", "<u4"), ("n_links", "<u4"), ("link_off", "<u4"), ("reserved", "<u4", (4,))])
MAPPOINT = np.dtype([("uuid", "u1", (16,)), ("ref_kf_uuid", "u1", (16,)), ("replaced_uuid", "u1", (16,)), ("mn_id", "<u8"),
                     ("first_kf_id", "<u8"), ("pos", "<f4", (3,)), ("normal", "<f4", (3,)), ("min_distance", "<f4"),
                     ("max_distance", "<f4"), ("descriptor", "u1", (32,)), ("creator_agent", "<i4"), ("flags", "<u4"), ("n_obs", "<u4"),
                     ("obs_off", "<u4"), ("reserved", "<u4", (4,))])
LINK = np.dtype([("uuid", "u1", (16,)), ("weight", "<i4"), ("kind", "<i4")])
OBS = np.dtype([("kf_uuid", "u1", (16,)), ("index", "<i4"), ("index_right", "<i4")])
assert HEADER.itemsize == 64 and KEYFRAME.itemsize == 192 and MAPPOINT.itemsize == 160 and LINK.itemsize == 24 and OBS.itemsize == 24
SECTION_DTYPES = [HEADER, KEYFRAME, MAPPOINT, capi.KP_DTYPE, np.dtype(("u1", (32,))), UUID, np.dtype("<i4"), np.dtype("<f8"),
                  np.dtype(("<i4", (2,))), np.dtype("<i4"), LINK, OBS]
MAGIC = 0x574D5644


class _Layout(C.Structure):
    _fields_ = [("offset", C.c_uint64 * 12), ("bytes", C.c_uint64 * 12), ("total_bytes", C.c_uint64)]


def _vp(a):
    return None if a is None or a.size == 0 else C.c_void_p(a.ctypes.data)


def layout(header: np.ndarray):
    L = _Layout()
    f = capi.lib().dvm_wire_layout
    f.restype = C.c_int32; f.argtypes = None
    capi.check(f(_vp(header), C.byref(L)))
    return list(L.offset), list(L.bytes), int(L.total_bytes)


def build(keyframes: list, mappoints: list, sender_agent=0, head_only=False) -> np.ndarray:
    """keyframes: dicts with the KEYFRAME scalar fields + kps (KP_DTYPE[n]), desc (u8[n,32]), kp_mappoint (u8[n,16]) [optional when
    head_only], bow_ids, bow_vals, fv (fv_nodes, fv_off, fv_feat as ORBVocabulary.transform returns them), links (LINK[]).
    mappoints: dicts with the MAPPOINT scalar fields + obs (OBS[]).  Returns the block as a uint8 array."""
    kf = np.zeros(len(keyframes), KEYFRAME); mp = np.zeros(len(mappoints), MAPPOINT)
    pools = {k: [] for k in ("kps", "desc", "kpmp", "bow_ids", "bow_vals", "fvn", "fvf", "links", "obs")}
    cnt = dict(kp=0, bow=0, fvn=0, fvf=0, links=0, obs=0)
    for i, k in enumerate(keyframes):
        for name in KEYFRAME.names:
            if name in k:
                kf[i][name] = k[name]
        n = int(k["n_kp"]) if "n_kp" in k and "kps" not in k else len(k["kps"])
        kf[i]["n_kp"], kf[i]["kp_off"] = n, cnt["kp"]; cnt["kp"] += n
        if "kps" in k:
            pools["kps"].append(np.ascontiguousarray(k["kps"], capi.KP_DTYPE)); pools["desc"].append(np.ascontiguousarray(k["desc"], np.uint8).reshape(-1, 32))
        else:
            pools["kps"].append(np.zeros(n, capi.KP_DTYPE)); pools["desc"].append(np.zeros((n, 32), np.uint8))
        pools["kpmp"].append(np.ascontiguousarray(k.get("kp_mappoint", np.zeros((n, 16), np.uint8)), np.uint8).reshape(-1, 16))
        bi = np.ascontiguousarray(k.get("bow_ids", []), np.int32)
        kf[i]["n_bow"], kf[i]["bow_off"] = len(bi), cnt["bow"]; cnt["bow"] += len(bi)
        pools["bow_ids"].append(bi); pools["bow_vals"].append(np.ascontiguousarray(k.get("bow_vals", []), np.float64))
        fv = k.get("fv")
        if fv is not None:
            nodes = np.ascontiguousarray(fv["fv_nodes"], np.int32); off = np.ascontiguousarray(fv["fv_off"], np.int32)
            pairs = np.stack([nodes, np.diff(off)], axis=1).astype(np.int32) if len(nodes) else np.zeros((0, 2), np.int32)
            feats = np.ascontiguousarray(fv["fv_feat"], np.int32)
        else:
            pairs = np.zeros((0, 2), np.int32); feats = np.zeros(0, np.int32)
        kf[i]["n_fv_nodes"], kf[i]["fv_node_off"], kf[i]["fv_feat_off"] = len(pairs), cnt["fvn"], cnt["fvf"]
        cnt["fvn"] += len(pairs); cnt["fvf"] += len(feats)
        pools["fvn"].append(pairs); pools["fvf"].append(feats)
        ln = np.ascontiguousarray(k.get("links", np.zeros(0, LINK)), LINK)
        kf[i]["n_links"], kf[i]["link_off"] = len(ln), cnt["links"]; cnt["links"] += len(ln)
        pools["links"].append(ln)
    for i, m in enumerate(mappoints):
        for name in MAPPOINT.names:
            if name in m:
                mp[i][name] = m[name]
        ob = np.ascontiguousarray(m.get("obs", np.zeros(0, OBS)), OBS)
        mp[i]["n_obs"], mp[i]["obs_off"] = len(ob), cnt["obs"]; cnt["obs"] += len(ob)
        pools["obs"].append(ob)
    cat = lambda xs, dt, shape=(0,): (np.concatenate(xs) if xs else np.zeros(shape, dt))
    P = dict(kps=cat(pools["kps"], capi.KP_DTYPE), desc=cat(pools["desc"], np.uint8, (0, 32)), kpmp=cat(pools["kpmp"], np.uint8, (0, 16)),
             bow_ids=cat(pools["bow_ids"], np.int32), bow_vals=cat(pools["bow_vals"], np.float64), fvn=cat(pools["fvn"], np.int32, (0, 2)),
             fvf=cat(pools["fvf"], np.int32), links=cat(pools["links"], LINK), obs=cat(pools["obs"], OBS))
    h = np.zeros(1, HEADER)
    h["n_keyframes"], h["n_mappoints"], h["n_keypoints"], h["n_bow"] = len(kf), len(mp), cnt["kp"], cnt["bow"]
    h["n_fv_nodes"], h["n_fv_feats"], h["n_links"], h["n_obs"], h["sender_agent"] = cnt["fvn"], cnt["fvf"], cnt["links"], cnt["obs"], sender_agent
    off, _, total = layout(h)
    out = np.zeros(off[3] if head_only else total, np.uint8)
    f = capi.lib().dvm_wire_build
    f.restype = C.c_int32; f.argtypes = None
    capi.check(f(_vp(h), _vp(kf), _vp(mp), _vp(P["kps"]), _vp(P["desc"]), _vp(P["kpmp"]), _vp(P["bow_ids"]), _vp(P["bow_vals"]), _vp(P["fvn"]),
                 _vp(P["fvf"]), _vp(P["links"]), _vp(P["obs"]), C.c_int32(int(head_only)), _vp(out), C.c_uint64(out.size)))
    return out


def validate(block: np.ndarray):
    f = capi.lib().dvm_wire_validate
    f.restype = C.c_int32; f.argtypes = None
    b = np.ascontiguousarray(block, np.uint8)
    capi.check(f(_vp(b), C.c_uint64(b.size)))


def sections(block: np.ndarray) -> list:
    """Validated block -> list of 12 numpy views (zero-copy) in section order."""
    validate(block)
    b = np.ascontiguousarray(block, np.uint8)
    h = b[:64].view(HEADER)
    off, nbytes, _ = layout(h)
    out = []
    for s in range(12):
        dt = SECTION_DTYPES[s]
        raw = b[off[s]:off[s] + nbytes[s]]
        out.append(raw.view(dt.base).reshape((-1,) + dt.shape) if dt.subdtype else raw.view(dt))
    return out


def parse(block: np.ndarray):
    """Validated block -> (header, keyframes, mappoints) with per-record slices of the pooled sections (views)."""
    S = sections(block)
    kfs = []
    for k in S[1]:
        a, n = int(k["kp_off"]), int(k["n_kp"])
        pairs = S[8][int(k["fv_node_off"]):int(k["fv_node_off"]) + int(k["n_fv_nodes"])]
        nf = int(pairs[:, 1].sum()) if len(pairs) else 0
        kfs.append(dict(rec=k, kps=S[3][a:a + n], desc=S[4][a:a + n], kp_mappoint=S[5][a:a + n],
                        bow_ids=S[6][int(k["bow_off"]):int(k["bow_off"]) + int(k["n_bow"])],
                        bow_vals=S[7][int(k["bow_off"]):int(k["bow_off"]) + int(k["n_bow"])],
                        fv=dict(fv_nodes=pairs[:, 0], fv_off=np.concatenate([[0], np.cumsum(pairs[:, 1])]).astype(np.int32),
                                fv_feat=S[9][int(k["fv_feat_off"]):int(k["fv_feat_off"]) + nf]),
                        links=S[10][int(k["link_off"]):int(k["link_off"]) + int(k["n_links"])]))
    mps = [dict(rec=m, obs=S[11][int(m["obs_off"]):int(m["obs_off"]) + int(m["n_obs"])]) for m in S[2]]
    return S[0][0], kfs, mps


def gather_keypoints_device(d_block_ptr: int, first_kf: int, count: int, d_kps: int, kps_stride: int, d_desc: int, desc_stride: int, stream=None):
    """dvm_wire_gather_keypoints: device pointers as integers (e.g. torch tensor .data_ptr() / dvm_orb_result_device)."""
    f = capi.lib().dvm_wire_gather_keypoints
    f.restype = C.c_int32; f.argtypes = None
    capi.check(f(C.c_void_p(d_block_ptr), C.c_int32(first_kf), C.c_int32(count), C.c_void_p(d_kps), C.c_int64(kps_stride), C.c_void_p(d_desc),
                 C.c_int64(desc_stride), C.c_void_p(stream or 0)))
