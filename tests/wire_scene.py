"""Random map deltas (keyframes + map points) for the DVMW wire-format tests."""
import numpy as np


def make_delta(wire, capi, seed=0, n_kf=3, n_mp=40, agent=1):
    rng = np.random.default_rng(seed)
    uu = lambda: rng.integers(0, 256, 16, dtype=np.uint8)
    kfs, mps = [], []
    kf_uuids = [uu() for _ in range(n_kf)]
    mp_uuids = [uu() for _ in range(n_mp)]
    for i in range(n_kf):
        n = int(rng.integers(5, 90))
        kps = np.zeros(n, capi.KP_DTYPE)
        for f in ("x", "y", "size", "angle", "response"):
            kps[f] = rng.uniform(0, 600, n).astype(np.float32)
        kps["octave"] = rng.integers(0, 8, n); kps["class_id"] = -1
        kpmp = np.zeros((n, 16), np.uint8)
        for j in range(n):
            if n_mp and rng.random() < 0.6:
                kpmp[j] = mp_uuids[int(rng.integers(0, n_mp))]
        nb = int(rng.integers(0, 60))
        bow_ids = np.sort(rng.choice(100000, nb, replace=False)).astype(np.int32)
        nodes = np.sort(rng.choice(5000, int(rng.integers(0, 20)), replace=False)).astype(np.int32)
        cnts = rng.integers(1, 5, len(nodes))
        fv = dict(fv_nodes=nodes, fv_off=np.concatenate([[0], np.cumsum(cnts)]).astype(np.int32),
                  fv_feat=rng.integers(0, n, int(cnts.sum())).astype(np.int32))
        links = np.zeros(int(rng.integers(0, 6)), wire.LINK)
        for l in links:
            l["uuid"] = kf_uuids[int(rng.integers(0, n_kf))]; l["weight"] = rng.integers(15, 300); l["kind"] = rng.integers(0, 4)
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        kfs.append(dict(uuid=kf_uuids[i], parent_uuid=kf_uuids[max(i - 1, 0)], mn_id=100 + i, frame_id=1000 + 7 * i, timestamp=12.5 + i,
                        tcw=rng.normal(size=3).astype(np.float32), qcw=q.astype(np.float32), fx=149.0, fy=149.0, cx=320.0, cy=240.0,
                        min_x=0.0, max_x=640.0, min_y=0.0, max_y=480.0, scale_factor=1.2, log_scale_factor=float(np.log(np.float32(1.2))),
                        n_levels=8, creator_agent=agent, origin_map_id=3, flags=int(rng.integers(0, 8)), kps=kps,
                        desc=rng.integers(0, 256, (n, 32), dtype=np.uint8), kp_mappoint=kpmp, bow_ids=bow_ids,
                        bow_vals=rng.uniform(0, 1, nb), fv=fv, links=links))
    for i in range(n_mp):
        obs = np.zeros(int(rng.integers(0, 5)) if n_kf else 0, wire.OBS)
        for o in obs:
            k = int(rng.integers(0, n_kf))
            o["kf_uuid"] = kf_uuids[k]; o["index"] = rng.integers(0, len(kfs[k]["kps"])); o["index_right"] = -1
        mps.append(dict(uuid=mp_uuids[i], ref_kf_uuid=kf_uuids[int(rng.integers(0, n_kf))] if n_kf else uu(), replaced_uuid=np.zeros(16, np.uint8), mn_id=5000 + i,
                        first_kf_id=100, pos=rng.normal(size=3).astype(np.float32), normal=rng.normal(size=3).astype(np.float32),
                        min_distance=float(rng.uniform(0.5, 2)), max_distance=float(rng.uniform(4, 20)),
                        descriptor=rng.integers(0, 256, 32, dtype=np.uint8), creator_agent=agent, flags=int(rng.integers(0, 2)), obs=obs))
    return kfs, mps


def assert_equal_delta(wire, kfs, mps, parsed):
    h, pk, pm = parsed
    assert int(h["magic"]) == wire.MAGIC and int(h["n_keyframes"]) == len(kfs) and int(h["n_mappoints"]) == len(mps)
    for a, b in zip(kfs, pk):
        for name in wire.KEYFRAME.names:
            if name in a:
                assert np.array_equal(np.asarray(a[name], wire.KEYFRAME[name].base).reshape(b["rec"][name].shape), b["rec"][name]), name
        assert np.array_equal(a["kps"], b["kps"]) and np.array_equal(a["desc"], b["desc"]) and np.array_equal(a["kp_mappoint"], b["kp_mappoint"])
        assert np.array_equal(a["bow_ids"], b["bow_ids"]) and np.array_equal(a["bow_vals"], b["bow_vals"])
        for k in ("fv_nodes", "fv_off", "fv_feat"):
            assert np.array_equal(a["fv"][k], b["fv"][k]), k
        assert np.array_equal(a["links"], b["links"])
    for a, b in zip(mps, pm):
        for name in wire.MAPPOINT.names:
            if name in a:
                assert np.array_equal(np.asarray(a[name], wire.MAPPOINT[name].base).reshape(b["rec"][name].shape), b["rec"][name]), name
        assert np.array_equal(a["obs"], b["obs"])
