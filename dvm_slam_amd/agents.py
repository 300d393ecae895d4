"""Decentralised merge round between agents (BASELINE.json config 4: "agents one-per-GPU, decentralized front-end + local BA,
descriptor / pose exchange over RCCL/xGMI"), over torch.distributed -- RCCL on GPUs (backend "nccl"), gloo in the CPU tests.
It is the traffic DVM-SLAM already exchanges between peers (reference src/slam_system/src/orb_slam3_wrapper.cpp:212-384
new keyframes, :457-618 merge attempt, :920-949 coordinate-frame change), expressed with the pieces of this package:

  1. every agent publishes the BoW vector of its current keyframe                      all_gather (ragged)
  2. every agent tests every peer's vector against ITS OWN keyframe database             KeyFrameDatabase::DetectMergePossibility
  3. an agent that sees a merge ships the candidate keyframe + its map points            DVMW block (wire.py), all_gather (ragged)
  4. the agent whose keyframe was recognised solves the similarity                       merge.py: SearchByBoW -> Sim3 RANSAC ->
                                                                                         OptimizeSim3 -> SearchBySim3
  5. and announces the frame change                                                      broadcast of (q, t, s)

No collective touches the per-frame hot path; this round runs when a keyframe is inserted."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import exchange, wire


def _id_to_uuid(i):
    u = np.zeros(16, np.uint8)
    u[:8] = np.array([int(i)], np.int64).view(np.uint8)
    return u


def _uuid_to_id(u):
    return int(np.ascontiguousarray(u[:8]).view(np.int64)[0])


def pack_candidate(kf, pts, agent):
    """Candidate keyframe + the map points its keypoints observe -> DVMW block."""
    n = len(kf["kps"])
    has = kf["mp"] >= 0
    kpmp = np.zeros((n, 16), np.uint8)
    for j in np.flatnonzero(has):
        kpmp[j] = _id_to_uuid(kf["mp"][j])
    T = np.asarray(kf["Tcw"], np.float32).reshape(7)   # Sophus::SE3f as stored: the wire record carries exactly these 7 floats
    rec = dict(uuid=_id_to_uuid(kf["uuid"]), mn_id=kf["mn_id"], tcw=T[4:], qcw=T[:4], fx=kf["K"][0], fy=kf["K"][1], cx=kf["K"][2], cy=kf["K"][3],
               min_x=kf["bounds"][0], max_x=kf["bounds"][1], min_y=kf["bounds"][2], max_y=kf["bounds"][3], scale_factor=1.2,
               log_scale_factor=kf["log_scale_factor"], n_levels=len(kf["scale_factors"]), creator_agent=agent, origin_map_id=kf["map_id"],
               kps=kf["kps"], desc=kf["desc"], kp_mappoint=kpmp, fv=kf["fv"], bow_ids=kf["bow"][0], bow_vals=kf["bow"][1])
    mps = []
    for j in np.flatnonzero(has):
        obs = np.zeros(1, wire.OBS); obs["kf_uuid"] = rec["uuid"]; obs["index"] = j; obs["index_right"] = -1
        mps.append(dict(uuid=_id_to_uuid(kf["mp"][j]), ref_kf_uuid=rec["uuid"], mn_id=int(kf["mp"][j]), pos=pts["pos"][j],
                        normal=pts["normal"][j], min_distance=pts["min_dist"][j], max_distance=pts["max_dist"][j], descriptor=pts["desc"][j],
                        creator_agent=agent, flags=int(kf["bad"][j]) if kf.get("bad") is not None else 0, obs=obs))
    return wire.build([rec], mps, sender_agent=agent)


def unpack_candidate(block, template):
    """DVMW block -> (keyframe dict, per-keypoint map point data) in the form merge.py works on; `template` supplies the scale
    tables (every agent runs the same extractor configuration)."""
    h, kfs, mps = wire.parse(block)
    k = kfs[0]
    rec = k["rec"]
    n = int(rec["n_kp"])
    mp = np.full(n, -1, np.int32); bad = np.zeros(n, np.uint8)
    pts = dict(pos=np.zeros((n, 3), np.float32), normal=np.zeros((n, 3), np.float32), min_dist=np.ones(n, np.float32),
               max_dist=np.ones(n, np.float32), desc=np.zeros((n, 32), np.uint8))
    for m in mps:
        if len(m["obs"]) < 1:
            raise ValueError("DVMW candidate block: map point without an observation")
        j = int(m["obs"][0]["index"])
        if not 0 <= j < n:
            raise ValueError(f"DVMW candidate block: observation index {j} outside the keyframe's {n} keypoints")
        mp[j] = _uuid_to_id(m["rec"]["uuid"]); bad[j] = int(m["rec"]["flags"]) & 1
        pts["pos"][j], pts["normal"][j] = m["rec"]["pos"], m["rec"]["normal"]
        pts["min_dist"][j], pts["max_dist"][j], pts["desc"][j] = m["rec"]["min_distance"], m["rec"]["max_distance"], m["rec"]["descriptor"]
    kf = dict(kps=np.array(k["kps"]), desc=np.array(k["desc"]), mp=mp, bad=bad, fv={a: np.array(b) for a, b in k["fv"].items()},
              Tcw=np.concatenate([np.asarray(rec["qcw"], np.float32), np.asarray(rec["tcw"], np.float32)]),
              K=np.array([rec["fx"], rec["fy"], rec["cx"], rec["cy"]], np.float32),
              bounds=np.array([rec["min_x"], rec["max_x"], rec["min_y"], rec["max_y"]], np.float32), scale_factors=template["scale_factors"],
              level_sigma2=template["level_sigma2"], inv_level_sigma2=template["inv_level_sigma2"], log_scale_factor=float(rec["log_scale_factor"]),
              uuid=_uuid_to_id(rec["uuid"]), map_id=int(rec["origin_map_id"]), mn_id=int(rec["mn_id"]))
    return kf, pts, int(h["sender_agent"])


def _gather_arrays(arrs, device):
    """all_gather a list of numpy arrays of one dtype each (ragged): returns per-rank lists."""
    out = []
    for a in arrs:
        raw = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(device)
        parts = exchange.all_gather_varlen(raw)
        out.append([p.cpu().numpy().view(a.dtype) for p in parts])
    return out


def merge_round(ops, me, levelsup, triples, device="cpu"):
    """One round on every rank.  `me`: dict(kf, kf_points (per keypoint), peers (this agent's keyframes), peer_pts, db (its
    KeyFrameDatabase, filled by merge.fill_database)).  Returns dict(seen: the merges THIS agent's database recognised,
    solved: merge.merge_with_peer-style results for the peers that recognised THIS agent's keyframe, sim3: the announced
    frame changes by rank)."""
    from . import merge
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    tr = ops.transform(me["kf"]["desc"], levelsup)
    ids_all, vals_all, who = _gather_arrays([tr["bow_ids"].astype(np.int32), tr["bow_vals"].astype(np.float64),
                                             np.array([me["kf"]["uuid"]], np.int64)], device)
    # 2. my database against every peer's current keyframe
    seen, blocks = {}, []
    for r in range(world):
        if r == rank:
            continue
        ok, best, score, base = me["db"].detect_merge_possibility(ids_all[r], vals_all[r], int(who[r][0]), me["peers"][0]["map_id"])
        if ok and best >= 0:
            seen[r] = (best, score, base)
    # 3. ship one candidate per recognised peer (header-only empty block otherwise), everybody gets everything
    payload = []
    for r in range(world):
        if r in seen:
            payload.append(pack_candidate(me["peers"][seen[r][0]], me["peer_pts"][seen[r][0]], rank))
        else:
            payload.append(wire.build([], [], sender_agent=rank))
    sizes = np.array([len(b) for b in payload], np.int64)
    cat = torch.from_numpy(np.concatenate(payload)).to(device)
    got_sizes = _gather_arrays([sizes], device)[0]
    got = exchange.all_gather_varlen(cat)
    # 4. solve the similarity for every peer that recognised MY keyframe
    solved = {}
    for r in range(world):
        if r == rank:
            continue
        off = int(np.sum(got_sizes[r][:rank]))
        blk = got[r].cpu().numpy()[off:off + int(got_sizes[r][rank])]
        if int(blk[:64].view(wire.HEADER)["n_keyframes"][0]) == 0:
            continue
        pk, pp, sender = unpack_candidate(blk, me["kf"])
        res = merge.solve_against_candidate(ops, dict(me["kf"], fv={k: tr[k] for k in ("fv_nodes", "fv_off", "fv_feat")}), me["kf_points"], pk, pp, triples)
        solved[sender] = res
    # 5. announce the frame changes (one broadcast per rank: zeros = nothing to announce)
    sim3 = {}
    for r in range(world):
        t = torch.zeros(9, dtype=torch.float64, device=device)
        if r == rank:
            good = [(p_, r_) for p_, r_ in solved.items() if r_.get("n_sim3_inliers", 0) >= 20]   # geometric verification passed
            if good:
                peer, res = max(good, key=lambda x: x[1]["n_sim3_inliers"])
                t[:8] = torch.from_numpy(res["S12"]); t[8] = float(peer) + 1
        if dist.is_initialized() and world > 1:
            dist.broadcast(t, r)
        if t[8] > 0:
            sim3[r] = (int(t[8].item()) - 1, t[:8].cpu().numpy())
    return dict(seen=seen, solved=solved, sim3=sim3)
