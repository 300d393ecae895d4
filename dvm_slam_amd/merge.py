"""Inter-agent map-merge candidate pipeline (BASELINE.json config 3: "2-agent inter-map merge: BoW candidates + BF Hamming
cross-match + Sim3 solve"), composed from the accelerated pieces in the order the reference runs them
(reference src/slam_system/src/orb_slam3_wrapper.cpp:457-618 -> KeyFrameDatabase::DetectMergePossibility
KeyFrameDatabase.cc:789-808; LoopClosing::DetectCommonRegionsFromBoW LoopClosing.cc -> ORBmatcher::SearchByBoW :709-834,
Sim3Solver Sim3Solver.cc:210-412, Optimizer::OptimizeSim3 Optimizer.cc:1960-2212, ORBmatcher::SearchBySim3 :1347-1551):

  1. BoW vector of the current keyframe (vocabulary transform) -> merge test against the peer's keyframe database
  2. SearchByBoW(current KF, candidate KF) -> map point correspondences
  3. Sim3Solver: all RANSAC hypotheses in one launch (Horn on minimal sets + inlier count), best one kept
  4. OptimizeSim3 on the correspondences (7-DoF LM)
  5. SearchBySim3 with the refined similarity -> more correspondences

The steps are injected (`ops`), so the same composition runs over the HIP library (GpuOps, below) and -- in the tests --
over the CPU oracle; nothing here computes on the host except index bookkeeping."""
from __future__ import annotations

import numpy as np

from . import synth


class GpuOps:
    """The accelerated implementation of every step (dvm_slam_amd.capi: host mirrors + C ABI)."""

    def __init__(self, voc, device=0):
        from . import capi
        self.capi, self.voc, self.device = capi, voc, device

    def transform(self, desc, levelsup):
        return self.capi.vocab_transform_host(self.voc, desc, levelsup, self.device)

    def new_database(self):
        return self.capi.HostKeyFrameDatabase(self.device)

    def search_by_bow(self, a, b, nnratio):
        n, m, _ = self.capi.search_by_bow_kf_kf(self.capi.keyframe_view(dict(a)), self.capi.keyframe_view(dict(b)), nnratio, True, self.device)
        return n, m

    def sim3_hypotheses(self, P1c, P2c, e1, e2, K1, K2, triples):
        return self.capi.sim3_hypotheses(P1c, P2c, e1, e2, K1, K2, triples, False, self.device)

    def optimize_sim3(self, S12, P1c, P2c, o1, o2, w1, w2, K1, K2, th2):
        return self.capi.optimize_sim3(S12, False, P1c, P2c, o1, o2, w1, w2, K1, K2, th2, self.device)

    def search_by_sim3(self, a, pa, b, pb, m12, idx2, S12, th):
        c = self.capi
        return c.search_by_sim3(c.keyframe_view(dict(a)), c.keyframe_view(dict(b)), c.map_points_view(pa), c.map_points_view(pb), m12, idx2, S12, th,
                                self.device)


def fill_database(ops, kfs, levelsup):
    """Peer side: BoW / feature vectors of its keyframes + its KeyFrameDatabase."""
    db = ops.new_database()
    for i, kf in enumerate(kfs):
        tr = ops.transform(kf["desc"], levelsup)
        kf["bow"] = (tr["bow_ids"], tr["bow_vals"]); kf["fv"] = {k: tr[k] for k in ("fv_nodes", "fv_off", "fv_feat")}
        slot = db.add(tr["bow_ids"], tr["bow_vals"], kf["map_id"], kf["uuid"], kf["mn_id"])
        assert slot == i
    for i, kf in enumerate(kfs):
        db.set_neighbours(i, kf.get("neigh", np.zeros(0, np.int32)))
    return db


def _quat_from_R(R):
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(np.asarray(R, np.float64).reshape(3, 3)).as_quat()
    return -q if q[3] < 0 else q


def merge_with_peer(ops, kf, kf_points, peer_kfs, peer_points, peer_db, levelsup, triples, nnratio=0.75, th_sim3=7.5):
    """kf: the current keyframe of this agent (keyframe dict: kps, desc, mp, bad, Tcw (7-float SE3f), K, bounds, scale tables, uuid,
    map_id ...); kf_points / peer_points[j]: map point data PER KEYPOINT (pos in the owner's world frame, min / max
    distance, descriptor).  Returns a dict with every intermediate result (None fields when the merge is rejected)."""
    out = dict(candidate=-1)
    tr = ops.transform(kf["desc"], levelsup)
    kf = dict(kf, fv={k: tr[k] for k in ("fv_nodes", "fv_off", "fv_feat")})
    ok, best, score, base = peer_db.detect_merge_possibility(tr["bow_ids"], tr["bow_vals"], kf["uuid"], peer_kfs[0]["map_id"])
    out.update(merge_possible=ok, candidate=best, score=score, baseline=base)
    if best < 0:
        return out
    out.update(solve_against_candidate(ops, kf, kf_points, peer_kfs[best], peer_points[best], triples, nnratio, th_sim3))
    return out


def solve_against_candidate(ops, kf, kf_points, pk, pp, triples, nnratio=0.75, th_sim3=7.5):
    """Steps 2-5 for one candidate keyframe `pk` (with per-keypoint map point data `pp`) of the peer; `kf` carries its feature
    vector.  Returns the intermediate results (see merge_with_peer)."""
    out = {}
    n12, m12 = ops.search_by_bow(kf, pk, nnratio)
    out.update(n_bow_matches=n12, bow_matches=m12)
    sel = np.flatnonzero(m12 >= 0)
    if len(sel) < 20:
        return out
    idx2_of_id = {int(v): j for j, v in enumerate(pk["mp"]) if v >= 0}
    i2 = np.array([idx2_of_id[int(v)] for v in m12[sel]])
    # Sim3Solver.cc:66-90: mvX3Dc = Rcw * X3Dw + tcw with the keyframes' rotation matrices
    R1, t1 = synth.Rt_from_se3(kf["Tcw"])
    R2, t2 = synth.Rt_from_se3(pk["Tcw"])
    P1c = (kf_points["pos"][sel].astype(np.float64) @ R1.T + t1).astype(np.float32)
    P2c = (pp["pos"][i2].astype(np.float64) @ R2.T + t2).astype(np.float32)
    e1 = np.floor(9.210 * kf["level_sigma2"][kf["kps"]["octave"][sel]]).astype(np.float32)     # Sim3Solver.cc:106-107
    e2 = np.floor(9.210 * pk["level_sigma2"][pk["kps"]["octave"][i2]]).astype(np.float32)
    T, nin, mask = ops.sim3_hypotheses(P1c, P2c, e1, e2, kf["K"], pk["K"], triples % len(sel))
    h = int(np.argmax(nin))
    out.update(hyp_T=T, hyp_inliers=nin, best_hyp=h)
    if nin[h] < 20:
        return out
    s, R, t = float(T[h, 0]), T[h, 1:10].astype(np.float64), T[h, 10:13].astype(np.float64)
    S12 = np.concatenate([_quat_from_R(R), t, [s]])
    o1 = np.stack([kf["kps"]["x"][sel], kf["kps"]["y"][sel]], 1).astype(np.float64)
    o2 = np.stack([pk["kps"]["x"][i2], pk["kps"]["y"][i2]], 1).astype(np.float64)
    w1 = kf["inv_level_sigma2"][kf["kps"]["octave"][sel]].astype(np.float64)
    w2 = pk["inv_level_sigma2"][pk["kps"]["octave"][i2]].astype(np.float64)
    S, inl, n_in = ops.optimize_sim3(S12, P1c.astype(np.float64), P2c.astype(np.float64), o1, o2, w1, w2, kf["K"].astype(np.float64),
                                     pk["K"].astype(np.float64), 10.0)
    out.update(S12=S, sim3_inliers=inl, n_sim3_inliers=n_in, pairs=(sel, i2))
    if n_in < 20:
        return out
    from scipy.spatial.transform import Rotation
    Rr = Rotation.from_quat(S[:4]).as_matrix().astype(np.float32)
    m_in = np.full(len(kf["kps"]), -1, np.int32); idx2 = np.full(len(kf["kps"]), -1, np.int32)
    keep = sel[inl.astype(bool)]
    m_in[keep] = m12[keep]; idx2[keep] = i2[inl.astype(bool)]
    S12f = synth.sim3_from_sRt(S[7], Rr, S[4:7])     # the Sophus::Sim3f handed to SearchBySim3 (LoopClosing.cc: gScm -> Sim3f)
    nf, m_all = ops.search_by_sim3(kf, kf_points, pk, pp, m_in, idx2, S12f, th_sim3)
    out.update(n_sim3_new=nf, matches=m_all)
    return out
