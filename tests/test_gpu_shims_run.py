"""The reference-side binding, EXECUTED: every monocular non-inertial static of ORB_SLAM3::Optimizer (include/Optimizer.h:48-92) as
defined by dvm_slam_amd/host/Optimizer_shim.h is compiled against behaving mock KeyFrame / MapPoint / Map / Frame classes
(tests/stubs/), linked with libdvmslam_hip.so and run on a synthetic map (tests/shim_driver/).  The map it leaves behind --
poses, points, erased observations, mvpMapPoints tables, bookkeeping members -- is compared with what the reference's own
gathering and write-back rules (restated here in Python, independently of the shim) give when the CPU oracle does the numerics.
Reference behaviour matched: Optimizer.cc:55-356 (BundleAdjustment), :1030-1387 (LocalBundleAdjustment), :3257-3675 (welding
LocalBundleAdjustment), :744-1028 (PoseOptimization), :1960-2212 (OptimizeSim3), :1389-1652 and :1653-1958 (OptimizeEssentialGraph)."""
import numpy as np
import pytest

import ref_dutils
import shim_world as sw
from dvm_slam_amd import synth

pytestmark = pytest.mark.gpu

CHI2_MONO = 5.991
BORDER = 2e-3     # an observation whose chi2 lies this close to the gate may fall on either side (BA parity is 1e-6, not bitwise)


def _close_pose(got_tq, exp_tq):
    """Poses written through Sophus::SE3f(q.cast<float>(), t.cast<float>()): float rounding + the 1e-6 BA tolerance."""
    exp = np.asarray(exp_tq, np.float64).copy()
    exp[3:] /= np.linalg.norm(exp[3:])
    g = np.asarray(got_tq, np.float64)
    if np.dot(g[3:], exp[3:]) < 0:
        g = np.concatenate([g[:3], -g[3:]])
    return np.allclose(g, exp, rtol=3e-6, atol=3e-6)


def _problem(seed=3, n_kf=16, n_pts=400):
    return synth.ba_problem(n_kf=n_kf, n_pts=n_pts, k_obs=5, seed=seed, radius=12.0)


def _edges_for(W, oracle, cams, fixed, landmarks, keep):
    """Edge arrays the reference would build: for every landmark its observations in the cameras `keep(kf)` admits."""
    slot = {k: i for i, k in enumerate(cams)}
    ep, el, ob, w, owner = [], [], [], [], []
    inv_sigma2 = W.tables[2]
    for pt, mp in enumerate(landmarks):
        for k, idx in sorted(W.mp[mp]["obs"].items()):
            if k not in slot or not keep(k, mp, idx):
                continue
            kp = W.kf[k]["kps"][idx]
            ep.append(slot[k]); el.append(pt); ob.append((float(kp["x"]), float(kp["y"]))); w.append(float(inv_sigma2[kp["octave"]]))
            owner.append((k, mp))
    edges = oracle.make_edges(np.asarray(ep, np.int32), np.asarray(el, np.int32), np.asarray(ob, np.float64).reshape(-1, 2), np.asarray(w, np.float64))
    poses = np.array([W.kf[k]["pose"] for k in cams], np.float64)
    points = np.array([W.mp[m]["pos"] for m in landmarks], np.float64)
    return poses, np.asarray(fixed, np.uint8), points, edges, owner


def _check_tables(W):
    """mvpMapPoints of every keyframe and the observations of every map point against the mirror."""
    for k in range(len(W.kf)):
        assert list(W.kf_matches(k)) == W.kf[k]["matches"], f"mvpMapPoints of keyframe {k}"
    for m in range(len(W.mp)):
        assert W.mp_observations(m) == W.mp[m]["obs"], f"observations of map point {m}"
        assert W.get_mp(m)["bad"] == W.mp[m]["bad"], f"bad flag of map point {m}"


@pytest.mark.parametrize("pooled", [False, True])
def test_local_bundle_adjustment(oracle, pooled):
    """pooled: the window goes through a registered dvm_ba_pool (the several-agents-on-one-GPU form, Optimizer_shim.h
    set_local_ba_pool) -- same tables, same rejected observations, poses and points within the same tolerance."""
    pr = _problem()
    # keyframe 9 is bad, keyframes 13..15 belong to another map, map point 7 is bad: none of them may enter the window
    W = sw.world_from_problem(pr, kf_ids=[10 + 3 * k for k in range(16)], map_of_kf=[0] * 13 + [1] * 3, init_kf_id=10, bad_kf={9}, bad_mp={7})
    main = 5
    own = W.kf[main]["map"]
    usable = lambda k: not W.kf[k]["bad"] and W.kf[k]["map"] == own
    neigh = [n for n, _ in W.kf[main]["covis"]]
    free = [main] + [n for n in neigh if usable(n)]
    assert 9 in neigh and not usable(9)
    landmarks, seen = [], set()
    for k in free:
        for mp in W.kf[k]["matches"]:
            if mp >= 0 and not W.mp[mp]["bad"] and W.mp[mp]["map"] == own and mp not in seen:
                seen.add(mp); landmarks.append(mp)
    tagged, anchors = set([main] + neigh), []
    for mp in landmarks:
        for k in sorted(W.mp[mp]["obs"]):
            if k not in tagged:
                tagged.add(k)
                if usable(k):
                    anchors.append(k)
    holds_initial = any(W.kf[k]["id"] == 10 for k in free)
    cams = free + anchors
    fixed = [W.kf[k]["id"] == 10 for k in free] + [True] * len(anchors)
    poses, fx, points, edges, owner = _edges_for(W, oracle, cams, fixed, landmarks, lambda k, mp, idx: usable(k))
    huber = float(np.float32(np.sqrt(np.float32(5.991))))
    po, xo, st, chi = oracle.ba_optimize(poses, fx, points, edges, pr["intrinsics"], huber, 10)
    _, front = oracle.ba_edge_chi2(po, xo, edges, pr["intrinsics"])

    assert W.L.sw_set_local_ba_pool(int(pooled)) == 0
    try:
        out = W.local_ba(main, own)
    finally:
        W.L.sw_set_local_ba_pool(0)
    assert out == dict(num_fixedKF=len(anchors) + int(holds_initial), num_OptKF=len(free), num_MPs=len(landmarks), num_edges=len(edges))
    assert len(anchors) > 0 and len(free) > 4 and st["iterations"] >= 3

    sure = np.abs(chi - CHI2_MONO) > BORDER
    rejected = [(k, mp) for (k, mp), c, f, s in zip(owner, chi, front, sure) if s and (c > CHI2_MONO or not f)]
    assert len(rejected) > 20
    assert sure.all(), "pick another seed: an observation sits on the chi2 gate"
    for k, mp in rejected:
        sw.mirror_erase(W, k, mp)
    _check_tables(W)
    for i, k in enumerate(free):
        g = W.get_kf(k)
        assert g["set_pose"] == 1 and _close_pose(g["pose"], po[i]), f"pose of free keyframe {k}"
    for k in range(len(W.kf)):
        if k not in free:
            g = W.get_kf(k)
            assert g["set_pose"] == 0 and np.array_equal(g["pose"], W.kf[k]["pose"]), f"keyframe {k} must not move"
    for i, mp in enumerate(landmarks):
        g = W.get_mp(mp)
        assert g["set_pos"] == 1 and g["update_normal"] == 1
        assert np.allclose(g["pos"], xo[i], rtol=3e-6, atol=3e-6), f"position of map point {mp}"
    for mp in range(len(W.mp)):
        if mp not in seen:
            assert W.get_mp(mp)["set_pos"] == 0
    assert W.L.sw_map_change_index(W.h, own) == 1
    opt = np.zeros(64, np.uint64); fxd = np.zeros(64, np.uint64); nf = np.zeros(1, np.int32)
    n = W.L.sw_map_opt_fixed(W.h, own, sw._p(opt), sw._p(fxd), 64, sw._p(nf))
    assert sorted(opt[:n]) == sorted(W.kf[k]["id"] for k in free) and sorted(fxd[:nf[0]]) == sorted(W.kf[k]["id"] for k in anchors)


def test_local_bundle_adjustment_pooled_from_several_threads():
    """Four agents' LocalMapping threads calling LocalBundleAdjustment at once with a registered pool: every world ends with the
    bits of its solo pooled call (the result of a window does not depend on the launch it rode in)."""
    import threading
    seeds = [3, 5, 8, 13]

    def world(seed):
        return sw.world_from_problem(_problem(seed=seed), init_kf_id=0)

    def state(W):
        return ([W.get_kf(k)["pose"].copy() for k in range(len(W.kf))], [W.get_mp(m)["pos"].copy() for m in range(len(W.mp))],
                [W.mp_observations(m) for m in range(len(W.mp))])

    probe = world(seeds[0])
    assert probe.L.sw_set_local_ba_pool(1) == 0
    try:
        solo = []
        for s in seeds:
            W = world(s)
            out = W.local_ba(5, 0)
            assert out["num_edges"] > 0
            solo.append((out, state(W)))
        worlds = [world(s) for s in seeds]
        outs = [None] * len(seeds)

        def job(i):
            outs[i] = worlds[i].local_ba(5, 0)
        th = [threading.Thread(target=job, args=(i,)) for i in range(len(seeds))]
        for t in th: t.start()
        for t in th: t.join()
    finally:
        probe.L.sw_set_local_ba_pool(0)
    for i in range(len(seeds)):
        assert outs[i] == solo[i][0]
        got = state(worlds[i])
        for a, b in zip(got[0], solo[i][1][0]):
            assert np.array_equal(a, b)
        for a, b in zip(got[1], solo[i][1][1]):
            assert np.array_equal(a, b)
        assert got[2] == solo[i][1][2]


def test_local_bundle_adjustment_without_gauge_or_with_stop_flag(oracle):
    """No fixed keyframe -> the reference returns before building anything (Optimizer.cc:1088-1091); a raised stop flag -> it
    returns after counting (:1306-1308).  Nothing may be written in either case."""
    pr = _problem(seed=5, n_kf=6, n_pts=120)
    W = sw.world_from_problem(pr, init_kf_id=999)            # the initial keyframe is not in the window and every observer is a neighbour
    out = W.local_ba(2, 0)
    assert out["num_fixedKF"] == 0 and out["num_OptKF"] == -1
    W2 = sw.world_from_problem(_problem(), init_kf_id=0)
    stop = np.ones(1, np.uint8)
    out = W2.local_ba(5, 0, stop=stop)
    assert out["num_edges"] > 0
    for W_ in (W, W2):
        assert all(W_.get_kf(k)["set_pose"] == 0 for k in range(len(W_.kf)))
        assert all(W_.get_mp(m)["set_pos"] == 0 for m in range(len(W_.mp)))


@pytest.mark.parametrize("direct,robust", [(True, True), (False, False)])
def test_global_bundle_adjustment(oracle, direct, robust):
    pr = _problem(seed=11)
    W = sw.world_from_problem(pr, kf_ids=[2 * k for k in range(16)], init_kf_id=0, bad_kf={4}, bad_mp={3, 50})
    W.set_origin(0, 0)
    loop_kf = 0 if direct else 22                                # == origin keyframe's id: results go to the entities themselves
    cams = [k for k in range(16) if not W.kf[k]["bad"]]
    newest = max(W.kf[k]["id"] for k in cams)
    landmarks = []
    for mp in range(len(W.mp)):
        if W.mp[mp]["bad"]:
            continue
        if any((not W.kf[k]["bad"]) and W.kf[k]["id"] <= newest for k in W.mp[mp]["obs"]):
            landmarks.append(mp)
    fixed = [W.kf[k]["id"] == 0 for k in cams]
    poses, fx, points, edges, _ = _edges_for(W, oracle, cams, fixed, landmarks, lambda k, mp, idx: not W.kf[k]["bad"])
    huber = float(np.float32(np.sqrt(np.float32(5.99)))) if robust else 0.0
    po, xo, st, _ = oracle.ba_optimize(poses, fx, points, edges, pr["intrinsics"], huber, 10)
    W.global_ba(0, 10, loop_kf, robust)
    for i, k in enumerate(cams):
        g = W.get_kf(k)
        if direct:
            assert g["set_pose"] == 1 and _close_pose(g["pose"], po[i])
        else:
            assert g["set_pose"] == 0 and g["gba_for"] == loop_kf and _close_pose(g["gba"], po[i])
            assert np.array_equal(g["pose"], W.kf[k]["pose"])
    assert W.get_kf(4)["set_pose"] == 0 and W.get_kf(4)["gba_for"] == 0
    for i, mp in enumerate(landmarks):
        g = W.get_mp(mp)
        if direct:
            assert (g["set_pos"], g["update_normal"]) == (1, 1) and np.allclose(g["pos"], xo[i], rtol=3e-6, atol=3e-6)
        else:
            assert g["set_pos"] == 0 and g["gba_for"] == loop_kf and np.allclose(g["gba"], xo[i], rtol=3e-6, atol=3e-6)
    assert W.get_mp(3)["set_pos"] == 0 and W.get_mp(3)["gba_for"] == 0
    _check_tables(W)                                             # global BA erases nothing


def test_welding_bundle_adjustment(oracle):
    """Optimizer::LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF, pbStopFlag), Optimizer.cc:3257-3675: two rounds on one graph."""
    pr = _problem(seed=21, n_kf=16, n_pts=500)
    W = sw.world_from_problem(pr, kf_ids=[100 + k for k in range(16)], init_kf_id=100, bad_kf={6}, bad_mp={11})
    main, adjust, fixed_l = 8, [7, 8, 9, 10, 6, 11], [3, 4, 5]      # the bad keyframe 6 is named but must be skipped
    usable = lambda k: not W.kf[k]["bad"]
    cams, fixed, landmarks, seen = [], [], [], set()
    for lst, fx_ in ((fixed_l, True), (adjust, False)):
        for k in lst:
            if not usable(k):
                continue
            cams.append(k); fixed.append(fx_)
            for mp in sorted(set(m for m in W.kf[k]["matches"] if m >= 0 and not W.mp[m]["bad"])):
                if mp not in seen:
                    seen.add(mp); landmarks.append(mp)
    newest = max(W.kf[k]["id"] for k in cams)
    keep = lambda k, mp, idx: usable(k) and W.kf[k]["id"] <= newest and W.kf[k]["matches"][idx] >= 0
    poses, fx, points, edges, owner = _edges_for(W, oracle, cams, fixed, landmarks, keep)
    huber = float(np.float32(np.sqrt(np.float32(5.99))))
    K = pr["intrinsics"]
    p1, x1, st1, chi1 = oracle.ba_optimize(poses, fx, points, edges, K, huber, 5)
    _, front1 = oracle.ba_edge_chi2(p1, x1, edges, K)
    level0 = ~((chi1 > CHI2_MONO) | (front1 == 0))
    assert (np.abs(chi1 - CHI2_MONO) > BORDER).all(), "pick another seed: an observation sits on the chi2 gate after round 1"
    assert 20 < (~level0).sum() < len(edges) // 3
    # round 2: the level-0 edges only, no kernel; a landmark / camera without a level-0 edge simply does not move
    p2, x2, st2, chi2_act = oracle.ba_optimize(p1, fx, x1, edges[level0], K, 0.0, 10)
    chi2 = chi1.copy(); chi2[level0] = chi2_act                  # a level-1 edge keeps the chi2 of its last evaluation
    _, front2 = oracle.ba_edge_chi2(p2, x2, edges, K)
    assert (np.abs(chi2 - CHI2_MONO) > BORDER).all()
    rejected = [(k, mp) for (k, mp), c, f in zip(owner, chi2, front2) if c > CHI2_MONO or not f]

    W.welding_ba(main, adjust, fixed_l)
    for k, mp in rejected:
        sw.mirror_erase(W, k, mp)
    _check_tables(W)
    for i, k in enumerate(cams):
        g = W.get_kf(k)
        if fixed[i]:
            assert g["set_pose"] == 0 and np.array_equal(g["pose"], W.kf[k]["pose"])
        else:
            assert g["set_pose"] == 1 and _close_pose(g["pose"], p2[i]), f"keyframe {k}"
    assert W.get_kf(6)["set_pose"] == 0
    moved = 0
    for i, mp in enumerate(landmarks):
        g = W.get_mp(mp)
        if W.mp[mp]["bad"]:                                      # flagged bad by the erasures: the reference skips it
            assert g["set_pos"] == 0
            continue
        assert (g["set_pos"], g["update_normal"]) == (1, 1) and np.allclose(g["pos"], x2[i], rtol=3e-6, atol=3e-6), f"map point {mp}"
        moved += 1
    assert moved > 100 and st1["iterations"] == 5 and st2["iterations"] >= 2
    assert W.L.sw_map_change_index(W.h, 0) == 0                  # (the welding BA does not touch the change index)


# ------------------------------------------------------------------------------------------------------------ essential graph
def _ring_world(n, seed, bad=(), moved=None):
    """Keyframes on a drifted loop (synth.pose_graph), spanning tree i -> i - 1, covisibility links above and below the 100-point
    threshold, two old loop edges; map point l is observed by keyframes l % n (its reference keyframe) and (l + 1) % n."""
    pg = synth.pose_graph(n=n, noise=0.0, seed=seed, drift=0.01)
    W = sw.World()
    W.add_map(0)
    K = np.array([149, 149, 320, 240], np.float32)
    rng = np.random.default_rng(seed)
    for i in range(n):
        S = pg["S0"][i]
        pose = np.concatenate([S[4:7] / S[7], S[:4]]) if moved is None or i not in moved else moved[i]
        W.add_keyframe(0, 3 * i, pose, K, np.zeros(6, sw.KEYPOINT_DTYPE), bad=i in bad)
    sym = {}
    for i in range(n):
        for j in range(max(0, i - 4), min(n, i + 5)):
            if j != i:
                sym.setdefault((min(i, j), max(i, j)), int(rng.choice([40, 99, 100, 180, 260])))   # (symmetric, like KeyFrame::GetWeight)
    for i in range(n):
        nb = sorted([(j, sym[(min(i, j), max(i, j))]) for j in range(max(0, i - 4), min(n, i + 5)) if j != i], key=lambda t: -t[1])
        W.set_covisible(i, [j for j, _ in nb], [w for _, w in nb])
        if i > 0:
            W.set_parent(i, i - 1)
    W.add_loop_edge(n // 2, 2); W.add_loop_edge(n // 2 + 5, 4)
    for l in range(3 * n):
        m = W.add_mappoint(0, l, rng.uniform(-8, 8, 3))
        W.observe(l % n, m, l // n)
        W.observe((l + 1) % n, m, 3 + l // n)
    return W


def _weight(W, a, b):
    return dict(W.kf[a]["covis"]).get(b, 0)


def _solve_pg(capi, S, fixed, edges, fix_scale):
    ev = np.array([(i, j) for i, j, _ in edges], np.int32); em = np.array([m for _, _, m in edges], np.float64)
    return capi.pose_graph_optimize(np.array(S, np.float64), np.array(fixed, np.uint8), ev, em, fix_scale=fix_scale, iterations=20)


def _se3_tq_from_sim3(o, div_in_float):
    """Sim3 [sR t] -> SE3 [R t / s] as (t, q): the loop-closure form divides in float, the merge form in double."""
    q = np.asarray(o[:4], np.float64)
    t = (np.asarray(o[4:7], np.float32) / np.float32(o[7])) if div_in_float else (np.asarray(o[4:7], np.float64) / o[7])
    return np.concatenate([np.asarray(t, np.float64), q])


def _tq(S):
    return np.array(list(S[4:7]) + list(S[:4]))


@pytest.mark.parametrize("fix_scale", [False, True])
def test_essential_graph_after_loop_closure(capi, oracle, fix_scale):
    n = 36
    W = _ring_world(n, seed=7, bad={9})
    cur, loop = n - 1, 0
    ids = [W.kf[k]["id"] for k in range(n)]
    # LoopClosing::CorrectLoop hands over: the corrected Sim3 of the current keyframe's neighbourhood, their poses before the
    # correction, and the new connections the fused points created
    fixq = np.array([0.01, -0.02, 0.015, 1.0]); fixq /= np.linalg.norm(fixq)
    corr = [float(v) for v in fixq] + [0.3, -0.2, 0.1, 1.0 if fix_scale else 0.93]
    non_corrected = {k: sw.sim3_of_pose(W.kf[k]["pose"]) for k in (cur, cur - 1, cur - 2)}
    corrected = {k: sw.s_mul(corr, non_corrected[k]) for k in non_corrected}
    connections = [(cur, loop), (cur, 1), (cur - 1, loop), (cur - 1, 9)]
    for mp in (0, 1, 2):                                         # points LoopClosing already moved with the current keyframe
        W.L.sw_mp_set_corrected(W.h, mp, sw.C.c_ulong(ids[cur]), sw.C.c_ulong(ids[cur - 1]))
    # ---- the graph by the reference's rules (Optimizer.cc:1389-1652)
    verts = [k for k in range(n) if not W.kf[k]["bad"]]
    slot = {k: i for i, k in enumerate(verts)}
    before = [corrected.get(k, sw.sim3_of_pose(W.kf[k]["pose"])) for k in verts]
    uncorr = lambda k: non_corrected.get(k, before[slot[k]])
    edges, closing = [], set()
    for a, b in connections:
        if not ((a == cur and b == loop) or _weight(W, a, b) >= 100) or a not in slot or b not in slot:
            continue
        edges.append((slot[a], slot[b], sw.s_mul(before[slot[b]], sw.s_inv(before[slot[a]]))))
        closing.add((min(a, b), max(a, b)))
    for k in verts:
        Swi = sw.s_inv(uncorr(k))
        par = W.kf[k]["parent"]
        if par is not None and par in slot:
            edges.append((slot[k], slot[par], sw.s_mul(uncorr(par), Swi)))
        for l in sorted(W.kf[k]["loops"]):
            if l < k and l in slot:
                edges.append((slot[k], slot[l], sw.s_mul(uncorr(l), Swi)))
        for nb, w in W.kf[k]["covis"]:
            if w < 100 or nb == par or W.kf[nb]["parent"] == k or W.kf[nb]["bad"] or nb >= k or (min(k, nb), max(k, nb)) in closing:
                continue
            edges.append((slot[k], slot[nb], sw.s_mul(uncorr(nb), Swi)))
    fixed = [W.kf[k]["id"] == 0 for k in verts]
    So, st = _solve_pg(capi, before, fixed, edges, fix_scale)
    assert st["iterations"] >= 2 and st["chi2_final"] < 0.2 * st["chi2_initial"]

    W.essential_graph_loop(0, loop, cur, non_corrected, corrected, connections, fix_scale)
    for i, k in enumerate(verts):
        g = W.get_kf(k)
        assert g["set_pose"] == 1 and _close_pose(g["pose"], _se3_tq_from_sim3(So[i], True)), f"keyframe {k}"
    assert W.get_kf(9)["set_pose"] == 0
    for mp in range(len(W.mp)):
        ref = cur - 1 if mp < 3 else W.mp[mp]["ref"]
        P = [float(v) for v in W.mp[mp]["pos"]]
        exp = P if ref not in slot else sw.s_map(sw.s_inv(list(So[slot[ref]])), sw.s_map(before[slot[ref]], P))
        g = W.get_mp(mp)
        assert (g["set_pos"], g["update_normal"]) == (1, 1) and np.allclose(g["pos"], exp, rtol=2e-6, atol=2e-6), f"map point {mp}"
    assert W.L.sw_map_change_index(W.h, 0) == 1
    # the device solve itself against the oracle, on this graph: same start, same level reached (the numeric Jacobians bound anything tighter)
    ev = np.array([(i, j) for i, j, _ in edges], np.int32); em = np.array([m for _, _, m in edges], np.float64)
    _, sto = oracle.pose_graph_optimize(np.array(before), np.array(fixed, np.uint8), ev, em, fix_scale=fix_scale, iterations=20)
    assert abs(st["chi2_initial"] - sto[2]) <= 1e-9 * sto[2] and st["chi2_final"] <= 1.5 * sto[3] + 1e-12 and sto[3] <= 1.5 * st["chi2_final"] + 1e-12


def test_essential_graph_after_map_merge(capi, oracle):
    """Optimizer::OptimizeEssentialGraph(pCurKF, vpFixedKFs, vpFixedCorrectedKFs, vpNonFixedKFs, vpNonCorrectedMPs), Optimizer.cc:1653-1958."""
    n = 40
    # keyframes 0..7: the welding window of the map merged INTO (fixed, corrected pose only); 8..13: the window of the old map
    # (fixed; already moved by the merge transform, the pose before it kept in mTcwBefMerge); 14..: the rest of the old map
    fixed_l, fixed_corr, non_fixed = list(range(0, 8)), list(range(8, 14)), list(range(14, n)) + [10]   # (10 is named twice)
    W0 = _ring_world(n, seed=19)
    Tq = np.array([0.02, 0.01, -0.015, 1.0]); Tq /= np.linalg.norm(Tq)
    merge_inv = sw.s_inv([float(v) for v in Tq] + [0.25, 0.1, -0.15, 1.0])
    bef = {k: np.asarray(W0.kf[k]["pose"], np.float32) for k in fixed_corr}
    moved = {k: _tq(sw.s_mul(sw.sim3_of_pose(bef[k]), merge_inv)).astype(np.float32) for k in fixed_corr}   # Tcw_new = Tcw_old * merge^-1
    W = _ring_world(n, seed=19, bad={30}, moved=moved)
    for k in fixed_corr:
        W.set_bef_merge(k, bef[k], _tq(sw.s_inv(sw.sim3_of_pose(bef[k]))))
    mps = [m for m in range(len(W.mp)) if W.mp[m]["ref"] >= 8]      # the old map's points
    # ---- the graph by the reference's rules
    V, slot = [], {}

    def vertex(k, fx_):
        if k not in slot:
            slot[k] = len(V); V.append(dict(est=sw.sim3_of_pose(W.kf[k]["pose"]), fixed=fx_, good=False, badp=False))
        return V[slot[k]]
    for k in fixed_l:
        if not W.kf[k]["bad"]:
            v = vertex(k, True); v["corr_wc"] = sw.s_inv(sw.sim3_of_pose(W.kf[k]["pose"])); v["good"], v["badp"] = True, False
    entered = set()
    for k in fixed_corr:
        if not W.kf[k]["bad"]:
            v = vertex(k, True); v["corr_wc"] = sw.s_inv(sw.sim3_of_pose(W.kf[k]["pose"]))
            v["unc_cw"] = sw.sim3_of_pose(bef[k]); v["good"], v["badp"] = True, True
            entered.add(k)
    for k in non_fixed:
        if W.kf[k]["bad"] or k in entered:
            continue
        v = vertex(k, False); v["unc_cw"] = sw.sim3_of_pose(W.kf[k]["pose"]); v["good"], v["badp"] = False, True
        entered.add(k)
    allk = fixed_l + fixed_corr + non_fixed
    members = set(allk)
    edges = []
    for k in allk:
        if k not in slot:
            continue
        vi = V[slot[k]]
        Swi = sw.s_inv(vi["unc_cw"]) if vi["badp"] else [0., 0., 0., 1., 0., 0., 0., 1.]

        def link(j):
            if j not in slot:
                return
            vj = V[slot[j]]
            if vi["good"] and vj["good"]:
                Sjw = sw.s_inv(vj["corr_wc"])
            elif vi["badp"] and vj["badp"]:
                Sjw = vj["unc_cw"]
            else:
                return
            edges.append((slot[k], slot[j], sw.s_mul(Sjw, Swi)))
        par = W.kf[k]["parent"]
        if par is not None and par in members:
            link(par)
        for l in sorted(W.kf[k]["loops"]):
            if l in members and l < k:
                link(l)
        for nb, w in W.kf[k]["covis"]:
            if w < 100 or nb == par or W.kf[nb]["parent"] == k or nb in W.kf[k]["loops"] or nb not in members:
                continue
            if not W.kf[nb]["bad"] and nb < k:
                link(nb)
    So, st = _solve_pg(capi, [v["est"] for v in V], [v["fixed"] for v in V], edges, False)
    # (edges between two FIXED vertices carry the reference's own measurement expressions and dominate chi2 as a constant: the
    #  relative-decrease stopping rule fires early -- in the reference as well)
    assert st["iterations"] >= 2 and st["chi2_final"] < st["chi2_initial"]

    before_pose = {k: W.get_kf(k)["pose"].copy() for k in range(n)}
    W.essential_graph_merge(non_fixed[0], fixed_l, fixed_corr, non_fixed, mps)
    written = [k for k in non_fixed if not W.kf[k]["bad"]]
    for k in range(n):
        g = W.get_kf(k)
        if k in written:
            assert g["set_pose"] == 1 and _close_pose(g["pose"], _se3_tq_from_sim3(So[slot[k]], False)), f"keyframe {k}"
            assert np.array_equal(g["bef_merge"], before_pose[k])             # mTcwBefMerge <- the pose it had (also for 10: its own record is overwritten)
        else:
            assert g["set_pose"] == 0 and np.array_equal(g["pose"], before_pose[k])
    assert max(np.abs(W.get_kf(k)["pose"][:3] - before_pose[k][:3]).max() for k in written) > 1e-3
    # map points: P' = Twr(now) * (mTwcBefMerge)^-1 * P through the reference keyframe, where that keyframe has an "uncorrected" pose.
    # A point whose reference keyframe (30) is bad loses that observation -- one observer left: the point goes bad -- and is then carried
    # over through the remaining observer, exactly as the reference's loop does.
    for m in range(len(W.mp)):
        g = W.get_mp(m)
        ref = W.mp[m]["ref"]
        if ref == 30 and m in mps:
            ref = 31
            assert g["bad"] and W.mp_observations(m) == {}
        if m not in mps or not V[slot[ref]]["badp"]:
            assert g["set_pos"] == 0, m
            continue
        new_cw = sw.sim3_of_pose(W.get_kf(ref)["pose"])
        old_cw = sw.sim3_of_pose(W.get_kf(ref)["bef_merge"])
        exp = sw.s_map(sw.s_inv(new_cw), sw.s_map(old_cw, [float(v) for v in W.mp[m]["pos"]]))
        assert (g["set_pos"], g["update_normal"]) == (1, 1) and np.allclose(g["pos"], exp, rtol=1e-5, atol=1e-5), f"map point {m}"


# ------------------------------------------------------------------------------------------- PoseOptimization, OptimizeSim3
def test_pose_optimization(oracle):
    """Optimizer::PoseOptimization(Frame*): keypoints without a map point are skipped, outlier flags land on the keypoints."""
    pr = synth.ba_problem(n_kf=24, n_pts=500, k_obs=8, seed=4, radius=12.0)
    rng = np.random.default_rng(8)
    sel = np.flatnonzero(pr["edge_pose"] == 3)
    obs = pr["obs"][sel].copy()
    bad = rng.random(len(obs)) < 0.15
    obs[bad] += rng.choice([-1.0, 1.0], size=(int(bad.sum()), 2)) * 35.0
    octave = np.rint(-np.log(pr["inv_sigma2"][sel]) / (2 * np.log(1.2))).astype(np.int32)
    W = sw.World(); W.add_map(0)
    mps = [W.add_mappoint(0, i, pr["points_gt"][pr["edge_point"][e]]) for i, e in enumerate(sel)]
    N = len(sel) + len(sel) // 2                                  # every third keypoint carries no map point
    kps = np.zeros(N, sw.KEYPOINT_DTYPE); match = np.full(N, -1, np.int32)
    slots = np.sort(rng.choice(N, len(sel), replace=False))
    for j, s in enumerate(slots):
        kps[s]["x"], kps[s]["y"], kps[s]["octave"] = obs[j, 0], obs[j, 1], octave[j]
        match[s] = mps[j]
    f = W.add_frame(pr["poses"][3], pr["intrinsics"], kps)
    W.frame_set_matches(f, match, outlier=np.ones(N, np.uint8))
    pose32 = np.asarray(pr["poses"][3], np.float32)
    Xw = np.array([W.mp[m]["pos"] for m in mps], np.float64)
    ob32 = np.stack([kps["x"][slots], kps["y"][slots]], axis=1).astype(np.float64)
    w = W.tables[2][octave].astype(np.float64)
    po, oo, no = oracle.pose_optimize(pose32.astype(np.float64), Xw, ob32, w, pr["intrinsics"])
    n = W.pose_optimization(f)
    g = W.get_frame(f)
    assert n == no and 0.5 * len(sel) < no < len(sel)
    assert np.array_equal(g["outlier"][slots], oo)
    assert g["outlier"][match < 0].all()                         # untouched where there is no map point (they were set before the call)
    assert g["set_pose"] == 1 and _close_pose(g["pose"], po)
    assert np.array_equal(g["mp"], match)
    # fewer than three correspondences: returns 0 and leaves the pose alone
    f2 = W.add_frame(pr["poses"][3], pr["intrinsics"], kps[:4])
    W.frame_set_matches(f2, np.array([mps[0], -1, -1, -1], np.int32))
    assert W.pose_optimization(f2) == 0 and W.get_frame(f2)["set_pose"] == 0


def _rot32(q):
    """Eigen::Quaternionf::toRotationMatrix in float, operation by operation (tests/stubs/Eigen/Core)."""
    f = np.float32
    x, y, z, w = [f(v) for v in q]
    tx, ty, tz = f(2) * x, f(2) * y, f(2) * z
    twx, twy, twz, txx, txy, txz, tyy, tyz, tzz = tx * w, ty * w, tz * w, tx * x, ty * x, tz * x, ty * y, tz * y, tz * z
    return np.array([[f(1) - (tyy + tzz), txy - twz, txz + twy], [txy + twz, f(1) - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, f(1) - (txx + tyy)]], np.float32)


@pytest.mark.parametrize("all_points,fix_scale", [(False, False), (True, True)])
def test_optimize_sim3(oracle, all_points, fix_scale):
    """Optimizer::OptimizeSim3(pKF1, pKF2, vpMatches1, g2oS12, th2, bFixScale, H, bAllPoints): the correspondences are gathered
    from the keyframes' own tables; rejected pairs are cleared from vpMatches1."""
    from scipy.spatial.transform import Rotation as Rot
    from dvm_slam_amd.synth import _quat_from_rot, _rot_from_axis_angle
    rng = np.random.default_rng(13)
    N = 220
    K = np.array([149.0, 149.0, 320.0, 240.0])
    # two keyframes with their own world frames; P1c = s R P2c + t relates the two CAMERA frames
    R = _rot_from_axis_angle(np.array([0.05, -0.2, 0.1])); t = np.array([0.3, -0.1, 0.2]); s = 1.0 if fix_scale else 1.15
    T1 = np.concatenate([[0.1, 0.2, -0.1], _quat_from_rot(_rot_from_axis_angle(np.array([0.02, 0.03, -0.01])))]).astype(np.float32)
    T2 = np.concatenate([[-0.2, 0.05, 0.3], _quat_from_rot(_rot_from_axis_angle(np.array([-0.03, 0.01, 0.02])))]).astype(np.float32)
    P2c = np.c_[rng.uniform(-3, 3, N), rng.uniform(-2, 2, N), rng.uniform(4, 12, N)]
    P1c = (s * (R @ P2c.T)).T + t
    proj = lambda P: np.c_[K[0] * P[:, 0] / P[:, 2] + K[2], K[1] * P[:, 1] / P[:, 2] + K[3]]
    o1 = proj(P1c) + rng.normal(0, 0.6, (N, 2)); o2 = proj(P2c) + rng.normal(0, 0.6, (N, 2))
    out = rng.random(N) < 0.1
    o1[out] += rng.choice([-1, 1], (int(out.sum()), 2)) * 25.0

    def to_world(T, Pc):
        Rcw = Rot.from_quat(T[3:].astype(np.float64)).as_matrix()
        return ((Rcw.T @ (Pc - T[:3].astype(np.float64)).T).T).astype(np.float32)
    X1, X2 = to_world(T1, P1c), to_world(T2, P2c)
    W = sw.World(); W.add_map(0)
    oct1, oct2 = rng.integers(0, 8, N), rng.integers(0, 8, N)
    k1 = np.zeros(N, sw.KEYPOINT_DTYPE); k1["x"], k1["y"], k1["octave"] = o1[:, 0], o1[:, 1], oct1
    has_kp2 = np.ones(N, bool) if not all_points else rng.random(N) < 0.7
    idx2 = np.cumsum(has_kp2) - 1
    k2 = np.zeros(int(has_kp2.sum()), sw.KEYPOINT_DTYPE); k2["x"], k2["y"], k2["octave"] = o2[has_kp2, 0], o2[has_kp2, 1], oct2[has_kp2]
    a = W.add_keyframe(0, 1, T1, K, k1); b = W.add_keyframe(0, 2, T2, K, k2)
    m1 = [W.add_mappoint(0, i, X1[i]) for i in range(N)]
    m2 = [W.add_mappoint(0, 1000 + i, X2[i], bad=(i == 7)) for i in range(N)]
    for i in range(N):
        W.observe(a, m1[i], i)
        if has_kp2[i]:
            W.observe(b, m2[i], int(idx2[i]))
    matches = np.array(m2, np.int32)
    matches[5] = -1                                               # keypoint 5 has no match; the match of keypoint 7 is a bad point
    S0 = np.r_[_quat_from_rot(_rot_from_axis_angle(rng.normal(0, 0.03, 3)) @ R), t + rng.normal(0, 0.05, 3), s * (1.0 if fix_scale else 1.04)]
    # ---- the correspondences by the reference's rules (Optimizer.cc:1990-2070): camera-frame points through the float rotation
    # matrix / translation of each keyframe, keypoints (or, with bAllPoints, the normalised projection of point 2)

    def cam_frame(T, X):
        Rm = _rot32(T[3:])
        return np.array([[(Rm[r, 0] * x[0] + (Rm[r, 1] * x[1] + Rm[r, 2] * x[2])) + T[r] for r in range(3)] for x in X], np.float32).astype(np.float64)
    C1, C2 = cam_frame(T1, X1), cam_frame(T2, X2)
    use = [i for i in range(N) if matches[i] >= 0 and i != 7 and (has_kp2[i] or all_points) and not np.float32(C2[i, 2]) < 0]
    px1 = np.stack([k1["x"][use], k1["y"][use]], axis=1).astype(np.float64)
    w1 = W.tables[2][oct1[use]].astype(np.float64)
    px2, w2 = [], []
    for i in use:
        if has_kp2[i]:
            j = int(idx2[i]); px2.append((float(k2["x"][j]), float(k2["y"][j]))); w2.append(float(W.tables[2][k2["octave"][j]]))
        else:
            iz = np.float32(1) / np.float32(C2[i, 2])
            px2.append((float(np.float32(C2[i, 0]) * iz), float(np.float32(C2[i, 1]) * iz))); w2.append(float(W.tables[2][0]))
    So, io, no = oracle.optimize_sim3(S0, fix_scale, C1[use], C2[use], px1, np.array(px2), w1, np.array(w2), K, K, 10.0)
    n, m_out, Sg = W.optimize_sim3(a, b, matches, S0, 10.0, fix_scale, all_points)
    assert n == no and n > 0.5 * has_kp2.sum()
    exp = matches.copy()
    for e, i in enumerate(use):
        if not io[e]:
            exp[i] = -1
    assert np.array_equal(m_out, exp)
    assert np.abs(Sg - So).max() < 1e-6, np.abs(Sg - So).max()


@pytest.mark.parametrize("fix_scale", [False, True])
def test_sim3_solver_class(oracle, fix_scale):
    """ORB_SLAM3::Sim3Solver as LoopClosing uses it -- constructor (gathers the correspondences from the keyframes' tables),
    SetRansacParameters, iterate(n, bNoMore, vbInliers, nInliers, bConverge) in a loop, the getters -- against the oracle's
    hypotheses evaluated on the SAME minimal sets (libc's rand() after srand(seed), drawn as Sim3Solver.cc:166-181 draws them)
    with the reference's sequential rule applied in Python."""
    import ctypes
    from scipy.spatial.transform import Rotation as Rot
    from dvm_slam_amd.synth import _quat_from_rot, _rot_from_axis_angle
    rng = np.random.default_rng(5)
    N = 160
    K = np.array([149.0, 149.0, 320.0, 240.0])
    R = _rot_from_axis_angle(np.array([0.1, -0.15, 0.05])); t = np.array([0.4, -0.2, 0.1]); s = 1.0 if fix_scale else 1.2
    T1 = np.concatenate([[0.1, 0.2, -0.1], _quat_from_rot(_rot_from_axis_angle(np.array([0.02, 0.03, -0.01])))]).astype(np.float32)
    T2 = np.concatenate([[-0.2, 0.05, 0.3], _quat_from_rot(_rot_from_axis_angle(np.array([-0.03, 0.01, 0.02])))]).astype(np.float32)
    P2c = np.c_[rng.uniform(-3, 3, N), rng.uniform(-2, 2, N), rng.uniform(4, 12, N)]
    P1c = (s * (R @ P2c.T)).T + t
    bad = rng.random(N) < 0.35                                    # wrong associations: RANSAC has something to reject
    P1c[bad] += rng.normal(0, 1.5, (int(bad.sum()), 3))

    def to_world(T, Pc):
        Rcw = Rot.from_quat(T[3:].astype(np.float64)).as_matrix()
        return ((Rcw.T @ (Pc - T[:3].astype(np.float64)).T).T).astype(np.float32)
    X1, X2 = to_world(T1, P1c), to_world(T2, P2c)
    W = sw.World(); W.add_map(0)
    oct1, oct2 = rng.integers(0, 8, N), rng.integers(0, 8, N)
    k1 = np.zeros(N, sw.KEYPOINT_DTYPE); k1["octave"] = oct1
    k2 = np.zeros(N, sw.KEYPOINT_DTYPE); k2["octave"] = oct2
    a = W.add_keyframe(0, 1, T1, K, k1); b = W.add_keyframe(0, 2, T2, K, k2)
    m1 = [W.add_mappoint(0, i, X1[i]) for i in range(N)]
    m2 = [W.add_mappoint(0, 1000 + i, X2[i], bad=(i == 9)) for i in range(N)]
    for i in range(N):
        W.observe(a, m1[i], i); W.observe(b, m2[i], i)
    matches = np.array(m2, np.int32); matches[4] = -1
    use = [i for i in range(N) if matches[i] >= 0 and i != 9]     # the constructor's rules (:81-96)

    def cam_frame(T, X):
        Rm = _rot32(T[3:])
        return np.array([[(Rm[r, 0] * x[0] + (Rm[r, 1] * x[1] + Rm[r, 2] * x[2])) + T[r] for r in range(3)] for x in X], np.float32)
    C1, C2 = cam_frame(T1, X1)[use], cam_frame(T2, X2)[use]
    sig2 = W.tables[1].astype(np.float32)                          # mvLevelSigma2
    e1 = np.array([float(int(9.210 * float(sig2[oct1[i]]))) for i in use], np.float32)    # vector<size_t>: truncated
    e2 = np.array([float(int(9.210 * float(sig2[oct2[i]]))) for i in use], np.float32)
    n = len(use)
    min_inl, max_its, per_call, seed = 20, 300, 20, 1234
    # SetRansacParameters (:135-152)
    eps = np.float32(min_inl) / np.float32(n)
    its = int(np.ceil(np.log(1 - 0.99) / np.log(1 - float(eps) ** 3)))
    max_its_eff = max(1, min(its, max_its))
    # the expectation draws with the reference's own DUtils::Random when oracle/_ref holds it (and then the driver is linked
    # against the same object code: sw_dutils_is_reference), with the restatement tests/test_ref_dutils.py pins otherwise
    ref_rng = ref_dutils.load_reference()
    libc = ctypes.CDLL("libc.so.6")
    if ref_rng is not None:
        assert W.L.sw_dutils_is_reference() == 1, "oracle/_ref/libdutils_ref.so is there but the shim driver was built without it"
        ref_rng.seed(seed)
        rand_int = ref_rng.random_int
    else:
        libc.srand(seed)

        def rand_int(lo, hi):
            return ref_dutils.python_random_int(libc, lo, hi)
    done, exp = 0, None
    best_n = 0
    calls = 0
    while exp is None and done < max_its_eff:
        H = min(per_call, max_its_eff - done)
        tr = []
        for _ in range(H):
            avail = list(range(n))
            trip = []
            for _i in range(3):
                r_ = rand_int(0, len(avail) - 1)
                trip.append(avail[r_]); avail[r_] = avail[-1]; avail.pop()
            tr.append(trip)
        calls += 1
        T, nin, mask = oracle.sim3_hypotheses(C1, C2, e1, e2, K, K, np.array(tr, np.int32), fix_scale)
        for h in range(H):
            done += 1
            if nin[h] >= best_n:
                best_n = int(nin[h]); best = (T[h].copy(), mask[h].copy())
                if nin[h] > min_inl:
                    exp = (T[h].copy(), mask[h].copy(), int(nin[h])); break
    assert exp is not None, "the scene should converge"
    Tg, est, inl, info = W.sim3_solver(a, b, matches, fix_scale, min_inl, max_its, per_call, seed, N)
    assert info["converged"] and not info["no_more"] and info["calls"] == calls
    assert abs(info["n_inliers"] - exp[2]) <= 2                                   # Horn's closed form: tolerance parity
    want = np.zeros(N, bool); want[np.array(use)[exp[1].astype(bool)]] = True
    assert (inl != want).sum() <= 2 and not inl[4] and not inl[9]
    sR = exp[0][0] * exp[0][1:10].reshape(3, 3)
    assert np.abs(Tg[:3, :3] - sR).max() < 1e-4 and np.abs(Tg[:3, 3] - exp[0][10:13]).max() < 1e-4 and np.array_equal(Tg[3], [0, 0, 0, 1])
    assert np.abs(est[:9].reshape(3, 3) - exp[0][1:10].reshape(3, 3)).max() < 1e-4 and abs(est[12] - exp[0][0]) < 1e-4
    assert abs(est[12] - s) < 0.05 and (fix_scale is False or est[12] == 1.0)
