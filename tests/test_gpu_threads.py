"""The boundary's threading contract: one extractor instance per tracking thread, the static Optimizer
functions called concurrently from Tracking (PoseOptimization), LocalMapping (local BA) and LoopClosing (OptimizeSim3).
Five host threads hammer the C ABI at once (ctypes drops the GIL during the calls); every result must equal the one the
same call gives when it runs alone."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_callers(capi, oracle, frames):
    from dvm_slam_amd import synth
    orc = oracle.OrbOracle()
    ref_ext = [orc.extract(f) for f in frames]
    pr = synth.ba_problem(n_kf=14, n_pts=400, seed=5)
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    delta = float(np.sqrt(5.991))

    def run_ba():
        ba = capi.BundleAdjuster()
        ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
        st = ba.optimize(6)
        p, x = ba.result()
        ba.close()
        return st["trials"], p, x

    ba_ref = run_ba()
    rng = np.random.default_rng(0)
    Xw = rng.uniform(-3, 3, (300, 3)) + [0, 0, 8]
    K = np.array([500.0, 500.0, 320.0, 240.0])
    obs = np.stack([K[0] * Xw[:, 0] / Xw[:, 2] + K[2], K[1] * Xw[:, 1] / Xw[:, 2] + K[3]], 1) + rng.normal(0, 0.5, (300, 2))
    pose0 = np.array([0.05, -0.03, 0.1, 0.0, 0.0, 0.0, 1.0])
    po_ref = capi.pose_optimize(pose0[None], Xw[None], obs[None], np.ones((1, 300)), [300], K)
    errors = []

    def guard(fn):
        def w():
            try:
                fn()
            except Exception as ex:   # noqa: BLE001
                errors.append(repr(ex))
        return w

    def t_extract(k):
        def f():
            ext = capi.OrbExtractor(max_batch=1)
            for it in range(12):
                i = (it + k) % len(frames)
                n, kp, d, m = ext.extract(frames[i])
                n_o, k_o, d_o, m_o = ref_ext[i]
                assert (n, m) == (n_o, m_o) and np.array_equal(d, d_o) and np.array_equal(kp["x"], k_o["x"]) and np.array_equal(kp["angle"], k_o["angle"])
            ext.close()
        return f

    big_img = synth.small_image(77, 600, 800)
    big_ref = oracle.OrbOracle(8000, 1.2, 8, 12, 5).extract(big_img, cap=4 * 8000 + 256)

    def t_extract_big():     # a configuration whose octree needs > 48 KB of LDS, next to the default one (shared per-kernel LDS cap)
        ext = capi.OrbExtractor(8000, 1.2, 8, 12, 5, max_batch=1)
        for _ in range(6):
            n, kp, d, m = ext.extract(big_img)
            assert (n, m) == (big_ref[0], big_ref[3]) and np.array_equal(d, big_ref[2])
        ext.close()

    def t_ba():
        for _ in range(4):
            tr, p, x = run_ba()
            assert tr == ba_ref[0] and np.array_equal(p, ba_ref[1]) and np.array_equal(x, ba_ref[2])

    def t_pose():
        for _ in range(40):
            p, o, n = capi.pose_optimize(pose0[None], Xw[None], obs[None], np.ones((1, 300)), [300], K)
            assert np.array_equal(p, po_ref[0]) and np.array_equal(o, po_ref[1]) and n[0] == po_ref[2][0]

    threads = [threading.Thread(target=guard(f)) for f in (t_extract(0), t_extract(1), t_extract_big, t_ba, t_pose)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors


def test_keyframe_database_concurrent_add_and_query(capi):
    """dvm_host::KeyFrameDatabase under the reference's threading (LocalMapping adds / erases keyframes while LoopClosing and
    the merge callback query): three threads on ONE database.  Queries against map 0 -- which nobody modifies -- must give the
    same answer as on a quiet database, whatever map 1 is going through (reallocation of the device CSR included)."""
    from kfdb_scene import fill, make_db_scene
    kfs = make_db_scene(5, n_maps=2, kf_per_map=30, n_words=3000, words_per_kf=120)
    quiet = capi.HostKeyFrameDatabase()
    fill(quiet, kfs)
    qs = [k for k in kfs if k["map_id"] == 1][:6]     # queries from map 1's keyframes against map 0
    want = [quiet.detect_merge_possibility(q["ids"], q["vals"], q["uuid"], 0) for q in qs]
    quiet.close()
    db = capi.HostKeyFrameDatabase()
    fill(db, kfs)
    errors, stop = [], threading.Event()

    def adder():
        try:
            rng = np.random.default_rng(1)
            for it in range(400):
                ids = np.unique(rng.integers(0, 3000, 400)).astype(np.int32)      # big vectors: force the CSR to grow
                s = db.add(ids, rng.random(len(ids)), 1, int(rng.integers(1, 2 ** 62)), 1000 + it)
                assert s >= 0
                if it % 3 == 0:
                    db.erase(s)
        except Exception as ex:   # noqa: BLE001
            errors.append(repr(ex))
        finally:
            stop.set()

    def querier():
        try:
            while not stop.is_set():
                for q, w in zip(qs, want):
                    assert db.detect_merge_possibility(q["ids"], q["vals"], q["uuid"], 0) == w
        except Exception as ex:   # noqa: BLE001
            errors.append(repr(ex))
            stop.set()

    threads = [threading.Thread(target=adder), threading.Thread(target=querier), threading.Thread(target=querier)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    db.close()
    assert not errors, errors
    with pytest.raises(capi.DvmError):     # uuid 0 is the reset value of mnPlaceRecognitionQuery: rejected, not silently blind
        db2 = capi.HostKeyFrameDatabase()
        try:
            db2.detect_merge_possibility(qs[0]["ids"], qs[0]["vals"], 0, 0)
        finally:
            db2.close()


def test_pooled_extractor_equals_one_extractor_per_frame(capi):
    """dvm_orb_pool_*: K threads hand their frames to ONE shared extractor, which runs frames that arrive together as one batch; every
    caller must get exactly what dvm_orb_extract gives for its frame -- whatever batch it happened to ride in (mixed image sizes included:
    a frame of another shape waits for the next batch)."""
    import threading
    from dvm_slam_amd import synth
    frames = synth.frame_stream(12)
    small = [synth.small_image(50 + i, 240, 320) for i in range(4)]
    ref_ext = capi.OrbExtractor(max_batch=1)
    ref = [ref_ext.extract(f) for f in frames]
    ref_small = [ref_ext.extract(f) for f in small]
    ref_ext.close()
    pool = capi.OrbPool(max_batch=8, window_us=200)
    K, per = 8, 24
    errors, sizes = [], []

    def agent(k):
        try:
            for i in range(per):
                if k == 7 and i % 3 == 0:            # one agent with another camera now and then
                    j = (i // 3) % 4
                    n, kp, d, m, bs = pool.extract(small[j])
                    r = ref_small[j]
                else:
                    j = (k + i) % 12
                    n, kp, d, m, bs = pool.extract(frames[j])
                    r = ref[j]
                sizes.append(bs)
                if n != r[0] or m != r[3] or kp.tobytes() != r[1].tobytes() or not np.array_equal(d, r[2]):
                    errors.append((k, i, n, r[0]))
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    ths = [threading.Thread(target=agent, args=(k,)) for k in range(K)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ths), "a pooled call did not return"
    assert not errors, errors[:3]
    assert max(sizes) > 1, "the frames never shared a batch"
    # a lone caller still works (pays the window)
    n, kp, d, m, bs = pool.extract(frames[3])
    assert bs == 1 and n == ref[3][0] and kp.tobytes() == ref[3][1].tobytes()
    pool.close()


def test_pooled_pose_optimization_equals_single_calls(capi):
    """dvm_pose_pool_*: K threads' PoseOptimization calls run as one launch when they arrive together; every caller gets the bits
    dvm_pose_optimize gives for its frame alone (frames of different sizes and, for one agent, another camera)."""
    import threading
    import bench_legs
    cases = [bench_legs._pose_case(300 + i, n_pts=int(n)) for i, n in enumerate((300, 120, 700, 40, 300, 900, 9, 250))]
    ref = []
    for c in cases:
        p, o, ni = capi.pose_optimize(c[0][None], c[1][None], c[2][None], c[3][None], np.array([len(c[1])], np.int32), c[4], 0)
        ref.append((p[0].copy(), o[0][:len(c[1])].copy(), int(ni[0])))
    other_cam = np.array(cases[0][4], np.float64).copy(); other_cam[0] *= 1.01
    c0 = cases[0]
    p, o, ni = capi.pose_optimize(c0[0][None], c0[1][None], c0[2][None], c0[3][None], np.array([len(c0[1])], np.int32), other_cam, 0)
    ref_other = (p[0].copy(), o[0][:len(c0[1])].copy(), int(ni[0]))
    pool = capi.PosePool(max_batch=8, window_us=200)
    errors, sizes = [], []

    def agent(k):
        try:
            for i in range(20):
                if k == 5 and i % 4 == 0:
                    po, ol, ni, bs = pool.optimize(c0[0], c0[1], c0[2], c0[3], other_cam)
                    r = ref_other
                else:
                    j = (k + i) % len(cases)
                    c = cases[j]
                    po, ol, ni, bs = pool.optimize(c[0], c[1], c[2], c[3], c[4])
                    r = ref[j]
                sizes.append(bs)
                if po.tobytes() != r[0].tobytes() or not np.array_equal(ol, r[1]) or ni != r[2]:
                    errors.append((k, i))
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    ths = [threading.Thread(target=agent, args=(k,)) for k in range(8)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ths), "a pooled call did not return"
    assert not errors, errors[:3]
    assert max(sizes) > 1, "the calls never shared a launch"
    pool.close()


def test_pooled_search_by_projection_equals_single_calls(capi, oracle):
    """dvm_match_pool_* behind dvmh_set_match_pool: K threads' SearchByProjection(Cur, Last) calls share launches; every call returns what
    the per-thread call returns (which the oracle confirms), sparse and crowded frames, with and without keypoints claimed at entry."""
    import threading
    from matcher_scene import make_scene
    scenes = [make_scene(oracle, 20 + i, n_last=int(nl), n_cur=int(nc)) for i, (nl, nc) in enumerate(((1000, 1100), (400, 500), (1000, 300), (1500, 1800)))]
    ths_ = (15.0, 7.0, 40.0, 15.0)
    for sc in scenes[:2]:
        sc["mp_c"] = np.full_like(sc["mp_c"], -1)     # nothing claimed at entry (the TrackWithMotionModel case): no skip array travels
    ref = [capi.search_by_projection_frames(th=t, **sc)[:2] for sc, t in zip(scenes, ths_)]
    for (n_g, mp_g), sc, t in zip(ref, scenes, ths_):
        n_o, mp_o = oracle.search_by_projection_frames(th=t, **sc)
        assert n_g == n_o and np.array_equal(mp_g, mp_o)
    pool = capi.MatchPool(max_batch=8, kp_cap=2048, q_cap=2048, window_us=200)
    pool.use()
    errors = []

    def agent(k):
        try:
            for i in range(16):
                j = (k + i) % len(scenes)
                n, mp, _ = capi.search_by_projection_frames(th=ths_[j], **scenes[j])
                if n != ref[j][0] or not np.array_equal(mp, ref[j][1]):
                    errors.append((k, i, j))
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    try:
        ths = [threading.Thread(target=agent, args=(k,)) for k in range(8)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in ths), "a pooled call did not return"
        assert not errors, errors[:3]
    finally:
        pool.close()
    n, mp, _ = capi.search_by_projection_frames(th=ths_[0], **scenes[0])   # back on the per-thread call
    assert n == ref[0][0] and np.array_equal(mp, ref[0][1])
