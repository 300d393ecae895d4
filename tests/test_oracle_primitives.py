"""Pins the CPU oracle (oracle/*.cpp) against independent brute-force numpy restatements
(tests/ref_numpy.py) and against constants derivable from the reference source alone.
The reference itself pins nothing: these known-answer tests are authored here."""
import math

import numpy as np
import pytest

import ref_numpy as ref


def test_constructor_tables_640x480(oracle):
    """ORBextractor.cc:282-339 with (1000, 1.2, 8, 20, 7): values computed in SURVEY.md section 8."""
    o = oracle.OrbOracle()
    t = o.tables()
    assert list(t["nfeat"]) == [217, 181, 151, 126, 105, 87, 73, 60] and int(t["nfeat"].sum()) == 1000
    assert list(t["umax"]) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3] == ref.umax_table()
    sc = np.float32(1.0)
    for i in range(8):
        assert t["scale"][i] == sc
        assert t["inv_scale"][i] == np.float32(1.0) / sc and t["sigma2"][i] == sc * sc
        sc = np.float32(np.float64(sc) * np.float64(np.float32(1.2)))
    o.extract(np.zeros((480, 640), np.uint8))
    dims = [o.level_dims(l) for l in range(8)]
    assert dims == [(480, 640), (400, 533), (333, 444), (278, 370), (231, 309), (193, 257), (161, 214), (134, 179)]
    assert sum(r * c for r, c in dims) == 950532


def test_gaussian_kernel_and_blur(oracle):
    assert list(oracle.gaussian_kernel7()) == [18, 34, 48, 56, 48, 34, 18] == ref.gaussian_kernel7_ed()
    rng = np.random.default_rng(0)
    for shape in [(23, 31), (40, 40), (8, 64)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        assert np.array_equal(oracle.gaussian_blur7(img), ref.gaussian_blur7_fixed(img))
    flat = np.full((20, 20), 201, np.uint8)
    assert np.array_equal(oracle.gaussian_blur7(flat), flat)  # kernel sums to 256 exactly


@pytest.mark.parametrize("src_shape,dst", [((480, 640), (533, 400)), ((400, 533), (444, 333)), ((37, 53), (44, 31)),
                                           ((20, 20), (17, 17)), ((31, 47), (60, 40))])
def test_resize_linear(oracle, src_shape, dst):
    rng = np.random.default_rng(src_shape[0])
    src = rng.integers(0, 256, src_shape, dtype=np.uint8)
    got = oracle.resize_linear(src, dst[0], dst[1])
    assert np.array_equal(got, ref.resize_linear_u8(src, dst[0], dst[1]))
    # sanity vs exact bilinear in float: fixed point stays within 1 grey level
    sh, sw = src_shape
    fx = np.clip((np.arange(dst[0]) + 0.5) * sw / dst[0] - 0.5, 0, sw - 1)
    fy = np.clip((np.arange(dst[1]) + 0.5) * sh / dst[1] - 0.5, 0, sh - 1)
    x0 = np.floor(fx).astype(int); y0 = np.floor(fy).astype(int)
    x1 = np.minimum(x0 + 1, sw - 1); y1 = np.minimum(y0 + 1, sh - 1)
    ax = (fx - x0)[None, :]; ay = (fy - y0)[:, None]
    s = src.astype(np.float64)
    exact = (s[y0][:, x0] * (1 - ax) + s[y0][:, x1] * ax) * (1 - ay) + (s[y1][:, x0] * (1 - ax) + s[y1][:, x1] * ax) * ay
    assert np.abs(got.astype(np.float64) - exact).max() <= 1.01


def test_fast_vs_bruteforce_definition(oracle):
    rng = np.random.default_rng(42)
    imgs = [rng.integers(0, 256, (22, 27), dtype=np.uint8), (rng.integers(0, 40, (25, 25)) + 100).astype(np.uint8)]
    blob = np.full((24, 24), 50, np.uint8); blob[8:15, 9:16] = 200; blob[3, 3] = 255
    imgs.append(blob)
    for img in imgs:
        for th in (7, 20, 60):
            xo, yo, so = oracle.fast9_16(img, th)
            xr, yr, sr = ref.fast_detect(img, th)
            assert np.array_equal(xo, xr) and np.array_equal(yo, yr) and np.array_equal(so, sr), th
    # closed form used by oracle and GPU: score = max(A,B) - 1 equals "largest t that is still a corner"
    img = imgs[0]
    sc = ref.fast_score_bruteforce(img)
    xo, yo, so = oracle.fast9_16(img, 1)
    assert all(sc[y, x] == s for x, y, s in zip(xo, yo, so))


def test_fast_too_small_roi(oracle):
    assert len(oracle.fast9_16(np.zeros((6, 30), np.uint8), 20)[0]) == 0


def test_cvround_atan2_sincos(oracle):
    L = oracle.lib()
    for v, e in [(0.5, 0), (1.5, 2), (2.5, 2), (-0.5, 0), (-1.5, -2), (3.4999, 3), (-2.5001, -3)]:
        assert L.orc_cv_round(v) == e
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = rng.integers(-70000, 70000, 2)
        a = oracle.fast_atan2(float(y), float(x))
        t = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(a - t); d = min(d, 360 - d)
        assert d < 0.3 or (x == 0 and y == 0)
    assert oracle.fast_atan2(0.0, 1.0) == 0.0 and oracle.fast_atan2(0.0, -1.0) == 180.0
    # sincos spec vs libm float cos/sin: at most 1 ulp apart, almost always identical
    ang = np.linspace(0, 360, 40001, dtype=np.float32)
    mism = 0
    for a in ang[::7]:
        c, s = oracle.sincos_deg(float(a))
        rad = np.float32(a) * np.float32(math.pi / np.float32(180.0))
        c0, s0 = np.float32(math.cos(float(rad))), np.float32(math.sin(float(rad)))
        for got, want in ((c, c0), (s, s0)):
            if np.float32(got) != want:
                mism += 1
                assert abs(float(got) - float(want)) <= np.spacing(np.float32(abs(want)) + np.float32(1e-30)) * 1.01
    assert mism <= 2  # correctly rounded on both sides for all but a handful of arguments


def test_ic_angle_and_descriptor(oracle):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    pat = np.array([int(v) for line in open(oracle._HERE + "/../dvm_slam_amd/csrc/orb_pattern_31.inc") if not line.startswith("//")
                    for v in line.strip().strip(",").split(",") if v], np.int32).reshape(256, 4)
    assert pat.shape == (256, 4) and np.abs(pat).max() <= 13
    for (cx, cy) in [(32, 32), (20, 40), (41, 23)]:
        m01, m10 = ref.ic_moments(img, cx, cy)
        ang = oracle.ic_angle(img, cx, cy)
        assert ang == oracle.fast_atan2(float(m01), float(m10))
        blur = oracle.gaussian_blur7(img)
        d = oracle.brief_descriptor(blur, cx, cy, ang)
        c, s = oracle.sincos_deg(ang)
        c, s = np.float32(c), np.float32(s)
        bits = []
        for (x0, y0, x1, y1) in pat:
            def val(x, y):
                x, y = np.float32(x), np.float32(y)
                iy = int(np.rint(np.float32(np.float32(x * s) + np.float32(y * c))))
                ix = int(np.rint(np.float32(np.float32(x * c) - np.float32(y * s))))
                return int(blur[cy + iy, cx + ix])
            bits.append(1 if val(x0, y0) < val(x1, y1) else 0)
        want = np.packbits(np.array(bits, np.uint8).reshape(32, 8)[:, ::-1], axis=1).ravel()  # LSB-first per byte
        assert np.array_equal(d, want)


def test_octree_vs_python_lists(oracle):
    rng = np.random.default_rng(4)
    checked = 0
    for trial in range(60):
        n = int(rng.integers(1, 400))
        W, H = int(rng.integers(60, 640)), int(rng.integers(60, 480))
        if round(W / H) < 1:
            continue
        pts = set()
        while len(pts) < n:
            pts.add((int(rng.integers(0, W)), int(rng.integers(0, H))))
        pts = sorted(pts, key=lambda p: (p[1] // 40, p[0] // 40, p[1], p[0]))  # cell-major like the extractor
        xs = np.array([p[0] for p in pts], np.int32); ys = np.array([p[1] for p in pts], np.int32)
        sc = rng.integers(7, 120, n).astype(np.int32)
        N = int(rng.integers(1, 150))
        want, ties = ref.distribute_octree(xs, ys, sc, 16, 16 + W, 16, 16 + H, N)
        if ties:
            continue  # std::sort's order among equal keys is implementation-defined; not comparable to Python's sort
        got = oracle.distribute_octree(xs, ys, sc, 16, 16 + W, 16, 16 + H, N)
        assert np.array_equal(got, want), trial
        checked += 1
    assert checked >= 15


def test_hamming_and_grid(oracle):
    rng = np.random.default_rng(3)
    A = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    B = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    D = oracle.hamming_matrix(A, B)
    for i in (0, 7, 39):
        for j in (0, 11, 49):
            assert D[i, j] == ref.descriptor_distance(A[i], B[j])
    assert oracle.hamming_matrix(A, A).diagonal().max() == 0
    assert oracle.hamming_matrix(np.zeros((1, 32), np.uint8), np.full((1, 32), 255, np.uint8))[0, 0] == 256
    kps = np.zeros(300, oracle.KP_DTYPE)
    kps["x"] = rng.uniform(-3, 643, 300).astype(np.float32)
    kps["y"] = rng.uniform(-3, 483, 300).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, 300)
    g = oracle.Grid(kps)
    for _ in range(60):
        x, y = rng.uniform(-50, 700), rng.uniform(-50, 530)
        r = float(rng.choice([1.0, 8.0, 40.0, 300.0]))
        lo, hi = int(rng.integers(-1, 8)), int(rng.integers(-1, 8))
        assert list(g.features_in_area(x, y, r, lo, hi)) == ref.grid_features_in_area(kps, x, y, r, lo, hi)


def test_extract_invariants(oracle, frames):
    o = oracle.OrbOracle()
    n, k, d, mono = o.extract(frames[0])
    assert mono == 0 and 1000 <= n <= 1000 + 3 * 8            # every level may overshoot its quota by <= 3 nodes
    assert (k["class_id"] == -1).all() and (k["angle"] >= 0).all() and (k["angle"] <= 360).all()
    assert list(k["octave"]) == sorted(k["octave"], reverse=True)  # reverse fill: level 7 first (Appendix B.1)
    assert d.shape == (n, 32)
    # lapping area {0,0}: nothing is "lapping" -> forward order, monoIndex == n
    n2, k2, d2, mono2 = o.extract(frames[0], lap=(0, 0))
    assert mono2 == n2 == n and np.array_equal(k2["octave"], k["octave"][::-1]) and np.array_equal(d2, d[::-1])
    assert o.extract(np.zeros((0, 0), np.uint8))[0] == -1


def test_search_by_projection_frames_oracle_semantics(oracle):
    """Whole-function SearchByProjection(Cur, Last) restatement: independent numpy re-derivation on a scene without
    competing queries (where the sequential claim rule cannot matter) + the rotation-histogram filter."""
    from matcher_scene import make_scene
    po = oracle
    sc = make_scene(po, 7, dup_frac=0.0, zero_obs_frac=0.0)
    sc["mp_c"][:] = -1
    n, mp = po.search_by_projection_frames(th=15.0, check_ori=False, **sc)
    # brute force: for every valid last-frame map point, best current keypoint in the window, in order, with claims
    kc, dc, kl, mps = sc["kps_c"], sc["desc_c"], sc["kps_l"], sc["mps"]
    K = sc["K"]
    Xc_all = po.se3_act(sc["Tcw"], mps["pos"])     # the projection arithmetic itself is pinned in test_pose_arithmetic.py
    grid = po.Grid(kc)
    ref = np.full(len(kc), -1, np.int32); cnt = 0
    for i in range(len(kl)):
        if sc["mp_l"][i] < 0 or sc["outlier_l"][i]:
            continue
        X = mps["pos"][i]
        xc = Xc_all[i]
        if xc[2] < 0:
            continue
        u = np.float32(np.float32(K[0] * xc[0]) / xc[2]) + K[2]; v = np.float32(np.float32(K[1] * xc[1]) / xc[2]) + K[3]
        if not (0 <= u <= 640 and 0 <= v <= 480):
            continue
        o = int(kl["octave"][i])
        cand = grid.features_in_area(u, v, np.float32(15.0) * sc["scale_factors"][o], o - 1, o + 1)
        best, bi = 256, -1
        for j in cand:
            if ref[j] >= 0:
                continue
            dd = int(np.unpackbits(mps["desc"][i] ^ dc[j]).sum())
            if dd < best:
                best, bi = dd, j
        if best <= 100:
            ref[bi] = i; cnt += 1
    assert n == cnt and np.array_equal(mp, ref)
    # with the orientation check: ~20 % of true matches have a random rotation -> most of them are removed
    n2, mp2 = po.search_by_projection_frames(th=15.0, check_ori=True, **sc)
    assert 0.6 * n < n2 < n
    kept = mp2 >= 0
    assert np.array_equal(mp2[kept], mp[kept])


def test_search_by_projection_points_oracle_semantics(oracle):
    """SearchByProjection(F, vpMapPoints): with no competing points the result must equal the per-point window search
    (orc_match_window) followed by the reference's accept rule (TH_HIGH, same-level ratio test)."""
    from matcher_scene import make_local_map_scene
    sc = make_local_map_scene(oracle, 11, dup_frac=0.0)
    sc["pts"]["n_obs"] = 0                       # nothing ever claims a keypoint -> order independent
    sc["mp"][:] = -1
    n, mp = oracle.search_by_projection_points(th=3.0, nnratio=0.8, **sc)
    pts = sc["pts"]
    sel = np.flatnonzero((pts["in_view"] == 1) & (pts["bad"] == 0))
    r = np.where(pts["view_cos"][sel] > 0.998, np.float32(2.5), np.float32(4.0)) * np.float32(3.0)
    qr = (r * sc["scale_factors"][pts["level"][sel]]).astype(np.float32)
    g = oracle.Grid(sc["kps"])
    m = g.match_window(sc["desc"], pts["desc"][sel], pts["proj_x"][sel], pts["proj_y"][sel], qr, pts["level"][sel] - 1, pts["level"][sel])
    ref = np.full(len(sc["kps"]), -1, np.int32); cnt = 0
    for k, i in enumerate(sel):
        bd, sd = int(m["best_dist"][k]), int(m["second_dist"][k])
        if bd > 100:
            continue
        same = m["best_level"][k] == m["second_level"][k]
        if same and np.float32(bd) > np.float32(0.8) * np.float32(sd):
            continue
        ref[m["best_idx"][k]] = i; cnt += 1
    assert n == cnt and np.array_equal(mp, ref)
    assert n > 100


def test_distinctive_descriptors_oracle_vs_numpy(oracle):
    """ComputeDistinctiveDescriptors: numpy restatement (unpackbits distances, np.sort rows, floor((N-1)/2)-th entry,
    first strict minimum) on ragged sets incl. N = 1, 2 and sets full of duplicates."""
    rng = np.random.default_rng(21)
    sizes = [1, 2, 3, 8, 15, 16, 40, 0, 65, 5]
    descs, off = [], [0]
    for n in sizes:
        base = rng.integers(0, 256, (max(n, 1), 32), dtype=np.uint8)
        d = base[rng.integers(0, max(1, n // 2 + 1), n)] if n else base[:0]   # many duplicates -> median ties
        flip = rng.random((n, 32)) < 0.05
        d = np.where(flip, rng.integers(0, 256, (n, 32), dtype=np.uint8), d).astype(np.uint8)
        descs.append(d); off.append(off[-1] + n)
    desc = np.concatenate(descs)
    bi, bm = oracle.distinctive_descriptors(desc, off)
    for p, n in enumerate(sizes):
        if n == 0:
            assert bi[p] == -1 and bm[p] == -1
            continue
        d = desc[off[p]:off[p + 1]]
        D = np.unpackbits(d[:, None, :] ^ d[None, :, :], axis=2).sum(axis=2)
        med = np.sort(D, axis=1)[:, (n - 1) // 2]
        assert bm[p] == med.min() and bi[p] == int(np.argmin(med)), p


def test_vocab_transform_oracle_vs_numpy(oracle):
    """DBoW2 transform restatement: greedy descent (first child wins ties), TF-IDF accumulation, L1 normalisation and
    L1 score, against a literal numpy / dict re-derivation on a small ragged vocabulary."""
    from dvm_slam_amd import synth
    voc = synth.vocabulary(k=6, L=3, seed=5)
    rng = np.random.default_rng(8)
    feats = voc["desc"][rng.integers(1, voc["n_nodes"], 300)].copy()        # near-duplicates of node descriptors -> ties
    feats[rng.random(feats.shape) < 0.03] ^= 0x10
    for levelsup in (0, 1, 2, 3, 5):
        r = oracle.vocab_transform(voc, feats, levelsup)
        bow, fv = {}, {}
        for i, f in enumerate(feats):
            cur, lvl, nid = 0, 0, (0 if voc["L"] - levelsup <= 0 else -1)
            while voc["child_off"][cur + 1] > voc["child_off"][cur]:
                lvl += 1
                ch = voc["children"][voc["child_off"][cur]:voc["child_off"][cur + 1]]
                d = np.unpackbits(voc["desc"][ch] ^ f, axis=1).sum(axis=1)
                cur = int(ch[int(np.argmin(d))])                                # argmin = first minimum
                if lvl == voc["L"] - levelsup:
                    nid = cur
            assert r["word"][i] == voc["word_id"][cur] and r["node"][i] == nid and r["weight"][i] == voc["weight"][cur]
            if voc["weight"][cur] > 0:
                bow[int(voc["word_id"][cur])] = bow.get(int(voc["word_id"][cur]), 0.0) + voc["weight"][cur]
                fv.setdefault(nid, []).append(i)
        ids = sorted(bow)
        norm = 0.0
        for k in ids:
            norm += abs(bow[k])
        assert list(r["bow_ids"]) == ids
        assert np.array_equal(r["bow_vals"], np.array([bow[k] / norm for k in ids]))
        assert list(r["fv_nodes"]) == sorted(fv)
        for m, nd in enumerate(sorted(fv)):
            assert list(r["fv_feat"][r["fv_off"][m]:r["fv_off"][m + 1]]) == fv[nd]
    a = oracle.vocab_transform(voc, feats[:150], 1); b = oracle.vocab_transform(voc, feats[100:], 1)
    s = oracle.bow_score(a["bow_ids"], a["bow_vals"], b["bow_ids"], b["bow_vals"])
    da = dict(zip(a["bow_ids"], a["bow_vals"])); db = dict(zip(b["bow_ids"], b["bow_vals"]))
    ref = 1.0 - 0.5 * sum(abs(da.get(k, 0.0) - db.get(k, 0.0)) for k in set(da) | set(db))
    assert 0.0 < s < 1.0 and abs(s - ref) < 1e-12
    assert abs(oracle.bow_score(a["bow_ids"], a["bow_vals"], a["bow_ids"], a["bow_vals"]) - 1.0) < 1e-12
