"""GPU parity: Hamming / grid-window matching kernels (through the C ABI) vs the CPU oracle.  Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("na,nb", [(1, 1), (37, 1001), (1000, 1000), (3, 2048)])
def test_hamming_matrix(capi, oracle, na, nb):
    rng = np.random.default_rng(na * 7 + nb)
    A = rng.integers(0, 256, (na, 32), dtype=np.uint8)
    B = rng.integers(0, 256, (nb, 32), dtype=np.uint8)
    B[: min(na, nb)] = A[: min(na, nb)]  # exact zeros on the diagonal
    D = capi.hamming_matrix(A, B)
    assert np.array_equal(D, oracle.hamming_matrix(A, B))
    assert D[0, 0] == 0
    # identity: popcount(a^b) via numpy unpackbits
    ref = np.unpackbits(A[:, None, :] ^ B[None, : min(nb, 64), :], axis=2).sum(axis=2)
    assert np.array_equal(D[:, : min(nb, 64)], ref)


def test_hamming_empty(capi):
    D = capi.hamming_matrix(np.zeros((0, 32), np.uint8), np.zeros((5, 32), np.uint8))
    assert D.shape == (0, 5)


def _check_matches(g, o):
    for k in ("best_idx", "best_dist", "second_dist", "best_level", "second_level"):
        assert np.array_equal(g[k].astype(np.int64), o[k].astype(np.int64)), k


def test_match_window_vs_oracle(capi, oracle, frames):
    orc = oracle.OrbOracle()
    _, k0, d0, _ = orc.extract(frames[0])
    _, k1, d1, _ = orc.extract(frames[1])
    grid_g = capi.FrameGrid(capacity=2048)
    grid_g.build(k1, d1)
    grid_o = oracle.Grid(k1)
    scale = orc.tables()["scale"]
    # TrackWithMotionModel-style queries (ORBmatcher.cc:1596-1611): th=15, octaves [o-1, o+1]
    qx, qy = k0["x"], k0["y"]
    qr = (np.float32(15) * scale[k0["octave"]]).astype(np.float32)
    qmin, qmax = k0["octave"] - 1, k0["octave"] + 1
    _check_matches(grid_g.match_window(d0, qx, qy, qr, qmin, qmax), grid_o.match_window(d1, d0, qx, qy, qr, qmin, qmax))
    # SearchByProjection(F, mappoints)-style: levels [l-1, l], radius 2.5/4 * scale
    qr2 = (np.float32(4.0) * scale[k0["octave"]]).astype(np.float32)
    _check_matches(grid_g.match_window(d0, qx, qy, qr2, k0["octave"] - 1, k0["octave"]),
                   grid_o.match_window(d1, d0, qx, qy, qr2, k0["octave"] - 1, k0["octave"]))
    # random windows incl. out-of-image centres, huge / tiny radii, unbounded levels, a skip mask
    rng = np.random.default_rng(3)
    nq = 700
    qd = d0[rng.integers(0, len(d0), nq)]
    qx = rng.uniform(-80, 720, nq).astype(np.float32)
    qy = rng.uniform(-80, 560, nq).astype(np.float32)
    qr = rng.choice([0.5, 3, 15, 50, 100, 1000], nq).astype(np.float32)
    qmin = rng.integers(-1, 8, nq).astype(np.int32)
    qmax = rng.integers(-1, 8, nq).astype(np.int32)
    skip = (rng.random(len(k1)) < 0.3).astype(np.uint8)
    _check_matches(grid_g.match_window(qd, qx, qy, qr, qmin, qmax, skip=skip),
                   grid_o.match_window(d1, qd, qx, qy, qr, qmin, qmax, skip=skip))
    # the runner-up's index (dvm_match_window_top2): what the scan returns when the best candidate is masked as well
    g2, second = grid_g.match_window(qd, qx, qy, qr, qmin, qmax, skip=skip, top2=True)
    _check_matches(g2, grid_o.match_window(d1, qd, qx, qy, qr, qmin, qmax, skip=skip))
    assert np.all((second >= 0) == (g2["second_dist"] < 256))
    checked = 0
    for q in np.flatnonzero(g2["best_idx"] >= 0)[:200]:
        sk2 = skip.copy(); sk2[g2["best_idx"][q]] = 1
        o = grid_o.match_window(d1, qd[q:q + 1], qx[q:q + 1], qy[q:q + 1], qr[q:q + 1], qmin[q:q + 1], qmax[q:q + 1], skip=sk2)
        assert int(o["best_idx"][0]) == int(second[q]) and (second[q] < 0 or int(o["best_dist"][0]) == int(g2["second_dist"][q]))
        checked += second[q] >= 0
    assert checked > 50
    # the ranked list (dvm_match_window_ranked): entry c is what the scan returns when entries 0..c-1 are masked as well
    ridx, rdist = grid_g.match_window_ranked(qd, qx, qy, qr, qmin, qmax, skip=skip)
    assert np.array_equal(ridx[:, 0], g2["best_idx"]) and np.array_equal(rdist[:, 0][ridx[:, 0] >= 0], g2["best_dist"][ridx[:, 0] >= 0])
    assert np.array_equal(ridx[:, 1], second)
    deep = 0
    for q in np.flatnonzero(ridx[:, 1] >= 0)[:150]:
        sk2 = skip.copy()
        for c in range(4):
            o = grid_o.match_window(d1, qd[q:q + 1], qx[q:q + 1], qy[q:q + 1], qr[q:q + 1], qmin[q:q + 1], qmax[q:q + 1], skip=sk2)
            assert int(o["best_idx"][0]) == int(ridx[q, c]), (q, c)
            if ridx[q, c] < 0:
                assert np.all(ridx[q, c:] < 0) and np.all(rdist[q, c:] == 256)
                break
            assert int(o["best_dist"][0]) == int(rdist[q, c])
            sk2[ridx[q, c]] = 1
            deep += c == 3
    assert deep > 20
    grid_g.close()


def test_match_window_ties(capi, oracle):
    """Many identical descriptors: the winner must be the FIRST candidate in GetFeaturesInArea order."""
    rng = np.random.default_rng(11)
    n = 1500
    kps = np.zeros(n, oracle.KP_DTYPE)
    kps["x"] = rng.uniform(0, 640, n).astype(np.float32)
    kps["y"] = rng.uniform(0, 480, n).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, n)
    base = rng.integers(0, 256, (4, 32), dtype=np.uint8)
    desc = base[rng.integers(0, 4, n)]            # only 4 distinct descriptors -> ties everywhere
    grid_g = capi.FrameGrid(capacity=2048)
    grid_g.build(kps, desc)
    grid_o = oracle.Grid(kps)
    nq = 500
    qd = base[rng.integers(0, 4, nq)]
    qx = rng.uniform(0, 640, nq).astype(np.float32)
    qy = rng.uniform(0, 480, nq).astype(np.float32)
    qr = rng.choice([20, 60, 200], nq).astype(np.float32)
    neg = np.full(nq, -1, np.int32)
    _check_matches(grid_g.match_window(qd, qx, qy, qr, neg, neg), grid_o.match_window(desc, qd, qx, qy, qr, neg, neg))
    grid_g.close()


def test_grid_edge_cases(capi, oracle):
    # empty frame, single keypoint, keypoints outside the grid (dropped by PosInGrid), ragged n
    for n in (0, 1, 63, 65, 1025):
        rng = np.random.default_rng(n)
        kps = np.zeros(n, oracle.KP_DTYPE)
        kps["x"] = rng.uniform(-5, 645, n).astype(np.float32)  # some round to column 64 -> not indexed
        kps["y"] = rng.uniform(-5, 485, n).astype(np.float32)
        kps["octave"] = rng.integers(0, 8, n)
        desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        g = capi.FrameGrid(capacity=2048)
        g.build(kps, desc)
        o = oracle.Grid(kps)
        nq = 64
        qd = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        qx = rng.uniform(0, 640, nq).astype(np.float32)
        qy = rng.uniform(0, 480, nq).astype(np.float32)
        qr = np.full(nq, 700, np.float32)
        neg = np.full(nq, -1, np.int32)
        _check_matches(g.match_window(qd, qx, qy, qr, neg, neg), o.match_window(desc, qd, qx, qy, qr, neg, neg))
        g.close()


def test_match_frames_batch(capi, oracle, frames):
    """extract(batch) -> grid build (batch) -> frame-to-frame search, all device resident."""
    import torch
    B = len(frames)
    e = capi.OrbExtractor(max_batch=B)
    e.extract_batch_host(frames)
    k_ptr, d_ptr, n_ptr, cap = e.result_device(0)
    grid = capi.FrameGrid(capacity=2048, slots=B)
    st = e.stream()
    grid.build_batch_device(0, B, k_ptr, cap, d_ptr, cap * 32, n_ptr, (0.0, 640.0, 0.0, 480.0), stream=st)
    out = torch.zeros((B, cap, 4), dtype=torch.int32, device="cuda")
    nq = torch.zeros(B, dtype=torch.int32, device="cuda")
    grid.match_frames_batch(0, B, k_ptr, cap, d_ptr, cap * 32, n_ptr, None, cap, 15.0, e.scale_factors_device(), 8,
                            out.data_ptr(), cap, nq.data_ptr(), stream=st)
    e.sync()
    torch.cuda.synchronize()
    res = out.cpu().numpy().view(capi.MATCH_DTYPE).reshape(B, cap)
    nqh = nq.cpu().numpy()
    orc = oracle.OrbOracle()
    scale = orc.tables()["scale"]
    ext = [orc.extract(f) for f in frames]
    assert nqh[0] == 0
    for i in range(1, B):
        _, kq, dq, _ = ext[i - 1]
        _, kt, dt, _ = ext[i]
        assert nqh[i] == len(kq)
        go = oracle.Grid(kt)
        ref = go.match_window(dt, dq, kq["x"], kq["y"], (np.float32(15) * scale[kq["octave"]]).astype(np.float32),
                              kq["octave"] - 1, kq["octave"] + 1)
        _check_matches(res[i][: len(kq)], ref)
        # the synthetic stream moves <= 8 px/frame: most keypoints must find a good match
        assert (ref["best_dist"] <= 50).mean() > 0.5
    grid.close()
    e.close()


def test_match_lists_bow_style(capi, oracle):
    """SearchByBoW inner loop: per query an explicit candidate list (same vocabulary node), incl. empty lists,
    masked (-1) entries, duplicates and heavy ties; reference semantics = sequential strict-'<' scan."""
    rng = np.random.default_rng(21)
    nt, nq = 900, 400
    base = rng.integers(0, 256, (6, 32), dtype=np.uint8)
    td = np.where(rng.random((nt, 1)) < 0.5, base[rng.integers(0, 6, nt)], rng.integers(0, 256, (nt, 32), dtype=np.uint8)).astype(np.uint8)
    qd = np.where(rng.random((nq, 1)) < 0.5, base[rng.integers(0, 6, nq)], rng.integers(0, 256, (nq, 32), dtype=np.uint8)).astype(np.uint8)
    lens = rng.choice([0, 1, 2, 5, 30, 200], nq)
    off = np.zeros(nq + 1, np.int32)
    off[1:] = np.cumsum(lens)
    cand = rng.integers(0, nt, off[-1]).astype(np.int32)
    cand[rng.random(len(cand)) < 0.1] = -1
    got = capi.match_lists(td, qd, off, cand)
    for q in range(nq):
        best, second, bidx = 256, 256, -1
        for idx in cand[off[q]:off[q + 1]]:
            if idx < 0:
                continue
            d = int(np.unpackbits(td[idx] ^ qd[q]).sum())
            if d < best:
                second, best, bidx = best, d, int(idx)
            elif d < second:
                second = d
        assert (got["best_idx"][q], got["best_dist"][q], got["second_dist"][q]) == (bidx, best, second), q


def test_is_in_frustum(capi, oracle):
    """Frame::isInFrustum (matrix form mRcw * P + mtcw, Frame.cc:585): every output field bit-identical to the oracle --
    both sides evaluate Eigen's a0 + (a1 + a2) sums and the shared logf of MapPoint::PredictScale.  mRcw / mOw are derived
    from the SE3f pose by each side's own UpdatePoseMatrices."""
    from dvm_slam_amd import synth
    rng = np.random.default_rng(8)
    n = 20000
    ang = 0.3
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    tcw = np.array([0.2, -0.1, 0.5], np.float32)
    Tcw = synth.se3_from_Rt(Rcw, tcw)
    K = (149.0, 149.0, 320.0, 240.0)
    P = rng.uniform(-15, 15, (n, 3)).astype(np.float32)
    normal = rng.normal(size=(n, 3)).astype(np.float32)
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    maxd = rng.uniform(5, 30, n).astype(np.float32)
    mind = (maxd / np.float32(1.2) ** 7).astype(np.float32)
    for a, b in zip(oracle.pose_matrices(Tcw), capi.pose_matrices(Tcw)):
        assert np.array_equal(a, b)
    Fo = oracle.make_frustum_frame(Tcw, K)
    Fg = oracle.make_frustum_frame(Tcw, K, cls=capi.FrustumFrame, matrices=capi.pose_matrices)
    ref = oracle.is_in_frustum(Fo, P, normal, mind, maxd, 0.5)
    got = capi.is_in_frustum(Fg, P, normal, mind, maxd, 0.5)
    assert ref["in_view"].sum() > 200
    for f in ("in_view", "level", "proj_x", "proj_y", "proj_xr", "depth", "view_cos"):
        assert np.array_equal(got[f], ref[f]), f
    # points behind the camera are never in view and keep proj = -1
    Pc = (Rcw @ P.T).T + tcw
    behind = Pc[:, 2] < -0.1
    assert (got["in_view"][behind] == 0).all() and (got["proj_x"][behind] == -1).all()


def test_distinctive_descriptors(capi, oracle):
    """dvm_distinctive_descriptors (MapPoint::ComputeDistinctiveDescriptors, batched) vs the oracle: ragged observation
    counts 0..300, duplicate-heavy sets (median ties -> first index wins), one set above the 512 limit (-2)."""
    rng = np.random.default_rng(33)
    sizes = [1, 2, 3, 7, 0, 64, 65, 130, 300, 12, 600] + list(rng.integers(1, 40, 200))
    descs, off = [], [0]
    for n in sizes:
        n = int(n)
        base = rng.integers(0, 256, (max(n, 1), 32), dtype=np.uint8)
        d = base[rng.integers(0, max(1, n // 3 + 1), n)] if n else base[:0]
        flip = rng.random((n, 32)) < 0.04
        d = np.where(flip, rng.integers(0, 256, (n, 32), dtype=np.uint8), d).astype(np.uint8)
        descs.append(d); off.append(off[-1] + n)
    desc = np.concatenate(descs)
    bi_g, bm_g = capi.distinctive_descriptors(desc, off)
    bi_o, bm_o = oracle.distinctive_descriptors(desc, off)
    big = np.array(sizes) > 512
    assert np.all(bi_g[big] == -2) and np.all(bm_g[big] == -2)
    assert np.array_equal(bi_g[~big], bi_o[~big]) and np.array_equal(bm_g[~big], bm_o[~big])
    assert capi.distinctive_descriptors(np.zeros((0, 32), np.uint8), [0])[0].shape == (0,)


@pytest.mark.parametrize("k,L", [(10, 3), (6, 4), (10, 1)])
def test_vocab_transform(capi, oracle, k, L):
    """dvm_vocab_transform (DBoW2 tree descent on the device) and the host mirror's BowVector / FeatureVector / score
    vs the oracle restatement of TemplatedVocabulary::transform.  Word / node ids and weights identical, BoW values
    bit-identical doubles."""
    from dvm_slam_amd import synth
    voc = synth.vocabulary(k=k, L=L, seed=k * 10 + L)
    rng = np.random.default_rng(k + L)
    feats = voc["desc"][rng.integers(1, voc["n_nodes"], 1500)].copy()
    feats[rng.random(feats.shape) < 0.05] ^= 0x81
    feats[:50] = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    v = capi.Vocabulary(voc)
    for levelsup in (0, 2, 4, 7):
        wg, ng, wtg = v.transform(feats, levelsup)
        r = oracle.vocab_transform(voc, feats, levelsup)
        assert np.array_equal(wg, r["word"]) and np.array_equal(ng, r["node"]) and np.array_equal(wtg, r["weight"])
        h = capi.vocab_transform_host(voc, feats, levelsup)
        for key in ("bow_ids", "bow_vals", "fv_nodes", "fv_off", "fv_feat"):
            assert np.array_equal(h[key], r[key]), key
    assert v.transform(np.zeros((0, 32), np.uint8), 4)[0].shape == (0,)
    v.close()
    a = capi.vocab_transform_host(voc, feats[:800], 4); b = capi.vocab_transform_host(voc, feats[600:], 4)
    assert capi.bow_score_host(a["bow_ids"], a["bow_vals"], b["bow_ids"], b["bow_vals"]) == \
        oracle.bow_score(a["bow_ids"], a["bow_vals"], b["bow_ids"], b["bow_vals"])


def _write_dbow2_text(voc, path, k):
    """The vocabulary in DBoW2's text form (saveToTextFile, TemplatedVocabulary.h:1290-1315): "k L scoring weighting", then one line per
    node in id order: parent, isLeaf, 32 descriptor bytes, weight.  synth.vocabulary numbers its nodes breadth-first, children after
    parents, as a file read by loadFromTextFile does."""
    parent = np.zeros(voc["n_nodes"], np.int64)
    for n in range(voc["n_nodes"]):
        parent[voc["children"][voc["child_off"][n]:voc["child_off"][n + 1]]] = n
    with open(path, "w") as f:
        f.write(f"{k} {voc['L']}  0 0\n")
        for n in range(1, voc["n_nodes"]):
            leaf = int(voc["word_id"][n] >= 0)
            f.write(f"{parent[n]} {leaf} " + " ".join(str(int(b)) for b in voc["desc"][n]) + f" {float(voc['weight'][n])!r}\n")
        f.write("\n")                                  # ORBvoc.txt ends in a newline; the loader must not turn it into a node


def test_vocabulary_from_text_file(capi, oracle, tmp_path):
    """ORBVocabulary::loadFromTextFile: the tree read from DBoW2's text format gives the transform of the tree built from the arrays
    (word ids in leaf order, children in node order), a damaged file is refused."""
    from dvm_slam_amd import synth
    voc = synth.vocabulary(k=9, L=3, seed=77)
    path = tmp_path / "voc.txt"
    _write_dbow2_text(voc, path, 9)
    hv = capi.HostVocabulary(str(path))
    assert (hv.k, hv.L, hv.nodes, hv.words) == (9, 3, voc["n_nodes"], int((voc["word_id"] >= 0).sum()))
    rng = np.random.default_rng(8)
    feats = voc["desc"][rng.integers(1, voc["n_nodes"], 900)].copy()
    feats[rng.random(feats.shape) < 0.04] ^= 0x42
    for levelsup in (0, 2, 4):
        h = hv.transform(feats, levelsup)
        r = oracle.vocab_transform(voc, feats, levelsup)
        for key in ("bow_ids", "bow_vals", "fv_nodes", "fv_off", "fv_feat"):
            assert np.array_equal(h[key], r[key]), key
    hv.close()
    text = path.read_text().splitlines()
    bad = tmp_path / "bad.txt"
    bad.write_text("\n".join([text[0]] + [text[1].replace(text[1].split()[0], "5", 1)] + text[2:]))      # a parent that does not exist yet
    with pytest.raises(RuntimeError):
        capi.HostVocabulary(str(bad))
    bad.write_text("30 3 0 0\n" + "\n".join(text[1:]))                                                   # k out of range
    with pytest.raises(RuntimeError):
        capi.HostVocabulary(str(bad))


def test_frame_grid_reports_truncated_device_counts(capi):
    """a device-side count above the handle's capacity is clamped by k_frame_build -- and counted, so the caller of
    the device-resident path learns about it (dvm_frame_overflows) instead of silently matching against a truncated frame."""
    import torch
    rng = np.random.default_rng(3)
    cap = 512
    g = capi.FrameGrid(capacity=cap, slots=2)
    k = np.zeros(2 * 600, capi.KP_DTYPE)
    k["x"] = rng.uniform(0, 640, len(k)); k["y"] = rng.uniform(0, 480, len(k))
    d_k = torch.from_numpy(k.view(np.uint8).copy()).cuda()
    d_d = torch.from_numpy(rng.integers(0, 256, (len(k), 32), dtype=np.uint8)).cuda()
    d_n = torch.tensor([500, 400], dtype=torch.int32).cuda()
    g.build_batch_device(0, 2, d_k.data_ptr(), 600, d_d.data_ptr(), 600 * 32, d_n.data_ptr(), (0.0, 640.0, 0.0, 480.0))
    torch.cuda.synchronize()
    assert g.overflows() == 0
    d_n = torch.tensor([500, 600], dtype=torch.int32).cuda()          # second frame: 600 keypoints into 512 slots
    g.build_batch_device(0, 2, d_k.data_ptr(), 600, d_d.data_ptr(), 600 * 32, d_n.data_ptr(), (0.0, 640.0, 0.0, 480.0))
    torch.cuda.synchronize()
    with pytest.raises(capi.DvmError) as e:
        g.overflows()
    assert "truncated" in str(e.value)
    g.close()
