"""GPU parity: HIP ORB extractor (through the C ABI) vs the CPU oracle, stage by stage and end to end.

Bit-exact bar for every integer stage (pyramid pixels, FAST candidates and scores, octree selection,
blur pixels, descriptors, output order); keypoint floats (pt, angle) must be bitwise equal too since
both sides perform the same IEEE single-rounding sequence (-ffp-contract=off).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same_kps(a, b):
    assert len(a) == len(b)
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(a[f], b[f]), f


@pytest.fixture(scope="module")
def ext(capi):
    e = capi.OrbExtractor(max_batch=4)
    yield e
    e.close()


def test_tables_match_oracle(capi, oracle, ext):
    t = ext.tables()
    o = oracle.OrbOracle().tables()
    for k in ("scale", "inv_scale", "sigma2", "inv_sigma2", "nfeat"):
        assert np.array_equal(t[k], o[k]), k


def test_stagewise_parity_640x480(capi, oracle, ext, frames):
    orc = oracle.OrbOracle()
    n_o, k_o, d_o, m_o = orc.extract(frames[0])
    n_g, k_g, d_g, m_g = ext.extract(frames[0])
    for l in range(8):
        assert np.array_equal(ext.debug_level(0, l), orc.level(l)), f"pyramid level {l}"
        assert np.array_equal(ext.debug_level(0, l, bordered=True), orc.level(l, bordered=True)), f"border {l}"
        xg, yg, sg = ext.debug_candidates(0, l)
        xo, yo, so = orc.candidates(l)
        assert np.array_equal(xg, xo) and np.array_equal(yg, yo) and np.array_equal(sg, so), f"FAST level {l}"
        lk_g, lk_o = ext.debug_level_keypoints(0, l), orc.level_keypoints(l)
        _same_kps(lk_g, lk_o)
        # blurred image is only consumed within 18 px of a keypoint... but the whole level must match
        bo = orc.blurred(l)
        if bo is not None:
            assert np.array_equal(ext.debug_blurred(0, l), bo), f"blur level {l}"
    assert (n_g, m_g) == (n_o, m_o)
    _same_kps(k_g, k_o)
    assert np.array_equal(d_g, d_o)


@pytest.mark.parametrize("shape", [(480, 640), (376, 1241), (120, 160), (97, 131), (480, 752)])
def test_end_to_end_sizes(capi, oracle, shape):
    from dvm_slam_amd import synth
    img = synth.small_image(7 + shape[0], *shape)
    e = capi.OrbExtractor(max_batch=1)
    orc = oracle.OrbOracle()
    for lap in ((0, 1000), (0, 0), (100, 300)):
        n_o, k_o, d_o, m_o = orc.extract(img, lap=lap)
        n_g, k_g, d_g, m_g = e.extract(img, lap=lap)
        assert (n_g, m_g) == (n_o, m_o), (shape, lap)
        _same_kps(k_g, k_o)
        assert np.array_equal(d_g, d_o)
    for l in range(8):   # REFLECT_101 frame at every width alignment (vectorised border kernel)
        assert np.array_equal(e.debug_level(0, l, bordered=True), orc.level(l, bordered=True)), (shape, l)
    e.close()


def test_batch_equals_single(capi, oracle, frames):
    e = capi.OrbExtractor(max_batch=4)
    e.extract_batch_host(frames)
    orc = oracle.OrbOracle()
    for f in range(len(frames)):
        n_g, k_g, d_g, m_g = e.download(f)
        n_o, k_o, d_o, m_o = orc.extract(frames[f])
        assert (n_g, m_g) == (n_o, m_o)
        _same_kps(k_g, k_o)
        assert np.array_equal(d_g, d_o)
    e.close()


def test_chunked_pipeline_equals_single(capi, oracle, frames):
    """DVM_CHUNKS=n cuts a batch into a two-lane chunk pipeline (orb_pipeline.cpp, off by default): every frame
    must still be exact, twice in a row (buffer reuse across calls), including a ragged last chunk."""
    import os
    os.environ["DVM_CHUNKS"] = "3"
    try:
        _chunked(capi, oracle, frames)
    finally:
        del os.environ["DVM_CHUNKS"]


@pytest.mark.parametrize("groups", ["1,3", "0,2", "2,8"])
def test_level_group_pipelining_equals_single(capi, oracle, frames, groups):
    """DVM_GROUPS=a,b launches FAST per level group and the octree of a group on a second stream under the next
    group's FAST cells (orb_pipeline.cpp, off by default): results must not change."""
    import os
    os.environ["DVM_GROUPS"] = groups
    try:
        _chunked(capi, oracle, frames, sizes=(64,))
    finally:
        del os.environ["DVM_GROUPS"]


def _chunked(capi, oracle, frames, sizes=(64, 97, 256)):
    orc = oracle.OrbOracle()
    ref = [orc.extract(f) for f in frames]
    for B in sizes:
        batch = np.stack([frames[i % len(frames)] for i in range(B)])
        e = capi.OrbExtractor(max_batch=B)
        for rep in range(2):
            e.extract_batch_host(batch)
            for f in (0, 1, B // 2 - 1, B // 2, B // 2 + 1, B - 2, B - 1):
                n_g, k_g, d_g, m_g = e.download(f)
                n_o, k_o, d_o, m_o = ref[f % len(frames)]
                assert (n_g, m_g) == (n_o, m_o), (B, f)
                _same_kps(k_g, k_o)
                assert np.array_equal(d_g, d_o)
        e.close()


def test_staged_ingest_pipeline(capi, oracle, frames):
    """dvm_orb_staging / dvm_orb_extract_staged: frames written into the handle's pinned buffer, H2D on the copy stream; the
    buffer is refilled with DIFFERENT frames while the previous batch is still computing (no sync in between), two handles
    alternate as in bench.py's PCIe-inclusive leg.  Every batch must still be exact."""
    from dvm_slam_amd import synth
    orc = oracle.OrbOracle()
    B = 8
    sets = [np.stack([synth.small_image(50 * k + j, 480, 640) for j in range(B)]) for k in range(4)]
    ref = [[orc.extract(img) for img in st] for st in sets]
    exts = [capi.OrbExtractor(max_batch=B), capi.OrbExtractor(max_batch=B)]
    pending = [None, None]

    def check(k, e):
        for f in (0, 3, B - 1):
            n_g, k_g, d_g, m_g = e.download(f)
            n_o, k_o, d_o, m_o = ref[k][f]
            assert (n_g, m_g) == (n_o, m_o), (k, f)
            _same_kps(k_g, k_o)
            assert np.array_equal(d_g, d_o)

    for rep in range(2):
        for k in range(4):
            e = exts[k & 1]
            if pending[k & 1] is not None:
                check(pending[k & 1], e)
            buf = e.staging(B, 480, 640)
            buf[:] = sets[k]
            e.extract_staged(B, 480, 640)
            buf2 = e.staging(B, 480, 640)      # returns as soon as the copy has left the pinned buffer ...
            buf2[:] = 0                        # ... so scribbling over it must not disturb the batch in flight
            pending[k & 1] = k
    for i, e in enumerate(exts):
        check(pending[i], e)
        e.close()


def test_edge_cases(capi, oracle):
    e = capi.OrbExtractor(max_batch=1)
    orc = oracle.OrbOracle()
    # empty image -> -1 like the reference (ORBextractor.cc:879-880)
    assert e.extract(np.zeros((0, 0), np.uint8))[0] == -1
    # flat image: no corner anywhere, zero keypoints, no crash
    flat = np.full((480, 640), 127, np.uint8)
    n_g, k_g, d_g, m_g = e.extract(flat)
    n_o, _, _, m_o = orc.extract(flat)
    assert (n_g, m_g) == (n_o, m_o) == (0, 0)
    # salt noise: maximum candidate density, exercises every cell's capacity and the iniTh path
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    n_g, k_g, d_g, m_g = e.extract(noise)
    n_o, k_o, d_o, m_o = orc.extract(noise)
    assert (n_g, m_g) == (n_o, m_o)
    _same_kps(k_g, k_o)
    assert np.array_equal(d_g, d_o)
    # low-contrast texture: cells fall through to minThFAST
    low = (120 + (noise.astype(np.int32) % 23)).astype(np.uint8)
    n_g, k_g, d_g, m_g = e.extract(low)
    n_o, k_o, d_o, m_o = orc.extract(low)
    assert (n_g, m_g) == (n_o, m_o)
    _same_kps(k_g, k_o)
    assert np.array_equal(d_g, d_o)
    # non-contiguous rows (stride > cols)
    big = np.zeros((480, 700), np.uint8)
    big[:, :640] = noise
    n_o, k_o, d_o, m_o = orc.extract(noise)
    n_s, k_s, d_s, _ = e.extract(big[:, :640])
    _same_kps(k_s, k_o)
    assert np.array_equal(d_s, d_o)
    e.close()


def test_other_parameters(capi, oracle):
    from dvm_slam_amd import synth
    img = synth.frame_stream(1, start=17)[0]
    for (nf, sf, nl, it, mt) in [(1500, 1.2, 8, 20, 7), (2000, 1.2, 8, 20, 7), (500, 1.5, 4, 15, 5), (5000, 1.2, 8, 20, 7)]:
        e = capi.OrbExtractor(nf, sf, nl, it, mt, max_batch=1)
        orc = oracle.OrbOracle(nf, sf, nl, it, mt)
        n_o, k_o, d_o, m_o = orc.extract(img, cap=3 * nf)
        n_g, k_g, d_g, m_g = e.extract(img)
        assert (n_g, m_g) == (n_o, m_o), (nf, sf, nl)
        _same_kps(k_g, k_o)
        assert np.array_equal(d_g, d_o)
        e.close()


def test_soak_regressions(capi, oracle):
    """Cases found by tools/soak_parity.py: (a) small quotas on wide images -- the first octree sweep splits all
    nIni = round(W / H) root nodes before the node count is compared with the quota, so a level may keep up to 4 * nIni
    keypoints (> quota + 2); (b) level quotas above 1 528 (up to 2 680: the whole 160 KB of LDS for one workgroup) still run
    on the device, larger ones are refused (no silent CPU path);
    (c) portrait sizes with nIni = 0 (the reference divides by zero) are rejected."""
    from dvm_slam_amd import synth
    for (h, w, nf, sf, nl, it, mt) in [(326, 895, 100, 1.3, 8, 31, 19), (200, 664, 100, 1.2, 8, 36, 3), (126, 1192, 100, 1.3, 4, 20, 20),
                                       (410, 766, 3000, 1.3, 2, 33, 24), (480, 640, 3000, 1.3, 2, 19, 17), (600, 800, 12000, 1.2, 8, 12, 5),
                                       (546, 670, 2680, 1.2, 1, 33, 15)]:
        img = synth.small_image(h + w, h, w)
        e = capi.OrbExtractor(nf, sf, nl, it, mt, max_batch=1)
        orc = oracle.OrbOracle(nf, sf, nl, it, mt)
        n_o, k_o, d_o, m_o = orc.extract(img, cap=4 * nf + 256)
        n_g, k_g, d_g, m_g = e.extract(img)
        assert (n_g, m_g) == (n_o, m_o), (h, w, nf)
        _same_kps(k_g, k_o)
        assert np.array_equal(d_g, d_o)
        e.close()
    e = capi.OrbExtractor(max_batch=1)
    with pytest.raises(capi.DvmError, match="too narrow"):
        e.extract(synth.small_image(3, 900, 300))
    e.close()
    for nf, nl in ((20000, 8), (3000, 1), (2681, 1)):
        e = capi.OrbExtractor(nf, 1.2, nl, 20, 7, max_batch=1)
        with pytest.raises(capi.DvmError, match="octree capacity"):
            e.extract(synth.small_image(4, 480, 640))
        e.close()


def test_reconfigure_same_handle(capi, oracle):
    """One handle, changing image sizes and batch sizes between calls (device buffers are re-planned per size)."""
    from dvm_slam_amd import synth
    e = capi.OrbExtractor(max_batch=4)
    orc = oracle.OrbOracle()
    for i, ((h, w), b) in enumerate([((480, 640), 1), ((240, 320), 3), ((376, 1241), 4), ((480, 640), 2), ((240, 320), 1)]):
        imgs = np.stack([synth.small_image(100 * i + j, h, w) for j in range(b)])
        if b == 1:
            res = [e.extract(imgs[0])]
        else:
            e.extract_batch_host(imgs)
            res = [e.download(f) for f in range(b)]
        for f in range(b):
            n_o, k_o, d_o, m_o = orc.extract(imgs[f])
            n_g, k_g, d_g, m_g = res[f]
            assert (n_g, m_g) == (n_o, m_o), (h, w, b, f)
            _same_kps(k_g, k_o)
            assert np.array_equal(d_g, d_o)
    e.close()
