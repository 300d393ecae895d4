"""GPU parity: host C++ mirror dvm_host::ORBmatcher::SearchByProjection(CurrentFrame, LastFrame) (device window search +
sequential claim / rotation-histogram epilogue) vs the sequential oracle restatement.  Exact: nmatches and every
mvpMapPoints entry."""
import numpy as np
import pytest

from matcher_scene import make_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,th,ori", [(0, 15.0, True), (1, 7.0, True), (2, 15.0, False), (3, 30.0, True)])
def test_search_by_projection_frames(capi, oracle, seed, th, ori):
    sc = make_scene(oracle, seed)
    n_o, mp_o = oracle.search_by_projection_frames(th=th, check_ori=ori, **sc)
    n_g, mp_g, req = capi.search_by_projection_frames(th=th, check_ori=ori, **sc)
    assert n_g == n_o
    assert np.array_equal(mp_g, mp_o)
    assert n_o > 300


def test_search_by_projection_frames_crowded(capi, oracle):
    """Several times more projected map points than keypoints and wide windows: the four best candidates the device returns per query
    are often all taken by earlier queries of the call, so the host's fallback search (the full window again, claims applied) runs -- and
    the result is still the sequential oracle's."""
    total = 0
    for seed, n_cur, th in ((7, 160, 60.0), (8, 220, 40.0), (9, 300, 80.0)):
        sc = make_scene(oracle, seed, n_last=1000, n_cur=n_cur)
        n_o, mp_o = oracle.search_by_projection_frames(th=th, check_ori=True, **sc)
        n_g, mp_g, req = capi.search_by_projection_frames(th=th, check_ori=True, **sc)
        assert n_g == n_o and np.array_equal(mp_g, mp_o), (seed, n_cur)
        total += req
    assert total > 20, "the scenes must exercise the fallback search"


def test_search_by_projection_frames_degenerate(capi, oracle):
    sc = make_scene(oracle, 5)
    # nothing to project
    sc2 = dict(sc); sc2["mp_l"] = np.full_like(sc["mp_l"], -1)
    n_g, mp_g, _ = capi.search_by_projection_frames(th=15.0, **sc2)
    assert n_g == 0 and np.array_equal(mp_g, sc["mp_c"])
    # every point behind the camera
    sc3 = dict(sc); mps = sc["mps"].copy(); mps["pos"][:, 2] = -np.abs(mps["pos"][:, 2]); sc3["mps"] = mps
    n_o, mp_o = oracle.search_by_projection_frames(th=15.0, **sc3)
    n_g, mp_g, _ = capi.search_by_projection_frames(th=15.0, **sc3)
    assert n_g == n_o == 0 and np.array_equal(mp_g, mp_o)


def test_search_by_projection_real_frames(capi, oracle, frames):
    """Real extractor output of two consecutive synthetic frames; map points back-projected at random depth."""
    orc = oracle.OrbOracle()
    _, k0, d0, _ = orc.extract(frames[0])
    _, k1, d1, _ = orc.extract(frames[1])
    rng = np.random.default_rng(9)
    K = np.array([500.0, 500.0, 320.0, 240.0], np.float32)
    z = rng.uniform(3, 9, len(k0)).astype(np.float32)
    mps = np.zeros(len(k0), oracle.MAP_POINT_DTYPE)
    mps["pos"][:, 0] = (k0["x"] - K[2]) / K[0] * z
    mps["pos"][:, 1] = (k0["y"] - K[3]) / K[1] * z
    mps["pos"][:, 2] = z
    mps["desc"] = d0
    mps["n_obs"] = rng.integers(0, 4, len(k0))
    args = dict(kps_c=k1, desc_c=d1, mp_c=np.full(len(k1), -1, np.int32),
                Tcw=np.array([0, 0, 0, 1, 0, 0, 0], np.float32), K=K, bounds=np.array([0, 640, 0, 480], np.float32),
                scale_factors=orc.tables()["scale"], kps_l=k0, mp_l=np.arange(len(k0), dtype=np.int32), outlier_l=None, mps=mps)
    n_o, mp_o = oracle.search_by_projection_frames(th=15.0, **args)
    n_g, mp_g, _ = capi.search_by_projection_frames(th=15.0, **args)
    assert n_g == n_o and np.array_equal(mp_g, mp_o)
    assert n_o > 200


@pytest.mark.parametrize("seed,th,ratio,far", [(0, 1.0, 0.8, False), (1, 3.0, 0.8, False), (2, 5.0, 0.9, True), (3, 1.0, 0.6, True)])
def test_search_by_projection_points(capi, oracle, seed, th, ratio, far):
    """Whole SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints) (TrackLocalMap / relocalisation form)."""
    from matcher_scene import make_local_map_scene
    sc = make_local_map_scene(oracle, seed)
    n_o, mp_o = oracle.search_by_projection_points(th=th, nnratio=ratio, far_points=far, th_far=9.0, **sc)
    n_g, mp_g, req = capi.search_by_projection_points(th=th, nnratio=ratio, far_points=far, th_far=9.0, **sc)
    assert n_g == n_o
    assert np.array_equal(mp_g, mp_o)
    assert n_o > 100
    if seed == 1:
        assert req > 0, "scene must exercise the claimed-keypoint re-query path"
