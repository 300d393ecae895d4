"""dvm_triangulate_matches (k_triangulate_matches) against the oracle: LocalMapping::CreateNewMapPoints' per-match geometry
(LocalMapping.cc:598-741, GeometricTools.cc:48-67).  Kernel and oracle run the same float / double operation sequence, so every
status and every coordinate must be IDENTICAL; against the reference's Eigen::JacobiSVD the parity is a tolerance one
(tests/test_oracle_triangulation.py pins the oracle to np.linalg.svd)."""
import numpy as np
import pytest

import tri_scene

pytestmark = pytest.mark.gpu


def _args(S):
    return (S["K1"], S["K2"], S["T1w"], S["T2w"], S["Ow1"], S["Ow2"], S["kps1"], S["kps2"], S["pairs"], S["sigma2_1"], S["sigma2_2"], S["sf1"],
            S["sf2"], S["ratio_factor"])


@pytest.mark.parametrize("seed,n,baseline", [(0, 700, 0.5), (1, 3000, 0.2), (2, 129, 1.5), (3, 1, 0.4), (4, 5000, 0.05)])
def test_bit_exact_against_oracle(capi, oracle, seed, n, baseline):
    S = tri_scene.scene(seed=seed, n=n, baseline=baseline)
    Xo, so = oracle.triangulate_matches(*_args(S))
    Xg, sg = capi.triangulate_matches(*_args(S))
    assert np.array_equal(sg, so)
    assert np.array_equal(Xg.view(np.uint32), Xo.view(np.uint32))
    if n >= 700 and baseline >= 0.5:
        assert (so == 0).sum() > n // 4
    if baseline <= 0.05:
        assert (so == 1).sum() > n // 5                                   # a short baseline: many rays are too parallel


def test_far_points_inertial_gate_and_mixed_tables(capi, oracle):
    S = tri_scene.scene(seed=9, n=1500)
    S["sf2"] = (1.25 ** np.arange(8)).astype(np.float32); S["sigma2_2"] = (S["sf2"] ** 2).astype(np.float32)   # a peer with another pyramid
    S["K2"] = np.array([430.0, 431.0, 350.0, 240.0], np.float32)
    for kw in (dict(far_points=True, th_far=7.5), dict(cos_parallax_max=0.9996), dict(far_points=True, th_far=1e9)):
        Xo, so = oracle.triangulate_matches(*_args(S), **kw)
        Xg, sg = capi.triangulate_matches(*_args(S), **kw)
        assert np.array_equal(sg, so) and np.array_equal(Xg.view(np.uint32), Xo.view(np.uint32))
    assert (so == 8).sum() == 0


def test_threshold_neighbourhoods(capi, oracle):
    """Keypoints nudged so that the reprojection error sits within a few ulp of 5.991 sigma2, depths near zero, parallax at the gate:
    kernel and oracle still decide alike (they share the arithmetic), whichever way."""
    S = tri_scene.scene(seed=21, n=2000, noise_px=0.0, wrong_frac=0.0, far_frac=0.3)
    rng = np.random.default_rng(5)
    sig = S["sigma2_1"][S["kps1"]["octave"]]
    r = np.sqrt(5.991 * sig.astype(np.float64))
    # the baseline is along x, so a displacement in y cannot be absorbed by the depth: it splits over the two images, and 2 r puts
    # both reprojection errors next to 5.991 sigma2
    S["kps1"]["y"] += (2 * r).astype(np.float32) * rng.choice([0.0, 0.98, 0.999, 1.0, 1.001, 1.02], len(r)).astype(np.float32)
    Xo, so = oracle.triangulate_matches(*_args(S))
    Xg, sg = capi.triangulate_matches(*_args(S))
    assert np.array_equal(sg, so) and np.array_equal(Xg.view(np.uint32), Xo.view(np.uint32))
    assert {0, 1, 5, 6} <= set(np.unique(so).tolist())


def test_empty_and_invalid(capi, oracle):
    from dvm_slam_amd import capi as c
    S = tri_scene.scene(seed=2, n=20)
    X, st = capi.triangulate_matches(*_args(dict(S, pairs=np.zeros((0, 2), np.int32))))
    assert X.shape == (0, 3) and st.shape == (0,)
    bad = S["pairs"].copy(); bad[3, 1] = 20                               # index past the keypoint array
    with pytest.raises(c.DvmError):
        capi.triangulate_matches(*_args(dict(S, pairs=bad)))
    with pytest.raises(c.DvmError):
        capi.triangulate_matches(*_args(dict(S, K1=np.array([0, 1, 2, 3], np.float32))))
    k = S["kps1"].copy(); k["octave"][S["pairs"][0, 0]] = 9              # octave outside the tables: reported per match, nothing read
    X, st = capi.triangulate_matches(*_args(dict(S, kps1=k)))
    assert st[0] == -1 and np.all(X[0] == 0)
