"""oracle/ba_oracle.cpp::orc_triangulate_matches (LocalMapping::CreateNewMapPoints' per-match geometry, LocalMapping.cc:598-741 +
GeometricTools::Triangulate) against an independent float64 numpy statement with np.linalg.svd: same decision wherever the deciding
comparison is not within float noise of its threshold, same point within the float error of a 4x4 null vector."""
import numpy as np
import pytest

import tri_scene
from oracle import pyoracle as po


def _call(S, **kw):
    return po.triangulate_matches(S["K1"], S["K2"], S["T1w"], S["T2w"], S["Ow1"], S["Ow2"], S["kps1"], S["kps2"], S["pairs"],
                                  S["sigma2_1"], S["sigma2_2"], S["sf1"], S["sf2"], S["ratio_factor"], **kw)


@pytest.mark.parametrize("seed", range(6))
def test_matches_numpy_statement(seed):
    S = tri_scene.scene(seed=seed, n=500, baseline=0.3 + 0.2 * seed)
    X, st = _call(S)
    Xr, sr, margin = tri_scene.numpy_reference(S)
    clear = margin > 2e-3
    assert clear.mean() > 0.9
    assert np.array_equal(st[clear], sr[clear])
    tri = clear & (sr != 1)
    assert tri.sum() > 100
    rel = np.linalg.norm(X[tri] - Xr[tri], axis=1) / np.linalg.norm(Xr[tri], axis=1)
    # float inputs, well-conditioned (parallax above the gate): the null vector of A is good to ~1e-4 relative
    assert rel.max() < 2e-3 and np.median(rel) < 2e-5
    assert set(np.unique(st)) >= {0, 1, 5} and (st == 0).sum() > 100
    assert np.all(X[st == 1] == 0)


def test_far_points_and_empty():
    S = tri_scene.scene(seed=11, n=400)
    X, st = _call(S, far_points=True, th_far=8.0)
    Xr, sr, margin = tri_scene.numpy_reference(S, far_points=True, th_far=8.0)
    clear = margin > 2e-3
    assert np.array_equal(st[clear], sr[clear]) and (st == 8).sum() > 10
    S0 = dict(S, pairs=np.zeros((0, 2), np.int32))
    X0, st0 = _call(S0)
    assert X0.shape == (0, 3) and st0.shape == (0,)


def test_decisions_follow_the_documented_order():
    """A match behind camera 1 reports 3 even if it would also fail later tests; the parallax gate comes first."""
    S = tri_scene.scene(seed=3, n=300, wrong_frac=0.5)
    X, st = _call(S)
    T1, T2 = S["T1w"].astype(np.float64), S["T2w"].astype(np.float64)
    for m in np.flatnonzero(st == 3):
        assert T1[2, :3] @ X[m].astype(np.float64) + T1[2, 3] <= 1e-6
    for m in np.flatnonzero(st == 4):
        assert T1[2, :3] @ X[m].astype(np.float64) + T1[2, 3] > -1e-6 and T2[2, :3] @ X[m].astype(np.float64) + T2[2, 3] <= 1e-6
