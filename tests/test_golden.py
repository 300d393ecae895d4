"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU part: the oracle still reproduces them.  GPU part (-m gpu): the HIP path reproduces them through
the C ABI -- bit-exact for keypoints / descriptors / matches, 1e-6 for BA states."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ORB_CASES = ["orb_160x120.npz", "orb_320x240.npz"]


def _params(z):
    p = z["params"]
    return int(p[0]), float(p[1]), int(p[2]), int(p[3]), int(p[4])


def _check_orb(z, n, k, d, mono):
    assert (n, mono) == (int(z["n"]), int(z["mono"]))
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(k[f], z["kp_" + f]), f
    assert np.array_equal(d, z["desc"])


@pytest.mark.parametrize("case", ORB_CASES)
def test_oracle_reproduces_orb_golden(oracle, case):
    z = np.load(os.path.join(G, case))
    o = oracle.OrbOracle(*_params(z))
    _check_orb(z, *o.extract(z["image"]))
    assert [o.level_dims(l) for l in range(_params(z)[2])] == [tuple(r) for r in z["level_dims"]]


def test_oracle_reproduces_match_golden(oracle):
    z = np.load(os.path.join(G, "match_320x240.npz"))
    g = oracle.Grid(z["k1"], 0.0, 320.0, 0.0, 240.0)
    m = g.match_window(z["d1"], z["d0"], z["k0"]["x"], z["k0"]["y"], z["qr"], z["k0"]["octave"] - 1, z["k0"]["octave"] + 1)
    for f in ("best_idx", "best_dist", "second_dist", "best_level", "second_level"):
        assert np.array_equal(m[f], z[f]), f


def test_oracle_reproduces_ba_golden(oracle):
    z = np.load(os.path.join(G, "ba_10kf_200pt.npz"))
    poses, pts, st, chi = oracle.ba_optimize(z["poses0"], z["fixed"], z["points0"], z["edges"], z["intrinsics"], float(z["delta"]), int(z["iters"]))
    assert st["trials"] == list(z["trials"])
    assert np.allclose(poses, z["poses"], atol=1e-9) and np.allclose(pts, z["points"], atol=1e-9)
    assert np.allclose(st["chi2"], z["chi2"], rtol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ORB_CASES)
def test_hip_reproduces_orb_golden(capi, case):
    z = np.load(os.path.join(G, case))
    e = capi.OrbExtractor(*_params(z), max_batch=1)
    _check_orb(z, *e.extract(z["image"]))
    e.close()


@pytest.mark.gpu
def test_hip_reproduces_match_golden(capi):
    z = np.load(os.path.join(G, "match_320x240.npz"))
    g = capi.FrameGrid(capacity=1024)
    g.build(z["k1"], z["d1"], bounds=(0.0, 320.0, 0.0, 240.0))
    m = g.match_window(z["d0"], z["k0"]["x"], z["k0"]["y"], z["qr"], z["k0"]["octave"] - 1, z["k0"]["octave"] + 1)
    for f in ("best_idx", "best_dist", "second_dist", "best_level", "second_level"):
        assert np.array_equal(m[f].astype(np.int64), z[f].astype(np.int64)), f
    g.close()


@pytest.mark.gpu
def test_hip_reproduces_ba_golden(capi):
    z = np.load(os.path.join(G, "ba_10kf_200pt.npz"))
    ba = capi.BundleAdjuster()
    ba.set_problem(z["poses0"], z["fixed"], z["points0"], z["edges"], z["intrinsics"], float(z["delta"]))
    st = ba.optimize(int(z["iters"]))
    p, pts = ba.result()
    assert st["trials"] == list(z["trials"])
    assert np.abs(p - z["poses"]).max() < 1e-6 and np.abs(pts - z["points"]).max() < 1e-6
    assert np.allclose(st["chi2"], z["chi2"], rtol=1e-9)
    ba.close()
