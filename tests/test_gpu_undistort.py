"""dvm_undistort_keypoints / dvm_image_bounds (Frame::UndistortKeyPoints, ComputeImageBounds; Frame.cc:791-848) against the
oracle: bit-exact (double arithmetic in the same order, one rounding to float), all other keypoint fields untouched, and the
undistorted keypoints + bounds feeding the frame grid give the oracle's grid."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_undistort import CAMS  # noqa: E402


def _kps(capi, rng, n, w=752, h=480):
    k = np.zeros(n, capi.KP_DTYPE)
    k["x"] = rng.uniform(0, w, n).astype(np.float32); k["y"] = rng.uniform(0, h, n).astype(np.float32)
    k["size"] = 31.0; k["angle"] = rng.uniform(0, 360, n).astype(np.float32); k["response"] = rng.integers(7, 255, n)
    k["octave"] = rng.integers(0, 8, n); k["class_id"] = -1
    return k


@pytest.mark.parametrize("name", list(CAMS))
def test_undistort_keypoints_bit_exact(capi, oracle, name):
    rng = np.random.default_rng(21)
    cam = CAMS[name]
    for n in (1, 63, 1000, 5000):
        k = _kps(capi, rng, n)
        k["x"][:1] = 0.0; k["y"][:1] = 0.0
        got = capi.undistort_keypoints(cam, k)
        want = oracle.undistort_points(cam, np.stack([k["x"], k["y"]], 1))
        assert np.array_equal(got["x"], want[:, 0]) and np.array_equal(got["y"], want[:, 1])
        for f in ("size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(got[f], k[f])
    assert np.array_equal(capi.image_bounds(cam, 752, 480), oracle.image_bounds(cam, 752, 480))
    assert len(capi.undistort_keypoints(cam, np.zeros(0, capi.KP_DTYPE))) == 0


def test_zero_k1_copies_and_degenerate_inputs(capi, oracle):
    rng = np.random.default_rng(22)
    cam = (500.0, 500.0, 320.0, 240.0, 0.0, 0.3, 0.01, 0.01, 0.0)
    k = _kps(capi, rng, 300, 640, 480)
    assert np.array_equal(capi.undistort_keypoints(cam, k), k)
    assert np.array_equal(capi.image_bounds(cam, 640, 480), np.float32([0, 640, 0, 480]))
    neg = (100.0, 100.0, 50.0, 50.0, -5.0, 0.0, 0.0, 0.0, 0.0)      # icdist < 0 branch
    k2 = _kps(capi, rng, 200, 200, 200)
    got = capi.undistort_keypoints(neg, k2)
    want = oracle.undistort_points(neg, np.stack([k2["x"], k2["y"]], 1))
    assert np.array_equal(got["x"], want[:, 0]) and np.array_equal(got["y"], want[:, 1])
    with pytest.raises(RuntimeError):
        capi.undistort_keypoints((0.0, 500.0, 1, 1, 0.1, 0, 0, 0, 0), k)


def test_device_pointers_in_place_and_grid(capi, oracle):
    """The Frame constructor's order (Frame.cc:411-467): extract -> UndistortKeyPoints -> ComputeImageBounds -> grid."""
    import torch
    rng = np.random.default_rng(23)
    cam = CAMS["euroc"]
    k = _kps(capi, rng, 1200)
    d = torch.from_numpy(k.view(np.uint8).copy()).cuda()
    capi.undistort_keypoints(cam, None, d_in=d.data_ptr(), d_out=d.data_ptr(), n=len(k), stream=None)
    torch.cuda.synchronize()
    got = d.cpu().numpy().view(capi.KP_DTYPE)
    want = oracle.undistort_points(cam, np.stack([k["x"], k["y"]], 1))
    assert np.array_equal(got["x"], want[:, 0]) and np.array_equal(got["y"], want[:, 1])
    # window search on the undistorted grid with the undistorted bounds (negative mnMinX / mnMinY: the grid origin moves)
    b = tuple(float(v) for v in capi.image_bounds(cam, 752, 480))
    assert b[0] < 0 and b[2] < 0
    desc = rng.integers(0, 256, (len(k), 32), dtype=np.uint8)
    g = capi.FrameGrid(capacity=2048)
    g.build(got, desc, bounds=b)
    go = oracle.Grid(got, *b)
    nq = 400
    qi = rng.integers(0, len(k), nq)
    qd = desc[qi] ^ rng.integers(0, 2, (nq, 32), dtype=np.uint8)
    qx = got["x"][qi] + rng.normal(0, 3, nq).astype(np.float32); qy = got["y"][qi] + rng.normal(0, 3, nq).astype(np.float32)
    qr = np.full(nq, 12.0, np.float32)
    qmin = np.full(nq, -1, np.int32); qmax = np.full(nq, -1, np.int32)
    mg = g.match_window(qd, qx, qy, qr, qmin, qmax)
    mo = go.match_window(desc, qd, qx, qy, qr, qmin, qmax)
    g.close()
    for f in ("best_idx", "best_dist", "second_dist", "best_level", "second_level"):
        assert np.array_equal(mg[f].astype(np.int32), mo[f]), f
    assert (mg["best_idx"] >= 0).sum() > 300
