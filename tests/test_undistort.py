"""Frame::UndistortKeyPoints / ComputeImageBounds (reference src/Frame.cc:791-848): the oracle's restatement of
cv::undistortPoints against an independent numpy evaluation of the same published algorithm (Python floats are IEEE doubles:
same operations in the same order give the same bits) and against the forward distortion model it inverts."""
import numpy as np

from oracle import pyoracle as po

CAMS = {  # (fx, fy, cx, cy, k1, k2, p1, p2, k3)
    "euroc": (458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0),
    "tum": (517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314),
    "mild": (500.0, 500.0, 320.0, 240.0, 0.05, 0.0, 0.0, 0.0, 0.0),
}


def _undistort_numpy(cam, xy):
    cam = [float(np.float32(c)) for c in cam]
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = cam
    out = np.zeros((len(xy), 2), np.float32)
    for i, (uf, vf) in enumerate(np.asarray(xy, np.float32)):
        u, v = float(uf), float(vf)
        ifx, ify = 1.0 / fx, 1.0 / fy
        x, y = (u - cx) * ifx, (v - cy) * ify
        x0, y0 = x, y
        for _ in range(5):
            r2 = x * x + y * y
            icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
            if icdist < 0:
                x, y = (u - cx) * ifx, (v - cy) * ify
                break
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x, y = (x0 - dx) * icdist, (y0 - dy) * icdist
        out[i] = np.float32(fx * x + cx), np.float32(fy * y + cy)
    return out


def _distort(cam, xy):
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = [float(np.float32(c)) for c in cam]
    x, y = (xy[:, 0].astype(np.float64) - cx) / fx, (xy[:, 1].astype(np.float64) - cy) / fy
    r2 = x * x + y * y
    cd = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([xd * fx + cx, yd * fy + cy], 1)


def test_oracle_matches_numpy_restatement_bit_for_bit():
    rng = np.random.default_rng(11)
    for name, cam in CAMS.items():
        xy = np.concatenate([rng.uniform(0, 752, (3000, 2)), [[0, 0], [752, 0], [0, 480], [752, 480], [cam[2], cam[3]]]]).astype(np.float32)
        got = po.undistort_points(cam, xy)
        want = _undistort_numpy(cam, xy)
        assert np.array_equal(got, want), name


def test_oracle_inverts_the_distortion_model():
    rng = np.random.default_rng(12)
    for name in ("euroc", "mild"):
        cam = CAMS[name]
        ideal = rng.uniform(60, 600, (2000, 2)) * [1.0, 0.7]
        dist = _distort(cam, ideal).astype(np.float32)
        back = po.undistort_points(cam, dist)
        assert np.abs(back - ideal).max() < (0.1 if name == "euroc" else 1e-3), name   # 5 iterations: 0.06 px short at the rim of the EuRoC lens


def test_zero_k1_is_a_copy_and_bounds():
    cam = (500.0, 500.0, 320.0, 240.0, 0.0, 0.3, 0.01, 0.01, 0.0)     # the reference tests ONLY k1 (Frame.cc:792)
    xy = np.random.default_rng(13).uniform(0, 640, (100, 2)).astype(np.float32)
    assert np.array_equal(po.undistort_points(cam, xy), xy)
    assert np.array_equal(po.image_bounds(cam, 640, 480), np.float32([0, 640, 0, 480]))
    b = po.image_bounds(CAMS["euroc"], 752, 480)
    c = po.undistort_points(CAMS["euroc"], np.float32([[0, 0], [752, 0], [0, 480], [752, 480]]))
    assert b[0] == min(c[0, 0], c[2, 0]) and b[1] == max(c[1, 0], c[3, 0]) and b[2] == min(c[0, 1], c[1, 1]) and b[3] == max(c[2, 1], c[3, 1])
    assert b[0] < 0 and b[1] > 752 and b[2] < 0 and b[3] > 480        # barrel distortion: the undistorted image is larger


def test_negative_icdist_keeps_the_normalised_input():
    cam = (100.0, 100.0, 50.0, 50.0, -5.0, 0.0, 0.0, 0.0, 0.0)       # 1 + k1 r2 < 0 for r2 > 0.2
    xy = np.float32([[150.0, 150.0], [52.0, 51.0]])
    got = po.undistort_points(cam, xy)
    assert np.array_equal(got, _undistort_numpy(cam, xy))
    assert np.array_equal(got[0], xy[0])
