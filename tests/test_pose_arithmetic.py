"""The float pose arithmetic the matchers go through (Sophus so3 / se3 / rxso3 / sim3.hpp + the Eigen 3.4.0 pieces they call),
CPU only: the oracle's restatement (oracle/sophus_oracle.h) against float64 definitions, the product's restatement
(dvm_slam_amd/csrc/pose_f32.h, here through the host library) bit-identical to the oracle's, and the reason the ABI carries
quaternions: the quaternion action and the rotation-matrix product differ in the last ulp often enough to flip gates."""
import numpy as np
import pytest

from dvm_slam_amd import capi, synth
from oracle import pyoracle as po


def _rand_se3(rng, trans=3.0):
    from scipy.spatial.transform import Rotation
    R = Rotation.from_rotvec(rng.normal(0, 1.0, 3)).as_matrix()
    return synth.se3_from_Rt(R, rng.normal(0, trans, 3)), R


def _mat_form_f32(R, t, P):
    """R p + t in float32, a row's sum left to right -- the form the ABI carried before (and Eigen never evaluates)."""
    R = R.astype(np.float32); t = t.astype(np.float32); P = P.astype(np.float32)
    out = np.zeros_like(P)
    for r in range(3):
        acc = (R[r, 0] * P[:, 0] + R[r, 1] * P[:, 1]).astype(np.float32)
        acc = (acc + R[r, 2] * P[:, 2]).astype(np.float32)
        out[:, r] = acc + t[r]
    return out


def test_oracle_pose_functions_vs_float64():
    rng = np.random.default_rng(0)
    for _ in range(20):
        T, R = _rand_se3(rng)
        R64, t64 = synth.Rt_from_se3(T)
        P = rng.normal(0, 5, (200, 3)).astype(np.float32)
        got = po.se3_act(T, P)
        assert np.allclose(got, P.astype(np.float64) @ R64.T + t64, rtol=0, atol=2e-5)
        Ti = po.se3_inverse(T)
        Ri, ti = synth.Rt_from_se3(Ti)
        assert np.allclose(Ri, R64.T, atol=1e-6) and np.allclose(ti, -R64.T @ t64, atol=2e-6)
        assert abs(np.linalg.norm(Ti[:4].astype(np.float64)) - 1) < 2e-7
        Rcw, tcw, Ow = po.pose_matrices(T)
        assert np.allclose(Rcw, R64, atol=1e-6) and np.array_equal(tcw, T[4:]) and np.array_equal(Ow, Ti[4:])
        # a similarity with this rotation
        s = float(rng.uniform(0.3, 3.0))
        S = synth.sim3_from_sRt(s, R64, t64 * s)
        assert np.allclose(po.sim3_act(S, P), s * (P.astype(np.float64) @ R64.T) + s * t64, rtol=0, atol=1e-4)
        Tcw, Ow2 = po.sim3_to_se3(S)          # SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale())
        R2, t2 = synth.Rt_from_se3(Tcw)
        assert np.allclose(R2, R64, atol=1e-6) and np.allclose(t2, t64, atol=1e-5) and np.allclose(Ow2, -R64.T @ t64, atol=1e-5)
        Si = po.sim3_inverse(S)
        back = po.sim3_act(Si, po.sim3_act(S, P))
        assert np.allclose(back, P, atol=1e-4)


def test_quaternion_from_matrix_all_branches():
    """Eigen's Shoemake conversion (sim3_to_se3 goes matrix -> quaternion): trace > 0 and each of the three i-branches."""
    from scipy.spatial.transform import Rotation
    for axis in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]):
        for ang in (0.1, 3.0, np.pi - 1e-3):
            R = Rotation.from_rotvec(np.array(axis, float) / np.linalg.norm(axis) * ang).as_matrix()
            S = synth.sim3_from_sRt(1.7, R, [0.1, 0.2, 0.3])
            Tcw, _ = po.sim3_to_se3(S)
            R2, _ = synth.Rt_from_se3(Tcw)
            assert np.allclose(R2, R, atol=2e-6), (axis, ang)


def test_product_pose_arithmetic_equals_oracle_bitwise():
    rng = np.random.default_rng(1)
    for k in range(200):
        T, R = _rand_se3(rng, trans=10.0 if k % 2 else 0.5)
        P = (rng.normal(0, 8, (64, 3)) * 10.0 ** rng.integers(-2, 2)).astype(np.float32)
        assert np.array_equal(capi.se3_act(T, P), po.se3_act(T, P))
        assert np.array_equal(capi.se3_inverse(T), po.se3_inverse(T))
        for a, b in zip(capi.pose_matrices(T), po.pose_matrices(T)):
            assert np.array_equal(a, b)
        S = synth.sim3_from_sRt(float(rng.uniform(0.2, 5.0)), R, rng.normal(0, 4, 3))
        assert np.array_equal(capi.sim3_act(S, P), po.sim3_act(S, P))
        assert np.array_equal(capi.sim3_inverse(S), po.sim3_inverse(S))
        for a, b in zip(capi.sim3_to_se3(S), po.sim3_to_se3(S)):
            assert np.array_equal(a, b)


def test_quaternion_and_matrix_forms_differ_in_the_last_ulp():
    """Why poses cross the ABI as (q, t): Tcw * p (so3.hpp:356-367) != mRcw * p + mtcw bit for bit on a large share of points."""
    rng = np.random.default_rng(2)
    T, _ = _rand_se3(rng)
    Rcw, tcw, _ = po.pose_matrices(T)
    P = rng.normal(0, 5, (20000, 3)).astype(np.float32)
    q = po.se3_act(T, P)
    m = _mat_form_f32(Rcw, tcw, P)
    differ = np.any(q != m, axis=1)
    assert differ.mean() > 0.3
    assert np.allclose(q, m, rtol=0, atol=1e-5)


def test_shared_logf():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(1e-3, 1e3, 200000), np.float32(1.2) ** np.arange(-10, 10), [1.0, 1e-38, 3e38, 1e-45]]).astype(np.float32)
    got = po.logf(x)
    want = np.log(x.astype(np.float64)).astype(np.float32)
    ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1 and (ulp == 0).mean() > 0.9999
    sub = x[::37]
    assert np.array_equal(capi.logf_shared(sub), po.logf(sub))
    assert po.logf([0.0])[0] == -np.inf and np.isnan(po.logf([-1.0])[0]) and po.logf([np.inf])[0] == np.inf
    assert np.array_equal(capi.logf_shared([0.0, np.inf]), po.logf([0.0, np.inf])) and np.isnan(capi.logf_shared([-2.0])[0])
