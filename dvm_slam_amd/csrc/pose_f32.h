// dvm_slam_amd/csrc/pose_f32.h -- the float pose arithmetic of the reference's matchers, for kernels and host mirrors.
//
// The reference keeps poses as Sophus::SE3f / Sim3f (quaternion + translation) and transforms points with the QUATERNION
// form  p + w*(2 q x p) + q x (2 q x p) (+ t)  (Thirdparty/Sophus/sophus/so3.hpp:356-367, se3.hpp:319-324,
// rxso3.hpp:265-273, sim3.hpp:226-229) -- not with a rotation matrix.  The two differ in the last float ulp, which
// decides image-bound / window / depth gates, so the product carries (q, t) through the C ABI (dvm_se3f / dvm_sim3f)
// and evaluates exactly that form.  Where the reference does use matrices (Frame::isInFrustum with mRcw, the
// fundamental matrix of SearchForTriangulation) the Eigen evaluation order is followed: a three-term sum is
// a0 + (a1 + a2), a four-term quaternion sum is (x + z) + (y + w) (Eigen 3.4.0, SSE2 packets).
// Every function needs -ffp-contract=off on its translation unit (both Makefiles set it).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define DVM_HD __host__ __device__ __forceinline__
#else
#define DVM_HD inline
#endif

namespace dvm_pose {

DVM_HD float sum3(float a, float b, float c) { return a + (b + c); }
DVM_HD float dot3(const float* a, const float* b) { return sum3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
DVM_HD float sqnorm4(const float* q) { return (q[0] * q[0] + q[2] * q[2]) + (q[1] * q[1] + q[3] * q[3]); }

// q = (x, y, z, w).  rot(q, p) = SO3f * p;  with scale = 1 exactly the unit-quaternion form, otherwise RxSO3f * p
DVM_HD void quat_rotate(const float* q, const float* p, float* out) {
  float a0 = q[1] * p[2] - q[2] * p[1], a1 = q[2] * p[0] - q[0] * p[2], a2 = q[0] * p[1] - q[1] * p[0];
  a0 = a0 + a0; a1 = a1 + a1; a2 = a2 + a2;
  const float b0 = q[1] * a2 - q[2] * a1, b1 = q[2] * a0 - q[0] * a2, b2 = q[0] * a1 - q[1] * a0;
  out[0] = (p[0] + q[3] * a0) + b0;
  out[1] = (p[1] + q[3] * a1) + b1;
  out[2] = (p[2] + q[3] * a2) + b2;
}
DVM_HD void se3_apply(const float* q, const float* t, const float* p, float* out) {
  float r[3];
  quat_rotate(q, p, r);
  out[0] = r[0] + t[0]; out[1] = r[1] + t[1]; out[2] = r[2] + t[2];
}
DVM_HD void rxso3_rotate(const float* q, const float* p, float* out) {
  const float s = sqnorm4(q);
  float a0 = q[1] * p[2] - q[2] * p[1], a1 = q[2] * p[0] - q[0] * p[2], a2 = q[0] * p[1] - q[1] * p[0];
  a0 = a0 + a0; a1 = a1 + a1; a2 = a2 + a2;
  const float b0 = q[1] * a2 - q[2] * a1, b1 = q[2] * a0 - q[0] * a2, b2 = q[0] * a1 - q[1] * a0;
  out[0] = s * p[0] + (q[3] * a0 + b0);
  out[1] = s * p[1] + (q[3] * a1 + b1);
  out[2] = s * p[2] + (q[3] * a2 + b2);
}
DVM_HD void sim3_apply(const float* q, const float* t, const float* p, float* out) {
  float r[3];
  rxso3_rotate(q, p, r);
  out[0] = r[0] + t[0]; out[1] = r[1] + t[1]; out[2] = r[2] + t[2];
}
DVM_HD void quat_div(const float* q, float d, float* out) { out[0] = q[0] / d; out[1] = q[1] / d; out[2] = q[2] / d; out[3] = q[3] / d; }

// SE3f::inverse(): normalised conjugate (SO3's quaternion ctor re-normalises), translation = qinv * (t * -1)
DVM_HD void se3_inverse(const float* q, const float* t, float* qi, float* ti) {
  const float c[4] = {-q[0], -q[1], -q[2], q[3]};
  quat_div(c, sqrtf(sqnorm4(c)), qi);
  const float nt[3] = {t[0] * -1.f, t[1] * -1.f, t[2] * -1.f};
  quat_rotate(qi, nt, ti);
}
// SE3f * SE3f
DVM_HD void se3_compose(const float* qa, const float* ta, const float* qb, const float* tb, float* q, float* t) {
  float r[4];
  r[3] = qa[3] * qb[3] - qa[0] * qb[0] - qa[1] * qb[1] - qa[2] * qb[2];
  r[0] = qa[3] * qb[0] + qa[0] * qb[3] + qa[1] * qb[2] - qa[2] * qb[1];
  r[1] = qa[3] * qb[1] + qa[1] * qb[3] + qa[2] * qb[0] - qa[0] * qb[2];
  r[2] = qa[3] * qb[2] + qa[2] * qb[3] + qa[0] * qb[1] - qa[1] * qb[0];
  quat_div(r, sqrtf(sqnorm4(r)), q);
  float rt[3];
  quat_rotate(qa, tb, rt);
  t[0] = ta[0] + rt[0]; t[1] = ta[1] + rt[1]; t[2] = ta[2] + rt[2];
}
// Quaternionf::toRotationMatrix(), row-major
DVM_HD void quat_matrix(const float* q, float* R) {
  const float tx = 2.f * q[0], ty = 2.f * q[1], tz = 2.f * q[2];
  const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1.f - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.f - (txx + tyy);
}
// Quaternionf(Matrix3f) -- what SO3f(R) stores
DVM_HD void matrix_quat(const float* R, float* q) {
  float t = sum3(R[0], R[4], R[8]);
  if (t > 0.f) {
    t = sqrtf(t + 1.f);
    q[3] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    return;
  }
  int i = R[4] > R[0] ? 1 : 0;
  if (R[8] > R[4 * i]) i = 2;
  const int j = (i + 1) % 3, k = (j + 1) % 3;
  t = sqrtf(R[4 * i] - R[4 * j] - R[4 * k] + 1.f);
  q[i] = 0.5f * t;
  t = 0.5f / t;
  q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
  q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
  q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
}
// Tcw = SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale()), Ow = Tcw.inverse().translation()
// (ORBmatcher.cc:403-404, 505-506, 1245-1246)
DVM_HD void sim3_decompose(const float* qs, const float* ts, float* q, float* t, float* Ow) {
  const float z = sqnorm4(qs);
  float n[4] = {qs[0], qs[1], qs[2], qs[3]};
  if (z > 0.f) quat_div(qs, sqrtf(z), n);
  float R[9];
  quat_matrix(n, R);
  matrix_quat(R, q);
  t[0] = ts[0] / z; t[1] = ts[1] / z; t[2] = ts[2] / z;
  float qi[4];
  se3_inverse(q, t, qi, Ow);
}
// Sim3f::inverse()
DVM_HD void sim3_inverse(const float* q, const float* t, float* qi, float* ti) {
  const float n2 = sqnorm4(q);
  const float c[4] = {-q[0], -q[1], -q[2], q[3]};
  quat_div(c, n2, qi);
  const float nt[3] = {t[0] * -1.f, t[1] * -1.f, t[2] * -1.f};
  rxso3_rotate(qi, nt, ti);
}
// 3x3 (row-major) products / inverse in Eigen's evaluation order
DVM_HD void mat3_mul(const float* A, const float* B, float* C) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) C[3 * r + c] = sum3(A[3 * r] * B[c], A[3 * r + 1] * B[3 + c], A[3 * r + 2] * B[6 + c]);
}
DVM_HD void mat3_apply(const float* A, const float* x, float* y) {
  y[0] = sum3(A[0] * x[0], A[1] * x[1], A[2] * x[2]);
  y[1] = sum3(A[3] * x[0], A[4] * x[1], A[5] * x[2]);
  y[2] = sum3(A[6] * x[0], A[7] * x[1], A[8] * x[2]);
}
DVM_HD float mat3_cofactor(const float* M, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return M[3 * i1 + j1] * M[3 * i2 + j2] - M[3 * i1 + j2] * M[3 * i2 + j1];
}
DVM_HD void mat3_inverse(const float* M, float* out) {
  const float c0 = mat3_cofactor(M, 0, 0), c1 = mat3_cofactor(M, 1, 0), c2 = mat3_cofactor(M, 2, 0);
  const float invdet = 1.f / sum3(c0 * M[0], c1 * M[3], c2 * M[6]);
  out[0] = c0 * invdet; out[1] = c1 * invdet; out[2] = c2 * invdet;
  out[3] = mat3_cofactor(M, 0, 1) * invdet; out[4] = mat3_cofactor(M, 1, 1) * invdet; out[5] = mat3_cofactor(M, 2, 1) * invdet;
  out[6] = mat3_cofactor(M, 0, 2) * invdet; out[7] = mat3_cofactor(M, 1, 2) * invdet; out[8] = mat3_cofactor(M, 2, 2) * invdet;
}

// Shared logf (MapPoint::PredictScale, MapPoint.cc:573-587): double evaluation of fdlibm's log polynomial on
// x = 2^k (1 + f), sqrt(1/2) < 1 + f <= sqrt(2), rounded to float once -- the same operation sequence on host and device.
DVM_HD float logf_shared(float xf) {
  if (!(xf > 0.f)) return xf == 0.f ? -INFINITY : NAN;
  if (xf > 3.4028234663852886e38f) return xf;
  const double x = (double)xf;
  uint64_t bits;
  memcpy(&bits, &x, 8);
  int k = (int)((bits >> 52) & 0x7FF) - 1023;
  bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
  double m;
  memcpy(&m, &bits, 8);
  if (m > 1.41421356237309514547) { m = m * 0.5; k = k + 1; }
  const double f = m - 1.0, s = f / (2.0 + f), z = s * s, w = z * z;
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 = z * (6.666666666666735130e-01 +
                         w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)k;
  return (float)(dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f));
}
// MapPoint::PredictScale: ceil(log(mfMaxDistance / dist) / logScaleFactor) clamped to [0, nLevels)
DVM_HD int predict_scale(float max_dist, float dist, float log_scale_factor, int n_levels) {
  const float ratio = max_dist / dist;
  int nScale = (int)ceilf(logf_shared(ratio) / log_scale_factor);
  return nScale < 0 ? 0 : (nScale >= n_levels ? n_levels - 1 : nScale);
}

}  // namespace dvm_pose
