// oracle/sophus_oracle.h -- f32 restatement of the Sophus / Eigen pose arithmetic the reference's matchers go through.
//
// TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).
//
// Sophus is vendored in the reference (Thirdparty/Sophus/sophus/*.hpp, paths below are relative to
// /root/reference/src/slam_system/orb_slam3/); Eigen is NOT (find_package(Eigen3 3.1.0 REQUIRED), CMakeLists.txt) -- the
// pieces of it that decide a rounding are restated from Eigen 3.4.0 (the version of Ubuntu 22.04 / ROS 2 Humble, which the
// reference targets) as compiled for x86-64 (SSE2 packets of 4 floats, no FMA contraction):
//   * a sum of THREE terms (Vector3f::dot / norm / squaredNorm / trace, a 3x3 * 3x1 or 3x3 * 3x3 coefficient) is
//     a0 + (a1 + a2): Core/Redux.h redux_novec_unroller splits [0,3) into [0,1) and [1,3);
//   * a sum of FOUR terms over a Quaternionf's coefficients (x,y,z,w) is (x + z) + (y + w): one Packet4f, SSE predux =
//     movehl + add, then shuffle + add_ss (arch/SSE/PacketMath.h);
//   * cross(a,b) = (a1*b2 - a2*b1, a2*b0 - a0*b2, a0*b1 - a1*b0)        (Geometry/OrthoMethods.h);
//   * Quaternion::toRotationMatrix, Quaternion = Matrix3 (Shoemake), Quaternion::inverse, MatrixBase::normalize
//     (Geometry/Quaternion.h, Core/Dot.h), 3x3 inverse by cofactors (LU/InverseImpl.h).
// PARITY UNPINNED: no Eigen / Sophus build exists in this image; a reference built with -march=native on an FMA host
// contracts some of these products (GCC -ffp-contract=fast) -- that variant is host-dependent and not modelled.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace sophus_oracle {

struct Quat { float x, y, z, w; };            // Eigen::Quaternionf coeffs() order
struct SE3 { Quat q; float t[3]; };           // Sophus::SE3f: unit quaternion + translation
struct Sim3 { Quat q; float t[3]; };          // Sophus::Sim3f: RxSO3f quaternion (scale = |q|^2) + translation

inline float sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }
inline float dot3(const float* a, const float* b) { return sum3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
inline float norm3(const float* a) { return std::sqrt(dot3(a, a)); }
inline float quat_sqnorm(const Quat& q) { return (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w); }
inline void cross3(const float* a, const float* b, float* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// SO3Base::operator*(point), Thirdparty/Sophus/sophus/so3.hpp:356-367: uv = q.vec x p; uv += uv;
// return p + q.w * uv + q.vec x uv   (left to right: (p + w*uv) + cross)
inline void so3_act(const Quat& q, const float* p, float* out) {
  const float v[3] = {q.x, q.y, q.z};
  float uv[3], c[3];
  cross3(v, p, uv);
  for (int i = 0; i < 3; i++) uv[i] = uv[i] + uv[i];
  cross3(v, uv, c);
  for (int i = 0; i < 3; i++) out[i] = (p[i] + q.w * uv[i]) + c[i];
}
// SE3Base::operator*(point), se3.hpp:319-324: so3() * p + translation()
inline void se3_act(const SE3& T, const float* p, float* out) {
  float r[3];
  so3_act(T.q, p, r);
  for (int i = 0; i < 3; i++) out[i] = r[i] + T.t[i];
}
// SO3Base::normalize, so3.hpp:297-303: length = coeffs.norm(); coeffs /= length
inline Quat quat_normalized_sophus(const Quat& q) {
  const float len = std::sqrt(quat_sqnorm(q));
  return Quat{q.x / len, q.y / len, q.z / len, q.w / len};
}
// SO3Base::inverse, so3.hpp:229-231: SO3(unit_quaternion().conjugate()) -- the explicit ctor (:481-487) re-normalises
inline Quat so3_inverse(const Quat& q) { return quat_normalized_sophus(Quat{-q.x, -q.y, -q.z, q.w}); }
// SE3Base::inverse, se3.hpp:208-211: invR = so3().inverse(); SE3(invR, invR * (translation() * -1))
inline SE3 se3_inverse(const SE3& T) {
  SE3 r;
  r.q = so3_inverse(T.q);
  const float nt[3] = {T.t[0] * -1.f, T.t[1] * -1.f, T.t[2] * -1.f};
  so3_act(r.q, nt, r.t);
  return r;
}
// SO3Base::operator*(SO3), so3.hpp:324-339 (explicit quaternion product, then the normalising ctor)
inline Quat so3_mul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return quat_normalized_sophus(r);
}
// SE3Base::operator*(SE3), se3.hpp:303-308
inline SE3 se3_mul(const SE3& A, const SE3& B) {
  SE3 r;
  r.q = so3_mul(A.q, B.q);
  float rt[3];
  so3_act(A.q, B.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = A.t[i] + rt[i];
  return r;
}
// Eigen::QuaternionBase::toRotationMatrix (Geometry/Quaternion.h), row-major out
inline void quat_to_matrix(const Quat& q, float* R) {
  const float tx = 2.f * q.x, ty = 2.f * q.y, tz = 2.f * q.z;
  const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.f - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.f - (txx + tyy);
}
// Eigen quaternionbase_assign_impl<Matrix3f,3,3>::run (Shoemake), used by SO3(Matrix3 const&), so3.hpp:469
inline Quat quat_from_matrix(const float* R) {
  Quat q;
  float c[4];
  float t = sum3(R[0], R[4], R[8]);   // mat.trace()
  if (t > 0.f) {
    t = std::sqrt(t + 1.f);
    q.w = 0.5f * t;
    t = 0.5f / t;
    q.x = (R[7] - R[5]) * t;
    q.y = (R[2] - R[6]) * t;
    q.z = (R[3] - R[1]) * t;
    return q;
  }
  int i = 0;
  if (R[4] > R[0]) i = 1;
  if (R[8] > R[4 * i]) i = 2;
  const int j = (i + 1) % 3, k = (j + 1) % 3;
  t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.f);
  c[i] = 0.5f * t;
  t = 0.5f / t;
  c[3] = (R[3 * k + j] - R[3 * j + k]) * t;
  c[j] = (R[3 * j + i] + R[3 * i + j]) * t;
  c[k] = (R[3 * k + i] + R[3 * i + k]) * t;
  q.x = c[0]; q.y = c[1]; q.z = c[2]; q.w = c[3];
  return q;
}
// RxSO3Base::rotationMatrix, rxso3.hpp:341-345: norm_quad = quaternion(); norm_quad.normalize() (Eigen: z = squaredNorm;
// if z > 0: coeffs /= sqrt(z)); toRotationMatrix
inline void rxso3_rotation_matrix(const Quat& q, float* R) {
  Quat n = q;
  const float z = quat_sqnorm(q);
  if (z > 0.f) { const float s = std::sqrt(z); n = Quat{q.x / s, q.y / s, q.z / s, q.w / s}; }
  quat_to_matrix(n, R);
}
// The decomposition every Sim3 consumer of ORBmatcher does (ORBmatcher.cc:403-404,505-506,1245-1246):
//   Tcw = SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale()); Ow = Tcw.inverse().translation()
inline void sim3_to_se3(const Sim3& S, SE3& Tcw, float* Ow) {
  float R[9];
  rxso3_rotation_matrix(S.q, R);
  Tcw.q = quat_from_matrix(R);
  const float s = quat_sqnorm(S.q);   // RxSO3Base::scale, rxso3.hpp:350
  for (int i = 0; i < 3; i++) Tcw.t[i] = S.t[i] / s;
  const SE3 Twc = se3_inverse(Tcw);
  for (int i = 0; i < 3; i++) Ow[i] = Twc.t[i];
}
// RxSO3Base::operator*(point), rxso3.hpp:265-273: scale * p + (q.w * tvcp + q.vec x tvcp)
inline void rxso3_act(const Quat& q, const float* p, float* out) {
  const float scale = quat_sqnorm(q);
  const float v[3] = {q.x, q.y, q.z};
  float tv[3], c[3];
  cross3(v, p, tv);
  for (int i = 0; i < 3; i++) tv[i] = tv[i] + tv[i];
  cross3(v, tv, c);
  for (int i = 0; i < 3; i++) out[i] = scale * p[i] + (q.w * tv[i] + c[i]);
}
// Sim3Base::operator*(point), sim3.hpp:226-229
inline void sim3_act(const Sim3& S, const float* p, float* out) {
  float r[3];
  rxso3_act(S.q, p, r);
  for (int i = 0; i < 3; i++) out[i] = r[i] + S.t[i];
}
// Sim3Base::inverse, sim3.hpp:129-132 with RxSO3Base::inverse rxso3.hpp:156-158 and Eigen Quaternion::inverse
// (conjugate().coeffs() / squaredNorm())
inline Sim3 sim3_inverse(const Sim3& S) {
  Sim3 r;
  const float n2 = quat_sqnorm(S.q);
  r.q = Quat{-S.q.x / n2, -S.q.y / n2, -S.q.z / n2, S.q.w / n2};
  const float nt[3] = {S.t[0] * -1.f, S.t[1] * -1.f, S.t[2] * -1.f};
  rxso3_act(r.q, nt, r.t);
  return r;
}
// 3x3 products as Eigen evaluates them (lazy coefficient-based product, row-major storage here)
inline void mat3_mul(const float* A, const float* B, float* C) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) C[3 * r + c] = sum3(A[3 * r] * B[c], A[3 * r + 1] * B[3 + c], A[3 * r + 2] * B[6 + c]);
}
inline void mat3_vec(const float* A, const float* x, float* y) {
  for (int r = 0; r < 3; r++) y[r] = sum3(A[3 * r] * x[0], A[3 * r + 1] * x[1], A[3 * r + 2] * x[2]);
}
// Eigen compute_inverse<Matrix3f> (LU/InverseImpl.h, compute_inverse_size3_helper): cofactor(i,j) =
// m(i1,j1)*m(i2,j2) - m(i1,j2)*m(i2,j1), i1=(i+1)%3 ...; det = sum3 over column 0; result(j,i) = cofactor(i,j) * invdet
inline void mat3_inverse(const float* M, float* out) {
  auto cof = [&](int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return M[3 * i1 + j1] * M[3 * i2 + j2] - M[3 * i1 + j2] * M[3 * i2 + j1];
  };
  const float c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
  const float det = sum3(c00 * M[0], c10 * M[3], c20 * M[6]);
  const float invdet = 1.f / det;
  out[3 * 1 + 0] = cof(0, 1) * invdet;
  out[3 * 1 + 1] = cof(1, 1) * invdet;
  out[3 * 2 + 0] = cof(0, 2) * invdet;
  out[3 * 1 + 2] = cof(2, 1) * invdet;
  out[3 * 2 + 1] = cof(1, 2) * invdet;
  out[3 * 2 + 2] = cof(2, 2) * invdet;
  out[0] = c00 * invdet; out[1] = c10 * invdet; out[2] = c20 * invdet;
}

// logf as both sides of the parity test compute it (MapPoint::PredictScale, MapPoint.cc:573-587 calls log(float) = glibc
// logf; HIP has no bit-identical libm, so -- like the descriptor's sincos -- a shared spec replaces it): the argument is
// promoted to double, x = 2^k (1+f) with sqrt(1/2) < 1+f <= sqrt(2), fdlibm e_log.c's polynomial in Horner form with
// separate mul / add, the result rounded to float.  Agrees with glibc logf except on rare 1-ulp cases.
inline float logf_spec(float xf) {
  if (!(xf > 0.f)) return xf == 0.f ? -INFINITY : NAN;
  if (std::isinf(xf)) return xf;
  const double x = (double)xf;
  uint64_t bits;
  std::memcpy(&bits, &x, 8);
  int k = (int)((bits >> 52) & 0x7FF) - 1023;
  bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
  double m;
  std::memcpy(&m, &bits, 8);
  if (m > 1.41421356237309514547) { m = m * 0.5; k = k + 1; }
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 = z * (6.666666666666735130e-01 +
                         w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  const double r = dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
  return (float)r;
}

}  // namespace sophus_oracle
