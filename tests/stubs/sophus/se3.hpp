// test mock: Sophus::SE3 as the shims and the mock ORB_SLAM3 classes use it (see tests/stubs/README.md).  Own code; the float
// operation order follows what oracle/sophus_oracle.h documents for Sophus 1.x (so3.hpp / se3.hpp of the reference tree).
#pragma once
#include <Eigen/Core>
namespace Sophus {
template <class T> class SE3 {
 public:
  SE3() {}
  // (the normalising constructor: SO3(quaternion) divides by the norm)
  SE3(const Eigen::Quaternion<T>& q, const Eigen::Matrix<T, 3, 1>& t) : q_(q.normalized()), t_(t) {}
  static SE3 raw(const Eigen::Quaternion<T>& q, const Eigen::Matrix<T, 3, 1>& t) { SE3 o; o.q_ = q; o.t_ = t; return o; }   // mock only: members as given
  const Eigen::Quaternion<T>& unit_quaternion() const { return q_; }
  const Eigen::Matrix<T, 3, 1>& translation() const { return t_; }
  Eigen::Matrix<T, 3, 3> rotationMatrix() const { return q_.toRotationMatrix(); }
  // (Sophus: SE3<U>(so3().cast<U>(), ...), and SO3<U>(quaternion) normalises -- a float pose cast to double has a unit quaternion to 1e-16)
  template <class U> SE3<U> cast() const { return SE3<U>(q_.template cast<U>(), t_.template cast<U>()); }
  SE3 inverse() const {
    const Eigen::Quaternion<T> qi = q_.conjugate().normalized();
    return raw(qi, qi * (t_ * T(-1)));
  }
  SE3 operator*(const SE3& b) const { return raw((q_ * b.q_).normalized(), t_ + q_ * b.t_); }
  Eigen::Matrix<T, 3, 1> operator*(const Eigen::Matrix<T, 3, 1>& p) const { return q_ * p + t_; }
 private:
  Eigen::Quaternion<T> q_;
  Eigen::Matrix<T, 3, 1> t_;
};
typedef SE3<float> SE3f;
typedef SE3<double> SE3d;
}  // namespace Sophus
