// test stub: Sophus::SE3 as the shims use it (see tests/stubs/README.md)
#pragma once
#include <Eigen/Core>
namespace Sophus {
template <class T> class SE3 {
 public:
  SE3() {}
  SE3(const Eigen::Quaternion<T>& q, const Eigen::Matrix<T, 3, 1>& t) : q_(q), t_(t) {}
  const Eigen::Quaternion<T>& unit_quaternion() const { return q_; }
  const Eigen::Matrix<T, 3, 1>& translation() const { return t_; }
  template <class U> SE3<U> cast() const { return SE3<U>(q_.template cast<U>(), t_.template cast<U>()); }
 private:
  Eigen::Quaternion<T> q_;
  Eigen::Matrix<T, 3, 1> t_;
};
typedef SE3<float> SE3f;
typedef SE3<double> SE3d;
}  // namespace Sophus
