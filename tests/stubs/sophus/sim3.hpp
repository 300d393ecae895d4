// test stub: Sophus::Sim3 as the shims use it (see tests/stubs/README.md)
#pragma once
#include <Eigen/Core>
namespace Sophus {
template <class T> class RxSO3 {
 public:
  const Eigen::Quaternion<T>& quaternion() const { return q_; }
  Eigen::Quaternion<T> q_;
};
template <class T> class Sim3 {
 public:
  const RxSO3<T>& rxso3() const { return r_; }
  const Eigen::Quaternion<T>& quaternion() const { return r_.q_; }
  const Eigen::Matrix<T, 3, 1>& translation() const { return t_; }
 private:
  RxSO3<T> r_;
  Eigen::Matrix<T, 3, 1> t_;
};
typedef Sim3<float> Sim3f;
}  // namespace Sophus
