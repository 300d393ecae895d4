// test mock: Sophus::Sim3 / RxSO3 as the shims use them (see tests/stubs/README.md): the quaternion of an RxSO3 carries the
// scale as its squared norm.
#pragma once
#include <Eigen/Core>
namespace Sophus {
template <class T> class RxSO3 {
 public:
  RxSO3() {}
  explicit RxSO3(const Eigen::Quaternion<T>& q) : q_(q) {}
  const Eigen::Quaternion<T>& quaternion() const { return q_; }
  T scale() const { return q_.squaredNorm(); }
  Eigen::Quaternion<T> q_;
};
template <class T> class Sim3 {
 public:
  Sim3() {}
  Sim3(const RxSO3<T>& r, const Eigen::Matrix<T, 3, 1>& t) : r_(r), t_(t) {}
  const RxSO3<T>& rxso3() const { return r_; }
  const Eigen::Quaternion<T>& quaternion() const { return r_.q_; }
  const Eigen::Matrix<T, 3, 1>& translation() const { return t_; }
  T scale() const { return r_.scale(); }
 private:
  RxSO3<T> r_;
  Eigen::Matrix<T, 3, 1> t_;
};
typedef Sim3<float> Sim3f;
typedef RxSO3<float> RxSO3f;
}  // namespace Sophus
