// test mock: g2o::Sim3 (rotation, translation, scale in double) with the three operations the Optimizer shim uses
// (see tests/stubs/README.md).  Semantics of the reference's Thirdparty/g2o/g2o/types/sim3.h: inverse = (r*, r* (-t / s), 1 / s),
// product = (r r', s (r t') + t, s s'), map(x) = s (r x) + t.
#pragma once
#include <Eigen/Core>
namespace g2o {
struct Sim3 {
  Eigen::Quaterniond r; Eigen::Vector3d t; double s = 1;
  Sim3() {}
  Sim3(const Eigen::Quaterniond& r_, const Eigen::Vector3d& t_, double s_) : r(r_), t(t_), s(s_) {}
  const Eigen::Quaterniond& rotation() const { return r; }
  const Eigen::Vector3d& translation() const { return t; }
  const double& scale() const { return s; }
  Sim3 inverse() const { return Sim3(r.conjugate(), r.conjugate() * ((-1. / s) * t), 1. / s); }
  Sim3 operator*(const Sim3& o) const { return Sim3(r * o.r, s * (r * o.t) + t, s * o.s); }
  Eigen::Vector3d map(const Eigen::Vector3d& xyz) const { return s * (r * xyz) + t; }
};
}  // namespace g2o
