// test stub: g2o::Sim3 as Optimizer::OptimizeSim3 passes it (see tests/stubs/README.md)
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
  Sim3 inverse() const;
  Sim3 operator*(const Sim3& other) const;
  Eigen::Vector3d map(const Eigen::Vector3d& xyz) const;
};
}  // namespace g2o
