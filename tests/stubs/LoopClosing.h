// test stub (tests/stubs/README.md): the one type of ORB_SLAM3::LoopClosing the Optimizer shim names
#pragma once
#include <map>
#include "orbslam3_stub.h"
#include "Thirdparty/g2o/g2o/types/sim3.h"
namespace ORB_SLAM3 {
class LoopClosing {
 public:
  typedef std::map<KeyFrame*, g2o::Sim3> KeyFrameAndPose;   // (with an Eigen aligned allocator in the reference)
};
}  // namespace ORB_SLAM3
