// test stub (tests/stubs/README.md): the eight monocular, non-inertial statics of ORB_SLAM3::Optimizer (include/Optimizer.h:48-101) the shim defines
#pragma once
#include "orbslam3_stub.h"
#include "Thirdparty/g2o/g2o/types/sim3.h"
#include "LoopClosing.h"
namespace ORB_SLAM3 {
class Optimizer {
 public:
  void static BundleAdjustment(const std::vector<KeyFrame*>& vpKF, const std::vector<MapPoint*>& vpMP, int nIterations = 5,
                               bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true);
  void static GlobalBundleAdjustemnt(Map* pMap, int nIterations = 5, bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0,
                                     const bool bRobust = true);
  void static LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs,
                                    int& num_edges);
  void static LocalBundleAdjustment(KeyFrame* pMainKF, std::vector<KeyFrame*> vpAdjustKF, std::vector<KeyFrame*> vpFixedKF, bool* pbStopFlag);
  int static PoseOptimization(Frame* pFrame);
  static int OptimizeSim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches1, g2o::Sim3& g2oS12, const float th2,
                          const bool bFixScale, Eigen::Matrix<double, 7, 7>& mAcumHessian, const bool bAllPoints = false);
  void static OptimizeEssentialGraph(Map* pMap, KeyFrame* pLoopKF, KeyFrame* pCurKF, const LoopClosing::KeyFrameAndPose& NonCorrectedSim3,
                                     const LoopClosing::KeyFrameAndPose& CorrectedSim3,
                                     const std::map<KeyFrame*, std::set<KeyFrame*>>& LoopConnections, const bool& bFixScale);
  void static OptimizeEssentialGraph(KeyFrame* pCurKF, std::vector<KeyFrame*>& vpFixedKFs, std::vector<KeyFrame*>& vpFixedCorrectedKFs,
                                     std::vector<KeyFrame*>& vpNonFixedKFs, std::vector<MapPoint*>& vpNonCorrectedMPs);
};
}  // namespace ORB_SLAM3
