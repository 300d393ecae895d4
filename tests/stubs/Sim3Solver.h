// test mock (tests/stubs/README.md) of ORB_SLAM3::Sim3Solver (include/Sim3Solver.h:31-140): the members host/Sim3Solver_shim.h's
// iterate() / find() read and write, plus MOCK bodies of what stays in the reference's src/Sim3Solver.cc (constructor,
// SetRansacParameters, getters) so that the shim can be linked and executed here.  The mock bodies are test infrastructure written
// from the documented behaviour (which correspondences are kept, in which order, with which error bounds) -- nothing of them ships.
#pragma once
#include <cmath>
#include <vector>
#include "orbslam3_stub.h"
namespace ORB_SLAM3 {
using std::vector;
class Sim3Solver {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  // keeps, in keypoint order of pKF1, every match whose two map points are alive and observed by their keyframes (pKF2, or the
  // per-match keyframe of vpKeyFrameMatchedMP); per kept match: both points in their camera frames, the chi2(2 dof, 1 %) bound
  // 9.210 * sigma^2(octave) truncated to size_t, the keypoint index in pKF1
  Sim3Solver(KeyFrame* pKF1, KeyFrame* pKF2, const std::vector<MapPoint*>& vpMatched12, const bool bFixScale = true,
             const vector<KeyFrame*> vpKeyFrameMatchedMP = vector<KeyFrame*>())
      : mN1((int)vpMatched12.size()), mnIterations(0), mnBestInliers(0), mbFixScale(bFixScale), pCamera1(pKF1->mpCamera), pCamera2(pKF2->mpCamera) {
    const std::vector<MapPoint*> own = pKF1->GetMapPointMatches();
    const Sophus::SE3f T1 = pKF1->GetPose(), T2 = pKF2->GetPose();
    const Eigen::Matrix3f R1 = T1.rotationMatrix(), R2 = T2.rotationMatrix();
    struct Kept { int i1, k1, k2; KeyFrame* other; };
    std::vector<Kept> kept;
    for (int i1 = 0; i1 < mN1; i1++) {
      MapPoint *a = own[i1], *b = vpMatched12[i1];
      if (!a || !b || a->isBad() || b->isBad()) continue;
      KeyFrame* other = vpKeyFrameMatchedMP.empty() ? pKF2 : vpKeyFrameMatchedMP[i1];
      const int k1 = std::get<0>(a->GetIndexInKeyFrame(pKF1)), k2 = std::get<0>(b->GetIndexInKeyFrame(other));
      if (k1 >= 0 && k2 >= 0) kept.push_back(Kept{i1, k1, k2, other});
    }
    for (const Kept& c : kept) {
      MapPoint *a = own[c.i1], *b = vpMatched12[c.i1];
      mvnMaxError1.push_back((size_t)(9.210 * pKF1->mvLevelSigma2[pKF1->mvKeysUn[c.k1].octave]));
      mvnMaxError2.push_back((size_t)(9.210 * c.other->mvLevelSigma2[c.other->mvKeysUn[c.k2].octave]));
      mvpMapPoints1.push_back(a);
      mvpMapPoints2.push_back(b);
      mvnIndices1.push_back((size_t)c.i1);
      mvX3Dc1.push_back(R1 * a->GetWorldPos() + T1.translation());
      mvX3Dc2.push_back(R2 * b->GetWorldPos() + T2.translation());
      mvAllIndices.push_back(mvAllIndices.size());
    }
    SetRansacParameters();
  }
  // iteration budget from the inlier ratio the caller hopes for: ceil(log(1 - p) / log(1 - eps^3)), eps = minInliers / N in float
  void SetRansacParameters(double probability = 0.99, int minInliers = 6, int maxIterations = 300) {
    mRansacProb = probability;
    mRansacMinInliers = minInliers;
    N = (int)mvpMapPoints1.size();
    mvbInliersi.assign(N, false);
    int wanted = 1;
    if (minInliers != N) {
      const float eps = (float)minInliers / N;
      wanted = (int)std::ceil(std::log(1 - probability) / std::log(1 - std::pow(eps, 3)));
    }
    mRansacMaxIts = std::max(1, std::min(wanted, maxIterations));
    mnIterations = 0;
  }
  Eigen::Matrix4f find(std::vector<bool>& vbInliers12, int& nInliers);                                        // host/Sim3Solver_shim.h
  Eigen::Matrix4f iterate(int nIterations, bool& bNoMore, std::vector<bool>& vbInliers, int& nInliers);       // host/Sim3Solver_shim.h
  Eigen::Matrix4f iterate(int nIterations, bool& bNoMore, vector<bool>& vbInliers, int& nInliers, bool& bConverge);   // host/Sim3Solver_shim.h
  Eigen::Matrix4f GetEstimatedTransformation() { return mBestT12; }
  Eigen::Matrix3f GetEstimatedRotation() { return mBestRotation; }
  Eigen::Vector3f GetEstimatedTranslation() { return mBestTranslation; }
  float GetEstimatedScale() { return mBestScale; }

 protected:
  std::vector<Eigen::Vector3f> mvX3Dc1, mvX3Dc2;
  std::vector<MapPoint*> mvpMapPoints1, mvpMapPoints2;
  std::vector<size_t> mvnIndices1, mvnMaxError1, mvnMaxError2, mvAllIndices;
  int N, mN1;
  Eigen::Matrix3f mR12i, mBestRotation;
  Eigen::Vector3f mt12i, mBestTranslation;
  float ms12i, mBestScale;
  Eigen::Matrix4f mT12i, mBestT12;
  std::vector<bool> mvbInliersi, mvbBestInliers;
  int mnInliersi, mnIterations, mnBestInliers;
  bool mbFixScale;
  double mRansacProb;
  int mRansacMinInliers, mRansacMaxIts;
  GeometricCamera *pCamera1, *pCamera2;
};
}  // namespace ORB_SLAM3
