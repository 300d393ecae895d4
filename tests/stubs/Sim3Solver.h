// test stub (tests/stubs/README.md): the class surface of ORB_SLAM3::Sim3Solver (include/Sim3Solver.h:31-140) -- the members the
// shim's function bodies read and write, nothing else
#pragma once
#include <vector>
#include "orbslam3_stub.h"
namespace ORB_SLAM3 {
using std::vector;
class Sim3Solver {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  Sim3Solver(KeyFrame* pKF1, KeyFrame* pKF2, const std::vector<MapPoint*>& vpMatched12, const bool bFixScale = true,
             const vector<KeyFrame*> vpKeyFrameMatchedMP = vector<KeyFrame*>());
  void SetRansacParameters(double probability = 0.99, int minInliers = 6, int maxIterations = 300);
  Eigen::Matrix4f find(std::vector<bool>& vbInliers12, int& nInliers);
  Eigen::Matrix4f iterate(int nIterations, bool& bNoMore, std::vector<bool>& vbInliers, int& nInliers);
  Eigen::Matrix4f iterate(int nIterations, bool& bNoMore, vector<bool>& vbInliers, int& nInliers, bool& bConverge);
  Eigen::Matrix4f GetEstimatedTransformation();
  Eigen::Matrix3f GetEstimatedRotation();
  Eigen::Vector3f GetEstimatedTranslation();
  float GetEstimatedScale();

 protected:
  KeyFrame* mpKF1;
  KeyFrame* mpKF2;
  std::vector<Eigen::Vector3f> mvX3Dc1, mvX3Dc2;
  std::vector<MapPoint*> mvpMapPoints1, mvpMapPoints2, mvpMatches12;
  std::vector<size_t> mvnIndices1, mvSigmaSquare1, mvSigmaSquare2, mvnMaxError1, mvnMaxError2;
  int N;
  int mN1;
  Eigen::Matrix3f mR12i;
  Eigen::Vector3f mt12i;
  float ms12i;
  Eigen::Matrix4f mT12i, mT21i;
  std::vector<bool> mvbInliersi;
  int mnInliersi;
  int mnIterations;
  std::vector<bool> mvbBestInliers;
  int mnBestInliers;
  Eigen::Matrix4f mBestT12;
  Eigen::Matrix3f mBestRotation;
  Eigen::Vector3f mBestTranslation;
  float mBestScale;
  bool mbFixScale;
  std::vector<size_t> mvAllIndices;
  std::vector<Eigen::Vector2f> mvP1im1, mvP2im2;
  double mRansacProb;
  int mRansacMinInliers;
  int mRansacMaxIts;
  float mTh;
  float mSigma2;
  GeometricCamera *pCamera1, *pCamera2;
};
}  // namespace ORB_SLAM3
