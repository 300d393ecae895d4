// test stub: the members of ORB_SLAM3::Frame / KeyFrame / MapPoint / Map / GeometricCamera the shims touch, with the
// reference's names and types (include/Frame.h, KeyFrame.h, MapPoint.h, Map.h, CameraModels/GeometricCamera.h).  See
// tests/stubs/README.md.  The two accessors marked (+) are the ONLY additions a maintainer makes to the reference.
#pragma once
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <tuple>
#include <vector>

#include <Eigen/Core>
#include <opencv2/core/core.hpp>
#include <sophus/se3.hpp>
#include <sophus/sim3.hpp>

#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64
using namespace std;   // the reference does this at global scope (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:36)

namespace ORB_SLAM3 {
class KeyFrame;
class Map;
class Frame;

class GeometricCamera {
 public:
  float getParameter(int i) { return p_[i]; }
  float p_[8] = {};
};

class MapPoint {
 public:
  Eigen::Vector3f GetWorldPos();
  void SetWorldPos(const Eigen::Vector3f& Pos);
  Eigen::Vector3f GetNormal();
  std::map<KeyFrame*, std::tuple<int, int>> GetObservations();
  int Observations();
  void EraseObservation(KeyFrame* pKF, int minObservationsBeforeDeletion = 3);
  void AddObservation(KeyFrame* pKF, int idx);
  void Replace(MapPoint* pMP);
  std::tuple<int, int> GetIndexInKeyFrame(KeyFrame* pKF);
  bool IsInKeyFrame(KeyFrame* pKF);
  bool isBad();
  cv::Mat GetDescriptor();
  void UpdateNormalAndDepth();
  float GetMinDistanceInvariance();
  float GetMaxDistanceInvariance();
  float GetMinDistance();   // (+) returns mfMinDistance (MapPoint::PredictScale reads the raw value, MapPoint.cc:573-587)
  float GetMaxDistance();   // (+) returns mfMaxDistance
  Map* GetMap();
  KeyFrame* GetReferenceKeyFrame();
  long unsigned int mnCorrectedByKF, mnCorrectedReference;
  long unsigned int mnId;
  float mTrackProjX, mTrackProjY, mTrackDepth, mTrackDepthR, mTrackProjXR, mTrackProjYR;
  bool mbTrackInView, mbTrackInViewR;
  int mnTrackScaleLevel, mnTrackScaleLevelR;
  float mTrackViewCos, mTrackViewCosR;
  long unsigned int mnTrackReferenceForFrame, mnLastFrameSeen;
  long unsigned int mnBALocalForKF;
  Eigen::Vector3f mPosGBA;
  long unsigned int mnBAGlobalForKF;
};

class KeyFrame {
 public:
  Sophus::SE3f GetPose();
  Sophus::SE3f GetPoseInverse();
  Eigen::Vector3f GetCameraCenter();
  Eigen::Matrix3f GetRotation();
  Eigen::Vector3f GetTranslation();
  void SetPose(const Sophus::SE3f& Tcw);
  std::vector<MapPoint*> GetMapPointMatches();
  MapPoint* GetMapPoint(const size_t& idx);
  void EraseMapPointMatch(MapPoint* pMP);
  void AddMapPoint(MapPoint* pMP, const size_t& idx);
  std::vector<KeyFrame*> GetVectorCovisibleKeyFrames();
  std::vector<KeyFrame*> GetCovisiblesByWeight(const int& w);
  int GetWeight(KeyFrame* pKF);
  KeyFrame* GetParent();
  bool hasChild(KeyFrame* pKF);
  std::set<KeyFrame*> GetLoopEdges();
  bool bImu = false;
  KeyFrame* mPrevKF = nullptr;
  bool isBad();
  Map* GetMap();
  long unsigned int mnId;
  long unsigned int mnBALocalForKF, mnBAFixedForKF, mnBAGlobalForKF;
  Sophus::SE3f mTcwGBA;
  const float fx = 0, fy = 0, cx = 0, cy = 0, invfx = 0, invfy = 0, mbf = 0, mb = 0, mThDepth = 0;
  const int N = 0;
  const std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
  const std::vector<float> mvuRight, mvDepth;
  const cv::Mat mDescriptors;
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  const int mnScaleLevels = 8;
  const float mfScaleFactor = 1.2f, mfLogScaleFactor = 0;
  const std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  const int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
  GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
  const int NLeft = -1, NRight = -1;
};

class Map {
 public:
  std::vector<KeyFrame*> GetAllKeyFrames();
  std::vector<MapPoint*> GetAllMapPoints();
  long unsigned int GetInitKFid();
  KeyFrame* GetOriginKF();
  void IncreaseChangeIndex();
  bool IsInertial();
  std::mutex mMutexMapUpdate;
  std::set<long unsigned int> msOptKFs, msFixedKFs;
};

class Frame {
 public:
  Sophus::SE3f GetPose() const;
  void SetPose(const Sophus::SE3<float>& Tcw);
  bool isInFrustum(MapPoint* pMP, float viewingCosLimit);
  void UndistortKeyPoints();                          // private in the reference (Frame.h): the shim is compiled into Frame.cc
  void ComputeImageBounds(const cv::Mat& imLeft);
  cv::Mat mK, mDistCoef;
  int N = 0;
  std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
  std::vector<float> mvuRight, mvDepth;
  cv::Mat mDescriptors;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  static float fx, fy, cx, cy, invfx, invfy;
  float mbf = 0, mb = 0;
  static float mfGridElementWidthInv, mfGridElementHeightInv;
  std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
  int mnScaleLevels = 8;
  float mfScaleFactor = 1.2f, mfLogScaleFactor = 0;
  std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  static float mnMinX, mnMaxX, mnMinY, mnMaxY;
  GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
  int Nleft = -1, Nright = -1;
  long unsigned int mnId = 0;
  Eigen::Matrix3f mRcw;      // private in the reference: the Frame_grid shim is compiled as part of Frame.cc
  Eigen::Vector3f mtcw, mOw;
};
}  // namespace ORB_SLAM3
