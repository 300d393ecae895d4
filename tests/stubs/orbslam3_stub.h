// test mock: ORB_SLAM3::Frame / KeyFrame / MapPoint / Map / GeometricCamera with the reference's member names and types
// (include/Frame.h, KeyFrame.h, MapPoint.h, Map.h, CameraModels/GeometricCamera.h) and BEHAVING containers behind them:
// observations, keypoint -> map point tables, covisibility lists, spanning tree, bad flags -- enough for the shims under
// dvm_slam_amd/host/ to run against a synthetic map (tests/shim_driver/, tests/test_gpu_shims_run.py).  Own code written
// from the semantics of the reference's methods (cited per method), no reference source.  See tests/stubs/README.md.
// The two accessors marked (+) are the ONLY additions a maintainer makes to the reference.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
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
#include <boost/uuid/uuid.hpp>
#include "dvmslam_hip.h"      // dvm_device_frame (the optional Frame member)

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64
using namespace std;   // the reference does this at global scope (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:36)

namespace ORB_SLAM3 {
class KeyFrame;
class Map;
class Frame;
class MapPoint;

class GeometricCamera {
 public:
  float getParameter(int i) { return p_[i]; }
  float p_[8] = {};
};

class Map {
 public:
  std::vector<KeyFrame*> GetAllKeyFrames() { return mock_kfs; }
  std::vector<MapPoint*> GetAllMapPoints() { return mock_mps; }
  long unsigned int GetInitKFid() { return mock_init_kf_id; }
  long unsigned int GetMaxKFid() { return mock_max_kf_id; }
  KeyFrame* GetOriginKF() { return mock_origin; }
  void IncreaseChangeIndex() { mock_change_index++; }
  bool IsInertial() { return false; }
  bool IsBad() { return mock_bad; }
  bool mock_bad = false;
  void EraseMapPoint(MapPoint* p) { mock_mps.erase(std::remove(mock_mps.begin(), mock_mps.end(), p), mock_mps.end()); }
  std::mutex mMutexMapUpdate;
  std::set<long unsigned int> msOptKFs, msFixedKFs;
  // mock state
  std::vector<KeyFrame*> mock_kfs;
  std::vector<MapPoint*> mock_mps;
  long unsigned int mock_init_kf_id = 0, mock_max_kf_id = 0;
  KeyFrame* mock_origin = nullptr;
  int mock_change_index = 0;
};

class MapPoint {
 public:
  MapPoint(long unsigned int id, const Eigen::Vector3f& pos, Map* map) : mnId(id), mWorldPos(pos), mpMap(map) {
    mDescriptor.create(1, 32, CV_8U);
    std::memset(mDescriptor.data, 0, 32);
  }
  Eigen::Vector3f GetWorldPos() { return mWorldPos; }
  void SetWorldPos(const Eigen::Vector3f& Pos) { mWorldPos = Pos; mock_set_pos++; }
  Eigen::Vector3f GetNormal() { return mNormalVector; }
  // (MapPoint.cc:272-275, 361-367: both take mMutexFeatures, a NON-recursive mutex -- the mock does too, and a caller that already
  //  holds it is reported instead of left to hang)
  std::map<KeyFrame*, std::tuple<int, int>> GetObservations() { MockFeatureLock l(this, "GetObservations"); return mObservations; }
  int Observations() { return nObs; }
  inline void EraseObservation(KeyFrame* pKF, int minObservationsBeforeDeletion = 3);
  inline void AddObservation(KeyFrame* pKF, int idx);
  inline void SetBadFlag();
  inline void Replace(MapPoint* pMP);
  void ComputeDistinctiveDescriptors();          // defined by host/MapPoint_shim.h
  static void ComputeDistinctiveDescriptorsBatch(const std::vector<MapPoint*>& points);   // (+) declared for host/MapPoint_shim.h
  std::tuple<int, int> GetIndexInKeyFrame(KeyFrame* pKF) {       // (MapPoint.cc: (-1, -1) when the keyframe does not observe the point)
    const auto it = mObservations.find(pKF);
    return it == mObservations.end() ? std::tuple<int, int>(-1, -1) : it->second;
  }
  bool IsInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) != 0; }
  bool isBad() { MockFeatureLock l(this, "isBad"); return mbBad; }
  cv::Mat GetDescriptor() { MockFeatureLock l(this, "GetDescriptor"); return mDescriptor.clone(); }
  void UpdateNormalAndDepth() { mock_update_normal++; }
  float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
  float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
  float GetMinDistance() { return mfMinDistance; }   // (+) returns mfMinDistance (MapPoint::PredictScale reads the raw value, MapPoint.cc:573-587)
  float GetMaxDistance() { return mfMaxDistance; }   // (+) returns mfMaxDistance
  Map* GetMap() { return mpMap; }
  KeyFrame* GetReferenceKeyFrame() { return mpRefKF; }
  long unsigned int mnCorrectedByKF = 0, mnCorrectedReference = 0;
  long unsigned int mnId;
  float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackDepthR = 0, mTrackProjXR = 0, mTrackProjYR = 0;
  bool mbTrackInView = false, mbTrackInViewR = false;
  int mnTrackScaleLevel = 0, mnTrackScaleLevelR = 0;
  float mTrackViewCos = 0, mTrackViewCosR = 0;
  long unsigned int mnTrackReferenceForFrame = 0, mnLastFrameSeen = 0;
  long unsigned int mnBALocalForKF = ~0ul, mnBALocalForMerge = ~0ul;
  Eigen::Vector3f mPosGBA;
  long unsigned int mnBAGlobalForKF = 0;
  // mock state (the reference keeps these protected; the mock leaves the ones no shim may touch public for the test driver)
  Eigen::Vector3f mWorldPos, mNormalVector;
  int nObs = 0;
  MapPoint* mpReplaced = nullptr;
  float mfMinDistance = 0, mfMaxDistance = 0;
  struct MockAccess;          // test driver only (tests/shim_driver/): reads / writes the protected members below

 protected:                   // as in include/MapPoint.h:210-248: a free function that touches these does not compile
  std::map<KeyFrame*, std::tuple<int, int>> mObservations;
  bool mbBad = false;
  cv::Mat mDescriptor;
  std::mutex mMutexFeatures;
  struct MockFeatureLock {    // std::mutex cannot name its owner: in the single-threaded shim tests a failed try_lock IS a self-deadlock
    MapPoint* p;
    MockFeatureLock(MapPoint* p_, const char* who) : p(p_) {
      if (!p->mMutexFeatures.try_lock()) {
        std::fprintf(stderr, "mock MapPoint::%s(): mMutexFeatures is already held by the caller -- against the reference this deadlocks\n", who);
        std::abort();
      }
    }
    ~MockFeatureLock() { p->mMutexFeatures.unlock(); }
  };

 public:
  Map* mpMap;
  KeyFrame* mpRefKF = nullptr;
  int mock_set_pos = 0, mock_update_normal = 0;
};

struct MapPoint::MockAccess {
  static bool& bad(MapPoint* p) { return p->mbBad; }
  static cv::Mat& descriptor(MapPoint* p) { return p->mDescriptor; }
  static std::map<KeyFrame*, std::tuple<int, int>>& observations(MapPoint* p) { return p->mObservations; }
};

struct KeyFrameInit {   // mock only: what the reference's KeyFrame(Frame&, Map*, KeyFrameDatabase*) copies out of the frame
  long unsigned int id = 0;
  float fx = 0, fy = 0, cx = 0, cy = 0;
  std::vector<cv::KeyPoint> keysUn;
  cv::Mat descriptors;
  std::vector<float> scaleFactors, levelSigma2, invLevelSigma2;
  float logScaleFactor = 0;
  int minX = 0, minY = 0, maxX = 0, maxY = 0;
};

class KeyFrame {
 public:
  KeyFrame(const KeyFrameInit& s, Map* map)
      : mnId(s.id), fx(s.fx), fy(s.fy), cx(s.cx), cy(s.cy), invfx(1.0f / s.fx), invfy(1.0f / s.fy), N((int)s.keysUn.size()), mvKeys(s.keysUn),
        mvKeysUn(s.keysUn), mvuRight(s.keysUn.size(), -1.0f), mvDepth(s.keysUn.size(), -1.0f), mDescriptors(s.descriptors),
        mnScaleLevels((int)s.scaleFactors.size()), mfScaleFactor(s.scaleFactors.size() > 1 ? s.scaleFactors[1] : 1.2f), mfLogScaleFactor(s.logScaleFactor),
        mvScaleFactors(s.scaleFactors), mvLevelSigma2(s.levelSigma2), mvInvLevelSigma2(s.invLevelSigma2), mnMinX(s.minX), mnMinY(s.minY), mnMaxX(s.maxX),
        mnMaxY(s.maxY), mvpMapPoints(s.keysUn.size(), static_cast<MapPoint*>(nullptr)), mpMap(map) {}
  Sophus::SE3f GetPose() { return mTcw; }
  Sophus::SE3f GetPoseInverse() { return mTwc; }
  Eigen::Vector3f GetCameraCenter() { return mTwc.translation(); }
  Eigen::Matrix3f GetRotation() { return mTcw.rotationMatrix(); }
  Eigen::Vector3f GetTranslation() { return mTcw.translation(); }
  void SetPose(const Sophus::SE3f& Tcw) { mTcw = Tcw; mTwc = Tcw.inverse(); mock_set_pose++; }   // (KeyFrame.cc:224-236)
  void mock_pose(const Sophus::SE3f& Tcw, const Sophus::SE3f& Twc) { mTcw = Tcw; mTwc = Twc; }  // both as given (tests hand over a precomputed inverse)
  std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
  std::set<MapPoint*> GetMapPoints() {                                 // (KeyFrame.cc: the good points among the matches)
    std::set<MapPoint*> s;
    for (MapPoint* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p);
    return s;
  }
  MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
  void EraseMapPointMatch(const int& idx) { mvpMapPoints[idx] = nullptr; }
  void EraseMapPointMatch(MapPoint* pMP) {                             // (KeyFrame.cc: through the point's own index in this keyframe)
    const int left = std::get<0>(pMP->GetIndexInKeyFrame(this));
    if (left != -1) mvpMapPoints[left] = nullptr;
  }
  void ReplaceMapPointMatch(const int& idx, MapPoint* pMP) { mvpMapPoints[idx] = pMP; }
  void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }
  std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { return mvpOrderedConnectedKeyFrames; }
  std::vector<KeyFrame*> GetBestCovisibilityKeyFrames(const int& N) {   // (KeyFrame.cc: the first N of the ordered list)
    if ((int)mvpOrderedConnectedKeyFrames.size() < N) return mvpOrderedConnectedKeyFrames;
    return std::vector<KeyFrame*>(mvpOrderedConnectedKeyFrames.begin(), mvpOrderedConnectedKeyFrames.begin() + N);
  }
  std::set<KeyFrame*> GetConnectedKeyFrames() { return mock_connected; }   // (KeyFrame.cc: the keys of mConnectedKeyFrameWeights)
  boost::uuids::uuid uuid = boost::uuids::nil_uuid();
  std::set<KeyFrame*> mock_connected;
  std::vector<KeyFrame*> GetCovisiblesByWeight(const int& w) {          // (KeyFrame.cc: the ordered list down to weight w)
    std::vector<KeyFrame*> out;
    for (size_t i = 0; i < mvpOrderedConnectedKeyFrames.size(); i++) if (mvOrderedWeights[i] >= w) out.push_back(mvpOrderedConnectedKeyFrames[i]);
    return out;
  }
  int GetWeight(KeyFrame* pKF) {
    for (size_t i = 0; i < mvpOrderedConnectedKeyFrames.size(); i++) if (mvpOrderedConnectedKeyFrames[i] == pKF) return mvOrderedWeights[i];
    return 0;
  }
  KeyFrame* GetParent() { return mpParent; }
  bool hasChild(KeyFrame* pKF) { return mspChildrens.count(pKF) != 0; }
  std::set<KeyFrame*> GetLoopEdges() { return mspLoopEdges; }
  bool bImu = false;
  KeyFrame* mPrevKF = nullptr;
  bool isBad() { return mbBad; }
  Map* GetMap() { return mpMap; }
  void UpdateMap(Map* pMap) { mpMap = pMap; }      // (KeyFrame.cc: under mMutexMap)
  long unsigned int mnId;
  long unsigned int mnBALocalForKF = ~0ul, mnBAFixedForKF = ~0ul, mnBAGlobalForKF = 0, mnBALocalForMerge = ~0ul;
  Sophus::SE3f mTcwGBA, mTcwBefMerge, mTwcBefMerge;
  const float fx, fy, cx, cy, invfx, invfy, mbf = 0, mb = 0, mThDepth = 0;
  const int N;
  const std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
  const std::vector<float> mvuRight, mvDepth;
  const cv::Mat mDescriptors;
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  const int mnScaleLevels;
  const float mfScaleFactor, mfLogScaleFactor;
  const std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  const int mnMinX, mnMinY, mnMaxX, mnMaxY;
  GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
  const int NLeft = -1, NRight = -1;
  // mock state (protected in the reference)
  Sophus::SE3f mTcw, mTwc;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<KeyFrame*> mvpOrderedConnectedKeyFrames;
  std::vector<int> mvOrderedWeights;
  KeyFrame* mpParent = nullptr;
  std::set<KeyFrame*> mspChildrens, mspLoopEdges;
  bool mbBad = false;
  Map* mpMap;
  int mock_set_pose = 0;
};

// (MapPoint.cc:209-233, monocular keyframes)
inline void MapPoint::AddObservation(KeyFrame* pKF, int idx) {
  std::tuple<int, int> ind(-1, -1);
  const auto it = mObservations.find(pKF);
  if (it != mObservations.end()) ind = it->second;
  std::get<0>(ind) = idx;
  mObservations[pKF] = ind;
  nObs++;
}
// (MapPoint.cc:235-270)  the observation goes; a point left with fewer than minObservationsBeforeDeletion observations, or
// without a reference keyframe, is flagged bad
inline void MapPoint::EraseObservation(KeyFrame* pKF, int minObservationsBeforeDeletion) {
  bool bad = false;
  const auto it = mObservations.find(pKF);
  if (it != mObservations.end()) {
    if (std::get<0>(it->second) != -1) nObs--;
    mObservations.erase(it);
    if (mpRefKF == pKF) mpRefKF = mObservations.empty() ? nullptr : mObservations.begin()->first;
    if (!mpRefKF) bad = true;
    if (nObs < minObservationsBeforeDeletion) bad = true;
  }
  if (bad) SetBadFlag();
}
// (MapPoint.cc:282-303)
inline void MapPoint::SetBadFlag() {
  mbBad = true;
  const std::map<KeyFrame*, std::tuple<int, int>> obs = mObservations;
  mObservations.clear();
  for (const auto& o : obs)
    if (std::get<0>(o.second) != -1) o.first->EraseMapPointMatch(std::get<0>(o.second));
  if (mpMap) mpMap->EraseMapPoint(this);
}
// (MapPoint.cc:311-365)  every observation of this point moves to pMP (or is dropped where pMP is already observed)
inline void MapPoint::Replace(MapPoint* pMP) {
  if (pMP->mnId == mnId) return;
  const std::map<KeyFrame*, std::tuple<int, int>> obs = mObservations;
  mObservations.clear();
  mbBad = true;
  mpReplaced = pMP;
  for (const auto& o : obs) {
    KeyFrame* pKF = o.first;
    const int left = std::get<0>(o.second);
    if (!pMP->IsInKeyFrame(pKF)) {
      if (left != -1) { pKF->ReplaceMapPointMatch(left, pMP); pMP->AddObservation(pKF, left); }
    } else if (left != -1) {
      pKF->EraseMapPointMatch(left);
    }
  }
  if (mpMap) mpMap->EraseMapPoint(this);
}

class Frame {
 public:
  Sophus::SE3f GetPose() const { return mTcw; }
  void SetPose(const Sophus::SE3<float>& Tcw) {                        // (Frame.cc:537-559: SetPose + UpdatePoseMatrices)
    mTcw = Tcw;
    const Sophus::SE3f Twc = mTcw.inverse();
    mRcw = mTcw.rotationMatrix(); mtcw = mTcw.translation(); mOw = Twc.translation();
    mock_set_pose++;
  }
  bool isInFrustum(MapPoint* pMP, float viewingCosLimit);
  void UndistortKeyPoints();                          // private in the reference (Frame.h): the shim is compiled into Frame.cc
  void ComputeImageBounds(const cv::Mat& imLeft);
  cv::Mat mK, mDistCoef;
  dvm_device_frame mDvmDevice{};     // (+) optional: `mDvmDevice = mpORBextractorLeft->LastDeviceResult();` after ExtractORB (Frame.cc:411), INTEGRATION.md section 0
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
  // mock state
  Sophus::SE3f mTcw;
  int mock_set_pose = 0;
};
}  // namespace ORB_SLAM3
