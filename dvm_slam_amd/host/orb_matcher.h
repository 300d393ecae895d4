// dvm_slam_amd/host/orb_matcher.h -- host-side mirror of ORB_SLAM3::ORBmatcher for the accelerated path.
//
// Same names / argument meaning as the reference (include/ORBmatcher.h:37-95), on plain structs instead
// of Frame* / MapPoint* (the reference's classes drag in OpenCV, Sophus, DBoW2; a maintainer maps
// Frame::mvKeysUn, mDescriptors, mvpMapPoints, ... onto FrameView 1:1).
// The descriptor search runs on the GPU through the C ABI (dvm_frame_build + dvm_match_window); the
// sequential semantics of the reference loop (a keypoint claimed by an earlier query is skipped by later
// ones, rotation histogram, three-maxima filter) are reproduced exactly on the host.
#pragma once
#include <cstdint>
#include <vector>

#include "dvmslam_hip.h"
#include "dvmslam_host.h"

namespace dvm_host {

// The POD views live in the public C header (include/dvmslam_host.h, where the C entry points of this library are declared);
// here they get constructors that zero them and the reference's defaults.
typedef dvmh_map_point MapPointPOD;          // MapPoint::GetWorldPos(), GetDescriptor(), Observations()
typedef dvmh_tracked_point TrackedPointPOD;  // a local map point as Tracking::SearchLocalPoints leaves it for the matcher (mTrack* fields, isBad, ...)

// The members of ORB_SLAM3::Frame the matcher touches (mono).
struct FrameView : dvmh_frame_view {
  FrameView() : dvmh_frame_view() { nLevels = 8; }
  FrameView(const dvmh_frame_view& v) : dvmh_frame_view(v) {}
};
// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) flattened: node ids ascending, the features of node k are
// feat[off[k] .. off[k+1]) in insertion order (dvm_host::ORBVocabulary::transform produces exactly this).
struct FeatureVectorView : dvmh_feature_vector_view {
  FeatureVectorView() : dvmh_feature_vector_view() {}
  FeatureVectorView(const dvmh_feature_vector_view& v) : dvmh_feature_vector_view(v) {}
};
// The members of ORB_SLAM3::KeyFrame the matcher touches (mono: NLeft == -1, no mpCamera2, mvuRight < 0).
// GetCameraCenter() = Twc.t (KeyFrame.cc:224-257)
struct KeyFrameView : dvmh_keyframe_view {
  KeyFrameView() : dvmh_keyframe_view() { nLevels = 8; }
  KeyFrameView(const dvmh_keyframe_view& v) : dvmh_keyframe_view(v) {}
  void SetPose(const dvm_se3f& T);       // KeyFrame::SetPose: mTcw = T; mTwc = mTcw.inverse()
};
// A list of map points as the projection searches read them (SoA): GetWorldPos, GetNormal, mfMinDistance / mfMaxDistance
// (GetMin/MaxDistanceInvariance = 0.8 / 1.2 times these), GetDescriptor, isBad, and an id standing for the pointer.
struct MapPointsView : dvmh_map_points_view {
  MapPointsView() : dvmh_map_points_view() {}
  MapPointsView(const dvmh_map_points_view& v) : dvmh_map_points_view(v) {}
};

typedef dvm_sim3f Sim3View;   // Sophus::Sim3f as stored: RxSO3 quaternion (x,y,z,w; scale = |q|^2) + translation

// Pose helpers in the reference's own float arithmetic (csrc/pose_f32.h restates Sophus / Eigen):
// Frame::UpdatePoseMatrices (Frame.cc:553-559): mRcw = mTcw.rotationMatrix(), mtcw, mOw = mTcw.inverse().translation()
void PoseMatrices(const dvm_se3f& Tcw, float* Rcw, float* tcw, float* Ow);
// Tcw = SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale()), Ow = Tcw.inverse().translation() (ORBmatcher.cc:403-404)
void Sim3ToSE3(const dvm_sim3f& Scw, dvm_se3f& Tcw, float* Ow);
dvm_se3f InverseSE3(const dvm_se3f& T);

// process-wide: route SearchByProjection(Cur, Last)'s grid build + window search through a shared search service (NULL: per-thread calls)
void set_match_pool(dvm_match_pool* pool);

// the window queries of SearchByProjection(CurrentFrame, LastFrame): qi = index in LastFrame, (qx, qy, qr) = projection and radius,
// [qmin, qmax] = octave range, qdesc = the map points' descriptors
struct FrameQueries {
  std::vector<int> qi;
  std::vector<float> qx, qy, qr;
  std::vector<int32_t> qmin, qmax;
  std::vector<uint8_t> qdesc;
};
void BuildFrameQueries(const FrameView& Cur, const FrameView& Last, const MapPointPOD* MPs, float th, FrameQueries& Q);

class ORBmatcher {
 public:
  static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;   // ORBmatcher.cc:36-38
  ORBmatcher(float nnratio = 0.6f, bool checkOri = true, int device = 0);
  ~ORBmatcher();
  ORBmatcher(const ORBmatcher&) = delete;

  // ORBmatcher::DescriptorDistance (ORBmatcher.cc:1900-1914)
  static int DescriptorDistance(const uint8_t* a, const uint8_t* b);

  // int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono)
  // (ORBmatcher.cc:1553-1748, monocular): fills CurrentFrame.mvpMapPoints, returns nmatches.
  // Returns < 0 (dvm_status) if the device call fails.
  int SearchByProjection(FrameView& CurrentFrame, const FrameView& LastFrame, const MapPointPOD* mapPoints, float th,
                         bool bMono = true);

  // int SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, const float th, const bool bFarPoints,
  //                        const float thFarPoints)   (ORBmatcher.cc:44-205, monocular: Nleft == -1, no right image).
  // F.mvpMapPoints[idx] receives the INDEX of the matched point in vpMapPoints (offset by mp_index_base); entries
  // already >= 0 refer to `claimedObs`: claimedObs[j] != 0 <=> F.mvpMapPoints[j]->Observations() > 0 at entry.
  int SearchByProjection(FrameView& F, const TrackedPointPOD* vpMapPoints, int nMapPoints, const uint8_t* claimedObs, float th,
                         bool bFarPoints, float thFarPoints, int mp_index_base = 0);
  static float RadiusByViewingCos(float viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; }   // :207-212

  // int SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12,
  //                             int windowSize)   (ORBmatcher.cc:605-707).  vbPrevMatched: 2 floats per F1 keypoint (in/out),
  // vnMatches12: F1.N ints.  The level-0 x level-0 Hamming table comes from the device (dvm_hamming_matrix); the mutual-best
  // bookkeeping (vMatchedDistance / vnMatches21) is sequential and stays on the host.
  int SearchForInitialization(const FrameView& F1, const FrameView& F2, float* vbPrevMatched, int32_t* vnMatches12, int windowSize = 10);

  // int SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)   (:214-393, mono).
  // vpMapPointMatches: F.N map point ids (-1 = NULL).
  int SearchByBoW(const KeyFrameView& KF, const FrameView& F, const FeatureVectorView& FfeatVec, int32_t* vpMapPointMatches);
  // int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12)   (:709-834).  vpMatches12: KF1.N ids of
  // KF2 map points.
  int SearchByBoW(const KeyFrameView& KF1, const KeyFrameView& KF2, int32_t* vpMatches12);

  // int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, vector<pair<size_t,size_t>>& vMatchedPairs,
  //                            const bool bOnlyStereo, const bool bCoarse)   (:836-1058, mono => bOnlyStereo finds nothing).
  // vMatchedPairs: up to KF1.N (idx1, idx2) pairs ordered by idx1.  Whole search on the device (dvm_match_triangulation).
  int SearchForTriangulation(const KeyFrameView& KF1, const KeyFrameView& KF2, int32_t* vMatchedPairs, bool bOnlyStereo = false,
                             bool bCoarse = false);
  // the geometry it derives from the two poses (:841-862, Pinhole.cpp:106-110) through Sophus' SE3 products and Eigen's
  // 3x3 inverse / products: R12, t12, epipole in image 2, F12
  static void TriangulationGeometry(const KeyFrameView& KF1, const KeyFrameView& KF2, float* R12, float* t12, float* ep, float* F12);

  // int Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th, const bool bRight = false)   (:1060-1234),
  // search part: vBestIdx[i] = keypoint of pKF the i-th point would be fused into (bestDist <= TH_LOW), -1 otherwise; the
  // return value counts them.  Points already in the keyframe (IsInKeyFrame) must be flagged in `inKF` (may be null).  The
  // Replace / AddObservation step mutates the map graph and stays with the caller, who replays vBestIdx in order (skipping
  // points that turned bad through an earlier Replace).
  int Fuse(const KeyFrameView& KF, const MapPointsView& vpMapPoints, const uint8_t* inKF, float th, int32_t* vBestIdx);
  // int Fuse(KeyFrame* pKF, Sophus::Sim3f& Scw, const vector<MapPoint*>& vpPoints, float th, vector<MapPoint*>& vpReplacePoint)
  // (:1236-1345), whole function: KF.mvpMapPoints receives the added points, vpReplacePoint[i] the id to replace (-1 none).
  int Fuse(KeyFrameView& KF, const Sim3View& Scw, const MapPointsView& vpPoints, float th, int32_t* vpReplacePoint);
  // int SearchByProjection(KeyFrame* pKF, Sophus::Sim3f& Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched,
  //                        int th, float ratioHamming)   (:395-496), whole function.  vpMatched: KF.N ids (in/out).
  int SearchByProjection(const KeyFrameView& KF, const Sim3View& Scw, const MapPointsView& vpPoints, int32_t* vpMatched, int th,
                         float ratioHamming = 1.f);
  // ... and the overload that also records which keyframe each point came from (:498-603): vpPointsKFs[i] -> vpMatchedKF[idx]
  int SearchByProjection(const KeyFrameView& KF, const Sim3View& Scw, const MapPointsView& vpPoints, const int32_t* vpPointsKFs,
                         int32_t* vpMatched, int32_t* vpMatchedKF, int th, float ratioHamming = 1.f);
  // int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, const float th,
  //                        const int ORBdist)   (:1750-1860, relocalisation), whole function.  MPs: the map point of every
  // keypoint of pKF; sAlreadyFound: ascending ids; CurrentFrame.mvpMapPoints receives ids.
  int SearchByProjection(FrameView& CurrentFrame, const KeyFrameView& KF, const MapPointsView& MPs, const int32_t* sAlreadyFound,
                         int nAlreadyFound, float th, int ORBdist);
  // int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12, const Sophus::Sim3f& S12, const float th)
  // (:1347-1551), whole function.  MPsN: the map point of every keypoint of KFN (entry i is read only where
  // KFN.mvpMapPoints[i] >= 0); vnIdxInKF2[i] = get<0>(vpMatches12[i]->GetIndexInKeyFrame(pKF2)) for the entries set at entry
  // (may be null).  Both directions are one dvm_project_search call each (cam.sim3_pair); the mutual check is host code.
  int SearchBySim3(const KeyFrameView& KF1, const KeyFrameView& KF2, const MapPointsView& MPs1, const MapPointsView& MPs2,
                   int32_t* vpMatches12, const int32_t* vnIdxInKF2, const Sim3View& S12, float th);

  int last_requeried = 0;  // queries re-issued on the host because an earlier match claimed their keypoint

 private:
  void ComputeThreeMaxima(std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3);  // :1862-1896
  void ComputeThreeMaxima(const int* sizes, int L, int& ind1, int& ind2, int& ind3);   // the same on the bins' sizes
  float mfNNratio;
  bool mbCheckOrientation;
  int device_;
 public:
  bool last_grid_from_device = false;     // the last call built its grid from a dvm_device_frame (no upload)
 private:
  dvm_frame* grid_ = nullptr;
  int grid_cap_ = 0;
  int ensure_handle(int N);                 // the calling thread's cached grid handle, at least N keypoints
  bool resident(const FrameView& F) const;  // F's keypoints + descriptors are still in HBM on this matcher's device
  int ensure_grid(const FrameView& F);
  int ensure_grid(const KeyFrameView& KF);
  int project_search(const KeyFrameView& KF, const dvm_se3f& Tcw, const float* Ow, const MapPointsView& P,
                     const uint8_t* valid, const uint8_t* skip, float th, bool gate, std::vector<dvm_match>& res,
                     std::vector<dvm_projection>& proj);
};

}  // namespace dvm_host
