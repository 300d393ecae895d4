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

namespace dvm_host {

struct MapPointPOD {
  float pos[3];        // MapPoint::GetWorldPos()
  uint8_t desc[32];    // MapPoint::GetDescriptor()
  int32_t n_obs;       // MapPoint::Observations()
};

// A local map point as Tracking::SearchLocalPoints leaves it for the matcher: the mTrack* fields written by
// Frame::isInFrustum (dvm_is_in_frustum fills the same values), isBad(), descriptor, Observations().
struct TrackedPointPOD {
  float mTrackProjX, mTrackProjY, mTrackDepth, mTrackViewCos;
  int32_t mnTrackScaleLevel;
  uint8_t mbTrackInView, bad, pad_[2];
  uint8_t desc[32];
  int32_t n_obs;
};

// The members of ORB_SLAM3::Frame the matcher touches (mono).
struct FrameView {
  int N = 0;
  const dvm_keypoint* mvKeysUn = nullptr;   // undistorted keypoints
  const uint8_t* mDescriptors = nullptr;    // N x 32
  int32_t* mvpMapPoints = nullptr;          // index into the map-point array, -1 = NULL
  const uint8_t* mvbOutlier = nullptr;      // may be null (no outliers)
  float Rcw[9], tcw[3];                     // GetPose()
  float fx, fy, cx, cy;                     // pinhole mpCamera
  float mnMinX, mnMaxX, mnMinY, mnMaxY;
  const float* mvScaleFactors = nullptr;
  int nLevels = 8;
};

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

  int last_requeried = 0;  // queries re-issued on the host because an earlier match claimed their keypoint

 private:
  void ComputeThreeMaxima(std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3);  // :1862-1896
  float mfNNratio;
  bool mbCheckOrientation;
  int device_;
  dvm_frame* grid_ = nullptr;
  int grid_cap_ = 0;
  int ensure_grid(const FrameView& F);
};

}  // namespace dvm_host
