// dvm_slam_amd/host/Frame_grid_shim.h -- the Frame members of the accelerated path (reference include/Frame.h:72-74,221-251,
// src/Frame.cc:443-506,575-636,712-782) over the dvmslam_hip C ABI.
//
//   void Frame::UndistortKeyPoints()                                    same name / signature (Frame.cc:791-818): mvKeys -> mvKeysUn
//   void Frame::ComputeImageBounds(const cv::Mat& imLeft)                same name / signature (Frame.cc:820-848)
//        cv::undistortPoints(pts, K, mDistCoef, noArray(), mK) on the device (dvm_undistort_keypoints / dvm_image_bounds);
//        both keep the reference's k1 == 0 shortcut
//   bool Frame::isInFrustum(MapPoint* pMP, float viewingCosLimit)        same name / signature: one map point, device call
//   int  Frame_isInFrustumBatch(Frame&, const std::vector<MapPoint*>&, float viewingCosLimit)
//        what Tracking::SearchLocalPoints (Tracking.cc) should call instead of its per-point loop: every local map point in
//        one dvm_is_in_frustum launch, mbTrackInView / mTrackProj* / mnTrackScaleLevel / mTrackViewCos written back
//
// The FRAME GRID itself (AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea) has no host-visible replacement: its only
// hot callers are the ORBmatcher functions, and ORBmatcher_shim.h builds the grid on the device per call (dvm_frame_build,
// keypoints sorted by cell) and searches it there (dvm_match_window / dvm_project_search).  Frame::mGrid and the host
// GetFeaturesInArea stay compiled for the reference's remaining cold callers; nothing in the accelerated path reads them.
//
// Compile into Frame.cc's translation unit in place of Frame::isInFrustum (the function reads the private mRcw / mtcw / mOw,
// which Frame::UpdatePoseMatrices derives from mTcw exactly as before).
#pragma once
#include <stdexcept>
#include <vector>

#include "Frame.h"
#include "MapPoint.h"
#include "dvm_device.h"
#include "dvmslam_hip.h"

namespace ORB_SLAM3 {
namespace dvm_frame_detail {
inline dvm_frustum_frame frustum(const Frame& F, const Eigen::Matrix3f& mRcw, const Eigen::Vector3f& mtcw, const Eigen::Vector3f& mOw) {
  dvm_frustum_frame f;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) f.Rcw[3 * r + c] = mRcw(r, c);
    f.tcw[r] = mtcw(r); f.Ow[r] = mOw(r);
  }
  f.fx = F.fx; f.fy = F.fy; f.cx = F.cx; f.cy = F.cy;
  f.min_x = F.mnMinX; f.max_x = F.mnMaxX; f.min_y = F.mnMinY; f.max_y = F.mnMaxY;
  f.bf = F.mbf; f.log_scale_factor = F.mfLogScaleFactor; f.n_levels = F.mnScaleLevels;
  return f;
}
inline int run(const dvm_frustum_frame& f, MapPoint* const* pts, int n, float viewingCosLimit) {
  std::vector<float> P(3 * (size_t)n), Nn(3 * (size_t)n), mind(n), maxd(n);
  for (int i = 0; i < n; i++) {
    const Eigen::Vector3f X = pts[i]->GetWorldPos(), N = pts[i]->GetNormal();
    for (int k = 0; k < 3; k++) { P[3 * i + k] = X(k); Nn[3 * i + k] = N(k); }
    mind[i] = pts[i]->GetMinDistance(); maxd[i] = pts[i]->GetMaxDistance();   // raw mfMin/MaxDistance (see ORBmatcher_shim.h)
  }
  std::vector<dvm_track_point> out(n);
  dvm_host::use_device();
  if (dvm_is_in_frustum(&f, P.data(), Nn.data(), mind.data(), maxd.data(), n, viewingCosLimit, out.data(), 0, NULL) != DVM_OK)
    throw std::runtime_error(dvm_last_error());
  int nin = 0;
  for (int i = 0; i < n; i++) {
    MapPoint* p = pts[i];
    const dvm_track_point& o = out[i];
    p->mbTrackInView = o.in_view != 0;
    p->mTrackProjX = o.proj_x; p->mTrackProjY = o.proj_y;     // -1 / the projection reached before a later gate failed (:578-603)
    if (o.in_view) {
      p->mTrackProjXR = o.proj_xr; p->mTrackDepth = o.depth; p->mnTrackScaleLevel = o.level; p->mTrackViewCos = o.view_cos;
      nin++;
    }
  }
  return nin;
}
}  // namespace dvm_frame_detail

namespace dvm_frame_detail {
// K = Pinhole::toK() (== mK for the pinhole camera DVM-SLAM runs), D = mDistCoef: 4 x 1 (k1 k2 p1 p2) or 5 x 1 (+ k3), CV_32F
inline dvm_distortion distortion(const Frame& F) {
  dvm_distortion d;
  d.fx = F.mK.at<float>(0, 0); d.fy = F.mK.at<float>(1, 1); d.cx = F.mK.at<float>(0, 2); d.cy = F.mK.at<float>(1, 2);
  d.k1 = F.mDistCoef.at<float>(0); d.k2 = F.mDistCoef.at<float>(1); d.p1 = F.mDistCoef.at<float>(2); d.p2 = F.mDistCoef.at<float>(3);
  d.k3 = F.mDistCoef.rows * F.mDistCoef.cols > 4 ? F.mDistCoef.at<float>(4) : 0.0f;
  return d;
}
}  // namespace dvm_frame_detail

inline void Frame::UndistortKeyPoints() {
  if (mDistCoef.at<float>(0) == 0.0) {
    mvKeysUn = mvKeys;
    return;
  }
  static_assert(sizeof(cv::KeyPoint) == sizeof(dvm_keypoint), "cv::KeyPoint layout");
  const dvm_distortion d = dvm_frame_detail::distortion(*this);
  mvKeysUn.resize(N);
  if (N == 0) return;
  dvm_host::use_device();
  if (dvm_undistort_keypoints(&d, reinterpret_cast<const dvm_keypoint*>(mvKeys.data()), reinterpret_cast<dvm_keypoint*>(mvKeysUn.data()), N, 0, NULL) != DVM_OK)
    throw std::runtime_error(dvm_last_error());
}

inline void Frame::ComputeImageBounds(const cv::Mat& imLeft) {
  const dvm_distortion d = dvm_frame_detail::distortion(*this);
  float b[4];
  if (dvm_image_bounds(&d, imLeft.cols, imLeft.rows, b) != DVM_OK) throw std::runtime_error(dvm_last_error());
  mnMinX = b[0]; mnMaxX = b[1]; mnMinY = b[2]; mnMaxY = b[3];
}

// bool Frame::isInFrustum(MapPoint* pMP, float viewingCosLimit), mono branch (Nleft == -1)
inline bool Frame::isInFrustum(MapPoint* pMP, float viewingCosLimit) {
  if (Nleft != -1) throw std::runtime_error("Frame::isInFrustum: fisheye stereo pairs are outside the accelerated path");
  const dvm_frustum_frame f = dvm_frame_detail::frustum(*this, mRcw, mtcw, mOw);
  return dvm_frame_detail::run(f, &pMP, 1, viewingCosLimit) == 1;
}

// Every candidate of Tracking::SearchLocalPoints at once.  Returns the number of points in view.
inline int Frame_isInFrustumBatch(Frame& F, const std::vector<MapPoint*>& vpMPs, float viewingCosLimit) {
  if (vpMPs.empty()) return 0;
  const dvm_frustum_frame f = dvm_frame_detail::frustum(F, F.mRcw, F.mtcw, F.mOw);
  return dvm_frame_detail::run(f, vpMPs.data(), (int)vpMPs.size(), viewingCosLimit);
}

}  // namespace ORB_SLAM3
