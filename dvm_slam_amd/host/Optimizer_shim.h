// dvm_slam_amd/host/Optimizer_shim.h -- the monocular, non-inertial statics of ORB_SLAM3::Optimizer (reference
// include/Optimizer.h:48-101, src/Optimizer.cc) over the dvmslam_hip C ABI:
//   BundleAdjustment (:55-356)  GlobalBundleAdjustemnt (:44-53)  LocalBundleAdjustment (:1030-1387)
//   PoseOptimization (:744-1028)  OptimizeSim3 (:1960-2212)
// Same names, same signatures: a maintainer deletes these five bodies from src/Optimizer.cc and compiles this header into
// the same translation unit (every other static -- inertial, essential graph -- stays where it is).  Each function keeps the
// reference's graph GATHERING (which keyframes / map points / observations enter, which vertices are fixed) and its
// WRITE-BACK (outlier erasure, SetPose / SetWorldPos / mTcwGBA ...) statement for statement; only the block between
// "create optimizer" and "recover optimized data" is the device solve.  Stereo / fisheye-pair observations are not part of
// the accelerated path (DVM-SLAM is monocular): a keyframe that carries them is rejected with an exception.
#pragma once
#include <cmath>
#include <list>
#include <map>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "Frame.h"
#include "KeyFrame.h"
#include "Map.h"
#include "MapPoint.h"
#include "Optimizer.h"
#include "dvmslam_hip.h"

namespace ORB_SLAM3 {
namespace dvm_optimizer_detail {

inline void check(int rc) { if (rc != DVM_OK) throw std::runtime_error(dvm_last_error()); }
// (t, q_xyzw) as doubles <- g2o::SE3Quat(Tcw.unit_quaternion().cast<double>(), Tcw.translation().cast<double>())
inline void pose7(const Sophus::SE3f& T, double* p) {
  for (int i = 0; i < 3; i++) p[i] = (double)T.translation()(i);
  for (int i = 0; i < 4; i++) p[3 + i] = (double)T.unit_quaternion().coeffs()(i);
}
// Sophus::SE3f(SE3quat.rotation().cast<float>(), SE3quat.translation().cast<float>())
inline Sophus::SE3f se3f(const double* p) {
  Eigen::Quaterniond q(p[6], p[3], p[4], p[5]);
  Eigen::Vector3d t;
  t(0) = p[0]; t(1) = p[1]; t(2) = p[2];
  return Sophus::SE3f(q.cast<float>(), t.cast<float>());
}
inline void require_mono(KeyFrame* pKF, int leftIndex) {
  if (pKF->mpCamera2 || (leftIndex != -1 && !pKF->mvuRight.empty() && pKF->mvuRight[leftIndex] >= 0))
    throw std::runtime_error("Optimizer shim: stereo / two-camera observation outside the accelerated (monocular) path");
}

// the device solve shared by BundleAdjustment and LocalBundleAdjustment
struct BAProblem {
  std::vector<KeyFrame*> kfs;
  std::vector<uint8_t> fixed;
  std::vector<MapPoint*> mps;
  std::unordered_map<KeyFrame*, int32_t> kf_index;
  std::vector<dvm_ba_edge> edges;
  std::vector<KeyFrame*> edge_kf;
  std::vector<MapPoint*> edge_mp;
  std::vector<double> poses, points, chi2;
  std::vector<uint8_t> depth_pos;
  int add_kf(KeyFrame* pKF, bool fix) {
    kf_index[pKF] = (int32_t)kfs.size();
    kfs.push_back(pKF); fixed.push_back(fix ? 1 : 0);
    return (int)kfs.size() - 1;
  }
  void solve(int iterations, bool* pbStopFlag, double huber_delta) {
    poses.resize(7 * kfs.size()); points.resize(3 * mps.size());
    for (size_t i = 0; i < kfs.size(); i++) pose7(kfs[i]->GetPose(), &poses[7 * i]);
    for (size_t i = 0; i < mps.size(); i++) {
      const Eigen::Vector3d X = mps[i]->GetWorldPos().cast<double>();
      for (int k = 0; k < 3; k++) points[3 * i + k] = X(k);
    }
    KeyFrame* k0 = kfs[0];
    dvm_ba_camera cam = {k0->fx, k0->fy, k0->cx, k0->cy, huber_delta};
    dvm_ba* ba = NULL;
    check(dvm_ba_create(0, &ba));
    int rc = dvm_ba_set_problem(ba, poses.data(), fixed.data(), (int)kfs.size(), points.data(), (int)mps.size(), edges.data(),
                                (int)edges.size(), &cam);
    dvm_ba_stats st;
    static_assert(sizeof(bool) == 1, "bool* pbStopFlag is read as a byte");
    if (rc == DVM_OK) rc = dvm_ba_optimize(ba, iterations, reinterpret_cast<const volatile uint8_t*>(pbStopFlag), &st);
    if (rc == DVM_OK) rc = dvm_ba_get_result(ba, poses.data(), points.data());
    chi2.resize(edges.size()); depth_pos.resize(edges.size());
    if (rc == DVM_OK) rc = dvm_ba_edge_chi2(ba, chi2.data(), depth_pos.data());
    dvm_ba_destroy(ba);
    check(rc);
  }
};

}  // namespace dvm_optimizer_detail

inline void Optimizer::GlobalBundleAdjustemnt(Map* pMap, int nIterations, bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust) {
  vector<KeyFrame*> vpKFs = pMap->GetAllKeyFrames();
  vector<MapPoint*> vpMP = pMap->GetAllMapPoints();
  BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust);
}

inline void Optimizer::BundleAdjustment(const vector<KeyFrame*>& vpKFs, const vector<MapPoint*>& vpMP, int nIterations, bool* pbStopFlag,
                                        const unsigned long nLoopKF, const bool bRobust) {
  using namespace dvm_optimizer_detail;
  vector<bool> vbNotIncludedMP;
  vbNotIncludedMP.resize(vpMP.size());
  Map* pMap = vpKFs[0]->GetMap();
  BAProblem B;
  long unsigned int maxKFid = 0;
  // Set KeyFrame vertices
  for (size_t i = 0; i < vpKFs.size(); i++) {
    KeyFrame* pKF = vpKFs[i];
    if (pKF->isBad()) continue;
    B.add_kf(pKF, pKF->mnId == pMap->GetInitKFid());
    if (pKF->mnId > maxKFid) maxKFid = pKF->mnId;
  }
  const float thHuber2D = sqrt(5.99);
  std::vector<int32_t> mp_vertex(vpMP.size(), -1);
  // Set MapPoint vertices + edges
  for (size_t i = 0; i < vpMP.size(); i++) {
    MapPoint* pMP = vpMP[i];
    if (pMP->isBad()) continue;
    const map<KeyFrame*, tuple<int, int>> observations = pMP->GetObservations();
    int nEdges = 0;
    const int32_t vid = (int32_t)B.mps.size();
    for (map<KeyFrame*, tuple<int, int>>::const_iterator mit = observations.begin(); mit != observations.end(); mit++) {
      KeyFrame* pKF = mit->first;
      if (pKF->isBad() || pKF->mnId > maxKFid) continue;
      auto kit = B.kf_index.find(pKF);
      if (kit == B.kf_index.end()) continue;            // optimizer.vertex(pKF->mnId) == NULL
      nEdges++;
      const int leftIndex = get<0>(mit->second);
      require_mono(pKF, leftIndex);
      if (leftIndex != -1) {
        const cv::KeyPoint& kpUn = pKF->mvKeysUn[leftIndex];
        const float& invSigma2 = pKF->mvInvLevelSigma2[kpUn.octave];
        dvm_ba_edge e = {kit->second, vid, kpUn.pt.x, kpUn.pt.y, invSigma2};
        B.edges.push_back(e); B.edge_kf.push_back(pKF); B.edge_mp.push_back(pMP);
      }
    }
    if (nEdges == 0) {
      vbNotIncludedMP[i] = true;                         // optimizer.removeVertex(vPoint)
      while (!B.edges.empty() && B.edges.back().point == vid) { B.edges.pop_back(); B.edge_kf.pop_back(); B.edge_mp.pop_back(); }
    } else {
      vbNotIncludedMP[i] = false;
      mp_vertex[i] = vid;
      B.mps.push_back(pMP);
    }
  }
  // Optimize!
  B.solve(nIterations, pbStopFlag, bRobust ? (double)thHuber2D : 0.0);
  // Recover optimized data: keyframes
  for (size_t i = 0; i < B.kfs.size(); i++) {
    KeyFrame* pKF = B.kfs[i];
    if (nLoopKF == pMap->GetOriginKF()->mnId) {
      pKF->SetPose(se3f(&B.poses[7 * i]));
    } else {
      pKF->mTcwGBA = se3f(&B.poses[7 * i]);   // Sophus::SE3d(...).cast<float>()
      pKF->mnBAGlobalForKF = nLoopKF;
      // (the reference's per-keyframe bad / good point census under `dist > 1` only fills local counters: nothing to write back)
    }
  }
  // Points
  for (size_t i = 0; i < vpMP.size(); i++) {
    if (vbNotIncludedMP[i]) continue;
    MapPoint* pMP = vpMP[i];
    if (pMP->isBad() || mp_vertex[i] < 0) continue;
    Eigen::Vector3d X;
    for (int k = 0; k < 3; k++) X(k) = B.points[3 * mp_vertex[i] + k];
    if (nLoopKF == pMap->GetOriginKF()->mnId) {
      pMP->SetWorldPos(X.cast<float>());
      pMP->UpdateNormalAndDepth();
    } else {
      pMP->mPosGBA = X.cast<float>();
      pMP->mnBAGlobalForKF = nLoopKF;
    }
  }
}

inline void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs,
                                             int& num_edges) {
  using namespace dvm_optimizer_detail;
  // Local KeyFrames: First Breath Search from Current Keyframe
  list<KeyFrame*> lLocalKeyFrames;
  lLocalKeyFrames.push_back(pKF);
  pKF->mnBALocalForKF = pKF->mnId;
  Map* pCurrentMap = pKF->GetMap();
  const vector<KeyFrame*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
  for (int i = 0, iend = vNeighKFs.size(); i < iend; i++) {
    KeyFrame* pKFi = vNeighKFs[i];
    pKFi->mnBALocalForKF = pKF->mnId;
    if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lLocalKeyFrames.push_back(pKFi);
  }
  // Local MapPoints seen in Local KeyFrames
  num_fixedKF = 0;
  list<MapPoint*> lLocalMapPoints;
  for (list<KeyFrame*>::iterator lit = lLocalKeyFrames.begin(), lend = lLocalKeyFrames.end(); lit != lend; lit++) {
    KeyFrame* pKFi = *lit;
    if (pKFi->mnId == pMap->GetInitKFid()) num_fixedKF = 1;
    vector<MapPoint*> vpMPs = pKFi->GetMapPointMatches();
    for (vector<MapPoint*>::iterator vit = vpMPs.begin(), vend = vpMPs.end(); vit != vend; vit++) {
      MapPoint* pMP = *vit;
      if (pMP)
        if (!pMP->isBad() && pMP->GetMap() == pCurrentMap)
          if (pMP->mnBALocalForKF != pKF->mnId) {
            lLocalMapPoints.push_back(pMP);
            pMP->mnBALocalForKF = pKF->mnId;
          }
    }
  }
  // Fixed Keyframes. Keyframes that see Local MapPoints but that are not Local Keyframes
  list<KeyFrame*> lFixedCameras;
  for (list<MapPoint*>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++) {
    map<KeyFrame*, tuple<int, int>> observations = (*lit)->GetObservations();
    for (map<KeyFrame*, tuple<int, int>>::iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
      KeyFrame* pKFi = mit->first;
      if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
        pKFi->mnBAFixedForKF = pKF->mnId;
        if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lFixedCameras.push_back(pKFi);
      }
    }
  }
  num_fixedKF = lFixedCameras.size() + num_fixedKF;
  if (num_fixedKF == 0) return;   // "LM-LBA: There are 0 fixed KF in the optimizations, LBA aborted"
  if (pMap->IsInertial()) throw std::runtime_error("Optimizer shim: inertial maps use the reference's own solver");

  BAProblem B;
  pCurrentMap->msOptKFs.clear();
  pCurrentMap->msFixedKFs.clear();
  // Set Local KeyFrame vertices, then the fixed ones
  for (list<KeyFrame*>::iterator lit = lLocalKeyFrames.begin(), lend = lLocalKeyFrames.end(); lit != lend; lit++) {
    KeyFrame* pKFi = *lit;
    B.add_kf(pKFi, pKFi->mnId == pMap->GetInitKFid());
    pCurrentMap->msOptKFs.insert(pKFi->mnId);
  }
  num_OptKF = lLocalKeyFrames.size();
  for (list<KeyFrame*>::iterator lit = lFixedCameras.begin(), lend = lFixedCameras.end(); lit != lend; lit++) {
    KeyFrame* pKFi = *lit;
    B.add_kf(pKFi, true);
    pCurrentMap->msFixedKFs.insert(pKFi->mnId);
  }
  const float thHuberMono = sqrt(5.991);
  int nPoints = 0, nEdges = 0;
  for (list<MapPoint*>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++) {
    MapPoint* pMP = *lit;
    const int32_t vid = (int32_t)B.mps.size();
    B.mps.push_back(pMP);
    nPoints++;
    const map<KeyFrame*, tuple<int, int>> observations = pMP->GetObservations();
    for (map<KeyFrame*, tuple<int, int>>::const_iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
      KeyFrame* pKFi = mit->first;
      if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) {
        const int leftIndex = get<0>(mit->second);
        require_mono(pKFi, leftIndex);
        if (leftIndex != -1) {   // Monocular observation
          const cv::KeyPoint& kpUn = pKFi->mvKeysUn[leftIndex];
          const float& invSigma2 = pKFi->mvInvLevelSigma2[kpUn.octave];
          dvm_ba_edge e = {B.kf_index.at(pKFi), vid, kpUn.pt.x, kpUn.pt.y, invSigma2};
          B.edges.push_back(e); B.edge_kf.push_back(pKFi); B.edge_mp.push_back(pMP);
          nEdges++;
        }
      }
    }
  }
  num_MPs = nPoints;
  num_edges = nEdges;
  if (pbStopFlag)
    if (*pbStopFlag) return;

  B.solve(10, pbStopFlag, (double)thHuberMono);

  vector<pair<KeyFrame*, MapPoint*>> vToErase;
  vToErase.reserve(B.edges.size());
  // Check inlier observations
  for (size_t i = 0, iend = B.edges.size(); i < iend; i++) {
    MapPoint* pMP = B.edge_mp[i];
    if (pMP->isBad()) continue;
    if (B.chi2[i] > 5.991 || !B.depth_pos[i]) vToErase.push_back(make_pair(B.edge_kf[i], pMP));
  }
  // Get Map Mutex
  unique_lock<mutex> lock(pMap->mMutexMapUpdate);
  if (!vToErase.empty()) {
    for (size_t i = 0; i < vToErase.size(); i++) {
      KeyFrame* pKFi = vToErase[i].first;
      MapPoint* pMPi = vToErase[i].second;
      pKFi->EraseMapPointMatch(pMPi);
      pMPi->EraseObservation(pKFi);
    }
  }
  // Recover optimized data: keyframes, points
  for (list<KeyFrame*>::iterator lit = lLocalKeyFrames.begin(), lend = lLocalKeyFrames.end(); lit != lend; lit++) {
    KeyFrame* pKFi = *lit;
    pKFi->SetPose(se3f(&B.poses[7 * B.kf_index.at(pKFi)]));
  }
  {
    size_t i = 0;
    for (list<MapPoint*>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++, i++) {
      MapPoint* pMP = *lit;
      Eigen::Vector3d X;
      for (int k = 0; k < 3; k++) X(k) = B.points[3 * i + k];
      pMP->SetWorldPos(X.cast<float>());
      pMP->UpdateNormalAndDepth();
    }
  }
  pMap->IncreaseChangeIndex();
}

inline int Optimizer::PoseOptimization(Frame* pFrame) {
  using namespace dvm_optimizer_detail;
  int nInitialCorrespondences = 0;
  const int N = pFrame->N;
  std::vector<double> Xw, obs, w;
  std::vector<size_t> vnIndexEdgeMono;
  {
    for (int i = 0; i < N; i++) {
      MapPoint* pMP = pFrame->mvpMapPoints[i];
      if (!pMP) continue;
      if (pFrame->mpCamera2 || (!pFrame->mvuRight.empty() && pFrame->mvuRight[i] >= 0))
        throw std::runtime_error("Optimizer shim: stereo observation outside the accelerated (monocular) path");
      nInitialCorrespondences++;
      pFrame->mvbOutlier[i] = false;
      const cv::KeyPoint& kpUn = pFrame->mvKeysUn[i];
      obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y);
      w.push_back(pFrame->mvInvLevelSigma2[kpUn.octave]);
      const Eigen::Vector3d X = pMP->GetWorldPos().cast<double>();
      Xw.push_back(X(0)); Xw.push_back(X(1)); Xw.push_back(X(2));
      vnIndexEdgeMono.push_back(i);
    }
  }
  if (nInitialCorrespondences < 3) return 0;
  double pose_in[7], pose_out[7];
  pose7(pFrame->GetPose(), pose_in);
  const int32_t n = nInitialCorrespondences;
  std::vector<uint8_t> outlier(n);
  int32_t ret = 0;
  dvm_ba_camera cam = {pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy, 0.0};
  check(dvm_pose_optimize(0, pose_in, Xw.data(), obs.data(), w.data(), &n, n, 1, &cam, pose_out, outlier.data(), &ret));
  for (int k = 0; k < n; k++) pFrame->mvbOutlier[vnIndexEdgeMono[k]] = outlier[k] != 0;
  // Recover optimized pose and return number of inliers
  pFrame->SetPose(se3f(pose_out));
  return ret;
}

inline int Optimizer::OptimizeSim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches1, g2o::Sim3& g2oS12, const float th2,
                                   const bool bFixScale, Eigen::Matrix<double, 7, 7>& mAcumHessian, const bool bAllPoints) {
  using namespace dvm_optimizer_detail;
  // Camera poses (matrix form, as the reference: P3D1c = R1w * P3D1w + t1w in float, Eigen's a0 + (a1 + a2) sums)
  const Eigen::Matrix3f R1w = pKF1->GetRotation();
  const Eigen::Vector3f t1w = pKF1->GetTranslation();
  const Eigen::Matrix3f R2w = pKF2->GetRotation();
  const Eigen::Vector3f t2w = pKF2->GetTranslation();
  auto to_cam = [](const Eigen::Matrix3f& R, const Eigen::Vector3f& t, const Eigen::Vector3f& P, double* out) {
    for (int r = 0; r < 3; r++) out[r] = (double)((R(r, 0) * P(0) + (R(r, 1) * P(1) + R(r, 2) * P(2))) + t(r));
  };
  const int N = vpMatches1.size();
  const vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
  std::vector<double> P1c, P2c, obs1, obs2, w1, w2;
  std::vector<size_t> vnIndexEdge;
  int nCorrespondences = 0;
  for (int i = 0; i < N; i++) {
    if (!vpMatches1[i]) continue;
    MapPoint* pMP1 = vpMapPoints1[i];
    MapPoint* pMP2 = vpMatches1[i];
    const int i2 = get<0>(pMP2->GetIndexInKeyFrame(pKF2));
    double c1[3], c2[3];
    if (pMP1 && pMP2) {
      if (!pMP1->isBad() && !pMP2->isBad()) {
        to_cam(R1w, t1w, pMP1->GetWorldPos(), c1);
        to_cam(R2w, t2w, pMP2->GetWorldPos(), c2);
      } else {
        continue;
      }
    } else {
      continue;   // the 3D position in KF1 doesn't exist: the reference adds an unconnected vertex only
    }
    if (i2 < 0 && !bAllPoints) continue;
    if ((float)c2[2] < 0) continue;
    nCorrespondences++;
    const cv::KeyPoint& kpUn1 = pKF1->mvKeysUn[i];
    obs1.push_back(kpUn1.pt.x); obs1.push_back(kpUn1.pt.y);
    w1.push_back(pKF1->mvInvLevelSigma2[kpUn1.octave]);
    if (i2 >= 0) {
      const cv::KeyPoint& kpUn2 = pKF2->mvKeysUn[i2];
      obs2.push_back(kpUn2.pt.x); obs2.push_back(kpUn2.pt.y);
      w2.push_back(pKF2->mvInvLevelSigma2[kpUn2.octave]);
    } else {
      const float invz = 1 / (float)c2[2];
      const float x = (float)c2[0] * invz, y = (float)c2[1] * invz;
      obs2.push_back(x); obs2.push_back(y);
      // kpUn2 = cv::KeyPoint(cv::Point2f(x, y), pMP2->mnTrackScaleLevel): the SIZE argument, so octave = 0
      w2.push_back(pKF2->mvInvLevelSigma2[0]);
    }
    for (int k = 0; k < 3; k++) { P1c.push_back(c1[k]); P2c.push_back(c2[k]); }
    vnIndexEdge.push_back(i);
  }
  if (nCorrespondences == 0) return 0;
  double S[8] = {g2oS12.rotation().x(), g2oS12.rotation().y(), g2oS12.rotation().z(), g2oS12.rotation().w(),
                 g2oS12.translation()(0), g2oS12.translation()(1), g2oS12.translation()(2), g2oS12.scale()};
  const double K1[4] = {pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy}, K2[4] = {pKF2->fx, pKF2->fy, pKF2->cx, pKF2->cy};
  std::vector<uint8_t> inlier(nCorrespondences);
  int32_t nIn = 0;
  check(dvm_optimize_sim3(0, S, bFixScale ? 1 : 0, P1c.data(), P2c.data(), obs1.data(), obs2.data(), w1.data(), w2.data(), nCorrespondences,
                          K1, K2, (double)th2, inlier.data(), &nIn));
  for (int k = 0; k < nCorrespondences; k++)
    if (!inlier[k]) vpMatches1[vnIndexEdge[k]] = static_cast<MapPoint*>(NULL);
  if (nIn == 0) return 0;      // fewer than 10 pairs survived the first round: g2oS12 stays as it was
  mAcumHessian.setZero();      // mAcumHessian = Eigen::MatrixXd::Zero(7, 7) (never accumulated in the reference either)
  Eigen::Vector3d t;
  t(0) = S[4]; t(1) = S[5]; t(2) = S[6];
  g2oS12 = g2o::Sim3(Eigen::Quaterniond(S[3], S[0], S[1], S[2]), t, S[7]);
  return nIn;
}

}  // namespace ORB_SLAM3
