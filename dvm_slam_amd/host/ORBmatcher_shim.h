// dvm_slam_amd/host/ORBmatcher_shim.h -- drop-in for ORB_SLAM3::ORBmatcher (reference include/ORBmatcher.h:37-95,
// src/ORBmatcher.cc) over dvm_host::ORBmatcher (host/orb_matcher.h) and the dvmslam_hip C ABI.
//
// Same class name, same public signatures: compile it inside the reference tree in place of src/ORBmatcher.cc and link
// libdvmslam_host.so + libdvmslam_hip.so; Tracking / LocalMapping / LoopClosing keep calling it unchanged.  Every method
// gathers the Frame / KeyFrame / MapPoint members the reference function reads into the POD views of orb_matcher.h
// (map points become small integer ids for the duration of the call), runs the mirror -- one batched device search + the
// reference's sequential bookkeeping replayed on the host -- and writes the results back through the reference's own
// setters.  Poses are handed over as Sophus stores them (quaternion + translation): the mirror evaluates Sophus' own
// point action (csrc/pose_f32.h).  Monocular only, like DVM-SLAM (src/slam_system/src/ros_mono.cpp).
//
// The reference needs two one-line accessors for this file: MapPoint::GetMinDistance() / GetMaxDistance() returning
// mfMinDistance / mfMaxDistance (MapPoint::PredictScale divides the RAW mfMaxDistance, MapPoint.cc:573-587; only the
// 0.8x / 1.2x "invariance" values have getters today).
#pragma once
#include <algorithm>
#include <cstring>
#include <set>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
#include "dvm_device.h"
#include "orb_matcher.h"

namespace ORB_SLAM3 {

class ORBmatcher {
 public:
  ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri), m_(nnratio, checkOri, dvm_host::device()) {}

  // Computes the Hamming distance between two ORB descriptors
  bool dvmLastGridFromDevice() const { return m_.last_grid_from_device; }   // (diagnostic: the last search built its grid from Frame::mDvmDevice)
  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
    return dvm_host::ORBmatcher::DescriptorDistance(a.ptr<uint8_t>(), b.ptr<uint8_t>());
  }

  // (ORBmatcher.cc:44-205) TrackLocalMap: the mTrack* fields were filled by Frame::isInFrustum
  int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3, const bool bFarPoints = false,
                         const float thFarPoints = 50.0f) {
    std::vector<dvm_host::TrackedPointPOD> pts(vpMapPoints.size());
    for (size_t i = 0; i < vpMapPoints.size(); i++) {
      MapPoint* p = vpMapPoints[i];
      dvm_host::TrackedPointPOD& t = pts[i];
      std::memset(&t, 0, sizeof(t));
      t.mTrackProjX = p->mTrackProjX; t.mTrackProjY = p->mTrackProjY; t.mTrackDepth = p->mTrackDepth; t.mTrackViewCos = p->mTrackViewCos;
      t.mnTrackScaleLevel = p->mnTrackScaleLevel; t.mbTrackInView = p->mbTrackInView; t.bad = p->isBad();
      if (t.mbTrackInView && !t.bad) { copy_desc(t.desc, p); t.n_obs = p->Observations(); }
    }
    // entries of F.mvpMapPoints: 0 = keeps the map point it has, -1 = NULL, i + 1 = receives vpMapPoints[i]
    std::vector<int32_t> mp(F.N);
    std::vector<uint8_t> claimed(F.N, 0);
    for (int j = 0; j < F.N; j++) {
      mp[j] = F.mvpMapPoints[j] ? 0 : -1;
      if (F.mvpMapPoints[j] && F.mvpMapPoints[j]->Observations() > 0) claimed[j] = 1;
    }
    dvm_host::FrameView V = view(F, mp.data());
    const int n = check(m_.SearchByProjection(V, pts.data(), (int)pts.size(), claimed.data(), th, bFarPoints, thFarPoints, 1));
    for (int j = 0; j < F.N; j++)
      if (mp[j] >= 1) F.mvpMapPoints[j] = vpMapPoints[mp[j] - 1];
    return n;
  }

  // (ORBmatcher.cc:1553-1748) TrackWithMotionModel
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) {
    Registry R;
    std::vector<int32_t> mpl(LastFrame.N), mpc(CurrentFrame.N);
    for (int i = 0; i < LastFrame.N; i++) mpl[i] = R.id(LastFrame.mvpMapPoints[i]);
    for (int i = 0; i < CurrentFrame.N; i++) mpc[i] = R.id(CurrentFrame.mvpMapPoints[i]);
    std::vector<dvm_host::MapPointPOD> pods(R.size());
    for (int k = 0; k < R.size(); k++) {
      MapPoint* p = R.ptr(k);
      const Eigen::Vector3f X = p->GetWorldPos();
      pods[k].pos[0] = X(0); pods[k].pos[1] = X(1); pods[k].pos[2] = X(2);
      copy_desc(pods[k].desc, p);
      pods[k].n_obs = p->Observations();
    }
    std::vector<uint8_t> outl(LastFrame.N);
    for (int i = 0; i < LastFrame.N; i++) outl[i] = LastFrame.mvbOutlier[i];
    dvm_host::FrameView C = view(CurrentFrame, mpc.data());
    dvm_host::FrameView L = view(const_cast<Frame&>(LastFrame), mpl.data());
    L.mvbOutlier = outl.data();
    // (the reference takes nLastOctave from LastFrame.mvKeys[i]: undistortion moves pt only, octave / angle are those of mvKeysUn)
    const int n = check(m_.SearchByProjection(C, L, pods.data(), th, bMono));
    for (int i = 0; i < CurrentFrame.N; i++) CurrentFrame.mvpMapPoints[i] = R.ptr(mpc[i]);
    return n;
  }

  // (ORBmatcher.cc:1750-1860) relocalisation
  int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist) {
    Registry R;
    KFPack K(R, pKF);
    std::vector<int32_t> mpc(CurrentFrame.N);
    for (int i = 0; i < CurrentFrame.N; i++) mpc[i] = R.id(CurrentFrame.mvpMapPoints[i]);
    std::vector<int32_t> already;
    for (MapPoint* p : sAlreadyFound) if (p) already.push_back(R.id(p));
    std::sort(already.begin(), already.end());
    PointPack P(K.per_keypoint_points());
    dvm_host::FrameView C = view(CurrentFrame, mpc.data());
    const int n = check(m_.SearchByProjection(C, K.v, P.v, already.data(), (int)already.size(), th, ORBdist));
    for (int i = 0; i < CurrentFrame.N; i++) CurrentFrame.mvpMapPoints[i] = R.ptr(mpc[i]);
    return n;
  }

  // (ORBmatcher.cc:395-496) loop detection
  int SearchByProjection(KeyFrame* pKF, Sophus::Sim3<float>& Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched,
                         int th, float ratioHamming = 1.0) {
    Registry R;
    KFPack K(R, pKF);
    PointPack P(R, vpPoints);
    std::vector<int32_t> matched(vpMatched.size());
    for (size_t i = 0; i < vpMatched.size(); i++) matched[i] = R.id(vpMatched[i]);
    const int n = check(m_.SearchByProjection(K.v, sim3(Scw), P.v, matched.data(), th, ratioHamming));
    for (size_t i = 0; i < vpMatched.size(); i++) vpMatched[i] = R.ptr(matched[i]);
    return n;
  }

  // (ORBmatcher.cc:498-603) place recognition: also records the keyframe each matched point came from
  int SearchByProjection(KeyFrame* pKF, Sophus::Sim3<float>& Scw, const std::vector<MapPoint*>& vpPoints,
                         const std::vector<KeyFrame*>& vpPointsKFs, std::vector<MapPoint*>& vpMatched, std::vector<KeyFrame*>& vpMatchedKF,
                         int th, float ratioHamming = 1.0) {
    Registry R;
    KFPack K(R, pKF);
    PointPack P(R, vpPoints);
    std::vector<int32_t> matched(vpMatched.size()), pkf(vpPoints.size()), mkf(vpMatchedKF.size(), -1);
    for (size_t i = 0; i < vpMatched.size(); i++) matched[i] = R.id(vpMatched[i]);
    for (size_t i = 0; i < vpPoints.size(); i++) pkf[i] = (int32_t)i;          // "keyframe id" = position in vpPointsKFs
    const int n = check(m_.SearchByProjection(K.v, sim3(Scw), P.v, pkf.data(), matched.data(), mkf.data(), th, ratioHamming));
    for (size_t i = 0; i < vpMatched.size(); i++) {
      vpMatched[i] = R.ptr(matched[i]);
      if (mkf[i] >= 0) vpMatchedKF[i] = vpPointsKFs[mkf[i]];
    }
    return n;
  }

  // (ORBmatcher.cc:214-393) relocalisation / loop detection
  int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
    Registry R;
    KFPack K(R, pKF);
    FVPack fv(F.mFeatVec);
    std::vector<int32_t> mp(F.N, -1);
    dvm_host::FrameView V = view(F, mp.data());
    std::vector<int32_t> out(F.N, -1);
    const int n = check(m_.SearchByBoW(K.v, V, fv.v, out.data()));
    vpMapPointMatches.assign(F.N, static_cast<MapPoint*>(NULL));
    for (int i = 0; i < F.N; i++) vpMapPointMatches[i] = R.ptr(out[i]);
    return n;
  }
  // (ORBmatcher.cc:709-834)
  int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
    Registry R;
    KFPack K1(R, pKF1), K2(R, pKF2);
    std::vector<int32_t> out(K1.v.N, -1);
    const int n = check(m_.SearchByBoW(K1.v, K2.v, out.data()));
    vpMatches12.assign(K1.v.N, static_cast<MapPoint*>(NULL));
    for (int i = 0; i < K1.v.N; i++) vpMatches12[i] = R.ptr(out[i]);
    return n;
  }

  // (ORBmatcher.cc:605-707) map initialisation
  int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10) {
    std::vector<int32_t> m1(F1.N, -1), m2(F2.N, -1), out(F1.N, -1);
    dvm_host::FrameView A = view(F1, m1.data()), B = view(F2, m2.data());
    static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "cv::Point2f layout");
    const int n = check(m_.SearchForInitialization(A, B, reinterpret_cast<float*>(vbPrevMatched.data()), out.data(), windowSize));
    vnMatches12.assign(out.begin(), out.end());
    return n;
  }

  // (ORBmatcher.cc:836-1058) new map points
  int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo,
                             const bool bCoarse = false) {
    Registry R;
    KFPack K1(R, pKF1), K2(R, pKF2);
    std::vector<int32_t> pairs(2 * (size_t)std::max(K1.v.N, 1));
    const int n = check(m_.SearchForTriangulation(K1.v, K2.v, pairs.data(), bOnlyStereo, bCoarse));
    vMatchedPairs.clear();
    vMatchedPairs.reserve(n);
    for (int i = 0; i < n; i++) vMatchedPairs.push_back(make_pair((size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]));
    return n;
  }

  // (ORBmatcher.cc:1347-1551)
  int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const Sophus::Sim3f& S12, const float th) {
    Registry R;
    KFPack K1(R, pKF1), K2(R, pKF2);
    PointPack P1(K1.per_keypoint_points()), P2(K2.per_keypoint_points());
    std::vector<int32_t> m12(K1.v.N, -1), idx2(K1.v.N, -1);
    for (int i = 0; i < K1.v.N; i++)
      if (vpMatches12[i]) { m12[i] = R.id(vpMatches12[i]); idx2[i] = std::get<0>(vpMatches12[i]->GetIndexInKeyFrame(pKF2)); }
    const int n = check(m_.SearchBySim3(K1.v, K2.v, P1.v, P2.v, m12.data(), idx2.data(), sim3(S12), th));
    for (int i = 0; i < K1.v.N; i++) vpMatches12[i] = R.ptr(m12[i]);
    return n;
  }

  // (ORBmatcher.cc:1060-1234) the search runs on the device, Replace / AddObservation are replayed here in the reference's order
  int Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th = 3.0, const bool bRight = false) {
    if (bRight) throw std::runtime_error("ORBmatcher::Fuse(bRight): stereo / fisheye pairs are outside the accelerated path");
    Registry R;
    KFPack K(R, pKF);
    PointPack P(R, vpMapPoints);
    std::vector<uint8_t> inKF(vpMapPoints.size(), 0);
    for (size_t i = 0; i < vpMapPoints.size(); i++) inKF[i] = vpMapPoints[i] && !vpMapPoints[i]->isBad() && vpMapPoints[i]->IsInKeyFrame(pKF);
    std::vector<int32_t> best(vpMapPoints.size(), -1);
    check(m_.Fuse(K.v, P.v, inKF.data(), th, best.data()));
    int nFused = 0;
    for (size_t i = 0; i < vpMapPoints.size(); i++) {
      MapPoint* pMP = vpMapPoints[i];
      if (best[i] < 0 || !pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;   // state may have changed through an earlier Replace
      MapPoint* pMPinKF = pKF->GetMapPoint(best[i]);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) {
          if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
          else pMPinKF->Replace(pMP);
        }
      } else {
        pMP->AddObservation(pKF, best[i]);
        pKF->AddMapPoint(pMP, best[i]);
      }
      nFused++;
    }
    return nFused;
  }

  // (ORBmatcher.cc:1236-1345)
  int Fuse(KeyFrame* pKF, Sophus::Sim3f& Scw, const std::vector<MapPoint*>& vpPoints, float th, vector<MapPoint*>& vpReplacePoint) {
    Registry R;
    KFPack K(R, pKF);
    PointPack P(R, vpPoints);
    const std::vector<int32_t> before = K.mp;
    std::vector<int32_t> rep(vpPoints.size(), -1);
    const int n = check(m_.Fuse(K.v, sim3(Scw), P.v, th, rep.data()));
    for (size_t i = 0; i < vpPoints.size(); i++)
      if (rep[i] >= 0) vpReplacePoint[i] = R.ptr(rep[i]);
    for (int j = 0; j < K.v.N; j++)
      if (K.mp[j] != before[j] && K.mp[j] >= 0) {   // a point was added to keypoint j
        MapPoint* pMP = R.ptr(K.mp[j]);
        pMP->AddObservation(pKF, j);
        pKF->AddMapPoint(pMP, j);
      }
    return n;
  }

 public:
  static const int TH_LOW = 50;
  static const int TH_HIGH = 100;
  static const int HISTO_LENGTH = 30;
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW

 protected:
  float RadiusByViewingCos(const float& viewCos) { return dvm_host::ORBmatcher::RadiusByViewingCos(viewCos); }

  float mfNNratio;
  bool mbCheckOrientation;

 private:
  dvm_host::ORBmatcher m_;

  static int check(int n) {
    if (n < 0) throw std::runtime_error(dvm_last_error());
    return n;
  }
  static const dvm_keypoint* kps(const std::vector<cv::KeyPoint>& v) {
    static_assert(sizeof(cv::KeyPoint) == sizeof(dvm_keypoint), "cv::KeyPoint layout");
    return reinterpret_cast<const dvm_keypoint*>(v.data());
  }
  static void copy_desc(uint8_t* dst, MapPoint* p) {
    const cv::Mat d = p->GetDescriptor();
    std::memcpy(dst, d.ptr<uint8_t>(), 32);
  }
  static dvm_se3f se3(const Sophus::SE3f& T) {
    dvm_se3f o;
    for (int i = 0; i < 4; i++) o.q[i] = T.unit_quaternion().coeffs()(i);   // (x, y, z, w)
    for (int i = 0; i < 3; i++) o.t[i] = T.translation()(i);
    return o;
  }
  static dvm_sim3f sim3(const Sophus::Sim3f& S) {
    dvm_sim3f o;
    for (int i = 0; i < 4; i++) o.q[i] = S.rxso3().quaternion().coeffs()(i);   // |q|^2 = scale
    for (int i = 0; i < 3; i++) o.t[i] = S.translation()(i);
    return o;
  }
  static dvm_host::FrameView view(Frame& F, int32_t* mp) {
    dvm_host::FrameView V;
    V.N = F.N; V.mvKeysUn = kps(F.mvKeysUn); V.mDescriptors = F.mDescriptors.data; V.mvpMapPoints = mp;
    V.Tcw = se3(F.GetPose());
    V.fx = F.fx; V.fy = F.fy; V.cx = F.cx; V.cy = F.cy;
    V.mnMinX = F.mnMinX; V.mnMaxX = F.mnMaxX; V.mnMinY = F.mnMinY; V.mnMaxY = F.mnMaxY;
    V.mvScaleFactors = F.mvScaleFactors.data(); V.nLevels = F.mnScaleLevels;
    // the extractor's result is still in HBM and these are its keypoints unchanged (no distortion: mvKeysUn == mvKeys, Frame.cc:792-795)
    V.dev = (F.mDvmDevice.handle_id && (F.mDistCoef.empty() || F.mDistCoef.at<float>(0) == 0.0f)) ? &F.mDvmDevice : nullptr;
    return V;
  }

  // MapPoint* <-> small integer id for the duration of one call (NULL <-> -1)
  struct Registry {
    std::vector<MapPoint*> v;
    std::unordered_map<MapPoint*, int32_t> m;
    int32_t id(MapPoint* p) {
      if (!p) return -1;
      auto it = m.find(p);
      if (it != m.end()) return it->second;
      m[p] = (int32_t)v.size();
      v.push_back(p);
      return (int32_t)v.size() - 1;
    }
    MapPoint* ptr(int32_t i) const { return i < 0 ? static_cast<MapPoint*>(NULL) : v[i]; }
    int size() const { return (int)v.size(); }
  };
  // DBoW2::FeatureVector flattened
  struct FVPack {
    std::vector<int32_t> node, off, feat;
    dvm_host::FeatureVectorView v;
    explicit FVPack(const DBoW2::FeatureVector& fv) {
      off.push_back(0);
      for (const auto& kv : fv) {
        node.push_back((int32_t)kv.first);
        for (unsigned f : kv.second) feat.push_back((int32_t)f);
        off.push_back((int32_t)feat.size());
      }
      v.n = (int)node.size(); v.node = node.data(); v.off = off.data(); v.feat = feat.data();
    }
  };
  // map point data as the projection searches read it
  struct PointData {
    std::vector<int32_t> id;
    std::vector<uint8_t> bad, desc;
    std::vector<float> pos, normal, mind, maxd;
    void push(int32_t i, MapPoint* p) {
      id.push_back(i);
      const bool ok = p != NULL;
      bad.push_back(ok ? (uint8_t)p->isBad() : 1);
      Eigen::Vector3f X, Nn;
      X.setZero(); Nn.setZero();
      float mn = 1.f, mx = 1.f;
      uint8_t d[32] = {0};
      if (ok) { X = p->GetWorldPos(); Nn = p->GetNormal(); mn = p->GetMinDistance(); mx = p->GetMaxDistance(); copy_desc(d, p); }
      for (int k = 0; k < 3; k++) { pos.push_back(X(k)); normal.push_back(Nn(k)); }
      mind.push_back(mn); maxd.push_back(mx);
      desc.insert(desc.end(), d, d + 32);
    }
  };
  struct PointPack {
    PointData d;
    dvm_host::MapPointsView v;
    PointPack(Registry& R, const std::vector<MapPoint*>& pts) {
      for (MapPoint* p : pts) d.push(R.id(p), p);
      bind();
    }
    explicit PointPack(PointData&& pd) : d(std::move(pd)) { bind(); }
    void bind() {
      v.n = (int)d.id.size(); v.id = d.id.data(); v.bad = d.bad.data(); v.pos = d.pos.data(); v.normal = d.normal.data();
      v.min_dist = d.mind.data(); v.max_dist = d.maxd.data(); v.desc = d.desc.data();
    }
  };
  // the members of a KeyFrame the matcher reads
  struct KFPack {
    std::vector<MapPoint*> mps;
    std::vector<int32_t> mp;
    std::vector<uint8_t> bad;
    FVPack fv;
    dvm_host::KeyFrameView v;
    Registry& R;
    KFPack(Registry& R_, KeyFrame* pKF) : mps(pKF->GetMapPointMatches()), fv(pKF->mFeatVec), R(R_) {
      mp.resize(mps.size()); bad.resize(mps.size());
      for (size_t i = 0; i < mps.size(); i++) { mp[i] = R.id(mps[i]); bad[i] = mps[i] ? (uint8_t)mps[i]->isBad() : 0; }
      v.N = pKF->N; v.mvKeysUn = kps(pKF->mvKeysUn); v.mDescriptors = pKF->mDescriptors.data;
      v.mvpMapPoints = mp.data(); v.mpBad = bad.data(); v.mFeatVec = fv.v;
      v.Tcw = se3(pKF->GetPose()); v.Twc = se3(pKF->GetPoseInverse());
      v.fx = pKF->fx; v.fy = pKF->fy; v.cx = pKF->cx; v.cy = pKF->cy;
      v.mnMinX = (float)pKF->mnMinX; v.mnMaxX = (float)pKF->mnMaxX; v.mnMinY = (float)pKF->mnMinY; v.mnMaxY = (float)pKF->mnMaxY;
      v.mvScaleFactors = pKF->mvScaleFactors.data(); v.mvLevelSigma2 = pKF->mvLevelSigma2.data();
      v.mvInvLevelSigma2 = pKF->mvInvLevelSigma2.data(); v.mfLogScaleFactor = pKF->mfLogScaleFactor; v.nLevels = pKF->mnScaleLevels;
    }
    PointData per_keypoint_points() const {   // the map point each keypoint observes (placeholder where there is none)
      PointData d;
      for (size_t i = 0; i < mps.size(); i++) d.push(mp[i], mps[i]);
      return d;
    }
  };
};

}  // namespace ORB_SLAM3
