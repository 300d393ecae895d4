// dvm_slam_amd/host/Sim3Solver_shim.h -- the RANSAC loop of ORB_SLAM3::Sim3Solver (src/Sim3Solver.cc:154-216: both iterate()
// overloads and find()) on the HIP library.  NOTHING ELSE of the class moves: the constructor, SetRansacParameters and the
// getters stay where they are in src/Sim3Solver.cc (they are host bookkeeping on KeyFrame* / MapPoint*); a maintainer deletes
// the bodies of iterate() x2 and find() there and includes this header.  ComputeCentroid / ComputeSim3 /
// CheckInliers / Project are then no longer called: Horn's closed form on three correspondences and the reprojection test of all
// N points in both images (:294-408) run on the device, one wave per hypothesis (dvm_sim3_hypotheses).
//
// What a call does differently from the reference, and why the caller cannot tell: the reference draws ONE minimal set per
// RANSAC iteration and stops drawing at the first hypothesis with more than mRansacMinInliers inliers.  Here all sets a call may
// need (nIterations, or what mRansacMaxIts leaves) are drawn up front with the reference's own DUtils::Random::RandomInt
// procedure (:166-181), all hypotheses are evaluated by one launch, and the outcome is read off the per-hypothesis inlier
// counts: the running best is taken over by every hypothesis that ties or beats it, the first such hypothesis above the
// acceptance bar ends the call -- the rule of :186-203.  Returned transform, inlier flags, mnIterations and the best-so-far
// members are therefore the reference's for those draws; only libc's rand() stream has advanced past the sets behind the
// accepted one (a stream all threads of the reference share and interleave in anyway).
#pragma once
#include <algorithm>
#include <stdexcept>
#include <vector>

#include "KeyFrame.h"
#include "MapPoint.h"
#include "Sim3Solver.h"
#include "Thirdparty/DBoW2/DUtils/Random.h"
#include "dvm_device.h"
#include "dvmslam_hip.h"

namespace ORB_SLAM3 {
namespace dvm_sim3solver_detail {

// the hypotheses of one call as the device returns them: per hypothesis s, R (row-major), t, the inlier count and one flag per point
struct Batch {
  int H = 0, N = 0;
  std::vector<float> T12;          // 13 floats per hypothesis
  std::vector<int32_t> inliers;
  std::vector<uint8_t> mask;       // H x N
  float scale(int h) const { return T12[13 * (size_t)h]; }
  Eigen::Matrix3f rotation(int h) const {
    Eigen::Matrix3f R;
    for (int k = 0; k < 9; k++) R(k / 3, k % 3) = T12[13 * (size_t)h + 1 + k];
    return R;
  }
  Eigen::Vector3f translation(int h) const {
    Eigen::Vector3f t;
    for (int k = 0; k < 3; k++) t(k) = T12[13 * (size_t)h + 10 + k];
    return t;
  }
  Eigen::Matrix4f matrix(int h) const {                       // [sR | t; 0 0 0 1]
    Eigen::Matrix4f M = Eigen::Matrix4f::Identity();
    const Eigen::Matrix3f R = rotation(h);
    const Eigen::Vector3f t = translation(h);
    const float s = scale(h);
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) M(r, c) = s * R(r, c);
      M(r, 3) = t(r);
    }
    return M;
  }
  std::vector<bool> flags(int h) const {
    std::vector<bool> f(N);
    for (int i = 0; i < N; i++) f[i] = mask[(size_t)h * N + i] != 0;
    return f;
  }
};

}  // namespace dvm_sim3solver_detail

inline Eigen::Matrix4f Sim3Solver::iterate(int nIterations, bool& bNoMore, vector<bool>& vbInliers, int& nInliers, bool& bConverge) {
  using dvm_sim3solver_detail::Batch;
  bNoMore = bConverge = false;
  nInliers = 0;
  vbInliers.assign(mN1, false);
  if (N < mRansacMinInliers) {
    bNoMore = true;
    return Eigen::Matrix4f::Identity();
  }
  Batch b;
  b.N = N;
  b.H = std::max(0, std::min(nIterations, mRansacMaxIts - mnIterations));
  Eigen::Matrix4f result = Eigen::Matrix4f::Identity();      // (the reference returns an uninitialised matrix when nothing ties the best)
  if (b.H > 0) {
    // minimal sets: three distinct correspondence indices each, RandomInt over the shrinking pool (:166-181)
    std::vector<int32_t> sets(3 * (size_t)b.H);
    std::vector<size_t> pool;
    for (int32_t* s = sets.data(); s != sets.data() + sets.size(); s += 3) {
      pool = mvAllIndices;
      for (int k = 0; k < 3; k++) {
        const int pick = DUtils::Random::RandomInt(0, pool.size() - 1);
        s[k] = (int32_t)pool[pick];
        pool[pick] = pool.back();
        pool.pop_back();
      }
    }
    // camera-frame points and per-point error bounds as flat arrays; the kernel projects them itself
    std::vector<float> P1(3 * (size_t)N), P2(3 * (size_t)N), bound1(N), bound2(N);
    for (int i = 0; i < N; i++) {
      for (int k = 0; k < 3; k++) {
        P1[3 * (size_t)i + k] = mvX3Dc1[i](k);
        P2[3 * (size_t)i + k] = mvX3Dc2[i](k);
      }
      bound1[i] = (float)mvnMaxError1[i];
      bound2[i] = (float)mvnMaxError2[i];
    }
    float K1[4], K2[4];
    for (int k = 0; k < 4; k++) { K1[k] = pCamera1->getParameter(k); K2[k] = pCamera2->getParameter(k); }
    b.T12.resize(13 * (size_t)b.H);
    b.inliers.resize(b.H);
    b.mask.resize((size_t)b.H * N);
    if (dvm_sim3_hypotheses(dvm_host::device(), P1.data(), P2.data(), bound1.data(), bound2.data(), N, K1, K2, sets.data(), b.H,
                            mbFixScale ? 1 : 0, b.T12.data(), b.inliers.data(), b.mask.data()) != DVM_OK)
      throw std::runtime_error(dvm_last_error());
    // read the outcome off the counts: `champion` = last hypothesis that tied or beat the running best, `accepted` = the first
    // of those above the bar (the call ends there and the hypotheses behind it were never "iterated")
    int champion = -1, accepted = -1, best = mnBestInliers;
    for (int h = 0; h < b.H && accepted < 0; h++) {
      if (b.inliers[h] < best) continue;
      best = b.inliers[h];
      champion = h;
      if (best > mRansacMinInliers) accepted = h;
    }
    const int last = accepted >= 0 ? accepted : b.H - 1;       // the hypothesis the per-iteration members describe on return
    mnIterations += last + 1;
    ms12i = b.scale(last);
    mR12i = b.rotation(last);
    mt12i = b.translation(last);
    mT12i = b.matrix(last);
    mnInliersi = b.inliers[last];
    mvbInliersi = b.flags(last);
    if (champion >= 0) {
      mnBestInliers = best;
      mvbBestInliers = b.flags(champion);
      mBestScale = b.scale(champion);
      mBestRotation = b.rotation(champion);
      mBestTranslation = b.translation(champion);
      mBestT12 = b.matrix(champion);
      result = mBestT12;
    }
    if (accepted >= 0) {
      nInliers = mnBestInliers;
      for (int i = 0; i < N; i++)
        if (mvbBestInliers[i]) vbInliers[mvnIndices1[i]] = true;
      bConverge = true;
      return result;
    }
  }
  bNoMore = mnIterations >= mRansacMaxIts;
  return result;
}

// the four-argument form hands out the identity unless a hypothesis passed (:154-207)
inline Eigen::Matrix4f Sim3Solver::iterate(int nIterations, bool& bNoMore, vector<bool>& vbInliers, int& nInliers) {
  bool converged = false;
  const Eigen::Matrix4f T = iterate(nIterations, bNoMore, vbInliers, nInliers, converged);
  if (!converged) return Eigen::Matrix4f::Identity();
  return T;
}

inline Eigen::Matrix4f Sim3Solver::find(vector<bool>& vbInliers12, int& nInliers) {
  bool exhausted = false;
  return iterate(mRansacMaxIts, exhausted, vbInliers12, nInliers);
}

}  // namespace ORB_SLAM3
