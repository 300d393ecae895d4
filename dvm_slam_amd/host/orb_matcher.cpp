// dvm_slam_amd/host/orb_matcher.cpp -- see orb_matcher.h.  Host C++ (g++), links libdvmslam_hip.so.
#include "orb_matcher.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace dvm_host {

namespace {
const int kCols = 64, kRows = 48;  // FRAME_GRID_COLS / ROWS, Frame.h:44-45

// Frame::GetFeaturesInArea on the host, used only for the rare re-queries (Frame.cc:712-770)
struct HostGrid {
  std::vector<int> cell[kCols][kRows];
  float minX, minY, wInv, hInv;
  const dvm_keypoint* kps;
  void build(const FrameView& F) {
    kps = F.mvKeysUn;
    minX = F.mnMinX; minY = F.mnMinY;
    wInv = static_cast<float>(kCols) / static_cast<float>(F.mnMaxX - F.mnMinX);
    hInv = static_cast<float>(kRows) / static_cast<float>(F.mnMaxY - F.mnMinY);
    for (int i = 0; i < F.N; i++) {
      const int px = (int)std::round((kps[i].x - minX) * wInv), py = (int)std::round((kps[i].y - minY) * hInv);
      if (px < 0 || px >= kCols || py < 0 || py >= kRows) continue;
      cell[px][py].push_back(i);
    }
  }
  void query(float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) const {
    out.clear();
    const int c0 = std::max(0, (int)std::floor((x - minX - r) * wInv));
    if (c0 >= kCols) return;
    const int c1 = std::min(kCols - 1, (int)std::ceil((x - minX + r) * wInv));
    if (c1 < 0) return;
    const int r0 = std::max(0, (int)std::floor((y - minY - r) * hInv));
    if (r0 >= kRows) return;
    const int r1 = std::min(kRows - 1, (int)std::ceil((y - minY + r) * hInv));
    if (r1 < 0) return;
    const bool check = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = c0; ix <= c1; ix++)
      for (int iy = r0; iy <= r1; iy++)
        for (int idx : cell[ix][iy]) {
          const dvm_keypoint& kp = kps[idx];
          if (check) {
            if (kp.octave < minLevel) continue;
            if (maxLevel >= 0 && kp.octave > maxLevel) continue;
          }
          if (std::fabs(kp.x - x) < r && std::fabs(kp.y - y) < r) out.push_back(idx);
        }
  }
};
}  // namespace

ORBmatcher::ORBmatcher(float nnratio, bool checkOri, int device) : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device) {}
ORBmatcher::~ORBmatcher() { if (grid_) dvm_frame_destroy(grid_); }

int ORBmatcher::DescriptorDistance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t x, y;
    std::memcpy(&x, a + 4 * i, 4);
    std::memcpy(&y, b + 4 * i, 4);
    dist += __builtin_popcount(x ^ y);
  }
  return dist;
}

void ORBmatcher::ComputeThreeMaxima(std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) ind3 = -1;
}

int ORBmatcher::SearchByProjection(FrameView& Cur, const FrameView& Last, const MapPointPOD* MPs, float th, bool bMono) {
  (void)bMono;  // DVM-SLAM is monocular (src/slam_system/src/ros_mono.cpp:19): bForward = bBackward = false
  int nmatches = 0;
  last_requeried = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (auto& h : rotHist) h.reserve(500);
  const float factor = 1.0f / HISTO_LENGTH;

  // ---- queries, exactly the loop header of :1573-1611
  std::vector<int> qi;                 // index in LastFrame
  std::vector<float> qx, qy, qr;
  std::vector<int32_t> qmin, qmax;
  std::vector<uint8_t> qdesc;
  for (int i = 0; i < Last.N; i++) {
    const int mp = Last.mvpMapPoints[i];
    if (mp < 0) continue;
    if (Last.mvbOutlier && Last.mvbOutlier[i]) continue;
    const float* X = MPs[mp].pos;
    const float xc = (Cur.Rcw[0] * X[0] + Cur.Rcw[1] * X[1] + Cur.Rcw[2] * X[2]) + Cur.tcw[0];
    const float yc = (Cur.Rcw[3] * X[0] + Cur.Rcw[4] * X[1] + Cur.Rcw[5] * X[2]) + Cur.tcw[1];
    const float zc = (Cur.Rcw[6] * X[0] + Cur.Rcw[7] * X[1] + Cur.Rcw[8] * X[2]) + Cur.tcw[2];
    const float invzc = (float)(1.0 / zc);
    if (invzc < 0) continue;
    const float u = Cur.fx * xc / zc + Cur.cx, v = Cur.fy * yc / zc + Cur.cy;
    if (u < Cur.mnMinX || u > Cur.mnMaxX) continue;
    if (v < Cur.mnMinY || v > Cur.mnMaxY) continue;
    const int nLastOctave = Last.mvKeysUn[i].octave;
    const float radius = th * Cur.mvScaleFactors[nLastOctave];
    qi.push_back(i); qx.push_back(u); qy.push_back(v); qr.push_back(radius);
    qmin.push_back(nLastOctave - 1); qmax.push_back(nLastOctave + 1);
    qdesc.insert(qdesc.end(), MPs[mp].desc, MPs[mp].desc + 32);
  }
  const int nq = (int)qi.size();
  if (nq == 0) return 0;

  // ---- one batched device search against CurrentFrame's grid (claims known at entry are masked)
  int rc = ensure_grid(Cur);
  if (rc != DVM_OK) return rc;
  std::vector<uint8_t> claimed(grid_cap_, 0);
  for (int j = 0; j < Cur.N; j++)
    if (Cur.mvpMapPoints[j] >= 0 && MPs[Cur.mvpMapPoints[j]].n_obs > 0) claimed[j] = 1;
  std::vector<dvm_match> res(nq);
  rc = dvm_match_window(grid_, 0, claimed.data(), qdesc.data(), qx.data(), qy.data(), qr.data(), qmin.data(), qmax.data(), nq,
                        nullptr, res.data(), 0, nullptr);
  if (rc != DVM_OK) return rc;

  // ---- sequential epilogue in query order (:1613-1664): a keypoint claimed by an earlier match of THIS call is
  // skipped by later queries, so a result whose best candidate has been claimed meanwhile is recomputed
  HostGrid hg;
  bool hg_built = false;
  std::vector<uint8_t> claimed_now = claimed;
  std::vector<int> cand;
  for (int q = 0; q < nq; q++) {
    int bestIdx2 = res[q].best_idx, bestDist = res[q].best_dist;
    if (bestIdx2 >= 0 && claimed_now[bestIdx2] && !claimed[bestIdx2]) {
      if (!hg_built) { hg.build(Cur); hg_built = true; }
      last_requeried++;
      hg.query(qx[q], qy[q], qr[q], qmin[q], qmax[q], cand);
      bestDist = 256; bestIdx2 = -1;
      for (int i2 : cand) {
        if (claimed_now[i2]) continue;
        const int dist = DescriptorDistance(&qdesc[32 * (size_t)q], Cur.mDescriptors + 32 * (size_t)i2);
        if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
      }
    }
    if (bestDist <= TH_HIGH) {
      const int i = qi[q];
      const int mp = Last.mvpMapPoints[i];
      Cur.mvpMapPoints[bestIdx2] = mp;
      if (MPs[mp].n_obs > 0) claimed_now[bestIdx2] = 1;
      nmatches++;
      if (mbCheckOrientation) {
        float rot = Last.mvKeysUn[i].angle - Cur.mvKeysUn[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int idx : rotHist[i]) { Cur.mvpMapPoints[idx] = -1; nmatches--; }
  }
  return nmatches;
}

int ORBmatcher::ensure_grid(const FrameView& F) {
  if (!grid_ || grid_cap_ < F.N) {
    if (grid_) dvm_frame_destroy(grid_);
    grid_ = nullptr;
    grid_cap_ = std::max(2048, F.N);
    int rc = dvm_frame_create(device_, grid_cap_, 1, &grid_);
    if (rc != DVM_OK) return rc;
  }
  return dvm_frame_build(grid_, 0, F.mvKeysUn, F.mDescriptors, F.N, nullptr, F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY, 0, nullptr);
}

int ORBmatcher::SearchByProjection(FrameView& F, const TrackedPointPOD* MPs, int nMP, const uint8_t* claimedObs, float th,
                                   bool bFarPoints, float thFarPoints, int mp_index_base) {
  int nmatches = 0;
  last_requeried = 0;
  const bool bFactor = th != 1.0;
  // ---- queries: the loop header of :50-73
  std::vector<int> qi;
  std::vector<float> qx, qy, qr;
  std::vector<int32_t> qmin, qmax;
  std::vector<uint8_t> qdesc;
  for (int i = 0; i < nMP; i++) {
    const TrackedPointPOD& mp = MPs[i];
    if (!mp.mbTrackInView) continue;
    if (bFarPoints && mp.mTrackDepth > thFarPoints) continue;
    if (mp.bad) continue;
    const int nPredictedLevel = mp.mnTrackScaleLevel;
    float r = RadiusByViewingCos(mp.mTrackViewCos);
    if (bFactor) r *= th;
    qi.push_back(i); qx.push_back(mp.mTrackProjX); qy.push_back(mp.mTrackProjY);
    qr.push_back(r * F.mvScaleFactors[nPredictedLevel]);
    qmin.push_back(nPredictedLevel - 1); qmax.push_back(nPredictedLevel);
    qdesc.insert(qdesc.end(), mp.desc, mp.desc + 32);
  }
  const int nq = (int)qi.size();
  if (nq == 0) return 0;
  int rc = ensure_grid(F);
  if (rc != DVM_OK) return rc;
  std::vector<uint8_t> claimed(grid_cap_, 0);
  for (int j = 0; j < F.N; j++)
    if (F.mvpMapPoints[j] >= 0 && claimedObs && claimedObs[j]) claimed[j] = 1;
  std::vector<dvm_match> res(nq);
  rc = dvm_match_window(grid_, 0, claimed.data(), qdesc.data(), qx.data(), qy.data(), qr.data(), qmin.data(), qmax.data(), nq,
                        nullptr, res.data(), 0, nullptr);
  if (rc != DVM_OK) return rc;
  // ---- sequential epilogue (:75-131).  A keypoint claimed earlier in THIS call (by a point with observations) is
  // skipped by later queries and may have been their best OR second-best candidate, so a query whose window contains
  // such a keypoint is recomputed on the host; all others keep the device result.
  HostGrid hg;
  bool hg_built = false;
  std::vector<uint8_t> claimed_now = claimed;
  std::vector<int> fresh;   // keypoints claimed during this call
  std::vector<int> cand;
  for (int q = 0; q < nq; q++) {
    int bestDist = res[q].best_dist, bestDist2 = res[q].second_dist, bestLevel = res[q].best_level, bestLevel2 = res[q].second_level;
    int bestIdx = res[q].best_idx;
    bool touched = false;
    for (int j : fresh) {
      const dvm_keypoint& kp = F.mvKeysUn[j];
      if (kp.octave < qmin[q] || kp.octave > qmax[q]) continue;
      if (std::fabs(kp.x - qx[q]) < qr[q] && std::fabs(kp.y - qy[q]) < qr[q]) { touched = true; break; }
    }
    if (touched) {
      if (!hg_built) { hg.build(F); hg_built = true; }
      last_requeried++;
      hg.query(qx[q], qy[q], qr[q], qmin[q], qmax[q], cand);
      bestDist = 256; bestLevel = -1; bestDist2 = 256; bestLevel2 = -1; bestIdx = -1;
      for (int idx : cand) {
        if (claimed_now[idx]) continue;
        const int dist = DescriptorDistance(&qdesc[32 * (size_t)q], F.mDescriptors + 32 * (size_t)idx);
        if (dist < bestDist) {
          bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F.mvKeysUn[idx].octave; bestIdx = idx;
        } else if (dist < bestDist2) {
          bestLevel2 = F.mvKeysUn[idx].octave; bestDist2 = dist;
        }
      }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
      if (bestLevel != bestLevel2 || bestDist <= mfNNratio * bestDist2) {
        F.mvpMapPoints[bestIdx] = mp_index_base + qi[q];
        if (MPs[qi[q]].n_obs > 0 && !claimed_now[bestIdx]) { claimed_now[bestIdx] = 1; fresh.push_back(bestIdx); }
        nmatches++;
      }
    }
  }
  return nmatches;
}

}  // namespace dvm_host

// ---- C entry point for the Python harness (tests only; a C++ caller uses the class directly)
extern "C" int dvmh_search_by_projection_frames(int device, int Nc, const dvm_keypoint* kps_c, const uint8_t* desc_c,
                                                int32_t* mp_c, const float* Rcw, const float* tcw, const float* K,
                                                const float* bounds, const float* scale_factors, int nlevels, int Nl,
                                                const dvm_keypoint* kps_l, const int32_t* mp_l, const uint8_t* outlier_l,
                                                const dvm_host::MapPointPOD* mps, float th, int check_ori, int* requeried) {
  dvm_host::FrameView C, L;
  C.N = Nc; C.mvKeysUn = kps_c; C.mDescriptors = desc_c; C.mvpMapPoints = mp_c;
  std::memcpy(C.Rcw, Rcw, 36); std::memcpy(C.tcw, tcw, 12);
  C.fx = K[0]; C.fy = K[1]; C.cx = K[2]; C.cy = K[3];
  C.mnMinX = bounds[0]; C.mnMaxX = bounds[1]; C.mnMinY = bounds[2]; C.mnMaxY = bounds[3];
  C.mvScaleFactors = scale_factors; C.nLevels = nlevels;
  L = C;
  L.N = Nl; L.mvKeysUn = kps_l; L.mDescriptors = nullptr; L.mvpMapPoints = const_cast<int32_t*>(mp_l); L.mvbOutlier = outlier_l;
  dvm_host::ORBmatcher m(0.9f, check_ori != 0, device);
  const int n = m.SearchByProjection(C, L, mps, th, true);
  if (requeried) *requeried = m.last_requeried;
  return n;
}

extern "C" int dvmh_search_by_projection_points(int device, int N, const dvm_keypoint* kps, const uint8_t* desc, int32_t* mp,
                                                const uint8_t* claimed_obs, const float* bounds, const float* scale_factors,
                                                int nlevels, const dvm_host::TrackedPointPOD* pts, int npts, float th,
                                                float nnratio, int far_points, float th_far, int* requeried) {
  dvm_host::FrameView F;
  F.N = N; F.mvKeysUn = kps; F.mDescriptors = desc; F.mvpMapPoints = mp;
  F.mnMinX = bounds[0]; F.mnMaxX = bounds[1]; F.mnMinY = bounds[2]; F.mnMaxY = bounds[3];
  F.mvScaleFactors = scale_factors; F.nLevels = nlevels;
  dvm_host::ORBmatcher m(nnratio, true, device);
  const int n = m.SearchByProjection(F, pts, npts, claimed_obs, th, far_points != 0, th_far, 0);
  if (requeried) *requeried = m.last_requeried;
  return n;
}
