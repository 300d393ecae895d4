// dvm_slam_amd/host/orb_matcher.cpp -- see orb_matcher.h.  Host C++ (g++), links libdvmslam_hip.so.
#include "orb_matcher.h"

#include "../csrc/pose_f32.h"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>

namespace dvm_host {

namespace {
const int kCols = 64, kRows = 48;  // FRAME_GRID_COLS / ROWS, Frame.h:44-45

// Frame::GetFeaturesInArea on the host, used only for the rare re-queries (Frame.cc:712-770)
struct HostGrid {
  // the grid of Frame::AssignFeaturesToGrid as two flat arrays (counting sort: a cell's indices stay in insertion order, the
  // order GetFeaturesInArea walks them in).  64 x 48 std::vectors, as the reference keeps it, cost ~0.1 ms to fill and free per
  // call -- as much as the device search this grid only backs up
  std::vector<int32_t> start, idx;     // start[ix * kRows + iy] .. start[ix * kRows + iy + 1] into idx
  float minX, minY, wInv, hInv;
  const dvm_keypoint* kps;
  void build(const FrameView& F) { build(F.mvKeysUn, F.N, F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY); }
  void build(const dvm_keypoint* k, int N, float mnMinX, float mnMaxX, float mnMinY, float mnMaxY) {
    kps = k;
    minX = mnMinX; minY = mnMinY;
    wInv = static_cast<float>(kCols) / static_cast<float>(mnMaxX - mnMinX);
    hInv = static_cast<float>(kRows) / static_cast<float>(mnMaxY - mnMinY);
    start.assign(kCols * kRows + 1, 0);
    std::vector<int32_t> cell_of(N);
    for (int i = 0; i < N; i++) {
      const int px = (int)std::round((kps[i].x - minX) * wInv), py = (int)std::round((kps[i].y - minY) * hInv);
      const bool in = !(px < 0 || px >= kCols || py < 0 || py >= kRows);
      cell_of[i] = in ? px * kRows + py : -1;
      if (in) start[cell_of[i] + 1]++;
    }
    for (int c = 0; c < kCols * kRows; c++) start[c + 1] += start[c];
    idx.resize(start[kCols * kRows]);
    std::vector<int32_t> fill(start.begin(), start.end() - 1);
    for (int i = 0; i < N; i++)
      if (cell_of[i] >= 0) idx[fill[cell_of[i]]++] = i;
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
        for (int32_t p = start[ix * kRows + iy]; p < start[ix * kRows + iy + 1]; p++) {
          const int id = idx[p];
          const dvm_keypoint& kp = kps[id];
          if (check) {
            if (kp.octave < minLevel) continue;
            if (maxLevel >= 0 && kp.octave > maxLevel) continue;
          }
          if (std::fabs(kp.x - x) < r && std::fabs(kp.y - y) < r) out.push_back(id);
        }
  }
};
}  // namespace

static std::atomic<dvm_match_pool*> g_match_pool{nullptr};   // dvmh_set_match_pool
void set_match_pool(dvm_match_pool* pool) { g_match_pool.store(pool, std::memory_order_release); }

ORBmatcher::ORBmatcher(float nnratio, bool checkOri, int device) : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device) {}
// The device grid belongs to the calling THREAD, not to the matcher object: ORB-SLAM3 constructs an ORBmatcher as a local in
// most of its callers (one per tracked frame), and creating / destroying a frame handle is a dozen device allocations.
namespace {
struct GridCache {
  dvm_frame* g = nullptr;
  int cap = 0, device = -1;
  ~GridCache() { if (g) dvm_frame_destroy(g); }
};
GridCache& grid_cache() {
  thread_local GridCache c;
  return c;
}
}  // namespace

ORBmatcher::~ORBmatcher() {}

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

void ORBmatcher::ComputeThreeMaxima(const int* sizes, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = sizes[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) ind3 = -1;
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

// The loop header of SearchByProjection(CurrentFrame, LastFrame) (:1573-1611): LastFrame's map points projected with CurrentFrame's
// pose, the ones in front of the camera and inside the image bounds become window queries.
void BuildFrameQueries(const FrameView& Cur, const FrameView& Last, const MapPointPOD* MPs, float th, FrameQueries& Q) {
  std::vector<int>& qi = Q.qi;
  std::vector<float>&qx = Q.qx, &qy = Q.qy, &qr = Q.qr;
  std::vector<int32_t>&qmin = Q.qmin, &qmax = Q.qmax;
  std::vector<uint8_t>& qdesc = Q.qdesc;
  qi.clear(); qx.clear(); qy.clear(); qr.clear(); qmin.clear(); qmax.clear(); qdesc.clear();
  qi.reserve(Last.N); qx.reserve(Last.N); qy.reserve(Last.N); qr.reserve(Last.N); qmin.reserve(Last.N); qmax.reserve(Last.N);
  qdesc.reserve((size_t)Last.N * 32);
  for (int i = 0; i < Last.N; i++) {
    const int mp = Last.mvpMapPoints[i];
    if (mp < 0) continue;
    if (Last.mvbOutlier && Last.mvbOutlier[i]) continue;
    const float* X = MPs[mp].pos;
    float x3Dc[3];   // Tcw * x3Dw: Sophus' quaternion action (:1577, so3.hpp:356-367)
    dvm_pose::se3_apply(Cur.Tcw.q, Cur.Tcw.t, X, x3Dc);
    const float xc = x3Dc[0], yc = x3Dc[1], zc = x3Dc[2];
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
}

int ORBmatcher::SearchByProjection(FrameView& Cur, const FrameView& Last, const MapPointPOD* MPs, float th, bool bMono) {
  const auto T0 = std::chrono::steady_clock::now();
  const bool dbg = std::getenv("DVM_HOST_DEBUG_TIMING") != nullptr;
  auto mark = [&](const char* w) { if (dbg) std::fprintf(stderr, "SBP %-14s %8.3f ms\n", w, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count()); };
  (void)bMono;  // DVM-SLAM is monocular (src/slam_system/src/ros_mono.cpp:19): bForward = bBackward = false
  int nmatches = 0;
  last_requeried = 0;
  // rotHist[bin] of :1566-1568 as (keypoint, bin) pairs + counts: the 30 vectors only ever feed ComputeThreeMaxima's sizes and the
  // final sweep over the losing bins
  std::vector<std::pair<int, int>> rotPairs;
  int rotCount[HISTO_LENGTH] = {0};
  const float factor = 1.0f / HISTO_LENGTH;

  // ---- queries, exactly the loop header of :1573-1611
  FrameQueries Q;
  BuildFrameQueries(Cur, Last, MPs, th, Q);
  std::vector<int>& qi = Q.qi;
  std::vector<float>&qx = Q.qx, &qy = Q.qy, &qr = Q.qr;
  std::vector<int32_t>&qmin = Q.qmin, &qmax = Q.qmax;
  std::vector<uint8_t>& qdesc = Q.qdesc;
  const int nq = (int)qi.size();
  if (nq == 0) return 0;

  // ---- one batched device search against CurrentFrame's grid (claims known at entry are masked)
  mark("queries");
  // the grid of CurrentFrame and the search over it are ONE chain on the calling thread's stream (this is the per-frame call of
  // Tracking: two staged calls with a synchronisation each cost more than the two kernels)
  int rc = ensure_handle(Cur.N);
  if (rc != DVM_OK) return rc;
  std::vector<uint8_t> claimed(grid_cap_, 0);
  bool any_claimed = false;
  for (int j = 0; j < Cur.N; j++)
    if (Cur.mvpMapPoints[j] >= 0 && MPs[Cur.mvpMapPoints[j]].n_obs > 0) { claimed[j] = 1; any_claimed = true; }
  // (TrackWithMotionModel clears CurrentFrame.mvpMapPoints before it searches, Tracking.cc:2606: nothing is claimed at entry, and the
  // call then queues no copy at all -- every other array is read in place)
  const uint8_t* skip = any_claimed ? claimed.data() : nullptr;
  std::vector<uint32_t> ranked((size_t)nq * 4);   // the four best candidates per query: dist << 16 | index, best first
  // keypoints + descriptors still in HBM where ORBextractor::operator() left them (Frame.cc:411 -> here): the grid is built from there
  const bool res = resident(Cur);
  last_grid_from_device = res;
  dvm_match_pool* pool = g_match_pool.load(std::memory_order_acquire);
  int pool_kp = 0, pool_q = 0;
  if (pool && !res) dvm_match_pool_capacity(pool, &pool_kp, &pool_q);
  if (pool && !res && Cur.N <= pool_kp && nq <= pool_q)   // several agents on this GPU: the search rides in the shared service's batch
    rc = dvm_match_pool_build_match_ranked(pool, Cur.mvKeysUn, Cur.mDescriptors, Cur.N, Cur.mnMinX, Cur.mnMaxX, Cur.mnMinY, Cur.mnMaxY, skip, qdesc.data(),
                                           qx.data(), qy.data(), qr.data(), qmin.data(), qmax.data(), nq, ranked.data(), nullptr);
  else
    rc = dvm_frame_build_match_window_ranked(grid_, 0, res ? Cur.dev->d_kps : Cur.mvKeysUn, res ? Cur.dev->d_desc : Cur.mDescriptors, Cur.N,
                                             Cur.mnMinX, Cur.mnMaxX, Cur.mnMinY, Cur.mnMaxY, skip, qdesc.data(), qx.data(), qy.data(),
                                             qr.data(), qmin.data(), qmax.data(), nq, ranked.data(), res ? 1 : 0);
  if (rc != DVM_OK) return rc;
  mark("match");

  // ---- sequential epilogue in query order (:1613-1664): a keypoint claimed by an earlier match of THIS call is
  // skipped by later queries, so a result whose best candidate has been claimed meanwhile is recomputed
  HostGrid hg;
  bool hg_built = false;
  double requery_ms = 0;
  std::vector<uint8_t> claimed_now = claimed;
  std::vector<int> cand;
  for (int q = 0; q < nq; q++) {
    // the scan of :1613-1650 skips keypoints an earlier query of this call has taken and keeps the smallest (distance, position) of
    // the rest: the first entry of the ranked list that is still free.  Only when all four are taken (and the list may go on) is the
    // window searched again here.
    int bestIdx2 = -1, bestDist = 256;
    bool exhausted = true;
    for (int c = 0; c < 4; c++) {
      const uint32_t key = ranked[(size_t)q * 4 + c];
      const int dist = (int)(key >> 16);
      if (dist >= 256) { exhausted = false; break; }   // end of the list: nothing else in the window
      const int idx = (int)(key & 0xFFFFu);
      if (!claimed_now[idx]) { bestIdx2 = idx; bestDist = dist; exhausted = false; break; }
    }
    if (exhausted) {
      const auto tq0 = dbg ? std::chrono::steady_clock::now() : T0;
      if (!hg_built) { hg.build(Cur); hg_built = true; }
      last_requeried++;
      hg.query(qx[q], qy[q], qr[q], qmin[q], qmax[q], cand);
      bestDist = 256; bestIdx2 = -1;
      for (int i2 : cand) {
        if (claimed_now[i2]) continue;
        const int dist = DescriptorDistance(&qdesc[32 * (size_t)q], Cur.mDescriptors + 32 * (size_t)i2);
        if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
      }
      if (dbg) requery_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count();
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
        rotPairs.emplace_back(bestIdx2, bin);
        rotCount[bin]++;
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotCount, HISTO_LENGTH, ind1, ind2, ind3);
    for (const auto& pr : rotPairs)
      if (pr.second != ind1 && pr.second != ind2 && pr.second != ind3) { Cur.mvpMapPoints[pr.first] = -1; nmatches--; }
  }
  mark("epilogue");
  if (dbg) std::fprintf(stderr, "SBP queries %d matches %d re-queried %d (%.3f ms in the host re-queries)\n", nq, nmatches, last_requeried, requery_ms);
  return nmatches;
}

int ORBmatcher::ensure_handle(int N) {
  GridCache& c = grid_cache();
  if (!c.g || c.cap < N || c.device != device_) {
    if (c.g) dvm_frame_destroy(c.g);
    c.g = nullptr;
    c.cap = std::max(2048, N);
    c.device = device_;
    int rc = dvm_frame_create(device_, c.cap, 1, &c.g);
    if (rc != DVM_OK) { c.cap = 0; return rc; }
  }
  grid_ = c.g;
  grid_cap_ = c.cap;
  return DVM_OK;
}
bool ORBmatcher::resident(const FrameView& F) const {
  return F.dev && F.dev->n == F.N && F.dev->device == device_ && dvm_device_frame_valid(F.dev);
}

int ORBmatcher::ensure_grid(const FrameView& F) {
  int rc0 = ensure_handle(F.N);
  if (rc0 != DVM_OK) return rc0;
  // the frame's keypoints + descriptors are still in HBM where ORBextractor::operator() left them (Frame.cc:411 -> ORBmatcher.cc:1553
  // without a round trip through the host): the grid is built from there, on the default stream the searches follow on
  if (resident(F)) {
    last_grid_from_device = true;
    return dvm_frame_build(grid_, 0, F.dev->d_kps, F.dev->d_desc, F.N, nullptr, F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY, 1, nullptr);
  }
  last_grid_from_device = false;
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

// --------------------------------------------------------------------------------------------------------------------
// the remaining whole functions of ORBmatcher.  Pattern: everything that is a descriptor
// distance runs on the device in ONE batched call (Hamming table / candidate-list search / projection + window search /
// triangulation search); the reference's sequential bookkeeping (claims, mutual best, rotation histogram) is replayed
// on the host in the reference's order, and a query whose device result was computed with a candidate that an earlier
// query claimed meanwhile is re-evaluated over its (short) candidate list.
// --------------------------------------------------------------------------------------------------------------------
namespace {
int RotBin(float a1, float a2) {
  const float factor = 1.0f / ORBmatcher::HISTO_LENGTH;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * factor);
  if (bin == ORBmatcher::HISTO_LENGTH) bin = 0;
  return bin;
}
int LowerBound(const FeatureVectorView& fv, int from, int key) { return (int)(std::lower_bound(fv.node + from, fv.node + fv.n, key) - fv.node); }

// Walk two FeatureVectors like the while loops of :232-364 / :735-818 / :890-1031 and call f(a, b) for every common node.
template <class Fn>
void ForEachCommonNode(const FeatureVectorView& A, const FeatureVectorView& B, Fn f) {
  int a = 0, b = 0;
  while (a < A.n && b < B.n) {
    if (A.node[a] == B.node[b]) { f(a, b); a++; b++; }
    else if (A.node[a] < B.node[b]) a = LowerBound(A, a, B.node[b]);
    else b = LowerBound(B, b, A.node[a]);
  }
}
}  // namespace

int ORBmatcher::SearchForInitialization(const FrameView& F1, const FrameView& F2, float* vbPrevMatched, int32_t* vnMatches12,
                                        int windowSize) {
  int nmatches = 0;
  for (int i = 0; i < F1.N; i++) vnMatches12[i] = -1;
  // level-0 keypoints of both frames (level1 > 0 -> continue; GetFeaturesInArea(.., level1, level1) keeps octave 0 only)
  std::vector<int> rows, col_of(F2.N, -1);
  std::vector<uint8_t> d1, d2;
  for (int i = 0; i < F1.N; i++)
    if (F1.mvKeysUn[i].octave <= 0) { rows.push_back(i); d1.insert(d1.end(), F1.mDescriptors + 32 * (size_t)i, F1.mDescriptors + 32 * (size_t)i + 32); }
  int ncol = 0;
  for (int j = 0; j < F2.N; j++)
    if (F2.mvKeysUn[j].octave == 0) { col_of[j] = ncol++; d2.insert(d2.end(), F2.mDescriptors + 32 * (size_t)j, F2.mDescriptors + 32 * (size_t)j + 32); }
  if (rows.empty() || ncol == 0) return 0;
  std::vector<uint16_t> D((size_t)rows.size() * ncol);
  if (int rcd = dvm_set_device(device_)) return rcd;   // stateless entry point: runs on the calling thread's current device
  int rc = dvm_hamming_matrix(d1.data(), (int)rows.size(), d2.data(), ncol, D.data(), 0, nullptr);
  if (rc != DVM_OK) return rc;

  std::vector<int> rotHist[HISTO_LENGTH];
  for (auto& h : rotHist) h.reserve(500);
  std::vector<int> vMatchedDistance(F2.N, INT32_MAX), vnMatches21(F2.N, -1);
  HostGrid hg;
  hg.build(F2);
  std::vector<int> vIndices2;
  for (size_t r = 0; r < rows.size(); r++) {
    const int i1 = rows[r];
    const int level1 = F1.mvKeysUn[i1].octave;
    hg.query(vbPrevMatched[2 * i1], vbPrevMatched[2 * i1 + 1], (float)windowSize, level1, level1, vIndices2);
    if (vIndices2.empty()) continue;
    const uint16_t* Drow = &D[r * (size_t)ncol];
    int bestDist = INT32_MAX, bestDist2 = INT32_MAX, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      const int dist = Drow[col_of[i2]];
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * mfNNratio) {
        if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        vnMatches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (mbCheckOrientation) rotHist[RotBin(F1.mvKeysUn[i1].angle, F2.mvKeysUn[bestIdx2].angle)].push_back(i1);
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx1 : rotHist[i])
        if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
    }
  }
  for (int i1 = 0; i1 < F1.N; i1++)
    if (vnMatches12[i1] >= 0) { vbPrevMatched[2 * i1] = F2.mvKeysUn[vnMatches12[i1]].x; vbPrevMatched[2 * i1 + 1] = F2.mvKeysUn[vnMatches12[i1]].y; }
  return nmatches;
}

int ORBmatcher::SearchByBoW(const KeyFrameView& KF, const FrameView& F, const FeatureVectorView& Ffv, int32_t* vpMapPointMatches) {
  for (int i = 0; i < F.N; i++) vpMapPointMatches[i] = -1;
  last_requeried = 0;
  // queries in walk order; the candidate list of a query is its node's feature list in F (CSR into Ffv.feat)
  std::vector<int> qkf, qnodeF;
  std::vector<int32_t> off(1, 0), cand;
  std::vector<uint8_t> qdesc;
  ForEachCommonNode(KF.mFeatVec, Ffv, [&](int a, int b) {
    for (int k = KF.mFeatVec.off[a]; k < KF.mFeatVec.off[a + 1]; k++) {
      const int realIdxKF = KF.mFeatVec.feat[k];
      if (KF.mvpMapPoints[realIdxKF] < 0) continue;
      if (KF.mpBad && KF.mpBad[realIdxKF]) continue;
      qkf.push_back(realIdxKF); qnodeF.push_back(b);
      qdesc.insert(qdesc.end(), KF.mDescriptors + 32 * (size_t)realIdxKF, KF.mDescriptors + 32 * (size_t)realIdxKF + 32);
      cand.insert(cand.end(), Ffv.feat + Ffv.off[b], Ffv.feat + Ffv.off[b + 1]);
      off.push_back((int32_t)cand.size());
    }
  });
  const int nq = (int)qkf.size();
  if (nq == 0) return 0;
  if (cand.empty()) cand.push_back(-1);   // every common node is empty on the other side: nothing to scan, not an error
  std::vector<dvm_match> res(nq);
  if (int rcd = dvm_set_device(device_)) return rcd;   // stateless entry point: runs on the calling thread's current device
  int rc = dvm_match_lists(F.mDescriptors, F.N, qdesc.data(), nq, off.data(), cand.data(), res.data(), 0, nullptr);
  if (rc != DVM_OK) return rc;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (auto& h : rotHist) h.reserve(500);
  for (int q = 0; q < nq; q++) {
    int bestDist1 = res[q].best_dist, bestDist2 = res[q].second_dist, bestIdxF = res[q].best_idx;
    bool touched = false;
    for (int p = off[q]; p < off[q + 1] && !touched; p++) touched = vpMapPointMatches[cand[p]] >= 0;
    if (touched) {   // a feature of this node was matched by an earlier query: it is skipped by this one (:265-266)
      last_requeried++;
      bestDist1 = 256; bestDist2 = 256; bestIdxF = -1;
      for (int p = off[q]; p < off[q + 1]; p++) {
        const int realIdxF = cand[p];
        if (vpMapPointMatches[realIdxF] >= 0) continue;
        const int dist = DescriptorDistance(&qdesc[32 * (size_t)q], F.mDescriptors + 32 * (size_t)realIdxF);
        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
        else if (dist < bestDist2) bestDist2 = dist;
      }
    }
    if (bestDist1 <= TH_LOW) {
      if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
        vpMapPointMatches[bestIdxF] = KF.mvpMapPoints[qkf[q]];
        if (mbCheckOrientation) rotHist[RotBin(KF.mvKeysUn[qkf[q]].angle, F.mvKeysUn[bestIdxF].angle)].push_back(bestIdxF);
        nmatches++;
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx : rotHist[i]) { vpMapPointMatches[idx] = -1; nmatches--; }
    }
  }
  return nmatches;
}

int ORBmatcher::SearchByBoW(const KeyFrameView& KF1, const KeyFrameView& KF2, int32_t* vpMatches12) {
  for (int i = 0; i < KF1.N; i++) vpMatches12[i] = -1;
  last_requeried = 0;
  std::vector<int> q1;
  std::vector<int32_t> off(1, 0), cand;
  std::vector<uint8_t> qdesc;
  ForEachCommonNode(KF1.mFeatVec, KF2.mFeatVec, [&](int a, int b) {
    for (int k = KF1.mFeatVec.off[a]; k < KF1.mFeatVec.off[a + 1]; k++) {
      const int idx1 = KF1.mFeatVec.feat[k];
      if (KF1.mvpMapPoints[idx1] < 0) continue;
      if (KF1.mpBad && KF1.mpBad[idx1]) continue;
      q1.push_back(idx1);
      qdesc.insert(qdesc.end(), KF1.mDescriptors + 32 * (size_t)idx1, KF1.mDescriptors + 32 * (size_t)idx1 + 32);
      for (int k2 = KF2.mFeatVec.off[b]; k2 < KF2.mFeatVec.off[b + 1]; k2++) {
        const int idx2 = KF2.mFeatVec.feat[k2];
        if (KF2.mvpMapPoints[idx2] < 0) continue;            // !pMP2
        if (KF2.mpBad && KF2.mpBad[idx2]) continue;           // pMP2->isBad()
        cand.push_back(idx2);
      }
      off.push_back((int32_t)cand.size());
    }
  });
  const int nq = (int)q1.size();
  if (nq == 0) return 0;
  if (cand.empty()) cand.push_back(-1);
  std::vector<dvm_match> res(nq);
  if (int rcd = dvm_set_device(device_)) return rcd;   // stateless entry point: runs on the calling thread's current device
  int rc = dvm_match_lists(KF2.mDescriptors, KF2.N, qdesc.data(), nq, off.data(), cand.data(), res.data(), 0, nullptr);
  if (rc != DVM_OK) return rc;
  int nmatches = 0;
  std::vector<uint8_t> vbMatched2(KF2.N, 0);
  std::vector<int> rotHist[HISTO_LENGTH];
  for (auto& h : rotHist) h.reserve(500);
  for (int q = 0; q < nq; q++) {
    int bestDist1 = res[q].best_dist, bestDist2 = res[q].second_dist, bestIdx2 = res[q].best_idx;
    bool touched = false;
    for (int p = off[q]; p < off[q + 1] && !touched; p++) touched = vbMatched2[cand[p]] != 0;
    if (touched) {
      last_requeried++;
      bestDist1 = 256; bestDist2 = 256; bestIdx2 = -1;
      for (int p = off[q]; p < off[q + 1]; p++) {
        const int idx2 = cand[p];
        if (vbMatched2[idx2]) continue;
        const int dist = DescriptorDistance(&qdesc[32 * (size_t)q], KF2.mDescriptors + 32 * (size_t)idx2);
        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
        else if (dist < bestDist2) bestDist2 = dist;
      }
    }
    if (bestDist1 < TH_LOW) {
      if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
        vpMatches12[q1[q]] = KF2.mvpMapPoints[bestIdx2];
        vbMatched2[bestIdx2] = 1;
        if (mbCheckOrientation) rotHist[RotBin(KF1.mvKeysUn[q1[q]].angle, KF2.mvKeysUn[bestIdx2].angle)].push_back(q1[q]);
        nmatches++;
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx : rotHist[i]) { vpMatches12[idx] = -1; nmatches--; }
    }
  }
  return nmatches;
}

void ORBmatcher::TriangulationGeometry(const KeyFrameView& KF1, const KeyFrameView& KF2, float* R12, float* t12, float* ep, float* F12) {
  // Cw = pKF1->GetCameraCenter(); C2 = T2w * Cw; ep = pKF2->mpCamera->project(C2)   (:842-848)
  float C2[3];
  dvm_pose::se3_apply(KF2.Tcw.q, KF2.Tcw.t, KF1.Twc.t, C2);
  ep[0] = KF2.fx * C2[0] / C2[2] + KF2.cx;
  ep[1] = KF2.fy * C2[1] / C2[2] + KF2.cy;
  // T12 = T1w * Tw2; R12 = T12.rotationMatrix(); t12 = T12.translation()   (:856-859)
  float q12[4];
  dvm_pose::se3_compose(KF1.Tcw.q, KF1.Tcw.t, KF2.Twc.q, KF2.Twc.t, q12, t12);
  dvm_pose::quat_matrix(q12, R12);
  // F12 = K1.transpose().inverse() * t12x * R12 * K2.inverse()   (Pinhole.cpp:106-110; Eigen cofactor inverse, products left to right)
  const float t12x[9] = {0.f, -t12[2], t12[1], t12[2], 0.f, -t12[0], -t12[1], t12[0], 0.f};   // SO3f::hat
  const float K1T[9] = {KF1.fx, 0.f, 0.f, 0.f, KF1.fy, 0.f, KF1.cx, KF1.cy, 1.f};
  const float K2[9] = {KF2.fx, 0.f, KF2.cx, 0.f, KF2.fy, KF2.cy, 0.f, 0.f, 1.f};
  float K1Tinv[9], K2inv[9], A[9], B[9];
  dvm_pose::mat3_inverse(K1T, K1Tinv);
  dvm_pose::mat3_inverse(K2, K2inv);
  dvm_pose::mat3_mul(K1Tinv, t12x, A);
  dvm_pose::mat3_mul(A, R12, B);
  dvm_pose::mat3_mul(B, K2inv, F12);
}

int ORBmatcher::SearchForTriangulation(const KeyFrameView& KF1, const KeyFrameView& KF2, int32_t* vMatchedPairs, bool bOnlyStereo,
                                       bool bCoarse) {
  if (bOnlyStereo) return 0;   // monocular keyframes have no stereo keypoints (mvuRight < 0): every idx1 is skipped (:896-898)
  float R12[9], t12[3], ep[2], F12[9];
  TriangulationGeometry(KF1, KF2, R12, t12, ep, F12);
  std::vector<int32_t> qidx, off(1, 0), cand;
  ForEachCommonNode(KF1.mFeatVec, KF2.mFeatVec, [&](int a, int b) {
    for (int k = KF1.mFeatVec.off[a]; k < KF1.mFeatVec.off[a + 1]; k++) {
      const int idx1 = KF1.mFeatVec.feat[k];
      if (KF1.mvpMapPoints[idx1] >= 0) continue;   // already a MapPoint
      qidx.push_back(idx1);
      for (int k2 = KF2.mFeatVec.off[b]; k2 < KF2.mFeatVec.off[b + 1]; k2++) {
        const int idx2 = KF2.mFeatVec.feat[k2];
        if (KF2.mvpMapPoints[idx2] >= 0) continue;
        cand.push_back(idx2);
      }
      off.push_back((int32_t)cand.size());
    }
  });
  const int nq = (int)qidx.size();
  if (nq == 0) return 0;
  if (cand.empty()) cand.push_back(-1);
  std::vector<int32_t> bi(nq), bd(nq);
  if (int rcd = dvm_set_device(device_)) return rcd;   // stateless entry point: runs on the calling thread's current device
  int rc = dvm_match_triangulation(KF1.mDescriptors, KF1.mvKeysUn, KF1.N, qidx.data(), nq, KF2.mDescriptors, KF2.mvKeysUn, KF2.N,
                                   off.data(), cand.data(), F12, ep, bCoarse ? 1 : 0, KF2.mvScaleFactors, KF2.mvLevelSigma2, KF2.nLevels,
                                   bi.data(), bd.data(), 0, nullptr);
  if (rc != DVM_OK) return rc;
  int nmatches = 0;
  std::vector<int> vMatches12(KF1.N, -1);
  std::vector<int> rotHist[HISTO_LENGTH];
  for (auto& h : rotHist) h.reserve(500);
  for (int q = 0; q < nq; q++) {
    if (bi[q] < 0) continue;
    vMatches12[qidx[q]] = bi[q];
    nmatches++;
    if (mbCheckOrientation) rotHist[RotBin(KF1.mvKeysUn[qidx[q]].angle, KF2.mvKeysUn[bi[q]].angle)].push_back(qidx[q]);
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx : rotHist[i]) { vMatches12[idx] = -1; nmatches--; }
    }
  }
  int k = 0;
  for (int i = 0; i < KF1.N; i++)
    if (vMatches12[i] >= 0) { vMatchedPairs[2 * k] = i; vMatchedPairs[2 * k + 1] = vMatches12[i]; k++; }
  return nmatches;
}

int ORBmatcher::ensure_grid(const KeyFrameView& KF) {
  FrameView F;
  F.N = KF.N; F.mvKeysUn = KF.mvKeysUn; F.mDescriptors = KF.mDescriptors;
  F.mnMinX = KF.mnMinX; F.mnMaxX = KF.mnMaxX; F.mnMinY = KF.mnMinY; F.mnMaxY = KF.mnMaxY;
  return ensure_grid(F);
}

int ORBmatcher::project_search(const KeyFrameView& KF, const dvm_se3f& Tcw, const float* Ow, const MapPointsView& P,
                               const uint8_t* valid, const uint8_t* skip, float th, bool gate, std::vector<dvm_match>& res,
                               std::vector<dvm_projection>& proj) {
  int rc = ensure_grid(KF);
  if (rc != DVM_OK) return rc;
  dvm_kf_camera cam;
  std::memset(&cam, 0, sizeof(cam));
  cam.Tcw = Tcw; std::memcpy(cam.Ow, Ow, 12);
  cam.fx = KF.fx; cam.fy = KF.fy; cam.cx = KF.cx; cam.cy = KF.cy;
  cam.min_x = KF.mnMinX; cam.max_x = KF.mnMaxX; cam.min_y = KF.mnMinY; cam.max_y = KF.mnMaxY;
  cam.log_scale_factor = KF.mfLogScaleFactor; cam.n_levels = KF.nLevels;
  res.resize(P.n); proj.resize(P.n);
  std::vector<uint8_t> skip_cap;
  if (skip) { skip_cap.assign(grid_cap_, 0); std::memcpy(skip_cap.data(), skip, KF.N); }
  return dvm_project_search(grid_, 0, skip ? skip_cap.data() : nullptr, &cam, P.pos, P.normal, P.min_dist, P.max_dist, P.desc, valid, P.n,
                            th, KF.mvScaleFactors, gate ? KF.mvInvLevelSigma2 : nullptr, 5.99, res.data(), proj.data(), 0, nullptr);
}

int ORBmatcher::Fuse(const KeyFrameView& KF, const MapPointsView& P, const uint8_t* inKF, float th, int32_t* vBestIdx) {
  if (P.n == 0) return 0;
  std::vector<uint8_t> valid(P.n, 1);
  for (int i = 0; i < P.n; i++)
    if ((P.id && P.id[i] < 0) || (P.bad && P.bad[i]) || (inKF && inKF[i])) valid[i] = 0;   // !pMP, isBad(), IsInKeyFrame(pKF)
  std::vector<dvm_match> res;
  std::vector<dvm_projection> proj;
  int rc = project_search(KF, KF.Tcw, KF.Twc.t, P, valid.data(), nullptr, th, true, res, proj);
  if (rc != DVM_OK) return rc;
  int nFused = 0;
  for (int i = 0; i < P.n; i++) {
    vBestIdx[i] = -1;
    if (valid[i] && res[i].best_idx >= 0 && res[i].best_dist <= TH_LOW) { vBestIdx[i] = res[i].best_idx; nFused++; }
  }
  return nFused;
}

void KeyFrameView::SetPose(const dvm_se3f& T) { Tcw = T; Twc = InverseSE3(T); }
dvm_se3f InverseSE3(const dvm_se3f& T) {
  dvm_se3f r;
  dvm_pose::se3_inverse(T.q, T.t, r.q, r.t);
  return r;
}
void PoseMatrices(const dvm_se3f& Tcw, float* Rcw, float* tcw, float* Ow) {
  const dvm_se3f Twc = InverseSE3(Tcw);
  dvm_pose::quat_matrix(Tcw.q, Rcw);
  std::memcpy(tcw, Tcw.t, 12);
  std::memcpy(Ow, Twc.t, 12);
}
void Sim3ToSE3(const dvm_sim3f& Scw, dvm_se3f& Tcw, float* Ow) { dvm_pose::sim3_decompose(Scw.q, Scw.t, Tcw.q, Tcw.t, Ow); }

int ORBmatcher::Fuse(KeyFrameView& KF, const Sim3View& Scw, const MapPointsView& P, float th, int32_t* vpReplacePoint) {
  if (P.n == 0) return 0;
  dvm_se3f Tcw;
  float Ow[3];
  Sim3ToSE3(Scw, Tcw, Ow);
  std::vector<int32_t> already(KF.mvpMapPoints, KF.mvpMapPoints + KF.N);   // spAlreadyFound = pKF->GetMapPoints()
  std::sort(already.begin(), already.end());
  std::vector<uint8_t> valid(P.n, 1);
  for (int i = 0; i < P.n; i++) {
    vpReplacePoint[i] = -1;
    if ((P.bad && P.bad[i]) || std::binary_search(already.begin(), already.end(), P.id[i])) valid[i] = 0;
  }
  std::vector<dvm_match> res;
  std::vector<dvm_projection> proj;
  int rc = project_search(KF, Tcw, Ow, P, valid.data(), nullptr, th, false, res, proj);
  if (rc != DVM_OK) return rc;
  int nFused = 0;
  std::vector<uint8_t> fresh(KF.N, 0);
  for (int i = 0; i < P.n; i++) {
    if (!valid[i] || res[i].best_idx < 0 || res[i].best_dist > TH_LOW) continue;
    const int bestIdx = res[i].best_idx;
    const int pMPinKF = KF.mvpMapPoints[bestIdx];
    if (pMPinKF >= 0) {
      if (fresh[bestIdx] || !(KF.mpBad && KF.mpBad[bestIdx])) vpReplacePoint[i] = pMPinKF;
    } else {
      KF.mvpMapPoints[bestIdx] = P.id[i];   // pMP->AddObservation(pKF, bestIdx); pKF->AddMapPoint(pMP, bestIdx)
      fresh[bestIdx] = 1;
    }
    nFused++;
  }
  return nFused;
}

int ORBmatcher::SearchByProjection(const KeyFrameView& KF, const Sim3View& Scw, const MapPointsView& P, int32_t* vpMatched, int th,
                                   float ratioHamming) {
  return SearchByProjection(KF, Scw, P, nullptr, vpMatched, nullptr, th, ratioHamming);
}

int ORBmatcher::SearchByProjection(const KeyFrameView& KF, const Sim3View& Scw, const MapPointsView& P, const int32_t* vpPointsKFs,
                                   int32_t* vpMatched, int32_t* vpMatchedKF, int th, float ratioHamming) {
  if (P.n == 0) return 0;
  last_requeried = 0;
  dvm_se3f Tcw;
  float Ow[3];
  Sim3ToSE3(Scw, Tcw, Ow);
  std::vector<int32_t> already(vpMatched, vpMatched + KF.N);   // spAlreadyFound (fixed at entry)
  std::sort(already.begin(), already.end());
  std::vector<uint8_t> valid(P.n, 1), skip(KF.N, 0);
  for (int i = 0; i < P.n; i++)
    if ((P.bad && P.bad[i]) || (P.id[i] >= 0 && std::binary_search(already.begin(), already.end(), P.id[i]))) valid[i] = 0;
  for (int j = 0; j < KF.N; j++) skip[j] = vpMatched[j] >= 0;
  std::vector<dvm_match> res;
  std::vector<dvm_projection> proj;
  int rc = project_search(KF, Tcw, Ow, P, valid.data(), skip.data(), (float)th, false, res, proj);
  if (rc != DVM_OK) return rc;
  int nmatches = 0;
  HostGrid hg;
  bool hg_built = false;
  std::vector<int> cand;
  for (int i = 0; i < P.n; i++) {
    if (!valid[i] || proj[i].level < 0) continue;
    int bestIdx = res[i].best_idx, bestDist = res[i].best_dist;
    if (bestIdx >= 0 && vpMatched[bestIdx] >= 0) {   // claimed by an earlier point of this call: search again without it
      if (!hg_built) { hg.build(KF.mvKeysUn, KF.N, KF.mnMinX, KF.mnMaxX, KF.mnMinY, KF.mnMaxY); hg_built = true; }
      last_requeried++;
      hg.query(proj[i].u, proj[i].v, proj[i].radius, proj[i].level - 1, proj[i].level, cand);
      bestDist = 256; bestIdx = -1;
      for (int idx : cand) {
        if (vpMatched[idx] >= 0) continue;
        const int dist = DescriptorDistance(P.desc + 32 * (size_t)i, KF.mDescriptors + 32 * (size_t)idx);
        if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
      }
    }
    if (bestIdx >= 0 && bestDist <= TH_LOW * ratioHamming) {
      vpMatched[bestIdx] = P.id[i];
      if (vpMatchedKF && vpPointsKFs) vpMatchedKF[bestIdx] = vpPointsKFs[i];
      nmatches++;
    }
  }
  return nmatches;
}

int ORBmatcher::SearchBySim3(const KeyFrameView& KF1, const KeyFrameView& KF2, const MapPointsView& MPs1, const MapPointsView& MPs2,
                             int32_t* vpMatches12, const int32_t* vnIdxInKF2, const Sim3View& S12, float th) {
  const int N1 = KF1.N, N2 = KF2.N;
  dvm_sim3f S21;   // S21 = S12.inverse()   (:1359, sim3.hpp:129-132)
  dvm_pose::sim3_inverse(S12.q, S12.t, S21.q, S21.t);
  std::vector<uint8_t> valid1(N1, 0), valid2(N2, 0), am2(N2, 0);
  for (int i = 0; i < N1; i++) {
    if (vpMatches12[i] >= 0) {
      const int idx2 = vnIdxInKF2 ? vnIdxInKF2[i] : -1;
      if (idx2 >= 0 && idx2 < N2) am2[idx2] = 1;
    } else if (KF1.mvpMapPoints[i] >= 0 && !(KF1.mpBad && KF1.mpBad[i])) valid1[i] = 1;
  }
  for (int i = 0; i < N2; i++)
    if (KF2.mvpMapPoints[i] >= 0 && !am2[i] && !(KF2.mpBad && KF2.mpBad[i])) valid2[i] = 1;
  auto direction = [&](const KeyFrameView& from, const KeyFrameView& into, const MapPointsView& P, const std::vector<uint8_t>& valid,
                       const dvm_sim3f& S2, std::vector<dvm_match>& res) -> int {
    int rc = ensure_grid(into);
    if (rc != DVM_OK) return rc;
    dvm_kf_camera cam;
    std::memset(&cam, 0, sizeof(cam));
    cam.Tcw = from.Tcw;
    cam.fx = KF1.fx; cam.fy = KF1.fy; cam.cx = KF1.cx; cam.cy = KF1.cy;   // the reference uses pKF1's calibration in both directions
    cam.min_x = into.mnMinX; cam.max_x = into.mnMaxX; cam.min_y = into.mnMinY; cam.max_y = into.mnMaxY;
    cam.log_scale_factor = into.mfLogScaleFactor; cam.n_levels = into.nLevels;
    cam.sim3_pair = 1;
    cam.S2 = S2;
    res.resize(P.n);
    return dvm_project_search(grid_, 0, nullptr, &cam, P.pos, P.normal ? P.normal : P.pos, P.min_dist, P.max_dist, P.desc, valid.data(), P.n,
                              th, into.mvScaleFactors, nullptr, 0.0, res.data(), nullptr, 0, nullptr);
  };
  std::vector<dvm_match> r1, r2;
  int rc = direction(KF1, KF2, MPs1, valid1, S21, r1);
  if (rc != DVM_OK) return rc;
  rc = direction(KF2, KF1, MPs2, valid2, S12, r2);
  if (rc != DVM_OK) return rc;
  int nFound = 0;
  for (int i1 = 0; i1 < N1; i1++) {
    if (!valid1[i1] || r1[i1].best_idx < 0 || r1[i1].best_dist > TH_HIGH) continue;
    const int idx2 = r1[i1].best_idx;
    if (valid2[idx2] && r2[idx2].best_dist <= TH_HIGH && r2[idx2].best_idx == i1) { vpMatches12[i1] = KF2.mvpMapPoints[idx2]; nFound++; }
  }
  return nFound;
}

int ORBmatcher::SearchByProjection(FrameView& Cur, const KeyFrameView& KF, const MapPointsView& P, const int32_t* sAlreadyFound,
                                   int nAlreadyFound, float th, int ORBdist) {
  if (KF.N == 0) return 0;
  last_requeried = 0;
  const dvm_se3f Twc = InverseSE3(Cur.Tcw);   // Ow = Tcw.inverse().translation()   (:1755)
  const float* Ow = Twc.t;
  std::vector<uint8_t> valid(KF.N, 0), skip;
  for (int i = 0; i < KF.N; i++) {
    const int id = KF.mvpMapPoints[i];
    if (id < 0 || (KF.mpBad && KF.mpBad[i])) continue;
    if (std::binary_search(sAlreadyFound, sAlreadyFound + nAlreadyFound, id)) continue;
    valid[i] = 1;
  }
  int rc = ensure_grid(Cur);
  if (rc != DVM_OK) return rc;
  skip.assign(grid_cap_, 0);
  for (int j = 0; j < Cur.N; j++) skip[j] = Cur.mvpMapPoints[j] >= 0;
  dvm_kf_camera cam;
  std::memset(&cam, 0, sizeof(cam));
  cam.Tcw = Cur.Tcw; std::memcpy(cam.Ow, Ow, 12);
  cam.fx = Cur.fx; cam.fy = Cur.fy; cam.cx = Cur.cx; cam.cy = Cur.cy;
  cam.min_x = Cur.mnMinX; cam.max_x = Cur.mnMaxX; cam.min_y = Cur.mnMinY; cam.max_y = Cur.mnMaxY;
  cam.log_scale_factor = KF.mfLogScaleFactor; cam.n_levels = Cur.nLevels;
  cam.sim3_pair = 2;
  std::vector<dvm_match> res(KF.N);
  std::vector<dvm_projection> proj(KF.N);
  rc = dvm_project_search(grid_, 0, skip.data(), &cam, P.pos, P.normal ? P.normal : P.pos, P.min_dist, P.max_dist, P.desc, valid.data(), KF.N, th,
                          Cur.mvScaleFactors, nullptr, 0.0, res.data(), proj.data(), 0, nullptr);
  if (rc != DVM_OK) return rc;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (auto& h : rotHist) h.reserve(500);
  HostGrid hg;
  bool hg_built = false;
  std::vector<int> cand;
  for (int i = 0; i < KF.N; i++) {
    if (!valid[i] || proj[i].level < 0) continue;
    int bestIdx2 = res[i].best_idx, bestDist = res[i].best_dist;
    if (bestIdx2 >= 0 && Cur.mvpMapPoints[bestIdx2] >= 0) {   // claimed by an earlier point of this call
      if (!hg_built) { hg.build(Cur); hg_built = true; }
      last_requeried++;
      hg.query(proj[i].u, proj[i].v, proj[i].radius, proj[i].level - 1, proj[i].level + 1, cand);
      bestDist = 256; bestIdx2 = -1;
      for (int i2 : cand) {
        if (Cur.mvpMapPoints[i2] >= 0) continue;
        const int dist = DescriptorDistance(P.desc + 32 * (size_t)i, Cur.mDescriptors + 32 * (size_t)i2);
        if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
      }
    }
    if (bestIdx2 >= 0 && bestDist <= ORBdist) {
      Cur.mvpMapPoints[bestIdx2] = KF.mvpMapPoints[i];
      nmatches++;
      if (mbCheckOrientation) rotHist[RotBin(KF.mvKeysUn[i].angle, Cur.mvKeysUn[bestIdx2].angle)].push_back(bestIdx2);
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

}  // namespace dvm_host

// ---- C entry points (include/dvmslam_host.h): the same functions for callers without the C++ classes -- the Python harness of
// tests/ and bench.py, a C or FFI binding.  A C++ caller uses the class directly.
using dvm_host::FeatureVectorView;
using dvm_host::FrameView;
using dvm_host::KeyFrameView;
using dvm_host::MapPointsView;
extern "C" {
void dvmh_set_match_pool(dvm_match_pool* pool) { dvm_host::set_match_pool(pool); }
int dvmh_search_by_projection_frames(int device, int Nc, const dvm_keypoint* kps_c, const uint8_t* desc_c, int32_t* mp_c, const dvm_se3f* Tcw,
                                     const float* K, const float* bounds, const float* scale_factors, int nlevels, int Nl, const dvm_keypoint* kps_l,
                                     const int32_t* mp_l, const uint8_t* outlier_l, const dvmh_map_point* mps, float th, int check_ori, int* requeried) {
  FrameView C, L;
  C.N = Nc; C.mvKeysUn = kps_c; C.mDescriptors = desc_c; C.mvpMapPoints = mp_c;
  C.Tcw = *Tcw;
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
int dvmh_search_by_projection_frames_dev(int device, int Nc, const dvm_keypoint* kps_c, const uint8_t* desc_c, int32_t* mp_c, const dvm_se3f* Tcw,
                                         const float* K, const float* bounds, const float* scale_factors, int nlevels, int Nl, const dvm_keypoint* kps_l,
                                         const int32_t* mp_l, const uint8_t* outlier_l, const dvmh_map_point* mps, float th, int check_ori, int* requeried,
                                         const dvm_device_frame* dev_c, int* grid_from_device) {
  FrameView C, L;
  C.N = Nc; C.mvKeysUn = kps_c; C.mDescriptors = desc_c; C.mvpMapPoints = mp_c;
  C.Tcw = *Tcw;
  C.fx = K[0]; C.fy = K[1]; C.cx = K[2]; C.cy = K[3];
  C.mnMinX = bounds[0]; C.mnMaxX = bounds[1]; C.mnMinY = bounds[2]; C.mnMaxY = bounds[3];
  C.mvScaleFactors = scale_factors; C.nLevels = nlevels;
  L = C;
  C.dev = dev_c;
  L.N = Nl; L.mvKeysUn = kps_l; L.mDescriptors = nullptr; L.mvpMapPoints = const_cast<int32_t*>(mp_l); L.mvbOutlier = outlier_l;
  dvm_host::ORBmatcher m(0.9f, check_ori != 0, device);
  const int n = m.SearchByProjection(C, L, mps, th, true);
  if (requeried) *requeried = m.last_requeried;
  if (grid_from_device) *grid_from_device = m.last_grid_from_device ? 1 : 0;
  return n;
}
int dvmh_search_by_projection_points(int device, int N, const dvm_keypoint* kps, const uint8_t* desc, int32_t* mp, const uint8_t* claimed_obs,
                                     const float* bounds, const float* scale_factors, int nlevels, const dvmh_tracked_point* pts, int npts, float th,
                                     float nnratio, int far_points, float th_far, int* requeried) {
  FrameView F;
  F.N = N; F.mvKeysUn = kps; F.mDescriptors = desc; F.mvpMapPoints = mp;
  F.mnMinX = bounds[0]; F.mnMaxX = bounds[1]; F.mnMinY = bounds[2]; F.mnMaxY = bounds[3];
  F.mvScaleFactors = scale_factors; F.nLevels = nlevels;
  dvm_host::ORBmatcher m(nnratio, true, device);
  const int n = m.SearchByProjection(F, pts, npts, claimed_obs, th, far_points != 0, th_far, 0);
  if (requeried) *requeried = m.last_requeried;
  return n;
}
int dvmh_search_for_initialization(int device, const dvmh_frame_view* F1, const dvmh_frame_view* F2, float* prev_matched, int32_t* matches12, int window,
                                   float nnratio, int check_ori) {
  dvm_host::ORBmatcher m(nnratio, check_ori != 0, device);
  return m.SearchForInitialization(FrameView(*F1), FrameView(*F2), prev_matched, matches12, window);
}
int dvmh_search_by_bow_kf_frame(int device, const dvmh_keyframe_view* KF, const dvmh_frame_view* F, const dvmh_feature_vector_view* Ffv, float nnratio,
                                int check_ori, int32_t* matches, int* requeried) {
  dvm_host::ORBmatcher m(nnratio, check_ori != 0, device);
  const int n = m.SearchByBoW(KeyFrameView(*KF), FrameView(*F), FeatureVectorView(*Ffv), matches);
  if (requeried) *requeried = m.last_requeried;
  return n;
}
int dvmh_search_by_bow_kf_kf(int device, const dvmh_keyframe_view* KF1, const dvmh_keyframe_view* KF2, float nnratio, int check_ori, int32_t* matches12,
                             int* requeried) {
  dvm_host::ORBmatcher m(nnratio, check_ori != 0, device);
  const int n = m.SearchByBoW(KeyFrameView(*KF1), KeyFrameView(*KF2), matches12);
  if (requeried) *requeried = m.last_requeried;
  return n;
}
void dvmh_pose_matrices(const dvm_se3f* Tcw, float* Rcw, float* tcw, float* Ow) { dvm_host::PoseMatrices(*Tcw, Rcw, tcw, Ow); }
void dvmh_se3_inverse(const dvm_se3f* T, dvm_se3f* out) { *out = dvm_host::InverseSE3(*T); }
void dvmh_sim3_to_se3(const dvm_sim3f* S, dvm_se3f* Tcw, float* Ow) { dvm_host::Sim3ToSE3(*S, *Tcw, Ow); }
void dvmh_se3_apply(const dvm_se3f* T, const float* p, int n, float* out) {
  for (int i = 0; i < n; i++) dvm_pose::se3_apply(T->q, T->t, p + 3 * i, out + 3 * i);
}
void dvmh_sim3_apply(const dvm_sim3f* S, const float* p, int n, float* out) {
  for (int i = 0; i < n; i++) dvm_pose::sim3_apply(S->q, S->t, p + 3 * i, out + 3 * i);
}
void dvmh_sim3_inverse(const dvm_sim3f* S, dvm_sim3f* out) { dvm_pose::sim3_inverse(S->q, S->t, out->q, out->t); }
float dvmh_logf(float x) { return dvm_pose::logf_shared(x); }
void dvmh_triangulation_geometry(const dvmh_keyframe_view* KF1, const dvmh_keyframe_view* KF2, float* R12, float* t12, float* ep, float* F12) {
  dvm_host::ORBmatcher::TriangulationGeometry(KeyFrameView(*KF1), KeyFrameView(*KF2), R12, t12, ep, F12);
}
int dvmh_search_for_triangulation(int device, const dvmh_keyframe_view* KF1, const dvmh_keyframe_view* KF2, int coarse, int check_ori, int32_t* pairs) {
  dvm_host::ORBmatcher m(0.6f, check_ori != 0, device);
  return m.SearchForTriangulation(KeyFrameView(*KF1), KeyFrameView(*KF2), pairs, false, coarse != 0);
}
int dvmh_fuse(int device, const dvmh_keyframe_view* KF, const dvmh_map_points_view* P, const uint8_t* inKF, float th, int32_t* best_idx) {
  dvm_host::ORBmatcher m(0.6f, true, device);
  return m.Fuse(KeyFrameView(*KF), MapPointsView(*P), inKF, th, best_idx);
}
int dvmh_fuse_sim3(int device, dvmh_keyframe_view* KF, const dvm_sim3f* Scw, const dvmh_map_points_view* P, float th, int32_t* replace) {
  dvm_host::ORBmatcher m(0.6f, true, device);
  KeyFrameView K(*KF);                    // (the function writes through K.mvpMapPoints, the caller's array)
  return m.Fuse(K, *Scw, MapPointsView(*P), th, replace);
}
int dvmh_search_by_projection_reloc(int device, dvmh_frame_view* Cur, const dvmh_keyframe_view* KF, const dvmh_map_points_view* P, const int32_t* already,
                                    int n_already, float th, int orb_dist, int check_ori, int* requeried) {
  dvm_host::ORBmatcher m(0.9f, check_ori != 0, device);
  FrameView C(*Cur);
  const int n = m.SearchByProjection(C, KeyFrameView(*KF), MapPointsView(*P), already, n_already, th, orb_dist);
  if (requeried) *requeried = m.last_requeried;
  return n;
}
int dvmh_search_by_sim3(int device, const dvmh_keyframe_view* KF1, const dvmh_keyframe_view* KF2, const dvmh_map_points_view* P1,
                        const dvmh_map_points_view* P2, int32_t* matches12, const int32_t* idx_in_kf2, const dvm_sim3f* S12, float th) {
  dvm_host::ORBmatcher m(0.6f, true, device);
  return m.SearchBySim3(KeyFrameView(*KF1), KeyFrameView(*KF2), MapPointsView(*P1), MapPointsView(*P2), matches12, idx_in_kf2, *S12, th);
}
int dvmh_search_by_projection_sim3(int device, const dvmh_keyframe_view* KF, const dvm_sim3f* Scw, const dvmh_map_points_view* P, const int32_t* point_kf,
                                   int32_t* matched, int32_t* matched_kf, int th, float ratio_hamming, int* requeried) {
  dvm_host::ORBmatcher m(0.6f, true, device);
  const int n = m.SearchByProjection(KeyFrameView(*KF), *Scw, MapPointsView(*P), point_kf, matched, matched_kf, th, ratio_hamming);
  if (requeried) *requeried = m.last_requeried;
  return n;
}
}
