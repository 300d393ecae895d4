// oracle/match_oracle.cpp -- CPU restatement of the reference matching primitives.
//
// TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  Integer work; PARITY UNPINNED against the real
// reference binary only in the sense that the reference holds no test vector -- the arithmetic is
// fully contained in the cited reference lines (no third-party code involved).
//
// Reference files followed (under /root/reference/src/slam_system/orb_slam3/):
//   src/ORBmatcher.cc:1900-1914  ORBmatcher::DescriptorDistance  -> descriptor_distance()
//   src/Frame.cc:443-444         grid inverse cell sizes          -> orc_grid ctor
//   src/Frame.cc:481-506         AssignFeaturesToGrid             -> orc_grid ctor
//   src/Frame.cc:773-782         PosInGrid (round, drop if out)   -> orc_grid ctor
//   src/Frame.cc:712-770         GetFeaturesInArea                -> features_in_area()
//   src/ORBmatcher.cc:70-115, :1604-1639  best / second-best loops -> orc_match_window()
//   src/ORBmatcher.cc:1553-1748  SearchByProjection(CurrentFrame, LastFrame, th, bMono=true), whole function,
//   src/ORBmatcher.cc:1862-1896  ComputeThreeMaxima              -> orc_search_by_projection_frames()
//   src/ORBmatcher.cc:44-212     SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints), mono branch,
//                                RadiusByViewingCos              -> orc_search_by_projection_points()
//   src/ORBmatcher.cc:605-707    SearchForInitialization          -> orc_search_for_initialization()
//   src/ORBmatcher.cc:214-393    SearchByBoW(KF, F, vpMapPointMatches), mono -> orc_search_by_bow_kf_frame()
//   src/ORBmatcher.cc:709-834    SearchByBoW(KF1, KF2, vpMatches12)          -> orc_search_by_bow_kf_kf()
//   src/ORBmatcher.cc:836-1058   SearchForTriangulation, mono; CameraModels/Pinhole.cpp:104-127 epipolarConstrain
//                                                                  -> orc_search_for_triangulation()
//   src/ORBmatcher.cc:1060-1234  Fuse(KF, vpMapPoints, th) search part (up to bestDist / bestIdx), :1236-1345 Fuse(KF, Scw, ...)
//                                whole function, :395-496 SearchByProjection(KF, Scw, vpPoints, vpMatched, th, ratioHamming)
//                                whole function; KeyFrame.cc:750-794 GetFeaturesInArea / IsInImage; MapPoint.cc:573-587
//                                PredictScale              -> orc_project_search(), orc_fuse_sim3(), orc_search_by_projection_sim3()
//   src/ORBmatcher.cc:1347-1551  SearchBySim3, whole function      -> orc_search_by_sim3()
//   src/ORBmatcher.cc:1750-1860  SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) -> orc_search_by_projection_reloc()
//   src/MapPoint.cc:384-453      MapPoint::ComputeDistinctiveDescriptors -> orc_distinctive_descriptors()
//   Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1025-1138 transform (TF_IDF, L1), BowVector.cpp:32-72,
//   FeatureVector.cpp:27-38, ScoringObject.cpp:23-63 L1Scoring::score, FORB.cpp:80-97 -> orc_vocab_transform(), orc_bow_score()
#include "oracle.h"
#include "sophus_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

namespace {
namespace so = sophus_oracle;
const int kGridCols = 64, kGridRows = 48;  // Frame.h:44-45
// poses cross the oracle's C interface as 7 floats: quaternion coeffs (x, y, z, w) as Sophus stores them, then translation
so::SE3 load_se3(const float* T) { return so::SE3{so::Quat{T[0], T[1], T[2], T[3]}, {T[4], T[5], T[6]}}; }
so::Sim3 load_sim3(const float* S) { return so::Sim3{so::Quat{S[0], S[1], S[2], S[3]}, {S[4], S[5], S[6]}}; }
// MapPoint::PredictScale (MapPoint.cc:573-587); log(float) is the shared logf spec (sophus_oracle.h)
int predict_scale(float max_dist, float dist, float log_scale_factor, int n_levels) {
  const float ratio = max_dist / dist;
  int nScale = (int)std::ceil(so::logf_spec(ratio) / log_scale_factor);
  if (nScale < 0) nScale = 0; else if (nScale >= n_levels) nScale = n_levels - 1;
  return nScale;
}

int descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t wa, wb;
    std::memcpy(&wa, a + 4 * i, 4);
    std::memcpy(&wb, b + 4 * i, 4);
    uint32_t v = wa ^ wb;
    v = v - ((v >> 1) & 0x55555555u);
    v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
    dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
  }
  return dist;
}
}  // namespace

struct orc_grid {
  std::vector<orc_keypoint> kps;
  float minX, minY, wInv, hInv;
  std::vector<int> cell[kGridCols][kGridRows];
};

extern "C" {

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

void orc_hamming_matrix(const uint8_t* A, int nA, const uint8_t* B, int nB, uint16_t* D) {
  for (int i = 0; i < nA; i++)
    for (int j = 0; j < nB; j++) D[(size_t)i * nB + j] = (uint16_t)descriptor_distance(A + 32 * i, B + 32 * j);
}

orc_grid* orc_grid_create(const orc_keypoint* kps, int n, float minX, float maxX, float minY, float maxY) {
  orc_grid* g = new orc_grid;
  g->kps.assign(kps, kps + n);
  g->minX = minX;
  g->minY = minY;
  g->wInv = static_cast<float>(kGridCols) / static_cast<float>(maxX - minX);
  g->hInv = static_cast<float>(kGridRows) / static_cast<float>(maxY - minY);
  for (int i = 0; i < n; i++) {
    int px = (int)std::round((kps[i].x - minX) * g->wInv);
    int py = (int)std::round((kps[i].y - minY) * g->hInv);
    if (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) continue;
    g->cell[px][py].push_back(i);
  }
  return g;
}
void orc_grid_destroy(orc_grid* g) { delete g; }

static void features_in_area(const orc_grid* g, float x, float y, float r, int minLevel, int maxLevel,
                             std::vector<int>& out) {
  out.clear();
  const int nMinCellX = std::max(0, (int)std::floor((x - g->minX - r) * g->wInv));
  if (nMinCellX >= kGridCols) return;
  const int nMaxCellX = std::min(kGridCols - 1, (int)std::ceil((x - g->minX + r) * g->wInv));
  if (nMaxCellX < 0) return;
  const int nMinCellY = std::max(0, (int)std::floor((y - g->minY - r) * g->hInv));
  if (nMinCellY >= kGridRows) return;
  const int nMaxCellY = std::min(kGridRows - 1, (int)std::ceil((y - g->minY + r) * g->hInv));
  if (nMaxCellY < 0) return;
  const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
      for (int idx : g->cell[ix][iy]) {
        const orc_keypoint& kp = g->kps[idx];
        if (checkLevels) {
          if (kp.octave < minLevel) continue;
          if (maxLevel >= 0 && kp.octave > maxLevel) continue;
        }
        const float dx = kp.x - x, dy = kp.y - y;
        if (std::fabs(dx) < r && std::fabs(dy) < r) out.push_back(idx);
      }
}

int orc_grid_features_in_area(const orc_grid* g, float x, float y, float r, int minLevel, int maxLevel,
                              int32_t* out, int cap) {
  std::vector<int> v;
  features_in_area(g, x, y, r, minLevel, maxLevel, v);
  for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
  return (int)v.size();
}

void orc_match_window(const orc_grid* g, const uint8_t* tdesc, const uint8_t* skip, const uint8_t* qdesc,
                      const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                      const int32_t* qmax, int nq, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist,
                      int32_t* best_level, int32_t* second_level) {
  std::vector<int> cand;
  for (int q = 0; q < nq; q++) {
    features_in_area(g, qx[q], qy[q], qr[q], qmin[q], qmax[q], cand);
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int idx : cand) {
      if (skip && skip[idx]) continue;
      const int dist = descriptor_distance(qdesc + 32 * (size_t)q, tdesc + 32 * (size_t)idx);
      if (dist < bestDist) {
        bestDist2 = bestDist;
        bestDist = dist;
        bestLevel2 = bestLevel;
        bestLevel = g->kps[idx].octave;
        bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = g->kps[idx].octave;
        bestDist2 = dist;
      }
    }
    best_idx[q] = bestIdx;
    best_dist[q] = bestDist;
    second_dist[q] = bestDist2;
    best_level[q] = bestLevel;
    second_level[q] = bestLevel2;
  }
}

// Frame::isInFrustum (mono branch, Frame.cc:575-636) + MapPoint::PredictScale (MapPoint.cc:573-587) +
// Get{Min,Max}DistanceInvariance (0.8f*min, 1.2f*max, MapPoint.cc:545-553).  All float, like the reference.
void orc_is_in_frustum(const orc_frustum_frame* F, const float* P, const float* normal, const float* min_dist,
                       const float* max_dist, int n, float viewing_cos_limit, orc_track_point* out) {
  for (int i = 0; i < n; i++) {
    orc_track_point& o = out[i];
    o.in_view = 0; o.proj_x = -1; o.proj_y = -1; o.proj_xr = 0; o.depth = 0; o.level = -1; o.view_cos = 0;
    const float* p = P + 3 * i;
    float Pc[3];
    so::mat3_vec(F->Rcw, p, Pc);                                 // mRcw * P + mtcw (:585): matrix form, Eigen's sum order
    for (int r = 0; r < 3; r++) Pc[r] = Pc[r] + F->tcw[r];
    const float Pc_dist = so::norm3(Pc);
    const float PcZ = Pc[2];
    const float invz = 1.0f / PcZ;
    if (PcZ < 0.0f) continue;
    const float u = F->fx * Pc[0] / Pc[2] + F->cx, v = F->fy * Pc[1] / Pc[2] + F->cy;
    if (u < F->min_x || u > F->max_x) continue;
    if (v < F->min_y || v > F->max_y) continue;
    o.proj_x = u; o.proj_y = v;
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    const float PO[3] = {p[0] - F->Ow[0], p[1] - F->Ow[1], p[2] - F->Ow[2]};
    const float dist = so::norm3(PO);
    if (dist < minDistance || dist > maxDistance) continue;
    const float* Pn = normal + 3 * i;
    const float viewCos = so::dot3(PO, Pn) / dist;
    if (viewCos < viewing_cos_limit) continue;
    const int nScale = predict_scale(max_dist[i], dist, F->log_scale_factor, F->n_levels);
    o.in_view = 1; o.proj_xr = u - F->bf * invz; o.depth = Pc_dist; o.level = nScale; o.view_cos = viewCos;
  }
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) -- ORBmatcher.cc:1553-1748,
// monocular (bForward = bBackward = false, no right image, no Nleft).  Plain sequential restatement: the claim check
// `CurrentFrame.mvpMapPoints[i2]->Observations() > 0` is evaluated against the state left by earlier iterations.
int orc_search_by_projection_frames(int Nc, const orc_keypoint* kps_c, const uint8_t* desc_c, int32_t* mp_c, const float* Tcw7,
                                    const float* K, const float* bounds, const float* scale_factors,
                                    int Nl, const orc_keypoint* kps_l, const int32_t* mp_l, const uint8_t* outlier_l,
                                    const orc_map_point* mps, float th, int check_ori) {
  const int HISTO_LENGTH = 30, TH_HIGH = 100;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  orc_grid* g = orc_grid_create(kps_c, Nc, bounds[0], bounds[1], bounds[2], bounds[3]);
  std::vector<int> vIndices2;
  const so::SE3 Tcw = load_se3(Tcw7);   // CurrentFrame.GetPose() (:1562)
  for (int i = 0; i < Nl; i++) {
    const int pMP = mp_l[i];
    if (pMP < 0) continue;
    if (outlier_l && outlier_l[i]) continue;
    const float* X = mps[pMP].pos;
    float x3Dc[3];
    so::se3_act(Tcw, X, x3Dc);           // Tcw * x3Dw (:1577): Sophus' quaternion action
    const float invzc = 1.0 / x3Dc[2];
    if (invzc < 0) continue;
    const float u = K[0] * x3Dc[0] / x3Dc[2] + K[2], v = K[1] * x3Dc[1] / x3Dc[2] + K[3];  // Pinhole::project
    if (u < bounds[0] || u > bounds[1]) continue;
    if (v < bounds[2] || v > bounds[3]) continue;
    const int nLastOctave = kps_l[i].octave;
    const float radius = th * scale_factors[nLastOctave];
    features_in_area(g, u, v, radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
    if (vIndices2.empty()) continue;
    int bestDist = 256, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (mp_c[i2] >= 0 && mps[mp_c[i2]].n_obs > 0) continue;
      const int dist = descriptor_distance(mps[pMP].desc, desc_c + 32 * (size_t)i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= TH_HIGH) {
      mp_c[bestIdx2] = pMP;
      nmatches++;
      if (check_ori) {
        float rot = kps_l[i].angle - kps_c[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < HISTO_LENGTH; i++) {
      const int s = (int)rotHist[i].size();
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) ind3 = -1;
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (size_t j = 0; j < rotHist[i].size(); j++) { mp_c[rotHist[i][j]] = -1; nmatches--; }
  }
  orc_grid_destroy(g);
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th, bFarPoints, thFarPoints) --
// ORBmatcher.cc:44-205, monocular (F.Nleft == -1, mvuRight < 0).  mp[idx] = index of the matched point (-1 = NULL);
// claimed_obs[idx] = (F.mvpMapPoints[idx]->Observations() > 0) for the entries that are set at entry.
int orc_search_by_projection_points(int N, const orc_keypoint* kps, const uint8_t* desc, int32_t* mp, const uint8_t* claimed_obs,
                                    const float* bounds, const float* scale_factors, const orc_tracked_point* pts, int npts,
                                    float th, float nnratio, int far_points, float th_far) {
  const int TH_HIGH = 100;
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  orc_grid* g = orc_grid_create(kps, N, bounds[0], bounds[1], bounds[2], bounds[3]);
  std::vector<uint8_t> has_obs(N, 0);
  for (int j = 0; j < N; j++) has_obs[j] = (mp[j] >= 0 && claimed_obs && claimed_obs[j]) ? 1 : 0;
  std::vector<int> vIndices;
  for (int iMP = 0; iMP < npts; iMP++) {
    const orc_tracked_point& pMP = pts[iMP];
    if (!pMP.in_view) continue;
    if (far_points && pMP.depth > th_far) continue;
    if (pMP.bad) continue;
    const int nPredictedLevel = pMP.level;
    float r = pMP.view_cos > 0.998 ? 2.5 : 4.0;
    if (bFactor) r *= th;
    features_in_area(g, pMP.proj_x, pMP.proj_y, r * scale_factors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, vIndices);
    if (vIndices.empty()) continue;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int idx : vIndices) {
      if (mp[idx] >= 0 && has_obs[idx]) continue;
      const int dist = descriptor_distance(pMP.desc, desc + 32 * (size_t)idx);
      if (dist < bestDist) {
        bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kps[idx].octave; bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = kps[idx].octave; bestDist2 = dist;
      }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
      if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
        mp[bestIdx] = iMP;
        has_obs[bestIdx] = pMP.n_obs > 0 ? 1 : 0;
        nmatches++;
      }
    }
  }
  orc_grid_destroy(g);
  return nmatches;
}

// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:420-447) for a batch of map points: the N x N table, every row
// sorted with std::sort, median = vDists[0.5 * (N - 1)], first strictly smaller median wins.
void orc_distinctive_descriptors(const uint8_t* desc, const int32_t* off, int npts, int32_t* best_idx, int32_t* best_median) {
  for (int p = 0; p < npts; p++) {
    const int N = off[p + 1] - off[p];
    if (N <= 0) { best_idx[p] = -1; best_median[p] = -1; continue; }
    const uint8_t* D = desc + (size_t)off[p] * 32;
    std::vector<float> Distances((size_t)N * N);
    for (int i = 0; i < N; i++) {
      Distances[(size_t)i * N + i] = 0;
      for (int j = i + 1; j < N; j++) {
        const int distij = descriptor_distance(D + 32 * i, D + 32 * j);
        Distances[(size_t)i * N + j] = distij;
        Distances[(size_t)j * N + i] = distij;
      }
    }
    int BestMedian = 0x7fffffff, BestIdx = 0;
    for (int i = 0; i < N; i++) {
      std::vector<int> vDists(Distances.begin() + (size_t)i * N, Distances.begin() + (size_t)(i + 1) * N);
      std::sort(vDists.begin(), vDists.end());
      const int median = vDists[(size_t)(0.5 * (N - 1))];
      if (median < BestMedian) { BestMedian = median; BestIdx = i; }
    }
    best_idx[p] = BestIdx; best_median[p] = BestMedian;
  }
}

// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (TemplatedVocabulary.h:1098-1138)
static void vocab_transform_one(const int32_t* child_off, const int32_t* children, const uint8_t* node_desc, const double* weight,
                                const int32_t* word_id, int L, const uint8_t* feature, int levelsup, int32_t* wid, double* w,
                                int32_t* nid) {
  const int nid_level = L - levelsup;
  *nid = (nid_level <= 0) ? 0 : -1;   // (the reference leaves it unset in the second case)
  int final_id = 0, current_level = 0;
  do {
    ++current_level;
    const int c0 = child_off[final_id], c1 = child_off[final_id + 1];
    final_id = children[c0];
    double best_d = descriptor_distance(feature, node_desc + 32 * (size_t)final_id);
    for (int c = c0 + 1; c < c1; c++) {
      const int id = children[c];
      const double d = descriptor_distance(feature, node_desc + 32 * (size_t)id);
      if (d < best_d) { best_d = d; final_id = id; }
    }
    if (current_level == nid_level) *nid = final_id;
  } while (child_off[final_id + 1] > child_off[final_id]);
  *wid = word_id[final_id];
  *w = weight[final_id];
}

// per-feature outputs + the whole transform(features, BowVector&, FeatureVector&, levelsup) with TF_IDF / L1_NORM,
// flattened: bow (ids ascending, values), fv (node ids ascending, CSR feature lists)
void orc_vocab_transform(int n_nodes, const int32_t* child_off, const int32_t* children, const uint8_t* node_desc,
                         const double* weight, const int32_t* word_id, int L, const uint8_t* features, int n, int levelsup,
                         int32_t* out_word, int32_t* out_node, double* out_weight, int32_t* bow_ids, double* bow_vals,
                         int32_t* n_bow, int32_t* fv_nodes, int32_t* fv_off, int32_t* fv_feat, int32_t* n_fv) {
  (void)n_nodes;
  std::map<unsigned, double> v;
  std::map<unsigned, std::vector<unsigned>> fv;
  for (int i = 0; i < n; i++) {
    int32_t id, nid; double w;
    vocab_transform_one(child_off, children, node_desc, weight, word_id, L, features + 32 * (size_t)i, levelsup, &id, &w, &nid);
    out_word[i] = id; out_node[i] = nid; out_weight[i] = w;
    if (w > 0) {
      auto vit = v.lower_bound((unsigned)id);
      if (vit != v.end() && !(v.key_comp()((unsigned)id, vit->first))) vit->second += w;
      else v.insert(vit, std::make_pair((unsigned)id, w));
      auto fit = fv.lower_bound((unsigned)nid);
      if (fit != fv.end() && fit->first == (unsigned)nid) fit->second.push_back(i);
      else { fit = fv.insert(fit, std::make_pair((unsigned)nid, std::vector<unsigned>())); fit->second.push_back(i); }
    }
  }
  double norm = 0.0;
  for (auto& e : v) norm += std::fabs(e.second);
  if (norm > 0.0) for (auto& e : v) e.second /= norm;
  int k = 0;
  for (auto& e : v) { bow_ids[k] = (int32_t)e.first; bow_vals[k] = e.second; k++; }
  *n_bow = k;
  int m = 0, t = 0;
  fv_off[0] = 0;
  for (auto& e : fv) { fv_nodes[m] = (int32_t)e.first; for (unsigned f : e.second) fv_feat[t++] = (int32_t)f; fv_off[++m] = t; }
  *n_fv = m;
}

double orc_bow_score(const int32_t* ids1, const double* vals1, int n1, const int32_t* ids2, const double* vals2, int n2) {
  std::map<unsigned, double> v1, v2;
  for (int i = 0; i < n1; i++) v1[(unsigned)ids1[i]] = vals1[i];
  for (int i = 0; i < n2; i++) v2[(unsigned)ids2[i]] = vals2[i];
  auto v1_it = v1.begin(), v2_it = v2.begin();
  double score = 0;
  while (v1_it != v1.end() && v2_it != v2.end()) {
    const double vi = v1_it->second, wi = v2_it->second;
    if (v1_it->first == v2_it->first) { score += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi); ++v1_it; ++v2_it; }
    else if (v1_it->first < v2_it->first) v1_it = v1.lower_bound(v2_it->first);
    else v2_it = v2.lower_bound(v1_it->first);
  }
  return -score / 2.0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Whole matcher functions of SURVEY.md 8(a) rows M4-M7 (monocular: Nleft == -1, no mpCamera2, mvuRight < 0).
 * Map points are indices (-1 = NULL); "bad" flags stand for MapPoint::isBad().  FeatureVectors are CSR:
 * node ids ascending (std::map order), features of node k = feat[off[k] .. off[k+1]).
 * ------------------------------------------------------------------------------------------------------------------ */
namespace {
const int kHisto = 30, kThLow = 50;
void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {  // ORBmatcher.cc:1862-1896
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
int rot_bin(float a1, float a2) {
  const float factor = 1.0f / kHisto;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * factor);
  if (bin == kHisto) bin = 0;
  return bin;
}
// std::map::lower_bound over the CSR node list
int fv_lower_bound(const int32_t* nodes, int n, int from, int key) { return (int)(std::lower_bound(nodes + from, nodes + n, key) - nodes); }
}  // namespace

int orc_search_for_initialization(int N1, const orc_keypoint* kps1, const uint8_t* desc1, int N2, const orc_keypoint* kps2,
                                  const uint8_t* desc2, const float* bounds, float* prev_matched, int32_t* matches12,
                                  int windowSize, float nnratio, int check_ori) {
  int nmatches = 0;
  for (int i = 0; i < N1; i++) matches12[i] = -1;
  std::vector<int> rotHist[kHisto];
  std::vector<int> vMatchedDistance(N2, INT32_MAX), vnMatches21(N2, -1);
  orc_grid* g = orc_grid_create(kps2, N2, bounds[0], bounds[1], bounds[2], bounds[3]);
  std::vector<int> vIndices2;
  for (int i1 = 0; i1 < N1; i1++) {
    const int level1 = kps1[i1].octave;
    if (level1 > 0) continue;
    features_in_area(g, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)windowSize, level1, level1, vIndices2);
    if (vIndices2.empty()) continue;
    int bestDist = INT32_MAX, bestDist2 = INT32_MAX, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      const int dist = descriptor_distance(desc1 + 32 * (size_t)i1, desc2 + 32 * (size_t)i2);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
    if (bestDist <= kThLow) {
      if (bestDist < (float)bestDist2 * nnratio) {
        if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        matches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (check_ori) rotHist[rot_bin(kps1[i1].angle, kps2[bestIdx2].angle)].push_back(i1);
      }
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, kHisto, ind1, ind2, ind3);
    for (int i = 0; i < kHisto; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx1 : rotHist[i])
        if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
    }
  }
  for (int i1 = 0; i1 < N1; i1++)
    if (matches12[i1] >= 0) { prev_matched[2 * i1] = kps2[matches12[i1]].x; prev_matched[2 * i1 + 1] = kps2[matches12[i1]].y; }
  orc_grid_destroy(g);
  return nmatches;
}

int orc_search_by_bow_kf_frame(const orc_keypoint* kps_kf, const uint8_t* desc_kf, const int32_t* mp_kf, const uint8_t* bad_kf,
                               const int32_t* fv_nodes_kf, const int32_t* fv_off_kf, const int32_t* fv_feat_kf, int nn_kf, int N_f,
                               const orc_keypoint* kps_f, const uint8_t* desc_f, const int32_t* fv_nodes_f, const int32_t* fv_off_f,
                               const int32_t* fv_feat_f, int nn_f, float nnratio, int check_ori, int32_t* matches) {
  for (int i = 0; i < N_f; i++) matches[i] = -1;
  int nmatches = 0;
  std::vector<int> rotHist[kHisto];
  int a = 0, b = 0;
  while (a < nn_kf && b < nn_f) {
    if (fv_nodes_kf[a] == fv_nodes_f[b]) {
      for (int iKF = fv_off_kf[a]; iKF < fv_off_kf[a + 1]; iKF++) {
        const int realIdxKF = fv_feat_kf[iKF];
        const int pMP = mp_kf[realIdxKF];
        if (pMP < 0) continue;
        if (bad_kf && bad_kf[realIdxKF]) continue;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (int iF = fv_off_f[b]; iF < fv_off_f[b + 1]; iF++) {
          const int realIdxF = fv_feat_f[iF];
          if (matches[realIdxF] >= 0) continue;
          const int dist = descriptor_distance(desc_kf + 32 * (size_t)realIdxKF, desc_f + 32 * (size_t)realIdxF);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 <= kThLow) {
          if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
            matches[bestIdxF] = pMP;
            if (check_ori) rotHist[rot_bin(kps_kf[realIdxKF].angle, kps_f[bestIdxF].angle)].push_back(bestIdxF);
            nmatches++;
          }
        }
      }
      a++; b++;
    } else if (fv_nodes_kf[a] < fv_nodes_f[b]) {
      a = fv_lower_bound(fv_nodes_kf, nn_kf, a, fv_nodes_f[b]);
    } else {
      b = fv_lower_bound(fv_nodes_f, nn_f, b, fv_nodes_kf[a]);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, kHisto, ind1, ind2, ind3);
    for (int i = 0; i < kHisto; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx : rotHist[i]) { matches[idx] = -1; nmatches--; }
    }
  }
  return nmatches;
}

int orc_search_by_bow_kf_kf(int N1, const orc_keypoint* kps1, const uint8_t* desc1, const int32_t* mp1, const uint8_t* bad1,
                            const int32_t* fv_nodes1, const int32_t* fv_off1, const int32_t* fv_feat1, int nn1, int N2,
                            const orc_keypoint* kps2, const uint8_t* desc2, const int32_t* mp2, const uint8_t* bad2,
                            const int32_t* fv_nodes2, const int32_t* fv_off2, const int32_t* fv_feat2, int nn2, float nnratio,
                            int check_ori, int32_t* matches12) {
  for (int i = 0; i < N1; i++) matches12[i] = -1;
  std::vector<uint8_t> vbMatched2(N2, 0);
  std::vector<int> rotHist[kHisto];
  int nmatches = 0, a = 0, b = 0;
  while (a < nn1 && b < nn2) {
    if (fv_nodes1[a] == fv_nodes2[b]) {
      for (int i1 = fv_off1[a]; i1 < fv_off1[a + 1]; i1++) {
        const int idx1 = fv_feat1[i1];
        if (mp1[idx1] < 0) continue;
        if (bad1 && bad1[idx1]) continue;
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (int i2 = fv_off2[b]; i2 < fv_off2[b + 1]; i2++) {
          const int idx2 = fv_feat2[i2];
          if (vbMatched2[idx2] || mp2[idx2] < 0) continue;
          if (bad2 && bad2[idx2]) continue;
          const int dist = descriptor_distance(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 < kThLow) {
          if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
            matches12[idx1] = mp2[bestIdx2];
            vbMatched2[bestIdx2] = 1;
            if (check_ori) rotHist[rot_bin(kps1[idx1].angle, kps2[bestIdx2].angle)].push_back(idx1);
            nmatches++;
          }
        }
      }
      a++; b++;
    } else if (fv_nodes1[a] < fv_nodes2[b]) {
      a = fv_lower_bound(fv_nodes1, nn1, a, fv_nodes2[b]);
    } else {
      b = fv_lower_bound(fv_nodes2, nn2, b, fv_nodes1[a]);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, kHisto, ind1, ind2, ind3);
    for (int i = 0; i < kHisto; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx : rotHist[i]) { matches12[idx] = -1; nmatches--; }
    }
  }
  return nmatches;
}

// Geometry of SearchForTriangulation (ORBmatcher.cc:841-862) + Pinhole::epipolarConstrain's fundamental matrix
// (Pinhole.cpp:106-110): Cw = pKF1->GetCameraCenter() = T1w.inverse().translation() (KeyFrame.cc:224-257), C2 = T2w * Cw,
// T12 = T1w * Tw2 (Sophus SE3 product: normalised quaternion product), R12 = T12.rotationMatrix(), F12 =
// K1.transpose().inverse() * hat(t12) * R12 * K2.inverse() with Eigen's cofactor inverse and left-to-right 3x3 products.
void orc_triangulation_geometry(const float* T1w7, const float* T2w7, const float* K1, const float* K2, float* R12, float* t12,
                                float* ep, float* F12) {
  const so::SE3 T1w = load_se3(T1w7), T2w = load_se3(T2w7);
  const so::SE3 Tw1 = so::se3_inverse(T1w), Tw2 = so::se3_inverse(T2w);
  float C2[3];
  so::se3_act(T2w, Tw1.t, C2);
  ep[0] = K2[0] * C2[0] / C2[2] + K2[2];
  ep[1] = K2[1] * C2[1] / C2[2] + K2[3];
  const so::SE3 T12 = so::se3_mul(T1w, Tw2);
  so::quat_to_matrix(T12.q, R12);
  for (int r = 0; r < 3; r++) t12[r] = T12.t[r];
  const float t12x[9] = {0.f, -t12[2], t12[1], t12[2], 0.f, -t12[0], -t12[1], t12[0], 0.f};   // SO3f::hat
  const float K1T[9] = {K1[0], 0.f, 0.f, 0.f, K1[1], 0.f, K1[2], K1[3], 1.f};
  const float K2m[9] = {K2[0], 0.f, K2[2], 0.f, K2[1], K2[3], 0.f, 0.f, 1.f};
  float K1Tinv[9], K2inv[9], A[9], B[9];
  so::mat3_inverse(K1T, K1Tinv);
  so::mat3_inverse(K2m, K2inv);
  so::mat3_mul(K1Tinv, t12x, A);
  so::mat3_mul(A, R12, B);
  so::mat3_mul(B, K2inv, F12);
}

int orc_search_for_triangulation(int N1, const orc_keypoint* kps1, const uint8_t* desc1, const int32_t* mp1, const int32_t* fv_nodes1,
                                 const int32_t* fv_off1, const int32_t* fv_feat1, int nn1, int N2, const orc_keypoint* kps2,
                                 const uint8_t* desc2, const int32_t* mp2, const int32_t* fv_nodes2, const int32_t* fv_off2,
                                 const int32_t* fv_feat2, int nn2, const float* F12, const float* ep, const float* scale_factors2,
                                 const float* level_sigma2_2, int coarse, int check_ori, int32_t* pairs) {
  int nmatches = 0;
  std::vector<uint8_t> vbMatched2(N2, 0);   // never set by the reference either (:878, :913)
  std::vector<int> vMatches12(N1, -1);
  std::vector<int> rotHist[kHisto];
  int a = 0, b = 0;
  while (a < nn1 && b < nn2) {
    if (fv_nodes1[a] == fv_nodes2[b]) {
      for (int i1 = fv_off1[a]; i1 < fv_off1[a + 1]; i1++) {
        const int idx1 = fv_feat1[i1];
        if (mp1[idx1] >= 0) continue;
        const orc_keypoint& kp1 = kps1[idx1];
        int bestDist = kThLow, bestIdx2 = -1;
        for (int i2 = fv_off2[b]; i2 < fv_off2[b + 1]; i2++) {
          const int idx2 = fv_feat2[i2];
          if (vbMatched2[idx2] || mp2[idx2] >= 0) continue;
          const int dist = descriptor_distance(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
          if (dist > kThLow || dist > bestDist) continue;
          const orc_keypoint& kp2 = kps2[idx2];
          const float distex = ep[0] - kp2.x, distey = ep[1] - kp2.y;
          if (distex * distex + distey * distey < 100 * scale_factors2[kp2.octave]) continue;
          bool ok = coarse != 0;
          if (!ok) {  // Pinhole::epipolarConstrain
            const float la = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
            const float lb = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
            const float lc = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
            const float num = la * kp2.x + lb * kp2.y + lc;
            const float den = la * la + lb * lb;
            if (den == 0) ok = false;
            else { const float dsqr = num * num / den; ok = dsqr < 3.84 * level_sigma2_2[kp2.octave]; }
          }
          if (ok) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
          vMatches12[idx1] = bestIdx2;
          nmatches++;
          if (check_ori) rotHist[rot_bin(kp1.angle, kps2[bestIdx2].angle)].push_back(idx1);
        }
      }
      a++; b++;
    } else if (fv_nodes1[a] < fv_nodes2[b]) {
      a = fv_lower_bound(fv_nodes1, nn1, a, fv_nodes2[b]);
    } else {
      b = fv_lower_bound(fv_nodes2, nn2, b, fv_nodes1[a]);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, kHisto, ind1, ind2, ind3);
    for (int i = 0; i < kHisto; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx : rotHist[i]) { vMatches12[idx] = -1; nmatches--; }
    }
  }
  int k = 0;
  for (int i = 0; i < N1; i++)
    if (vMatches12[i] >= 0) { pairs[2 * k] = i; pairs[2 * k + 1] = vMatches12[i]; k++; }
  return nmatches;
}

// Projection gates + window search shared by Fuse x2 and SearchByProjection(KF, Scw, ...) (see the file header).
// best_idx / best_dist per point (-1 / 256: nothing found or rejected by a gate); proj = (u, v, radius, level) with
// level -1 when a gate rejected the point.  skip[idx] != 0 removes a keyframe keypoint from every window.
void orc_project_search(int N, const orc_keypoint* kps, const uint8_t* desc, const float* bounds, const uint8_t* skip, const float* Tcw7,
                        const float* Ow, const float* K, int n, const float* P, const float* normal,
                        const float* min_dist, const float* max_dist, const uint8_t* pdesc, const uint8_t* valid, float th,
                        const float* scale_factors, float log_scale_factor, int n_levels, const float* gate_inv_sigma2, double gate,
                        int32_t* best_idx, int32_t* best_dist, float* proj) {
  orc_grid* g = orc_grid_create(kps, N, bounds[0], bounds[1], bounds[2], bounds[3]);
  std::vector<int> vIndices;
  const so::SE3 Tcw = load_se3(Tcw7);
  for (int i = 0; i < n; i++) {
    best_idx[i] = -1; best_dist[i] = 256;
    float* pr = proj ? proj + 4 * i : nullptr;
    if (pr) { pr[0] = -1.f; pr[1] = -1.f; pr[2] = 0.f; pr[3] = -1.f; }
    if (valid && !valid[i]) continue;
    const float* p = P + 3 * i;
    float c[3];
    so::se3_act(Tcw, p, c);   // p3Dc = Tcw * p3Dw (:424,527,1107,1267)
    if (c[2] < 0.0f) continue;
    const float u = K[0] * c[0] / c[2] + K[2], v = K[1] * c[1] / c[2] + K[3];
    if (!(u >= bounds[0] && u < bounds[1] && v >= bounds[2] && v < bounds[3])) continue;   // KeyFrame::IsInImage
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    const float PO[3] = {p[0] - Ow[0], p[1] - Ow[1], p[2] - Ow[2]};
    const float dist3D = so::norm3(PO);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const float dot = so::dot3(PO, normal + 3 * i);
    if (dot < 0.5 * dist3D) continue;
    const int nScale = predict_scale(max_dist[i], dist3D, log_scale_factor, n_levels);
    const float radius = th * scale_factors[nScale];
    if (pr) { pr[0] = u; pr[1] = v; pr[2] = radius; pr[3] = (float)nScale; }
    features_in_area(g, u, v, radius, -1, -1, vIndices);                       // KeyFrame::GetFeaturesInArea: no level filter
    int bestDist = 256, bestIdx = -1;
    for (int idx : vIndices) {
      if (skip && skip[idx]) continue;
      const int kpLevel = kps[idx].octave;
      if (kpLevel < nScale - 1 || kpLevel > nScale) continue;
      if (gate_inv_sigma2) {
        const float ex = u - kps[idx].x, ey = v - kps[idx].y;
        const float e2 = ex * ex + ey * ey;
        if (e2 * gate_inv_sigma2[kpLevel] > gate) continue;
      }
      const int dist = descriptor_distance(pdesc + 32 * (size_t)i, desc + 32 * (size_t)idx);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    best_idx[i] = bestIdx; best_dist[i] = bestDist;
  }
  orc_grid_destroy(g);
}

// ORBmatcher::Fuse(KF, Scw, vpPoints, th, vpReplacePoint): kf_mp (in/out) = KF map point ids per keypoint, kf_mp_bad = their
// isBad(); point i has id point_id[i]; replace[i] receives the id of the KF map point to be replaced (-1 none).
int orc_fuse_sim3(int N, const orc_keypoint* kps, const uint8_t* desc, const float* bounds, int32_t* kf_mp, const uint8_t* kf_mp_bad,
                  const float* Scw7, const float* K, int n, const int32_t* point_id,
                  const uint8_t* point_bad, const float* P, const float* normal, const float* min_dist, const float* max_dist,
                  const uint8_t* pdesc, float th, const float* scale_factors, float log_scale_factor, int n_levels, int32_t* replace) {
  so::SE3 Tcw;   // Tcw = SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale()); Ow = Tcw.inverse().translation() (:1245-1246)
  float Ow[3];
  so::sim3_to_se3(load_sim3(Scw7), Tcw, Ow);
  const float Tcw7[7] = {Tcw.q.x, Tcw.q.y, Tcw.q.z, Tcw.q.w, Tcw.t[0], Tcw.t[1], Tcw.t[2]};
  std::vector<uint8_t> valid(n, 1);
  std::vector<int32_t> already(kf_mp, kf_mp + N);   // spAlreadyFound = pKF->GetMapPoints() at entry
  std::sort(already.begin(), already.end());
  for (int i = 0; i < n; i++) {
    replace[i] = -1;
    if ((point_bad && point_bad[i]) || std::binary_search(already.begin(), already.end(), point_id[i])) valid[i] = 0;
  }
  std::vector<int32_t> bi(n), bd(n);
  orc_project_search(N, kps, desc, bounds, nullptr, Tcw7, Ow, K, n, P, normal, min_dist, max_dist, pdesc, valid.data(), th,
                     scale_factors, log_scale_factor, n_levels, nullptr, 0.0, bi.data(), bd.data(), nullptr);
  int nFused = 0;
  std::vector<uint8_t> fresh(N, 0);   // keypoints that received a point in this call (never bad)
  for (int i = 0; i < n; i++) {
    if (!valid[i] || bi[i] < 0 || bd[i] > kThLow) continue;
    const int pMPinKF = kf_mp[bi[i]];
    if (pMPinKF >= 0) {
      if (fresh[bi[i]] || !(kf_mp_bad && kf_mp_bad[bi[i]])) replace[i] = pMPinKF;
    } else {
      kf_mp[bi[i]] = point_id[i];
      fresh[bi[i]] = 1;
    }
    nFused++;
  }
  return nFused;
}

// ORBmatcher::SearchByProjection(KF, Scw, vpPoints, vpMatched, th, ratioHamming) (:395-496): matched (in/out) = point id
// per KF keypoint (-1 = NULL); a keypoint matched earlier (at entry or in this call) is skipped by later points.
int orc_search_by_projection_sim3(int N, const orc_keypoint* kps, const uint8_t* desc, const float* bounds, int32_t* matched,
                                  const float* Scw7, const float* K, int n, const int32_t* point_id,
                                  const uint8_t* point_bad, const float* P, const float* normal, const float* min_dist,
                                  const float* max_dist, const uint8_t* pdesc, int th, float ratioHamming, const float* scale_factors,
                                  float log_scale_factor, int n_levels) {
  so::SE3 Tcw;   // (:403-404)
  float Ow[3];
  so::sim3_to_se3(load_sim3(Scw7), Tcw, Ow);
  const float Tcw7[7] = {Tcw.q.x, Tcw.q.y, Tcw.q.z, Tcw.q.w, Tcw.t[0], Tcw.t[1], Tcw.t[2]};
  std::vector<int32_t> already(matched, matched + N);
  std::sort(already.begin(), already.end());
  int nmatches = 0;
  std::vector<uint8_t> skip(N), one(1, 1);
  for (int i = 0; i < n; i++) {
    if ((point_bad && point_bad[i]) || (point_id[i] >= 0 && std::binary_search(already.begin(), already.end(), point_id[i]))) continue;
    for (int j = 0; j < N; j++) skip[j] = matched[j] >= 0;
    int32_t bi, bd;
    orc_project_search(N, kps, desc, bounds, skip.data(), Tcw7, Ow, K, 1, P + 3 * i, normal + 3 * i, min_dist + i, max_dist + i,
                       pdesc + 32 * (size_t)i, one.data(), (float)th, scale_factors, log_scale_factor, n_levels, nullptr, 0.0, &bi, &bd,
                       nullptr);
    if (bi >= 0 && bd <= kThLow * ratioHamming) { matched[bi] = point_id[i]; nmatches++; }
  }
  return nmatches;
}

// ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (:1347-1551).  Per-keypoint map point data: mpN[i] id (-1 NULL),
// badN[i], PN (3 per keypoint), minN / maxN (mfMin/MaxDistance), mdescN (32 per keypoint).  idx_in_kf2[i] =
// get<0>(vpMatches12[i]->GetIndexInKeyFrame(pKF2)) for the entries of matches12 that are set at entry.  Poses and S12 are
// Sophus objects (7 floats each); S21 = S12.inverse() (sim3.hpp:129-132); p3Dc2 = S21 * (T1w * p3Dw) with the quaternion
// actions of se3.hpp:319-324 / rxso3.hpp:265-273.
namespace {
void sim3_direction(int Na, const int32_t* mpa, const uint8_t* bada, const uint8_t* already_a, const float* Pa, const float* mina,
                    const float* maxa, const uint8_t* mdesca, const so::SE3& Taw, const so::Sim3& Sba,
                    int Nb, const orc_keypoint* kpsb, const uint8_t* descb, const float* bounds, const float* K, float th,
                    const float* scale_factors, float log_scale_factor, int n_levels, std::vector<int>& vnMatch) {
  orc_grid* g = orc_grid_create(kpsb, Nb, bounds[0], bounds[1], bounds[2], bounds[3]);
  std::vector<int> vIndices;
  vnMatch.assign(Na, -1);
  for (int i = 0; i < Na; i++) {
    if (mpa[i] < 0 || already_a[i]) continue;
    if (bada && bada[i]) continue;
    const float* p = Pa + 3 * i;
    float c1[3], c2[3];
    so::se3_act(Taw, p, c1);      // p3Dc1 = T1w * p3Dw   (:1394)
    so::sim3_act(Sba, c1, c2);    // p3Dc2 = S21 * p3Dc1  (:1395)
    if (c2[2] < 0.0) continue;
    const float invz = 1.0 / c2[2];
    const float x = c2[0] * invz, y = c2[1] * invz;
    const float u = K[0] * x + K[2], v = K[1] * y + K[3];
    if (!(u >= bounds[0] && u < bounds[1] && v >= bounds[2] && v < bounds[3])) continue;
    const float maxDistance = 1.2f * maxa[i], minDistance = 0.8f * mina[i];
    const float dist3D = so::norm3(c2);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const int nScale = predict_scale(maxa[i], dist3D, log_scale_factor, n_levels);
    const float radius = th * scale_factors[nScale];
    features_in_area(g, u, v, radius, -1, -1, vIndices);
    int bestDist = INT32_MAX, bestIdx = -1;
    for (int idx : vIndices) {
      if (kpsb[idx].octave < nScale - 1 || kpsb[idx].octave > nScale) continue;
      const int dist = descriptor_distance(mdesca + 32 * (size_t)i, descb + 32 * (size_t)idx);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= 100) vnMatch[i] = bestIdx;   // TH_HIGH
  }
  orc_grid_destroy(g);
}
}  // namespace

int orc_search_by_sim3(int N1, const orc_keypoint* kps1, const uint8_t* desc1, const int32_t* mp1, const uint8_t* bad1, const float* P1,
                       const float* min1, const float* max1, const uint8_t* mdesc1, const float* T1w7, int N2,
                       const orc_keypoint* kps2, const uint8_t* desc2, const int32_t* mp2, const uint8_t* bad2, const float* P2,
                       const float* min2, const float* max2, const uint8_t* mdesc2, const float* T2w7,
                       const float* bounds, const float* K, const float* S12_7, float th,
                       const float* scale_factors, float log_scale_factor, int n_levels, int32_t* matches12, const int32_t* idx_in_kf2) {
  const so::SE3 T1w = load_se3(T1w7), T2w = load_se3(T2w7);
  const so::Sim3 S12 = load_sim3(S12_7), S21 = so::sim3_inverse(S12);
  std::vector<uint8_t> am1(N1, 0), am2(N2, 0);
  for (int i = 0; i < N1; i++)
    if (matches12[i] >= 0) {
      am1[i] = 1;
      const int idx2 = idx_in_kf2 ? idx_in_kf2[i] : -1;
      if (idx2 >= 0 && idx2 < N2) am2[idx2] = 1;
    }
  std::vector<int> vnMatch1, vnMatch2;
  sim3_direction(N1, mp1, bad1, am1.data(), P1, min1, max1, mdesc1, T1w, S21, N2, kps2, desc2, bounds, K, th, scale_factors,
                 log_scale_factor, n_levels, vnMatch1);
  sim3_direction(N2, mp2, bad2, am2.data(), P2, min2, max2, mdesc2, T2w, S12, N1, kps1, desc1, bounds, K, th, scale_factors,
                 log_scale_factor, n_levels, vnMatch2);
  int nFound = 0;
  for (int i1 = 0; i1 < N1; i1++) {
    const int idx2 = vnMatch1[i1];
    if (idx2 >= 0 && vnMatch2[idx2] == i1) { matches12[i1] = mp2[idx2]; nFound++; }
  }
  return nFound;
}

// ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (:1750-1860, relocalisation): the map
// points of pKF (per keypoint: id -1 = NULL, bad flag, position, distances, descriptor) are projected with the current
// frame's pose; already (sorted ids) = sAlreadyFound; mp_c (in/out) = CurrentFrame.mvpMapPoints as ids.
int orc_search_by_projection_reloc(int Nc, const orc_keypoint* kps_c, const uint8_t* desc_c, int32_t* mp_c, const float* bounds,
                                   const float* Tcw7, const float* K, int Nk, const orc_keypoint* kps_k,
                                   const int32_t* mp_k, const uint8_t* bad_k, const float* P, const float* min_dist, const float* max_dist,
                                   const uint8_t* pdesc, const int32_t* already, int n_already, float th, int ORBdist,
                                   const float* scale_factors, float log_scale_factor, int n_levels, int check_ori) {
  int nmatches = 0;
  std::vector<int> rotHist[kHisto];
  orc_grid* g = orc_grid_create(kps_c, Nc, bounds[0], bounds[1], bounds[2], bounds[3]);
  std::vector<int> vIndices2;
  const so::SE3 Tcw = load_se3(Tcw7), Twc = so::se3_inverse(Tcw);   // Ow = Tcw.inverse().translation() (:1754-1755)
  const float* Ow = Twc.t;
  for (int i = 0; i < Nk; i++) {
    if (mp_k[i] < 0) continue;
    if ((bad_k && bad_k[i]) || std::binary_search(already, already + n_already, mp_k[i])) continue;
    const float* p = P + 3 * i;
    float c[3];
    so::se3_act(Tcw, p, c);   // x3Dc = Tcw * x3Dw (:1772)
    const float u = K[0] * c[0] / c[2] + K[2], v = K[1] * c[1] / c[2] + K[3];
    if (u < bounds[0] || u > bounds[1]) continue;
    if (v < bounds[2] || v > bounds[3]) continue;
    const float PO[3] = {p[0] - Ow[0], p[1] - Ow[1], p[2] - Ow[2]};
    const float dist3D = so::norm3(PO);
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const int nPredictedLevel = predict_scale(max_dist[i], dist3D, log_scale_factor, n_levels);
    const float radius = th * scale_factors[nPredictedLevel];
    features_in_area(g, u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1, vIndices2);
    if (vIndices2.empty()) continue;
    int bestDist = 256, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (mp_c[i2] >= 0) continue;
      const int dist = descriptor_distance(pdesc + 32 * (size_t)i, desc_c + 32 * (size_t)i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= ORBdist) {
      mp_c[bestIdx2] = mp_k[i];
      nmatches++;
      if (check_ori) rotHist[rot_bin(kps_k[i].angle, kps_c[bestIdx2].angle)].push_back(bestIdx2);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, kHisto, ind1, ind2, ind3);
    for (int i = 0; i < kHisto; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int idx : rotHist[i]) { mp_c[idx] = -1; nmatches--; }
  }
  orc_grid_destroy(g);
  return nmatches;
}

// Pose helpers for the tests (the reference's own derivations, sophus_oracle.h): all 7-float poses
void orc_se3_inverse(const float* T7, float* out7) {
  const so::SE3 r = so::se3_inverse(load_se3(T7));
  out7[0] = r.q.x; out7[1] = r.q.y; out7[2] = r.q.z; out7[3] = r.q.w; out7[4] = r.t[0]; out7[5] = r.t[1]; out7[6] = r.t[2];
}
void orc_pose_matrices(const float* Tcw7, float* Rcw, float* tcw, float* Ow) {   // Frame::UpdatePoseMatrices, Frame.cc:553-559
  const so::SE3 T = load_se3(Tcw7), Twc = so::se3_inverse(T);
  so::quat_to_matrix(T.q, Rcw);
  for (int r = 0; r < 3; r++) { tcw[r] = T.t[r]; Ow[r] = Twc.t[r]; }
}
void orc_sim3_to_se3(const float* S7, float* Tcw7, float* Ow) {
  so::SE3 T;
  so::sim3_to_se3(load_sim3(S7), T, Ow);
  Tcw7[0] = T.q.x; Tcw7[1] = T.q.y; Tcw7[2] = T.q.z; Tcw7[3] = T.q.w; Tcw7[4] = T.t[0]; Tcw7[5] = T.t[1]; Tcw7[6] = T.t[2];
}
void orc_sim3_inverse(const float* S7, float* out7) {
  const so::Sim3 r = so::sim3_inverse(load_sim3(S7));
  out7[0] = r.q.x; out7[1] = r.q.y; out7[2] = r.q.z; out7[3] = r.q.w; out7[4] = r.t[0]; out7[5] = r.t[1]; out7[6] = r.t[2];
}
void orc_se3_act(const float* T7, const float* p, int n, float* out) {
  const so::SE3 T = load_se3(T7);
  for (int i = 0; i < n; i++) so::se3_act(T, p + 3 * i, out + 3 * i);
}
void orc_sim3_act(const float* S7, const float* p, int n, float* out) {
  const so::Sim3 S = load_sim3(S7);
  for (int i = 0; i < n; i++) so::sim3_act(S, p + 3 * i, out + 3 * i);
}
float orc_logf(float x) { return so::logf_spec(x); }

// ---------------------------------------------------------------------------------------------------------------------
// Frame::UndistortKeyPoints / ComputeImageBounds (Frame.cc:791-848).  cv::undistortPoints is OpenCV (not in /root/reference):
// restated from OpenCV 4.x modules/calib3d/src/undistort.dispatch.cpp, cvUndistortPointsInternal, as the public
// cv::undistortPoints(src, dst, K, D, R = noArray(), P) runs it -- TermCriteria(MAX_ITER, 5, 0.01): exactly five passes of
//   r2 = x^2 + y^2;  icdist = (1 + ((k7 r2 + k6) r2 + k5) r2) / (1 + ((k4 r2 + k1) r2 + k0) r2);
//   dX = 2 k2 x y + k3 (r2 + 2 x^2) + k8 r2 + k9 r2^2;  dY = k2 (r2 + 2 y^2) + 2 k3 x y + k10 r2 + k11 r2^2;
//   x = (x0 - dX) icdist;  y = (y0 - dY) icdist
// in double on k = (k1, k2, p1, p2, k3, 0 ...) converted from CV_32F, then (x, y, 1) through P * R (here K * I) and one
// rounding to float.  PARITY UNPINNED: OpenCV is not available to run against.
static void undistort_one(const float* cam, float uf, float vf, float* xo, float* yo) {
  double A[3][3] = {{(double)cam[0], 0, (double)cam[2]}, {0, (double)cam[1], (double)cam[3]}, {0, 0, 1}};
  double k[14] = {0};
  for (int i = 0; i < 5; i++) k[i] = (double)cam[4 + i];
  double RR[3][3];
  for (int r = 0; r < 3; r++)          // cvMatMul(&_PP, &_RR, &_RR) with RR = I, PP = K
    for (int c = 0; c < 3; c++) {
      double acc = 0;
      for (int m = 0; m < 3; m++) acc += A[r][m] * (m == c ? 1.0 : 0.0);
      RR[r][c] = acc;
    }
  const double fx = A[0][0], fy = A[1][1], ifx = 1. / fx, ify = 1. / fy, cx = A[0][2], cy = A[1][2];
  double x = uf, y = vf;
  const double u = x, v = y;
  x = (x - cx) * ifx;
  y = (y - cy) * ify;
  const double x0 = x, y0 = y;     // tilt model off: invMatTilt = I, invProj = 1
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
    if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
    const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
    const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
  const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
  const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
  *xo = (float)(xx * ww);
  *yo = (float)(yy * ww);
}
// cam = {fx, fy, cx, cy, k1, k2, p1, p2, k3}; xy in / out: n x 2 floats.  k1 == 0: copy (Frame.cc:792-795)
void orc_undistort_points(const float* cam, const float* xy_in, int n, float* xy_out) {
  for (int i = 0; i < n; i++) {
    if (cam[4] == 0.0f) { xy_out[2 * i] = xy_in[2 * i]; xy_out[2 * i + 1] = xy_in[2 * i + 1]; }
    else undistort_one(cam, xy_in[2 * i], xy_in[2 * i + 1], &xy_out[2 * i], &xy_out[2 * i + 1]);
  }
}
// Frame::ComputeImageBounds (Frame.cc:820-848): out = {mnMinX, mnMaxX, mnMinY, mnMaxY}
void orc_image_bounds(const float* cam, int cols, int rows, float* out) {
  if (cam[4] != 0.0f) {
    const float in[8] = {0.0f, 0.0f, (float)cols, 0.0f, 0.0f, (float)rows, (float)cols, (float)rows};
    float m[8];
    orc_undistort_points(cam, in, 4, m);
    out[0] = std::min(m[0], m[4]); out[1] = std::max(m[2], m[6]);
    out[2] = std::min(m[1], m[3]); out[3] = std::max(m[5], m[7]);
  } else {
    out[0] = 0.0f; out[1] = (float)cols; out[2] = 0.0f; out[3] = (float)rows;
  }
}

}  // extern "C"
