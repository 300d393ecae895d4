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
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {
const int kGridCols = 64, kGridRows = 48;  // Frame.h:44-45

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
    for (int r = 0; r < 3; r++) Pc[r] = (F->Rcw[3 * r] * p[0] + F->Rcw[3 * r + 1] * p[1] + F->Rcw[3 * r + 2] * p[2]) + F->tcw[r];
    const float Pc_dist = std::sqrt(Pc[0] * Pc[0] + Pc[1] * Pc[1] + Pc[2] * Pc[2]);
    const float PcZ = Pc[2];
    const float invz = 1.0f / PcZ;
    if (PcZ < 0.0f) continue;
    const float u = F->fx * Pc[0] / Pc[2] + F->cx, v = F->fy * Pc[1] / Pc[2] + F->cy;
    if (u < F->min_x || u > F->max_x) continue;
    if (v < F->min_y || v > F->max_y) continue;
    o.proj_x = u; o.proj_y = v;
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    const float PO[3] = {p[0] - F->Ow[0], p[1] - F->Ow[1], p[2] - F->Ow[2]};
    const float dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    if (dist < minDistance || dist > maxDistance) continue;
    const float* Pn = normal + 3 * i;
    const float viewCos = (PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) / dist;
    if (viewCos < viewing_cos_limit) continue;
    const float ratio = max_dist[i] / dist;
    int nScale = (int)std::ceil(std::log(ratio) / F->log_scale_factor);
    if (nScale < 0) nScale = 0; else if (nScale >= F->n_levels) nScale = F->n_levels - 1;
    o.in_view = 1; o.proj_xr = u - F->bf * invz; o.depth = Pc_dist; o.level = nScale; o.view_cos = viewCos;
  }
}

}  // extern "C"
