// dvm_slam_amd/csrc/match_kernels.h -- launchers of the matching kernels (match_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orb_kernels.h"
#include "undistort_f64.h"

struct dvm_frame;

namespace dvm {

constexpr int kFrameCap = 8192;  // keypoints per frame slot (13-bit index inside the sort key)

struct dvm_match_pod {  // == dvm_match
  int32_t best_idx, best_dist, second_dist;
  int16_t best_level, second_level;
};

// Device view of a set of frame slots, each `cap` keypoints, sorted in GetFeaturesInArea order.
constexpr int kGridCols = 64, kGridRows = 48;  // Frame.h:44-45 (FRAME_GRID_COLS / FRAME_GRID_ROWS)

struct FrameView {
  float4* skp;           // (x, y, octave bits, cell bits) per sorted position
  int32_t* sidx;         // original keypoint index
  uint8_t* sdesc;        // 32 B descriptors, sorted
  int32_t* cellx_start;  // [65] first sorted position with grid column >= c
  int32_t* n_sorted;     // keypoints inside the grid
  int32_t* n_total;      // keypoints given
  int32_t* n_overflow;   // one per handle: device-side counts (d_n) that exceeded cap and were clamped (sticky; dvm_frame_overflows)
  int32_t cap;
  float minX, minY, wInv, hInv;
  __host__ __device__ FrameView slot(int s) const {
    FrameView v = *this;
    v.skp += (int64_t)s * cap;
    v.sidx += (int64_t)s * cap;
    v.sdesc += (int64_t)s * cap * 32;
    v.cellx_start += (int64_t)s * 80;
    v.n_sorted += s;
    v.n_total += s;
    return v;
  }
};

// Query source of the frame-to-frame search: pair i matches the keypoints of frame i-1 (arrays with
// a per-frame stride) -- or of the carry frame for pair 0 -- against train slot first_slot + i.
struct PairQueries {
  const dvm_keypoint_pod* kps;
  const uint8_t* desc;
  const int32_t* n;
  int64_t kps_stride, desc_stride;
  const dvm_keypoint_pod* carry_kps;
  const uint8_t* carry_desc;
  const int32_t* carry_n;
  int32_t* n_out;  // [count] number of queries of each pair (may be null)
  int32_t cap;
};

struct FrustumFrame {  // == dvm_frustum_frame
  float Rcw[9], tcw[3], Ow[3], fx, fy, cx, cy, min_x, max_x, min_y, max_y, bf, log_scale_factor;
  int32_t n_levels;
};
struct TrackPoint {  // == dvm_track_point
  float proj_x, proj_y, proj_xr, depth, view_cos;
  int32_t level, in_view;
};
struct TriPair {  // == dvm_tri_pair
  double cos_parallax_max;
  float K1[4], K2[4], T1w[12], T2w[12], Ow1[3], Ow2[3];
  float ratio_factor, th_far;
  int32_t far_points, n_levels;
};
struct ProjectCam {  // == dvm_kf_camera (+ th)
  float q[4], t[3];       // Tcw as Sophus::SE3f stores it: unit quaternion (x, y, z, w) + translation
  float Ow[3], fx, fy, cx, cy, min_x, max_x, min_y, max_y, log_scale_factor;
  int32_t n_levels;
  int32_t sim3_pair;      // 1: SearchBySim3 form, p' = S2 * (Tcw * p), distance = |p'|, no viewing-angle test; 2: relocalisation form
  float q2[4], t2[3];     // S2 as Sophus::Sim3f stores it: RxSO3 quaternion (scale = |q2|^2) + translation
  float th;
};
struct Projection {  // == dvm_projection
  float u, v, radius;
  int32_t level;
};
struct TriGeom {
  float F12[9], ep[2];
  int32_t coarse, th_low;
};
void launch_project_search(hipStream_t s, const FrameView& F, int slot, const uint8_t* skip, const ProjectCam& C, const float* P,
                           const float* normal, const float* min_dist, const float* max_dist, const uint8_t* desc,
                           const uint8_t* valid, int n, const float* scale_factors, const float* gate_inv_sigma2, double gate,
                           dvm_match_pod* out, Projection* proj);
void launch_match_triangulation(hipStream_t s, const uint8_t* desc1, const dvm_keypoint_pod* kps1, const int32_t* qidx, int nq,
                                const uint8_t* desc2, const dvm_keypoint_pod* kps2, const int32_t* off, const int32_t* cand,
                                const TriGeom& G, const float* scale_factors2, const float* level_sigma2_2, int32_t* best_idx,
                                int32_t* best_dist);
void launch_undistort_keypoints(hipStream_t s, const dvm_undistort::Camera& cam, const float* in, float* out, int n);
void launch_triangulate_matches(hipStream_t s, const TriPair& P, const dvm_keypoint_pod* kps1, int n1, const dvm_keypoint_pod* kps2, int n2,
                                const int32_t* pairs, int n, const float* sigma2_1, const float* sigma2_2, const float* sf1,
                                const float* sf2, float* x3D, int32_t* status);
void launch_is_in_frustum(hipStream_t s, const FrustumFrame& F, const float* P, const float* normal, const float* min_dist,
                          const float* max_dist, int n, float cos_limit, TrackPoint* out);
void launch_frame_build(hipStream_t s, const dvm_keypoint_pod* kps, int64_t kps_stride, const uint8_t* desc,
                        int64_t desc_stride, int n, const int32_t* d_n, const FrameView& F, int first_slot, int count);
void launch_match_window(hipStream_t s, const FrameView& F, int slot, const uint8_t* skip, const uint8_t* qdesc,
                         const float* qx, const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax,
                         int nq, const int32_t* d_nq, int grid_q, dvm_match_pod* out, int32_t* second_idx = nullptr);
// the device view of a grid handle (capi.cpp; for the kernels of other translation units that search the same grid: track.cpp)
FrameView frame_view_of(const struct ::dvm_frame* f);
void launch_match_window_ranked(hipStream_t s, const FrameView& F, int slot, const uint8_t* skip, const uint8_t* qdesc, const float* qx,
                                const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax, int nq, uint32_t* ranked);
void launch_match_window_ranked_batch(hipStream_t s, const FrameView& F, int first_slot, int count, const uint8_t* skip, const int32_t* skip_on,
                                      int skip_stride, const uint8_t* qdesc, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                                      const int32_t* qmax, const int32_t* nq_arr, int qstride, uint32_t* ranked);
void launch_match_frames(hipStream_t s, const FrameView& F, int first_slot, int count, const PairQueries& pq, float th,
                         const float* scale_factors, int nlevels, dvm_match_pod* out, int64_t out_stride);
void launch_match_lists(hipStream_t s, const uint8_t* tdesc, const uint8_t* qdesc, const int32_t* off, const int32_t* cand,
                        int nq, dvm_match_pod* out);
void launch_bowdb_query(hipStream_t s, const int64_t* kf_off, const int32_t* kf_len, const int32_t* live, int n_live, const int32_t* ids,
                        const double* vals, const int32_t* qids, const double* qvals, int nq, int32_t* common, int32_t* first_word, float* score);
void launch_hamming_matrix(hipStream_t s, const uint8_t* A, int nA, const uint8_t* B, int nB, uint16_t* D);

void launch_distinctive(hipStream_t s, const uint8_t* desc, const int32_t* off, int npts, int32_t* best_idx, int32_t* best_median);
void launch_vocab_transform(hipStream_t s, const int32_t* child_off, const int32_t* children, const uint8_t* node_desc,
                            const double* weight, const int32_t* word_id, int L, const uint8_t* feat, int n, int levelsup,
                            int32_t* out_word, int32_t* out_node, double* out_weight);
}  // namespace dvm
