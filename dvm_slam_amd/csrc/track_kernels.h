// dvm_slam_amd/csrc/track_kernels.h -- launchers of track_kernels.hip (the device chain of dvm_track_finish).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "match_kernels.h" // FrameView
#include "orb_kernels.h"   // dvm_keypoint_pod

namespace dvm {
size_t track_claims_lds(int kp_cap, int nq);
// what k_track_claims needs to search a query's window again on the device: the grid slot the ranked lists came from + the query arrays
struct TrackRequery {
  FrameView F;          // F.skp == nullptr: no device re-search (res[1] then reports the exhausted query to the caller)
  const uint8_t* qdesc; const float *qx, *qy, *qr; const int32_t *qmin, *qmax;
};
// a batch of frames through the same launches: frame b's per-query arrays at b * qstride elements (nq_arr[b] of them), its keypoints at
// b * kps_stride; count = 1, nq_arr = nullptr: one frame (nq given directly)
struct TrackBatch { int count; int qstride; int64_t kps_stride; const int32_t* nq_arr; };
void launch_track_claims(hipStream_t s, const uint32_t* ranked, const uint8_t* q_claims, const float* q_angle, int nq, const TrackRequery& rq,
                         const dvm_keypoint_pod* kps, const int32_t* d_n, int kp_cap, int th_high, int check_ori, int32_t* assign, int32_t* res,
                         int32_t* assign_host, int32_t* res_host, const TrackBatch& B);
void launch_track_gather(hipStream_t s, const int32_t* assign, const dvm_keypoint_pod* kps_un, const int32_t* d_n, int kp_cap, const float* q_pos,
                         const float* inv_sigma2, int nlevels, double* Xw, double* obs, double* info, int32_t* edge_kp, int32_t* n_edges,
                         const int32_t* res, int min_matches, int32_t* n_edges_host, const TrackBatch& B);
void launch_track_finish(hipStream_t s, int32_t* assign, const int32_t* d_n, int kp_cap, const int32_t* edge_kp, const int32_t* n_edges,
                         const uint8_t* edge_outlier, const uint8_t* q_claims, uint8_t* outlier, int32_t* out, const int32_t* res, const TrackBatch& B);
}  // namespace dvm
