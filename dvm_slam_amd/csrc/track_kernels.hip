// dvm_slam_amd/csrc/track_kernels.hip -- the device side of dvm_track_begin / dvm_track_finish (include/dvmslam_hip.h): what the
// reference's Tracking::TrackWithMotionModel does between the window search and the pose it ends on, as kernels of ONE stream chain
// behind the extraction of the frame (no host step in between):
//   k_track_claims   ORBmatcher::SearchByProjection(CurrentFrame, LastFrame): the sequential epilogue of src/ORBmatcher.cc:1613-1664
//                    (a keypoint taken by an earlier query is skipped by the later ones) and the rotation histogram :1652-1663,
//                    :1730-1745, replayed from the ranked candidate lists of k_match_window_ranked
//   k_track_gather   Optimizer::PoseOptimization's edge list (src/Optimizer.cc:768-838): the matched keypoints in keypoint order
//   k_track_finish   the outlier flags back in keypoint order, Tracking.cc:2636-2660 (outlier matches dropped, nmatchesMap)
// The pose itself is k_pose_optimize (ba_kernels.hip), launched between the last two.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "match_kernels.h"
#include "track_kernels.h"

namespace dvm {

namespace {
constexpr int kHisto = 30;   // HISTO_LENGTH, ORBmatcher.cc:38
}

// One wave.  Queries are decided in blocks of 64, one per lane.  Inside a block a lane's choice depends on what the lanes before it
// take: every round each undecided lane proposes its first candidate that is still free; a lane whose proposal is also proposed by
// an EARLIER undecided lane that will take it (owner[] = smallest such lane) is blocked; the lanes in front of the first blocked lane
// are final and commit, the rest propose again.  The first undecided lane is never blocked, so every round commits at least one.
//   ranked[q][0..3]  dist << 16 | keypoint, best first, dist >= 256 = end of the list (k_match_window_ranked)
//   q_claims[q]      the query's map point has Observations() > 0: its match takes the keypoint (:1620-1622)
//   q_angle[q]       LastFrame.mvKeysUn[i].angle
//   RQ               the grid the lists were ranked on and the query arrays: a query that finds all four ranked candidates taken while
//                    its list may go on has its window scanned again by the whole wave at its turn (on the dense bench stream
//                    about one query in 60: every frame has some)
// Outputs: assign[j] = the query matched to keypoint j at the end of the call or -1; res[0] = nmatches, res[1] = 1 if such a query could
// not be searched again here (RQ.F.skp == nullptr: the caller then repeats the epilogue on the host), res[2] = matches before the
// rotation check, res[3] = queries searched again.
// blockIdx.x = frame of a batch (dvm_track_finish_batch: K agents' frames in one chain): frame b's per-query arrays lie at b * B.qstride
// elements, its keypoints at b * B.kps_stride, its grid in slot b, its results at b * kp_cap / b * 8; a single frame is the batch of one.
__global__ void __launch_bounds__(256) k_track_claims(const uint32_t* __restrict__ ranked, const uint8_t* __restrict__ q_claims,
                                                      const float* __restrict__ q_angle, int nq, TrackRequery RQ, const dvm_keypoint_pod* __restrict__ kps,
                                                      const int32_t* __restrict__ d_n, int kp_cap, int th_high, int check_ori,
                                                      int32_t* __restrict__ assign, int32_t* __restrict__ res, int32_t* __restrict__ assign_host,
                                                      int32_t* __restrict__ res_host, TrackBatch B) {
  extern __shared__ __attribute__((aligned(16))) uint8_t track_smem[];
  {
    const int b = blockIdx.x;
    if (B.nq_arr) nq = B.nq_arr[b];
    const size_t qo = (size_t)b * B.qstride;
    ranked += qo * 4; q_claims += qo; q_angle += qo;
    RQ.qdesc += qo * 32; RQ.qx += qo; RQ.qy += qo; RQ.qr += qo; RQ.qmin += qo; RQ.qmax += qo;
    if (RQ.F.skp) RQ.F = RQ.F.slot(b);
    kps += (size_t)b * B.kps_stride; d_n += b;
    assign += (size_t)b * kp_cap; assign_host += (size_t)b * kp_cap; res += 8 * b; res_host += 8 * b;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = min(*d_n, kp_cap);
  const int nq_pad = (nq + 63) & ~63;
  uint4* s_keys = reinterpret_cast<uint4*>(track_smem);                    // [nq_pad] the ranked lists
  int32_t* s_assign = reinterpret_cast<int32_t*>(s_keys + nq_pad);         // [kp_cap]
  uint32_t* s_owner = reinterpret_cast<uint32_t*>(s_assign + kp_cap);      // [kp_cap]
  uint32_t* s_qres = s_owner + kp_cap;                                     // [nq_pad]: keypoint | bin << 16, or 0xFFFFFFFF
  uint8_t* s_claimed = reinterpret_cast<uint8_t*>(s_qres + nq_pad);        // [kp_cap]
  uint8_t* s_qcl = s_claimed + kp_cap;                                     // [nq_pad] the query takes its keypoint
  __shared__ int s_rot[kHisto];
  __shared__ int s_ind[3];
  __shared__ int s_cnt[4];
  // ---- parallel prologue (four waves): everything the sequential part reads comes to LDS -- a global load inside a round would put its
  // latency on the chain of ~50 rounds per frame
  for (int j = tid; j < kp_cap; j += 256) { s_assign[j] = -1; s_owner[j] = 0xFFFFFFFFu; s_claimed[j] = 0; }
  for (int q = tid; q < nq_pad; q += 256) {
    s_qres[q] = 0xFFFFFFFFu;
    s_keys[q] = q < nq ? *reinterpret_cast<const uint4*>(ranked + 4 * (size_t)q) : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    s_qcl[q] = q < nq ? q_claims[q] : 0;
  }
  if (tid < kHisto) s_rot[tid] = 0;
  if (tid < 4) s_cnt[tid] = 0;
  __syncthreads();
  // ---- sequential part: wave 0 alone (the other waves wait at the barrier below; inside one wave LDS operations complete in order,
  // so the rounds need no barrier)
  int exhausted_any = 0, n_requeried = 0, n_rounds = 0;
  if (wave == 0) {
    for (int q0 = 0; q0 < nq; q0 += 64) {
      const int q = q0 + lane;
      bool decided = q >= nq;
      const uint4 k4 = s_keys[q];
      const uint32_t key0 = k4.x, key1 = k4.y, key2 = k4.z, key3 = k4.w;   // (selected by compares: an indexed array would live in scratch)
      const bool claims = !decided && s_qcl[q] != 0;
      // list entries: distance >= 256 ends the list; an index >= N cannot occur (the grid holds N keypoints), guarded all the same
      const int i0 = (int)(key0 & 0xFFFFu), i1 = (int)(key1 & 0xFFFFu), i2 = (int)(key2 & 0xFFFFu), i3 = (int)(key3 & 0xFFFFu);
      const bool v0 = (key0 >> 16) < 256u && i0 < N, v1 = v0 && (key1 >> 16) < 256u && i1 < N, v2 = v1 && (key2 >> 16) < 256u && i2 < N,
                 v3 = v2 && (key3 >> 16) < 256u && i3 < N;
      while (__ballot(!decided)) {
        n_rounds++;
        // first candidate of the list no earlier query has taken (the four flags are read side by side: one LDS latency per round)
        const bool f0 = v0 && !s_claimed[v0 ? i0 : 0], f1 = v1 && !s_claimed[v1 ? i1 : 0], f2 = v2 && !s_claimed[v2 ? i2 : 0],
                   f3 = v3 && !s_claimed[v3 ? i3 : 0];
        int prop = -1, pdist = 256;
        if (!decided) {
          if (f0) { prop = i0; pdist = (int)(key0 >> 16); }
          else if (f1) { prop = i1; pdist = (int)(key1 >> 16); }
          else if (f2) { prop = i2; pdist = (int)(key2 >> 16); }
          else if (f3) { prop = i3; pdist = (int)(key3 >> 16); }
        }
        bool exhausted = !decided && v3 && prop < 0;     // all four taken, the list may go on
        // A query whose four ranked candidates are all taken waits until every query in front of it is final -- it blocks the lanes behind
        // it meanwhile --, then the whole wave scans its window again, skipping what is taken by now: the reference's loop of
        // ORBmatcher.cc:1613-1650 at that query's turn (smallest (distance, scan position) among the free ones)
        const unsigned long long und = __ballot(!decided);
        const int first_und = (int)__builtin_ctzll(und);
        const bool requery = ((__ballot(exhausted) >> first_und) & 1ull) != 0ull;
        if (requery && RQ.F.skp) {
          const int qf = q0 + first_und;
          const float x = RQ.qx[qf], y = RQ.qy[qf], r = RQ.qr[qf];
          const int minLevel = RQ.qmin[qf], maxLevel = RQ.qmax[qf];
          const FrameView& F = RQ.F;
          uint32_t best = (256u << 16) | 0xFFFFu;
          const int nMinCellX = max(0, (int)floorf((x - F.minX - r) * F.wInv));
          const int nMaxCellX = min(kGridCols - 1, (int)ceilf((x - F.minX + r) * F.wInv));
          const int nMinCellY = max(0, (int)floorf((y - F.minY - r) * F.hInv));
          const int nMaxCellY = min(kGridRows - 1, (int)ceilf((y - F.minY + r) * F.hInv));
          const bool empty = nMinCellX >= kGridCols || nMaxCellX < 0 || nMinCellY >= kGridRows || nMaxCellY < 0;
          if (!empty && nMinCellX <= nMaxCellX) {
            const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
            const uint32_t* qd = reinterpret_cast<const uint32_t*>(RQ.qdesc + (size_t)qf * 32);
            uint32_t w[8];
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = qd[i];
            const int beg = F.cellx_start[nMinCellX], end = F.cellx_start[nMaxCellX + 1];
            for (int p = beg + lane; p < end; p += 64) {
              const float4 kp = F.skp[p];
              const int idx = F.sidx[p];
              const uint4* td = reinterpret_cast<const uint4*>(F.sdesc + (size_t)p * 32);
              const uint4 a = td[0], b = td[1];
              const int oct = __float_as_int(kp.z);
              const int iy = __float_as_int(kp.w) % kGridRows;
              if (iy < nMinCellY || iy > nMaxCellY) continue;
              if (checkLevels) {
                if (oct < minLevel) continue;
                if (maxLevel >= 0 && oct > maxLevel) continue;
              }
              const float dx = kp.x - x, dy = kp.y - y;
              if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
              if (idx >= N || s_claimed[idx]) continue;
              const int d = __popc(a.x ^ w[0]) + __popc(a.y ^ w[1]) + __popc(a.z ^ w[2]) + __popc(a.w ^ w[3]) +
                            __popc(b.x ^ w[4]) + __popc(b.y ^ w[5]) + __popc(b.z ^ w[6]) + __popc(b.w ^ w[7]);
              best = min(best, ((uint32_t)d << 16) | ((uint32_t)p & 0xFFFFu));
            }
          }
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o));
          const int bd = (int)(best >> 16);
          const int bidx = bd < 256 ? F.sidx[best & 0xFFFFu] : -1;
          if (lane == first_und) { exhausted = false; pdist = bd; prop = bidx; n_requeried++; }
        }
        const bool matched = prop >= 0 && pdist <= th_high;
        const bool takes = matched && claims;
        if (!decided && takes) atomicMin(&s_owner[prop], (uint32_t)lane);
        const bool blocked = !decided && ((prop >= 0 && s_owner[prop] < (uint32_t)lane) || (exhausted && RQ.F.skp != nullptr));
        const unsigned long long bm = __ballot(blocked);
        const int first_blocked = bm ? (int)__builtin_ctzll(bm) : 64;
        if (!decided && takes) s_owner[prop] = 0xFFFFFFFFu;
        if (!decided && lane < first_blocked) {
          decided = true;
          if (exhausted) exhausted_any = 1;
          if (matched) {
            atomicMax(&s_assign[prop], q);              // the last writer in query order stays (:1651: CurrentFrame.mvpMapPoints[bestIdx2] = pMP)
            if (takes) s_claimed[prop] = 1;
            s_qres[q] = (uint32_t)prop;
          }
        }
      }
    }
    const int any_exhausted = __ballot(exhausted_any != 0) != 0ull;
    int nrq = n_requeried;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) nrq += __shfl_xor(nrq, o);
    if (lane == 0) { s_cnt[0] = any_exhausted; s_cnt[1] = nrq; s_cnt[2] = n_rounds; }
  }
  __syncthreads();
  // ---- parallel epilogue.  Rotation histogram of the matches (:1652-1663): counts only, so the order of the additions is free
  if (check_ori) {
    for (int q = tid; q < nq; q += 256) {
      const uint32_t r = s_qres[q];
      if (r == 0xFFFFFFFFu) continue;
      float rot = q_angle[q] - kps[r].angle;
      if (rot < 0.0f) rot += 360.0f;
      int bin = (int)roundf(rot * (1.0f / (float)kHisto));
      if (bin == kHisto) bin = 0;
      atomicAdd(&s_rot[bin], 1);
      s_qres[q] = r | ((uint32_t)bin << 16);
    }
  }
  __syncthreads();
  // ComputeThreeMaxima (ORBmatcher.cc:1750-1802) on one lane, then the matches of the other bins are taken back (:1730-1745)
  if (tid == 0) {
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < kHisto; i++) {
      const int sv = s_rot[i];
      if (sv > max1) { max3 = max2; max2 = max1; max1 = sv; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (sv > max2) { max3 = max2; max2 = sv; ind3 = ind2; ind2 = i; }
      else if (sv > max3) { max3 = sv; ind3 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) ind3 = -1;
    s_ind[0] = ind1; s_ind[1] = ind2; s_ind[2] = ind3;
  }
  __syncthreads();
  int nmatched = 0, ndropped = 0;
  for (int q = tid; q < nq; q += 256) {
    const uint32_t r = s_qres[q];
    if (r == 0xFFFFFFFFu) continue;
    nmatched++;
    if (check_ori) {
      const int bin = (int)(r >> 16);
      if (bin != s_ind[0] && bin != s_ind[1] && bin != s_ind[2]) { ndropped++; s_assign[r & 0xFFFFu] = -1; }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { nmatched += __shfl_xor(nmatched, o); ndropped += __shfl_xor(ndropped, o); }
  __shared__ int s_nm[4], s_nd[4];
  if (lane == 0) { s_nm[wave] = nmatched; s_nd[wave] = ndropped; }
  __syncthreads();
  // (both copies: the device one feeds k_track_gather, the mapped one is what the host reads after the chain's one synchronisation)
  for (int j = tid; j < kp_cap; j += 256) {
    const int a = j < N ? s_assign[j] : -1;
    assign[j] = a;
    if (j < N) assign_host[j] = a;
  }
  if (tid == 0) {
    const int nm = s_nm[0] + s_nm[1] + s_nm[2] + s_nm[3], nd = s_nd[0] + s_nd[1] + s_nd[2] + s_nd[3];
    res[0] = nm - nd; res[1] = s_cnt[0]; res[2] = nm; res[3] = s_cnt[1];
    res_host[0] = nm - nd; res_host[1] = s_cnt[0]; res_host[2] = nm; res_host[3] = s_cnt[1]; res_host[4] = s_cnt[2];
  }
}

// PoseOptimization's edges (Optimizer.cc:768-838): keypoints with a map point, in keypoint order.  One workgroup, block scan.
//   Xw[e] = (double)pos of the query's map point, obs[e] = (double)mvKeysUn[i].pt, info[e] = (double)mvInvLevelSigma2[octave]
// edge_kp[e] = i.  n_edges[0] = count (0 if the frame has fewer than min_matches matches: the caller's retry / lost path).
__global__ void __launch_bounds__(256) k_track_gather(const int32_t* __restrict__ assign, const dvm_keypoint_pod* __restrict__ kps_un,
                                                      const int32_t* __restrict__ d_n, int kp_cap, const float* __restrict__ q_pos,
                                                      const float* __restrict__ inv_sigma2, int nlevels, double* __restrict__ Xw,
                                                      double* __restrict__ obs, double* __restrict__ info, int32_t* __restrict__ edge_kp,
                                                      int32_t* __restrict__ n_edges, const int32_t* __restrict__ res, int min_matches,
                                                      int32_t* __restrict__ n_edges_host, TrackBatch B) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  {
    const int b = blockIdx.x;
    assign += (size_t)b * kp_cap; kps_un += (size_t)b * B.kps_stride; d_n += b; q_pos += (size_t)b * B.qstride * 3;
    Xw += (size_t)b * kp_cap * 3; obs += (size_t)b * kp_cap * 2; info += (size_t)b * kp_cap; edge_kp += (size_t)b * kp_cap;
    n_edges += b; n_edges_host += b; res += 8 * b;
  }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int N = min(*d_n, kp_cap);
  if (tid == 0) s_base = 0;
  __syncthreads();
  const bool go = res[0] >= min_matches && res[1] == 0;
  for (int i0 = 0; i0 < N && go; i0 += 256) {
    const int i = i0 + tid;
    const int q = i < N ? assign[i] : -1;
    const unsigned long long m = __ballot(q >= 0);
    if (lane == 0) s_wave[wv] = __popcll(m);
    __syncthreads();
    int pos = s_base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    for (int w = 0; w < wv; w++) pos += s_wave[w];
    if (q >= 0) {
      const dvm_keypoint_pod kp = kps_un[i];
      Xw[3 * pos] = (double)q_pos[3 * q]; Xw[3 * pos + 1] = (double)q_pos[3 * q + 1]; Xw[3 * pos + 2] = (double)q_pos[3 * q + 2];
      obs[2 * pos] = (double)kp.x; obs[2 * pos + 1] = (double)kp.y;
      info[pos] = (double)inv_sigma2[min(max(kp.octave, 0), nlevels - 1)];
      edge_kp[pos] = i;
    }
    __syncthreads();
    if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  if (tid == 0) { n_edges[0] = s_base; n_edges_host[0] = s_base; }
}

// mvbOutlier in keypoint order; Tracking.cc:2636-2660: a match PoseOptimization marked an outlier loses its map point, the others
// with Observations() > 0 count into nmatchesMap.  out[0] = nmatchesMap, out[1] = nmatches (res[0]) - outliers.
__global__ void __launch_bounds__(256) k_track_finish(int32_t* __restrict__ assign, const int32_t* __restrict__ d_n, int kp_cap,
                                                      const int32_t* __restrict__ edge_kp, const int32_t* __restrict__ n_edges,
                                                      const uint8_t* __restrict__ edge_outlier, const uint8_t* __restrict__ q_claims,
                                                      uint8_t* __restrict__ outlier, int32_t* __restrict__ out, const int32_t* __restrict__ res, TrackBatch B) {
  __shared__ int s_cnt[2];
  {
    const int b = blockIdx.x;
    assign += (size_t)b * kp_cap; d_n += b; edge_kp += (size_t)b * kp_cap; n_edges += b; edge_outlier += (size_t)b * kp_cap;
    q_claims += (size_t)b * B.qstride; outlier += (size_t)b * kp_cap; out += 4 * b; res += 8 * b;
  }
  const int tid = threadIdx.x;
  const int N = min(*d_n, kp_cap), E = n_edges[0];
  if (tid < 2) s_cnt[tid] = 0;
  for (int j = tid; j < kp_cap; j += 256) outlier[j] = 0;
  __syncthreads();
  int map = 0, left = 0;
  for (int e = tid; e < E; e += 256) {
    const int i = edge_kp[e];
    if (i < 0 || i >= N) continue;
    if (edge_outlier[e]) { outlier[i] = 1; left++; }
    else if (q_claims[assign[i]]) map++;
  }
  atomicAdd(&s_cnt[0], map);
  atomicAdd(&s_cnt[1], left);     // (outlier edges)
  __syncthreads();
  // nmatches is SearchByProjection's count (two queries without observations may have matched the same keypoint: both counted,
  // ORBmatcher.cc:1651-1653) minus one per keypoint whose match PoseOptimization rejected (Tracking.cc:2645-2653)
  if (tid == 0) { out[0] = s_cnt[0]; out[1] = res[0] - s_cnt[1]; }
}

size_t track_claims_lds(int kp_cap, int nq) { const size_t qp = ((size_t)nq + 63) & ~(size_t)63; return qp * 16 + (size_t)kp_cap * 9 + qp * 5 + 16; }

void launch_track_claims(hipStream_t s, const uint32_t* ranked, const uint8_t* q_claims, const float* q_angle, int nq, const TrackRequery& rq,
                         const dvm_keypoint_pod* kps, const int32_t* d_n, int kp_cap, int th_high, int check_ori, int32_t* assign, int32_t* res,
                         int32_t* assign_host, int32_t* res_host, const TrackBatch& B) {
  const size_t lds = track_claims_lds(kp_cap, B.count > 1 || B.nq_arr ? B.qstride : nq);
  if (lds > 48 * 1024) raise_dynamic_lds(reinterpret_cast<const void*>(k_track_claims), (int)lds);
  hipLaunchKernelGGL(k_track_claims, dim3(B.count), dim3(256), lds, s, ranked, q_claims, q_angle, nq, rq, kps, d_n, kp_cap, th_high,
                     check_ori, assign, res, assign_host, res_host, B);
}
void launch_track_gather(hipStream_t s, const int32_t* assign, const dvm_keypoint_pod* kps_un, const int32_t* d_n, int kp_cap, const float* q_pos,
                         const float* inv_sigma2, int nlevels, double* Xw, double* obs, double* info, int32_t* edge_kp, int32_t* n_edges,
                         const int32_t* res, int min_matches, int32_t* n_edges_host, const TrackBatch& B) {
  hipLaunchKernelGGL(k_track_gather, dim3(B.count), dim3(256), 0, s, assign, kps_un, d_n, kp_cap, q_pos, inv_sigma2, nlevels, Xw, obs, info, edge_kp, n_edges,
                     res, min_matches, n_edges_host, B);
}
void launch_track_finish(hipStream_t s, int32_t* assign, const int32_t* d_n, int kp_cap, const int32_t* edge_kp, const int32_t* n_edges,
                         const uint8_t* edge_outlier, const uint8_t* q_claims, uint8_t* outlier, int32_t* out, const int32_t* res, const TrackBatch& B) {
  hipLaunchKernelGGL(k_track_finish, dim3(B.count), dim3(256), 0, s, assign, d_n, kp_cap, edge_kp, n_edges, edge_outlier, q_claims, outlier, out, res, B);
}

}  // namespace dvm
