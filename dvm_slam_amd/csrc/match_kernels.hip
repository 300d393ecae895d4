// dvm_slam_amd/csrc/match_kernels.hip -- gfx950 kernels of the matching path.
//
//   ORBmatcher::DescriptorDistance (reference src/ORBmatcher.cc:1900-1914) -> popcount over 8 dwords
//   Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:481-506,773-782)  -> k_frame_build
//   Frame::GetFeaturesInArea (src/Frame.cc:712-770) + the best / second-best loops of
//   ORBmatcher::SearchByProjection (src/ORBmatcher.cc:70-115,1604-1639)      -> k_match_window
//
// Layout: a frame's keypoints are stored SORTED by (grid column ix, grid row iy, keypoint index) --
// exactly the order in which GetFeaturesInArea enumerates candidates -- so "first candidate wins a
// tie" becomes "smallest sorted position wins", an associative rule a wavefront can reduce.
#include <hip/hip_runtime.h>

#include "match_kernels.h"
#include "pose_f32.h"
#include "undistort_f64.h"
#include "jacobi4.h"

namespace dvm {

constexpr uint32_t kInvalidKey = 0xFFFFFFFFu;

// One workgroup (1024 threads) per frame slot.  Bitonic sort of (cell << 13 | idx) keys in LDS, then
// gather.  Slot s = first_slot + blockIdx.x reads keypoints kps + blockIdx.x*kps_stride.
__global__ void __launch_bounds__(1024) k_frame_build(const dvm_keypoint_pod* __restrict__ kps_base, int64_t kps_stride,
                                                      const uint8_t* __restrict__ desc_base, int64_t desc_stride,
                                                      int n_host, const int32_t* __restrict__ d_n, FrameView FB,
                                                      int first_slot) {
  __shared__ uint32_t keys[kFrameCap];
  const int tid = threadIdx.x;
  const FrameView F = FB.slot(first_slot + blockIdx.x);
  const dvm_keypoint_pod* kps = kps_base + (int64_t)blockIdx.x * kps_stride;
  const uint8_t* desc = desc_base + (int64_t)blockIdx.x * desc_stride;
  int n = d_n ? d_n[blockIdx.x] : n_host;
  if (tid == 0 && n > F.cap) atomicAdd(F.n_overflow, 1);   // a truncated frame is reported, not hidden (dvm_frame_overflows)
  n = min(max(n, 0), F.cap);
  int P = 64;
  while (P < n) P <<= 1;
  for (int i = tid; i < P; i += 1024) {
    uint32_t key = kInvalidKey;
    if (i < n) {
      const dvm_keypoint_pod kp = kps[i];
      // PosInGrid: round() = half away from zero; keypoints outside the grid are not indexed
      int px = (int)roundf((kp.x - F.minX) * F.wInv);
      int py = (int)roundf((kp.y - F.minY) * F.hInv);
      if (px >= 0 && px < kGridCols && py >= 0 && py < kGridRows) key = ((uint32_t)(px * kGridRows + py) << 13) | (uint32_t)i;
    }
    keys[i] = key;
  }
  __syncthreads();
  // A wave owns the 64-aligned blocks of its lanes (i = tid + 1024 t), so a stage with partner distance j < 64 only touches keys
  // the same wave wrote: between two such stages the wave's own LDS order is enough.  The workgroup barrier (16 waves, ~0.25 us,
  // and the sort is 55 stages for 1 024 keys) stays where a stage reads or has written across waves: 14 of the 55.
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += 1024) {
        int ixj = i ^ j;
        if (ixj > i) {
          uint32_t a = keys[i], b = keys[ixj];
          bool up = ((i & k) == 0);
          if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
        }
      }
      const int jn = j > 1 ? (j >> 1) : k;   // partner distance of the next stage (stage k << 1 opens with j = k)
      if (j >= 64 || jn >= 64) {
        __syncthreads();
      } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  // number of indexed keypoints = first invalid key (binary search by thread 0 is fine: log2(8192))
  __shared__ int s_m;
  if (tid == 0) {
    int lo = 0, hi = P;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (keys[mid] == kInvalidKey) hi = mid; else lo = mid + 1;
    }
    s_m = lo;
    *F.n_sorted = lo;
    *F.n_total = n;
  }
  __syncthreads();
  const int m = s_m;
  if (tid <= kGridCols) {  // cellx_start[c] = first sorted position whose column >= c
    uint32_t want = (uint32_t)(tid * kGridRows) << 13;
    int lo = 0, hi = m;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    F.cellx_start[tid] = lo;
  }
  for (int p = tid; p < m; p += 1024) {
    const uint32_t key = keys[p];
    const int i = (int)(key & 0x1FFFu);
    const dvm_keypoint_pod kp = kps[i];
    F.skp[p] = make_float4(kp.x, kp.y, __int_as_float(kp.octave), __int_as_float((int)(key >> 13)));
    F.sidx[p] = i;
    const uint4* s = reinterpret_cast<const uint4*>(desc + (size_t)i * 32);
    uint4* d = reinterpret_cast<uint4*>(F.sdesc + (size_t)p * 32);
    d[0] = s[0];
    d[1] = s[1];
  }
}

__device__ __forceinline__ void top2_insert(uint32_t& k1, uint32_t& k2, uint32_t k) {
  if (k < k1) { k2 = k1; k1 = k; }
  else if (k < k2) k2 = k;
}

// GetFeaturesInArea(x, y, r, minLevel, maxLevel) of frame F scanned by one DPP row (16 lanes): best / second best
// (dist << 16 | sorted position) after the row reduction, identical on all 16 lanes.  gate_inv_sigma2 != nullptr adds the
// per-candidate reprojection gate of ORBmatcher::Fuse (ORBmatcher.cc:1187-1196): skip if (ex^2 + ey^2) * invSigma2[octave]
// > gate (the comparison is made in double there: 5.99 is a double literal).
__device__ __forceinline__ void window_top2(const FrameView& F, float x, float y, float r, int minLevel, int maxLevel,
                                            const uint8_t* __restrict__ qdesc32, const uint8_t* __restrict__ skip,
                                            const float* __restrict__ gate_inv_sigma2, double gate, int lane, uint32_t& k1_out,
                                            uint32_t& k2_out) {
  uint32_t k1 = (256u << 16) | 0xFFFFu, k2 = k1;
  // GetFeaturesInArea cell rectangle (with the reference's early-outs)
  const int nMinCellX = max(0, (int)floorf((x - F.minX - r) * F.wInv));
  const int nMaxCellX = min(kGridCols - 1, (int)ceilf((x - F.minX + r) * F.wInv));
  const int nMinCellY = max(0, (int)floorf((y - F.minY - r) * F.hInv));
  const int nMaxCellY = min(kGridRows - 1, (int)ceilf((y - F.minY + r) * F.hInv));
  const bool empty = nMinCellX >= kGridCols || nMaxCellX < 0 || nMinCellY >= kGridRows || nMaxCellY < 0;
  if (!empty && nMinCellX <= nMaxCellX) {
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    const uint32_t* qd = reinterpret_cast<const uint32_t*>(qdesc32);
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = qd[i];
    const int beg = F.cellx_start[nMinCellX], end = F.cellx_start[nMaxCellX + 1];
    for (int p = beg + lane; p < end; p += 16) {
      const float4 kp = F.skp[p];
      const int oct = __float_as_int(kp.z);
      const int cell = __float_as_int(kp.w);
      const int iy = cell % kGridRows;
      if (iy < nMinCellY || iy > nMaxCellY) continue;
      if (checkLevels) {
        if (oct < minLevel) continue;
        if (maxLevel >= 0 && oct > maxLevel) continue;
      }
      const float dx = kp.x - x, dy = kp.y - y;
      if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
      if (skip && skip[F.sidx[p]]) continue;
      if (gate_inv_sigma2) {
        const float e2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
        if ((double)__fmul_rn(e2, gate_inv_sigma2[oct]) > gate) continue;
      }
      const uint4* td = reinterpret_cast<const uint4*>(F.sdesc + (size_t)p * 32);
      const uint4 a = td[0], b = td[1];
      int d = __popc(a.x ^ w[0]) + __popc(a.y ^ w[1]) + __popc(a.z ^ w[2]) + __popc(a.w ^ w[3]) +
              __popc(b.x ^ w[4]) + __popc(b.y ^ w[5]) + __popc(b.z ^ w[6]) + __popc(b.w ^ w[7]);
      top2_insert(k1, k2, ((uint32_t)d << 16) | (uint32_t)p);
    }
  }
  // top-2 of the row: xor-1, xor-2 inside quads, then half-row and row mirrors (the merged sets are disjoint)
#define DVM_TOP2_STEP(CTRL)                                                              \
  {                                                                                      \
    const uint32_t o1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k1, CTRL, 0xF, 0xF, false); \
    const uint32_t o2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k2, CTRL, 0xF, 0xF, false); \
    const uint32_t n1 = min(k1, o1), n2 = min(max(k1, o1), min(k2, o2));                 \
    k1 = n1; k2 = n2;                                                                    \
  }
  DVM_TOP2_STEP(0xB1)    // quad_perm [1,0,3,2]
  DVM_TOP2_STEP(0x4E)    // quad_perm [2,3,0,1]
  DVM_TOP2_STEP(0x141)   // row_half_mirror
  DVM_TOP2_STEP(0x140)   // row_mirror
#undef DVM_TOP2_STEP
  k1_out = k1; k2_out = k2;
}

// Sixteen lanes (one DPP row) per query, four queries per wave: a window holds ~10 candidates out of the ~70 of its
// grid columns, so a full wave per query left most lanes idle and paid a 6-step cross-lane reduction; a row reduces
// its top-2 with four DPP steps and no LDS traffic.  QUERIES_FROM_KPS: queries are keypoints of another frame (frame-to-frame
// search, window th*scale[octave], octaves [o-1,o+1]); otherwise explicit query arrays.
template <bool QUERIES_FROM_KPS>
__global__ void __launch_bounds__(256) k_match_window(FrameView FB, int first_slot, const uint8_t* __restrict__ skip,
                                                      const uint8_t* __restrict__ qdesc, const float* __restrict__ qx,
                                                      const float* __restrict__ qy, const float* __restrict__ qr,
                                                      const int32_t* __restrict__ qmin, const int32_t* __restrict__ qmax,
                                                      int nq_host, const int32_t* __restrict__ d_nq, PairQueries PQ,
                                                      float th, const float* __restrict__ scale_factors, int nlevels,
                                                      dvm_match_pod* __restrict__ out, int64_t out_stride,
                                                      int32_t* __restrict__ second_idx) {
  const int lane = threadIdx.x & 15;               // lane within the query's row
  const int q = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int pair = blockIdx.y;
  const FrameView F = FB.slot(first_slot + pair);
  const dvm_keypoint_pod* qkps = nullptr;
  int nq;
  if (QUERIES_FROM_KPS) {
    // pair 0 takes its queries from the carry frame (previous batch), pair i>0 from frame i-1
    if (pair == 0) {
      if (!PQ.carry_kps) { if (q == 0 && lane == 0 && PQ.n_out) PQ.n_out[0] = 0; return; }
      qkps = PQ.carry_kps; qdesc = PQ.carry_desc; nq = *PQ.carry_n;
    } else {
      qkps = PQ.kps + (int64_t)(pair - 1) * PQ.kps_stride;
      qdesc = PQ.desc + (int64_t)(pair - 1) * PQ.desc_stride;
      nq = PQ.n[pair - 1];
    }
    if (q == 0 && lane == 0 && nq > PQ.cap) atomicAdd(F.n_overflow, 1);
    nq = min(nq, PQ.cap);
    if (q == 0 && lane == 0 && PQ.n_out) PQ.n_out[pair] = nq;
    out += (int64_t)pair * out_stride;
  } else {
    nq = d_nq ? *d_nq : nq_host;
  }
  if (q >= nq) return;
  float x, y, r;
  int minLevel, maxLevel;
  if (QUERIES_FROM_KPS) {
    const dvm_keypoint_pod kp = qkps[q];
    x = kp.x; y = kp.y;
    const int o = min(max(kp.octave, 0), nlevels - 1);
    r = th * scale_factors[o];
    minLevel = kp.octave - 1;
    maxLevel = kp.octave + 1;
  } else {
    x = qx[q]; y = qy[q]; r = qr[q];
    minLevel = qmin[q]; maxLevel = qmax[q];
  }
  uint32_t k1, k2;
  window_top2(F, x, y, r, minLevel, maxLevel, qdesc + (size_t)q * 32, skip, nullptr, 0.0, lane, k1, k2);
  if (lane == 0) {
    dvm_match_pod m;
    const int p1 = (int)(k1 & 0xFFFFu), p2 = (int)(k2 & 0xFFFFu);
    m.best_dist = (int)(k1 >> 16);
    m.second_dist = (int)(k2 >> 16);
    m.best_idx = (m.best_dist < 256) ? F.sidx[p1] : -1;
    m.best_level = (m.best_dist < 256) ? (int16_t)__float_as_int(F.skp[p1].z) : (int16_t)-1;
    m.second_level = (m.second_dist < 256) ? (int16_t)__float_as_int(F.skp[p2].z) : (int16_t)-1;
    out[q] = m;
    // the runner-up's index: what the reference's scan finds when the best candidate is excluded (a caller replaying claims)
    if (!QUERIES_FROM_KPS && second_idx) second_idx[q] = (m.second_dist < 256) ? F.sidx[p2] : -1;
  }
}

// The same scan returning the FOUR best candidates in the reference's order of preference (distance, then scan position: strict '<',
// first wins).  A caller that replays ORBmatcher.cc:1613-1664's claims in query order walks the list to the first keypoint no earlier
// query of the call has taken -- with the best two only (k_match_window), 15 % of a frame's queries had to be searched again on the host.
// ranked[q][c] = dist << 16 | train index, dist = 256 from the end of the list on (fewer than four candidates: the list is complete).
__device__ __forceinline__ void sort4(uint32_t (&k)[4]) {
  uint32_t t;
#define DVM_CE(a, b) t = min(k[a], k[b]); k[b] = max(k[a], k[b]); k[a] = t;
  DVM_CE(0, 1) DVM_CE(2, 3) DVM_CE(0, 2) DVM_CE(1, 3) DVM_CE(1, 2)
#undef DVM_CE
}
// blockIdx.y = frame of a batch (the shared search service, dvm_match_pool_*): frame b searches grid slot `slot + b` with its own query
// arrays at b * qstride elements, nq_arr[b] queries and -- where skip_on[b] is set -- the skip flags at b * skip_stride bytes.
// A single search is the batch of one (qstride = 0, nq_arr = skip_on = NULL).
__global__ void __launch_bounds__(256) k_match_window_ranked(FrameView FB, int slot, const uint8_t* __restrict__ skip,
                                                             const uint8_t* __restrict__ qdesc, const float* __restrict__ qx,
                                                             const float* __restrict__ qy, const float* __restrict__ qr,
                                                             const int32_t* __restrict__ qmin, const int32_t* __restrict__ qmax, int nq,
                                                             uint32_t* __restrict__ ranked, int qstride, const int32_t* __restrict__ nq_arr,
                                                             const int32_t* __restrict__ skip_on, int skip_stride) {
  const int lane = threadIdx.x & 15;               // lane within the query's DPP row
  const int q = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int b = blockIdx.y;
  if (nq_arr) nq = nq_arr[b];
  if (q >= nq) return;
  const FrameView F = FB.slot(slot + b);
  {
    const size_t o = (size_t)b * qstride;
    qdesc += o * 32; qx += o; qy += o; qr += o; qmin += o; qmax += o; ranked += o * 4;
    if (skip_on) skip = skip_on[b] ? skip + (size_t)b * skip_stride : nullptr;
  }
  const float x = qx[q], y = qy[q], r = qr[q];
  const int minLevel = qmin[q], maxLevel = qmax[q];
  const uint32_t none = (256u << 16) | 0xFFFFu;
  uint32_t k[4] = {none, none, none, none};
  const int nMinCellX = max(0, (int)floorf((x - F.minX - r) * F.wInv));
  const int nMaxCellX = min(kGridCols - 1, (int)ceilf((x - F.minX + r) * F.wInv));
  const int nMinCellY = max(0, (int)floorf((y - F.minY - r) * F.hInv));
  const int nMaxCellY = min(kGridRows - 1, (int)ceilf((y - F.minY + r) * F.hInv));
  const bool empty = nMinCellX >= kGridCols || nMaxCellX < 0 || nMinCellY >= kGridRows || nMaxCellY < 0;
  if (!empty && nMinCellX <= nMaxCellX) {
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    const uint32_t* qd = reinterpret_cast<const uint32_t*>(qdesc + (size_t)q * 32);
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = qd[i];
    const int beg = F.cellx_start[nMinCellX], end = F.cellx_start[nMaxCellX + 1];
    for (int p = beg + lane; p < end; p += 16) {
      const float4 kp = F.skp[p];
      const int oct = __float_as_int(kp.z);
      const int iy = __float_as_int(kp.w) % kGridRows;
      if (iy < nMinCellY || iy > nMaxCellY) continue;
      if (checkLevels) {
        if (oct < minLevel) continue;
        if (maxLevel >= 0 && oct > maxLevel) continue;
      }
      const float dx = kp.x - x, dy = kp.y - y;
      if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
      if (skip && skip[F.sidx[p]]) continue;
      const uint4* td = reinterpret_cast<const uint4*>(F.sdesc + (size_t)p * 32);
      const uint4 a = td[0], b = td[1];
      const int d = __popc(a.x ^ w[0]) + __popc(a.y ^ w[1]) + __popc(a.z ^ w[2]) + __popc(a.w ^ w[3]) +
                    __popc(b.x ^ w[4]) + __popc(b.y ^ w[5]) + __popc(b.z ^ w[6]) + __popc(b.w ^ w[7]);
      const uint32_t key = ((uint32_t)d << 16) | (uint32_t)p;
      if (key < k[3]) { k[3] = key; sort4(k); }
    }
  }
  // the four smallest of two ascending 4-lists: min(a[i], b[3 - i]) (a bitonic half-cleaner), sorted again; the sets are disjoint
#define DVM_TOP4_STEP(CTRL)                                                                                     \
  {                                                                                                             \
    uint32_t o[4];                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; i++) o[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k[i], CTRL, 0xF, 0xF, false); \
    _Pragma("unroll") for (int i = 0; i < 4; i++) k[i] = min(k[i], o[3 - i]);                                   \
    sort4(k);                                                                                                   \
  }
  DVM_TOP4_STEP(0xB1)    // quad_perm [1,0,3,2]
  DVM_TOP4_STEP(0x4E)    // quad_perm [2,3,0,1]
  DVM_TOP4_STEP(0x141)   // row_half_mirror
  DVM_TOP4_STEP(0x140)   // row_mirror
#undef DVM_TOP4_STEP
  if (lane < 4) {
    uint32_t mine = k[0];
#pragma unroll
    for (int i = 1; i < 4; i++) mine = lane == i ? k[i] : mine;
    const uint32_t d = mine >> 16;
    ranked[(size_t)q * 4 + lane] = d < 256 ? ((d << 16) | (uint32_t)F.sidx[mine & 0xFFFFu]) : (256u << 16);
  }
}

// Best / second-best over an EXPLICIT candidate list per query (CSR: cand[off[q] .. off[q+1]) in the
// reference's scan order).  This is the inner loop of the vocabulary-node restricted searches --
// ORBmatcher::SearchByBoW (reference src/ORBmatcher.cc:262-300,:760-800), SearchForTriangulation
// (:905-960), SearchBySim3, Fuse -- whose candidate enumeration (DBoW2 FeatureVector walk, epipolar /
// chi2 gates) stays with the caller.  Strict '<' first-wins ties == min over (dist << 20 | position).
__global__ void __launch_bounds__(256) k_match_lists(const uint8_t* __restrict__ tdesc, const uint8_t* __restrict__ qdesc,
                                                     const int32_t* __restrict__ off, const int32_t* __restrict__ cand,
                                                     int nq, dvm_match_pod* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const uint32_t* qd = reinterpret_cast<const uint32_t*>(qdesc + (size_t)q * 32);
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = qd[i];
  const int beg = off[q], end = off[q + 1];
  unsigned long long k1 = (256ull << 32) | 0xFFFFFFFFull, k2 = k1;
  for (int p = beg + lane; p < end; p += 64) {
    const int idx = cand[p];
    if (idx < 0) continue;  // caller-masked candidate
    const uint4* td = reinterpret_cast<const uint4*>(tdesc + (size_t)idx * 32);
    const uint4 a = td[0], b = td[1];
    const int d = __popc(a.x ^ w[0]) + __popc(a.y ^ w[1]) + __popc(a.z ^ w[2]) + __popc(a.w ^ w[3]) +
                  __popc(b.x ^ w[4]) + __popc(b.y ^ w[5]) + __popc(b.z ^ w[6]) + __popc(b.w ^ w[7]);
    const unsigned long long k = ((unsigned long long)d << 32) | (unsigned)(p - beg);
    if (k < k1) { k2 = k1; k1 = k; } else if (k < k2) k2 = k;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long o1 = __shfl_xor(k1, o), o2 = __shfl_xor(k2, o);
    const unsigned long long n1 = min(k1, o1), n2 = min(max(k1, o1), min(k2, o2));
    k1 = n1; k2 = n2;
  }
  if (lane == 0) {
    dvm_match_pod m;
    m.best_dist = (int)(k1 >> 32);
    m.second_dist = (int)(k2 >> 32);
    m.best_idx = m.best_dist < 256 ? cand[beg + (int)(k1 & 0xFFFFFFFFu)] : -1;
    m.best_level = -1;
    m.second_level = -1;
    out[q] = m;
  }
}

// Projection of map points into a keyframe + windowed best-descriptor search: the common body of ORBmatcher::Fuse(KF, MPs)
// (reference src/ORBmatcher.cc:1060-1234), Fuse(KF, Scw, ...) (:1236-1345), SearchByProjection(KF, Scw, ...) x2 (:395-603)
// -- depth > 0, KeyFrame::IsInImage, distance inside the scale-invariance range, viewing angle < 60 deg
// (PO.Pn >= 0.5 dist), MapPoint::PredictScale, radius = th * scaleFactor[level], candidates of octave [level-1, level],
// optional per-candidate chi2 gate (Fuse).  One DPP row (16 lanes) per map point; every lane repeats the projection.
__global__ void __launch_bounds__(256) k_project_search(FrameView FB, int slot, const uint8_t* __restrict__ skip, ProjectCam C,
                                                        const float* __restrict__ P, const float* __restrict__ normal,
                                                        const float* __restrict__ min_dist, const float* __restrict__ max_dist,
                                                        const uint8_t* __restrict__ desc, const uint8_t* __restrict__ valid, int n,
                                                        const float* __restrict__ scale_factors,
                                                        const float* __restrict__ gate_inv_sigma2, double gate,
                                                        dvm_match_pod* __restrict__ out, Projection* __restrict__ proj) {
  const int lane = threadIdx.x & 15;
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (i >= n) return;
  const FrameView F = FB.slot(slot);
  float out_u = -1.f, out_v = -1.f, out_r = 0.f;
  int out_level = -1;
  bool ok = valid == nullptr || valid[i] != 0;
  const float p0 = P[3 * i], p1 = P[3 * i + 1], p2 = P[3 * i + 2];
  // p3Dc = Tcw * p3Dw: Sophus' quaternion form (so3.hpp:356-367), never a rotation matrix
  const float pw[3] = {p0, p1, p2};
  float pc[3];
  dvm_pose::se3_apply(C.q, C.t, pw, pc);
  float X = pc[0], Y = pc[1], Z = pc[2];
  float u, v;
  if (C.sim3_pair == 1) {   // SearchBySim3 (:1395-1411): p3Dc2 = S21 * (T1w * p3Dw); u = fx * (X * invz) + cx with invz = 1.0 / Z
    float p2c[3];
    dvm_pose::sim3_apply(C.q2, C.t2, pc, p2c);
    X = p2c[0]; Y = p2c[1]; Z = p2c[2];
    const float invz = (float)(1.0 / (double)Z);
    u = C.fx * (X * invz) + C.cx;
    v = C.fy * (Y * invz) + C.cy;
  } else {
    u = C.fx * X / Z + C.cx;
    v = C.fy * Y / Z + C.cy;
  }
  // sim3_pair == 2: SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (:1750-1860) -- no depth test, bounds
  // inclusive at both ends
  const bool reloc = C.sim3_pair == 2;
  const bool in_strict = u >= C.min_x && u < C.max_x && v >= C.min_y && v < C.max_y;
  const bool in_loose = !(u < C.min_x || u > C.max_x) && !(v < C.min_y || v > C.max_y);
  const bool front = !(Z < 0.0f);
  ok = ok && (reloc ? in_loose : (front && in_strict));
  if (ok) {
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    float q0 = p0 - C.Ow[0], q1 = p1 - C.Ow[1], q2 = p2 - C.Ow[2];
    if (C.sim3_pair == 1) { q0 = X; q1 = Y; q2 = Z; }
    const float dist = sqrtf(dvm_pose::sum3(q0 * q0, q1 * q1, q2 * q2));   // Vector3f::norm(): a0 + (a1 + a2)
    ok = !(dist < minDistance || dist > maxDistance);
    if (ok) {
      const float dot = dvm_pose::sum3(q0 * normal[3 * i], q1 * normal[3 * i + 1], q2 * normal[3 * i + 2]);
      ok = C.sim3_pair != 0 || !((double)dot < 0.5 * (double)dist);
      if (ok) {
        const int nScale = dvm_pose::predict_scale(max_dist[i], dist, C.log_scale_factor, C.n_levels);
        out_u = u; out_v = v; out_level = nScale; out_r = C.th * scale_factors[nScale];
      }
    }
  }
  uint32_t k1 = (256u << 16) | 0xFFFFu, k2 = k1;
  if (out_level >= 0)   // uniform inside the row: all 16 lanes computed the same projection
    window_top2(F, out_u, out_v, out_r, out_level - 1, reloc ? out_level + 1 : out_level, desc + (size_t)i * 32, skip, gate_inv_sigma2, gate, lane, k1, k2);
  if (lane == 0) {
    dvm_match_pod m;
    const int p1i = (int)(k1 & 0xFFFFu), p2i = (int)(k2 & 0xFFFFu);
    m.best_dist = (int)(k1 >> 16);
    m.second_dist = (int)(k2 >> 16);
    m.best_idx = (m.best_dist < 256) ? F.sidx[p1i] : -1;
    m.best_level = (m.best_dist < 256) ? (int16_t)__float_as_int(F.skp[p1i].z) : (int16_t)-1;
    m.second_level = (m.second_dist < 256) ? (int16_t)__float_as_int(F.skp[p2i].z) : (int16_t)-1;
    out[i] = m;
    if (proj) {
      Projection pr;
      pr.u = out_u; pr.v = out_v; pr.radius = out_r; pr.level = out_level;
      proj[i] = pr;
    }
  }
}

// ORBmatcher::SearchForTriangulation inner loop (reference src/ORBmatcher.cc:905-998, monocular): query q = keypoint
// qidx[q] of KF1 scans KF2 keypoints cand[off[q] .. off[q+1]) (its vocabulary node's features that are neither matched
// nor carry a map point); a candidate is taken if dist <= TH_LOW, dist <= the best so far, it is not within
// 100*scaleFactor[octave] (squared px) of the epipole, and (bCoarse or) Pinhole::epipolarConstrain holds
// (CameraModels/Pinhole.cpp:104-127).  The reference never sets vbMatched2, so queries are independent; within a query the
// sequential rule "<= replaces" is "minimum distance, LAST position wins a tie" = min over (dist << 16 | 0xFFFF - pos).
__global__ void __launch_bounds__(256) k_match_triangulation(const uint8_t* __restrict__ desc1, const dvm_keypoint_pod* __restrict__ kps1,
                                                             const int32_t* __restrict__ qidx, int nq,
                                                             const uint8_t* __restrict__ desc2, const dvm_keypoint_pod* __restrict__ kps2,
                                                             const int32_t* __restrict__ off, const int32_t* __restrict__ cand, TriGeom G,
                                                             const float* __restrict__ scale_factors2,
                                                             const float* __restrict__ level_sigma2_2, int32_t* __restrict__ best_idx,
                                                             int32_t* __restrict__ best_dist) {
  const int lane = threadIdx.x & 15;
  const int q = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (q >= nq) return;
  const int idx1 = qidx[q];
  const uint32_t* qd = reinterpret_cast<const uint32_t*>(desc1 + (size_t)idx1 * 32);
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = qd[i];
  const float x1 = kps1[idx1].x, y1 = kps1[idx1].y;
  // epipolar line in the second image l = x1' F12 = [a b c]
  const float a = __fadd_rn(__fadd_rn(__fmul_rn(x1, G.F12[0]), __fmul_rn(y1, G.F12[3])), G.F12[6]);
  const float b = __fadd_rn(__fadd_rn(__fmul_rn(x1, G.F12[1]), __fmul_rn(y1, G.F12[4])), G.F12[7]);
  const float c = __fadd_rn(__fadd_rn(__fmul_rn(x1, G.F12[2]), __fmul_rn(y1, G.F12[5])), G.F12[8]);
  const float den = __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
  const int beg = off[q], end = off[q + 1];
  uint32_t k = 0xFFFFFFFFu;
  for (int p = beg + lane; p < end; p += 16) {
    const int idx2 = cand[p];
    if (idx2 < 0) continue;
    const uint4* td = reinterpret_cast<const uint4*>(desc2 + (size_t)idx2 * 32);
    const uint4 A = td[0], B = td[1];
    const int d = __popc(A.x ^ w[0]) + __popc(A.y ^ w[1]) + __popc(A.z ^ w[2]) + __popc(A.w ^ w[3]) +
                  __popc(B.x ^ w[4]) + __popc(B.y ^ w[5]) + __popc(B.z ^ w[6]) + __popc(B.w ^ w[7]);
    if (d > G.th_low) continue;
    const dvm_keypoint_pod kp2 = kps2[idx2];
    const float distex = __fsub_rn(G.ep[0], kp2.x), distey = __fsub_rn(G.ep[1], kp2.y);
    if (__fadd_rn(__fmul_rn(distex, distex), __fmul_rn(distey, distey)) < __fmul_rn(100.f, scale_factors2[kp2.octave])) continue;
    if (!G.coarse) {
      const float num = __fadd_rn(__fadd_rn(__fmul_rn(a, kp2.x), __fmul_rn(b, kp2.y)), c);
      if (den == 0.f) continue;
      const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
      if (!((double)dsqr < 3.84 * (double)level_sigma2_2[kp2.octave])) continue;
    }
    k = min(k, ((uint32_t)d << 16) | (uint32_t)(0xFFFF - (p - beg)));
  }
  k = min(k, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k, 0xB1, 0xF, 0xF, false));
  k = min(k, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k, 0x4E, 0xF, 0xF, false));
  k = min(k, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k, 0x141, 0xF, 0xF, false));
  k = min(k, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k, 0x140, 0xF, 0xF, false));
  if (lane == 0) {
    const bool hit = k != 0xFFFFFFFFu;
    best_idx[q] = hit ? cand[beg + (0xFFFF - (int)(k & 0xFFFFu))] : -1;
    best_dist[q] = hit ? (int)(k >> 16) : 256;
  }
}

// KeyFrameDatabase place-recognition queries (reference src/KeyFrameDatabase.cc:224-808: DetectCandidates,
// DetectNBestCandidates, CalculateMergeScore) walk an inverted file word -> keyframes to count, per keyframe, the words it
// shares with the query BowVector, then score the survivors with L1Scoring::score (DBoW2/ScoringObject.cpp:23-63).
// A keyframe is in the inverted list of a word iff its BowVector holds the word, so the count is |ids(query) n ids(kf)| and
// the position of a keyframe in lKFsSharingWords is ordered by (first shared word, insertion order): one wave per stored
// keyframe intersects its sorted word list with the query's (binary search per word) and produces count, first shared
// word and the score for ALL keyframes at once -- no inverted file on the device.  The double sum of the score is taken
// in ascending word order (the reference's merge walk) by walking the hit mask of each 64-word chunk.
__global__ void __launch_bounds__(256) k_bowdb_query(const int64_t* __restrict__ kf_off, const int32_t* __restrict__ kf_len,
                                                     const int32_t* __restrict__ live, int n_live,
                                                     const int32_t* __restrict__ ids_base, const double* __restrict__ vals_base,
                                                     const int32_t* __restrict__ qids, const double* __restrict__ qvals, int nq,
                                                     int32_t* __restrict__ common, int32_t* __restrict__ first_word,
                                                     float* __restrict__ score) {
  const int lane = threadIdx.x & 63;
  const int li = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (li >= n_live) return;
  const int kf = live[li];                         // erased slots are not launched (their results are preset by the host)
  const int len = kf_len[kf];
  const int32_t* __restrict__ ids = ids_base + kf_off[kf];
  const double* __restrict__ vals = vals_base + kf_off[kf];
  constexpr int off = 0;
  int n_common = 0, first = -1;
  double acc = 0.0;
  for (int base = 0; base < len; base += 64) {
    const int p = base + lane;
    bool hit = false;
    double term = 0.0;
    if (p < len) {
      const int id = ids[off + p];
      int lo = 0, hi = nq;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (qids[mid] < id) lo = mid + 1; else hi = mid;
      }
      if (lo < nq && qids[lo] == id) {
        hit = true;
        const double vi = qvals[lo], wi = vals[off + p];   // score(v1 = query, v2 = keyframe)
        term = fabs(vi - wi) - fabs(vi) - fabs(wi);
      }
    }
    unsigned long long mask = __ballot(hit);
    if (mask != 0ull && first < 0) first = ids[off + base + (int)__builtin_ctzll(mask)];
    n_common += __builtin_popcountll(mask);
    while (mask) {
      const int l = (int)__builtin_ctzll(mask);
      const int lo32 = __builtin_amdgcn_readlane(__double2loint(term), l), hi32 = __builtin_amdgcn_readlane(__double2hiint(term), l);
      acc += __hiloint2double(hi32, lo32);
      mask &= mask - 1;
    }
  }
  if (lane == 0) {
    common[kf] = len < 0 ? -1 : n_common;
    first_word[kf] = first;
    score[kf] = (float)(-acc / 2.0);
  }
}

// Frame::isInFrustum, mono branch (reference src/Frame.cc:575-636) + MapPoint::PredictScale
// (src/MapPoint.cc:573-587): thread per map point, float arithmetic in the reference's order.
__global__ void __launch_bounds__(256) k_is_in_frustum(FrustumFrame F, const float* __restrict__ P, const float* __restrict__ normal,
                                                       const float* __restrict__ min_dist, const float* __restrict__ max_dist,
                                                       int n, float cos_limit, TrackPoint* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  TrackPoint o;
  o.in_view = 0; o.proj_x = -1; o.proj_y = -1; o.proj_xr = 0; o.depth = 0; o.level = -1; o.view_cos = 0;
  const float p0 = P[3 * i], p1 = P[3 * i + 1], p2 = P[3 * i + 2];
  // Pc = mRcw * P + mtcw (Frame.cc:585): Eigen's 3x3 * 3x1 coefficient is a0 + (a1 + a2)
  const float X = dvm_pose::sum3(F.Rcw[0] * p0, F.Rcw[1] * p1, F.Rcw[2] * p2) + F.tcw[0];
  const float Y = dvm_pose::sum3(F.Rcw[3] * p0, F.Rcw[4] * p1, F.Rcw[5] * p2) + F.tcw[1];
  const float Z = dvm_pose::sum3(F.Rcw[6] * p0, F.Rcw[7] * p1, F.Rcw[8] * p2) + F.tcw[2];
  const float Pc_dist = sqrtf(dvm_pose::sum3(X * X, Y * Y, Z * Z));
  const float invz = 1.0f / Z;
  bool ok = !(Z < 0.0f);
  const float u = F.fx * X / Z + F.cx, v = F.fy * Y / Z + F.cy;
  ok = ok && !(u < F.min_x || u > F.max_x) && !(v < F.min_y || v > F.max_y);
  if (ok) {
    o.proj_x = u; o.proj_y = v;
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    const float q0 = p0 - F.Ow[0], q1 = p1 - F.Ow[1], q2 = p2 - F.Ow[2];
    const float dist = sqrtf(dvm_pose::sum3(q0 * q0, q1 * q1, q2 * q2));
    if (!(dist < minDistance || dist > maxDistance)) {
      const float viewCos = dvm_pose::sum3(q0 * normal[3 * i], q1 * normal[3 * i + 1], q2 * normal[3 * i + 2]) / dist;
      if (!(viewCos < cos_limit)) {
        const int nScale = dvm_pose::predict_scale(max_dist[i], dist, F.log_scale_factor, F.n_levels);
        o.in_view = 1; o.proj_xr = u - F.bf * invz; o.depth = Pc_dist; o.level = nScale; o.view_cos = viewCos;
      }
    }
  }
  out[i] = o;
}

// LocalMapping::CreateNewMapPoints, the geometry of one neighbour keyframe's matches (reference src/LocalMapping.cc:598-741, monocular
// pinhole branch; GeometricTools::Triangulate src/GeometricTools.cc:48-67): thread per match.  Parallax of the two rays, the
// homogeneous point (null vector of the 4x4 system: eigenvector of the smallest eigenvalue of A^T A by cyclic Jacobi in double --
// the reference runs Eigen::JacobiSVD<Matrix4f>, tolerance parity), depth, reprojection error and scale-consistency tests in float
// in Eigen's evaluation order, comparisons with double literals in double.  status: include/dvmslam_hip.h.
__global__ void __launch_bounds__(128) k_triangulate_matches(TriPair P, const dvm_keypoint_pod* __restrict__ kps1, int n1,
                                                             const dvm_keypoint_pod* __restrict__ kps2, int n2,
                                                             const int32_t* __restrict__ pairs, int n, const float* __restrict__ sigma2_1,
                                                             const float* __restrict__ sigma2_2, const float* __restrict__ sf1,
                                                             const float* __restrict__ sf2, float* __restrict__ x3D_out,
                                                             int32_t* __restrict__ status) {
  const int m = blockIdx.x * 128 + threadIdx.x;
  if (m >= n) return;
  using dvm_pose::sum3;
  float* X = x3D_out + 3 * (int64_t)m;
  X[0] = X[1] = X[2] = 0.0f;
  const int i1 = pairs[2 * m], i2 = pairs[2 * m + 1];
  if (i1 < 0 || i1 >= n1 || i2 < 0 || i2 >= n2) { status[m] = -1; return; }
  const dvm_keypoint_pod kp1 = kps1[i1], kp2 = kps2[i2];
  if (kp1.octave < 0 || kp1.octave >= P.n_levels || kp2.octave < 0 || kp2.octave >= P.n_levels) { status[m] = -1; return; }
  const float* T1w = P.T1w;
  const float* T2w = P.T2w;
  const float xn1[3] = {(kp1.x - P.K1[2]) / P.K1[0], (kp1.y - P.K1[3]) / P.K1[1], 1.0f};
  const float xn2[3] = {(kp2.x - P.K2[2]) / P.K2[0], (kp2.y - P.K2[3]) / P.K2[1], 1.0f};
  float r1[3], r2[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    r1[i] = sum3(T1w[i] * xn1[0], T1w[4 + i] * xn1[1], T1w[8 + i] * xn1[2]);
    r2[i] = sum3(T2w[i] * xn2[0], T2w[4 + i] * xn2[1], T2w[8 + i] * xn2[2]);
  }
  const float nr1 = sqrtf(sum3(r1[0] * r1[0], r1[1] * r1[1], r1[2] * r1[2])), nr2 = sqrtf(sum3(r2[0] * r2[0], r2[1] * r2[1], r2[2] * r2[2]));
  const float cosParallaxRays = sum3(r1[0] * r2[0], r1[1] * r2[1], r1[2] * r2[2]) / (nr1 * nr2);
  const float cosParallaxStereo = cosParallaxRays + 1;
  if (!(cosParallaxRays < cosParallaxStereo && cosParallaxRays > 0 && (double)cosParallaxRays < P.cos_parallax_max)) { status[m] = 1; return; }
  float A[4][4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    A[0][k] = xn1[0] * T1w[8 + k] - T1w[k];
    A[1][k] = xn1[1] * T1w[8 + k] - T1w[4 + k];
    A[2][k] = xn2[0] * T2w[8 + k] - T2w[k];
    A[3][k] = xn2[1] * T2w[8 + k] - T2w[4 + k];
  }
  double B[4][4], V[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 4; k++) acc += (double)A[k][i] * (double)A[k][j];
      B[i][j] = acc;
    }
  jacobi4_dev(B, V);
  int mi = 0;
#pragma unroll
  for (int k = 1; k < 4; k++) if (B[k][k] < B[mi][mi]) mi = k;
  float vh[4];
#pragma unroll
  for (int k = 0; k < 4; k++) vh[k] = (float)(mi == 0 ? V[k][0] : mi == 1 ? V[k][1] : mi == 2 ? V[k][2] : V[k][3]);
  if (vh[3] == 0) { status[m] = 2; return; }
  const float x3D[3] = {vh[0] / vh[3], vh[1] / vh[3], vh[2] / vh[3]};
  X[0] = x3D[0]; X[1] = x3D[1]; X[2] = x3D[2];
  const float z1 = sum3(T1w[8] * x3D[0], T1w[9] * x3D[1], T1w[10] * x3D[2]) + T1w[11];
  if (z1 <= 0) { status[m] = 3; return; }
  const float z2 = sum3(T2w[8] * x3D[0], T2w[9] * x3D[1], T2w[10] * x3D[2]) + T2w[11];
  if (z2 <= 0) { status[m] = 4; return; }
  {
    const float x1 = sum3(T1w[0] * x3D[0], T1w[1] * x3D[1], T1w[2] * x3D[2]) + T1w[3];
    const float y1 = sum3(T1w[4] * x3D[0], T1w[5] * x3D[1], T1w[6] * x3D[2]) + T1w[7];
    const float u = P.K1[0] * x1 / z1 + P.K1[2], v = P.K1[1] * y1 / z1 + P.K1[3];
    const float ex = u - kp1.x, ey = v - kp1.y;
    if ((double)(ex * ex + ey * ey) > 5.991 * (double)sigma2_1[kp1.octave]) { status[m] = 5; return; }
  }
  {
    const float x2 = sum3(T2w[0] * x3D[0], T2w[1] * x3D[1], T2w[2] * x3D[2]) + T2w[3];
    const float y2 = sum3(T2w[4] * x3D[0], T2w[5] * x3D[1], T2w[6] * x3D[2]) + T2w[7];
    const float u = P.K2[0] * x2 / z2 + P.K2[2], v = P.K2[1] * y2 / z2 + P.K2[3];
    const float ex = u - kp2.x, ey = v - kp2.y;
    if ((double)(ex * ex + ey * ey) > 5.991 * (double)sigma2_2[kp2.octave]) { status[m] = 6; return; }
  }
  const float d1[3] = {x3D[0] - P.Ow1[0], x3D[1] - P.Ow1[1], x3D[2] - P.Ow1[2]}, d2[3] = {x3D[0] - P.Ow2[0], x3D[1] - P.Ow2[1], x3D[2] - P.Ow2[2]};
  const float dist1 = sqrtf(sum3(d1[0] * d1[0], d1[1] * d1[1], d1[2] * d1[2])), dist2 = sqrtf(sum3(d2[0] * d2[0], d2[1] * d2[1], d2[2] * d2[2]));
  if (dist1 == 0 || dist2 == 0) { status[m] = 7; return; }
  if (P.far_points && (dist1 >= P.th_far || dist2 >= P.th_far)) { status[m] = 8; return; }
  const float ratioDist = dist2 / dist1;
  const float ratioOctave = sf1[kp1.octave] / sf2[kp2.octave];
  if (ratioDist * P.ratio_factor < ratioOctave || ratioDist > ratioOctave * P.ratio_factor) { status[m] = 9; return; }
  status[m] = 0;
}
void launch_triangulate_matches(hipStream_t s, const TriPair& P, const dvm_keypoint_pod* kps1, int n1, const dvm_keypoint_pod* kps2, int n2,
                                const int32_t* pairs, int n, const float* sigma2_1, const float* sigma2_2, const float* sf1,
                                const float* sf2, float* x3D, int32_t* status) {
  hipLaunchKernelGGL(k_triangulate_matches, dim3((n + 127) / 128), dim3(128), 0, s, P, kps1, n1, kps2, n2, pairs, n, sigma2_1, sigma2_2, sf1, sf2,
                     x3D, status);
}

// D[i][j] = Hamming(A[i], B[j]); one thread per pair, 64 columns x 4 rows per workgroup.
__global__ void __launch_bounds__(256) k_hamming_matrix(const uint8_t* __restrict__ A, int nA,
                                                        const uint8_t* __restrict__ B, int nB, uint16_t* __restrict__ D) {
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= nA || j >= nB) return;
  const uint4* a = reinterpret_cast<const uint4*>(A + (size_t)i * 32);
  const uint4* b = reinterpret_cast<const uint4*>(B + (size_t)j * 32);
  const uint4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
  int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
          __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
  D[(size_t)i * nB + j] = (uint16_t)d;
}

void launch_frame_build(hipStream_t s, const dvm_keypoint_pod* kps, int64_t kps_stride, const uint8_t* desc,
                        int64_t desc_stride, int n, const int32_t* d_n, const FrameView& F, int first_slot, int count) {
  hipLaunchKernelGGL(k_frame_build, dim3(count), dim3(1024), 0, s, kps, kps_stride, desc, desc_stride, n, d_n, F, first_slot);
}
void launch_match_window(hipStream_t s, const FrameView& F, int slot, const uint8_t* skip, const uint8_t* qdesc,
                         const float* qx, const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax,
                         int nq, const int32_t* d_nq, int grid_q, dvm_match_pod* out, int32_t* second_idx) {
  PairQueries pq{};
  hipLaunchKernelGGL(k_match_window<false>, dim3((grid_q + 15) / 16, 1), dim3(256), 0, s, F, slot, skip, qdesc, qx, qy, qr,
                     qmin, qmax, nq, d_nq, pq, 0.f, nullptr, 0, out, 0, second_idx);
}
void launch_match_window_ranked(hipStream_t s, const FrameView& F, int slot, const uint8_t* skip, const uint8_t* qdesc, const float* qx,
                                const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax, int nq, uint32_t* ranked) {
  hipLaunchKernelGGL(k_match_window_ranked, dim3((nq + 15) / 16), dim3(256), 0, s, F, slot, skip, qdesc, qx, qy, qr, qmin, qmax, nq, ranked, 0,
                     nullptr, nullptr, 0);
}
void launch_match_window_ranked_batch(hipStream_t s, const FrameView& F, int first_slot, int count, const uint8_t* skip, const int32_t* skip_on,
                                      int skip_stride, const uint8_t* qdesc, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                                      const int32_t* qmax, const int32_t* nq_arr, int qstride, uint32_t* ranked) {
  hipLaunchKernelGGL(k_match_window_ranked, dim3((qstride + 15) / 16, count), dim3(256), 0, s, F, first_slot, skip, qdesc, qx, qy, qr, qmin, qmax, 0,
                     ranked, qstride, nq_arr, skip_on, skip_stride);
}
void launch_match_frames(hipStream_t s, const FrameView& F, int first_slot, int count, const PairQueries& pq, float th,
                         const float* scale_factors, int nlevels, dvm_match_pod* out, int64_t out_stride) {
  hipLaunchKernelGGL(k_match_window<true>, dim3((pq.cap + 15) / 16, count), dim3(256), 0, s, F, first_slot, nullptr, nullptr,
                     nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, pq, th, scale_factors, nlevels, out, out_stride, nullptr);
}
// Frame::UndistortKeyPoints (Frame.cc:791-818): thread per keypoint, cv::undistortPoints in double (undistort_f64.h)
__global__ void __launch_bounds__(256) k_undistort_keypoints(dvm_undistort::Camera cam, const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float kp[7];
#pragma unroll
  for (int k = 0; k < 7; k++) kp[k] = in[7 * (int64_t)i + k];
  if (cam.k1 != 0.0f) dvm_undistort::undistort_point(cam, kp[0], kp[1], &kp[0], &kp[1]);
#pragma unroll
  for (int k = 0; k < 7; k++) out[7 * (int64_t)i + k] = kp[k];
}
void launch_undistort_keypoints(hipStream_t s, const dvm_undistort::Camera& cam, const float* in, float* out, int n) {
  hipLaunchKernelGGL(k_undistort_keypoints, dim3((n + 255) / 256), dim3(256), 0, s, cam, in, out, n);
}

void launch_is_in_frustum(hipStream_t s, const FrustumFrame& F, const float* P, const float* normal, const float* min_dist,
                          const float* max_dist, int n, float cos_limit, TrackPoint* out) {
  hipLaunchKernelGGL(k_is_in_frustum, dim3((n + 255) / 256), dim3(256), 0, s, F, P, normal, min_dist, max_dist, n, cos_limit, out);
}
void launch_match_lists(hipStream_t s, const uint8_t* tdesc, const uint8_t* qdesc, const int32_t* off, const int32_t* cand,
                        int nq, dvm_match_pod* out) {
  hipLaunchKernelGGL(k_match_lists, dim3((nq + 3) / 4), dim3(256), 0, s, tdesc, qdesc, off, cand, nq, out);
}
void launch_project_search(hipStream_t s, const FrameView& F, int slot, const uint8_t* skip, const ProjectCam& C, const float* P,
                           const float* normal, const float* min_dist, const float* max_dist, const uint8_t* desc,
                           const uint8_t* valid, int n, const float* scale_factors, const float* gate_inv_sigma2, double gate,
                           dvm_match_pod* out, Projection* proj) {
  hipLaunchKernelGGL(k_project_search, dim3((n + 15) / 16), dim3(256), 0, s, F, slot, skip, C, P, normal, min_dist, max_dist, desc,
                     valid, n, scale_factors, gate_inv_sigma2, gate, out, proj);
}
void launch_match_triangulation(hipStream_t s, const uint8_t* desc1, const dvm_keypoint_pod* kps1, const int32_t* qidx, int nq,
                                const uint8_t* desc2, const dvm_keypoint_pod* kps2, const int32_t* off, const int32_t* cand,
                                const TriGeom& G, const float* scale_factors2, const float* level_sigma2_2, int32_t* best_idx,
                                int32_t* best_dist) {
  hipLaunchKernelGGL(k_match_triangulation, dim3((nq + 15) / 16), dim3(256), 0, s, desc1, kps1, qidx, nq, desc2, kps2, off, cand, G,
                     scale_factors2, level_sigma2_2, best_idx, best_dist);
}
void launch_bowdb_query(hipStream_t s, const int64_t* kf_off, const int32_t* kf_len, const int32_t* live, int n_live, const int32_t* ids,
                        const double* vals, const int32_t* qids, const double* qvals, int nq, int32_t* common, int32_t* first_word, float* score) {
  if (n_live > 0)
    hipLaunchKernelGGL(k_bowdb_query, dim3((n_live + 3) / 4), dim3(256), 0, s, kf_off, kf_len, live, n_live, ids, vals, qids, qvals, nq,
                       common, first_word, score);
}
void launch_hamming_matrix(hipStream_t s, const uint8_t* A, int nA, const uint8_t* B, int nB, uint16_t* D) {
  hipLaunchKernelGGL(k_hamming_matrix, dim3((nB + 63) / 64, (nA + 3) / 4), dim3(256), 0, s, A, nA, B, nB, D);
}

// MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:384-453), batched over map points: point p owns
// the descriptors desc[off[p] .. off[p+1]) (its observations, in the reference's iteration order).  For every row i of
// the N x N Hamming table the median is vDists[0.5 * (N - 1)] of the SORTED row (self distance 0 included); the
// winner is the first i with the strictly smallest median.  One wavefront per map point: descriptors staged in LDS,
// row i = one distance per lane (chunks of 64), the k-th order statistic by a 9-step binary search over the value
// range [0, 256] with ballot counts -- no sort.  Points with more than kDistinctMaxN observations report -2.
constexpr int kDistinctMaxN = 512;
__global__ void __launch_bounds__(64) k_distinctive(const uint8_t* __restrict__ desc, const int32_t* __restrict__ off, int npts,
                                                    int32_t* __restrict__ best_idx, int32_t* __restrict__ best_median) {
  __shared__ uint32_t s_desc[kDistinctMaxN * 8];
  __shared__ uint16_t s_row[kDistinctMaxN];
  const int p = blockIdx.x, lane = threadIdx.x;
  if (p >= npts) return;
  const int o0 = off[p], N = off[p + 1] - o0;
  if (N <= 0) { if (lane == 0) { best_idx[p] = -1; best_median[p] = -1; } return; }
  if (N > kDistinctMaxN) { if (lane == 0) { best_idx[p] = -2; best_median[p] = -2; } return; }
  const uint32_t* g = reinterpret_cast<const uint32_t*>(desc + (size_t)o0 * 32);
  for (int i = lane; i < N * 8; i += 64) s_desc[i] = g[i];
  __syncthreads();
  const int k = (N - 1) >> 1;                     // (size_t)(0.5 * (N - 1))
  uint32_t best = 0xFFFFFFFFu;                    // median << 16 | i
  for (int i = 0; i < N; i++) {
    uint32_t di[8];
#pragma unroll
    for (int w = 0; w < 8; w++) di[w] = s_desc[i * 8 + w];
    for (int j = lane; j < N; j += 64) {
      int d = 0;
#pragma unroll
      for (int w = 0; w < 8; w++) d += __popc(di[w] ^ s_desc[j * 8 + w]);
      s_row[j] = (uint16_t)d;
    }
    __syncthreads();
    // smallest v with #{j : d_j <= v} > k
    int lo = 0, hi = 256;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      int cnt = 0;
      for (int base = 0; base < N; base += 64) {
        const int j = base + lane;
        cnt += __popcll(__ballot(j < N && (int)s_row[j] <= mid));
      }
      if (cnt > k) hi = mid; else lo = mid + 1;
    }
    best = min(best, ((uint32_t)lo << 16) | (uint32_t)i);
    __syncthreads();
  }
  if (lane == 0) { best_idx[p] = (int)(best & 0xFFFFu); best_median[p] = (int)(best >> 16); }
}
void launch_distinctive(hipStream_t s, const uint8_t* desc, const int32_t* off, int npts, int32_t* best_idx, int32_t* best_median) {
  if (npts > 0) hipLaunchKernelGGL(k_distinctive, dim3(npts), dim3(64), 0, s, desc, off, npts, best_idx, best_median);
}

// DBoW2 TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (reference
// Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1098-1138): walk the vocabulary tree from the root, at every node take
// the child with the smallest FORB::distance (first child wins ties: strict '<' scan), remember the node reached at
// level L - levelsup.  One DPP row (16 lanes) per feature: lane c scores child c (k = 10 in ORBvoc), the argmin of
// (distance << 16 | child position) is four DPP steps.  Nodes: CSR children lists, 32-B descriptors, weight, word id.
__global__ void __launch_bounds__(256) k_vocab_transform(const int32_t* __restrict__ child_off, const int32_t* __restrict__ children,
                                                         const uint8_t* __restrict__ node_desc, const double* __restrict__ weight,
                                                         const int32_t* __restrict__ word_id, int L, const uint8_t* __restrict__ feat,
                                                         int n, int levelsup, int32_t* __restrict__ out_word,
                                                         int32_t* __restrict__ out_node, double* __restrict__ out_weight) {
  const int lane = threadIdx.x & 15;
  const int f = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (f >= n) return;
  uint32_t w[8];
  {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(feat + (size_t)f * 32);
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = q[i];
  }
  const int nid_level = L - levelsup;
  int nid = nid_level <= 0 ? 0 : -1;   // the reference leaves *nid unset if the walk ends above nid_level
  int cur = 0, level = 0;
  int c0 = child_off[0], c1 = child_off[1];
  while (c1 > c0) {
    ++level;
    uint32_t best = 0xFFFFFFFFu;
    for (int base = c0; base < c1; base += 16) {
      const int c = base + lane;
      uint32_t key = 0xFFFFFFFFu;
      if (c < c1) {
        const uint4* d = reinterpret_cast<const uint4*>(node_desc + (size_t)children[c] * 32);
        const uint4 a = d[0], b = d[1];
        const int dist = __popc(a.x ^ w[0]) + __popc(a.y ^ w[1]) + __popc(a.z ^ w[2]) + __popc(a.w ^ w[3]) +
                         __popc(b.x ^ w[4]) + __popc(b.y ^ w[5]) + __popc(b.z ^ w[6]) + __popc(b.w ^ w[7]);
        key = ((uint32_t)dist << 16) | (uint32_t)(c - c0);
      }
      best = min(best, key);
    }
    best = min(best, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)best, 0xB1, 0xF, 0xF, false));
    best = min(best, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)best, 0x4E, 0xF, 0xF, false));
    best = min(best, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)best, 0x141, 0xF, 0xF, false));
    best = min(best, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)best, 0x140, 0xF, 0xF, false));
    cur = children[c0 + (int)(best & 0xFFFFu)];
    if (level == nid_level) nid = cur;
    c0 = child_off[cur]; c1 = child_off[cur + 1];
  }
  if (lane == 0) { out_word[f] = word_id[cur]; out_node[f] = nid; out_weight[f] = weight[cur]; }
}
void launch_vocab_transform(hipStream_t s, const int32_t* child_off, const int32_t* children, const uint8_t* node_desc,
                            const double* weight, const int32_t* word_id, int L, const uint8_t* feat, int n, int levelsup,
                            int32_t* out_word, int32_t* out_node, double* out_weight) {
  if (n > 0)
    hipLaunchKernelGGL(k_vocab_transform, dim3((n + 15) / 16), dim3(256), 0, s, child_off, children, node_desc, weight, word_id, L,
                       feat, n, levelsup, out_word, out_node, out_weight);
}

}  // namespace dvm
