// dvm_slam_amd/csrc/track.cpp -- dvm_tracker: ONE enqueue per tracked frame (include/dvmslam_hip.h, "tracking step").
//
// The reference's per-frame hot loop is Frame::Frame -> ExtractORB (src/Frame.cc:371-411) followed by
// Tracking::TrackWithMotionModel (src/Tracking.cc:2584-2667): SearchByProjection(CurrentFrame, LastFrame) (src/ORBmatcher.cc:1553-1748)
// -> Optimizer::PoseOptimization (src/Optimizer.cc:744-1028) -> outlier matches dropped.  Through the three separate calls of this
// library that is three blocking host <-> device round trips with the claim replay, the rotation histogram and the edge gathering on
// the host in between.  Here the whole step is one chain on the extractor's stream:
//   dvm_track_begin    queues the extraction of the frame and returns (the host builds the projection queries meanwhile: they need
//                      LastFrame's map points and the predicted pose, nothing of the new frame)
//   dvm_track_finish   queues [undistortion] -> grid (k_frame_build) -> ranked window search -> k_track_claims -> k_track_gather ->
//                      k_pose_optimize -> k_track_finish behind it, synchronises ONCE and hands everything back
// Results are those of the separate calls bit for bit (tests/test_gpu_track_frame.py).  Two cases are handed back to the caller
// unfinished, flagged in dvm_track_result::status: fewer than min_matches matches (the reference searches again with a doubled window,
// Tracking.cc:2616-2624: call dvm_track_finish again with the wider queries -- no new extraction), and a query whose four ranked
// candidates were all taken by earlier queries (the list may go on: the caller replays the epilogue from the ranked lists on the host).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/dvmslam_hip.h"
#include "ba_kernels.h"
#include "match_kernels.h"
#include "orb_pipeline.h"
#include "host_stage.h"   // HostPool
#include "track_kernels.h"

using namespace dvm;

struct dvm_tracker {
  int device = 0, kp_cap = 0, q_cap = 0, max_frames = 1;
  dvm_frame* grid = nullptr;       // max_frames slots
  uint8_t* d_buf = nullptr;        // device working set (one allocation)
  uint8_t *hm = nullptr, *hm_dev = nullptr;   // mapped page-locked buffer: queries in (staging of the one copy), results out
  size_t hm_bytes = 0;
  // device, [frame][...]
  uint32_t* d_ranked; int32_t* d_assign; int32_t* d_res; double *d_Xw, *d_obs, *d_info, *d_chi; int32_t *d_edge_kp, *d_nedges;
  uint8_t* d_edge_out; dvm_keypoint_pod* d_kps_un;
  // mapped (host address; device address = hm_dev + (p - hm)).  The query block is carved per call (stride = the call's largest nq).
  struct Mapped {
    uint8_t* qdesc; float *qx, *qy, *qr; int32_t *qmin, *qmax; uint8_t* q_claims; float *q_angle, *q_pos; double* pose_in; int32_t* nq_arr;
    float* inv_sigma2;
    int32_t* assign; uint8_t* outlier; int32_t* res; int32_t* fin; int32_t* nedges; double* pose_out; int32_t* n_inl;
    dvm_keypoint_pod* kps_un;
  } m;
  template <class T> T* dev(T* host_ptr) const { return reinterpret_cast<T*>(hm_dev + (reinterpret_cast<uint8_t*>(host_ptr) - hm)); }
  // the query block: built in the mapped buffer (page-locked), copied to the device by ONE asynchronous copy on a side stream while the
  // extraction runs; the kernels read the device copy (the one-wave claim replay walks it serially: a PCIe read per step would be its chain)
  uint8_t* d_q = nullptr; size_t q_bytes = 0;
  hipStream_t cstream = nullptr; hipEvent_t cev = nullptr;
  template <class T> T* qdev(T* host_ptr) const { return reinterpret_cast<T*>(d_q + (reinterpret_cast<uint8_t*>(host_ptr) - hm)); }
  int begun = 0;                   // frames of the batch whose extraction is queued
  int rows = 0, cols = 0;
};

namespace {
size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }
template <class T> T* carve(uint8_t*& p, size_t count) { T* r = reinterpret_cast<T*>(p); p += pad256(count * sizeof(T)); return r; }
}  // namespace

extern "C" {

int dvm_tracker_create_batch(int device, int max_frames, int max_keypoints, int max_queries, dvm_tracker** out) {
  if (!out || max_keypoints < 1 || max_queries < 1 || max_frames < 1 || max_frames > 256) return DVM_ERR_INVALID;
  *out = nullptr;
  if (max_keypoints > kFrameCap || max_queries > kFrameCap || track_claims_lds(max_keypoints, max_queries) > 150 * 1024) {
    set_error("dvm_tracker_create: capacity beyond what the claim replay keeps in LDS (9 B per keypoint + 21 B per query <= 150 KB)");
    return DVM_ERR_CAPACITY;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  if (device < 0 || device >= ndev) { set_error("device index out of range"); return DVM_ERR_INVALID; }
  DVM_HIP(hipSetDevice(device));
  dvm_tracker* t = new (std::nothrow) dvm_tracker();
  if (!t) return DVM_ERR_INVALID;
  t->device = device; t->kp_cap = max_keypoints; t->q_cap = max_queries; t->max_frames = max_frames;
  int rc = dvm_frame_create(device, max_keypoints, max_frames, &t->grid);
  if (rc != DVM_OK) { delete t; return rc; }
  const size_t K = (size_t)max_keypoints, Q = (size_t)max_queries, B = (size_t)max_frames;
  // device working set
  size_t dbytes = pad256(B * Q * 16) + pad256(B * K * 4) + pad256(B * 32) + pad256(B * K * 24) + pad256(B * K * 16) + 2 * pad256(B * K * 8) + pad256(B * K * 4) +
                  pad256(B * 4) + pad256(B * K) + pad256(B * K * sizeof(dvm_keypoint_pod));
  if (hipMalloc(reinterpret_cast<void**>(&t->d_buf), dbytes) != hipSuccess) { dvm_tracker_destroy(t); set_error("dvm_tracker_create: hipMalloc"); return DVM_ERR_HIP; }
  uint8_t* p = t->d_buf;
  t->d_ranked = carve<uint32_t>(p, B * Q * 4); t->d_assign = carve<int32_t>(p, B * K); t->d_res = carve<int32_t>(p, B * 8);
  t->d_Xw = carve<double>(p, B * K * 3); t->d_obs = carve<double>(p, B * K * 2); t->d_info = carve<double>(p, B * K); t->d_chi = carve<double>(p, B * K);
  t->d_edge_kp = carve<int32_t>(p, B * K); t->d_nedges = carve<int32_t>(p, B); t->d_edge_out = carve<uint8_t>(p, B * K);
  t->d_kps_un = carve<dvm_keypoint_pod>(p, B * K);
  // mapped buffer: [query block: carved per call] [results]
  t->q_bytes = pad256(B * Q * 32) + 3 * pad256(B * Q * 4) + 2 * pad256(B * Q * 4) + pad256(B * Q) + pad256(B * Q * 4) + pad256(B * Q * 12) + pad256(B * 56) +
               pad256(B * 4) + pad256(64 * 4);
  const size_t mbytes = t->q_bytes + pad256(B * K * 4) + pad256(B * K) + pad256(B * 32) + pad256(B * 16) + pad256(B * 4) + pad256(B * 56) + pad256(B * 4) +
                        pad256(K * sizeof(dvm_keypoint_pod));
  if (hipHostMalloc(reinterpret_cast<void**>(&t->hm), mbytes, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&t->hm_dev), t->hm, 0) != hipSuccess) {
    dvm_tracker_destroy(t); set_error("dvm_tracker_create: mapped host memory"); return DVM_ERR_HIP;
  }
  t->hm_bytes = mbytes;
  std::memset(t->hm, 0, mbytes);
  p = t->hm + t->q_bytes;
  auto& m = t->m;
  m.assign = carve<int32_t>(p, B * K); m.outlier = carve<uint8_t>(p, B * K);
  m.res = carve<int32_t>(p, B * 8); m.fin = carve<int32_t>(p, B * 4); m.nedges = carve<int32_t>(p, B); m.pose_out = carve<double>(p, B * 7);
  m.n_inl = carve<int32_t>(p, B); m.kps_un = carve<dvm_keypoint_pod>(p, K);
  if (hipMalloc(reinterpret_cast<void**>(&t->d_q), t->q_bytes) != hipSuccess || hipStreamCreateWithFlags(&t->cstream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&t->cev, hipEventDisableTiming) != hipSuccess) {
    dvm_tracker_destroy(t); set_error("dvm_tracker_create: query block"); return DVM_ERR_HIP;
  }
  *out = t;
  return DVM_OK;
}
int dvm_tracker_create(int device, int max_keypoints, int max_queries, dvm_tracker** out) {
  return dvm_tracker_create_batch(device, 1, max_keypoints, max_queries, out);
}

void dvm_tracker_destroy(dvm_tracker* t) {
  if (!t) return;
  hipSetDevice(t->device);
  if (t->grid) dvm_frame_destroy(t->grid);
  if (t->d_buf) hipFree(t->d_buf);
  if (t->d_q) hipFree(t->d_q);
  if (t->cev) hipEventDestroy(t->cev);
  if (t->cstream) hipStreamDestroy(t->cstream);
  if (t->hm) hipHostFree(t->hm);
  delete t;
}

int dvm_track_begin_batch(dvm_tracker* t, dvm_orb* h, const uint8_t* imgs, int count, int rows, int cols, int stride, int64_t frame_stride, int lap0, int lap1) {
  if (!t || !h || count < 1) return DVM_ERR_INVALID;
  if (count > t->max_frames) { set_error("dvm_track_begin_batch: more frames than the tracker was created for"); return DVM_ERR_CAPACITY; }
  t->begun = 0;
  const int rc = dvm_orb_extract_batch_host(h, imgs, count, rows, cols, stride, frame_stride, lap0, lap1);
  if (rc != DVM_OK) return rc;
  t->begun = count; t->rows = rows; t->cols = cols;
  return DVM_OK;
}
int dvm_track_begin_staged(dvm_tracker* t, dvm_orb* h, int count, int rows, int cols, int lap0, int lap1) {
  if (!t || !h || count < 1) return DVM_ERR_INVALID;
  if (count > t->max_frames) { set_error("dvm_track_begin_staged: more frames than the tracker was created for"); return DVM_ERR_CAPACITY; }
  t->begun = 0;
  const int rc = dvm_orb_extract_staged(h, count, rows, cols, lap0, lap1);
  if (rc != DVM_OK) return rc;
  t->begun = count; t->rows = rows; t->cols = cols;
  return DVM_OK;
}
int dvm_track_begin(dvm_tracker* t, dvm_orb* h, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1) {
  return dvm_track_begin_batch(t, h, img, 1, rows, cols, stride, (int64_t)rows * stride, lap0, lap1);
}

int dvm_track_finish_batch(dvm_tracker* t, dvm_orb* h, int count, const dvm_track_queries* qs, const dvm_track_frame_out* outs, dvm_track_result* res) {
  if (!t || !h || !qs || !outs || !res || count < 1) return DVM_ERR_INVALID;
  if (t->begun != count) { set_error("dvm_track_finish: no matching dvm_track_begin on this tracker"); return DVM_ERR_STATE; }
  const dvm_track_queries& q0 = qs[0];
  int nq_max = 0;
  for (int b = 0; b < count; b++) {
    const dvm_track_queries& q = qs[b];
    const dvm_track_frame_out& o = outs[b];
    if (!o.kps || !o.desc || !o.assign || !o.outlier) return DVM_ERR_INVALID;
    if (q.nq < 0 || q.nq > t->q_cap || q.nlevels < 1 || q.nlevels > 64 || !q.inv_level_sigma2) { set_error("dvm_track_finish: bad query set"); return DVM_ERR_INVALID; }
    if (q.nq && (!q.qdesc || !q.qx || !q.qy || !q.qr || !q.qmin || !q.qmax || !q.q_claims || !q.q_angle || !q.q_pos)) return DVM_ERR_INVALID;
    if (b && (std::memcmp(q.bounds, q0.bounds, 16) != 0 || std::memcmp(&q.cam, &q0.cam, sizeof(q.cam)) != 0 || q.th_high != q0.th_high || q.check_ori != q0.check_ori ||
              q.min_matches != q0.min_matches || q.nlevels != q0.nlevels || std::memcmp(q.inv_level_sigma2, q0.inv_level_sigma2, (size_t)q.nlevels * 4) != 0)) {
      set_error("dvm_track_finish_batch: the frames of a batch share camera, bounds, level table and matcher thresholds");
      return DVM_ERR_INVALID;
    }
    if (count > 1 && q.dist && q.dist->k1 != 0.0f) { set_error("dvm_track_finish_batch: distorted keypoints take the single-frame call"); return DVM_ERR_INVALID; }
    nq_max = std::max(nq_max, q.nq);
    std::memset(&res[b], 0, sizeof(res[b]));
  }
  DVM_HIP(hipSetDevice(t->device));
  hipStream_t s = (hipStream_t)dvm_orb_stream(h);
  const dvm_keypoint* d_kps = nullptr; const uint8_t* d_desc = nullptr; const int32_t* d_n = nullptr; int ocap = 0;
  int rc = dvm_orb_result_device(h, 0, &d_kps, &d_desc, &d_n, &ocap);
  if (rc != DVM_OK) return rc;
  if (ocap > t->kp_cap) { set_error("dvm_track_finish: the extractor's keypoint capacity exceeds the tracker's"); return DVM_ERR_CAPACITY; }
  int64_t kps_stride = ocap, desc_stride = (int64_t)ocap * 32;
  if (count > 1) {     // the batch layout dvm_orb_result_device exposes: frame i at a constant stride
    const dvm_keypoint* k1 = nullptr; const uint8_t* d1 = nullptr; const int32_t* n1 = nullptr; int c1 = 0;
    if ((rc = dvm_orb_result_device(h, 1, &k1, &d1, &n1, &c1)) != DVM_OK) return rc;
    kps_stride = k1 - d_kps; desc_stride = d1 - d_desc;
    if (n1 != d_n + 1) { set_error("dvm_track_finish_batch: unexpected result layout"); return DVM_ERR_STATE; }
  }
  const int Qs = (std::max(nq_max, 1) + 63) & ~63;      // the per-query arrays' stride for this call
  auto& m = t->m;
  {   // the query block of THIS call, packed: one copy (~75 KB per frame of 1 000 queries)
    uint8_t* p = t->hm;
    const size_t Qn = (size_t)count * Qs;
    m.qdesc = carve<uint8_t>(p, Qn * 32); m.qx = carve<float>(p, Qn); m.qy = carve<float>(p, Qn); m.qr = carve<float>(p, Qn);
    m.qmin = carve<int32_t>(p, Qn); m.qmax = carve<int32_t>(p, Qn); m.q_claims = carve<uint8_t>(p, Qn); m.q_angle = carve<float>(p, Qn);
    m.q_pos = carve<float>(p, Qn * 3); m.pose_in = carve<double>(p, (size_t)count * 7); m.nq_arr = carve<int32_t>(p, count); m.inv_sigma2 = carve<float>(p, 64);
  }
  HostPool::get().run((size_t)count, count >= 4 ? 8 : 1, [&](size_t b) {       // (a batch's query blocks: 2.4 MB for 32 frames)
    const dvm_track_queries& q = qs[b];
    const size_t o = b * Qs, nq = (size_t)q.nq;
    std::memcpy(m.qdesc + o * 32, q.qdesc, nq * 32); std::memcpy(m.qx + o, q.qx, nq * 4); std::memcpy(m.qy + o, q.qy, nq * 4);
    std::memcpy(m.qr + o, q.qr, nq * 4); std::memcpy(m.qmin + o, q.qmin, nq * 4); std::memcpy(m.qmax + o, q.qmax, nq * 4);
    std::memcpy(m.q_claims + o, q.q_claims, nq); std::memcpy(m.q_angle + o, q.q_angle, nq * 4); std::memcpy(m.q_pos + o * 3, q.q_pos, nq * 12);
    std::memcpy(m.pose_in + 7 * b, q.pose_in, 56);
    m.nq_arr[b] = q.nq;
  });
  std::memcpy(m.inv_sigma2, q0.inv_level_sigma2, (size_t)q0.nlevels * 4);
  {
    const size_t used = (size_t)(reinterpret_cast<uint8_t*>(m.inv_sigma2) - t->hm) + pad256(64 * 4);
    DVM_HIP(hipMemcpyAsync(t->d_q, t->hm, used, hipMemcpyHostToDevice, t->cstream));   // beside the extraction, not behind it
    DVM_HIP(hipEventRecord(t->cev, t->cstream));
    DVM_HIP(hipStreamWaitEvent(s, t->cev, 0));
  }
  // mvKeysUn: the extractor's keypoints themselves without distortion (Frame.cc:791-797), else undistorted on the device (:799-818)
  const bool undist = count == 1 && q0.dist && q0.dist->k1 != 0.0f;
  const dvm_keypoint* d_un = d_kps;
  if (undist) {
    rc = dvm_undistort_keypoints(q0.dist, d_kps, reinterpret_cast<dvm_keypoint*>(t->d_kps_un), ocap, 1, s);
    if (rc != DVM_OK) return rc;
    d_un = reinterpret_cast<const dvm_keypoint*>(t->d_kps_un);
    if (outs[0].kps_un) DVM_HIP(hipMemcpyAsync(m.kps_un, t->d_kps_un, (size_t)ocap * sizeof(dvm_keypoint_pod), hipMemcpyDeviceToHost, s));
  }
  rc = dvm_frame_build_batch(t->grid, 0, count, d_un, kps_stride, d_desc, desc_stride, d_n, q0.bounds[0], q0.bounds[1], q0.bounds[2], q0.bounds[3], s);
  if (rc != DVM_OK) return rc;
  const FrameView FV = frame_view_of(t->grid);    // (bounds of the build above)
  launch_match_window_ranked_batch(s, FV, 0, count, nullptr, nullptr, 0, t->qdev(m.qdesc), t->qdev(m.qx), t->qdev(m.qy), t->qdev(m.qr), t->qdev(m.qmin),
                                   t->qdev(m.qmax), t->qdev(m.nq_arr), Qs, t->d_ranked);
  TrackBatch TB{count, Qs, kps_stride, t->qdev(m.nq_arr)};
  TrackRequery rq{};
  rq.F = FV;
  { static const bool no_rq = std::getenv("DVM_TRACK_NO_REQUERY") != nullptr; if (no_rq) rq.F.skp = nullptr; }   /* timing experiment only */
  rq.qdesc = t->qdev(m.qdesc); rq.qx = t->qdev(m.qx); rq.qy = t->qdev(m.qy); rq.qr = t->qdev(m.qr); rq.qmin = t->qdev(m.qmin); rq.qmax = t->qdev(m.qmax);
  launch_track_claims(s, t->d_ranked, t->qdev(m.q_claims), t->qdev(m.q_angle), 0, rq, reinterpret_cast<const dvm_keypoint_pod*>(d_un), d_n, ocap, q0.th_high,
                      q0.check_ori, t->d_assign, t->d_res, t->dev(m.assign), t->dev(m.res), TB);
  launch_track_gather(s, t->d_assign, reinterpret_cast<const dvm_keypoint_pod*>(d_un), d_n, ocap, t->qdev(m.q_pos), t->qdev(m.inv_sigma2), q0.nlevels, t->d_Xw,
                      t->d_obs, t->d_info, t->d_edge_kp, t->d_nedges, t->d_res, q0.min_matches, t->dev(m.nedges), TB);
  ba_launch_pose_optimize(s, t->qdev(m.pose_in), t->d_Xw, t->d_obs, t->d_info, t->d_nedges, ocap, count, q0.cam.fx, q0.cam.fy, q0.cam.cx, q0.cam.cy,
                          t->dev(m.pose_out), t->d_edge_out, t->dev(m.n_inl), t->d_chi);
  launch_track_finish(s, t->d_assign, d_n, ocap, t->d_edge_kp, t->d_nedges, t->d_edge_out, t->qdev(m.q_claims), t->dev(m.outlier), t->dev(m.fin), t->d_res, TB);
  // (what the host wants back is written to mapped memory by the kernels themselves: no copy command behind the chain)
  rc = hip_check(hipGetLastError(), "tracking chain launch");
  if (rc != DVM_OK) return rc;
  {   // the extraction's results of all frames: one synchronisation (the whole chain is through), block copies
    std::vector<dvm_keypoint*> kp(count); std::vector<uint8_t*> dp(count); std::vector<int> caps(count), ns(count), monos(count);
    for (int b = 0; b < count; b++) { kp[b] = outs[b].kps; dp[b] = outs[b].desc; caps[b] = outs[b].cap; }
    rc = dvm_orb_download_batch(h, count, kp.data(), dp.data(), caps.data(), ns.data(), monos.data());
    if (rc != DVM_OK) return rc;
    for (int b = 0; b < count; b++) { res[b].n = ns[b]; res[b].mono_index = monos[b]; }
  }
  for (int b = 0; b < count; b++) {
    const dvm_track_queries& q = qs[b];
    const dvm_track_frame_out& o = outs[b];
    dvm_track_result& r = res[b];
    const int n = r.n;
    if (o.kps_un) std::memcpy(o.kps_un, undist ? reinterpret_cast<const dvm_keypoint*>(m.kps_un) : o.kps, (size_t)n * sizeof(dvm_keypoint));
    const size_t ko = (size_t)b * ocap;
    std::memcpy(o.assign, m.assign + ko, (size_t)n * 4);
    const int32_t* rs = m.res + 8 * b;
    r.nmatches = rs[0];
    r.nmatches_before_rotation = rs[2];
    r.n_requeried = rs[3];
    if (std::getenv("DVM_TRACK_DEBUG")) std::fprintf(stderr, "track: frame %d nq %d rounds %d requeried %d\n", b, q.nq, rs[4], rs[3]);
    if (rs[1]) {                        // a query ran out of ranked candidates and could not be searched again on the device (DVM_TRACK_NO_REQUERY)
      r.status = DVM_TRACK_REPLAY_ON_HOST;
      if (o.ranked && q.nq) DVM_HIP(hipMemcpy(o.ranked, t->d_ranked + (size_t)b * Qs * 4, (size_t)q.nq * 16, hipMemcpyDeviceToHost));
      std::memset(o.outlier, 0, (size_t)n);
      continue;
    }
    if (r.nmatches < q.min_matches) { r.status = DVM_TRACK_FEW_MATCHES; std::memset(o.outlier, 0, (size_t)n); continue; }
    r.status = DVM_TRACK_COMPLETE;
    std::memcpy(o.outlier, m.outlier + ko, (size_t)n);
    r.n_edges = m.nedges[b]; r.n_inliers = m.n_inl[b]; r.nmatches_map = m.fin[4 * b]; r.nmatches_after = m.fin[4 * b + 1];
    std::memcpy(r.pose, m.pose_out + 7 * (size_t)b, 56);
  }
  return DVM_OK;
}

int dvm_track_finish(dvm_tracker* t, dvm_orb* h, const dvm_track_queries* q, dvm_keypoint* kps, uint8_t* desc, int cap, dvm_keypoint* kps_un,
                     int32_t* assign, uint8_t* outlier, uint32_t* ranked, dvm_track_result* res) {
  if (!q || !res) return DVM_ERR_INVALID;
  dvm_track_frame_out o{kps, desc, cap, kps_un, assign, outlier, ranked};
  return dvm_track_finish_batch(t, h, 1, q, &o, res);
}

}  // extern "C"
