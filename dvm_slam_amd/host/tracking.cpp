// dvm_slam_amd/host/tracking.cpp -- dvmh_track_with_motion_model (include/dvmslam_host.h): the host side of the one-chain tracking
// step.  Reference: Frame::Frame -> ExtractORB (src/Frame.cc:371-411), Tracking::TrackWithMotionModel (src/Tracking.cc:2584-2667).
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dvmslam_host.h"
#include "orb_matcher.h"

using dvm_host::FrameQueries;
using dvm_host::FrameView;

namespace {
constexpr int TH_HIGH = 100;   // ORBmatcher::TH_HIGH, ORBmatcher.cc:36
}

extern "C" int dvmh_track_with_motion_model(dvm_tracker* t, dvm_orb* h, int device, const uint8_t* img, int rows, int cols, int stride, int lap0,
                                            int lap1, const dvm_se3f* Tcw_pred, const float* K, const float* bounds, const dvm_distortion* dist,
                                            const float* scale_factors, const float* inv_level_sigma2, int nlevels, int Nl,
                                            const dvm_keypoint* kps_l, const int32_t* mp_l, const uint8_t* outlier_l, const dvmh_map_point* mps,
                                            float th, int check_ori, dvm_keypoint* kps, uint8_t* desc, int cap, dvm_keypoint* kps_un,
                                            int32_t* mp_c, int32_t* dropped, dvmh_track_result* out) {
  if (!t || !h || !img || !Tcw_pred || !K || !bounds || !scale_factors || !inv_level_sigma2 || !kps || !desc || !mp_c || !dropped || !out ||
      (Nl && (!kps_l || !mp_l || !mps)))
    return DVM_ERR_INVALID;
  std::memset(out, 0, sizeof(*out));
  // 1. the extraction is on its way ...
  int rc = dvm_track_begin(t, h, img, rows, cols, stride, lap0, lap1);
  if (rc != DVM_OK) return rc;
  // 2. ... while the queries are built (they need LastFrame and the predicted pose only)
  FrameView C, L;
  C.N = 0; C.Tcw = *Tcw_pred;
  C.fx = K[0]; C.fy = K[1]; C.cx = K[2]; C.cy = K[3];
  C.mnMinX = bounds[0]; C.mnMaxX = bounds[1]; C.mnMinY = bounds[2]; C.mnMaxY = bounds[3];
  C.mvScaleFactors = scale_factors; C.nLevels = nlevels;
  L = C;
  L.N = Nl; L.mvKeysUn = kps_l; L.mvpMapPoints = const_cast<int32_t*>(mp_l); L.mvbOutlier = outlier_l;
  std::vector<dvm_keypoint> un_local;
  if (!kps_un) { un_local.resize((size_t)cap); kps_un = un_local.data(); }
  std::vector<int32_t> assign((size_t)cap);
  std::vector<uint8_t> outl((size_t)cap);
  std::vector<uint32_t> ranked;
  FrameQueries Q;
  std::vector<uint8_t> q_claims;
  std::vector<float> q_angle, q_pos;
  dvm_track_queries tq;
  dvm_track_result tr;
  float th_now = th;
  for (int attempt = 0; attempt < 2; attempt++) {
    dvm_host::BuildFrameQueries(C, L, mps, th_now, Q);
    const int nq = (int)Q.qi.size();
    q_claims.resize((size_t)nq); q_angle.resize((size_t)nq); q_pos.resize((size_t)nq * 3);
    for (int q = 0; q < nq; q++) {
      const int i = Q.qi[q], mp = mp_l[i];
      q_claims[q] = mps[mp].n_obs > 0;
      q_angle[q] = kps_l[i].angle;
      q_pos[3 * q] = mps[mp].pos[0]; q_pos[3 * q + 1] = mps[mp].pos[1]; q_pos[3 * q + 2] = mps[mp].pos[2];
    }
    ranked.resize((size_t)nq * 4 + 4);
    std::memset(&tq, 0, sizeof(tq));
    tq.nq = nq; tq.qdesc = Q.qdesc.data(); tq.qx = Q.qx.data(); tq.qy = Q.qy.data(); tq.qr = Q.qr.data(); tq.qmin = Q.qmin.data(); tq.qmax = Q.qmax.data();
    tq.q_claims = q_claims.data(); tq.q_angle = q_angle.data(); tq.q_pos = q_pos.data();
    std::memcpy(tq.bounds, bounds, 16);
    tq.dist = dist; tq.inv_level_sigma2 = inv_level_sigma2; tq.nlevels = nlevels;
    tq.cam.fx = K[0]; tq.cam.fy = K[1]; tq.cam.cx = K[2]; tq.cam.cy = K[3]; tq.cam.huber_delta = 0.0;
    // g2o::SE3Quat(Tcw.unit_quaternion().cast<double>(), Tcw.translation().cast<double>()) (Optimizer.cc:760)
    for (int k = 0; k < 3; k++) tq.pose_in[k] = (double)Tcw_pred->t[k];
    for (int k = 0; k < 4; k++) tq.pose_in[3 + k] = (double)Tcw_pred->q[k];
    tq.th_high = TH_HIGH; tq.check_ori = check_ori; tq.min_matches = 20;
    rc = dvm_track_finish(t, h, &tq, kps, desc, cap, kps_un, assign.data(), outl.data(), ranked.data(), &tr);
    if (rc != DVM_OK) return rc;
    out->n = tr.n; out->mono_index = tr.mono_index; out->n_requeried = tr.n_requeried;
    if (tr.status != DVM_TRACK_FEW_MATCHES || attempt == 1) break;
    th_now = 2 * th;                       // Tracking.cc:2616-2624: "Not enough matches, wider window search"
    out->wide_window = 1;
  }
  const int N = tr.n;
  for (int j = 0; j < N; j++) { mp_c[j] = -1; dropped[j] = -1; }
  if (tr.status == DVM_TRACK_REPLAY_ON_HOST) {
    // rare: a query ran out of ranked candidates.  The separate calls take over from the frame's host arrays (same results by
    // construction: they are what the chain is tested against)
    out->replayed_on_host = 1;
    C.N = N; C.mvKeysUn = kps_un; C.mDescriptors = desc; C.mvpMapPoints = mp_c;
    dvm_host::ORBmatcher m(0.9f, check_ori != 0, device);
    int nm = m.SearchByProjection(C, L, mps, th_now, true);
    if (nm < 0) return nm;
    if (nm < 20 && !out->wide_window) {
      for (int j = 0; j < N; j++) mp_c[j] = -1;
      out->wide_window = 1;
      nm = m.SearchByProjection(C, L, mps, 2 * th, true);
      if (nm < 0) return nm;
    }
    out->nmatches_search = nm;
    if (nm < 20) { out->nmatches = nm; out->tracked = 0; out->Tcw = *Tcw_pred; return DVM_OK; }
    std::vector<double> Xw, obs, w;
    std::vector<int> kp_of;
    for (int j = 0; j < N; j++) {
      if (mp_c[j] < 0) continue;
      for (int k = 0; k < 3; k++) Xw.push_back((double)mps[mp_c[j]].pos[k]);
      obs.push_back((double)kps_un[j].x); obs.push_back((double)kps_un[j].y);
      w.push_back((double)inv_level_sigma2[kps_un[j].octave]);
      kp_of.push_back(j);
    }
    const int32_t ne = (int32_t)kp_of.size();
    std::vector<uint8_t> rej((size_t)ne);
    int32_t inl = 0;
    rc = dvm_pose_optimize(device, tq.pose_in, Xw.data(), obs.data(), w.data(), &ne, ne, 1, &tq.cam, out->pose, rej.data(), &inl);
    if (rc != DVM_OK) return rc;
    out->n_inliers = inl;
    int left = nm, nmap = 0;
    for (int e = 0; e < ne; e++) {
      const int j = kp_of[e];
      if (rej[e]) { dropped[j] = mp_c[j]; mp_c[j] = -1; left--; }
      else if (mps[mp_c[j]].n_obs > 0) nmap++;
    }
    out->nmatches = left; out->nmatches_map = nmap; out->tracked = 1;
  } else {
    out->nmatches_search = tr.nmatches;
    if (tr.status == DVM_TRACK_FEW_MATCHES) {    // not tracked: the matches of the (doubled) search stay as SearchByProjection left them
      for (int j = 0; j < N; j++) if (assign[j] >= 0) mp_c[j] = mp_l[Q.qi[assign[j]]];
      out->nmatches = tr.nmatches; out->tracked = 0; out->Tcw = *Tcw_pred;
      for (int k = 0; k < 7; k++) out->pose[k] = tq.pose_in[k];
      return DVM_OK;
    }
    for (int j = 0; j < N; j++) {
      if (assign[j] < 0) continue;
      const int mp = mp_l[Q.qi[assign[j]]];
      if (outl[j]) dropped[j] = mp; else mp_c[j] = mp;
    }
    out->nmatches = tr.nmatches_after; out->nmatches_map = tr.nmatches_map; out->n_inliers = tr.n_inliers; out->tracked = 1;
    std::memcpy(out->pose, tr.pose, 56);
  }
  // Sophus::SE3f(SE3quat_recov.rotation().cast<float>(), SE3quat_recov.translation().cast<float>()) (Optimizer.cc:1023-1025)
  for (int k = 0; k < 3; k++) out->Tcw.t[k] = (float)out->pose[k];
  for (int k = 0; k < 4; k++) out->Tcw.q[k] = (float)out->pose[3 + k];
  return DVM_OK;
}


// ---- K agents' frames at one camera tick: dvmh_track_with_motion_model for `count` frames through ONE chain of batched launches
// (dvm_track_begin_batch / dvm_track_finish_batch).  Per frame exactly the steps of the single call; the queries of the frames are
// built by a few host threads while the batch extraction runs.
namespace {
struct AgentQueries {
  FrameQueries Q;
  std::vector<uint8_t> q_claims;
  std::vector<float> q_angle, q_pos;
  FrameView C, L;
  void build(const dvmh_track_in& in, const float* K, const float* bounds, const float* scale_factors, int nlevels, float th) {
    C = FrameView();
    C.N = 0; C.Tcw = *in.Tcw_pred;
    C.fx = K[0]; C.fy = K[1]; C.cx = K[2]; C.cy = K[3];
    C.mnMinX = bounds[0]; C.mnMaxX = bounds[1]; C.mnMinY = bounds[2]; C.mnMaxY = bounds[3];
    C.mvScaleFactors = scale_factors; C.nLevels = nlevels;
    L = C;
    L.N = in.Nl; L.mvKeysUn = in.kps_l; L.mvpMapPoints = const_cast<int32_t*>(in.mp_l); L.mvbOutlier = in.outlier_l;
    dvm_host::BuildFrameQueries(C, L, in.mps, th, Q);
    const int nq = (int)Q.qi.size();
    q_claims.resize((size_t)nq); q_angle.resize((size_t)nq); q_pos.resize((size_t)nq * 3);
    for (int q = 0; q < nq; q++) {
      const int i = Q.qi[q], mp = in.mp_l[i];
      q_claims[q] = in.mps[mp].n_obs > 0;
      q_angle[q] = in.kps_l[i].angle;
      q_pos[3 * q] = in.mps[mp].pos[0]; q_pos[3 * q + 1] = in.mps[mp].pos[1]; q_pos[3 * q + 2] = in.mps[mp].pos[2];
    }
  }
  void fill(dvm_track_queries& tq, const dvmh_track_in& in, const float* K, const float* bounds, const float* inv_level_sigma2, int nlevels, int check_ori) const {
    std::memset(&tq, 0, sizeof(tq));
    tq.nq = (int)Q.qi.size(); tq.qdesc = Q.qdesc.data(); tq.qx = Q.qx.data(); tq.qy = Q.qy.data(); tq.qr = Q.qr.data(); tq.qmin = Q.qmin.data(); tq.qmax = Q.qmax.data();
    tq.q_claims = q_claims.data(); tq.q_angle = q_angle.data(); tq.q_pos = q_pos.data();
    std::memcpy(tq.bounds, bounds, 16);
    tq.dist = nullptr; tq.inv_level_sigma2 = inv_level_sigma2; tq.nlevels = nlevels;
    tq.cam.fx = K[0]; tq.cam.fy = K[1]; tq.cam.cx = K[2]; tq.cam.cy = K[3]; tq.cam.huber_delta = 0.0;
    for (int k = 0; k < 3; k++) tq.pose_in[k] = (double)in.Tcw_pred->t[k];
    for (int k = 0; k < 4; k++) tq.pose_in[3 + k] = (double)in.Tcw_pred->q[k];
    tq.th_high = TH_HIGH; tq.check_ori = check_ori; tq.min_matches = 20;
  }
};
}  // namespace

#include <thread>

namespace {
// A few persistent host threads for the per-tick work of the batched chain (the agents' query lists are independent): run(n, width, fn) calls
// fn(i) for i in [0, n) on the caller and up to width - 1 workers and returns when all are done.  One job at a time; never destroyed (its
// sleeping workers end with the process -- a destructor that joined them would run in forked children, where they do not exist).
class TickPool {
 public:
  static TickPool& get() { static TickPool* p = new TickPool; return *p; }
  template <class F> void run(int n, int width, F&& fn) {
    if (n <= 0) return;
    if (width <= 1 || n == 1) { for (int i = 0; i < n; i++) fn(i); return; }
    std::lock_guard<std::mutex> job(job_mtx_);
    std::function<void(int)> f = fn;
    {
      std::lock_guard<std::mutex> l(mtx_);
      const int want = std::min(std::min(width - 1, n - 1), 31);
      while ((int)workers_.size() < want) spawn();
      fn_ = &f; n_ = n; next_.store(0); active_ = want; pending_ = want; gen_++;
    }
    cv_.notify_all();
    for (;;) { const int i = next_.fetch_add(1); if (i >= n) break; f(i); }
    std::unique_lock<std::mutex> l(mtx_);
    done_.wait(l, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }
 private:
  std::mutex job_mtx_, mtx_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> workers_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_ = 0, active_ = 0, pending_ = 0;
  std::atomic<int> next_{0};
  unsigned long gen_ = 0;
  void spawn() {
    const int id = (int)workers_.size();
    workers_.emplace_back([this, id] {
      unsigned long seen = 0;
      for (;;) {
        std::unique_lock<std::mutex> l(mtx_);
        cv_.wait(l, [&] { return gen_ != seen && id < active_; });
        seen = gen_;
        const std::function<void(int)>* f = fn_;
        const int n = n_;
        l.unlock();
        for (;;) { const int i = next_.fetch_add(1); if (i >= n) break; (*f)(i); }
        l.lock();
        if (--pending_ == 0) done_.notify_one();
      }
    });
    workers_.back().detach();
  }
};
}  // namespace

extern "C" int dvmh_track_with_motion_model_batch(dvm_tracker* t, dvm_orb* h, int device, int count, const uint8_t* imgs, int rows, int cols, int stride,
                                                  int64_t frame_stride, int lap0, int lap1, const float* K, const float* bounds, const float* scale_factors,
                                                  const float* inv_level_sigma2, int nlevels, float th, int check_ori, const dvmh_track_in* in,
                                                  const dvmh_track_out* outs, dvmh_track_result* res) {
  if (!t || !h || !K || !bounds || !scale_factors || !inv_level_sigma2 || !in || !outs || !res || count < 1) return DVM_ERR_INVALID;
  for (int b = 0; b < count; b++) {
    if (!in[b].Tcw_pred || (in[b].Nl && (!in[b].kps_l || !in[b].mp_l || !in[b].mps)) || !outs[b].kps || !outs[b].desc || !outs[b].mp_c || !outs[b].dropped) return DVM_ERR_INVALID;
    std::memset(&res[b], 0, sizeof(res[b]));
  }
  (void)device;
  static const bool timing = std::getenv("DVM_TRACK_BATCH_TIMING") != nullptr;       // host-side phase times of a tick on stderr
  using clk = std::chrono::steady_clock;
  clk::time_point tp0 = clk::now(), tp1, tp2, tp3, tp4;
  int rc = imgs ? dvm_track_begin_batch(t, h, imgs, count, rows, cols, stride, frame_stride, lap0, lap1) : dvm_track_begin_staged(t, h, count, rows, cols, lap0, lap1);
  if (rc != DVM_OK) return rc;
  tp1 = clk::now();
  std::vector<AgentQueries> AQ((size_t)count);
  // (persistent workers: spawning eight threads per tick and giving each four agents' queries took 0.45 ms of a 1.6 ms tick of 32 frames --
  //  longer than the extraction it was meant to run under)
  auto build_all = [&](float scale, const std::vector<uint8_t>* only) {
    TickPool::get().run(count, 24, [&](int b) { if (!only || (*only)[b]) AQ[b].build(in[b], K, bounds, scale_factors, nlevels, scale * th); });
  };
  build_all(1.0f, nullptr);
  tp2 = clk::now();
  std::vector<dvm_track_queries> tq((size_t)count);
  std::vector<dvm_track_frame_out> fo((size_t)count);
  std::vector<dvm_track_result> tr((size_t)count);
  std::vector<std::vector<int32_t>> assign((size_t)count);
  std::vector<std::vector<uint8_t>> outl((size_t)count);
  std::vector<std::vector<dvm_keypoint>> un_local((size_t)count);
  for (int b = 0; b < count; b++) {
    const int cap = outs[b].cap;
    assign[b].resize((size_t)cap); outl[b].resize((size_t)cap);
    dvm_keypoint* un = outs[b].kps_un;
    if (!un) { un_local[b].resize((size_t)cap); un = un_local[b].data(); }
    fo[b] = dvm_track_frame_out{outs[b].kps, outs[b].desc, cap, un, assign[b].data(), outl[b].data(), nullptr};
  }
  std::vector<uint8_t> wide((size_t)count, 0);
  for (int attempt = 0; attempt < 2; attempt++) {
    for (int b = 0; b < count; b++) AQ[b].fill(tq[b], in[b], K, bounds, inv_level_sigma2, nlevels, check_ori);
    if (attempt == 0) tp3 = clk::now();
    rc = dvm_track_finish_batch(t, h, count, tq.data(), fo.data(), tr.data());
    if (rc != DVM_OK) return rc;
    if (attempt == 0) tp4 = clk::now();
    bool any = false;
    if (attempt == 0)
      for (int b = 0; b < count; b++) if (tr[b].status == DVM_TRACK_FEW_MATCHES) { wide[b] = 1; any = true; }
    if (!any) break;
    build_all(2.0f, &wide);          // Tracking.cc:2616-2624: "Not enough matches, wider window search" -- for the frames that need it
  }
  for (int b = 0; b < count; b++) {
    dvmh_track_result* out = &res[b];
    const dvm_track_result& r = tr[b];
    const int N = r.n;
    int32_t* mp_c = outs[b].mp_c; int32_t* dropped = outs[b].dropped;
    out->n = r.n; out->mono_index = r.mono_index; out->n_requeried = r.n_requeried; out->wide_window = wide[b];
    for (int j = 0; j < N; j++) { mp_c[j] = -1; dropped[j] = -1; }
    out->nmatches_search = r.nmatches;
    if (r.status != DVM_TRACK_COMPLETE) {     // not tracked: the matches of the (doubled) search stay as SearchByProjection left them
      for (int j = 0; j < N; j++) if (assign[b][j] >= 0) mp_c[j] = in[b].mp_l[AQ[b].Q.qi[assign[b][j]]];
      out->nmatches = r.nmatches; out->tracked = 0; out->Tcw = *in[b].Tcw_pred;
      for (int k = 0; k < 3; k++) out->pose[k] = (double)in[b].Tcw_pred->t[k];
      for (int k = 0; k < 4; k++) out->pose[3 + k] = (double)in[b].Tcw_pred->q[k];
      continue;
    }
    for (int j = 0; j < N; j++) {
      if (assign[b][j] < 0) continue;
      const int mp = in[b].mp_l[AQ[b].Q.qi[assign[b][j]]];
      if (outl[b][j]) dropped[j] = mp; else mp_c[j] = mp;
    }
    out->nmatches = r.nmatches_after; out->nmatches_map = r.nmatches_map; out->n_inliers = r.n_inliers; out->tracked = 1;
    std::memcpy(out->pose, r.pose, 56);
    for (int k = 0; k < 3; k++) out->Tcw.t[k] = (float)out->pose[k];
    for (int k = 0; k < 4; k++) out->Tcw.q[k] = (float)out->pose[3 + k];
  }
  if (timing) {
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "track batch of %d: begin (images in + enqueue) %.3f  queries %.3f  tables + fill %.3f  finish (wait + results out) %.3f  rest %.3f ms\n", count,
                 ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3), ms(tp3, tp4), ms(tp4, clk::now()));
  }
  return DVM_OK;
}
