// tools/online_agents.cpp -- K agents tracking on ONE GPU through the drop-in C ABI, one host thread and one extractor handle each:
// per frame dvm_orb_extract (ORBextractor::operator()) -> dvmh_search_by_projection_frames (SearchByProjection(Cur, Last)) ->
// dvm_pose_optimize (PoseOptimization), host arrays in and out, every call synchronous as Tracking makes them (Tracking.cc:1423-1426,
// :2610, :2632).  The Python leg of the same name measures the same thing through ctypes and is bound by the interpreter lock above
// ~5 k frames/s; this is the C++ host the reference is.  Inputs come from a file bench_legs.online_agents writes (frames, the search
// inputs of every frame pair, the pose cases); every thread's results are reduced to a checksum and must equal thread 0's.
//   usage: online_agents <inputs.bin> <device> <frames_per_agent> K [K ...]     -> one JSON object on stdout
// Build: g++ -O2 -std=c++17 tools/online_agents.cpp -Iinclude -Ldvm_slam_amd/lib -ldvmslam_host -ldvmslam_hip -lpthread
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "dvmslam_hip.h"
#include "dvmslam_host.h"

namespace {
struct Pair {
  int Nc = 0, Nl = 0;
  std::vector<dvm_keypoint> kc, kl;
  std::vector<uint8_t> dc;
  std::vector<dvmh_map_point> mps;
};
struct PoseCase {
  int n = 0;
  double pose[7];
  std::vector<double> X, obs, w;
  dvm_ba_camera cam;
};
struct Inputs {
  int cyc = 0, rows = 0, cols = 0;
  std::vector<uint8_t> frames;   // (cyc + 1) x rows x cols
  float scale[8];
  std::vector<Pair> pairs;       // cyc
  std::vector<PoseCase> poses;   // cyc
};
template <class T> bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }
bool load(const char* path, Inputs& in) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  int32_t h[3];
  bool ok = rd(f, h, 3);
  in.cyc = h[0]; in.rows = h[1]; in.cols = h[2];
  in.frames.resize((size_t)(in.cyc + 1) * in.rows * in.cols);
  ok = ok && rd(f, in.frames.data(), in.frames.size()) && rd(f, in.scale, 8);
  for (int t = 0; ok && t < in.cyc; t++) {
    Pair p;
    int32_t n2[2];
    ok = rd(f, n2, 2);
    p.Nc = n2[0]; p.Nl = n2[1];
    p.kc.resize(p.Nc); p.dc.resize((size_t)p.Nc * 32); p.kl.resize(p.Nl); p.mps.resize(p.Nl);
    ok = ok && rd(f, p.kc.data(), p.Nc) && rd(f, p.dc.data(), p.dc.size()) && rd(f, p.kl.data(), p.Nl) && rd(f, p.mps.data(), p.Nl);
    in.pairs.push_back(std::move(p));
  }
  for (int t = 0; ok && t < in.cyc; t++) {
    PoseCase c;
    int32_t n;
    ok = rd(f, &n, 1);
    c.n = n;
    c.X.resize((size_t)n * 3); c.obs.resize((size_t)n * 2); c.w.resize(n);
    double k4[4];
    ok = ok && rd(f, c.pose, 7) && rd(f, c.X.data(), c.X.size()) && rd(f, c.obs.data(), c.obs.size()) && rd(f, c.w.data(), c.w.size()) && rd(f, k4, 4);
    c.cam = dvm_ba_camera{k4[0], k4[1], k4[2], k4[3], 0.0};
    in.poses.push_back(std::move(c));
  }
  fclose(f);
  return ok;
}
uint64_t mix(uint64_t h, const void* p, size_t n) {   // FNV-1a
  const uint8_t* b = static_cast<const uint8_t*>(p);
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
struct Gate {
  std::mutex m; std::condition_variable cv; int waiting = 0, gen = 0, parties = 0;
  void wait() {
    std::unique_lock<std::mutex> l(m);
    const int g = gen;
    if (++waiting == parties) { waiting = 0; gen++; cv.notify_all(); } else cv.wait(l, [&] { return gen != g; });
  }
};
}  // namespace

// one measurement: K agent threads, `nframes` frames each; pool != nullptr: extraction through the shared extractor (dvm_orb_pool_extract)
struct Result { double fps = 0, ms_per_frame = 0, mean[3] = {0, 0, 0}, med[3] = {0, 0, 0}, mean_batch = 0; int rc = 0; uint64_t sum0 = 0; bool same = true; };

static Result measure(const Inputs& in, int device, int nframes, int K, dvm_orb_pool* pool, dvm_pose_pool* ppool, bool one_chain = false) {
  const float K4[4] = {500.f, 500.f, 320.f, 240.f}, bounds[4] = {0.f, 640.f, 0.f, 480.f};
  const dvm_se3f Tcw{{0.f, 0.f, 0.f, 1.f}, {0.f, 0.f, 0.f}};
  Gate gate; gate.parties = K + 1;
  std::vector<uint64_t> sums(K, 0);
  std::vector<int> rcs(K, 0);
  double call_ms[3] = {0, 0, 0};   // agent 0: time inside each of the three calls (mean), and their medians
  std::vector<double> call_all[3];
  std::atomic<long> batch_frames{0}, batch_calls{0};
  auto agent = [&](int id) {
    dvm_set_device(device);
    dvm_orb_params P{1000, 1.2f, 8, 20, 7};
    dvm_orb* h = nullptr;
    int rc = pool ? 0 : dvm_orb_create(&P, device, 1, &h);
    const int cap = 4096;
    std::vector<dvm_keypoint> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int n = 0, mono = 0, bs = 0;
    const size_t fb = (size_t)in.rows * in.cols;
    float inv_s2[8];
    for (int l = 0; l < 8; l++) inv_s2[l] = 1.0f / (in.scale[l] * in.scale[l]);
    dvm_tracker* trk = nullptr;
    if (one_chain && rc == 0) rc = dvm_tracker_create(device, cap, cap, &trk);
    std::vector<dvm_keypoint> kun(cap);
    std::vector<int32_t> mp_t(cap), dropped(cap);
    dvmh_track_result tr{};
    auto track = [&](const uint8_t* img, const Pair& p, const int32_t* mpl) {   // the whole tracked frame as one device chain
      return dvmh_track_with_motion_model(trk, h, device, img, in.rows, in.cols, in.cols, 0, 1000, &Tcw, K4, bounds, nullptr, in.scale, inv_s2, 8, p.Nl,
                                          p.kl.data(), mpl, nullptr, p.mps.data(), 15.0f, 1, kps.data(), desc.data(), cap, kun.data(), mp_t.data(),
                                          dropped.data(), &tr);
    };
    auto extract = [&](const uint8_t* img) {
      return pool ? dvm_orb_pool_extract(pool, img, in.rows, in.cols, in.cols, 0, 1000, kps.data(), desc.data(), cap, &n, &mono, &bs)
                  : dvm_orb_extract(h, img, in.rows, in.cols, in.cols, 0, 1000, kps.data(), desc.data(), cap, &n, &mono);
    };
    auto pose_opt = [&](const PoseCase& c, double* po, uint8_t* ol, int32_t* ni) {
      int32_t nn = c.n;
      return ppool ? dvm_pose_pool_optimize(ppool, c.pose, c.X.data(), c.obs.data(), c.w.data(), c.n, &c.cam, po, ol, ni, nullptr)
                   : dvm_pose_optimize(device, c.pose, c.X.data(), c.obs.data(), c.w.data(), &nn, c.n, 1, &c.cam, po, ol, ni);
    };
    std::vector<int32_t> mp_c, mp_l;
    {  // one untimed frame: the extractor's buffers, this thread's grid handle and staging context (pinned allocations, a stream) exist
      if (rc == 0) rc = extract(in.frames.data());
      const Pair& p = in.pairs[0];
      mp_c.assign(p.Nc, -1); mp_l.resize(p.Nl);
      for (int j = 0; j < p.Nl; j++) mp_l[j] = j;
      if (rc == 0) {
        const int nm = dvmh_search_by_projection_frames(device, p.Nc, p.kc.data(), p.dc.data(), mp_c.data(), &Tcw, K4, bounds, in.scale, 8, p.Nl, p.kl.data(),
                                                        mp_l.data(), nullptr, p.mps.data(), 15.0f, 1, nullptr);
        if (nm < 0) rc = nm;
      }
      const PoseCase& c = in.poses[0];
      double po[7]; std::vector<uint8_t> ol(c.n); int32_t ni = 0;
      if (rc == 0) rc = pose_opt(c, po, ol.data(), &ni);
      if (rc == 0 && one_chain) rc = track(in.frames.data() + fb, p, mp_l.data());
    }
    gate.wait();   // all agents ready
    gate.wait();   // clock started
    uint64_t sum = 1469598103934665603ull;
    for (int i = 0; i < nframes && rc == 0; i++) {
      const int t = 1 + i % in.cyc;
      const auto c0 = std::chrono::steady_clock::now();
      if (one_chain) {
        const Pair& p = in.pairs[t - 1];
        mp_l.resize(p.Nl);
        for (int j = 0; j < p.Nl; j++) mp_l[j] = j;
        rc = track(in.frames.data() + (size_t)t * fb, p, mp_l.data());
        if (rc) break;
        if (id == 0) {
          const double d = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count();
          call_ms[0] += d; call_all[0].push_back(d);
        }
        if (i < in.cyc) {
          sum = mix(sum, &tr.n, 4); sum = mix(sum, kps.data(), (size_t)tr.n * sizeof(dvm_keypoint)); sum = mix(sum, desc.data(), (size_t)tr.n * 32);
          sum = mix(sum, &tr.nmatches, 4); sum = mix(sum, mp_t.data(), (size_t)tr.n * 4); sum = mix(sum, tr.pose, sizeof(tr.pose));
        }
        continue;
      }
      rc = extract(in.frames.data() + (size_t)t * fb);
      if (rc) break;
      if (pool) { batch_frames += bs; batch_calls += 1; }
      const auto c1 = std::chrono::steady_clock::now();
      const Pair& p = in.pairs[t - 1];
      mp_c.assign(p.Nc, -1);
      mp_l.resize(p.Nl);
      for (int j = 0; j < p.Nl; j++) mp_l[j] = j;
      const int nm = dvmh_search_by_projection_frames(device, p.Nc, p.kc.data(), p.dc.data(), mp_c.data(), &Tcw, K4, bounds, in.scale, 8, p.Nl,
                                                      p.kl.data(), mp_l.data(), nullptr, p.mps.data(), 15.0f, 1, nullptr);
      if (nm < 0) { rc = nm; break; }
      const auto c2 = std::chrono::steady_clock::now();
      const PoseCase& c = in.poses[t - 1];
      double pose_out[7];
      std::vector<uint8_t> outl(c.n);
      int32_t ninl = 0;
      rc = pose_opt(c, pose_out, outl.data(), &ninl);
      if (rc) break;
      if (id == 0) {
        const auto c3 = std::chrono::steady_clock::now();
        const double d[3] = {std::chrono::duration<double, std::milli>(c1 - c0).count(), std::chrono::duration<double, std::milli>(c2 - c1).count(),
                             std::chrono::duration<double, std::milli>(c3 - c2).count()};
        for (int q = 0; q < 3; q++) { call_ms[q] += d[q]; call_all[q].push_back(d[q]); }
      }
      if (i < in.cyc) {   // one cycle of results -> checksum
        sum = mix(sum, &n, 4); sum = mix(sum, kps.data(), (size_t)n * sizeof(dvm_keypoint)); sum = mix(sum, desc.data(), (size_t)n * 32);
        sum = mix(sum, &nm, 4); sum = mix(sum, mp_c.data(), mp_c.size() * 4); sum = mix(sum, pose_out, sizeof(pose_out)); sum = mix(sum, outl.data(), outl.size());
      }
    }
    sums[id] = sum; rcs[id] = rc;
    if (trk) dvm_tracker_destroy(trk);
    if (h) dvm_orb_destroy(h);
  };
  std::vector<std::thread> th;
  for (int k = 0; k < K; k++) th.emplace_back(agent, k);
  gate.wait();
  const auto t0 = std::chrono::steady_clock::now();
  gate.wait();
  for (auto& t : th) t.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  Result R;
  for (int k = 0; k < K; k++) {
    if (rcs[k]) { std::fprintf(stderr, "agent %d of %d failed: rc %d (%s)\n", k, K, rcs[k], dvm_last_error()); R.rc = rcs[k]; }
    R.same = R.same && sums[k] == sums[0];
  }
  R.sum0 = sums[0];
  R.fps = K * (double)nframes / dt; R.ms_per_frame = dt / nframes * 1e3;
  for (int q = 0; q < 3; q++) {
    R.mean[q] = call_ms[q] / nframes;
    if (!call_all[q].empty()) { std::sort(call_all[q].begin(), call_all[q].end()); R.med[q] = call_all[q][call_all[q].size() / 2]; }
  }
  R.mean_batch = batch_calls ? (double)batch_frames / (double)batch_calls : 0;
  return R;
}

// K agents' frames per camera tick through ONE batched chain (dvmh_track_with_motion_model_batch), one host thread
static Result measure_batched(const Inputs& in, int device, int nframes, int K, uint64_t* sum_agent0) {
  Result R;
  const float K4[4] = {500.f, 500.f, 320.f, 240.f}, bounds[4] = {0.f, 640.f, 0.f, 480.f};
  const dvm_se3f Tcw{{0.f, 0.f, 0.f, 1.f}, {0.f, 0.f, 0.f}};
  dvm_set_device(device);
  dvm_orb_params P{1000, 1.2f, 8, 20, 7};
  dvm_orb* h = nullptr;
  dvm_tracker* trk = nullptr;
  const int cap = 4096;
  int rc = dvm_orb_create(&P, device, K, &h);
  if (rc == 0) rc = dvm_tracker_create_batch(device, K, cap, cap, &trk);
  if (rc) { R.rc = rc; return R; }
  float inv_s2[8];
  for (int l = 0; l < 8; l++) inv_s2[l] = 1.0f / (in.scale[l] * in.scale[l]);
  const size_t fb = (size_t)in.rows * in.cols;
  std::vector<std::vector<dvm_keypoint>> kps(K, std::vector<dvm_keypoint>(cap)), kun(K, std::vector<dvm_keypoint>(cap));
  std::vector<std::vector<uint8_t>> desc(K, std::vector<uint8_t>((size_t)cap * 32));
  std::vector<std::vector<int32_t>> mp_t(K, std::vector<int32_t>(cap)), dropped(K, std::vector<int32_t>(cap)), mp_l(K);
  std::vector<dvmh_track_in> tin(K);
  std::vector<dvmh_track_out> tout(K);
  std::vector<dvmh_track_result> tr(K);
  for (int a = 0; a < K; a++) tout[a] = dvmh_track_out{kps[a].data(), desc[a].data(), cap, kun[a].data(), mp_t[a].data(), dropped[a].data()};
  uint64_t sum = 1469598103934665603ull;
  auto tick = [&](int i) {
    uint8_t* staging = nullptr;        // the cameras' frames go straight into the extractor's page-locked input buffer
    if (dvm_orb_staging(h, K, in.rows, in.cols, &staging) != 0) return -1;
    for (int a = 0; a < K; a++) {      // agent a is `a` frames ahead in the cycle: different frames and maps in one batch
      const int t = 1 + (i + a) % in.cyc;
      std::memcpy(staging + (size_t)a * fb, in.frames.data() + (size_t)t * fb, fb);
      const Pair& p = in.pairs[t - 1];
      mp_l[a].resize(p.Nl);
      for (int j = 0; j < p.Nl; j++) mp_l[a][j] = j;
      tin[a] = dvmh_track_in{&Tcw, p.Nl, p.kl.data(), mp_l[a].data(), nullptr, p.mps.data()};
    }
    return dvmh_track_with_motion_model_batch(trk, h, device, K, nullptr, in.rows, in.cols, in.cols, (int64_t)fb, 0, 1000, K4, bounds, in.scale, inv_s2, 8, 15.0f, 1,
                                              tin.data(), tout.data(), tr.data());
  };
  rc = tick(0);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < nframes && rc == 0; i++) {
    rc = tick(i);
    if (rc == 0 && i < in.cyc) {       // agent 0's results over one cycle -> the checksum the one-chain mode forms
      sum = mix(sum, &tr[0].n, 4); sum = mix(sum, kps[0].data(), (size_t)tr[0].n * sizeof(dvm_keypoint)); sum = mix(sum, desc[0].data(), (size_t)tr[0].n * 32);
      sum = mix(sum, &tr[0].nmatches, 4); sum = mix(sum, mp_t[0].data(), (size_t)tr[0].n * 4); sum = mix(sum, tr[0].pose, sizeof(tr[0].pose));
    }
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (rc) std::fprintf(stderr, "batched chain failed: rc %d (%s)\n", rc, dvm_last_error());
  R.rc = rc; R.fps = K * (double)nframes / dt; R.ms_per_frame = dt / nframes * 1e3;
  *sum_agent0 = sum;
  dvm_tracker_destroy(trk);
  dvm_orb_destroy(h);
  return R;
}

int main(int argc, char** argv) {
  int pool_batch = 0;
  if (argc > 1 && std::strncmp(argv[1], "--pool=", 7) == 0) { pool_batch = std::atoi(argv[1] + 7); argv++; argc--; }
  if (argc < 5) { std::fprintf(stderr, "usage: online_agents [--pool=<max_batch>] <inputs.bin> <device> <frames_per_agent> K [K ...]\n"); return 2; }
  Inputs in;
  if (!load(argv[1], in)) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  const int device = std::atoi(argv[2]), nframes = std::atoi(argv[3]);
  dvm_orb_pool* pool = nullptr;
  dvm_pose_pool* ppool = nullptr;
  dvm_match_pool* mpool = nullptr;
  if (pool_batch > 0) {
    dvm_orb_params P{1000, 1.2f, 8, 20, 7};
    if (dvm_orb_pool_create(&P, device, pool_batch, -1, &pool) != 0 || dvm_pose_pool_create(device, pool_batch, -1, &ppool) != 0 ||
        dvm_match_pool_create(device, pool_batch, 2048, 2048, -1, &mpool) != 0) {
      std::fprintf(stderr, "pool: %s\n", dvm_last_error());
      return 1;
    }
  }
  std::string out = "{\"unit\": \"frames/s over all agents (extract + SearchByProjection + PoseOptimization per frame, host arrays in -> out, one C++ thread per agent)\"";
  bool same = true;
  uint64_t ref = 0;
  uint64_t ref_chain = 0;
  for (int mode = 0; mode < 3; mode++) {
    if (mode == 1 && !pool) continue;
    // pooled: all three calls through the shared services; one_chain: dvmh_track_with_motion_model (one enqueue + one synchronisation per frame)
    out += mode == 0 ? ", \"by_agents\": {" : mode == 1 ? ", \"by_agents_pooled\": {" : ", \"by_agents_one_chain\": {";
    for (int a = 4; a < argc; a++) {
      const int K = std::atoi(argv[a]);
      dvmh_set_match_pool(mode == 1 ? mpool : nullptr);   // the search of SearchByProjection(Cur, Last) through the shared service (pooled mode)
      const Result R = measure(in, device, nframes, K, mode == 1 ? pool : nullptr, mode == 1 ? ppool : nullptr, mode == 2);
      if (R.rc) return 1;
      if (mode == 0 && a == 4) ref = R.sum0;
      if (mode == 2 && a == 4) ref_chain = R.sum0;
      same = same && R.same && R.sum0 == (mode == 2 ? ref_chain : ref);
      char buf[640];
      std::snprintf(buf, sizeof(buf), "%s\"%d\": {\"value\": %.1f, \"ms_per_frame_per_agent\": %.4f, \"agent0_ms_in_extract_search_pose\": {\"mean\": [%.4f, %.4f, %.4f], \"median\": [%.4f, %.4f, %.4f]}%s",
                    a == 4 ? "" : ", ", K, R.fps, R.ms_per_frame, R.mean[0], R.mean[1], R.mean[2], R.med[0], R.med[1], R.med[2], mode == 1 ? "" : "}");
      out += buf;
      if (mode == 1) { std::snprintf(buf, sizeof(buf), ", \"mean_frames_per_batch\": %.2f}", R.mean_batch); out += buf; }
    }
    out += "}";
  }
  {   // the same K agents, their frames of a tick through ONE batched chain (one host thread)
    out += ", \"by_agents_batched_chain\": {";
    for (int a = 4; a < argc; a++) {
      const int K = std::atoi(argv[a]);
      uint64_t s0 = 0;
      const Result R = measure_batched(in, device, std::max(nframes / 2, 20), K, &s0);
      if (R.rc) return 1;
      same = same && s0 == ref_chain;      // agent 0 sees the frames the one-chain mode's agents see: the same results
      char buf[256];
      std::snprintf(buf, sizeof(buf), "%s\"%d\": {\"value\": %.1f, \"ms_per_tick\": %.4f}", a == 4 ? "" : ", ", K, R.fps, R.ms_per_frame);
      out += buf;
    }
    out += "}";
  }
  if (pool) dvm_orb_pool_destroy(pool);
  if (ppool) dvm_pose_pool_destroy(ppool);
  dvmh_set_match_pool(nullptr);
  if (mpool) dvm_match_pool_destroy(mpool);
  out += std::string(", \"identical_results_across_agents\": ") + (same ? "true" : "false") + "}";
  std::puts(out.c_str());
  return same ? 0 : 1;
}
