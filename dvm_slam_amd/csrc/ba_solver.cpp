// dvm_slam_amd/csrc/ba_solver.cpp -- host driver of the MI355X bundle adjustment (dvm_ba_* C ABI).
//
// Mirrors Optimizer::BundleAdjustment / LocalBundleAdjustment (reference src/Optimizer.cc:55-356,
// 1030-1387) at the level of "build graph -> optimizer.optimize(n) -> read estimates / chi2 back",
// with g2o's generic hyper-graph replaced by flat SoA arrays in HBM.  The Levenberg-Marquardt control
// flow (reference Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:59-165 and
// sparse_optimizer.cpp:349-412) runs here on the host, one scalar read-back per trial step; every
// numerical step is a HIP kernel (ba_kernels.hip).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/dvmslam_hip.h"
#include "ba_kernels.h"
#include "f64_spec.h"
#include "ba_ordering.h"
#include "host_stage.h"
#include "orb_pipeline.h"  // set_error / hip_check / DVM_HIP

using namespace dvm;

struct dvm_ba {
  int device = 0;
  hipStream_t stream = nullptr;
  BaView V{};
  // Phase results travel to the host through page-locked memory mapped into the device (BaPublish): h_vals[0..7] scalars
  // (chi2, trial chi2, scale, max diagonal, ..., [6] = Cholesky failure flag), h_seq the sequence number the host spins on.
  double* h_vals = nullptr;                 // hipHostMalloc'ed [16]; [8] holds the sequence number
  double* d_vals = nullptr;                 // the same memory as the device sees it
  unsigned long long seq = 0;
  int solve_seq = 0;
  bool speculate = true;     // device-side accept / reject + the next trial's Schur complement enqueued ahead (DVM_BA_NO_SPECULATION=1: off)
  double* d_spec = nullptr;  // device [2]: BaPublish::spec
  bool fuse_levels = true;   // k_chol_trsm_update (solve + update of a level in one launch) until one of its waits times out
  // landmark-sharded mode (dvm_ba_set_problem_sharded): rank r of `world` owns the landmarks l with l % world == r
  int rank = 0, world = 1;
  // second set of linearisation buffers: a trial evaluates its state WITH Jacobians and accumulates Hpp / Hll into these, so
  // that an accepted trial's state is already linearised when the next iteration starts (swapped in with the state)
  double *alt_lin = nullptr, *alt_linA = nullptr, *alt_W = nullptr, *alt_Hpp = nullptr, *alt_bp = nullptr, *alt_Hll = nullptr, *alt_bl = nullptr;
  bool sharded_api = false;   // problem set through dvm_ba_set_problem_sharded: with a collective registered, even a single rank runs the sharded flow
  // optional HIP-event timing of the phases of a trial (dvm_ba_profile): [0] linearise, [1] Schur complement, [2] tile Cholesky +
  // back substitution, [3] landmarks + update + chi2; milliseconds accumulated over prof_trials trials / prof_iters iterations
  bool prof = false;
  hipEvent_t pev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  double prof_ms[4] = {0, 0, 0, 0};
  int prof_trials = 0, prof_iters = 0;
  dvm_allreduce_fn allreduce = nullptr;
  void* allreduce_ctx = nullptr;
  double* ar_buf = nullptr;      // caller's device buffer the collectives run on
  int64_t ar_cap = 0;
  unsigned int* d_counter = nullptr;        // arrival counters of the in-kernel reductions [4]
  double* d_dev_vals = nullptr;             // device copy of the phase results [8]
  int* d_fail = nullptr;
  uint8_t* d_depth = nullptr;
  uint8_t* d_flags = nullptr;               // per-edge level / robust-kernel flags (dvm_ba_set_edge_flags); V.e_flags points here once set
  uint8_t* h_flags = nullptr;               // ONE page-locked staging slot of E bytes for them, reserved with the problem and reused by every call
  bool have_problem = false;
  double ms_structure = 0;
  // Window mode (csrc/ba_window.hip): a problem with a handful of free cameras -- the two-keyframe global BA of a monocular
  // initialisation, the first local windows -- is solved by the sequential-order kernel, whose result is bit-identical to g2o's
  // summation order: with one or two free cameras the gauge is held by the damping alone and any other order lands 1e-3 and more
  // away.  The state then lives in these host copies between calls; the tile solver's device arrays are
  // brought up to date only when a call needs them (edge flags set: the welding BA's second round).
  bool win_mode = false, win_device_stale = false;
  std::vector<double> ws_poses, ws_points, ws_chi2;
  std::vector<uint8_t> ws_fixed, ws_depth;
  std::vector<dvm_ba_edge> ws_edges;
  dvm_ba_camera ws_cam{};
  BaTileSchedule sched;                               // level schedule of the tile Cholesky (host copy: launch sizes)
  double tile_fill = 1.0;                             // non-zero tiles / all lower tiles of the factor

  // Device memory of a problem comes from an arena the handle keeps across dvm_ba_set_problem calls: chunks are carved by a
  // bump pointer and only RESET when the problem is replaced.  A fresh hipMalloc of this size class is mapped lazily -- the
  // first optimize() after set_problem paid 18 ms of first-touch at 2 000 keyframes (0.5 GB of linearisation buffers) on
  // top of 9 ms of iterations, every time, because allocations that large are not recycled by the runtime.
  struct Chunk { uint8_t* base; size_t cap, used; };
  std::vector<Chunk> chunks;
  template <typename T>
  int dalloc(T** p, size_t n) {
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
    for (Chunk& c : chunks)
      if (c.cap - c.used >= bytes) { *p = reinterpret_cast<T*>(c.base + c.used); c.used += bytes; return DVM_OK; }
    size_t total = 0;
    for (const Chunk& c : chunks) total += c.cap;
    const size_t cap = std::max<size_t>({bytes, total, (size_t)8 << 20});
    void* q = nullptr;
    int rc = hip_check(hipMalloc(&q, cap), "hipMalloc(ba arena)");
    if (rc != DVM_OK) {   // memory pressure: fall back to an exact-size chunk
      rc = hip_check(hipMalloc(&q, bytes), "hipMalloc(ba)");
      if (rc != DVM_OK) return rc;
      chunks.push_back({static_cast<uint8_t*>(q), bytes, bytes});
    } else {
      chunks.push_back({static_cast<uint8_t*>(q), cap, bytes});
    }
    *p = reinterpret_cast<T*>(q);
    return DVM_OK;
  }
  // Uploads go through page-locked staging chunks (kept like the device arena) as asynchronous copies on the handle's
  // stream: a synchronous hipMemcpy from pageable memory pins, copies and unpins per call, and the runtime was still busy
  // with that when the first kernels of the following optimize() were launched (8-15 ms of launch delay at 2 000 keyframes).
  struct HostChunk { uint8_t* base; size_t cap, used; };
  std::vector<HostChunk> stage;
  uint8_t* stage_alloc(size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    for (HostChunk& c : stage)
      if (c.cap - c.used >= bytes) { uint8_t* p = c.base + c.used; c.used += bytes; return p; }
    size_t total = 0;
    for (const HostChunk& c : stage) total += c.cap;
    const size_t cap = std::max<size_t>({bytes, total, (size_t)4 << 20});
    void* q = nullptr;
    if (hipHostMalloc(&q, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    stage.push_back({static_cast<uint8_t*>(q), cap, bytes});
    return static_cast<uint8_t*>(q);
  }
  // The copies are queued and issued by flush_copies(): a problem set-up is two dozen small arrays, and neighbours in the device
  // arena are neighbours in the staging chunk too (both sides advance in 256-byte steps), so runs of them leave as ONE
  // hipMemcpyAsync -- each call costs the host ~5 us, a third of a local-BA window's whole set-up.
  struct Pending { uint8_t* dst; const uint8_t* src; size_t bytes; };
  std::vector<Pending> pending;
  int copy_in(void* dst, const void* src, size_t bytes) {
    if (!bytes) return DVM_OK;
    uint8_t* st = stage_alloc((bytes + 255) & ~(size_t)255);
    if (!st) {                                                            // no pinned memory left: plain copy, in order
      const int rc = flush_copies();
      return rc != DVM_OK ? rc : hip_check(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), "upload");
    }
    std::memcpy(st, src, bytes);
    if (bytes > ((size_t)512 << 10)) {       // a large array goes out at once: its DMA runs under the staging of the next one
      const int rc = flush_copies();
      return rc != DVM_OK ? rc : hip_check(hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, stream), "upload");
    }
    if (!pending.empty()) {
      Pending& l = pending.back();
      const size_t lp = (l.bytes + 255) & ~(size_t)255;
      if (l.dst + lp == static_cast<uint8_t*>(dst) && l.src + lp == st) { l.bytes = lp + bytes; return queued(bytes); }   // (the slack of the previous array travels along)
    }
    pending.push_back({static_cast<uint8_t*>(dst), st, bytes});
    return queued(bytes);
  }
  // a large problem's arrays are megabytes: they go out while the host is still staging the next ones
  size_t pending_bytes = 0;
  int queued(size_t bytes) {
    pending_bytes += bytes;
    return pending_bytes > ((size_t)2 << 20) ? flush_copies() : DVM_OK;
  }
  int flush_copies() {
    int rc = DVM_OK;
    for (const Pending& c : pending)
      if (rc == DVM_OK) rc = hip_check(hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyHostToDevice, stream), "upload");
    pending.clear();
    pending_bytes = 0;
    return rc;
  }
  template <typename T>
  int upload(const T** dst, const std::vector<T>& v) {
    T* p = nullptr;
    int rc = dalloc(&p, v.size());
    if (rc == DVM_OK) rc = copy_in(p, v.data(), v.size() * sizeof(T));
    *dst = p;
    return rc;
  }
  void free_problem() {           // the arenas stay: the next problem reuses them (callers have synchronised the stream)
    for (Chunk& c : chunks) c.used = 0;
    for (HostChunk& c : stage) c.used = 0;
    pending.clear();
    pending_bytes = 0;
    have_problem = false;
  }
  void release_arena() {
    for (Chunk& c : chunks) hipFree(c.base);
    chunks.clear();
    for (HostChunk& c : stage) hipHostFree(c.base);
    stage.clear();
  }
};

extern "C" {

int dvm_ba_create(int device, dvm_ba** out) {
  if (!out) return DVM_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  if (device < 0 || device >= n) { set_error("device index out of range"); return DVM_ERR_INVALID; }
  DVM_HIP(hipSetDevice(device));
  dvm_ba* h = new dvm_ba;
  h->device = device;
  int rc = hip_check(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking), "stream");
  if (rc == DVM_OK) rc = hip_check(hipMalloc(&h->d_fail, sizeof(int)), "malloc");
  if (rc == DVM_OK) rc = hip_check(hipMalloc(&h->d_counter, 4 * sizeof(unsigned int)), "malloc");
  if (rc == DVM_OK) rc = hip_check(hipMemset(h->d_counter, 0, 4 * sizeof(unsigned int)), "memset");
  if (rc == DVM_OK) rc = hip_check(hipMalloc(&h->d_dev_vals, 8 * sizeof(double)), "malloc");
  if (rc == DVM_OK) rc = hip_check(hipMemset(h->d_dev_vals, 0, 8 * sizeof(double)), "memset");
  if (rc == DVM_OK) rc = hip_check(hipMalloc(&h->d_spec, 2 * sizeof(double)), "malloc");
  if (rc == DVM_OK) rc = hip_check(hipMemset(h->d_spec, 0, 2 * sizeof(double)), "memset");
  h->speculate = std::getenv("DVM_BA_NO_SPECULATION") == nullptr;
  if (rc == DVM_OK) rc = hip_check(hipHostMalloc(&h->h_vals, 16 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent), "hostmalloc");
  if (rc == DVM_OK) { std::memset(h->h_vals, 0, 16 * sizeof(double)); rc = hip_check(hipHostGetDevicePointer((void**)&h->d_vals, h->h_vals, 0), "devptr"); }
  if (rc != DVM_OK) { dvm_ba_destroy(h); return rc; }
  *out = h;
  return DVM_OK;
}

void dvm_ba_destroy(dvm_ba* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  h->free_problem();
  h->release_arena();
  if (h->d_fail) hipFree(h->d_fail);
  for (auto& e : h->pev) if (e) hipEventDestroy(e);
  if (h->d_counter) hipFree(h->d_counter);
  if (h->d_dev_vals) hipFree(h->d_dev_vals);
  if (h->d_spec) hipFree(h->d_spec);
  if (h->h_vals) hipHostFree(h->h_vals);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

// Graph construction ("buildStructure", block_solver.hpp:143-295): vertex ordering, CSR incidence
// lists and the block pattern of the reduced camera matrix.  Done once per problem on the host.
// the default choice between the level launches and the flow form of the reduced solve (k_chol_flow); DVM_BA_FLOW overrides it
// Measured: the flow form costs ~15.5 us per level of a chain (the factorisation's own 7.7 us + the parent strip and
// its product in the same workgroup) and ~18 us per level with two children; the level launches ~17 us per level when a level is a handful of
// tiles and ~20 us when its trailing update is large, and they own the small problems (the top-pair kernel solves a local-BA window's whole
// reduced system in one workgroup).  From 5 levels on the flow form wins -- 100 keyframes 6 927 -> 7 114 it/s, 200: 5 536 -> 5 739, the
// 500-keyframe ring 4 036 -> 4 092, the same map with loop closures (13 levels after the kept landmarks) 2 404 -> 2 509 -- as long as its
// tasks are a few per workgroup: a task holds its workgroup while it waits, and at 1 000 keyframes with loop closures (1 848 tasks on 256
// workgroups) the level launches are ahead again (1 401 vs 1 313 it/s).
static bool kFlowDefault(const BaTileSchedule& SC, int workgroups) { return SC.nlevels >= 5 && (int)(SC.flow_tasks.size() / 8) <= 4 * workgroups; }
static int set_problem_impl(dvm_ba* h, const double* poses, const uint8_t* fixed, int P, const double* points, int L,
                            const dvm_ba_edge* edges, int E, const dvm_ba_camera* cam, int rank, int world) {
  if (!h || !poses || !fixed || !points || !edges || !cam || P < 1 || L < 1 || E < 1 || world < 1 || rank < 0 || rank >= world) {
    set_error("dvm_ba_set_problem: bad arguments");
    return DVM_ERR_INVALID;
  }
  h->rank = rank; h->world = world;
  DVM_HIP(hipSetDevice(h->device));
  DVM_HIP(hipStreamSynchronize(h->stream));
  h->free_problem();
  const auto t0 = std::chrono::steady_clock::now();
  const bool dbg_time = std::getenv("DVM_BA_DEBUG_SCHEDULE") != nullptr;
  auto mark = [&](const char* what) {
    if (dbg_time) std::fprintf(stderr, "set_problem: %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  BaView& V = h->V;
  std::memset(&V, 0, sizeof(V));
  V.P = P; V.L = L; V.E = E;
  V.fx = cam->fx; V.fy = cam->fy; V.cx = cam->cx; V.cy = cam->cy; V.delta = cam->huber_delta;
  std::vector<int32_t> e_pose(E), e_point(E);
  std::vector<double> e_obs(2 * (size_t)E), e_info(E);
  std::vector<int32_t> pt_cnt(L + 1, 0), ps_cnt(P + 1, 0);
  for (int k = 0; k < E; k++) {
    if (edges[k].pose < 0 || edges[k].pose >= P || edges[k].point < 0 || edges[k].point >= L) { set_error("edge index out of range"); return DVM_ERR_INVALID; }
    e_pose[k] = edges[k].pose; e_point[k] = edges[k].point;
    e_obs[2 * (size_t)k] = edges[k].u; e_obs[2 * (size_t)k + 1] = edges[k].v; e_info[k] = edges[k].inv_sigma2;
    pt_cnt[e_point[k] + 1]++; ps_cnt[e_pose[k] + 1]++;
  }
  std::vector<int32_t> pt_start(L + 1, 0), ps_start(P + 1, 0);
  for (int l = 0; l < L; l++) pt_start[l + 1] = pt_start[l] + pt_cnt[l + 1];
  for (int p = 0; p < P; p++) ps_start[p + 1] = ps_start[p] + ps_cnt[p + 1];
  std::vector<int32_t> pt_edges(E), ps_edges(E), pt_fill(pt_start.begin(), pt_start.end() - 1), ps_fill(ps_start.begin(), ps_start.end() - 1);
  for (int k = 0; k < E; k++) { pt_edges[pt_fill[e_point[k]]++] = k; ps_edges[ps_fill[e_pose[k]]++] = k; }
  // Free cameras.  g2o orders them by vertex id (sparse_optimizer.cpp:161-185) and leaves the fill-reducing
  // permutation to the sparse Cholesky; here the permutation is applied up front (ba_ordering.h): banding + nested
  // dissection at tile granularity, so that the dense-tile factorisation has a short dependency chain.
  std::vector<int32_t> nat_of(P, -1), nat_pose;
  for (int p = 0; p < P; p++)
    if (!fixed[p] && ps_cnt[p + 1] > 0) { nat_of[p] = (int32_t)nat_pose.size(); nat_pose.push_back(p); }
  V.nfree = (int)nat_pose.size();
  // ---- which camera pairs share a landmark.  One table over (camera, camera) in natural free-camera indices serves the
  // ordering (its adjacency lists, already free of duplicates) and, after the ordering, the block list of the reduced matrix.
  // (A std::map of vectors keyed by the block did this before: 29 of the 50 ms this function took at 500 keyframes, next to
  // 16 ms of ordering on adjacency lists with a million duplicate entries.)
  const int nf = V.nfree;
  std::vector<int32_t> dense_id;
  std::unordered_map<uint64_t, int32_t> sparse_id;
  const bool dense = (size_t)nf * nf <= ((size_t)1 << 24);
  std::vector<std::pair<int32_t, int32_t>> pair_cams;          // id -> (hi, lo) natural indices, in first-seen order
  auto pair_slot = [&](int hi, int lo) -> int32_t& {
    if (dense) return dense_id[(size_t)hi * nf + lo];
    auto it = sparse_id.find(((uint64_t)hi << 32) | (uint32_t)lo);
    if (it == sparse_id.end()) it = sparse_id.emplace(((uint64_t)hi << 32) | (uint32_t)lo, -1).first;
    return it->second;
  };
  auto touch = [&](int x, int y) -> int32_t {
    int32_t& id = pair_slot(std::max(x, y), std::min(x, y));
    if (id < 0) { id = (int32_t)pair_cams.size(); pair_cams.push_back({std::max(x, y), std::min(x, y)}); }
    return id;
  };
  std::vector<int32_t> e_cam(E);
  for (int k = 0; k < E; k++) e_cam[k] = nat_of[e_pose[k]];
  // KEPT LANDMARKS (ba_kernels.h, BaView::kept_*): when the elimination tree of the tile factorisation comes out deep (>= 12 levels: a
  // chain, not a bush) and a few landmarks are to blame -- landmarks that couple camera pairs hardly anything else couples (<= 2
  // co-observed landmarks) --, those landmarks (at most 210: ten tiles) are left out of the Schur complement and stay unknowns of the
  // reduced system; the structure is then built again without their couplings.  DVM_BA_BORDER=0 switches it off (A/B, tests).
  constexpr size_t kMaxKept = 210;                               // ten tiles of 21
  std::vector<int32_t> kept_slot(L, -1), kept_list;
  // (the landmark-chunk form of the Schur complement, DVM_BA_SCHUR_LM, has no workgroups for the kept landmarks' rows: the two do not combine)
  const bool try_border = world == 1 && nf > 60 && !std::getenv("DVM_BA_SCHUR_LM") &&
                          !(std::getenv("DVM_BA_BORDER") && std::atoi(std::getenv("DVM_BA_BORDER")) == 0);
  std::vector<int> cam_pos;
  int levels_all = 0;
  for (int pass = 0;; pass++) {
    if (dense) dense_id.assign((size_t)nf * nf, -1); else sparse_id.clear();
    pair_cams.clear();
    std::vector<int32_t> pair_cnt;                               // per pair: landmarks that see both cameras
    for (int i = 0; i < nf; i++) touch(i, i);
    std::vector<int32_t> lc;                                     // the free cameras of one landmark, in edge order
    for (int l = 0; l < L; l++) {
      if (kept_slot[l] >= 0) continue;
      lc.clear();
      for (int a = pt_start[l]; a < pt_start[l + 1]; a++) { const int ca = e_cam[pt_edges[a]]; if (ca >= 0) lc.push_back(ca); }
      for (size_t a = 1; a < lc.size(); a++)
        for (size_t b = 0; b < a; b++) {
          const int32_t id = touch(lc[a], lc[b]);
          if (pass == 0 && try_border) { if ((size_t)id >= pair_cnt.size()) pair_cnt.resize(pair_cams.size(), 0); pair_cnt[id]++; }
        }
    }
    std::vector<std::vector<int>> cam_adj(nf);
    for (const auto& pc : pair_cams)
      if (pc.first != pc.second) { cam_adj[pc.first].push_back(pc.second); cam_adj[pc.second].push_back(pc.first); }
    if (pass == 0) mark("incidence + adjacency");
    cam_pos = ba_order_cameras(cam_adj);
    if (pass == 0) mark("camera order");
    if (!try_border || pass == 2) break;
    // how deep does the tree of this structure come out?
    const int nct = (nf + kCamsPerTile - 1) / kCamsPerTile, nbt0 = ((int)kept_list.size() + 20) / 21;
    std::vector<std::vector<char>> T0(nct + nbt0 + 1, std::vector<char>(nct + nbt0 + 1, 0));
    for (const auto& pc : pair_cams) {
      const int t1 = cam_pos[pc.first] / kCamsPerTile, t2 = cam_pos[pc.second] / kCamsPerTile;
      T0[std::max(t1, t2)][std::min(t1, t2)] = 1;
    }
    for (size_t j = 0; j < kept_list.size(); j++) {
      const int l = kept_list[j], tb = nct + (int)j / 21;
      T0[tb][tb] = 1;
      for (int a = pt_start[l]; a < pt_start[l + 1]; a++) if (e_cam[pt_edges[a]] >= 0) T0[tb][cam_pos[e_cam[pt_edges[a]]] / kCamsPerTile] = 1;
    }
    const int levels = ba_tile_schedule(T0).nlevels;
    if (pass == 1) {
      if (4 * levels <= 3 * levels_all) break;                   // the tree is at least a quarter shorter: keep them
      std::fill(kept_slot.begin(), kept_slot.end(), -1); kept_list.clear();      // no gain: the plain structure, once more
      pass = 1;                                                   // (-> pass 2: build and leave)
      continue;
    }
    levels_all = levels;
    if (levels < 12) break;
    // the landmarks to blame, by the number of weak camera pairs each one is the (near) only reason for
    std::vector<std::pair<int32_t, int32_t>> cand;               // (score, landmark)
    for (int l = 0; l < L; l++) {
      int score = 0;
      bool dup = false;
      for (int a = pt_start[l]; a < pt_start[l + 1] && !dup; a++) {
        const int ca = e_cam[pt_edges[a]];
        if (ca < 0) continue;
        for (int b = pt_start[l]; b < a; b++) {
          const int cb = e_cam[pt_edges[b]];
          if (cb < 0) continue;
          if (cb == ca) { dup = true; break; }                  // two observations from one camera: its border block would be written twice
          if (pair_cnt[pair_slot(std::max(ca, cb), std::min(ca, cb))] <= 2) score++;
        }
      }
      if (score > 0 && !dup) cand.push_back({score, l});
    }
    if (cand.empty()) break;
    std::sort(cand.begin(), cand.end(), [](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) { return x.first != y.first ? x.first > y.first : x.second < y.second; });
    if (cand.size() > kMaxKept) cand.resize(kMaxKept);
    std::sort(cand.begin(), cand.end(), [](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) { return x.second < y.second; });
    for (const auto& c : cand) { kept_slot[c.second] = (int32_t)kept_list.size(); kept_list.push_back(c.second); }
    mark("kept landmarks chosen");
  }
  std::vector<int32_t> pidx(P, -1), free_pose(nf);
  for (int a = 0; a < nf; a++) { pidx[nat_pose[a]] = cam_pos[a]; free_pose[cam_pos[a]] = nat_pose[a]; }
  // non-zero lower blocks (i1 >= i2) in ascending (i1, i2) order -- k_schur deals contiguous runs of this list to the XCDs
  const int nblk = (int)pair_cams.size();
  std::vector<int32_t> blk_i1(nblk), blk_i2(nblk), order(nblk), rank_of(nblk);
  for (int u = 0; u < nblk; u++) {
    const int p1 = cam_pos[pair_cams[u].first], p2 = cam_pos[pair_cams[u].second];
    blk_i1[u] = std::max(p1, p2); blk_i2[u] = std::min(p1, p2);
    order[u] = u;
  }
  std::sort(order.begin(), order.end(), [&](int x, int y) { return blk_i1[x] != blk_i1[y] ? blk_i1[x] < blk_i1[y] : blk_i2[x] < blk_i2[y]; });
  {
    std::vector<int32_t> s1(nblk), s2(nblk);
    for (int r = 0; r < nblk; r++) { rank_of[order[r]] = r; s1[r] = blk_i1[order[r]]; s2[r] = blk_i2[order[r]]; }
    blk_i1.swap(s1); blk_i2.swap(s2);
  }
  // ---- landmark sharding: the STRUCTURE above (free cameras, ordering, block pattern, and below the tile schedule) comes from
  // all edges and is identical on every rank; the edge arrays, the incidence lists and the (edge, edge) pairs keep only the
  // edges of the landmarks this rank owns.  A landmark owned elsewhere has no local edge: for the kernels it is "not a vertex".
  std::vector<int32_t> loc_of;
  int El = E;
  if (world > 1) {
    loc_of.assign(E, -1);
    El = 0;
    for (int k = 0; k < E; k++) if (e_point[k] % world == rank) loc_of[k] = El++;
    if (El == 0) { set_error("dvm_ba_set_problem_sharded: this rank owns no observed landmark"); return DVM_ERR_INVALID; }
  }
  // the (edge of i1, edge of i2) pairs of every block, in landmark order: count, then fill (the order inside a block is the
  // one a map of vectors filled by the same loops had)
  std::vector<int32_t> blk_start(nblk + 1, 0);
  struct PairRec { int32_t blk, k1, k2; };
  std::vector<PairRec> recs;                       // one table lookup per pair: the count pass records what the fill pass needs
  {
    size_t npairs = 0;                               // (an exact reserve: the vector used to grow -- and copy itself -- once or twice on the way)
    for (int l = 0; l < L; l++) { const size_t d = (size_t)(pt_start[l + 1] - pt_start[l]); npairs += d * (d + 1) / 2; }
    recs.reserve(npairs);
  }
  struct LmEdge { int32_t k, c, pos; };
  std::vector<LmEdge> le;                          // the free-camera edges of one landmark (edge, natural camera, position), in edge order
  for (int l = 0; l < L; l++) {
    if (world > 1 && l % world != rank) continue;
    if (kept_slot[l] >= 0) continue;                 // not eliminated: no term in the Schur complement
    le.clear();
    for (int a = pt_start[l]; a < pt_start[l + 1]; a++) {
      const int k = pt_edges[a], c = e_cam[k];
      if (c >= 0) le.push_back({k, c, cam_pos[c]});
    }
    for (const LmEdge& x : le)
      for (const LmEdge& y : le) {
        if (y.pos > x.pos) continue;
        const int32_t blk = rank_of[pair_slot(std::max(x.c, y.c), std::min(x.c, y.c))];
        blk_start[blk + 1]++;
        recs.push_back({blk, x.k, y.k});
      }
  }
  for (int r = 0; r < nblk; r++) blk_start[r + 1] += blk_start[r];
  std::vector<int32_t> pair_k1(recs.size()), pair_k2(recs.size()), fill(blk_start.begin(), blk_start.end() - 1);
  for (const PairRec& r : recs) {
    const int t = fill[r.blk]++;
    pair_k1[t] = world > 1 ? loc_of[r.k1] : r.k1;
    pair_k2[t] = world > 1 ? loc_of[r.k2] : r.k2;
  }
  mark("block pairs");
  if (world > 1) {
    std::vector<int32_t> ep(El), el(El);
    std::vector<double> eo(2 * (size_t)El), ei(El);
    for (int k = 0; k < E; k++) {
      const int j = loc_of[k];
      if (j < 0) continue;
      ep[j] = e_pose[k]; el[j] = e_point[k]; eo[2 * (size_t)j] = e_obs[2 * (size_t)k]; eo[2 * (size_t)j + 1] = e_obs[2 * (size_t)k + 1]; ei[j] = e_info[k];
    }
    e_pose.swap(ep); e_point.swap(el); e_obs.swap(eo); e_info.swap(ei);
    std::fill(pt_start.begin(), pt_start.end(), 0); std::fill(ps_start.begin(), ps_start.end(), 0);
    for (int j = 0; j < El; j++) { pt_start[e_point[j] + 1]++; ps_start[e_pose[j] + 1]++; }
    for (int l = 0; l < L; l++) pt_start[l + 1] += pt_start[l];
    for (int p = 0; p < P; p++) ps_start[p + 1] += ps_start[p];
    pt_edges.assign(El, 0); ps_edges.assign(El, 0);
    std::vector<int32_t> pf(pt_start.begin(), pt_start.end() - 1), sf(ps_start.begin(), ps_start.end() - 1);
    for (int j = 0; j < El; j++) { pt_edges[pf[e_point[j]]++] = j; ps_edges[sf[e_pose[j]]++] = j; }
    E = El;
    V.E = El;
  }
  V.nblk = (int)blk_i1.size();
  const int n = 6 * V.nfree;
  // ---- landmark-chunk form of the Schur complement.  The per-block form gathers W1 | W2 | Dinv per (edge, edge) pair: 360 B for a pair,
  // 260 MB for the 720 k pairs of the 500-keyframe problem, of which 77 MB come from HBM -- every W row is fetched once per pair it
  // is in.  Here a workgroup takes a chunk of landmarks, loads their W rows ONCE (23 MB in all) and forms every pair product of those
  // landmarks from LDS; what a chunk contributes to a block is one partial 6x6 block (a "run"), and a second small launch adds a
  // block's runs in chunk order (a fixed order: no atomics).  Landmarks are sorted by their first camera, so a chunk touches the few
  // blocks of neighbouring cameras and a block collects from few chunks.
  std::vector<int32_t> sl_desc, sl_row_edge, sl_row_lm, sl_lm, sl_pairs, sl_runs, bp_start(V.nblk + 1, 0), bp_slots;
  V.n_slc = V.n_slrun = 0;
  if (std::getenv("DVM_BA_SCHUR_LM")) {        // experiment switch: 54 + 7 us against the per-block form's 53 us on the 500-keyframe problem
    std::unordered_map<uint64_t, int32_t> blk_of;
    blk_of.reserve((size_t)V.nblk * 2);
    for (int b = 0; b < V.nblk; b++) blk_of[((uint64_t)blk_i1[b] << 32) | (uint32_t)blk_i2[b]] = b;
    std::vector<std::pair<int32_t, int32_t>> lms;       // (first camera position, landmark) of every landmark with a free-camera edge here
    for (int l = 0; l < L; l++) {
      int first = 1 << 30;
      for (int a = pt_start[l]; a < pt_start[l + 1]; a++) { const int fi = pidx[e_pose[pt_edges[a]]]; if (fi >= 0) first = std::min(first, fi); }
      if (first < (1 << 30)) lms.push_back({first, l});
    }
    std::sort(lms.begin(), lms.end());
    std::vector<std::pair<int32_t, int32_t>> run_of_blk;   // (block, global run index), for the per-block lists
    size_t i = 0;
    bool chunks_ok = true;
    while (i < lms.size() && chunks_ok) {
      const int32_t row_off = (int32_t)sl_row_edge.size(), lm_off = (int32_t)sl_lm.size(), pair_off = (int32_t)sl_pairs.size(), run_off = (int32_t)sl_runs.size() / 2;
      int nrows = 0, nlm = 0, nrun_bound = 0;
      std::vector<std::pair<int32_t, int32_t>> pr;      // (block, row1 | row2 << 9 | landmark slot << 18)
      std::vector<int32_t> seen_blk;
      while (i < lms.size() && nlm < kSchurLmLandmarks) {
        const int l = lms[i].second;
        int deg = 0;
        for (int a = pt_start[l]; a < pt_start[l + 1]; a++) deg += pidx[e_pose[pt_edges[a]]] >= 0;
        if (deg > kSchurLmRows || deg * (deg + 1) / 2 > std::min(kSchurLmPairs, kSchurLmRuns)) { chunks_ok = false; break; }   // a landmark seen by > 44 free cameras: the per-block form takes the problem
        // the chunk's limits: rows and pairs (LDS), and blocks -- a landmark of degree d can open d (d + 1) / 2 new runs at most
        if (nrows + deg > kSchurLmRows || (int)pr.size() + deg * (deg + 1) / 2 > kSchurLmPairs || nrun_bound + deg * (deg + 1) / 2 > kSchurLmRuns) { if (nlm > 0) break; }
        const int r0 = nrows;
        for (int a = pt_start[l]; a < pt_start[l + 1]; a++) {
          const int k = pt_edges[a];
          if (pidx[e_pose[k]] < 0) continue;
          sl_row_edge.push_back(k); sl_row_lm.push_back(nlm); nrows++;
        }
        // block_solver.hpp:381-439: edge k1 (outer) x edge k2 (inner) of the landmark, lower blocks only
        for (int q1 = r0; q1 < nrows; q1++)
          for (int q2 = r0; q2 < nrows; q2++) {
            const int i1 = pidx[e_pose[sl_row_edge[row_off + q1]]], i2 = pidx[e_pose[sl_row_edge[row_off + q2]]];
            if (i2 > i1) continue;
            const int32_t bid = blk_of[((uint64_t)i1 << 32) | (uint32_t)i2];
            pr.push_back({bid, q1 | (q2 << 9) | (nlm << 18)});
            seen_blk.push_back(bid);
          }
        std::sort(seen_blk.begin(), seen_blk.end());
        seen_blk.erase(std::unique(seen_blk.begin(), seen_blk.end()), seen_blk.end());
        nrun_bound = (int)seen_blk.size();
        sl_lm.push_back(l); nlm++; i++;
      }
      std::stable_sort(pr.begin(), pr.end(), [](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) { return x.first < y.first; });
      int nruns = 0;
      for (size_t j = 0; j < pr.size(); j++) {
        if (j == 0 || pr[j].first != pr[j - 1].first) {
          run_of_blk.push_back({pr[j].first, (int32_t)sl_runs.size() / 2 - (int32_t)(sl_desc.size() / 8)});   // global run index = position minus the sentinels before it
          sl_runs.push_back(pr[j].first); sl_runs.push_back((int32_t)j); nruns++;
        }
        sl_pairs.push_back(pr[j].second);
      }
      sl_runs.push_back(-1); sl_runs.push_back((int32_t)pr.size());      // sentinel: where the chunk's last run ends
      const int32_t d[8] = {row_off, nrows, lm_off, nlm, pair_off, (int32_t)pr.size(), run_off, nruns};
      sl_desc.insert(sl_desc.end(), d, d + 8);
    }
    if (!chunks_ok) { sl_desc.clear(); sl_row_edge.clear(); sl_row_lm.clear(); sl_lm.clear(); sl_pairs.clear(); sl_runs.clear(); run_of_blk.clear(); }
    V.n_slc = (int)sl_desc.size() / 8;
    V.n_slrun = (int)run_of_blk.size();
    for (const auto& rb : run_of_blk) bp_start[rb.first + 1]++;
    for (int b = 0; b < V.nblk; b++) bp_start[b + 1] += bp_start[b];
    bp_slots.resize(run_of_blk.size());
    std::vector<int32_t> fillp(bp_start.begin(), bp_start.end() - 1);
    for (const auto& rb : run_of_blk) bp_slots[fillp[rb.first]++] = rb.second;     // run_of_blk is in chunk order: so is every block's list
  }
  mark("landmark chunks");
  // ---- tile space: 10 whole cameras per 64-row tile, one extra tile for the augmented rhs row
  const int ncamt = (V.nfree + kCamsPerTile - 1) / kCamsPerTile, nkeptt = ((int)kept_list.size() + 20) / 21, nkb = ncamt + nkeptt + 1;
  V.ncamt = ncamt; V.nkept = (int)kept_list.size();
  V.n_pad = 64 * (ncamt + nkeptt);
  V.per_tile = kCamsPerTile; V.dof = 6;
  V.ldS = 64 * nkb;
  // ---- symbolic tile Cholesky + level schedule.  SLAM reduced camera matrices are banded along the trajectory plus
  // a few loop-closure blocks; skipping zero tiles keeps the dense-tile solver exact while doing only the work the
  // structure requires, and the nested-dissection order makes most tile columns independent of each other.
  std::vector<std::vector<char>> T(nkb, std::vector<char>(nkb, 0));
  for (size_t b = 0; b < blk_i1.size(); b++) {
    const int tr = blk_i1[b] / kCamsPerTile, tc = blk_i2[b] / kCamsPerTile;
    T[std::max(tr, tc)][std::min(tr, tc)] = 1;
  }
  for (size_t j = 0; j < kept_list.size(); j++) {     // the tiles of the kept landmarks: coupled to the tiles of the cameras that see them
    const int l = kept_list[j], tb = ncamt + (int)j / 21;
    T[tb][tb] = 1;
    for (int a = pt_start[l]; a < pt_start[l + 1]; a++) {
      const int c = e_cam[pt_edges[a]];
      if (c >= 0) T[tb][cam_pos[c] / kCamsPerTile] = 1;
    }
  }
  mark("block arrays");
  h->sched = ba_tile_schedule(T);
  mark("tile schedule");
  const BaTileSchedule& SC = h->sched;
  h->tile_fill = SC.fill;
  V.nlevels = SC.nlevels;
  V.pair_ok = SC.pair_a >= 0 && std::getenv("DVM_BA_NO_PAIR") == nullptr; V.pair_a = SC.pair_a; V.pair_b = SC.pair_b;   // (DVM_BA_NO_PAIR: A/B switch)
  V.diag_in_level = std::getenv("DVM_BA_NO_DIAG_IN_LEVEL") == nullptr;
  V.n_root_raw = std::getenv("DVM_BA_NO_ROOT_RAW") ? 0 : SC.n_root_raw;   // (ba_ordering.h: the last launched level's panel solve left to the back substitution; the switch: A/B and tests)

  int rc = DVM_OK;
  auto ok = [&](int r) { if (rc == DVM_OK) rc = r; };
  ok(h->dalloc(&V.poses, 7 * (size_t)P)); ok(h->dalloc(&V.points, 3 * (size_t)L));
  ok(h->dalloc(&V.poses_new, 7 * (size_t)P)); ok(h->dalloc(&V.points_new, 3 * (size_t)L));
  ok(h->upload(&V.pidx, pidx)); ok(h->upload(&V.free_pose, free_pose));
  ok(h->upload(&V.e_pose, e_pose)); ok(h->upload(&V.e_point, e_point));
  ok(h->upload(&V.e_obs, e_obs)); ok(h->upload(&V.e_info, e_info));
  ok(h->upload(&V.pt_start, pt_start)); ok(h->upload(&V.pt_edges, pt_edges));
  {
    std::vector<int32_t> pt_fi(pt_edges.size());
    for (size_t i = 0; i < pt_edges.size(); i++) pt_fi[i] = pidx[e_pose[pt_edges[i]]];
    ok(h->upload(&V.pt_fi, pt_fi));
  }
  ok(h->upload(&V.ps_start, ps_start)); ok(h->upload(&V.ps_edges, ps_edges));
  ok(h->upload(&V.sl_desc, sl_desc)); ok(h->upload(&V.sl_row_edge, sl_row_edge)); ok(h->upload(&V.sl_row_lm, sl_row_lm)); ok(h->upload(&V.sl_lm, sl_lm));
  ok(h->upload(&V.sl_pairs, sl_pairs)); ok(h->upload(&V.sl_runs, sl_runs)); ok(h->upload(&V.bp_start, bp_start)); ok(h->upload(&V.bp_slots, bp_slots));
  ok(h->dalloc(&V.sl_part, 36 * (size_t)std::max(V.n_slrun, 1)));
  ok(h->upload(&V.blk_i1, blk_i1)); ok(h->upload(&V.blk_i2, blk_i2)); ok(h->upload(&V.blk_start, blk_start));
  ok(h->upload(&V.pair_k1, pair_k1)); ok(h->upload(&V.pair_k2, pair_k2));
  V.schur_wide = (V.nblk > 0 && V.nblk <= 512 && pair_k1.size() / (size_t)V.nblk >= 192) ? 1 : 0;   // few blocks, long pair lists: see k_schur
  {
    std::vector<int32_t> pair_pt(pair_k1.size());
    for (size_t t = 0; t < pair_k1.size(); t++) pair_pt[t] = e_point[pair_k1[t]];
    ok(h->upload(&V.pair_pt, pair_pt));
  }
  ok(h->dalloc(&V.e_chi2, (size_t)E)); ok(h->dalloc(&V.e_lin, (size_t)E * kEdgeLinStride)); ok(h->dalloc(&V.e_linA, (size_t)E * kEdgeLinStride)); ok(h->dalloc(&V.e_W, (size_t)E * 18));
  ok(h->dalloc(&V.Hpp, 36 * (size_t)V.nfree)); ok(h->dalloc(&V.bp, (size_t)n));
  ok(h->dalloc(&V.Hll, 9 * (size_t)L)); ok(h->dalloc(&V.bl, 3 * (size_t)L));
  ok(h->dalloc(&V.Dinv, 9 * (size_t)L)); ok(h->dalloc(&V.db, 3 * (size_t)L));
  ok(h->dalloc(&h->alt_lin, (size_t)E * kEdgeLinStride)); ok(h->dalloc(&h->alt_linA, (size_t)E * kEdgeLinStride)); ok(h->dalloc(&h->alt_W, (size_t)E * 18));
  ok(h->dalloc(&h->alt_Hpp, 36 * (size_t)V.nfree)); ok(h->dalloc(&h->alt_bp, (size_t)n));
  ok(h->dalloc(&h->alt_Hll, 9 * (size_t)L)); ok(h->dalloc(&h->alt_bl, 3 * (size_t)L));
  ok(h->dalloc(&V.S, (size_t)V.ldS * V.ldS)); ok(h->dalloc(&V.Linv, (size_t)(V.ldS / 64) * 64 * 64)); ok(h->dalloc(&V.ytmp, (size_t)V.n_pad + 64 + 2 * (SC.strips.size() / 2) + 2));
  ok(h->dalloc(&V.xrow, (size_t)V.n_pad + 64));
  ok(h->dalloc(&V.x, (size_t)n + 3 * (size_t)L));
  ok(h->dalloc(&V.partial, (size_t)(E + 255) / 256)); ok(h->dalloc(&V.partial2, (size_t)(8 * (size_t)L + 255) / 256 + (size_t)(V.nfree + 255) / 256 + (size_t)(V.nfree + 3) / 4 + 2));   // block partials of k_point_backsub / k_max_diag
  ok(h->dalloc(&h->d_depth, (size_t)E));
  ok(h->dalloc(&h->d_flags, (size_t)E));
  h->h_flags = h->stage_alloc((size_t)std::max(E, 1));      // (nullptr when pinned memory is exhausted: set_edge_flags then copies synchronously)
  ok(h->upload(&V.nz_tiles, SC.nz_tiles)); V.n_nz = (int)(SC.nz_tiles.size() / 2);
  ok(h->upload(&V.cols, SC.cols)); ok(h->upload(&V.strips, SC.strips)); ok(h->upload(&V.targets, SC.targets));
  ok(h->upload(&V.contrib, SC.contrib)); ok(h->upload(&V.colstrip_off, SC.colstrip_off)); ok(h->upload(&V.colstrips, SC.colstrips));
  ok(h->upload(&V.contrib_strip, SC.contrib_strip));
  if (V.nkept) { ok(h->upload(&V.kept_slot, kept_slot)); ok(h->upload(&V.kept_list, kept_list)); }
  // the flow form of the solve (k_chol_flow): task list, per-tile level ranges, flags (zeroed once: they are compared with the solve's sequence number)
  ok(h->upload(&V.flow_tasks, SC.flow_tasks)); ok(h->upload(&V.flow_contrib, SC.flow_contrib)); ok(h->upload(&V.flow_col, SC.flow_col)); ok(h->upload(&V.colstrip_id, SC.colstrip_id));
  V.n_flow_tasks = (int)(SC.flow_tasks.size() / 8); V.n_strips_total = (int)(SC.strips.size() / 2); V.n_tiles_total = nkb;
  ok(h->dalloc(&V.flow_flags, 4 * (size_t)V.n_strips_total + 2 * (size_t)nkb + 4));
  {
    // which form: the level launches pay ~17 us a level when a level is a handful of tiles and ~43 us when its trailing update is large; the flow
    // form pays one launch and a ~19 us chain per level whatever the level's size.  DVM_BA_FLOW=0 / 1 forces the choice (A/B switch).
    const char* e = std::getenv("DVM_BA_FLOW");
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
    V.flow_wgs = cus;
    const bool fits = (size_t)V.ldS * V.ldS * sizeof(double) < (size_t)0x7FFFFFF0;      // 32-bit buffer offsets into S
    // (the chains -- one per leaf of the elimination tree -- wait for tasks the OTHER workgroups draw: they must stay a minority)
    // several roots (a disconnected reduced camera graph): one tree's back substitution could write x before a diagonal tile of another
    // tree fails, and g2o's linear solver leaves x untouched when the factorisation fails -- such graphs keep the level launches
    int roots = 0;
    for (size_t k = 0; k + 1 < SC.flow_col.size() / 8; k++) roots += SC.flow_col[8 * k] < 0;
    V.flow = (e ? std::atoi(e) != 0 : kFlowDefault(SC, cus)) && fits && 4 * SC.flow_leaves <= cus && roots <= 1 ? 1 : 0;
  }
  V.strip_flags = reinterpret_cast<int32_t*>(V.ytmp + (size_t)V.n_pad + 64);   // behind the back substitution's words, cleared with them
  V.h_level_off = SC.level_off.data(); V.h_strip_off = SC.strip_off.data(); V.h_tgt_off = SC.tgt_off.data();
  if (std::getenv("DVM_BA_DEBUG_SCHEDULE")) {
    for (size_t l = 0; l + 1 < SC.tgt_off.size(); l++) {
      int mx = 0; long sum = 0;
      for (int t = SC.tgt_off[l]; t < SC.tgt_off[l + 1]; t++) { const int n = SC.targets[4 * t + 3] - SC.targets[4 * t + 2]; mx = std::max(mx, n); sum += n; }
      std::fprintf(stderr, "level %zu: cols %d strips %d targets %d contributors sum %ld max %d\n", l, SC.level_off[l + 1] - SC.level_off[l],
                   SC.strip_off[l + 1] - SC.strip_off[l], SC.tgt_off[l + 1] - SC.tgt_off[l], sum, mx);
    }
  }
  if (rc != DVM_OK) { h->free_problem(); return rc; }
  mark("allocations + uploads");
  // poses: normalise quaternions like SE3Quat's constructor (se3quat.h:261-266)
  std::vector<double> pn(poses, poses + 7 * (size_t)P);
  for (int p = 0; p < P; p++) {
    double* q = &pn[7 * (size_t)p + 3];
    if (q[3] < 0) for (int i = 0; i < 4; i++) q[i] = -q[i];
    const double nn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= nn;
  }
  ok(h->copy_in(V.poses, pn.data(), pn.size() * sizeof(double)));
  ok(h->copy_in(V.points, points, 3 * (size_t)L * sizeof(double)));
  // the trial buffers start as copies: fixed cameras and unobserved landmarks are never rewritten
  ok(h->copy_in(V.poses_new, pn.data(), pn.size() * sizeof(double)));
  ok(h->copy_in(V.points_new, points, 3 * (size_t)L * sizeof(double)));
  ok(h->flush_copies());                       // every array of the problem, merged into a few copies (copy_in)
  ok(hip_check(hipMemsetAsync(V.x, 0, ((size_t)n + 3 * (size_t)L) * sizeof(double), h->stream), "memset"));
  // (S is NOT cleared as a whole: every kernel touches structurally non-zero tiles only, and a trial's prologue clears exactly
  //  those.  The matrix is ldS^2 doubles -- 1.3 GB at 2 000 keyframes.)
  ok(hip_check(hipMemsetAsync(V.e_chi2, 0, (size_t)E * sizeof(double), h->stream), "memset"));
  ok(hip_check(hipMemsetAsync(V.ytmp, 0, ((size_t)V.n_pad + 64 + 2 * (SC.strips.size() / 2) + 2) * sizeof(double), h->stream), "memset"));   // ticket + hand-off flags of the back substitution
  ok(hip_check(hipMemsetAsync(V.flow_flags, 0, (4 * (size_t)V.n_strips_total + 2 * (size_t)nkb + 4) * sizeof(int32_t), h->stream), "memset"));
  h->solve_seq = 0; h->fuse_levels = true;
  ok(hip_check(hipStreamSynchronize(h->stream), "sync"));
  if (rc != DVM_OK) { h->free_problem(); return rc; }
  mark("state + memsets + sync");
  h->ms_structure = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  V.lambda = nullptr;   // damping travels by value (BaView::lambda_v)
  V.damp_s = rank == 0 ? 1.0 : 0.0;
  V.shard_rank = rank; V.shard_world = world;
  h->have_problem = true;
  {
    static const int max_free = [] { const char* e = std::getenv("DVM_BA_WINDOW_MAX_FREE"); return e ? std::atoi(e) : 6; }();
    h->win_mode = world == 1 && V.nfree <= std::min(max_free, 30) && !std::getenv("DVM_BA_NO_WINDOW");
    h->win_device_stale = false;
    if (h->win_mode) {
      h->ws_poses = pn; h->ws_points.assign(points, points + 3 * (size_t)L); h->ws_fixed.assign(fixed, fixed + P);
      h->ws_edges.assign(edges, edges + E); h->ws_cam = *cam;
      h->ws_chi2.assign(E, 0.0); h->ws_depth.assign(E, 0);
    }
  }
  return DVM_OK;
}

int dvm_ba_set_problem(dvm_ba* h, const double* poses, const uint8_t* fixed, int P, const double* points, int L,
                       const dvm_ba_edge* edges, int E, const dvm_ba_camera* cam) {
  const int rc = set_problem_impl(h, poses, fixed, P, points, L, edges, E, cam, 0, 1);
  if (rc == DVM_OK) h->sharded_api = false;
  return rc;
}
int dvm_ba_set_problem_sharded(dvm_ba* h, const double* poses, const uint8_t* fixed, int P, const double* points, int L,
                               const dvm_ba_edge* edges, int E, const dvm_ba_camera* cam, int rank, int world) {
  const int rc = set_problem_impl(h, poses, fixed, P, points, L, edges, E, cam, rank, world);
  if (rc == DVM_OK) h->sharded_api = true;
  return rc;
}
int dvm_ba_set_allreduce(dvm_ba* h, dvm_allreduce_fn fn, void* ctx, void* d_buf, int64_t cap_doubles) {
  if (!h) return DVM_ERR_INVALID;
  h->allreduce = fn; h->allreduce_ctx = ctx; h->ar_buf = static_cast<double*>(d_buf); h->ar_cap = cap_doubles;
  return DVM_OK;
}
// g2o's e->setLevel(1) / e->setRobustKernel(0) between two optimize() calls on the same graph (the two-round solves of
// Optimizer.cc:3474-3519 and :161-...): the flags take effect with the next dvm_ba_optimize; the state, the structure and
// the factorisation schedule stay as they are (a level-1 edge contributes exact zeros).
int dvm_ba_set_edge_flags(dvm_ba* h, const uint8_t* flags) {
  if (!h || !h->have_problem) { set_error("dvm_ba_set_edge_flags: no problem set"); return DVM_ERR_STATE; }
  if (h->world > 1) { set_error("dvm_ba_set_edge_flags: not available on a landmark-sharded problem"); return DVM_ERR_STATE; }
  DVM_HIP(hipSetDevice(h->device));
  if (!flags) { h->V.e_flags = nullptr; return DVM_OK; }
  // the flags change between the rounds of one problem (the welding BA's setLevel(1) round): they travel through the slot
  // reserved by set_problem -- staging them with copy_in() would take E more bytes of the pinned arena per call, which is only
  // rewound when the problem is replaced.  The slot may still be the source of the previous call's copy: wait for the stream.
  int rc = hip_check(hipStreamSynchronize(h->stream), "dvm_ba_set_edge_flags: sync");
  if (rc != DVM_OK) return rc;
  if (h->h_flags) {
    std::memcpy(h->h_flags, flags, (size_t)h->V.E);
    rc = hip_check(hipMemcpyAsync(h->d_flags, h->h_flags, (size_t)h->V.E, hipMemcpyHostToDevice, h->stream), "upload(flags)");
  } else {
    rc = hip_check(hipMemcpy(h->d_flags, flags, (size_t)h->V.E, hipMemcpyHostToDevice), "upload(flags)");   // the stream is idle: ordered
  }
  if (rc == DVM_OK) h->V.e_flags = h->d_flags;
  return rc;
}
int dvm_ba_schedule_info(const dvm_ba* h, int64_t* out) {
  if (!h || !h->have_problem || !out) return DVM_ERR_STATE;
  const BaTileSchedule& SC = h->sched;
  int64_t nlaunched = 0, ncols = 0, nstrips = 0, ntargets = 0, ncontrib = 0, ndiag_contrib = 0;
  for (int lv = 0; lv < SC.nlevels; lv++) {
    const int nc = SC.level_off[lv + 1] - SC.level_off[lv], ns = SC.strip_off[lv + 1] - SC.strip_off[lv], nt = SC.tgt_off[lv + 1] - SC.tgt_off[lv];
    if (lv == SC.nlevels - 1 && nc == 1 && ns == 0 && nt == 0) break;   // the rhs tile alone: not launched
    nlaunched++; ncols += nc; nstrips += ns; ntargets += nt;
    for (int t = SC.tgt_off[lv]; t < SC.tgt_off[lv + 1]; t++) {
      const int64_t n = SC.targets[4 * t + 3] - SC.targets[4 * t + 2];
      ncontrib += n;
      if (SC.targets[4 * t] == SC.targets[4 * t + 1]) ndiag_contrib += n;
    }
  }
  out[0] = nlaunched; out[1] = ncols; out[2] = nstrips; out[3] = ntargets; out[4] = ncontrib; out[5] = ndiag_contrib;
  out[6] = h->V.n_nz; out[7] = h->V.ldS; out[8] = h->V.nfree; out[9] = h->V.nblk; out[10] = h->V.E; out[11] = SC.ntiles;
  return DVM_OK;
}
int dvm_ba_solve_info(const dvm_ba* h, int64_t* out) {
  if (!h || !h->have_problem || !out) return DVM_ERR_STATE;
  out[0] = h->V.flow && h->fuse_levels ? 1 : 0; out[1] = h->V.n_flow_tasks; out[2] = h->sched.flow_leaves; out[3] = h->V.flow_wgs; out[4] = h->V.nkept; out[5] = h->V.ncamt;
  return DVM_OK;
}
int dvm_ba_profile(dvm_ba* h, int enable, double* ms4, int32_t* trials, int32_t* iters) {
  if (!h) return DVM_ERR_INVALID;
  if (ms4) for (int k = 0; k < 4; k++) ms4[k] = h->prof_ms[k];
  if (trials) *trials = h->prof_trials;
  if (iters) *iters = h->prof_iters;
  if (enable >= 0) {
    h->prof = enable != 0;
    if (h->prof && !h->pev[0]) for (auto& e : h->pev) if (hipEventCreate(&e) != hipSuccess) return DVM_ERR_HIP;
    for (double& v : h->prof_ms) v = 0;
    h->prof_trials = h->prof_iters = 0;
  }
  return DVM_OK;
}
int64_t dvm_ba_allreduce_doubles(const dvm_ba* h) {
  if (!h || !h->have_problem) return 0;
  return std::max<int64_t>({(int64_t)h->V.n_nz * 4096, 3 * (int64_t)h->V.L, 6 * (int64_t)h->V.nfree, 8});
}

// Wait until the device has released sequence number `seq` into the mapped host memory (BaPublish).  Spinning on host
// memory costs a few microseconds; hipStreamSynchronize + D2H copies cost ~60 us per LM iteration (r01 timeline).  The
// stream's status is polled sparsely so that a failed launch / device fault surfaces instead of hanging the caller.
static int wait_seq(dvm_ba* h, unsigned long long seq) {
  volatile unsigned long long* p = reinterpret_cast<volatile unsigned long long*>(h->h_vals + 8);
  const auto t0 = std::chrono::steady_clock::now();
  double next_query = 5e-3;   // seconds.  NOT earlier: hipStreamQuery puts a marker packet on the queue when work is pending, and a
                              // marker between two kernels of a trial (the next trial's k_schur is enqueued before this wait)
                              // costs the device ~6 us -- a healthy trial publishes within a millisecond
  for (unsigned spin = 0;; spin++) {
    if (*p >= seq) { std::atomic_thread_fence(std::memory_order_acquire); return DVM_OK; }
    if ((spin & 0xFFF) == 0xFFF) {
      const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (waited < next_query) continue;
      next_query = waited + 5e-3;
      const hipError_t q = hipStreamQuery(h->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return hip_check(q, "bundle adjustment stream");
      if (q == hipSuccess && *p < seq) {           // everything ran, nothing was published: should be impossible
        if (*p >= seq) continue;
        set_error("bundle adjustment: phase result was not published");
        return DVM_ERR_HIP;
      }
      if (waited > 30.0) {
        set_error("bundle adjustment: timed out waiting for the device");
        return DVM_ERR_HIP;
      }
    }
  }
}

// optimizer.optimize(iterations) with OptimizationAlgorithmLevenberg.  stop_flag mirrors g2o's
// setForceStopFlag(bool*): polled at the top of every iteration and after every trial.
int dvm_ba_optimize(dvm_ba* h, int iterations, const volatile uint8_t* stop_flag, dvm_ba_stats* st) {
  if (!h || !h->have_problem) { set_error("dvm_ba_optimize: no problem set"); return DVM_ERR_STATE; }
  DVM_HIP(hipSetDevice(h->device));
  BaView& V = h->V;
  hipStream_t s = h->stream;
  if (h->win_mode && !V.e_flags && !h->prof) {
    // the whole optimize() as one launch of the sequential-order kernel; the state continues from the previous call without a
    // second normalisation (g2o keeps the graph between optimize() calls)
    dvm_ba_window w{};
    w.n_poses = V.P; w.n_points = V.L; w.n_edges = V.E; w.iterations = iterations;
    w.poses = h->ws_poses.data(); w.fixed = h->ws_fixed.data(); w.points = h->ws_points.data(); w.edges = h->ws_edges.data(); w.cam = h->ws_cam;
    std::vector<double> po(h->ws_poses.size()), xo(h->ws_points.size());
    w.poses_out = po.data(); w.points_out = xo.data(); w.edge_chi2_out = h->ws_chi2.data(); w.depth_positive_out = h->ws_depth.data();
    dvm_ba_stats ws;
    const int rc = dvm_ba_optimize_windows_impl(h->device, &w, 1, stop_flag, &ws, /*normalize_input=*/false, /*fast=*/false, /*cluster=*/0);
    if (rc == DVM_ERR_CAPACITY && !h->win_device_stale) {
      // what the sequential-order kernel cannot hold (a landmark with more rows than a chunk: duplicate observations; an index block
      // beyond its staging area) the tile solver can: this problem is its from now on (round-4 advice)
      h->win_mode = false;
    } else {
      if (rc != DVM_OK) return rc;
      h->ws_poses.swap(po); h->ws_points.swap(xo);
      h->win_device_stale = true;
      if (st) { *st = ws; st->ms_structure = h->ms_structure; }
      return DVM_OK;
    }
  }
  if (h->win_mode && h->win_device_stale) {      // the tile solver takes over (edge flags, profiling): it continues from the window kernel's state
    DVM_HIP(hipStreamSynchronize(s));
    DVM_HIP(hipMemcpy(V.poses, h->ws_poses.data(), h->ws_poses.size() * sizeof(double), hipMemcpyHostToDevice));
    DVM_HIP(hipMemcpy(V.points, h->ws_points.data(), h->ws_points.size() * sizeof(double), hipMemcpyHostToDevice));
    DVM_HIP(hipMemcpy(V.poses_new, h->ws_poses.data(), h->ws_poses.size() * sizeof(double), hipMemcpyHostToDevice));
    DVM_HIP(hipMemcpy(V.points_new, h->ws_points.data(), h->ws_points.size() * sizeof(double), hipMemcpyHostToDevice));
    DVM_HIP(hipMemcpy(V.e_chi2, h->ws_chi2.data(), h->ws_chi2.size() * sizeof(double), hipMemcpyHostToDevice));
    h->win_device_stale = false;
  }
  if (st) { std::memset(st, 0, sizeof(*st)); st->ms_structure = h->ms_structure; }
  const auto t0 = std::chrono::steady_clock::now();
  const bool dbg_time = std::getenv("DVM_BA_DEBUG_SCHEDULE") != nullptr;
  auto mark = [&](const char* what, int a) {
    if (dbg_time) std::fprintf(stderr, "optimize: %-18s %d %9.3f ms\n", what, a, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  enum { S_CHI = 0, S_TMPCHI = 1, S_SCALE = 2, S_MAXDIAG = 3, S_FAIL = 6 };
  auto pub = [&](int slot, int counter, bool publish, bool with_fail) {
    BaPublish p;
    p.dev_vals = h->d_dev_vals; p.host_vals = h->d_vals; p.host_seq = reinterpret_cast<unsigned long long*>(h->d_vals + 8);
    p.seq = publish ? ++h->seq : 0; p.counter = h->d_counter + counter; p.d_fail = with_fail ? h->d_fail : nullptr;
    p.slot = slot; p.publish = publish ? 1 : 0;
    p.spec = nullptr; p.cur_chi = 0; p.lambda = 0; p.n_bad = 0; p.spec_mode = 0;
    return p;
  };
  // a one-rank job that registered a collective takes the sharded flow too (sum over one rank = identity): that is how the
  // RCCL path is exercised on a single-GPU box (tests/test_gpu_rccl.py)
  const bool sharded = h->world > 1 || (h->sharded_api && h->allreduce && h->ar_buf);
  if (sharded && (!h->allreduce || !h->ar_buf || h->ar_cap < dvm_ba_allreduce_doubles(h))) {
    set_error("dvm_ba_optimize: sharded problem without dvm_ba_set_allreduce (callback + device buffer of dvm_ba_allreduce_doubles())");
    return DVM_ERR_STATE;
  }
  // collectives of the sharded mode: in-place over ranks, on the caller's device buffer (ordered on the BA stream) or on a
  // few host doubles
  auto ar_dev = [&](int64_t n, int op) { return h->allreduce(h->allreduce_ctx, h->ar_buf, n, 0, op, (void*)s) == 0 ? DVM_OK : DVM_ERR_HIP; };
  auto ar_host = [&](double* v, int64_t n, int op) { return h->allreduce(h->allreduce_ctx, v, n, 1, op, (void*)s) == 0 ? DVM_OK : DVM_ERR_HIP; };
  double lambda = -1, ni = 2;
  int nBad = 0, it_done = 0, trials_total = 0, stop = 0;
  double chi_last = 0;
  auto terminate = [&]() { return stop_flag && *stop_flag; };
  // Speculative linearisation: every trial evaluates its state with the Jacobian kernel (chi2 is the same sum either way)
  // into the alternate buffers and accumulates Hpp / Hll there while the host waits for chi2 and decides.  Trials are
  // accepted far more often than not; an accepted trial hands the next iteration its state AND that state's linearisation
  // by swapping pointers, so the iteration starts at the Schur complement (computeActiveErrors + buildSystem already done,
  // on the same values g2o would compute them on); a rejected trial's buffers are simply overwritten by the next one.
  bool lin_ready = false;
  double spec_chi = 0;
  // Speculation on acceptance, one step further: the workgroup that publishes a trial's chi2 takes the accept / reject
  // decision and the next damping itself (BaPublish::spec), the linearisation's launch derives the landmark inverses of the
  // NEXT trial from it, and that trial's Schur complement is enqueued right behind -- all before the host has seen chi2, so an
  // accepted trial (the normal case) costs the device neither the host's decision latency nor a prologue launch.  The host
  // takes the same decision on the same numbers and only keeps what was enqueued if the device's damping equals its own bit
  // for bit; otherwise (a rejected trial: the speculative launch has done nothing; a damping that differs in the last bit: it
  // has filled S) the next trial starts the ordinary way, after emptying S if need be.
  const bool speculate = !sharded && h->speculate && !h->prof;   // (the profiling pass times every phase of a trial on its own)
  // g2o's buildStructure (iteration 0 of every optimize()) reallocates _x: "the last successful solve" starts empty
  DVM_HIP(hipMemsetAsync(V.x, 0, ((size_t)6 * V.nfree + 3 * (size_t)V.L) * sizeof(double), s));
  bool schur_enqueued = false;      // the NEXT trial's k_schur is already on the stream, for (V after the swap, lambda)
  bool tiles_clear = false;         // the structurally non-zero tiles of S are empty: k_accum's launch was the last to touch S
  for (int it = 0; it < iterations && !terminate(); it++) {
    int rc = DVM_OK;
    bool first_spec = false;
    if (!lin_ready) {
    // computeActiveErrors + activeRobustChi2 + buildSystem (one fused edge pass at the current state).  chi2 reaches the
    // host from the edge pass itself, so the accumulation kernels below run while the host prepares the first trial.
    if (h->prof) hipEventRecord(h->pev[0], s);
    ba_launch_edge_eval(s, V, true, pub(S_CHI, 0, it != 0 || sharded, false));
    if (it == 0 && !sharded) {
      // computeLambdaInit's max |diag| comes out of the accumulation launch itself; with speculation on, the first trial's
      // prologue and Schur complement follow at once on the damping the device derives from it (checked below)
      BaPublish pm = pub(S_MAXDIAG, 1, true, false);
      first_spec = speculate && iterations > 0;
      if (first_spec) { pm.spec = h->d_spec; pm.spec_mode = 1; }
      ba_launch_accum(s, V, nullptr, &pm);
      tiles_clear = true;
      if (first_spec) {
        BaView VA = V;
        VA.lambda = h->d_spec;
        ba_launch_schur(s, VA, h->d_fail);
        tiles_clear = false;
      }
    } else {
      ba_launch_accum(s, V);
      tiles_clear = true;
    }
    if (h->prof) hipEventRecord(h->pev[1], s);
    rc = hip_check(hipGetLastError(), "bundle adjustment launch");
    if (rc == DVM_OK) rc = wait_seq(h, h->seq);
    if (rc != DVM_OK) return rc;
    mark("linearised", it);
    if (h->prof) { hipEventSynchronize(h->pev[1]); float ms = 0; hipEventElapsedTime(&ms, h->pev[0], h->pev[1]); h->prof_ms[0] += ms; }
    if (sharded) {
      double v = h->h_vals[S_CHI];                       // chi2 of the local edges (read before the next publication rewrites the slots)
      if ((rc = ar_host(&v, 1, 0)) != DVM_OK) return rc;
      if (it == 0) {   // computeLambdaInit needs max |diag| of the SUMMED Hpp and of every Hll
        ba_launch_hpp_diag(s, V, h->ar_buf);
        if ((rc = ar_dev(6 * (int64_t)V.nfree, 0)) != DVM_OK) return rc;
        ba_launch_max_diag_sharded(s, V, h->ar_buf, pub(S_MAXDIAG, 1, true, false));
        if ((rc = wait_seq(h, h->seq)) != DVM_OK) return rc;
        double m = h->h_vals[S_MAXDIAG];
        if ((rc = ar_host(&m, 1, 1)) != DVM_OK) return rc;
        h->h_vals[S_MAXDIAG] = m;
      }
      h->h_vals[S_CHI] = v;
    }
    } else {
      h->h_vals[S_CHI] = spec_chi;     // the accepted trial's chi2: the same kernel on the same state
      lin_ready = false;
    }
    if (h->prof) h->prof_iters++;
    double currentChi = h->h_vals[S_CHI], tempChi = currentChi;
    const double iniChi = currentChi;
    if (it == 0) {
      if (st) st->chi2_initial = currentChi;
      lambda = 1e-5 * h->h_vals[S_MAXDIAG];  // computeLambdaInit, _tau = 1e-5
      ni = 2; nBad = 0;
      if (first_spec) {
        const double dev_l = h->h_vals[7];
        if (std::memcmp(&dev_l, &lambda, sizeof(double)) == 0) schur_enqueued = true;   // (else: it ran on another damping, S is cleared below)
        if (st) { st->spec_trials++; st->spec_kept += schur_enqueued ? 1 : 0; }
      }
    }
    double rho = 0;
    int qmax = 0;
    bool spec_now = false;
    do {
      // one trial = setLambda + Schur complement + reduced solve + landmarks + oplus into the TRIAL state + its chi2:
      // ~40 asynchronous launches, no copy, no host synchronisation inside (push / pop are a pointer swap)
      V.lambda_v = lambda;
      for (int attempt = 0;; attempt++) {
        if (h->prof) hipEventRecord(h->pev[0], s);
        if (schur_enqueued && attempt == 0) schur_enqueued = false;        // this trial's Schur complement is on the stream already
        else {
          // S must be empty where the Schur complement does not write: normally the linearisation's launch has seen to it; not
          // after a failed attempt (its solve has consumed S), a speculative launch on another damping, or a chi2-only evaluation
          if (!tiles_clear) ba_launch_clear_tiles(s, V);
          ba_launch_schur(s, V, h->d_fail);
        }
        tiles_clear = false;
        if (h->prof) hipEventRecord(h->pev[1], s);
        if (sharded) {   // sum the partial reduced systems (non-zero tiles incl. the rhs row): ~6 MB at 500 keyframes
          ba_launch_pack_tiles(s, V, h->ar_buf, false);
          if ((rc = ar_dev((int64_t)V.n_nz * 4096, 0)) != DVM_OK) return rc;
          ba_launch_pack_tiles(s, V, h->ar_buf, true);
        }
        {
          BaView VS = V;                                   // in-launch hand-offs (k_chol_trsm_update) only while they have never timed out
          // (the fused level launches assume their whole grid resident: not with several ranks on one GPU, nor after a timeout;
          //  every rank of a sharded solve solves the summed system redundantly)
          if (sharded || !h->fuse_levels) VS.strip_flags = nullptr;
          // (a sharded solve keeps the flow form -- every rank solves the summed system redundantly, on its own GPU; a wait that
          //  gives up on ANY rank is all-reduced below and every rank repeats the trial with the level launches)
          if (!h->fuse_levels) VS.flow = 0;
          ba_launch_cholesky_solve(s, VS, h->d_fail, ++h->solve_seq);
        }
        if (h->prof) hipEventRecord(h->pev[2], s);
        ba_launch_backsub_update(s, V, pub(S_SCALE, 2, false, false), h->d_fail);
        {
          BaView VT = V;                                   // the trial state, linearised into the alternate buffers
          VT.poses = V.poses_new; VT.points = V.points_new;
          VT.e_lin = h->alt_lin; VT.e_linA = h->alt_linA; VT.e_W = h->alt_W; VT.Hpp = h->alt_Hpp; VT.bp = h->alt_bp; VT.Hll = h->alt_Hll; VT.bl = h->alt_bl;
          // (no speculation into an iteration that will not run, nor on a repeated attempt)
          spec_now = speculate && attempt == 0 && it + 1 < iterations;
          BaPublish pe = pub(S_TMPCHI, 0, true, true);
          if (spec_now) { pe.spec = h->d_spec; pe.cur_chi = currentChi; pe.lambda = lambda; pe.n_bad = nBad; }
          if (it + 1 >= iterations && !sharded) {
            ba_launch_edge_eval(s, V, false, pe);          // the budget's last iteration: nobody will use a linearisation, chi2 alone (same sum)
          } else {
            ba_launch_edge_eval(s, VT, true, pe);
            ba_launch_accum(s, VT, spec_now ? h->d_spec : nullptr);   // runs while the host waits for chi2 and decides
            tiles_clear = true;
          }
          if (h->prof) hipEventRecord(h->pev[3], s);
          if (spec_now) {
            BaView VA = VT;                                // the next trial as it looks if this one is accepted
            VA.poses_new = V.poses; VA.points_new = V.points;
            VA.lambda = h->d_spec;
            ba_launch_schur_speculative(s, VA, h->d_fail);
            tiles_clear = false;                           // (true again below if the device rejected the trial: that launch does nothing)
          }
        }
        rc = hip_check(hipGetLastError(), "bundle adjustment launch");
        mark("trial launched", it);
        if (rc == DVM_OK) rc = wait_seq(h, h->seq);
        if (rc != DVM_OK) return rc;
        mark("trial published", it);
        bool timed_out = h->h_vals[S_FAIL] == 2.0;
        if (sharded) {
          // chi2 and the gain-ratio denominator are sums over ranks; the failure flag rides along (a sum of non-negative flags =
          // "any rank failed"; 1e6 = "a wait inside some rank's solve gave up"): every rank must take the SAME accept / reject /
          // repeat decision, or the LM state and the collectives of the following trials diverge
          double v[3] = {h->h_vals[S_TMPCHI], h->h_vals[S_SCALE], timed_out ? 1e6 : (h->h_vals[S_FAIL] != 0.0 ? 1.0 : 0.0)};
          if ((rc = ar_host(v, 3, 0)) != DVM_OK) return rc;
          timed_out = v[2] >= 1e6;
          h->h_vals[S_TMPCHI] = v[0]; h->h_vals[S_SCALE] = v[1]; h->h_vals[S_FAIL] = timed_out ? 2.0 : (v[2] != 0.0 ? 1.0 : 0.0);
        }
        // a wait inside the solve gave up (other work held the compute units its producer needed): nothing was decided on this
        // result -- the same trial runs again (on every rank of a sharded solve), with one launch per phase from now on
        if (timed_out && h->fuse_levels && attempt == 0) {
          h->fuse_levels = false;
          if (spec_now) {                                  // whatever was enqueued behind the failed attempt is void
            DVM_HIP(hipStreamSynchronize(s));
            spec_now = false;
          }
          continue;
        }
        break;
      }
      const double dev_next = spec_now ? h->h_vals[7] : -1.0;   // the device's decision: next damping, or -1 (rejected)
      if (spec_now && dev_next < 0) tiles_clear = true;
      if (h->prof) {
        hipEventSynchronize(h->pev[3]);
        for (int k = 0; k < 3; k++) { float ms = 0; hipEventElapsedTime(&ms, h->pev[k], h->pev[k + 1]); h->prof_ms[1 + k] += ms; }
        h->prof_trials++;
      }
      // A failed linear solve (optimization_algorithm_levenberg.cpp:107-127): g2o still applies update(x) -- x being whatever the
      // last successful solve left (zeros before the first) --, evaluates the errors there, then overrides tempChi with max()
      // and divides by computeScale() of that x.  The kernels did the same (k_chol_backsolve / k_point_backsub keep x on a
      // failure); max() is finite, so a negative scale would even accept the step, exactly as in the reference.
      const bool ok2 = (h->h_vals[S_FAIL] == 0.0);
      tempChi = ok2 ? h->h_vals[S_TMPCHI] : std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = h->h_vals[S_SCALE];
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - dvm::f64_cube(2 * rho - 1);   // pow(2 rho - 1, 3) as the spec both sides share (csrc/f64_spec.h: the correctly rounded cube)
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
        std::swap(V.poses, V.poses_new);       // discardTop(): the trial state becomes the estimate
        std::swap(V.points, V.points_new);
        std::swap(V.e_lin, h->alt_lin); std::swap(V.e_linA, h->alt_linA); std::swap(V.e_W, h->alt_W); std::swap(V.Hpp, h->alt_Hpp); std::swap(V.bp, h->alt_bp);
        std::swap(V.Hll, h->alt_Hll); std::swap(V.bl, h->alt_bl);
        lin_ready = true; spec_chi = tempChi;
        if (spec_now) {
          if (std::memcmp(&dev_next, &lambda, sizeof(double)) == 0) schur_enqueued = true;   // same decision, same damping: keep it
          if (st) { st->spec_trials++; st->spec_kept += schur_enqueued ? 1 : 0; }
        }
      } else {
        lambda *= ni;
        ni *= 2;                               // pop(): (poses, points) were never touched
        if (st && spec_now) st->spec_trials++;
      }
      qmax++;
      trials_total++;
    } while (rho < 0 && qmax < 10 && !terminate());
    it_done++;
    chi_last = currentChi;
    if (st && it < 64) { st->trials_per_iter[it] = qmax; st->chi2_per_iter[it] = currentChi; st->lambda_per_iter[it] = lambda; }
    if (qmax == 10 || rho == 0) { stop = 1; break; }
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) { stop = 2; break; }
  }
  if (sharded) {   // every rank ends with every landmark: owners contribute theirs, the sum is scattered back
    ba_launch_points_exchange(s, V, h->ar_buf, false);
    const int rc = ar_dev(3 * (int64_t)V.L, 0);
    if (rc != DVM_OK) return rc;
    ba_launch_points_exchange(s, V, h->ar_buf, true);
  }
  // The answer is complete: the last trial's publication is behind every launch that writes the state.  What may still be on
  // the stream is the speculative linearisation behind that trial (it leaves at once when the device saw the optimisation end);
  // every entry point that reads device memory synchronises the stream itself, so the blocking wait here -- 50-80 us of
  // interrupt latency on an almost idle stream, per call -- is only paid where a collective has to be finished.
  if (sharded) DVM_HIP(hipStreamSynchronize(s));
  if (h->win_mode) {     // a window-mode problem that took the tile solver for this call: the host copies follow the device state
    DVM_HIP(hipStreamSynchronize(s));
    DVM_HIP(hipMemcpy(h->ws_poses.data(), V.poses, h->ws_poses.size() * sizeof(double), hipMemcpyDeviceToHost));
    DVM_HIP(hipMemcpy(h->ws_points.data(), V.points, h->ws_points.size() * sizeof(double), hipMemcpyDeviceToHost));
    DVM_HIP(hipMemcpy(h->ws_chi2.data(), V.e_chi2, h->ws_chi2.size() * sizeof(double), hipMemcpyDeviceToHost));
  }
  mark("returning", it_done);
  if (st) {
    st->iterations = it_done; st->total_trials = trials_total; st->stop_reason = stop;
    st->chi2_final = chi_last; st->lambda_final = lambda;
    st->ms_optimize = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  return DVM_OK;
}

int dvm_ba_get_result(dvm_ba* h, double* poses, double* points) {
  if (!h || !h->have_problem) return DVM_ERR_STATE;
  DVM_HIP(hipSetDevice(h->device));
  if (h->win_mode && h->win_device_stale) {       // the state of the window kernel's last run
    if (poses) std::memcpy(poses, h->ws_poses.data(), h->ws_poses.size() * sizeof(double));
    if (points) std::memcpy(points, h->ws_points.data(), h->ws_points.size() * sizeof(double));
    return DVM_OK;
  }
  DVM_HIP(hipStreamSynchronize(h->stream));
  if (poses) DVM_HIP(hipMemcpy(poses, h->V.poses, 7 * (size_t)h->V.P * sizeof(double), hipMemcpyDeviceToHost));
  if (points) DVM_HIP(hipMemcpy(points, h->V.points, 3 * (size_t)h->V.L * sizeof(double), hipMemcpyDeviceToHost));
  return DVM_OK;
}

// e->chi2() of every edge as g2o reports it after optimize() (value at the last error evaluation)
// and e->isDepthPositive() at the final estimates (Optimizer.cc:1317-1354, :297-312).
int dvm_ba_edge_chi2(dvm_ba* h, double* chi2, uint8_t* depth_positive) {
  if (!h || !h->have_problem) return DVM_ERR_STATE;
  DVM_HIP(hipSetDevice(h->device));
  if (h->win_mode && h->win_device_stale) {
    if (chi2) std::memcpy(chi2, h->ws_chi2.data(), h->ws_chi2.size() * sizeof(double));
    if (depth_positive) std::memcpy(depth_positive, h->ws_depth.data(), h->ws_depth.size());
    return DVM_OK;
  }
  if (depth_positive) ba_launch_edge_depth(h->stream, h->V, h->d_depth);
  DVM_HIP(hipStreamSynchronize(h->stream));
  if (chi2) DVM_HIP(hipMemcpy(chi2, h->V.e_chi2, (size_t)h->V.E * sizeof(double), hipMemcpyDeviceToHost));
  if (depth_positive) DVM_HIP(hipMemcpy(depth_positive, h->d_depth, (size_t)h->V.E, hipMemcpyDeviceToHost));
  return DVM_OK;
}

void* dvm_ba_stream(dvm_ba* h) { return h ? (void*)h->stream : nullptr; }

int dvm_pose_optimize(int device, const double* pose_in, const double* Xw, const double* obs, const double* inv_sigma2,
                      const int32_t* n, int stride, int batch, const dvm_ba_camera* cam, double* pose_out,
                      uint8_t* outlier, int32_t* n_inliers) {
  if (!pose_in || !Xw || !obs || !inv_sigma2 || !n || !cam || !pose_out || !outlier || !n_inliers || stride < 1 || batch < 1) {
    set_error("dvm_pose_optimize: bad arguments");
    return DVM_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  if (device < 0 || device >= ndev) return DVM_ERR_INVALID;
  for (int f = 0; f < batch; f++) if (n[f] < 0 || n[f] > stride) { set_error("n[f] out of range"); return DVM_ERR_INVALID; }
  DVM_HIP(hipSetDevice(device));
  const size_t B = (size_t)batch, S = (size_t)stride;
  Stage st;   // the calling thread's staging context: one upload, one download, one synchronisation (host_stage.h)
  // A frame or a few (Tracking's per-frame call): k_pose_optimize reads every correspondence ONCE, into registers, and writes the
  // results once at the end, so the arrays stay in mapped host memory and no copy command brackets the kernel.  Beyond
  // kPoseEdgesPerThread x 256 correspondences per frame the kernel re-reads them every iteration: copied, as large batches are.
  const bool direct = S <= 1280 && B * S <= 8192;
  int iP, iX, iO, iW, iN, oP, oL, oI;
  if (direct) {
    iP = st.in_mapped(pose_in, B * 7 * 8); iX = st.in_mapped(Xw, B * S * 3 * 8); iO = st.in_mapped(obs, B * S * 2 * 8);
    iW = st.in_mapped(inv_sigma2, B * S * 8); iN = st.in_mapped(n, B * 4);
    oP = st.out_mapped(pose_out, B * 7 * 8); oL = st.out_mapped(outlier, B * S); oI = st.out_mapped(n_inliers, B * 4);
  } else {
    iP = st.in(pose_in, B * 7 * 8); iX = st.in(Xw, B * S * 3 * 8); iO = st.in(obs, B * S * 2 * 8); iW = st.in(inv_sigma2, B * S * 8);
    iN = st.in(n, B * 4);
    oP = st.out(pose_out, B * 7 * 8); oL = st.out(outlier, B * S); oI = st.out(n_inliers, B * 4);
  }
  const int sC = st.scratch(B * S * 8);
  int rc = st.upload();
  if (rc != DVM_OK) return rc;
  ba_launch_pose_optimize(st.stream(), st.ptr<double>(iP), st.ptr<double>(iX), st.ptr<double>(iO), st.ptr<double>(iW), st.ptr<int32_t>(iN), stride,
                          batch, cam->fx, cam->fy, cam->cx, cam->cy, st.ptr<double>(oP), st.ptr<uint8_t>(oL), st.ptr<int32_t>(oI),
                          st.ptr<double>(sC));
  rc = hip_check(hipGetLastError(), "pose_optimize launch");
  return rc == DVM_OK ? st.download() : rc;
}

int dvm_optimize_sim3(int device, double* S12, int fix_scale, const double* P1c, const double* P2c, const double* obs1,
                      const double* obs2, const double* w1, const double* w2, int N, const double* K1, const double* K2,
                      double th2, uint8_t* inlier, int32_t* n_inliers) {
  if (!S12 || !P1c || !P2c || !obs1 || !obs2 || !w1 || !w2 || !K1 || !K2 || !inlier || !n_inliers || N < 1) {
    set_error("dvm_optimize_sim3: bad arguments");
    return DVM_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  if (device < 0 || device >= ndev) return DVM_ERR_INVALID;
  DVM_HIP(hipSetDevice(device));
  const size_t n = (size_t)N;
  double K[8], S_out[8];
  std::memcpy(K, K1, 32); std::memcpy(K + 4, K2, 32);
  Stage st;   // S12 is refined in place on the device: staged in, fetched into S_out, handed over only if the solve succeeded
  const int ioS = st.add(S12, S_out, 64), iK = st.in(K, 64), iP1 = st.in(P1c, 24 * n), iP2 = st.in(P2c, 24 * n), iO1 = st.in(obs1, 16 * n),
            iO2 = st.in(obs2, 16 * n), iW1 = st.in(w1, 8 * n), iW2 = st.in(w2, 8 * n), oN = st.out(n_inliers, 4), oI = st.out(inlier, n),
            sChi = st.scratch(16 * n), sFl = st.scratch(2 * n);
  int rc = st.upload();
  if (rc != DVM_OK) return rc;
  ba_launch_optimize_sim3(nullptr, st.ptr<double>(ioS), fix_scale, st.ptr<double>(iP1), st.ptr<double>(iP2), st.ptr<double>(iO1), st.ptr<double>(iO2),
                          st.ptr<double>(iW1), st.ptr<double>(iW2), N, st.ptr<double>(iK), th2, st.ptr<uint8_t>(oI), st.ptr<int32_t>(oN),
                          st.ptr<double>(sChi), st.ptr<uint8_t>(sFl));
  rc = hip_check(hipGetLastError(), "optimize_sim3 launch");
  if (rc == DVM_OK) rc = st.download();
  if (rc == DVM_OK && *n_inliers > 0) std::memcpy(S12, S_out, 64);
  return rc;
}

int dvm_sim3_hypotheses(int device, const float* P1c, const float* P2c, const float* max_err1, const float* max_err2, int N,
                        const float* K1, const float* K2, const int32_t* triples, int H, int fix_scale, float* T12,
                        int32_t* n_inliers, uint8_t* inlier_mask) {
  if (!P1c || !P2c || !max_err1 || !max_err2 || !K1 || !K2 || !triples || !T12 || !n_inliers || !inlier_mask || N < 3 || H < 1) {
    set_error("dvm_sim3_hypotheses: bad arguments");
    return DVM_ERR_INVALID;
  }
  for (int i = 0; i < 3 * H; i++) if (triples[i] < 0 || triples[i] >= N) { set_error("dvm_sim3_hypotheses: sample index out of range"); return DVM_ERR_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  if (device < 0 || device >= ndev) return DVM_ERR_INVALID;
  DVM_HIP(hipSetDevice(device));
  const size_t n = (size_t)N, hh = (size_t)H;
  float K[8];
  std::memcpy(K, K1, 16); std::memcpy(K + 4, K2, 16);
  Stage st;
  const int iP1 = st.in(P1c, 12 * n), iP2 = st.in(P2c, 12 * n), iE1 = st.in(max_err1, 4 * n), iE2 = st.in(max_err2, 4 * n), iK = st.in(K, 32),
            iT = st.in(triples, 12 * hh), oT = st.out(T12, 52 * hh), oN = st.out(n_inliers, 4 * hh), oM = st.out(inlier_mask, hh * n);
  int rc = st.upload();
  if (rc != DVM_OK) return rc;
  ba_launch_sim3_hypotheses(nullptr, st.ptr<float>(iP1), st.ptr<float>(iP2), st.ptr<float>(iE1), st.ptr<float>(iE2), N, st.ptr<float>(iK),
                            st.ptr<int32_t>(iT), H, fix_scale, st.ptr<float>(oT), st.ptr<int32_t>(oN), st.ptr<uint8_t>(oM));
  rc = hip_check(hipGetLastError(), "sim3_hypotheses launch");
  return rc == DVM_OK ? st.download() : rc;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------- batch of independent problems
// K independent bundle adjustments solved CONCURRENTLY: the LocalBundleAdjustment windows of several agents that share this GPU
// (BASELINE.json config 4 with more agents than GPUs; in the reference every agent's LocalMapping thread runs its own
// Optimizer::LocalBundleAdjustment, Optimizer.cc:1030-1387).  One window on the tile solver is a chain of ~400 small launches with
// the host's LM decisions in between: 1.2 ms during which the GPU is mostly idle.  Up to `threads` host threads each take a solver
// handle from a process-wide pool (a handle keeps its buffers, streams and pinned words: creating one costs milliseconds) and pull
// windows from a shared counter, so the chains of different windows interleave on the device.  Every window gives exactly what
// dvm_ba_set_problem + dvm_ba_optimize(iterations) + dvm_ba_get_result + dvm_ba_edge_chi2 on its own handle give -- the routing of
// small problems to the sequential-order kernel included.
namespace {
struct BaPool {
  std::mutex m;
  std::vector<std::pair<int, dvm_ba*>> idle;   // (device, handle)
  dvm_ba* take(int device, int* rc) {
    {
      std::lock_guard<std::mutex> lock(m);
      for (size_t i = 0; i < idle.size(); i++)
        if (idle[i].first == device) { dvm_ba* h = idle[i].second; idle.erase(idle.begin() + (long)i); return h; }
    }
    dvm_ba* h = nullptr;
    *rc = dvm_ba_create(device, &h);
    return *rc == DVM_OK ? h : nullptr;
  }
  void give(int device, dvm_ba* h) {
    std::lock_guard<std::mutex> lock(m);
    // at most 8 idle handles a device (each still holds its last problem's device and pinned buffers: 64 of them were the process's for good)
    int same = 0;
    for (const auto& e : idle) same += e.first == device;
    if (same < 8) { idle.emplace_back(device, h); return; }
    dvm_ba_destroy(h);
  }
};
BaPool& ba_pool() {
  static BaPool* p = new BaPool();   // never destroyed: its handles own HIP objects, and the runtime may be gone at static destruction
  return *p;
}
int solve_one_window(dvm_ba* h, const dvm_ba_window& w, const volatile uint8_t* stop_flag, dvm_ba_stats* st) {
  int rc = dvm_ba_set_problem(h, w.poses, w.fixed, w.n_poses, w.points, w.n_points, w.edges, w.n_edges, &w.cam);
  if (rc != DVM_OK) return rc;
  rc = dvm_ba_optimize(h, w.iterations, stop_flag, st);
  if (rc != DVM_OK) return rc;
  if (w.poses_out || w.points_out) {
    // dvm_ba_get_result wants both arrays: a window that asks for one of them gets the other into scratch
    std::vector<double> tp, tq;
    double* po = w.poses_out;
    double* qo = w.points_out;
    if (!po) { tp.resize((size_t)w.n_poses * 7); po = tp.data(); }
    if (!qo) { tq.resize((size_t)std::max(w.n_points, 1) * 3); qo = tq.data(); }
    rc = dvm_ba_get_result(h, po, qo);
    if (rc != DVM_OK) return rc;
  }
  if (w.edge_chi2_out || w.depth_positive_out) rc = dvm_ba_edge_chi2(h, w.edge_chi2_out, w.depth_positive_out);
  return rc;
}
}  // namespace

extern "C" int dvm_ba_optimize_batch(int device, const dvm_ba_window* windows, int K, int threads, const volatile uint8_t* stop_flag,
                                     dvm_ba_stats* stats) {
  if (K < 0 || (K && !windows)) return DVM_ERR_INVALID;
  if (K == 0) return DVM_OK;
  for (int k = 0; k < K; k++) {
    const dvm_ba_window& w = windows[k];
    if (w.n_poses < 1 || w.n_points < 0 || w.n_edges < 0 || w.iterations < 0 || !w.poses || !w.fixed || (w.n_points && !w.points) ||
        (w.n_edges && !w.edges)) {
      set_error("dvm_ba_optimize_batch: window " + std::to_string(k) + " is incomplete");
      return DVM_ERR_INVALID;
    }
  }
  // default 4: the runtime multiplexes a process's streams onto 4 hardware queues (GPU_MAX_HW_QUEUES); measured on 32 LBA windows:
  // 1 thread 8.5 k it/s, 2: 15.5 k, 4: 22-23 k, 8: 19-27 k (25-27 k with 8 or 16 hardware queues), 32: 17-30 k
  int T = threads > 0 ? threads : 4;
  if (const char* e = getenv("DVM_BA_BATCH_THREADS")) T = std::max(1, atoi(e));
  T = std::max(1, std::min(T, K));
  std::atomic<int> next{0};
  std::vector<int> rcs((size_t)K, DVM_OK);
  std::vector<char> done((size_t)K, 0);
  std::vector<std::string> errs((size_t)T);
  std::vector<int> trc((size_t)T, DVM_OK);
  auto work = [&](int t) {
    if (hipSetDevice(device) != hipSuccess) { trc[t] = DVM_ERR_HIP; errs[t] = "hipSetDevice failed"; return; }
    int rc = DVM_OK;
    dvm_ba* h = ba_pool().take(device, &rc);
    if (!h) { trc[t] = rc; errs[t] = last_error_cstr(); return; }
    for (;;) {
      const int k = next.fetch_add(1);
      if (k >= K) break;
      rcs[k] = solve_one_window(h, windows[k], stop_flag, stats ? &stats[k] : nullptr);
      done[k] = 1;
      if (rcs[k] != DVM_OK && errs[t].empty()) errs[t] = "window " + std::to_string(k) + ": " + last_error_cstr();
    }
    ba_pool().give(device, h);
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < T; t++) pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool) th.join();
  // a worker that could not start (no handle, no device) is an error only if windows were left unsolved: the other workers drain the list
  if (!std::all_of(done.begin(), done.end(), [](char d) { return d != 0; }))
    for (int t = 0; t < T; t++)
      if (trc[t] != DVM_OK) { set_error("dvm_ba_optimize_batch: " + errs[t]); return trc[t]; }
  for (int k = 0; k < K; k++)
    if (rcs[k] != DVM_OK) {
      for (int t = 0; t < T; t++) if (!errs[t].empty()) { set_error("dvm_ba_optimize_batch: " + errs[t]); break; }
      return rcs[k];
    }
  return DVM_OK;
}
