// dvm_slam_amd/csrc/orb_pipeline.cpp -- host orchestration of the ORB extractor on one MI355X.
//
// Mirrors ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:47-91, src/ORBextractor.cc):
//   constructor tables      ORBextractor.cc:282-339 -> OrbPipeline::OrbPipeline
//   ComputePyramid          :957-976                -> launch_pyr_level0 / launch_pyr_resize
//   ComputeKeyPointsOctTree :612-715                -> launch_fast + launch_octree
//   operator()              :876-955                -> extract_device (stage order, output placement)
// Everything runs in HIP kernels (orb_kernels.hip, octree_kernel.hip); the host only sequences launches.
#include "orb_pipeline.h"
#include "host_stage.h"   // HostPool

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

namespace dvm {

// ------------------------------------------------------------------------------------- errors
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* last_error_cstr() { return g_err.c_str(); }
int hip_check(hipError_t e, const char* what) {
  if (e == hipSuccess) return DVM_OK;
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return DVM_ERR_HIP;
}

// ----------------------------------------------------------------------------------- profiler
hipEvent_t Profiler::get_event() {
  if (!pool_.empty()) {
    hipEvent_t e = pool_.back();
    pool_.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}
void Profiler::begin(hipStream_t s, const char* name) {
  if (!enabled) return;
  const bool live = !only_fast || std::strcmp(name, "fast") == 0;
  Pending p{name, live ? get_event() : nullptr, live ? get_event() : nullptr, s, false, live};
  if (live) hipEventRecord(p.a, s);
  pending_.push_back(p);
}
void Profiler::end(hipStream_t s) {   // closes the most recent open bracket of stream `s` (brackets of different streams nest)
  if (!enabled) return;
  for (auto it = pending_.rbegin(); it != pending_.rend(); ++it)
    if (!it->closed && it->stream == s) {
      if (it->live) hipEventRecord(it->b, s);
      it->closed = true;
      return;
    }
}
void Profiler::resolve() {
  for (auto& p : pending_) {
    float ms = 0;
    if (!p.live) continue;
    if (p.closed && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      auto& t = totals_[p.name];
      t.first += ms;
      t.second += 1;
    }
    pool_.push_back(p.a);
    pool_.push_back(p.b);
  }
  pending_.clear();
}
void Profiler::reset() {
  resolve();
  totals_.clear();
}
bool Profiler::get(const std::string& name, double* ms, int64_t* launches) {
  auto it = totals_.find(name);
  if (it == totals_.end()) return false;
  if (ms) *ms = it->second.first;
  if (launches) *launches = it->second.second;
  return true;
}
Profiler::~Profiler() {
  resolve();
  for (auto e : pool_) hipEventDestroy(e);
}

// ------------------------------------------------------------------------------- host octree
// DistributeOctTree, reference ORBextractor.cc:419-610, restated without std::list.
// Invariant used: after a round that splits the node sequence P = (p1..pm) (in processing order),
//   new list = reverse(children(p1) ++ ... ++ children(pm)) ++ (old list without P)
// because every child is push_front'ed (n1..n4 order) and the parent erased in place.  Phase 1
// processes all splittable nodes in list order; phase 2 processes them in the order given by
// std::sort(compareNodes) walked from the back, stopping as soon as the list reaches N nodes.
#ifdef DVM_DEBUG   // the host restatement of DistributeOctTree: debug / A-B builds only (make DEBUG=1); a release library has no host compute path
namespace {
struct ONode {
  int x0, y0, x1, y1;
  std::vector<int> keys;  // indices into the candidate array, parent's order preserved
};
struct OCtx {
  const uint32_t* cand;
  std::vector<ONode> nodes;
  void split(int id, int child_ids[4], int& nchild) {
    const int x0 = nodes[id].x0, y0 = nodes[id].y0, x1 = nodes[id].x1, y1 = nodes[id].y1;
    const int hx = (int)std::ceil(static_cast<float>(x1 - x0) / 2);
    const int hy = (int)std::ceil(static_cast<float>(y1 - y0) / 2);
    const int xm = x0 + hx, ym = y0 + hy;
    ONode ch[4];
    ch[0] = ONode{x0, y0, xm, ym, {}};
    ch[1] = ONode{xm, y0, x1, ym, {}};
    ch[2] = ONode{x0, ym, xm, y1, {}};
    ch[3] = ONode{xm, ym, x1, y1, {}};
    for (int k : nodes[id].keys) {
      int x, y, s;
      unpack_cand(cand[k], x, y, s);
      const int q = ((float)x < (float)xm ? 0 : 1) + ((float)y < (float)ym ? 0 : 2);
      ch[q].keys.push_back(k);
    }
    nchild = 0;
    for (int q = 0; q < 4; q++) {
      if (ch[q].keys.empty()) continue;
      child_ids[nchild++] = (int)nodes.size();
      nodes.push_back(std::move(ch[q]));
    }
  }
};
}  // namespace

void octree_select(const uint32_t* cand, int n, int minX, int maxX, int minY, int maxY, int N, std::vector<uint32_t>& out) {
  out.clear();
  if (n <= 0) return;
  OCtx cx;
  cx.cand = cand;
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  if (nIni <= 0) return;  // aspect < 0.5: the reference indexes an empty vector here (undefined)
  const float hX = static_cast<float>(maxX - minX) / nIni;
  for (int i = 0; i < nIni; i++)
    cx.nodes.push_back(ONode{(int)(hX * static_cast<float>(i)), 0, (int)(hX * static_cast<float>(i + 1)), maxY - minY, {}});
  for (int i = 0; i < n; i++) {
    int x, y, s;
    unpack_cand(cand[i], x, y, s);
    cx.nodes[(int)((float)x / hX)].keys.push_back(i);
  }
  std::vector<int> list;  // node ids in std::list order
  for (int i = 0; i < nIni; i++)
    if (!cx.nodes[i].keys.empty()) list.push_back(i);

  typedef std::pair<int, int> SizeId;  // (#keys, node id)
  auto less = [&](const SizeId& a, const SizeId& b) {
    if (a.first != b.first) return a.first < b.first;
    return cx.nodes[a.second].x0 < cx.nodes[b.second].x0;
  };
  std::vector<int> fresh, rest;
  std::vector<char> gone;
  std::vector<SizeId> expandable;
  // splits `proc` (processing order), stops early once the list would hold >= stopN nodes
  auto round = [&](const std::vector<int>& proc, int stopN) {
    fresh.clear();
    expandable.clear();
    gone.assign(cx.nodes.size() + 4 * proc.size() + 4, 0);
    int size = (int)list.size();
    for (int id : proc) {
      int ch[4], nc;
      cx.split(id, ch, nc);
      for (int k = 0; k < nc; k++) {
        fresh.push_back(ch[k]);
        if (cx.nodes[ch[k]].keys.size() > 1) expandable.push_back(SizeId((int)cx.nodes[ch[k]].keys.size(), ch[k]));
      }
      gone[id] = 1;
      size += nc - 1;
      if (stopN >= 0 && size >= stopN) break;
    }
    rest.clear();
    for (int id : list)
      if (!gone[id]) rest.push_back(id);
    list.assign(fresh.rbegin(), fresh.rend());
    list.insert(list.end(), rest.begin(), rest.end());
  };

  bool finish = false;
  std::vector<int> proc;
  while (!finish) {
    const int prev = (int)list.size();
    proc.clear();
    for (int id : list)
      if (cx.nodes[id].keys.size() > 1) proc.push_back(id);
    round(proc, -1);
    const int nToExpand = (int)expandable.size();
    if ((int)list.size() >= N || (int)list.size() == prev) {
      finish = true;
    } else if ((int)list.size() + nToExpand * 3 > N) {
      while (!finish) {
        const int prev2 = (int)list.size();
        std::vector<SizeId> order = expandable;
        std::sort(order.begin(), order.end(), less);
        proc.clear();
        for (int j = (int)order.size() - 1; j >= 0; j--) proc.push_back(order[j].second);
        round(proc, N);
        if ((int)list.size() >= N || (int)list.size() == prev2) finish = true;
      }
    }
  }
  out.reserve(list.size());
  for (int id : list) {
    const std::vector<int>& keys = cx.nodes[id].keys;
    int best = keys[0];
    int bs = (int)(cand[best] >> 24);
    for (size_t k = 1; k < keys.size(); k++) {
      int s = (int)(cand[keys[k]] >> 24);
      if (s > bs) { bs = s; best = keys[k]; }
    }
    out.push_back(cand[best]);
  }
}
#endif  // DVM_DEBUG


// -------------------------------------------------------------------------------- OrbPipeline
static inline int cv_round_f(float v) { return (int)std::nearbyintf(v); }
static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

OrbPipeline::OrbPipeline(const dvm_orb_params& p, int dev, int mb) : params(p), device(dev), max_batch(std::max(1, mb)) {
  const int L = params.nlevels;
  const double sf = (double)params.scale_factor;  // the reference stores scaleFactor as double (ORBextractor.h:84)
  scale.assign(L, 1.f); inv_scale.assign(L, 1.f); sigma2.assign(L, 1.f); inv_sigma2.assign(L, 1.f); nfeat.assign(L, 0);
  for (int i = 1; i < L; i++) {
    scale[i] = (float)(scale[i - 1] * sf);
    sigma2[i] = scale[i] * scale[i];
  }
  for (int i = 0; i < L; i++) {
    inv_scale[i] = 1.0f / scale[i];
    inv_sigma2[i] = 1.0f / sigma2[i];
  }
  const float factor = (float)(1.0f / sf);
  float per_scale = params.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)L));
  int sum = 0;
  for (int l = 0; l < L - 1; l++) {
    nfeat[l] = cv_round_f(per_scale);
    sum += nfeat[l];
    per_scale *= factor;
  }
  nfeat[L - 1] = std::max(params.nfeatures - sum, 0);
  // umax, :320-338
  const int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
  const int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
  std::fill(umax, umax + 16, 0);
  for (int v = 0; v <= vmax; ++v) umax[v] = (int)std::nearbyint(std::sqrt((double)kHalfPatch * kHalfPatch - v * v));
  for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

OrbPipeline::~OrbPipeline() {
  if (stream) hipStreamSynchronize(stream);
  free_all();
  if (copy_stream) { hipStreamSynchronize(copy_stream); hipStreamDestroy(copy_stream); }
  if (ev_copied) hipEventDestroy(ev_copied);
  if (ev_stage_free) hipEventDestroy(ev_stage_free);
  if (d_stage) hipFree(d_stage);
  if (h_stage) hipHostFree(h_stage);
  for (hipStream_t st : {lane_main[1], lane_side[0], lane_side[1], lat_aux})
    if (st) { hipStreamSynchronize(st); hipStreamDestroy(st); }
  for (int c = 0; c < kMaxChunks; c++)
    for (hipEvent_t e : {ev_compact[c], ev_fork[c], ev_join[c]})
      if (e) hipEventDestroy(e);
  for (hipEvent_t e : ev_group) if (e) hipEventDestroy(e);
  if (ev_start) hipEventDestroy(ev_start);
  if (ev_done) hipEventDestroy(ev_done);
  if (stream) hipStreamDestroy(stream);
}

int OrbPipeline::init() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device visible (libdvmslam_hip has no CPU path)");
    return DVM_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) {
    set_error("device index out of range");
    return DVM_ERR_INVALID;
  }
  DVM_HIP(hipSetDevice(device));
  DVM_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  lane_main[0] = stream;
  if (const char* e = getenv("DVM_BLUR_EARLY")) blur_early = (e[0] == '1');
  if (const char* e = getenv("DVM_GROUPS")) sscanf(e, "%d,%d", &group_split[0], &group_split[1]);
  if (const char* e = getenv("DVM_CHUNKS")) chunks = std::min(std::max(atoi(e), 1), (int)kMaxChunks);
  {  // side stream at the LOWEST priority: its kernels (the blur) only fill what the main chain leaves idle.
    // The runtime multiplexes streams onto 4 hardware queues per device: a handle owns TWO streams (main + side), so that
    // two handles -- ping-pong ingest, dvm_orb_extract_staged -- still get a queue per stream; the second pair exists only
    // for the opt-in chunk / level-group pipelines (a fifth stream in the process made every stage ~2x slower).
    int least = 0, greatest = 0;
    DVM_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    side_priority = least;   // lane_side[0] is created by the first call that forks the blur (batches of more than kLatencyBatch frames):
                             // a handle that only ever sees single frames keeps ONE stream -- K agents on a GPU are K + K staging streams, not 3 K
    if (chunks > 1 || group_split[0] < kMaxLevels) {
      DVM_HIP(hipStreamCreateWithPriority(&lane_side[1], hipStreamNonBlocking, least));
      DVM_HIP(hipStreamCreateWithFlags(&lane_main[1], hipStreamNonBlocking));
    }
    (void)greatest;
  }
  DVM_HIP(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
  DVM_HIP(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
  for (int g = 0; g < 4; g++) DVM_HIP(hipEventCreateWithFlags(&ev_group[g], hipEventDisableTiming));
  for (int c = 0; c < kMaxChunks; c++) {
    DVM_HIP(hipEventCreateWithFlags(&ev_compact[c], hipEventDisableTiming));
    DVM_HIP(hipEventCreateWithFlags(&ev_fork[c], hipEventDisableTiming));
    DVM_HIP(hipEventCreateWithFlags(&ev_join[c], hipEventDisableTiming));
  }
  if (const char* e = getenv("DVM_LATENCY_PATH")) latency_path = (e[0] != '0');
  if (const char* e = getenv("DVM_OCT_BLUR")) oct_blur = (e[0] != '0');
  if (const char* e = getenv("DVM_LAT_SPLIT")) lat_split = (e[0] != '0');
  if (const char* e = getenv("DVM_ZERO_COPY_IN")) zero_copy_in = (e[0] != '0');
  if (const char* e = getenv("DVM_SERIAL")) overlap_blur = (e[0] != '1');      // debug / A-B switch only
#ifdef DVM_DEBUG
  if (const char* e = getenv("DVM_HOST_OCTREE")) host_octree_forced = (e[0] == '1');  // debug / A-B switch, debug builds only
#endif
  host_octree = host_octree_forced;
  // orientation disc offsets (any order: the moments are exact integer sums)
  int8_t du[kDiscPixels], dv[kDiscPixels];
  int n = 0;
  for (int v = -kHalfPatch; v <= kHalfPatch; v++) {
    const int d = umax[std::abs(v)];
    for (int u = -d; u <= d; u++) { du[n] = (int8_t)u; dv[n] = (int8_t)v; n++; }
  }
  if (n != kDiscPixels) { set_error("disc size mismatch"); return DVM_ERR_STATE; }
  // 8.8 fixed-point Gaussian kernel, ksize 7, sigma 2 (OpenCV getGaussianKernelFixedPoint_ED)
  int g7[7];
  {
    double k[7], s = 0;
    for (int i = 0; i < 7; i++) { double x = i - 3.0; k[i] = std::exp(-0.5 / 4.0 * x * x); s += k[i]; }
    double err = 0; long long acc = 0;
    for (int i = 0; i < 3; i++) {
      double adj = k[i] * (1.0 / s) * 256.0 + err;
      long long v = (long long)std::nearbyint(adj);
      err = adj - (double)v;
      g7[i] = g7[6 - i] = (int)v;
      acc += v;
    }
    g7[3] = (int)(256 - 2 * acc);
  }
  for (int i = 0; i < 7; i++) gauss7[i] = g7[i];
  upload_constants(du, dv, g7);
  DVM_HIP(hipGetLastError());
  DVM_HIP(hipDeviceSynchronize());
  return DVM_OK;
}

void OrbPipeline::free_all() {
  void* dptrs[] = {d_pyr, d_blur, d_tabs, d_cells, d_tiles, d_cand, d_dense, d_cell_count, d_lvl_count, d_sel, d_nsel,
                   d_kps, d_desc, d_aux, d_n, d_mono, d_nid};
  for (void* p : dptrs) if (p) hipFree(p);
  void* hptrs[] = {h_cell_count, h_dense, h_sel, h_nsel, h_n, h_mono, (void*)h_err, h_kps_m, h_desc_m, h_kps_b, h_desc_b};
  for (void* p : hptrs) if (p) hipHostFree(p);
  d_pyr = d_blur = d_desc = nullptr; d_tabs = nullptr; d_cells = nullptr; d_tiles = nullptr;
  d_cand = d_dense = d_sel = nullptr; d_cell_count = d_lvl_count = d_nsel = d_n = d_mono = nullptr;
  d_kps = nullptr; d_aux = nullptr; d_nid = nullptr; d_err = nullptr; h_err = nullptr;
  h_cell_count = h_nsel = h_n = h_mono = nullptr; h_dense = h_sel = nullptr;
  h_kps_m = nullptr; h_desc_m = nullptr; last_mirrored = false;
  h_kps_b = nullptr; h_desc_b = nullptr; h_batch_cap = 0;
  configured = false;  // (the host-image staging buffers d_stage/h_stage live until the destructor)
}

// cv::resize INTER_LINEAR coefficient tables for one axis (OpenCV resize.cpp, 11-bit weights)
static void resize_axis_tables(int ssize, int dsize, bool clamp_like_x, std::vector<int32_t>& ofs, std::vector<int32_t>& wts) {
  const double inv_scale = (double)dsize / ssize;
  const double sc = 1. / inv_scale;
  ofs.resize(dsize);
  wts.resize(dsize);
  for (int d = 0; d < dsize; d++) {
    float f = (float)((d + 0.5) * sc - 0.5);
    int s = (int)std::floor(f);
    f -= s;
    if (clamp_like_x) {  // x axis: offsets are clamped and the fraction zeroed at the borders
      if (s < 0) { f = 0; s = 0; }
      if (s >= ssize - 1) { f = 0; s = ssize - 1; }
    }
    auto sat = [](float v) { int i = (int)std::nearbyintf(v); return std::min(std::max(i, -32768), 32767); };
    const int w0 = sat((1.f - f) * 2048), w1 = sat(f * 2048);
    ofs[d] = s;
    wts[d] = (int32_t)((uint32_t)(uint16_t)(int16_t)w0 | ((uint32_t)(uint16_t)(int16_t)w1 << 16));
  }
}

int OrbPipeline::configure(int rows, int cols) {
  if (configured && PD.rows == rows && PD.cols == cols) return DVM_OK;
  if (stream) hipStreamSynchronize(stream);
  free_all();
  const int L = params.nlevels;
  if (L < 1 || L > kMaxLevels) { set_error("nlevels out of range"); return DVM_ERR_INVALID; }
  if (cols + 2 * kEdge > 4095 || rows + 2 * kEdge > 4095) { set_error("image larger than 4057 px unsupported"); return DVM_ERR_INVALID; }
  std::memset(&PD, 0, sizeof(PD));
  PD.nlevels = L; PD.rows = rows; PD.cols = cols;
  PD.ini_th = std::min(std::max(params.ini_th_fast, 0), 255);
  PD.min_th = std::min(std::max(params.min_th_fast, 0), 255);
  cells.clear(); tiles.clear();
  std::vector<int32_t> tabs;
  int pyr_off = 0, blur_off = 0, cand_off = 0, sel_off = 0;
  for (int l = 0; l < L; l++) {
    LevelDesc& D = PD.lv[l];
    D.w = cv_round_f((float)cols * inv_scale[l]);   // ORBextractor.cc:959-960
    D.h = cv_round_f((float)rows * inv_scale[l]);
    if (D.w < 1 || D.h < 1) { set_error("pyramid level collapses to zero size"); return DVM_ERR_INVALID; }
    D.stride = align_up(D.w + 2 * kEdge, 64);
    D.pyr_off = pyr_off;
    pyr_off += align_up(D.stride * (D.h + 2 * kEdge), 256);
    D.blur_stride = align_up(D.w, 64);
    D.blur_off = blur_off;
    blur_off += align_up(D.blur_stride * D.h, 256);
    D.scale = scale[l];
    D.patch_size = (int)(31 * scale[l]);  // :700
    D.quota = nfeat[l];
    D.sel_off = sel_off;
    // a level keeps up to max(quota + 2, 4 * nIni) keypoints: the first octree sweep splits every root node before the
    // node count is compared with the quota (ORBextractor.cc:460-536), which matters for small quotas on wide images
    D.sel_cap = std::max(std::max(nfeat[l], 1), 4 * std::max(octree_root_nodes(D), 0)) + 4;
    sel_off += D.sel_cap;
    if (l > 0) {
      std::vector<int32_t> xo, xa, yo, yb;
      resize_axis_tables(PD.lv[l - 1].w, D.w, true, xo, xa);
      resize_axis_tables(PD.lv[l - 1].h, D.h, false, yo, yb);
      D.tab_off = (int)tabs.size();
      tabs.insert(tabs.end(), xo.begin(), xo.end());
      tabs.insert(tabs.end(), xa.begin(), xa.end());
      tabs.insert(tabs.end(), yo.begin(), yo.end());
      tabs.insert(tabs.end(), yb.begin(), yb.end());
    }
    // FAST cells, ORBextractor.cc:617-650
    D.cell_first = (int)cells.size();
    cand_off = align_up(cand_off, 32);   // a level's range starts on its own 128-byte line (k_octree writes and re-reads it)
    D.cand_off = cand_off;
    const int minBX = kEdge - 3, minBY = minBX, maxBX = D.w - kEdge + 3, maxBY = D.h - kEdge + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const float W = 35;
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols > 0 && nRows > 0) {
      const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
      for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = (float)maxBY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = (float)(minBX + j * wCell);
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBX - 6) continue;
          if (maxX > maxBX) maxX = (float)maxBX;
          CellDesc c{};
          c.level = (int16_t)l;
          c.x0 = (int16_t)iniX; c.y0 = (int16_t)iniY;
          c.rw = (int16_t)((int)maxX - (int)iniX); c.rh = (int16_t)((int)maxY - (int)iniY);
          if (c.rw > kMaxCellDim || c.rh > kMaxCellDim) { set_error("FAST cell larger than supported"); return DVM_ERR_INVALID; }
          const int ew = std::max(c.rw - 6, 0), eh = std::max(c.rh - 6, 0);
          c.cand_cap = ((ew + 1) / 2) * ((eh + 1) / 2);  // strict local maxima cannot be 8-adjacent
          c.cand_base = cand_off;
          cand_off += c.cand_cap;
          cells.push_back(c);
        }
      }
    }
    D.cell_count = (int)cells.size() - D.cell_first;
    D.cand_cap = cand_off - D.cand_off;
    for (int y = 0; y < D.h; y += kBlurTH)
      for (int x = 0; x < D.w; x += kBlurTW) tiles.push_back(TileDesc{(int16_t)l, (int16_t)x, (int16_t)y, 0});
  }
  max_cell_rw = max_cell_rh = 8;
  for (const CellDesc& c : cells) { max_cell_rw = std::max<int>(max_cell_rw, c.rw); max_cell_rh = std::max<int>(max_cell_rh, c.rh); }
  // DistributeOctTree starts from nIni = round((cols - 32) / (rows - 32)) root nodes per level (ORBextractor.cc:423);
  // for a portrait level nIni = 0 and the reference divides by it: reject such sizes instead of guessing
  for (int l = 0; l < PD.nlevels; l++)
    if (PD.lv[l].w > 2 * (kEdge - 3) && PD.lv[l].h > 2 * (kEdge - 3) && octree_root_nodes(PD.lv[l]) < 1) {   // (levels without a FAST area have no candidates)
      set_error("image too narrow for DistributeOctTree: round((cols-32)/(rows-32)) = 0 at some pyramid level");
      return DVM_ERR_INVALID;
    }
  // A level quota beyond the device octree's node capacity (2 680 keypoints on one level, i.e. ~12 300 features at the
  // usual 1.2 / 8 levels) is refused: there is no CPU path.  (A -DDVM_DEBUG build has DVM_HOST_OCTREE=1, the same algorithm on
  // the host, as an A-B switch.)
  host_octree = host_octree_forced;
  if (!host_octree && !octree_prepare_device(PD)) {
    set_error("a pyramid level's keypoint quota exceeds the device octree capacity (2680 nodes); lower nFeatures");
    return DVM_ERR_CAPACITY;
  }
  tiny_levels = false;
  for (int l = 0; l < PD.nlevels; l++) tiny_levels |= (PD.lv[l].w < 40 || PD.lv[l].h < 20);
  PD.ncells = (int)cells.size();
  PD.ntiles = (int)tiles.size();
  PD.pyr_frame_bytes = pyr_off;
  PD.blur_frame_bytes = blur_off;
  PD.cand_frame_slots = align_up(std::max(cand_off, 1), 32);
  PD.sel_frame_slots = sel_off;
  PD.kp_cap = sel_off;
  const size_t B = (size_t)max_batch;
  DVM_HIP(hipMalloc(&d_pyr, B * PD.pyr_frame_bytes + 256));   // + slack: 16-byte tile loads may run past the last ROI
  DVM_HIP(hipMalloc(&d_blur, B * PD.blur_frame_bytes));
  DVM_HIP(hipMalloc(&d_tabs, std::max<size_t>(tabs.size(), 1) * 4));
  DVM_HIP(hipMalloc(&d_cells, std::max<size_t>(cells.size(), 1) * sizeof(CellDesc)));
  DVM_HIP(hipMalloc(&d_tiles, std::max<size_t>(tiles.size(), 1) * sizeof(TileDesc)));
  DVM_HIP(hipMalloc(&d_cand, B * PD.cand_frame_slots * 4));
  DVM_HIP(hipMalloc(&d_dense, B * PD.cand_frame_slots * 4));
  DVM_HIP(hipMalloc(&d_cell_count, B * std::max(PD.ncells, 1) * 4));
  DVM_HIP(hipMalloc(&d_lvl_count, B * kMaxLevels * 4));
  DVM_HIP(hipMalloc(&d_sel, B * PD.sel_frame_slots * 4));
  DVM_HIP(hipMalloc(&d_nsel, B * L * 4));
  DVM_HIP(hipMalloc(&d_kps, B * PD.kp_cap * sizeof(dvm_keypoint_pod)));
  DVM_HIP(hipMalloc(&d_desc, B * PD.kp_cap * 32));
  DVM_HIP(hipMalloc(&d_aux, B * PD.kp_cap * sizeof(KpAux)));
  DVM_HIP(hipMalloc(&d_n, B * 4));
  DVM_HIP(hipMalloc(&d_mono, B * 4));
  DVM_HIP(hipMalloc(&d_nid, B * PD.cand_frame_slots * 4));
  // the octree's overflow flag lives in mapped host memory: the kernel stores to it (system scope), sync() reads it after
  // the stream has drained -- no copy, no launch
  DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_err), 4, hipHostMallocMapped));
  *h_err = 0;
  DVM_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_err), h_err, 0));
  DVM_HIP(hipHostMalloc(&h_cell_count, B * std::max(PD.ncells, 1) * 4));
  DVM_HIP(hipHostMalloc(&h_dense, B * PD.cand_frame_slots * 4));
  DVM_HIP(hipHostMalloc(&h_sel, B * PD.sel_frame_slots * 4));
  DVM_HIP(hipHostMalloc(&h_nsel, B * L * 4));
  DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_n), B * 4, hipHostMallocMapped));
  DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_mono), B * 4, hipHostMallocMapped));
  {
    const size_t mb = std::min<size_t>(B, kLatencyBatch);
    DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_kps_m), mb * PD.kp_cap * sizeof(dvm_keypoint_pod), hipHostMallocMapped));
    DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_desc_m), mb * PD.kp_cap * 32, hipHostMallocMapped));
    DVM_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&mirror_dev.kps), h_kps_m, 0));
    DVM_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&mirror_dev.desc), h_desc_m, 0));
    DVM_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&mirror_dev.n), h_n, 0));
    DVM_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&mirror_dev.mono), h_mono, 0));
  }
  if (!tabs.empty()) DVM_HIP(hipMemcpy(d_tabs, tabs.data(), tabs.size() * 4, hipMemcpyHostToDevice));
  if (!cells.empty()) DVM_HIP(hipMemcpy(d_cells, cells.data(), cells.size() * sizeof(CellDesc), hipMemcpyHostToDevice));
  if (!tiles.empty()) DVM_HIP(hipMemcpy(d_tiles, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
  // the blurred images are read up to 18 px around a keypoint only; clear once so debug dumps are defined
  DVM_HIP(hipMemset(d_blur, 0, B * PD.blur_frame_bytes));
  DVM_HIP(hipMemset(d_pyr, 0, B * PD.pyr_frame_bytes));
  // the handle's stream is non-blocking: null-stream memsets/copies above must land before it runs
  DVM_HIP(hipDeviceSynchronize());
  configured = true;
  return DVM_OK;
}

int OrbPipeline::ensure_stage(size_t need) {
  if (need > stage_bytes) {
    if (copy_stream) hipStreamSynchronize(copy_stream);
    if (stream) hipStreamSynchronize(stream);
    copy_pending = false; stage_free_valid = false;
    if (d_stage) hipFree(d_stage);
    if (h_stage) hipHostFree(h_stage);
    d_stage = nullptr; h_stage = nullptr; stage_bytes = 0; stage_view = nullptr;
    DVM_HIP(hipMalloc(&d_stage, need));
    DVM_HIP(hipHostMalloc(&h_stage, need));
    stage_bytes = need;
  }
  return DVM_OK;
}

int OrbPipeline::extract_host(const uint8_t* imgs, int batch, int rows, int cols, int stride, int64_t frame_stride,
                              int lap0, int lap1) {
  if (!imgs || rows <= 0 || cols <= 0) return DVM_ERR_EMPTY;
  if (batch < 1 || batch > max_batch || stride < cols) { set_error("bad batch/stride"); return DVM_ERR_INVALID; }
  DVM_HIP(hipSetDevice(device));
  const size_t need = (size_t)batch * rows * cols;
  const int rc = ensure_stage(need);
  if (rc != DVM_OK) return rc;
  if (copy_stream) DVM_HIP(hipStreamSynchronize(copy_stream));
  DVM_HIP(hipStreamSynchronize(stream));  // staging buffer reuse
  copy_pending = false;
  // (a batch of frames by a few pooled threads: 32 VGA frames are 9.8 MB -- 0.3 ms of a 1.4 ms tracking tick on one thread)
  HostPool::get().run((size_t)batch, batch >= 4 ? 8 : 1, [&](size_t f) {
    if (stride == cols) { std::memcpy(h_stage + f * (size_t)rows * cols, imgs + f * (size_t)frame_stride, (size_t)rows * cols); return; }
    for (int y = 0; y < rows; y++)
      std::memcpy(h_stage + (f * (size_t)rows + y) * cols, imgs + f * (size_t)frame_stride + (size_t)y * stride, cols);
  });
  if (latency_path && zero_copy_in && batch <= kLatencyBatch) {   // level 0 reads the pinned buffer itself (see orb_pipeline.h)
    if (!stage_view) DVM_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&stage_view), h_stage, 0));
    return extract_device(stage_view, batch, rows, cols, cols, (int64_t)rows * cols, lap0, lap1);
  }
  DVM_HIP(hipMemcpyAsync(d_stage, h_stage, need, hipMemcpyHostToDevice, stream));
  return extract_device(d_stage, batch, rows, cols, cols, (int64_t)rows * cols, lap0, lap1);
}

// Zero-copy ingest for a caller that can write its frames where the DMA engine reads them: staging() hands out the pinned
// buffer (batch x rows x cols, tight), extract_staged() queues the H2D copy and the extraction on the handle's stream
// and returns.  Two handles used alternately overlap one batch's PCIe transfer with the other's kernels.
int OrbPipeline::staging(int batch, int rows, int cols, uint8_t** host_ptr) {
  if (!host_ptr || batch < 1 || batch > max_batch || rows <= 0 || cols <= 0) return DVM_ERR_INVALID;
  DVM_HIP(hipSetDevice(device));
  const int rc = ensure_stage((size_t)batch * rows * cols);
  if (rc != DVM_OK) return rc;
  if (copy_pending) {   // only the previous copy OUT of the pinned buffer has to be through, not the batch's kernels
    DVM_HIP(hipEventSynchronize(ev_copied));
    copy_pending = false;
  }
  *host_ptr = h_stage;
  return DVM_OK;
}

// The copy runs on its own stream: behind an in-order copy -> kernel chain on ONE stream the runtime did not overlap the
// transfer with other work (measured: 2.7 ms per 256-frame step with two handles used alternately against 1.6 ms this
// way).  host: wait until the previous batch has read d_stage (its level-0 kernel); copy stream: H2D -> event; main
// stream: wait for the event -> extraction.  With two handles used alternately a batch crosses PCIe while the other
// handle's batch is in FAST / octree / ... (1.53 ms per 256-frame step against 1.30 ms with the frames resident in HBM).
int OrbPipeline::extract_staged(int batch, int rows, int cols, int lap0, int lap1) {
  if (batch < 1 || batch > max_batch || rows <= 0 || cols <= 0) return DVM_ERR_INVALID;
  const size_t need = (size_t)batch * rows * cols;
  if (!h_stage || need > stage_bytes) { set_error("extract_staged: call dvm_orb_staging for this size first"); return DVM_ERR_STATE; }
  DVM_HIP(hipSetDevice(device));
  if (!copy_stream) {
    DVM_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    DVM_HIP(hipEventCreateWithFlags(&ev_copied, hipEventDisableTiming));
    DVM_HIP(hipEventCreateWithFlags(&ev_stage_free, hipEventDisableTiming));
  }
  // d_stage may be overwritten once the previous batch's level-0 kernel has read it.  Waited for on the HOST: a device-side
  // hipStreamWaitEvent in front of the copy cost 0.4 ms per step when measured (1.94 vs 1.53 ms, two handles).
  if (stage_free_valid) DVM_HIP(hipEventSynchronize(ev_stage_free));
  DVM_HIP(hipMemcpyAsync(d_stage, h_stage, need, hipMemcpyHostToDevice, copy_stream));
  DVM_HIP(hipEventRecord(ev_copied, copy_stream));
  copy_pending = true;
  DVM_HIP(hipStreamWaitEvent(stream, ev_copied, 0));
  return extract_device(d_stage, batch, rows, cols, cols, (int64_t)rows * cols, lap0, lap1);
}

// The frames in the pinned buffer of staging(), read IN PLACE when they are few (the latency path's zero-copy ingest: no H2D copy queued);
// otherwise extract_staged().  For callers that synchronise before they touch the buffer again (the shared extractor of orb_pool.cpp).
int OrbPipeline::extract_staged_sync_owner(int batch, int rows, int cols, int lap0, int lap1) {
  if (!(latency_path && zero_copy_in && batch <= kLatencyBatch)) return extract_staged(batch, rows, cols, lap0, lap1);
  if (batch < 1 || rows <= 0 || cols <= 0) return DVM_ERR_INVALID;
  if (!h_stage || (size_t)batch * rows * cols > stage_bytes) { set_error("extract_staged: call dvm_orb_staging for this size first"); return DVM_ERR_STATE; }
  DVM_HIP(hipSetDevice(device));
  if (!stage_view) DVM_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&stage_view), h_stage, 0));
  return extract_device(stage_view, batch, rows, cols, cols, (int64_t)rows * cols, lap0, lap1);
}

int OrbPipeline::extract_device(const uint8_t* d_imgs, int batch, int rows, int cols, int stride, int64_t frame_stride,
                                int lap0, int lap1) {
  if (!d_imgs || rows <= 0 || cols <= 0) return DVM_ERR_EMPTY;
  if (batch < 1 || batch > max_batch || stride < cols) { set_error("bad batch/stride"); return DVM_ERR_INVALID; }
  DVM_HIP(hipSetDevice(device));
  int rc = configure(rows, cols);
  if (rc != DVM_OK) return rc;
  const int L = PD.nlevels;
  last_batch = batch;
  const bool small = latency_path && batch <= kLatencyBatch;
  HostMirror hm;
  last_mirrored = small;
  if (small) {
    hm = mirror_dev;   // (device addresses of the mapped result buffers, looked up once per configuration)
  }
  const bool blur_first = blur_early && !small;
  const bool split0 = small && lat_split && overlap_blur && !host_octree && !tiny_levels && L > 1 && chunks == 1 && group_split[0] >= L;
  if (split0 && !lat_aux) DVM_HIP(hipStreamCreateWithFlags(&lat_aux, hipStreamNonBlocking));

  // Per chunk: one in-order chain, except the blur: it depends only on the pyramid and on the per-level candidate
  // counts, so it runs on the lane's side stream concurrently with the latency-bound k_octree (one workgroup per
  // level whose critical path is a single-lane std::sort emulation) and joins before the descriptors.
  // DVM_SERIAL=1 keeps the blur on the main chain, DVM_CHUNKS=1 disables the chunk pipeline.
  int nck = host_octree ? 1 : std::min(chunks, std::max(1, batch / 32));   // chunks of >= 32 frames
  auto run_half = [&](hipStream_t st, hipStream_t side, int ck, int f0, int nb) -> int {
    uint8_t* pyr_f0 = d_pyr + (size_t)f0 * PD.pyr_frame_bytes;
    uint32_t* cand_f0 = d_cand + (size_t)f0 * PD.cand_frame_slots;
    int32_t* cnt_f0 = d_cell_count + (size_t)f0 * PD.ncells;
    // (tried: FAST of level 0 forked onto the side stream right after k_pyr_level0 so that it overlaps the seven resize
    // launches -- the resizes are VALU-bound too and slow down by as much as is gained: 1.740 -> 1.717 ms per step,
    // at the price of two k_fast_cells launches per batch; not kept.  launch_fast still takes a cell range.)
    prof.begin(st, "pyramid");
    launch_pyr_level0(st, d_imgs + (int64_t)f0 * frame_stride, rows, cols, stride, frame_stride, pyr_f0, PD, nb);
    uint32_t* dense_f0 = d_dense + (size_t)f0 * PD.cand_frame_slots;
    int32_t* lcnt_f0 = d_lvl_count + (size_t)f0 * kMaxLevels;
    int32_t* nid_f0 = d_nid + (size_t)f0 * PD.cand_frame_slots;
    uint32_t* sel_f0 = d_sel + (size_t)f0 * PD.sel_frame_slots;
    int32_t* nsel_f0 = d_nsel + (size_t)f0 * L;
    if (split0) {   // level 0: FAST cells + octree on the second stream, beside everything below up to k_assemble
      DVM_HIP(hipEventRecord(ev_group[0], st));
      DVM_HIP(hipStreamWaitEvent(lat_aux, ev_group[0], 0));
      launch_fast(lat_aux, pyr_f0, d_cells, PD, cand_f0, cnt_f0, nb, max_cell_rw, max_cell_rh, PD.lv[0].cell_first, PD.lv[0].cell_count, cells.data());
      launch_octree(lat_aux, cand_f0, cnt_f0, d_cells, dense_f0, lcnt_f0, PD, nid_f0, sel_f0, nsel_f0, d_err, nb, 0, 1, true);
      DVM_HIP(hipEventRecord(ev_group[3], lat_aux));
    }
    if (d_imgs == d_stage && ev_stage_free && f0 + nb >= batch) {   // the staged input has been consumed: the next batch's
      DVM_HIP(hipEventRecord(ev_stage_free, st));                   // H2D copy may overwrite it while this one computes
      stage_free_valid = true;
    }
    for (int l = 1; l < L; l++) launch_pyr_resize(st, pyr_f0, PD, l, d_tabs, nb);
    if (tiny_levels) launch_pyr_borders(st, pyr_f0, PD, nb);   // else: fused into the level kernels
    prof.end(st);
    const bool blur_forked = overlap_blur && !host_octree && side != nullptr && !small;   // latency path: one chain, no events (below)
    if (blur_forked && blur_first) {   // blur needs the pyramid only: low-priority side stream, fills the idle slots of
      DVM_HIP(hipEventRecord(ev_fork[ck], st));               // FAST's tail, the small scan kernels and the octree
      DVM_HIP(hipStreamWaitEvent(side, ev_fork[ck], 0));
      prof.begin(side, "blur");
      launch_blur(side, pyr_f0, (d_blur + (size_t)f0 * PD.blur_frame_bytes), d_tiles, PD, nullptr, nb);
      prof.end(side);
      DVM_HIP(hipEventRecord(ev_join[ck], side));
    }
    // FAST and the octree, pipelined over level groups: a level's octree (a latency chain of small scans and one
    // wavefront sort) needs that level's FAST cells only, so it runs on the auxiliary stream while the main stream is
    // already on the FAST cells of the next group; only the last group's octree is left after FAST has finished.
    //   main: FAST(G0) | FAST(G1) | FAST(G2) octree(G2) ... wait aux
    //   aux :            octree(G0) | octree(G1)
    hipStream_t aux = lane_main[st == lane_main[0] ? 1 : 0];
    const bool grouped = overlap_blur && !host_octree && chunks == 1 && aux != nullptr && group_split[0] < L;
    const int g_a = std::min(std::max(group_split[0], 0), L), g_b = std::min(std::max(group_split[1], g_a), L);
    const int gl[4] = {split0 ? 1 : 0, grouped ? g_a : L, grouped ? g_b : L, L};   // level groups [gl[i], gl[i+1])
    int oct_main_first = split0 ? 1 : 0;   // levels below this one have their octree on the auxiliary stream
    bool blur_done = false;
    prof.begin(st, "fast");   // one bracket over the (up to three) k_fast_cells launches of the batch
    for (int gi = 0; gi < 3; gi++) {
      const int la = gl[gi], lb = gl[gi + 1];
      if (la >= lb) continue;
      const int c_first = PD.lv[la].cell_first, c_end = PD.lv[lb - 1].cell_first + PD.lv[lb - 1].cell_count;
      launch_fast(st, pyr_f0, d_cells, PD, cand_f0, cnt_f0, nb, max_cell_rw, max_cell_rh, c_first, c_end - c_first, cells.data());
      if (grouped && lb < L) {
        DVM_HIP(hipEventRecord(ev_group[gi], st));
        DVM_HIP(hipStreamWaitEvent(aux, ev_group[gi], 0));
        launch_octree(aux, cand_f0, cnt_f0, d_cells, dense_f0, lcnt_f0, PD, nid_f0, sel_f0, nsel_f0, d_err, nb, la, lb - la, small);
        oct_main_first = lb;
      }
    }
    prof.end(st);
    if (nck > 1) DVM_HIP(hipEventRecord(ev_compact[ck], st));   // this chunk has left the throughput-bound stages
    // blur after FAST: only the event goes between FAST and the octree; the side stream's calls come after the octree's launch, so
    // the host is not still talking to the side stream while the main chain waits for its next kernel (latency path)
    auto fork_blur = [&]() -> int {
      DVM_HIP(hipStreamWaitEvent(side, ev_fork[ck], 0));
      prof.begin(side, "blur");
      launch_blur(side, (d_pyr + (size_t)f0 * PD.pyr_frame_bytes), (d_blur + (size_t)f0 * PD.blur_frame_bytes), d_tiles, PD, nullptr, nb);
      prof.end(side);
      DVM_HIP(hipEventRecord(ev_join[ck], side));
      return DVM_OK;
    };
    const bool blur_late = blur_forked && !blur_first;
    if (blur_late) {
      DVM_HIP(hipEventRecord(ev_fork[ck], st));
      if (host_octree) { const int rcb = fork_blur(); if (rcb != DVM_OK) return rcb; }
    }
    if (!host_octree) {
      prof.begin(st, "octree");   // the part of the octree work that is NOT hidden behind FAST
      const int la = oct_main_first;
      if (small && oct_blur && la == 0 && !blur_forked && octree_blur_fits(PD)) {   // one-frame path: octrees + blur tiles in one launch
        launch_octree_blur(st, cand_f0, cnt_f0, d_cells, dense_f0, lcnt_f0, PD, nid_f0, sel_f0, nsel_f0, d_err, nb, pyr_f0,
                           d_blur + (size_t)f0 * PD.blur_frame_bytes, d_tiles, gauss7);
        blur_done = true;
      } else {
        launch_octree(st, cand_f0, cnt_f0, d_cells, dense_f0, lcnt_f0, PD, nid_f0, sel_f0, nsel_f0, d_err, nb, la, L - la, small);
      }
      if (blur_late) { const int rcb = fork_blur(); if (rcb != DVM_OK) return rcb; }
      if (split0) {
        DVM_HIP(hipStreamWaitEvent(st, ev_group[3], 0));
      } else if (oct_main_first > 0) {
        DVM_HIP(hipEventRecord(ev_group[3], aux));
        DVM_HIP(hipStreamWaitEvent(st, ev_group[3], 0));
      }
      prof.end(st);
    }
#ifdef DVM_DEBUG
    else {
    // ---- DistributeOctTree on the host (K3): per-cell candidate lists -> vToDistributeKeys per level -> octree_select
      DVM_HIP(hipMemcpyAsync(h_cell_count, d_cell_count, (size_t)batch * PD.ncells * 4, hipMemcpyDeviceToHost, st));
      DVM_HIP(hipMemcpyAsync(h_dense, d_cand, (size_t)batch * PD.cand_frame_slots * 4, hipMemcpyDeviceToHost, st));
      DVM_HIP(hipStreamSynchronize(st));
      {
        const int jobs = batch * L;
        const int nthreads = std::max(1, std::min<int>((int)std::thread::hardware_concurrency(), std::min(jobs / 4, 16)));
        auto work = [&](int t) {
          std::vector<uint32_t> out;
          for (int j = t; j < jobs; j += nthreads) {
            const int f = j / L, l = j % L;
            const LevelDesc& D = PD.lv[l];
            std::vector<uint32_t> keys;   // concatenation in the reference's cell loop order
            for (int ci = D.cell_first; ci < D.cell_first + D.cell_count; ci++) {
              const uint32_t* sp = h_dense + (size_t)f * PD.cand_frame_slots + cells[ci].cand_base;
              keys.insert(keys.end(), sp, sp + h_cell_count[(size_t)f * PD.ncells + ci]);
            }
            octree_select(keys.data(), (int)keys.size(), kEdge - 3, D.w - kEdge + 3, kEdge - 3, D.h - kEdge + 3, D.quota, out);
            const int n = std::min<int>((int)out.size(), D.sel_cap);
            h_nsel[f * L + l] = n;
            std::memcpy(h_sel + (size_t)f * PD.sel_frame_slots + D.sel_off, out.data(), (size_t)n * 4);
          }
        };
        if (nthreads == 1) work(0);
        else {
          std::vector<std::thread> th;
          for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
          for (auto& t : th) t.join();
        }
      }
      DVM_HIP(hipMemcpyAsync(d_nsel, h_nsel, (size_t)batch * L * 4, hipMemcpyHostToDevice, st));
      DVM_HIP(hipMemcpyAsync(d_sel, h_sel, (size_t)batch * PD.sel_frame_slots * 4, hipMemcpyHostToDevice, st));
    }
#endif

    prof.begin(st, "assemble");
    launch_assemble(st, (d_sel + (size_t)f0 * PD.sel_frame_slots), (d_nsel + (size_t)f0 * L), PD, lap0, lap1, (d_kps + (size_t)f0 * PD.kp_cap), (d_aux + (size_t)f0 * PD.kp_cap), (d_n + f0), (d_mono + f0), nb, hm);
    prof.end(st);
    if (!blur_forked && !blur_done) {
      prof.begin(st, "blur");
      launch_blur(st, (d_pyr + (size_t)f0 * PD.pyr_frame_bytes), (d_blur + (size_t)f0 * PD.blur_frame_bytes), d_tiles, PD, nullptr, nb);
      prof.end(st);
    } else {
      DVM_HIP(hipStreamWaitEvent(st, ev_join[ck], 0));
    }
    prof.begin(st, "orient_desc");
    launch_orient_desc(st, (d_pyr + (size_t)f0 * PD.pyr_frame_bytes), (d_blur + (size_t)f0 * PD.blur_frame_bytes), PD, (d_aux + (size_t)f0 * PD.kp_cap), (d_n + f0), (d_kps + (size_t)f0 * PD.kp_cap), (d_desc + (size_t)f0 * PD.kp_cap * 32), nb, hm);
    prof.end(st);

    return DVM_OK;
  };
  if (!small && !lane_side[0] && overlap_blur) DVM_HIP(hipStreamCreateWithPriority(&lane_side[0], hipStreamNonBlocking, side_priority));
  if (nck == 1) {
    rc = run_half(stream, lane_side[0], 0, 0, batch);
    if (rc != DVM_OK) return rc;
  } else {
    DVM_HIP(hipEventRecord(ev_start, stream));              // everything queued on `stream` so far precedes the batch
    DVM_HIP(hipStreamWaitEvent(lane_main[1], ev_start, 0));
    const int per = (batch + nck - 1) / nck;
    for (int c = 0; c < nck; c++) {
      const int f0 = c * per, nb = std::min(per, batch - f0);
      if (nb <= 0) { nck = c; break; }
      hipStream_t st = lane_main[c & 1];
      if (c > 0) DVM_HIP(hipStreamWaitEvent(st, ev_compact[c - 1], 0));
      rc = run_half(st, lane_side[c & 1], c, f0, nb);
      if (rc != DVM_OK) return rc;
    }
    DVM_HIP(hipEventRecord(ev_done, lane_main[1]));         // `stream` stays the handle's ordering point
    DVM_HIP(hipStreamWaitEvent(stream, ev_done, 0));
  }
  DVM_HIP(hipGetLastError());
  return DVM_OK;
}

int OrbPipeline::sync() {
  DVM_HIP(hipSetDevice(device));
  DVM_HIP(hipStreamSynchronize(stream));
  prof.resolve();
  // the device octree's overflow flag: checked at every synchronisation point, so the device-resident consumers
  // (dvm_orb_result_device, dvm_orb_copy_result, the wire gather) cannot read a silently truncated level either
  if (h_err && *reinterpret_cast<volatile int32_t*>(h_err)) {
    set_error("device octree: node capacity exceeded (internal: configure() refuses such quotas)");
    return DVM_ERR_CAPACITY;
  }
  return DVM_OK;
}

int OrbPipeline::download(int frame, dvm_keypoint* kps, uint8_t* desc, int cap, int* n, int* mono) {
  if (!configured || frame < 0 || frame >= last_batch) { set_error("no results for that frame"); return DVM_ERR_STATE; }
  DVM_HIP(hipSetDevice(device));
  if (last_mirrored) {   // latency path: the kernels stored the results here as they produced them
    int rc = sync();
    if (rc != DVM_OK) return rc;
    const int N = h_n[frame];
    if (n) *n = N;
    if (mono) *mono = h_mono[frame];
    if (N > cap) { set_error("keypoint buffer too small"); return DVM_ERR_CAPACITY; }
    if (N > 0) {
      if (kps) std::memcpy(kps, h_kps_m + (size_t)frame * PD.kp_cap, (size_t)N * sizeof(dvm_keypoint));
      if (desc) std::memcpy(desc, h_desc_m + (size_t)frame * PD.kp_cap * 32, (size_t)N * 32);
    }
    return DVM_OK;
  }
  DVM_HIP(hipMemcpyAsync(h_n, d_n, (size_t)last_batch * 4, hipMemcpyDeviceToHost, stream));
  DVM_HIP(hipMemcpyAsync(h_mono, d_mono, (size_t)last_batch * 4, hipMemcpyDeviceToHost, stream));
  int rc = sync();
  if (rc != DVM_OK) return rc;
  const int N = h_n[frame];
  if (n) *n = N;
  if (mono) *mono = h_mono[frame];
  if (N > cap) { set_error("keypoint buffer too small"); return DVM_ERR_CAPACITY; }
  if (N > 0) {
    if (kps) DVM_HIP(hipMemcpy(kps, d_kps + (size_t)frame * PD.kp_cap, (size_t)N * sizeof(dvm_keypoint), hipMemcpyDeviceToHost));
    if (desc) DVM_HIP(hipMemcpy(desc, d_desc + (size_t)frame * PD.kp_cap * 32, (size_t)N * 32, hipMemcpyDeviceToHost));
  }
  return DVM_OK;
}

// the first `count` frames' results in one go: two copies of the whole [count][kp_cap] blocks into page-locked memory behind the handle's
// stream, ONE synchronisation, then the frames' N entries each (download() frame by frame is two blocking copies to pageable memory per
// frame: 32 frames of a batched tracking chain spent 1.3 ms there)
int OrbPipeline::download_batch(int count, dvm_keypoint* const* kps, uint8_t* const* desc, const int* caps, int* n, int* mono) {
  if (!configured || count < 1 || count > last_batch) { set_error("no results for those frames"); return DVM_ERR_STATE; }
  if (last_mirrored || count == 1) {
    for (int f = 0; f < count; f++) {
      const int rc = download(f, kps ? kps[f] : nullptr, desc ? desc[f] : nullptr, caps[f], n ? n + f : nullptr, mono ? mono + f : nullptr);
      if (rc != DVM_OK) return rc;
    }
    return DVM_OK;
  }
  DVM_HIP(hipSetDevice(device));
  const size_t need = (size_t)count * PD.kp_cap;
  if (need > h_batch_cap) {
    if (h_kps_b) hipHostFree(h_kps_b);
    if (h_desc_b) hipHostFree(h_desc_b);
    h_kps_b = nullptr; h_desc_b = nullptr; h_batch_cap = 0;
    DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_kps_b), need * sizeof(dvm_keypoint_pod)));
    DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_desc_b), need * 32));
    h_batch_cap = need;
  }
  DVM_HIP(hipMemcpyAsync(h_n, d_n, (size_t)last_batch * 4, hipMemcpyDeviceToHost, stream));
  DVM_HIP(hipMemcpyAsync(h_mono, d_mono, (size_t)last_batch * 4, hipMemcpyDeviceToHost, stream));
  DVM_HIP(hipMemcpyAsync(h_kps_b, d_kps, need * sizeof(dvm_keypoint_pod), hipMemcpyDeviceToHost, stream));
  DVM_HIP(hipMemcpyAsync(h_desc_b, d_desc, need * 32, hipMemcpyDeviceToHost, stream));
  const int rc = sync();
  if (rc != DVM_OK) return rc;
  for (int f = 0; f < count; f++) {
    const int N = h_n[f];
    if (n) n[f] = N;
    if (mono) mono[f] = h_mono[f];
    if (N > caps[f]) { set_error("keypoint buffer too small"); return DVM_ERR_CAPACITY; }
  }
  HostPool::get().run((size_t)count, count >= 4 ? 8 : 1, [&](size_t f) {
    const int N = h_n[f];
    if (N <= 0) return;
    if (kps && kps[f]) std::memcpy(kps[f], h_kps_b + f * PD.kp_cap, (size_t)N * sizeof(dvm_keypoint));
    if (desc && desc[f]) std::memcpy(desc[f], h_desc_b + f * PD.kp_cap * 32, (size_t)N * 32);
  });
  return DVM_OK;
}

}  // namespace dvm