// dvm_slam_amd/csrc/orb_pool.cpp -- a shared extractor for several agents' frames (dvm_orb_pool_*, include/dvmslam_hip.h).
//
// The reference runs one ORBextractor per agent (orb_slam3_wrapper.cpp: one System per agent; Tracking.cc:1423-1426 calls it once per
// frame).  K agents that share a GPU and each call dvm_orb_extract from their own tracking thread issue K chains of ~13 launches that the
// runtime serialises on its launch path: eight threads reach 6.5 k frames/s together, where ONE launch group of eight frames runs at
// 58 k frames/s.  The pool turns the former into the latter without changing what a caller sees: dvm_orb_pool_extract is the same
// blocking call (one image in, that frame's keypoints and descriptors out, the same bytes), but frames that arrive within a short
// window are extracted as ONE batch -- "group commit":
//   * a caller copies its image into the next free slot of the collecting lane's pinned staging buffer (outside the lock);
//   * the caller that took slot 0 leads the batch: it waits until the batch is full or no frame has joined for `window_us`, closes
//     it, lets the other lane start collecting, runs the batch (dvm_orb_extract_staged: H2D copy + the batch pipeline), fetches
//     all results with two copies into pinned memory and wakes the batch;
//   * every caller copies its own frame's results out (no GPU call, no lock) and the last one frees the lane.
// Two lanes: while one batch runs, the next one collects.  Frames of a batch must share size and lapping area; a frame that does not
// match the collecting batch waits for the next one.
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/dvmslam_hip.h"
#include "orb_pipeline.h"

using namespace dvm;

namespace {
struct Lane {
  OrbPipeline* p = nullptr;
  enum State { FREE, COLLECT, RUN, DONE } state = FREE;
  int count = 0, copied = 0, readers = 0, rc = DVM_OK;
  int rows = 0, cols = 0, lap0 = 0, lap1 = 0;
  uint8_t* stage = nullptr;                 // the lane's pinned staging buffer (max_batch frames of rows x cols)
  std::chrono::steady_clock::time_point last_join;
  // results of the batch in pinned host memory
  dvm_keypoint* h_kps = nullptr;
  uint8_t* h_desc = nullptr;
  int32_t* h_n = nullptr;
  int32_t* h_mono = nullptr;
  size_t res_cap = 0;                       // frames x kp_cap the result buffers hold
  int kp_cap = 0;
  std::string err;
};
}  // namespace

struct dvm_orb_pool {
  dvm_orb_params P;
  int device = 0, max_batch = 0, window_us = 0;
  std::mutex m;
  std::condition_variable cv;
  Lane lane[2];
  int cur = 0;   // the lane new frames join
};

static void lane_free_results(Lane& L) {
  if (L.h_kps) hipHostFree(L.h_kps);
  if (L.h_desc) hipHostFree(L.h_desc);
  if (L.h_n) hipHostFree(L.h_n);
  if (L.h_mono) hipHostFree(L.h_mono);
  L.h_kps = nullptr; L.h_desc = nullptr; L.h_n = L.h_mono = nullptr; L.res_cap = 0;
}

extern "C" int dvm_orb_pool_create(const dvm_orb_params* p, int device, int max_batch, int window_us, dvm_orb_pool** out) {
  if (!p || !out || max_batch < 1 || max_batch > 256 || p->nlevels < 1 || p->nlevels > kMaxLevels || p->nfeatures < 0 || !(p->scale_factor > 1.0f)) {
    set_error("dvm_orb_pool_create: bad parameters");
    return DVM_ERR_INVALID;
  }
  *out = nullptr;
  dvm_orb_pool* pool = new (std::nothrow) dvm_orb_pool();
  if (!pool) return DVM_ERR_INVALID;
  pool->P = *p; pool->device = device; pool->max_batch = max_batch; pool->window_us = window_us < 0 ? 20 : window_us;
  for (Lane& L : pool->lane) {
    L.p = new (std::nothrow) OrbPipeline(*p, device, max_batch);
    const int rc = L.p ? L.p->init() : DVM_ERR_INVALID;
    if (rc != DVM_OK) {
      for (Lane& K : pool->lane) delete K.p;
      delete pool;
      return rc;
    }
  }
  *out = pool;
  return DVM_OK;
}

extern "C" void dvm_orb_pool_destroy(dvm_orb_pool* pool) {
  if (!pool) return;
  hipSetDevice(pool->device);
  for (Lane& L : pool->lane) { lane_free_results(L); delete L.p; }
  delete pool;
}

// the leader's part: run the closed batch of lane L (state RUN, all images copied) and fetch its results
static int run_batch(dvm_orb_pool* pool, Lane& L) {
  OrbPipeline& P = *L.p;
  int rc = hip_check(hipSetDevice(pool->device), "hipSetDevice");
  if (rc == DVM_OK) rc = P.extract_staged(L.count, L.rows, L.cols, L.lap0, L.lap1);
  if (rc != DVM_OK) return rc;
  const size_t cap = (size_t)P.PD.kp_cap, need = (size_t)pool->max_batch * cap;
  if (need > L.res_cap || (int)cap != L.kp_cap) {
    lane_free_results(L);
    DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&L.h_kps), need * sizeof(dvm_keypoint)));
    DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&L.h_desc), need * 32));
    DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&L.h_n), (size_t)pool->max_batch * 4));
    DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&L.h_mono), (size_t)pool->max_batch * 4));
    L.res_cap = need; L.kp_cap = (int)cap;
  }
  const size_t B = (size_t)L.count;
  DVM_HIP(hipMemcpyAsync(L.h_n, P.d_n, B * 4, hipMemcpyDeviceToHost, P.stream));
  DVM_HIP(hipMemcpyAsync(L.h_mono, P.d_mono, B * 4, hipMemcpyDeviceToHost, P.stream));
  DVM_HIP(hipMemcpyAsync(L.h_kps, P.d_kps, B * cap * sizeof(dvm_keypoint), hipMemcpyDeviceToHost, P.stream));
  DVM_HIP(hipMemcpyAsync(L.h_desc, P.d_desc, B * cap * 32, hipMemcpyDeviceToHost, P.stream));
  return P.sync();
}

extern "C" int dvm_orb_pool_extract(dvm_orb_pool* pool, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1, dvm_keypoint* kps,
                                    uint8_t* desc, int cap, int* n, int* mono_index, int* batch_size) {
  if (!pool) return DVM_ERR_INVALID;
  if (n) *n = 0;
  if (mono_index) *mono_index = -1;
  if (batch_size) *batch_size = 0;
  if (!img || rows <= 0 || cols <= 0) return DVM_ERR_EMPTY;
  if (stride < cols) { set_error("bad stride"); return DVM_ERR_INVALID; }
  using clock = std::chrono::steady_clock;
  std::unique_lock<std::mutex> lk(pool->m);
  // ---- join the collecting batch (or open one)
  int li, slot;
  for (;;) {
    Lane& C = pool->lane[pool->cur];
    if (C.state == Lane::FREE) {
      // staging(): pinned buffer of the lane for this size (first use / a new size allocates; the lane is idle)
      uint8_t* sp = nullptr;
      int rc = hip_check(hipSetDevice(pool->device), "hipSetDevice");
      if (rc == DVM_OK) rc = C.p->staging(pool->max_batch, rows, cols, &sp);
      if (rc != DVM_OK) return rc;
      C.stage = sp; C.state = Lane::COLLECT; C.count = 0; C.copied = 0; C.readers = 0; C.rc = DVM_OK;
      C.rows = rows; C.cols = cols; C.lap0 = lap0; C.lap1 = lap1;
    }
    if (C.state == Lane::COLLECT && C.count < pool->max_batch && C.rows == rows && C.cols == cols && C.lap0 == lap0 && C.lap1 == lap1) {
      li = pool->cur; slot = C.count++; C.last_join = clock::now();
      break;
    }
    pool->cv.wait(lk);   // the collecting batch is full / of another shape / both lanes busy: the next state change wakes us
  }
  Lane& L = pool->lane[li];
  // ---- the image into this frame's slot (other callers copy theirs at the same time)
  lk.unlock();
  uint8_t* dst = L.stage + (size_t)slot * rows * cols;
  for (int y = 0; y < rows; y++) std::memcpy(dst + (size_t)y * cols, img + (size_t)y * stride, (size_t)cols);
  lk.lock();
  L.copied++;
  if (slot == 0) {
    // ---- leader: wait for the batch to fill or for the arrivals to pause, close it, run it
    const auto window = std::chrono::microseconds(pool->window_us);
    while (L.count < pool->max_batch) {
      const auto deadline = L.last_join + window;
      if (clock::now() >= deadline) break;
      pool->cv.wait_until(lk, deadline);
    }
    L.state = Lane::RUN;                               // closed: nobody joins any more
    if (pool->lane[li ^ 1].state == Lane::FREE || pool->lane[li ^ 1].state == Lane::COLLECT) pool->cur = li ^ 1;
    pool->cv.notify_all();                             // waiting callers may open the other lane
    while (L.copied < L.count) pool->cv.wait(lk);      // every joined frame is in the staging buffer
    lk.unlock();
    const int rc = run_batch(pool, L);
    const std::string err = rc == DVM_OK ? std::string() : std::string(last_error_cstr());
    lk.lock();
    L.rc = rc; L.err = err; L.readers = L.count; L.state = Lane::DONE;
    pool->cv.notify_all();
  } else {
    pool->cv.notify_all();                             // (the leader may be waiting for this copy or for this arrival)
    while (L.state != Lane::DONE) pool->cv.wait(lk);
  }
  // ---- this frame's results (the lane stays DONE until every frame of the batch has been read)
  const int rc = L.rc;
  const int count = L.count;
  std::string err = L.err;
  lk.unlock();
  int out_rc = rc;
  if (rc == DVM_OK) {
    const int N = L.h_n[slot];
    if (n) *n = N;
    if (mono_index) *mono_index = L.h_mono[slot];
    if (batch_size) *batch_size = count;
    if (N > cap) { set_error("keypoint buffer too small"); out_rc = DVM_ERR_CAPACITY; }
    else if (N > 0) {
      if (kps) std::memcpy(kps, L.h_kps + (size_t)slot * L.kp_cap, (size_t)N * sizeof(dvm_keypoint));
      if (desc) std::memcpy(desc, L.h_desc + (size_t)slot * L.kp_cap * 32, (size_t)N * 32);
    }
  } else {
    set_error("dvm_orb_pool_extract: " + err);
  }
  lk.lock();
  if (--L.readers == 0) {
    L.state = Lane::FREE;
    if (pool->lane[pool->cur].state != Lane::COLLECT) pool->cur = li;   // (nothing is collecting: the freed lane is the next to open)
    pool->cv.notify_all();
  }
  return out_rc;
}
