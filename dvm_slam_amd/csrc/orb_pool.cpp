// dvm_slam_amd/csrc/orb_pool.cpp -- shared per-GPU services for several agents' per-frame calls (dvm_orb_pool_*, dvm_pose_pool_*;
// include/dvmslam_hip.h), built on the group-commit protocol of group_commit.h.
//
// The reference runs one ORBextractor per agent and calls Optimizer::PoseOptimization from every agent's tracking thread
// (orb_slam3_wrapper.cpp: one System per agent; Tracking.cc:1423-1426, :2632).  K agents that share a GPU and make these calls from K
// threads issue K chains of small launches that the runtime serialises: eight threads reach 6.5 k frames/s together through the
// per-frame calls, where ONE launch group of eight frames extracts at 58 k frames/s.  The pools turn the former into the latter without
// changing what a caller sees: the same blocking call, the same bytes back, but calls that arrive within a short window run as ONE batch.
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/dvmslam_hip.h"
#include "ba_kernels.h"
#include "group_commit.h"
#include "orb_pipeline.h"

using namespace dvm;

// ------------------------------------------------------------------------------------------------ extraction
namespace {
struct OrbLane {
  OrbPipeline* p = nullptr;
  uint8_t* stage = nullptr;                 // the lane's pinned staging buffer (max_batch frames of rows x cols)
  // results of the batch in pinned host memory
  dvm_keypoint* h_kps = nullptr;
  uint8_t* h_desc = nullptr;
  int32_t* h_n = nullptr;
  int32_t* h_mono = nullptr;
  size_t res_cap = 0;                       // frames x kp_cap the result buffers hold
  int kp_cap = 0;
  void free_results() {
    if (h_kps) hipHostFree(h_kps);
    if (h_desc) hipHostFree(h_desc);
    if (h_n) hipHostFree(h_n);
    if (h_mono) hipHostFree(h_mono);
    h_kps = nullptr; h_desc = nullptr; h_n = h_mono = nullptr; res_cap = 0;
  }
};
}  // namespace

struct dvm_orb_pool {
  dvm_orb_params P;
  int device = 0;
  GroupCommit gc;
  OrbLane lane[GroupCommit::kLanes];
};

extern "C" int dvm_orb_pool_create(const dvm_orb_params* p, int device, int max_batch, int window_us, dvm_orb_pool** out) {
  if (!p || !out || max_batch < 1 || max_batch > 256 || p->nlevels < 1 || p->nlevels > kMaxLevels || p->nfeatures < 0 || !(p->scale_factor > 1.0f)) {
    set_error("dvm_orb_pool_create: bad parameters");
    return DVM_ERR_INVALID;
  }
  *out = nullptr;
  dvm_orb_pool* pool = new (std::nothrow) dvm_orb_pool();
  if (!pool) return DVM_ERR_INVALID;
  pool->P = *p; pool->device = device;
  pool->gc.max_batch = max_batch; pool->gc.window_us = window_us < 0 ? 20 : window_us;
  for (OrbLane& L : pool->lane) {
    L.p = new (std::nothrow) OrbPipeline(*p, device, max_batch);
    const int rc = L.p ? L.p->init() : DVM_ERR_INVALID;
    if (rc != DVM_OK) {
      for (OrbLane& K : pool->lane) delete K.p;
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
  for (OrbLane& L : pool->lane) { L.free_results(); delete L.p; }
  delete pool;
}

// the leader's part: run the closed batch (all images are in the staging buffer) and fetch its results with four copies
static int orb_run_batch(dvm_orb_pool* pool, OrbLane& L, int count, int rows, int cols, int lap0, int lap1) {
  OrbPipeline& P = *L.p;
  int rc = hip_check(hipSetDevice(pool->device), "hipSetDevice");
  if (rc == DVM_OK) rc = P.extract_staged_sync_owner(count, rows, cols, lap0, lap1);   // (a batch of up to four frames takes the one-frame path)
  if (rc != DVM_OK) return rc;
  const size_t cap = (size_t)P.PD.kp_cap, need = (size_t)pool->gc.max_batch * cap;
  if (need > L.res_cap || (int)cap != L.kp_cap) {
    L.free_results();
    DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&L.h_kps), need * sizeof(dvm_keypoint)));
    DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&L.h_desc), need * 32));
    DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&L.h_n), (size_t)pool->gc.max_batch * 4));
    DVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&L.h_mono), (size_t)pool->gc.max_batch * 4));
    L.res_cap = need; L.kp_cap = (int)cap;
  }
  const size_t B = (size_t)count;
  if (P.last_mirrored) {   // the kernels stored the results into mapped host memory as they produced them: no copy commands
    rc = P.sync();
    if (rc != DVM_OK) return rc;
    std::memcpy(L.h_n, P.h_n, B * 4); std::memcpy(L.h_mono, P.h_mono, B * 4);
    for (size_t b = 0; b < B; b++) {
      const size_t nb = (size_t)std::max(P.h_n[b], 0);
      std::memcpy(L.h_kps + b * cap, P.h_kps_m + b * cap, nb * sizeof(dvm_keypoint));
      std::memcpy(L.h_desc + b * cap * 32, P.h_desc_m + b * cap * 32, nb * 32);
    }
    return DVM_OK;
  }
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
  const int64_t key[4] = {rows, cols, lap0, lap1};   // the frames of a batch share size and lapping area
  int li = 0, slot = 0;
  int rc = pool->gc.join(key, [&](int l) {
    // the lane's pinned buffer for this size (first use / a new size allocates; the lane is idle)
    int r = hip_check(hipSetDevice(pool->device), "hipSetDevice");
    if (r == DVM_OK) r = pool->lane[l].p->staging(pool->gc.max_batch, rows, cols, &pool->lane[l].stage);
    return r;
  }, li, slot);
  if (rc != DVM_OK) return rc;
  OrbLane& L = pool->lane[li];
  uint8_t* dst = L.stage + (size_t)slot * rows * cols;   // this frame's slot (other callers copy theirs at the same time)
  for (int y = 0; y < rows; y++) std::memcpy(dst + (size_t)y * cols, img + (size_t)y * stride, (size_t)cols);
  if (pool->gc.arrive(li, slot)) {
    const int r = orb_run_batch(pool, L, pool->gc.batch_count(li), rows, cols, lap0, lap1);
    pool->gc.publish(li, r, r == DVM_OK ? std::string() : std::string(last_error_cstr()));
  }
  std::string err;
  int count = 0;
  rc = pool->gc.result(li, &err, &count);
  if (rc == DVM_OK) {
    const int N = L.h_n[slot];
    if (n) *n = N;
    if (mono_index) *mono_index = L.h_mono[slot];
    if (batch_size) *batch_size = count;
    if (N > cap) { set_error("keypoint buffer too small"); rc = DVM_ERR_CAPACITY; }
    else if (N > 0) {
      if (kps) std::memcpy(kps, L.h_kps + (size_t)slot * L.kp_cap, (size_t)N * sizeof(dvm_keypoint));
      if (desc) std::memcpy(desc, L.h_desc + (size_t)slot * L.kp_cap * 32, (size_t)N * 32);
    }
  } else {
    set_error("dvm_orb_pool_extract: " + err);
  }
  pool->gc.finish(li);
  return rc;
}

// ------------------------------------------------------------------------------------------------ PoseOptimization
// Optimizer::PoseOptimization (Optimizer.cc:744-1028) of several agents' current frames as ONE launch of k_pose_optimize (a workgroup
// per frame: the kernel is the batched form already, dvm_pose_optimize(batch)).  A lane keeps its slots in MAPPED host memory: a caller
// writes its correspondences straight into its slot, the kernel reads them once (register-resident up to kPoolPoseMaxN per frame)
// and writes pose, outlier flags and inlier count back in place -- no copy command around the launch.
namespace {
constexpr int kPoolPoseMaxN = 1280;   // correspondences per frame the kernel keeps in registers (ba_kernels.hip: kPoseEdgesPerThread x 256)
struct PoseLane {
  double *pose_in = nullptr, *X = nullptr, *obs = nullptr, *w = nullptr, *pose_out = nullptr, *chi = nullptr;   // mapped host (chi: device)
  int32_t *n = nullptr, *ninl = nullptr;
  uint8_t* outl = nullptr;
  hipStream_t stream = nullptr;
  dvm_ba_camera cam{};
};
}  // namespace

struct dvm_pose_pool {
  int device = 0;
  GroupCommit gc;
  PoseLane lane[GroupCommit::kLanes];
};

static void pose_lane_free(PoseLane& L) {
  for (void* p : {(void*)L.pose_in, (void*)L.X, (void*)L.obs, (void*)L.w, (void*)L.pose_out, (void*)L.n, (void*)L.ninl, (void*)L.outl})
    if (p) hipHostFree(p);
  if (L.chi) hipFree(L.chi);
  if (L.stream) hipStreamDestroy(L.stream);
  L = PoseLane{};
}

extern "C" int dvm_pose_pool_create(int device, int max_batch, int window_us, dvm_pose_pool** out) {
  if (!out || max_batch < 1 || max_batch > 256) { set_error("dvm_pose_pool_create: bad parameters"); return DVM_ERR_INVALID; }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  if (device < 0 || device >= ndev) return DVM_ERR_INVALID;
  DVM_HIP(hipSetDevice(device));
  dvm_pose_pool* pool = new (std::nothrow) dvm_pose_pool();
  if (!pool) return DVM_ERR_INVALID;
  pool->device = device;
  pool->gc.max_batch = max_batch; pool->gc.window_us = window_us < 0 ? 20 : window_us;
  const size_t B = (size_t)max_batch, S = kPoolPoseMaxN;
  int rc = DVM_OK;
  for (PoseLane& L : pool->lane) {
    auto mapped = [&](void** p, size_t bytes) { if (rc == DVM_OK) rc = hip_check(hipHostMalloc(p, bytes, hipHostMallocMapped), "hipHostMalloc"); };
    mapped(reinterpret_cast<void**>(&L.pose_in), B * 7 * 8); mapped(reinterpret_cast<void**>(&L.X), B * S * 3 * 8);
    mapped(reinterpret_cast<void**>(&L.obs), B * S * 2 * 8); mapped(reinterpret_cast<void**>(&L.w), B * S * 8);
    mapped(reinterpret_cast<void**>(&L.pose_out), B * 7 * 8); mapped(reinterpret_cast<void**>(&L.n), B * 4);
    mapped(reinterpret_cast<void**>(&L.ninl), B * 4); mapped(reinterpret_cast<void**>(&L.outl), B * S);
    if (rc == DVM_OK) rc = hip_check(hipMalloc(reinterpret_cast<void**>(&L.chi), B * S * 8), "hipMalloc");
    if (rc == DVM_OK) rc = hip_check(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking), "stream");
  }
  if (rc != DVM_OK) {
    for (PoseLane& L : pool->lane) pose_lane_free(L);
    delete pool;
    return rc;
  }
  *out = pool;
  return DVM_OK;
}

extern "C" void dvm_pose_pool_destroy(dvm_pose_pool* pool) {
  if (!pool) return;
  hipSetDevice(pool->device);
  for (PoseLane& L : pool->lane) { if (L.stream) hipStreamSynchronize(L.stream); pose_lane_free(L); }
  delete pool;
}

extern "C" int dvm_pose_pool_optimize(dvm_pose_pool* pool, const double* pose_in, const double* Xw, const double* obs, const double* inv_sigma2, int n,
                                      const dvm_ba_camera* cam, double* pose_out, uint8_t* outlier, int32_t* n_inliers, int* batch_size) {
  if (!pool || !pose_in || !cam || !pose_out || !n_inliers || n < 0 || (n && (!Xw || !obs || !inv_sigma2 || !outlier))) {
    set_error("dvm_pose_pool_optimize: bad arguments");
    return DVM_ERR_INVALID;
  }
  if (batch_size) *batch_size = 0;
  if (n > kPoolPoseMaxN)   // beyond what the kernel keeps in registers it re-reads the correspondences every iteration: not from host memory
    return dvm_pose_optimize(pool->device, pose_in, Xw, obs, inv_sigma2, &n, n, 1, cam, pose_out, outlier, n_inliers);
  int64_t key[4];   // the frames of a batch share the camera (the kernel takes one set of intrinsics)
  static_assert(sizeof(double) == sizeof(int64_t), "key");
  std::memcpy(&key[0], &cam->fx, 8); std::memcpy(&key[1], &cam->fy, 8); std::memcpy(&key[2], &cam->cx, 8); std::memcpy(&key[3], &cam->cy, 8);
  int li = 0, slot = 0;
  int rc = pool->gc.join(key, [&](int l) { pool->lane[l].cam = *cam; return 0; }, li, slot);
  if (rc != DVM_OK) return rc;
  PoseLane& L = pool->lane[li];
  const size_t S = kPoolPoseMaxN, s = (size_t)slot;
  std::memcpy(L.pose_in + 7 * s, pose_in, 7 * 8);
  if (n) {
    std::memcpy(L.X + s * S * 3, Xw, (size_t)n * 3 * 8);
    std::memcpy(L.obs + s * S * 2, obs, (size_t)n * 2 * 8);
    std::memcpy(L.w + s * S, inv_sigma2, (size_t)n * 8);
  }
  L.n[slot] = n;
  if (pool->gc.arrive(li, slot)) {
    const int count = pool->gc.batch_count(li);
    int r = hip_check(hipSetDevice(pool->device), "hipSetDevice");
    if (r == DVM_OK) {
      // (mapped allocations: the device address of a hipHostMalloc'ed buffer is the host address under unified addressing, taken explicitly)
      void *dP, *dX, *dO, *dW, *dN, *dPo, *dL, *dI;
      hipHostGetDevicePointer(&dP, L.pose_in, 0); hipHostGetDevicePointer(&dX, L.X, 0); hipHostGetDevicePointer(&dO, L.obs, 0);
      hipHostGetDevicePointer(&dW, L.w, 0); hipHostGetDevicePointer(&dN, L.n, 0); hipHostGetDevicePointer(&dPo, L.pose_out, 0);
      hipHostGetDevicePointer(&dL, L.outl, 0); hipHostGetDevicePointer(&dI, L.ninl, 0);
      ba_launch_pose_optimize(L.stream, static_cast<const double*>(dP), static_cast<const double*>(dX), static_cast<const double*>(dO),
                              static_cast<const double*>(dW), static_cast<const int32_t*>(dN), kPoolPoseMaxN, count, L.cam.fx, L.cam.fy, L.cam.cx,
                              L.cam.cy, static_cast<double*>(dPo), static_cast<uint8_t*>(dL), static_cast<int32_t*>(dI), L.chi);
      r = hip_check(hipGetLastError(), "pose_optimize launch");
      if (r == DVM_OK) r = hip_check(hipStreamSynchronize(L.stream), "pose_optimize");
    }
    pool->gc.publish(li, r, r == DVM_OK ? std::string() : std::string(last_error_cstr()));
  }
  std::string err;
  int count = 0;
  rc = pool->gc.result(li, &err, &count);
  if (rc == DVM_OK) {
    std::memcpy(pose_out, L.pose_out + 7 * s, 7 * 8);
    if (n) std::memcpy(outlier, L.outl + s * S, (size_t)n);
    *n_inliers = L.ninl[slot];
    if (batch_size) *batch_size = count;
  } else {
    set_error("dvm_pose_pool_optimize: " + err);
  }
  pool->gc.finish(li);
  return rc;
}
