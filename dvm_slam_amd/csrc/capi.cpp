// dvm_slam_amd/csrc/capi.cpp -- extern "C" boundary of libdvmslam_hip.so (include/dvmslam_hip.h).
// Thin: argument checks, handle lifetime, host<->device staging.  No compute happens on the host
// here; if no HIP device is visible every entry point that needs one fails with DVM_ERR_NO_DEVICE.
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/dvmslam_hip.h"
#include "match_kernels.h"
#include "orb_pipeline.h"

namespace dvm {
const char* last_error_cstr();
}
using namespace dvm;

struct dvm_orb {
  OrbPipeline* p;
  float* d_scale = nullptr;
  uint64_t id = 0, serial = 0;      // dvm_device_frame: which handle, which extraction
  int last_n = 0;
};
// handle id -> serial of its current result (dvm_device_frame_valid); ids are never reused
static std::mutex g_orb_live_mu;
static std::unordered_map<uint64_t, uint64_t> g_orb_live;
static uint64_t g_orb_next_id = 1;
static void orb_result_changed(dvm_orb* h, int n) {
  std::lock_guard<std::mutex> l(g_orb_live_mu);
  if (!h->id) h->id = g_orb_next_id++;
  g_orb_live[h->id] = ++h->serial;
  h->last_n = n;
}

struct dvm_frame {
  int device, cap, slots;
  FrameView view{};
  // scratch for the host-pointer convenience paths
  void* d_scratch = nullptr;
  size_t scratch_bytes = 0;
};

namespace dvm { FrameView frame_view_of(const ::dvm_frame* f) { return f->view; } }

static int need_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_error("no HIP device visible (libdvmslam_hip has no CPU path)");
    return DVM_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    set_error("device index out of range");
    return DVM_ERR_INVALID;
  }
  return hip_check(hipSetDevice(device), "hipSetDevice");
}

#include "group_commit.h"
#include "host_stage.h"

extern "C" {

const char* dvm_last_error(void) { return last_error_cstr(); }
const char* dvm_version(void) { return "dvmslam-hip 0.1 (gfx950)"; }
int dvm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int dvm_set_device(int device) { return need_device(device); }

// ------------------------------------------------------------------------------------------ ORB
int dvm_orb_create(const dvm_orb_params* p, int device, int max_batch, dvm_orb** out) {
  if (!p || !out || p->nlevels < 1 || p->nlevels > kMaxLevels || p->nfeatures < 0 || !(p->scale_factor > 1.0f)) {
    set_error("dvm_orb_create: bad parameters");
    return DVM_ERR_INVALID;
  }
  *out = nullptr;
  OrbPipeline* pipe = new (std::nothrow) OrbPipeline(*p, device, max_batch);
  if (!pipe) return DVM_ERR_INVALID;
  int rc = pipe->init();
  if (rc != DVM_OK) {
    delete pipe;
    return rc;
  }
  dvm_orb* h = new dvm_orb{pipe};
  if (hipMalloc(&h->d_scale, sizeof(float) * kMaxLevels) == hipSuccess)
    hipMemcpy(h->d_scale, pipe->scale.data(), sizeof(float) * p->nlevels, hipMemcpyHostToDevice);
  *out = h;
  return DVM_OK;
}
void dvm_orb_destroy(dvm_orb* h) {
  if (!h) return;
  if (h->id) { std::lock_guard<std::mutex> l(g_orb_live_mu); g_orb_live.erase(h->id); }
  if (h->d_scale) hipFree(h->d_scale);
  delete h->p;
  delete h;
}
int dvm_orb_tables(const dvm_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int32_t* nfeat) {
  if (!h) return DVM_ERR_INVALID;
  const OrbPipeline& P = *h->p;
  for (int i = 0; i < P.params.nlevels; i++) {
    if (scale) scale[i] = P.scale[i];
    if (inv_scale) inv_scale[i] = P.inv_scale[i];
    if (sigma2) sigma2[i] = P.sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = P.inv_sigma2[i];
    if (nfeat) nfeat[i] = P.nfeat[i];
  }
  return DVM_OK;
}
int dvm_orb_extract(dvm_orb* h, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1,
                    dvm_keypoint* kps, uint8_t* desc, int cap, int* n, int* mono_index) {
  if (!h) return DVM_ERR_INVALID;
  if (n) *n = 0;
  if (mono_index) *mono_index = -1;
  orb_result_changed(h, -1);                 // whatever a dvm_device_frame of this handle named is being overwritten
  int rc = h->p->extract_host(img, 1, rows, cols, stride, (int64_t)rows * stride, lap0, lap1);
  if (rc != DVM_OK) return rc;
  int nn = 0;
  rc = h->p->download(0, kps, desc, cap, &nn, mono_index);
  if (n) *n = nn;
  if (rc == DVM_OK) h->last_n = nn;
  return rc;
}
int dvm_orb_last_result(dvm_orb* h, dvm_device_frame* out) {
  if (!h || !out) return DVM_ERR_INVALID;
  if (!h->p->configured || h->last_n < 0 || !h->id) { set_error("dvm_orb_last_result: no single-frame result on this handle"); return DVM_ERR_STATE; }
  OrbPipeline& P = *h->p;
  out->d_kps = reinterpret_cast<const dvm_keypoint*>(P.d_kps); out->d_desc = P.d_desc;
  out->n = h->last_n; out->device = P.device; out->handle_id = h->id; out->serial = h->serial;
  return DVM_OK;
}
int dvm_device_frame_valid(const dvm_device_frame* ref) {
  if (!ref || !ref->handle_id || !ref->d_kps) return 0;
  std::lock_guard<std::mutex> l(g_orb_live_mu);
  const auto it = g_orb_live.find(ref->handle_id);
  return it != g_orb_live.end() && it->second == ref->serial ? 1 : 0;
}
int dvm_orb_extract_batch_device(dvm_orb* h, const uint8_t* d_imgs, int batch, int rows, int cols, int stride,
                                 int64_t frame_stride, int lap0, int lap1) {
  if (!h) return DVM_ERR_INVALID;
  orb_result_changed(h, -1);
  return h->p->extract_device(d_imgs, batch, rows, cols, stride, frame_stride, lap0, lap1);
}
int dvm_orb_extract_batch_host(dvm_orb* h, const uint8_t* imgs, int batch, int rows, int cols, int stride,
                               int64_t frame_stride, int lap0, int lap1) {
  if (!h) return DVM_ERR_INVALID;
  orb_result_changed(h, -1);
  return h->p->extract_host(imgs, batch, rows, cols, stride, frame_stride, lap0, lap1);
}
int dvm_orb_staging(dvm_orb* h, int batch, int rows, int cols, uint8_t** host_ptr) {
  if (!h) return DVM_ERR_INVALID;
  return h->p->staging(batch, rows, cols, host_ptr);
}
int dvm_orb_extract_staged(dvm_orb* h, int batch, int rows, int cols, int lap0, int lap1) {
  if (!h) return DVM_ERR_INVALID;
  orb_result_changed(h, -1);
  return h->p->extract_staged(batch, rows, cols, lap0, lap1);
}
int dvm_orb_sync(dvm_orb* h) { return h ? h->p->sync() : DVM_ERR_INVALID; }
int dvm_orb_result_device(dvm_orb* h, int frame, const dvm_keypoint** d_kps, const uint8_t** d_desc,
                          const int32_t** d_n, int* capacity) {
  if (!h || !h->p->configured || frame < 0 || frame >= h->p->max_batch) return DVM_ERR_STATE;
  OrbPipeline& P = *h->p;
  if (d_kps) *d_kps = reinterpret_cast<const dvm_keypoint*>(P.d_kps + (size_t)frame * P.PD.kp_cap);
  if (d_desc) *d_desc = P.d_desc + (size_t)frame * P.PD.kp_cap * 32;
  if (d_n) *d_n = P.d_n + frame;
  if (capacity) *capacity = P.PD.kp_cap;
  return DVM_OK;
}
int dvm_orb_copy_result(dvm_orb* h, int frame, dvm_keypoint* d_kps_dst, uint8_t* d_desc_dst, int32_t* d_n_dst) {
  if (!h || !h->p->configured || frame < 0 || frame >= h->p->max_batch || !d_kps_dst || !d_desc_dst || !d_n_dst) return DVM_ERR_STATE;
  OrbPipeline& P = *h->p;
  const size_t cap = (size_t)P.PD.kp_cap;
  int rc = hip_check(hipMemcpyAsync(d_kps_dst, P.d_kps + (size_t)frame * cap, cap * sizeof(dvm_keypoint), hipMemcpyDeviceToDevice, P.stream), "copy kps");
  if (rc == DVM_OK) rc = hip_check(hipMemcpyAsync(d_desc_dst, P.d_desc + (size_t)frame * cap * 32, cap * 32, hipMemcpyDeviceToDevice, P.stream), "copy desc");
  if (rc == DVM_OK) rc = hip_check(hipMemcpyAsync(d_n_dst, P.d_n + frame, 4, hipMemcpyDeviceToDevice, P.stream), "copy n");
  return rc;
}
const float* dvm_orb_scale_factors_device(dvm_orb* h) { return h ? h->d_scale : nullptr; }
int dvm_orb_download(dvm_orb* h, int frame, dvm_keypoint* kps, uint8_t* desc, int cap, int* n, int* mono_index) {
  if (!h) return DVM_ERR_INVALID;
  return h->p->download(frame, kps, desc, cap, n, mono_index);
}
int dvm_orb_download_batch(dvm_orb* h, int count, dvm_keypoint* const* kps, uint8_t* const* desc, const int* caps, int* n, int* mono_index) {
  if (!h || !caps) return DVM_ERR_INVALID;
  return h->p->download_batch(count, kps, desc, caps, n, mono_index);
}
int dvm_orb_pyramid(dvm_orb* h, int frame, int level, const uint8_t** d_ptr, int* rows, int* cols, int* stride) {
  if (!h || !h->p->configured) return DVM_ERR_STATE;
  OrbPipeline& P = *h->p;
  if (level < 0 || level >= P.PD.nlevels || frame < 0 || frame >= P.max_batch) return DVM_ERR_INVALID;
  const LevelDesc& L = P.PD.lv[level];
  if (d_ptr) *d_ptr = P.d_pyr + (size_t)frame * P.PD.pyr_frame_bytes + L.pyr_off + (size_t)kEdge * L.stride + kEdge;
  if (rows) *rows = L.h;
  if (cols) *cols = L.w;
  if (stride) *stride = L.stride;
  return DVM_OK;
}

int dvm_orb_debug_level(dvm_orb* h, int frame, int level, int bordered, uint8_t* out) {
  if (!h || !h->p->configured || !out) return DVM_ERR_STATE;
  OrbPipeline& P = *h->p;
  if (level < 0 || level >= P.PD.nlevels || frame < 0 || frame >= P.last_batch) return DVM_ERR_INVALID;
  int rc = P.sync();
  if (rc != DVM_OK) return rc;
  const LevelDesc& L = P.PD.lv[level];
  const uint8_t* base = P.d_pyr + (size_t)frame * P.PD.pyr_frame_bytes + L.pyr_off;
  if (bordered)
    return hip_check(hipMemcpy2D(out, L.w + 2 * kEdge, base, L.stride, L.w + 2 * kEdge, L.h + 2 * kEdge, hipMemcpyDeviceToHost), "memcpy2d");
  return hip_check(hipMemcpy2D(out, L.w, base + (size_t)kEdge * L.stride + kEdge, L.stride, L.w, L.h, hipMemcpyDeviceToHost), "memcpy2d");
}
int dvm_orb_debug_blurred(dvm_orb* h, int frame, int level, uint8_t* out) {
  if (!h || !h->p->configured || !out) return DVM_ERR_STATE;
  OrbPipeline& P = *h->p;
  if (level < 0 || level >= P.PD.nlevels || frame < 0 || frame >= P.last_batch) return DVM_ERR_INVALID;
  int rc = P.sync();
  if (rc != DVM_OK) return rc;
  const LevelDesc& L = P.PD.lv[level];
  return hip_check(hipMemcpy2D(out, L.w, P.d_blur + (size_t)frame * P.PD.blur_frame_bytes + L.blur_off, L.blur_stride, L.w, L.h,
                               hipMemcpyDeviceToHost), "memcpy2d");
}
int dvm_orb_debug_candidates(dvm_orb* h, int frame, int level, int32_t* xs, int32_t* ys, int32_t* scores, int cap, int* n) {
  if (!h || !h->p->configured) return DVM_ERR_STATE;
  OrbPipeline& P = *h->p;
  if (level < 0 || level >= P.PD.nlevels || frame < 0 || frame >= P.last_batch) return DVM_ERR_INVALID;
  int rc = P.sync();
  if (rc != DVM_OK) return rc;
  // vToDistributeKeys of the level = the per-cell lists of k_fast_cells in the reference's cell loop order
  const LevelDesc& LD = P.PD.lv[level];
  std::vector<int32_t> cc(std::max(LD.cell_count, 1));
  if (LD.cell_count > 0) {
    rc = hip_check(hipMemcpy(cc.data(), P.d_cell_count + (size_t)frame * P.PD.ncells + LD.cell_first, (size_t)LD.cell_count * 4, hipMemcpyDeviceToHost), "memcpy");
    if (rc != DVM_OK) return rc;
  }
  int cnt = 0;
  for (int c = 0; c < LD.cell_count; c++) cnt += cc[c];
  if (n) *n = cnt;
  if (cnt > cap) return DVM_ERR_CAPACITY;
  std::vector<uint32_t> tmp(cnt > 0 ? cnt : 1);
  for (int c = 0, o = 0; c < LD.cell_count; c++) {
    if (cc[c] > 0) {
      rc = hip_check(hipMemcpy(tmp.data() + o, P.d_cand + (size_t)frame * P.PD.cand_frame_slots + P.cells[LD.cell_first + c].cand_base,
                               (size_t)cc[c] * 4, hipMemcpyDeviceToHost), "memcpy");
      if (rc != DVM_OK) return rc;
      o += cc[c];
    }
  }
  for (int i = 0; i < cnt; i++) {
    int x, y, s;
    unpack_cand(tmp[i], x, y, s);
    xs[i] = x; ys[i] = y; scores[i] = s;
  }
  return DVM_OK;
}
int dvm_orb_debug_level_keypoints(dvm_orb* h, int frame, int level, dvm_keypoint* kps, int cap, int* n) {
  if (!h || !h->p->configured) return DVM_ERR_STATE;
  OrbPipeline& P = *h->p;
  if (level < 0 || level >= P.PD.nlevels || frame < 0 || frame >= P.last_batch) return DVM_ERR_INVALID;
  int rc = P.sync();
  if (rc != DVM_OK) return rc;
  const int L = P.PD.nlevels;
  std::vector<int32_t> nsel(L), nk(1);
  rc = hip_check(hipMemcpy(nsel.data(), P.d_nsel + (size_t)frame * L, (size_t)L * 4, hipMemcpyDeviceToHost), "memcpy");
  if (rc != DVM_OK) return rc;
  int start = 0;
  for (int l = 0; l < level; l++) start += nsel[l];
  const int cnt = nsel[level];
  if (n) *n = cnt;
  if (cnt > cap) return DVM_ERR_CAPACITY;
  if (cnt == 0) return DVM_OK;
  // keypoints g = start..start+cnt-1 in (level, octree) order live at aux[g].out_pos; undo the scaling
  std::vector<KpAux> aux(cnt);
  rc = hip_check(hipMemcpy(aux.data(), P.d_aux + (size_t)frame * P.PD.kp_cap + start, (size_t)cnt * sizeof(KpAux), hipMemcpyDeviceToHost), "memcpy");
  if (rc != DVM_OK) return rc;
  std::vector<dvm_keypoint> all(P.PD.kp_cap);
  rc = hip_check(hipMemcpy(all.data(), P.d_kps + (size_t)frame * P.PD.kp_cap, (size_t)P.PD.kp_cap * sizeof(dvm_keypoint), hipMemcpyDeviceToHost), "memcpy");
  if (rc != DVM_OK) return rc;
  for (int i = 0; i < cnt; i++) {
    dvm_keypoint k = all[aux[i].out_pos];
    k.x = (float)aux[i].cx;
    k.y = (float)aux[i].cy;
    kps[i] = k;
  }
  return DVM_OK;
}

int dvm_orb_profiling(dvm_orb* h, int enable) {
  if (!h) return DVM_ERR_INVALID;
  h->p->prof.enabled = enable != 0;
  h->p->prof.only_fast = enable == 2;
  return DVM_OK;
}
int dvm_orb_profile_get(dvm_orb* h, const char* name, double* total_ms, int64_t* launches) {
  if (!h || !name) return DVM_ERR_INVALID;
  if (total_ms) *total_ms = 0;
  if (launches) *launches = 0;
  return h->p->prof.get(name, total_ms, launches) ? DVM_OK : DVM_ERR_STATE;
}
int dvm_orb_profile_reset(dvm_orb* h) {
  if (!h) return DVM_ERR_INVALID;
  h->p->prof.reset();
  return DVM_OK;
}
void* dvm_orb_stream(dvm_orb* h) { return h ? (void*)h->p->stream : nullptr; }

// ------------------------------------------------------------------------------------- matching
int dvm_hamming_matrix(const uint8_t* A, int nA, const uint8_t* B, int nB, uint16_t* D, int on_device, void* stream) {
  if (nA < 0 || nB < 0 || (nA && !A) || (nB && !B)) return DVM_ERR_INVALID;
  if (nA == 0 || nB == 0) return DVM_OK;
  if (!D) return DVM_ERR_INVALID;
  int rc = need_device(0);
  if (on_device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return DVM_ERR_NO_DEVICE;
    launch_hamming_matrix((hipStream_t)stream, A, nA, B, nB, D);
    return hip_check(hipGetLastError(), "hamming launch");
  }
  if (rc != DVM_OK) return rc;
  Stage st;
  const int iA = st.in(A, (size_t)nA * 32), iB = st.in(B, (size_t)nB * 32), oD = st.out(D, (size_t)nA * nB * 2);
  rc = st.upload();
  if (rc != DVM_OK) return rc;
  launch_hamming_matrix(nullptr, st.ptr<uint8_t>(iA), nA, st.ptr<uint8_t>(iB), nB, st.ptr<uint16_t>(oD));
  rc = hip_check(hipGetLastError(), "hamming launch");
  return rc == DVM_OK ? st.download() : rc;
}

int dvm_frame_create(int device, int capacity, int slots, dvm_frame** out) {
  if (!out || capacity < 1 || capacity > kFrameCap || slots < 1) {
    set_error("dvm_frame_create: capacity must be 1..8192, slots >= 1");
    return DVM_ERR_INVALID;
  }
  *out = nullptr;
  int rc = need_device(device);
  if (rc != DVM_OK) return rc;
  dvm_frame* f = new dvm_frame{};
  f->device = device; f->cap = capacity; f->slots = slots;
  FrameView& V = f->view;
  V.cap = capacity;
  const size_t S = (size_t)slots;
  rc = hip_check(hipMalloc(&V.skp, S * capacity * sizeof(float4)), "hipMalloc");
  if (rc == DVM_OK) rc = hip_check(hipMalloc(&V.sidx, S * capacity * 4), "hipMalloc");
  if (rc == DVM_OK) rc = hip_check(hipMalloc(&V.sdesc, S * capacity * 32), "hipMalloc");
  if (rc == DVM_OK) rc = hip_check(hipMalloc(&V.cellx_start, S * 80 * 4), "hipMalloc");
  if (rc == DVM_OK) rc = hip_check(hipMalloc(&V.n_sorted, S * 4), "hipMalloc");
  if (rc == DVM_OK) rc = hip_check(hipMalloc(&V.n_total, S * 4), "hipMalloc");
  if (rc == DVM_OK) rc = hip_check(hipMalloc(&V.n_overflow, 4), "hipMalloc");
  if (rc == DVM_OK) rc = hip_check(hipMemset(V.n_overflow, 0, 4), "memset");
  if (rc == DVM_OK) rc = hip_check(hipMemset(V.cellx_start, 0, S * 80 * 4), "memset");
  if (rc == DVM_OK) rc = hip_check(hipMemset(V.n_sorted, 0, S * 4), "memset");
  if (rc == DVM_OK) rc = hip_check(hipMemset(V.n_total, 0, S * 4), "memset");
  if (rc != DVM_OK) {
    dvm_frame_destroy(f);
    return rc;
  }
  *out = f;
  return DVM_OK;
}
void dvm_frame_destroy(dvm_frame* f) {
  if (!f) return;
  FrameView& V = f->view;
  if (V.skp) hipFree(V.skp);
  if (V.sidx) hipFree(V.sidx);
  if (V.sdesc) hipFree(V.sdesc);
  if (V.cellx_start) hipFree(V.cellx_start);
  if (V.n_sorted) hipFree(V.n_sorted);
  if (V.n_total) hipFree(V.n_total);
  if (V.n_overflow) hipFree(V.n_overflow);
  if (f->d_scratch) hipFree(f->d_scratch);
  delete f;
}
static int frame_bounds(dvm_frame* f, float minX, float maxX, float minY, float maxY) {
  if (!(maxX > minX) || !(maxY > minY)) {
    set_error("frame bounds empty");
    return DVM_ERR_INVALID;
  }
  FrameView& V = f->view;
  V.minX = minX; V.minY = minY;
  V.wInv = static_cast<float>(64) / static_cast<float>(maxX - minX);  // Frame.cc:443-444
  V.hInv = static_cast<float>(48) / static_cast<float>(maxY - minY);
  return DVM_OK;
}
int dvm_frame_build(dvm_frame* f, int slot, const dvm_keypoint* kps, const uint8_t* desc, int n, const int32_t* d_n,
                    float minX, float maxX, float minY, float maxY, int on_device, void* stream) {
  if (!f || slot < 0 || slot >= f->slots || n < 0 || n > f->cap || (n && (!kps || !desc))) return DVM_ERR_INVALID;
  int rc = frame_bounds(f, minX, maxX, minY, maxY);
  if (rc != DVM_OK) return rc;
  rc = hip_check(hipSetDevice(f->device), "hipSetDevice");
  if (rc != DVM_OK) return rc;
  if (on_device) {
    launch_frame_build((hipStream_t)stream, reinterpret_cast<const dvm_keypoint_pod*>(kps), 0, desc, 0, n, d_n, f->view, slot, 1);
    return hip_check(hipGetLastError(), "frame_build launch");
  }
  // the keypoints / descriptors travel through the calling thread's staging context (asynchronous copy on its blocking
  // stream); the kernel follows on the legacy default stream.  The call returns once the grid is built: a consumer on a
  // NON-BLOCKING stream (the ORB pipeline's and the BA's streams are, and on_device = 1 callers pass any stream) is not
  // ordered behind the default stream and could otherwise read a half-built grid (~10 us for a 1000-keypoint frame)
  // (keypoints and descriptors are read once each by k_frame_build: mapped, no copy in front of the kernel)
  Stage st;
  const int iK = st.in_mapped(kps, (size_t)n * sizeof(dvm_keypoint)), iD = st.in_mapped(desc, (size_t)n * 32);
  rc = st.upload();
  if (rc != DVM_OK) return rc;
  launch_frame_build(st.stream(), st.ptr<dvm_keypoint_pod>(iK), 0, st.ptr<uint8_t>(iD), 0, n, nullptr, f->view, slot, 1);
  rc = hip_check(hipGetLastError(), "frame_build launch");
  return rc == DVM_OK ? hip_check(hipStreamSynchronize(st.stream()), "frame_build") : rc;
}
int dvm_frame_overflows(dvm_frame* f, int32_t* count) {
  if (!f || !count) return DVM_ERR_INVALID;
  int rc = hip_check(hipSetDevice(f->device), "hipSetDevice");
  if (rc != DVM_OK) return rc;
  rc = hip_check(hipMemcpy(count, f->view.n_overflow, 4, hipMemcpyDeviceToHost), "memcpy");   // synchronises with the null stream only
  if (rc != DVM_OK) return rc;
  if (*count) { set_error("frame grid: a device-side keypoint count exceeded the handle's capacity and was truncated"); return DVM_ERR_CAPACITY; }
  return DVM_OK;
}

int dvm_frame_build_batch(dvm_frame* f, int first_slot, int count, const dvm_keypoint* d_kps, int64_t kps_stride,
                          const uint8_t* d_desc, int64_t desc_stride, const int32_t* d_n, float minX, float maxX,
                          float minY, float maxY, void* stream) {
  if (!f || first_slot < 0 || count < 1 || first_slot + count > f->slots || !d_kps || !d_desc || !d_n) return DVM_ERR_INVALID;
  int rc = frame_bounds(f, minX, maxX, minY, maxY);
  if (rc != DVM_OK) return rc;
  launch_frame_build((hipStream_t)stream, reinterpret_cast<const dvm_keypoint_pod*>(d_kps), kps_stride, d_desc, desc_stride, 0,
                     d_n, f->view, first_slot, count);
  return hip_check(hipGetLastError(), "frame_build launch");
}

int dvm_match_window_top2(const dvm_frame* train, int slot, const uint8_t* skip, const uint8_t* qdesc, const float* qx,
                          const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax, int nq,
                          const int32_t* d_nq, dvm_match* out, int32_t* second_idx, int on_device, void* stream) {
  if (!train || slot < 0 || slot >= train->slots || nq < 0) return DVM_ERR_INVALID;
  if (nq == 0) return DVM_OK;
  if (!qdesc || !qx || !qy || !qr || !qmin || !qmax || !out) return DVM_ERR_INVALID;
  int rc = hip_check(hipSetDevice(train->device), "hipSetDevice");
  if (rc != DVM_OK) return rc;
  if (on_device) {
    launch_match_window((hipStream_t)stream, train->view, slot, skip, qdesc, qx, qy, qr, qmin, qmax, nq, d_nq, nq,
                        reinterpret_cast<dvm_match_pod*>(out), second_idx);
    return hip_check(hipGetLastError(), "match launch");
  }
  // host convenience path: stage queries, run, copy back
  const size_t qb = (size_t)nq;
  Stage st;
  // the query arrays are read once and the results written once: mapped; the skip flags are looked up per candidate: copied
  const int iD = st.in_mapped(qdesc, qb * 32), iX = st.in_mapped(qx, qb * 4), iY = st.in_mapped(qy, qb * 4), iR = st.in_mapped(qr, qb * 4),
            iMin = st.in_mapped(qmin, qb * 4), iMax = st.in_mapped(qmax, qb * 4), iS = st.in(skip, (size_t)train->cap),
            oM = st.out_mapped(out, qb * sizeof(dvm_match)), oS = second_idx ? st.out_mapped(second_idx, qb * 4) : -1;
  rc = st.upload();
  if (rc != DVM_OK) return rc;
  launch_match_window(st.stream(), train->view, slot, st.ptr<uint8_t>(iS), st.ptr<uint8_t>(iD), st.ptr<float>(iX), st.ptr<float>(iY),
                      st.ptr<float>(iR), st.ptr<int32_t>(iMin), st.ptr<int32_t>(iMax), nq, nullptr, nq, st.ptr<dvm_match_pod>(oM),
                      second_idx ? st.ptr<int32_t>(oS) : nullptr);
  rc = hip_check(hipGetLastError(), "match launch");
  return rc == DVM_OK ? st.download() : rc;
}
int dvm_match_window_ranked(const dvm_frame* train, int slot, const uint8_t* skip, const uint8_t* qdesc, const float* qx, const float* qy,
                            const float* qr, const int32_t* qmin, const int32_t* qmax, int nq, uint32_t* ranked, int on_device, void* stream) {
  if (!train || slot < 0 || slot >= train->slots || nq < 0) return DVM_ERR_INVALID;
  if (nq == 0) return DVM_OK;
  if (!qdesc || !qx || !qy || !qr || !qmin || !qmax || !ranked) return DVM_ERR_INVALID;
  int rc = hip_check(hipSetDevice(train->device), "hipSetDevice");
  if (rc != DVM_OK) return rc;
  if (on_device) {
    launch_match_window_ranked((hipStream_t)stream, train->view, slot, skip, qdesc, qx, qy, qr, qmin, qmax, nq, ranked);
    return hip_check(hipGetLastError(), "match launch");
  }
  const size_t qb = (size_t)nq;
  Stage st;   // the query arrays are read once and the results written once: mapped; the skip flags are looked up per candidate: copied
  const int iD = st.in_mapped(qdesc, qb * 32), iX = st.in_mapped(qx, qb * 4), iY = st.in_mapped(qy, qb * 4), iR = st.in_mapped(qr, qb * 4),
            iMin = st.in_mapped(qmin, qb * 4), iMax = st.in_mapped(qmax, qb * 4), iS = st.in(skip, (size_t)train->cap),
            oR = st.out_mapped(ranked, qb * 16);
  rc = st.upload();
  if (rc != DVM_OK) return rc;
  launch_match_window_ranked(st.stream(), train->view, slot, st.ptr<uint8_t>(iS), st.ptr<uint8_t>(iD), st.ptr<float>(iX), st.ptr<float>(iY),
                             st.ptr<float>(iR), st.ptr<int32_t>(iMin), st.ptr<int32_t>(iMax), nq, st.ptr<uint32_t>(oR));
  rc = hip_check(hipGetLastError(), "match launch");
  return rc == DVM_OK ? st.download() : rc;
}
int dvm_frame_build_match_window_ranked(dvm_frame* f, int slot, const dvm_keypoint* kps, const uint8_t* desc, int n, float minX,
                                        float maxX, float minY, float maxY, const uint8_t* skip, const uint8_t* qdesc,
                                        const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                                        const int32_t* qmax, int nq, uint32_t* ranked, int kps_on_device) {
  if (!f || slot < 0 || slot >= f->slots || n < 0 || n > f->cap || (n && (!kps || !desc)) || nq < 0) return DVM_ERR_INVALID;
  if (nq && (!qdesc || !qx || !qy || !qr || !qmin || !qmax || !ranked)) return DVM_ERR_INVALID;
  int rc = frame_bounds(f, minX, maxX, minY, maxY);
  if (rc != DVM_OK) return rc;
  rc = hip_check(hipSetDevice(f->device), "hipSetDevice");
  if (rc != DVM_OK) return rc;
  const size_t qb = (size_t)nq;
  Stage st;
  // (kps_on_device: the frame's keypoints + descriptors are device arrays already -- where the extractor left them)
  const int iK = kps_on_device ? -1 : st.in_mapped(kps, (size_t)n * sizeof(dvm_keypoint)), iDs = kps_on_device ? -1 : st.in_mapped(desc, (size_t)n * 32);
  const int iD = st.in_mapped(qdesc, qb * 32), iX = st.in_mapped(qx, qb * 4), iY = st.in_mapped(qy, qb * 4), iR = st.in_mapped(qr, qb * 4),
            iMin = st.in_mapped(qmin, qb * 4), iMax = st.in_mapped(qmax, qb * 4), iS = st.in(skip, (size_t)f->cap),
            oR = st.out_mapped(ranked, qb * 16);
  rc = st.upload();
  if (rc != DVM_OK) return rc;
  launch_frame_build(st.stream(), kps_on_device ? reinterpret_cast<const dvm_keypoint_pod*>(kps) : st.ptr<dvm_keypoint_pod>(iK), 0,
                     kps_on_device ? desc : st.ptr<uint8_t>(iDs), 0, n, nullptr, f->view, slot, 1);
  if (nq)
    launch_match_window_ranked(st.stream(), f->view, slot, st.ptr<uint8_t>(iS), st.ptr<uint8_t>(iD), st.ptr<float>(iX), st.ptr<float>(iY),
                               st.ptr<float>(iR), st.ptr<int32_t>(iMin), st.ptr<int32_t>(iMax), nq, st.ptr<uint32_t>(oR));
  rc = hip_check(hipGetLastError(), "frame_build + match launch");
  return rc == DVM_OK ? st.download() : rc;   // (download() synchronises the stream: the grid is complete for any later consumer)
}
void* dvm_thread_stream(int device) {
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  StageCtx& c = stage_ctx();
  return c.ensure(16) == DVM_OK ? (void*)c.s : nullptr;
}
int dvm_match_window(const dvm_frame* train, int slot, const uint8_t* skip, const uint8_t* qdesc, const float* qx,
                     const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax, int nq,
                     const int32_t* d_nq, dvm_match* out, int on_device, void* stream) {
  return dvm_match_window_top2(train, slot, skip, qdesc, qx, qy, qr, qmin, qmax, nq, d_nq, out, nullptr, on_device, stream);
}

int dvm_is_in_frustum(const dvm_frustum_frame* frame, const float* P, const float* normal, const float* min_dist,
                      const float* max_dist, int n, float viewing_cos_limit, dvm_track_point* out, int on_device, void* stream) {
  static_assert(sizeof(dvm_frustum_frame) == sizeof(FrustumFrame) && sizeof(dvm_track_point) == sizeof(TrackPoint), "layout");
  if (!frame || n < 0) return DVM_ERR_INVALID;
  if (n == 0) return DVM_OK;
  if (!P || !normal || !min_dist || !max_dist || !out) return DVM_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  FrustumFrame F;
  std::memcpy(&F, frame, sizeof(F));
  if (on_device) {
    launch_is_in_frustum((hipStream_t)stream, F, P, normal, min_dist, max_dist, n, viewing_cos_limit, reinterpret_cast<TrackPoint*>(out));
    return hip_check(hipGetLastError(), "is_in_frustum launch");
  }
  const size_t N = (size_t)n;
  Stage st;
  const int iP = st.in(P, N * 12), iN = st.in(normal, N * 12), imin = st.in(min_dist, N * 4), imax = st.in(max_dist, N * 4),
            oT = st.out(out, N * sizeof(TrackPoint));
  int rc = st.upload();
  if (rc != DVM_OK) return rc;
  launch_is_in_frustum(nullptr, F, st.ptr<float>(iP), st.ptr<float>(iN), st.ptr<float>(imin), st.ptr<float>(imax), n, viewing_cos_limit,
                       st.ptr<TrackPoint>(oT));
  rc = hip_check(hipGetLastError(), "launch");
  return rc == DVM_OK ? st.download() : rc;
}

int dvm_triangulate_matches(const dvm_tri_pair* pair, const dvm_keypoint* kps1, int n1, const dvm_keypoint* kps2, int n2,
                            const int32_t* pairs, int n, const float* sigma2_1, const float* sigma2_2, const float* scale_factors_1,
                            const float* scale_factors_2, float* x3D, int32_t* status, int on_device, void* stream) {
  static_assert(sizeof(dvm_tri_pair) == sizeof(TriPair) && sizeof(dvm_keypoint) == sizeof(dvm_keypoint_pod), "layout");
  if (!pair || n < 0 || n1 < 0 || n2 < 0) return DVM_ERR_INVALID;
  if (n == 0) return DVM_OK;
  if (!kps1 || !kps2 || !pairs || !sigma2_1 || !sigma2_2 || !scale_factors_1 || !scale_factors_2 || !x3D || !status) return DVM_ERR_INVALID;
  if (pair->n_levels <= 0 || pair->n_levels > 64) { set_error("dvm_triangulate_matches: n_levels out of range"); return DVM_ERR_INVALID; }
  if (!(pair->K1[0] != 0.0f) || !(pair->K1[1] != 0.0f) || !(pair->K2[0] != 0.0f) || !(pair->K2[1] != 0.0f)) {
    set_error("dvm_triangulate_matches: zero focal length");
    return DVM_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  TriPair P;
  std::memcpy(&P, pair, sizeof(P));
  if (on_device) {
    launch_triangulate_matches((hipStream_t)stream, P, reinterpret_cast<const dvm_keypoint_pod*>(kps1), n1, reinterpret_cast<const dvm_keypoint_pod*>(kps2), n2,
                               pairs, n, sigma2_1, sigma2_2, scale_factors_1, scale_factors_2, x3D, status);
    return hip_check(hipGetLastError(), "triangulate_matches launch");
  }
  for (int m = 0; m < n; m++)
    if (pairs[2 * m] < 0 || pairs[2 * m] >= n1 || pairs[2 * m + 1] < 0 || pairs[2 * m + 1] >= n2) {
      set_error("dvm_triangulate_matches: match index out of range");
      return DVM_ERR_INVALID;
    }
  const size_t N = (size_t)n, L = (size_t)P.n_levels;
  Stage st;
  const int iK1 = st.in(kps1, (size_t)n1 * sizeof(dvm_keypoint)), iK2 = st.in(kps2, (size_t)n2 * sizeof(dvm_keypoint)), iP = st.in(pairs, N * 8),
            iS1 = st.in(sigma2_1, L * 4), iS2 = st.in(sigma2_2, L * 4), iF1 = st.in(scale_factors_1, L * 4), iF2 = st.in(scale_factors_2, L * 4),
            oX = st.out(x3D, N * 12), oS = st.out(status, N * 4);
  int rc = st.upload();
  if (rc != DVM_OK) return rc;
  launch_triangulate_matches(nullptr, P, st.ptr<dvm_keypoint_pod>(iK1), n1, st.ptr<dvm_keypoint_pod>(iK2), n2, st.ptr<int32_t>(iP), n, st.ptr<float>(iS1),
                             st.ptr<float>(iS2), st.ptr<float>(iF1), st.ptr<float>(iF2), st.ptr<float>(oX), st.ptr<int32_t>(oS));
  rc = hip_check(hipGetLastError(), "launch");
  return rc == DVM_OK ? st.download() : rc;
}

int dvm_undistort_keypoints(const dvm_distortion* cam, const dvm_keypoint* kps_in, dvm_keypoint* kps_out, int n, int on_device, void* stream) {
  static_assert(sizeof(dvm_distortion) == sizeof(dvm_undistort::Camera) && sizeof(dvm_keypoint) == 28, "layout");
  if (!cam || n < 0) return DVM_ERR_INVALID;
  if (n == 0) return DVM_OK;
  if (!kps_in || !kps_out) return DVM_ERR_INVALID;
  if (!(cam->fx != 0.0f) || !(cam->fy != 0.0f)) { set_error("dvm_undistort_keypoints: zero focal length"); return DVM_ERR_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  dvm_undistort::Camera C;
  std::memcpy(&C, cam, sizeof(C));
  if (on_device) {
    launch_undistort_keypoints((hipStream_t)stream, C, reinterpret_cast<const float*>(kps_in), reinterpret_cast<float*>(kps_out), n);
    return hip_check(hipGetLastError(), "undistort launch");
  }
  Stage st;
  const int iK = st.in(kps_in, (size_t)n * 28), oK = st.out(kps_out, (size_t)n * 28);
  int rc = st.upload();
  if (rc != DVM_OK) return rc;
  launch_undistort_keypoints(nullptr, C, st.ptr<float>(iK), st.ptr<float>(oK), n);
  rc = hip_check(hipGetLastError(), "undistort launch");
  return rc == DVM_OK ? st.download() : rc;
}

int dvm_image_bounds(const dvm_distortion* cam, int cols, int rows, float bounds[4]) {
  if (!cam || !bounds || cols <= 0 || rows <= 0) return DVM_ERR_INVALID;
  if (cam->k1 == 0.0f) {   // Frame.cc:843-847
    bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows;
    return DVM_OK;
  }
  dvm_keypoint c[4];
  std::memset(c, 0, sizeof(c));
  c[0].x = 0.0f; c[0].y = 0.0f; c[1].x = (float)cols; c[1].y = 0.0f; c[2].x = 0.0f; c[2].y = (float)rows; c[3].x = (float)cols; c[3].y = (float)rows;
  const int rc = dvm_undistort_keypoints(cam, c, c, 4, 0, nullptr);
  if (rc != DVM_OK) return rc;
  bounds[0] = std::min(c[0].x, c[2].x); bounds[1] = std::max(c[1].x, c[3].x);
  bounds[2] = std::min(c[0].y, c[1].y); bounds[3] = std::max(c[2].y, c[3].y);
  return DVM_OK;
}

int dvm_match_lists(const uint8_t* tdesc, int nt, const uint8_t* qdesc, int nq, const int32_t* off, const int32_t* cand,
                    dvm_match* out, int on_device, void* stream) {
  if (nq < 0 || nt < 0) return DVM_ERR_INVALID;
  if (nq == 0) return DVM_OK;
  if (!tdesc || !qdesc || !off || !cand || !out) return DVM_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  if (on_device) {
    launch_match_lists((hipStream_t)stream, tdesc, qdesc, off, cand, nq, reinterpret_cast<dvm_match_pod*>(out));
    return hip_check(hipGetLastError(), "match_lists launch");
  }
  const int ncand = off[nq];
  if (ncand < 0) return DVM_ERR_INVALID;
  Stage st;
  const int iT = st.in(tdesc, (size_t)nt * 32), iQ = st.in(qdesc, (size_t)nq * 32), iO = st.in(off, (size_t)(nq + 1) * 4),
            iC = st.in(cand, (size_t)ncand * 4), oM = st.out(out, (size_t)nq * sizeof(dvm_match));
  int rc = st.upload();
  if (rc != DVM_OK) return rc;
  launch_match_lists(nullptr, st.ptr<uint8_t>(iT), st.ptr<uint8_t>(iQ), st.ptr<int32_t>(iO), st.ptr<int32_t>(iC), nq, st.ptr<dvm_match_pod>(oM));
  rc = hip_check(hipGetLastError(), "match_lists launch");
  return rc == DVM_OK ? st.download() : rc;
}

int dvm_project_search(const dvm_frame* train, int slot, const uint8_t* skip, const dvm_kf_camera* cam, const float* P,
                       const float* normal, const float* min_dist, const float* max_dist, const uint8_t* desc,
                       const uint8_t* valid, int n, float th, const float* scale_factors, const float* gate_inv_sigma2,
                       double gate, dvm_match* out, dvm_projection* proj, int on_device, void* stream) {
  static_assert(sizeof(dvm_kf_camera) + 4 == sizeof(ProjectCam) && sizeof(dvm_projection) == sizeof(Projection), "layout");
  if (!train || slot < 0 || slot >= train->slots || !cam || n < 0) return DVM_ERR_INVALID;
  if (n == 0) return DVM_OK;
  if (!P || !normal || !min_dist || !max_dist || !desc || !scale_factors || !out) return DVM_ERR_INVALID;
  if (cam->n_levels < 1 || cam->n_levels > 32) return DVM_ERR_INVALID;
  int rc = hip_check(hipSetDevice(train->device), "hipSetDevice");
  if (rc != DVM_OK) return rc;
  ProjectCam C;
  std::memcpy(&C, cam, sizeof(dvm_kf_camera));
  C.th = th;
  if (on_device) {
    launch_project_search((hipStream_t)stream, train->view, slot, skip, C, P, normal, min_dist, max_dist, desc, valid, n,
                          scale_factors, gate_inv_sigma2, gate, reinterpret_cast<dvm_match_pod*>(out), reinterpret_cast<Projection*>(proj));
    return hip_check(hipGetLastError(), "project_search launch");
  }
  Stage st;
  const size_t N = (size_t)n, L = (size_t)cam->n_levels;
  const int iP = st.in(P, N * 12), iN = st.in(normal, N * 12), imin = st.in(min_dist, N * 4), imax = st.in(max_dist, N * 4),
            iD = st.in(desc, N * 32), iV = st.in(valid, N), iS = st.in(scale_factors, L * 4), iG = st.in(gate_inv_sigma2, L * 4),
            iK = st.in(skip, (size_t)train->cap), oM = st.out(out, N * sizeof(dvm_match)), oP = st.out(proj, N * sizeof(dvm_projection));
  rc = st.upload();
  if (rc != DVM_OK) return rc;
  launch_project_search(nullptr, train->view, slot, st.ptr<uint8_t>(iK), C, st.ptr<float>(iP), st.ptr<float>(iN), st.ptr<float>(imin),
                        st.ptr<float>(imax), st.ptr<uint8_t>(iD), st.ptr<uint8_t>(iV), n, st.ptr<float>(iS), st.ptr<float>(iG), gate,
                        st.ptr<dvm_match_pod>(oM), st.ptr<Projection>(oP));
  rc = hip_check(hipGetLastError(), "project_search launch");
  return rc == DVM_OK ? st.download() : rc;
}

int dvm_match_triangulation(const uint8_t* desc1, const dvm_keypoint* kps1, int n1, const int32_t* qidx, int nq,
                            const uint8_t* desc2, const dvm_keypoint* kps2, int n2, const int32_t* off, const int32_t* cand,
                            const float* F12, const float* ep, int coarse, const float* scale_factors2,
                            const float* level_sigma2_2, int nlevels, int32_t* best_idx, int32_t* best_dist, int on_device,
                            void* stream) {
  if (nq < 0 || n1 < 0 || n2 < 0 || nlevels < 1) return DVM_ERR_INVALID;
  if (nq == 0) return DVM_OK;
  if (!desc1 || !kps1 || !qidx || !desc2 || !kps2 || !off || !cand || !F12 || !ep || !scale_factors2 || !level_sigma2_2 || !best_idx ||
      !best_dist)
    return DVM_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  TriGeom G;
  std::memcpy(G.F12, F12, 36);
  G.ep[0] = ep[0]; G.ep[1] = ep[1];
  G.coarse = coarse != 0; G.th_low = 50;   // ORBmatcher::TH_LOW
  if (on_device) {
    launch_match_triangulation((hipStream_t)stream, desc1, reinterpret_cast<const dvm_keypoint_pod*>(kps1), qidx, nq, desc2,
                               reinterpret_cast<const dvm_keypoint_pod*>(kps2), off, cand, G, scale_factors2, level_sigma2_2, best_idx, best_dist);
    return hip_check(hipGetLastError(), "match_triangulation launch");
  }
  const int ncand = off[nq];
  if (ncand < 0) return DVM_ERR_INVALID;
  Stage st;
  const int i1 = st.in(desc1, (size_t)n1 * 32), k1 = st.in(kps1, (size_t)n1 * sizeof(dvm_keypoint)), iq = st.in(qidx, (size_t)nq * 4),
            i2 = st.in(desc2, (size_t)n2 * 32), k2 = st.in(kps2, (size_t)n2 * sizeof(dvm_keypoint)), io = st.in(off, (size_t)(nq + 1) * 4),
            ic = st.in(cand, (size_t)ncand * 4), is = st.in(scale_factors2, (size_t)nlevels * 4), iv = st.in(level_sigma2_2, (size_t)nlevels * 4),
            oi = st.out(best_idx, (size_t)nq * 4), od = st.out(best_dist, (size_t)nq * 4);
  int rc = st.upload();
  if (rc != DVM_OK) return rc;
  launch_match_triangulation(nullptr, st.ptr<uint8_t>(i1), st.ptr<dvm_keypoint_pod>(k1), st.ptr<int32_t>(iq), nq, st.ptr<uint8_t>(i2),
                             st.ptr<dvm_keypoint_pod>(k2), st.ptr<int32_t>(io), st.ptr<int32_t>(ic), G, st.ptr<float>(is), st.ptr<float>(iv),
                             st.ptr<int32_t>(oi), st.ptr<int32_t>(od));
  rc = hip_check(hipGetLastError(), "match_triangulation launch");
  return rc == DVM_OK ? st.download() : rc;
}

struct dvm_bowdb {
  int device = 0;
  std::vector<int64_t> off;               // per slot: first word (64-bit: a long session stores more than 2^31 words)
  std::vector<int32_t> len;               // per slot (len < 0: erased)
  std::vector<int32_t> live;              // slots with len >= 0, ascending: what a query launches over
  size_t used = 0, cap = 0, dead = 0;     // words stored (live + erased) / capacity of d_ids, d_vals / words of erased slots
  int32_t* d_ids = nullptr;
  double* d_vals = nullptr;
  int64_t* d_off = nullptr;
  int32_t *d_len = nullptr, *d_live = nullptr, *d_common = nullptr, *d_first = nullptr;
  float* d_score = nullptr;
  size_t slot_cap = 0;
  bool meta_dirty = true;
};

void dvm_bowdb_destroy(dvm_bowdb* db) {
  if (!db) return;
  hipSetDevice(db->device);
  for (void* p : {(void*)db->d_ids, (void*)db->d_vals, (void*)db->d_off, (void*)db->d_len, (void*)db->d_live, (void*)db->d_common, (void*)db->d_first, (void*)db->d_score})
    if (p) hipFree(p);
  delete db;
}

int dvm_bowdb_create(int device, dvm_bowdb** out) {
  if (!out) return DVM_ERR_INVALID;
  *out = nullptr;
  int rc = need_device(device);
  if (rc != DVM_OK) return rc;
  dvm_bowdb* db = new dvm_bowdb;
  db->device = device;
  *out = db;
  return DVM_OK;
}

int dvm_bowdb_size(const dvm_bowdb* db) { return db ? (int)db->off.size() : 0; }

int dvm_bowdb_add(dvm_bowdb* db, const int32_t* word_ids, const double* values, int n, int32_t* slot) {
  if (!db || n < 0 || (n > 0 && (!word_ids || !values))) return DVM_ERR_INVALID;
  for (int i = 1; i < n; i++)
    if (word_ids[i] <= word_ids[i - 1]) { set_error("bowdb: word ids must be strictly ascending"); return DVM_ERR_INVALID; }
  DVM_HIP(hipSetDevice(db->device));
  // Erased keyframes leave their words behind (the reference culls keyframes all the time): once the dead words outnumber
  // the live ones, repack the live slots' ranges into fresh arrays, in slot order.  Slot numbers never change.
  const bool repack = db->dead >= ((size_t)1 << 16) && db->dead > db->used - db->dead;
  if (repack || db->used + (size_t)n > db->cap) {   // grow (doubling) and / or repack, keeping what is stored
    const size_t live_words = db->used - db->dead;
    const size_t keep = repack ? live_words : db->used;
    const size_t ncap = std::max<size_t>(std::max<size_t>(repack ? keep * 2 : db->cap * 2, keep + (size_t)n), 1 << 16);
    int32_t* ni = nullptr; double* nv = nullptr;
    DVM_HIP(hipMalloc(&ni, ncap * 4));
    if (hipMalloc(&nv, ncap * 8) != hipSuccess) { hipFree(ni); set_error("hipMalloc(bowdb)"); return DVM_ERR_HIP; }
    hipError_t e = hipSuccess;
    if (repack) {
      // the compacted offsets go to a scratch vector and replace db->off only once every copy has landed: a failure midway
      // leaves the database exactly as it was (old arrays, old offsets)
      std::vector<int64_t> noff(db->off);
      size_t w = 0;
      for (size_t k = 0; k < db->off.size() && e == hipSuccess; k++) {
        if (db->len[k] <= 0) { if (db->len[k] == 0) noff[k] = (int64_t)w; continue; }
        const size_t L = (size_t)db->len[k];
        e = hipMemcpyAsync(ni + w, db->d_ids + db->off[k], L * 4, hipMemcpyDeviceToDevice, nullptr);
        if (e == hipSuccess) e = hipMemcpyAsync(nv + w, db->d_vals + db->off[k], L * 8, hipMemcpyDeviceToDevice, nullptr);
        noff[k] = (int64_t)w;
        w += L;
      }
      if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
      if (e == hipSuccess) { db->off.swap(noff); db->used = w; db->dead = 0; }
    } else if (db->used) {
      e = hipMemcpy(ni, db->d_ids, db->used * 4, hipMemcpyDeviceToDevice);
      if (e == hipSuccess) e = hipMemcpy(nv, db->d_vals, db->used * 8, hipMemcpyDeviceToDevice);
    }
    if (e != hipSuccess) { hipFree(ni); hipFree(nv); set_error(std::string("bowdb grow / repack: ") + hipGetErrorString(e)); return DVM_ERR_HIP; }
    if (db->d_ids) hipFree(db->d_ids);
    if (db->d_vals) hipFree(db->d_vals);
    db->d_ids = ni; db->d_vals = nv; db->cap = ncap;
    db->meta_dirty = true;
  }
  if (n) {
    DVM_HIP(hipMemcpy(db->d_ids + db->used, word_ids, (size_t)n * 4, hipMemcpyHostToDevice));
    DVM_HIP(hipMemcpy(db->d_vals + db->used, values, (size_t)n * 8, hipMemcpyHostToDevice));
  }
  if (slot) *slot = (int32_t)db->off.size();
  db->off.push_back((int64_t)db->used);
  db->len.push_back(n);
  db->used += (size_t)n;
  db->meta_dirty = true;
  return DVM_OK;
}

int dvm_bowdb_erase(dvm_bowdb* db, int32_t slot) {
  if (!db || slot < 0 || slot >= (int32_t)db->off.size()) return DVM_ERR_INVALID;
  if (db->len[slot] > 0) db->dead += (size_t)db->len[slot];
  db->len[slot] = -1;
  db->meta_dirty = true;
  return DVM_OK;
}

int dvm_bowdb_query(dvm_bowdb* db, const int32_t* word_ids, const double* values, int n, int32_t* common, int32_t* first_word,
                    float* score) {
  if (!db || n < 0 || (n > 0 && (!word_ids || !values)) || !common || !first_word || !score) return DVM_ERR_INVALID;
  for (int i = 1; i < n; i++)   // the kernel binary-searches the query ids
    if (word_ids[i] <= word_ids[i - 1]) { set_error("bowdb: query word ids must be strictly ascending"); return DVM_ERR_INVALID; }
  const int N = (int)db->off.size();
  if (N == 0) return DVM_OK;
  DVM_HIP(hipSetDevice(db->device));
  if ((size_t)N > db->slot_cap) {
    for (void* p : {(void*)db->d_off, (void*)db->d_len, (void*)db->d_live, (void*)db->d_common, (void*)db->d_first, (void*)db->d_score})
      if (p) hipFree(p);
    db->d_off = nullptr; db->d_len = db->d_live = db->d_common = db->d_first = nullptr; db->d_score = nullptr;
    db->slot_cap = std::max<size_t>((size_t)N * 2, 1024);
    DVM_HIP(hipMalloc(&db->d_off, db->slot_cap * 8)); DVM_HIP(hipMalloc(&db->d_len, db->slot_cap * 4));
    DVM_HIP(hipMalloc(&db->d_live, db->slot_cap * 4));
    DVM_HIP(hipMalloc(&db->d_common, db->slot_cap * 4)); DVM_HIP(hipMalloc(&db->d_first, db->slot_cap * 4));
    DVM_HIP(hipMalloc(&db->d_score, db->slot_cap * 4));
    db->meta_dirty = true;
  }
  if (db->meta_dirty) {
    db->live.clear();
    for (int k = 0; k < N; k++) if (db->len[k] >= 0) db->live.push_back(k);
    DVM_HIP(hipMemcpy(db->d_off, db->off.data(), (size_t)N * 8, hipMemcpyHostToDevice));
    DVM_HIP(hipMemcpy(db->d_len, db->len.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    if (!db->live.empty()) DVM_HIP(hipMemcpy(db->d_live, db->live.data(), db->live.size() * 4, hipMemcpyHostToDevice));
    db->meta_dirty = false;
  }
  Stage st;
  const int iq = st.in(word_ids, (size_t)n * 4), iv = st.in(values, (size_t)n * 8);
  int rc = st.upload();
  if (rc != DVM_OK) return rc;
  // erased slots answer (common -1, first word -1, score 0) without costing a wavefront: the launch covers the live slots only
  DVM_HIP(hipMemsetAsync(db->d_common, 0xFF, (size_t)N * 4, nullptr));
  DVM_HIP(hipMemsetAsync(db->d_first, 0xFF, (size_t)N * 4, nullptr));
  DVM_HIP(hipMemsetAsync(db->d_score, 0, (size_t)N * 4, nullptr));
  launch_bowdb_query(nullptr, db->d_off, db->d_len, db->d_live, (int)db->live.size(), db->d_ids, db->d_vals, st.ptr<int32_t>(iq), st.ptr<double>(iv), n,
                     db->d_common, db->d_first, db->d_score);
  rc = hip_check(hipGetLastError(), "bowdb_query launch");
  if (rc != DVM_OK) return rc;
  DVM_HIP(hipMemcpy(common, db->d_common, (size_t)N * 4, hipMemcpyDeviceToHost));
  DVM_HIP(hipMemcpy(first_word, db->d_first, (size_t)N * 4, hipMemcpyDeviceToHost));
  DVM_HIP(hipMemcpy(score, db->d_score, (size_t)N * 4, hipMemcpyDeviceToHost));
  return DVM_OK;
}

int dvm_bowdb_stats(const dvm_bowdb* db, int64_t out[4]) {
  if (!db || !out) return DVM_ERR_INVALID;
  int64_t nlive = 0;
  for (int32_t l : db->len) nlive += l >= 0;
  out[0] = (int64_t)db->off.size(); out[1] = nlive; out[2] = (int64_t)db->used; out[3] = (int64_t)db->cap;
  return DVM_OK;
}

int dvm_distinctive_descriptors(const uint8_t* desc, const int32_t* off, int n_points, int32_t* best_idx, int32_t* best_median,
                                int on_device, void* stream) {
  if (n_points < 0) return DVM_ERR_INVALID;
  if (n_points == 0) return DVM_OK;
  if (!desc || !off || !best_idx || !best_median) return DVM_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  if (on_device) {
    launch_distinctive((hipStream_t)stream, desc, off, n_points, best_idx, best_median);
    return hip_check(hipGetLastError(), "distinctive launch");
  }
  const int total = off[n_points];
  if (total < 0) return DVM_ERR_INVALID;
  Stage st;
  const int iD = st.in(desc, (size_t)total * 32), iO = st.in(off, (size_t)(n_points + 1) * 4), oI = st.out(best_idx, (size_t)n_points * 4),
            oM = st.out(best_median, (size_t)n_points * 4);
  int rc = st.upload();
  if (rc != DVM_OK) return rc;
  launch_distinctive(nullptr, st.ptr<uint8_t>(iD), st.ptr<int32_t>(iO), n_points, st.ptr<int32_t>(oI), st.ptr<int32_t>(oM));
  rc = hip_check(hipGetLastError(), "distinctive launch");
  return rc == DVM_OK ? st.download() : rc;
}

struct dvm_vocab {
  int device = 0, n_nodes = 0, L = 0;
  int32_t *child_off = nullptr, *children = nullptr, *word_id = nullptr;
  uint8_t* desc = nullptr;
  double* weight = nullptr;
};

void dvm_vocab_destroy(dvm_vocab* v) {
  if (!v) return;
  hipSetDevice(v->device);
  for (void* p : {(void*)v->child_off, (void*)v->children, (void*)v->word_id, (void*)v->desc, (void*)v->weight})
    if (p) hipFree(p);
  delete v;
}

int dvm_vocab_create(int device, int n_nodes, const int32_t* child_off, const int32_t* children, const uint8_t* desc,
                     const double* weight, const int32_t* word_id, int L, dvm_vocab** out) {
  if (!out) return DVM_ERR_INVALID;
  *out = nullptr;
  if (n_nodes < 1 || !child_off || !children || !desc || !weight || !word_id || L < 0) return DVM_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  if (device < 0 || device >= ndev) return DVM_ERR_INVALID;
  const int nch = child_off[n_nodes];
  if (child_off[0] != 0 || nch < 0) { set_error("vocabulary: bad child_off"); return DVM_ERR_INVALID; }
  for (int i = 0; i < n_nodes; i++) if (child_off[i + 1] < child_off[i]) { set_error("vocabulary: child_off not monotone"); return DVM_ERR_INVALID; }
  for (int c = 0; c < nch; c++) if (children[c] <= 0 || children[c] >= n_nodes) { set_error("vocabulary: child id out of range"); return DVM_ERR_INVALID; }
  DVM_HIP(hipSetDevice(device));
  dvm_vocab* v = new dvm_vocab;
  v->device = device; v->n_nodes = n_nodes; v->L = L;
  int rc = DVM_OK;
  auto up = [&](auto** dst, const void* src, size_t bytes) {
    if (rc != DVM_OK) return;
    void* p = nullptr;
    rc = hip_check(hipMalloc(&p, std::max<size_t>(bytes, 16)), "hipMalloc(vocab)");
    if (rc == DVM_OK && bytes) rc = hip_check(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice), "memcpy(vocab)");
    *dst = static_cast<std::remove_reference_t<decltype(**dst)>*>(p);
  };
  up(&v->child_off, child_off, (size_t)(n_nodes + 1) * 4);
  up(&v->children, children, (size_t)nch * 4);
  up(&v->desc, desc, (size_t)n_nodes * 32);
  up(&v->weight, weight, (size_t)n_nodes * 8);
  up(&v->word_id, word_id, (size_t)n_nodes * 4);
  if (rc != DVM_OK) { dvm_vocab_destroy(v); return rc; }
  *out = v;
  return DVM_OK;
}

int dvm_vocab_transform(const dvm_vocab* v, const uint8_t* features, int n, int levelsup, int32_t* word_id, int32_t* node_id,
                        double* weight, int on_device, void* stream) {
  if (!v || n < 0) return DVM_ERR_INVALID;
  if (n == 0) return DVM_OK;
  if (!features || !word_id || !node_id || !weight) return DVM_ERR_INVALID;
  DVM_HIP(hipSetDevice(v->device));
  if (on_device) {
    launch_vocab_transform((hipStream_t)stream, v->child_off, v->children, v->desc, v->weight, v->word_id, v->L, features, n, levelsup,
                           word_id, node_id, weight);
    return hip_check(hipGetLastError(), "vocab_transform launch");
  }
  const size_t b_f = (size_t)n * 32, b_i = ((size_t)n * 4 + 15) & ~(size_t)15, b_w = (size_t)n * 8;
  uint8_t* d = nullptr;
  int rc = hip_check(hipMalloc(&d, b_f + 2 * b_i + b_w), "hipMalloc");
  if (rc != DVM_OK) return rc;
  rc = hip_check(hipMemcpy(d, features, b_f, hipMemcpyHostToDevice), "memcpy");
  int32_t* dw = reinterpret_cast<int32_t*>(d + b_f);
  int32_t* dn = reinterpret_cast<int32_t*>(d + b_f + b_i);
  double* dwt = reinterpret_cast<double*>(d + b_f + 2 * b_i);
  if (rc == DVM_OK) {
    launch_vocab_transform(nullptr, v->child_off, v->children, v->desc, v->weight, v->word_id, v->L, d, n, levelsup, dw, dn, dwt);
    rc = hip_check(hipGetLastError(), "vocab_transform launch");
  }
  if (rc == DVM_OK) rc = hip_check(hipMemcpy(word_id, dw, (size_t)n * 4, hipMemcpyDeviceToHost), "memcpy");
  if (rc == DVM_OK) rc = hip_check(hipMemcpy(node_id, dn, (size_t)n * 4, hipMemcpyDeviceToHost), "memcpy");
  if (rc == DVM_OK) rc = hip_check(hipMemcpy(weight, dwt, b_w, hipMemcpyDeviceToHost), "memcpy");
  hipFree(d);
  return rc;
}

int dvm_match_frames_batch(const dvm_frame* train, int first_slot, int count, const dvm_keypoint* d_kps,
                           int64_t kps_stride, const uint8_t* d_desc, int64_t desc_stride, const int32_t* d_n,
                           const dvm_keypoint* d_carry_kps, const uint8_t* d_carry_desc, const int32_t* d_carry_n,
                           int cap, float th, const float* d_scale_factors, int nlevels, dvm_match* d_out,
                           int64_t out_stride, int32_t* d_nq_out, void* stream) {
  if (!train || first_slot < 0 || count < 1 || first_slot + count > train->slots || cap < 1 || !d_out || !d_scale_factors)
    return DVM_ERR_INVALID;
  if (count > 1 && (!d_kps || !d_desc || !d_n)) return DVM_ERR_INVALID;
  PairQueries pq{};
  pq.kps = reinterpret_cast<const dvm_keypoint_pod*>(d_kps);
  pq.desc = d_desc; pq.n = d_n; pq.kps_stride = kps_stride; pq.desc_stride = desc_stride;
  pq.carry_kps = reinterpret_cast<const dvm_keypoint_pod*>(d_carry_kps);
  pq.carry_desc = d_carry_desc; pq.carry_n = d_carry_n; pq.n_out = d_nq_out; pq.cap = cap;
  launch_match_frames((hipStream_t)stream, train->view, first_slot, count, pq, th, d_scale_factors, nlevels,
                      reinterpret_cast<dvm_match_pod*>(d_out), out_stride);
  return hip_check(hipGetLastError(), "match_frames launch");
}


}  // extern "C"

// ------------------------------------------------------------------------------------------------ shared search service
// The grid build + ranked window search of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame) (dvm_frame_build_match_window_ranked) for
// several agents' tracking threads at once: calls that arrive together run as ONE launch pair (k_frame_build over the batch's grid slots,
// k_match_window_ranked with a frame per blockIdx.y) -- the third of the shared per-GPU services, see orb_pool.cpp / group_commit.h.
// A lane keeps every slot's arrays in MAPPED host memory (each is read or written once per call); the skip flags, which the search looks up
// per candidate, travel through a device buffer and only when some frame of the batch has any.

namespace {
struct MatchLane {
  dvm_frame* grid = nullptr;                 // max_batch slots
  dvm_keypoint* kps = nullptr; uint8_t* desc = nullptr; int32_t* n = nullptr;                      // [B][kp_cap], mapped
  uint8_t* qdesc = nullptr; float *qx = nullptr, *qy = nullptr, *qr = nullptr; int32_t *qmin = nullptr, *qmax = nullptr, *nq = nullptr;   // [B][q_cap]
  uint32_t* ranked = nullptr;                // [B][q_cap][4]
  int32_t* skip_on = nullptr;                // [B]
  uint8_t *h_skip = nullptr, *d_skip = nullptr;   // [B][kp_cap]: pinned / device
  hipStream_t stream = nullptr;
  float bounds[4] = {0, 0, 0, 0};
};
}  // namespace

struct dvm_match_pool {
  int device = 0, kp_cap = 0, q_cap = 0;
  GroupCommit gc;
  MatchLane lane[GroupCommit::kLanes];
};

static void match_lane_free(MatchLane& L) {
  for (void* p : {(void*)L.kps, (void*)L.desc, (void*)L.n, (void*)L.qdesc, (void*)L.qx, (void*)L.qy, (void*)L.qr, (void*)L.qmin, (void*)L.qmax, (void*)L.nq,
                  (void*)L.ranked, (void*)L.skip_on, (void*)L.h_skip})
    if (p) hipHostFree(p);
  if (L.d_skip) hipFree(L.d_skip);
  if (L.grid) dvm_frame_destroy(L.grid);
  if (L.stream) hipStreamDestroy(L.stream);
  L = MatchLane{};
}

extern "C" int dvm_match_pool_create(int device, int max_batch, int kp_cap, int q_cap, int window_us, dvm_match_pool** out) {
  if (!out || max_batch < 1 || max_batch > 256 || kp_cap < 1 || kp_cap > kFrameCap || q_cap < 1 || q_cap > 65536) {
    set_error("dvm_match_pool_create: bad parameters");
    return DVM_ERR_INVALID;
  }
  *out = nullptr;
  int rc = need_device(device);
  if (rc != DVM_OK) return rc;
  rc = hip_check(hipSetDevice(device), "hipSetDevice");
  if (rc != DVM_OK) return rc;
  dvm_match_pool* pool = new (std::nothrow) dvm_match_pool();
  if (!pool) return DVM_ERR_INVALID;
  pool->device = device; pool->kp_cap = kp_cap; pool->q_cap = (q_cap + 15) & ~15;
  pool->gc.max_batch = max_batch; pool->gc.window_us = window_us < 0 ? 20 : window_us;
  const size_t B = (size_t)max_batch, K = (size_t)kp_cap, Q = (size_t)pool->q_cap;
  for (MatchLane& L : pool->lane) {
    auto mapped = [&](void** p, size_t bytes) { if (rc == DVM_OK) rc = hip_check(hipHostMalloc(p, bytes, hipHostMallocMapped), "hipHostMalloc"); };
    if (rc == DVM_OK) rc = dvm_frame_create(device, kp_cap, max_batch, &L.grid);
    mapped(reinterpret_cast<void**>(&L.kps), B * K * sizeof(dvm_keypoint)); mapped(reinterpret_cast<void**>(&L.desc), B * K * 32);
    mapped(reinterpret_cast<void**>(&L.n), B * 4); mapped(reinterpret_cast<void**>(&L.qdesc), B * Q * 32);
    mapped(reinterpret_cast<void**>(&L.qx), B * Q * 4); mapped(reinterpret_cast<void**>(&L.qy), B * Q * 4); mapped(reinterpret_cast<void**>(&L.qr), B * Q * 4);
    mapped(reinterpret_cast<void**>(&L.qmin), B * Q * 4); mapped(reinterpret_cast<void**>(&L.qmax), B * Q * 4); mapped(reinterpret_cast<void**>(&L.nq), B * 4);
    mapped(reinterpret_cast<void**>(&L.ranked), B * Q * 16); mapped(reinterpret_cast<void**>(&L.skip_on), B * 4);
    if (rc == DVM_OK) rc = hip_check(hipHostMalloc(reinterpret_cast<void**>(&L.h_skip), B * K), "hipHostMalloc");
    if (rc == DVM_OK) rc = hip_check(hipMalloc(reinterpret_cast<void**>(&L.d_skip), B * K), "hipMalloc");
    if (rc == DVM_OK) rc = hip_check(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking), "stream");
  }
  if (rc != DVM_OK) {
    for (MatchLane& L : pool->lane) match_lane_free(L);
    delete pool;
    return rc;
  }
  *out = pool;
  return DVM_OK;
}
extern "C" void dvm_match_pool_destroy(dvm_match_pool* pool) {
  if (!pool) return;
  hipSetDevice(pool->device);
  for (MatchLane& L : pool->lane) { if (L.stream) hipStreamSynchronize(L.stream); match_lane_free(L); }
  delete pool;
}
extern "C" int dvm_match_pool_capacity(const dvm_match_pool* pool, int* kp_cap, int* q_cap) {
  if (!pool) return DVM_ERR_INVALID;
  if (kp_cap) *kp_cap = pool->kp_cap;
  if (q_cap) *q_cap = pool->q_cap;
  return DVM_OK;
}
extern "C" int dvm_match_pool_build_match_ranked(dvm_match_pool* pool, const dvm_keypoint* kps, const uint8_t* desc, int n, float minX, float maxX, float minY,
                                      float maxY, const uint8_t* skip, const uint8_t* qdesc, const float* qx, const float* qy, const float* qr,
                                      const int32_t* qmin, const int32_t* qmax, int nq, uint32_t* ranked, int* batch_size) {
  if (!pool || n < 0 || nq < 0 || (n && (!kps || !desc)) || (nq && (!qdesc || !qx || !qy || !qr || !qmin || !qmax || !ranked))) return DVM_ERR_INVALID;
  if (n > pool->kp_cap || nq > pool->q_cap) { set_error("dvm_match_pool: frame larger than the pool's capacity"); return DVM_ERR_CAPACITY; }
  if (!(maxX > minX) || !(maxY > minY)) { set_error("frame bounds empty"); return DVM_ERR_INVALID; }
  if (batch_size) *batch_size = 0;
  int64_t key[4] = {0, 0, 0, 0};   // the frames of a batch share the image bounds (one grid geometry per launch)
  std::memcpy(&key[0], &minX, 4); std::memcpy(&key[1], &maxX, 4); std::memcpy(&key[2], &minY, 4); std::memcpy(&key[3], &maxY, 4);
  int li = 0, slot = 0;
  int rc = pool->gc.join(key, [&](int l) {
    MatchLane& O = pool->lane[l];
    O.bounds[0] = minX; O.bounds[1] = maxX; O.bounds[2] = minY; O.bounds[3] = maxY;
    return 0;
  }, li, slot);
  if (rc != DVM_OK) return rc;
  MatchLane& L = pool->lane[li];
  const size_t K = (size_t)pool->kp_cap, Q = (size_t)pool->q_cap, s = (size_t)slot;
  if (n) { std::memcpy(L.kps + s * K, kps, (size_t)n * sizeof(dvm_keypoint)); std::memcpy(L.desc + s * K * 32, desc, (size_t)n * 32); }
  L.n[slot] = n; L.nq[slot] = nq;
  if (nq) {
    std::memcpy(L.qdesc + s * Q * 32, qdesc, (size_t)nq * 32);
    std::memcpy(L.qx + s * Q, qx, (size_t)nq * 4); std::memcpy(L.qy + s * Q, qy, (size_t)nq * 4); std::memcpy(L.qr + s * Q, qr, (size_t)nq * 4);
    std::memcpy(L.qmin + s * Q, qmin, (size_t)nq * 4); std::memcpy(L.qmax + s * Q, qmax, (size_t)nq * 4);
  }
  L.skip_on[slot] = skip ? 1 : 0;
  if (skip) std::memcpy(L.h_skip + s * K, skip, (size_t)n);
  if (pool->gc.arrive(li, slot)) {
    const int count = pool->gc.batch_count(li);
    int r = hip_check(hipSetDevice(pool->device), "hipSetDevice");
    if (r == DVM_OK) r = frame_bounds(L.grid, L.bounds[0], L.bounds[1], L.bounds[2], L.bounds[3]);
    if (r == DVM_OK) {
      bool any_skip = false;
      for (int b = 0; b < count; b++) any_skip = any_skip || L.skip_on[b];
      if (any_skip) r = hip_check(hipMemcpyAsync(L.d_skip, L.h_skip, (size_t)count * K, hipMemcpyHostToDevice, L.stream), "skip upload");
    }
    if (r == DVM_OK) {
      auto dev = [](void* h) { void* d = nullptr; hipHostGetDevicePointer(&d, h, 0); return d; };
      launch_frame_build(L.stream, static_cast<const dvm_keypoint_pod*>(dev(L.kps)), (int64_t)K, static_cast<const uint8_t*>(dev(L.desc)), (int64_t)K * 32, 0,
                         static_cast<const int32_t*>(dev(L.n)), L.grid->view, 0, count);
      launch_match_window_ranked_batch(L.stream, L.grid->view, 0, count, L.d_skip, static_cast<const int32_t*>(dev(L.skip_on)), (int)K,
                                       static_cast<const uint8_t*>(dev(L.qdesc)), static_cast<const float*>(dev(L.qx)), static_cast<const float*>(dev(L.qy)),
                                       static_cast<const float*>(dev(L.qr)), static_cast<const int32_t*>(dev(L.qmin)), static_cast<const int32_t*>(dev(L.qmax)),
                                       static_cast<const int32_t*>(dev(L.nq)), (int)Q, static_cast<uint32_t*>(dev(L.ranked)));
      r = hip_check(hipGetLastError(), "frame_build + match launch");
      if (r == DVM_OK) r = hip_check(hipStreamSynchronize(L.stream), "search");
    }
    pool->gc.publish(li, r, r == DVM_OK ? std::string() : std::string(last_error_cstr()));
  }
  std::string err;
  int count = 0;
  rc = pool->gc.result(li, &err, &count);
  if (rc == DVM_OK) {
    if (nq) std::memcpy(ranked, L.ranked + s * Q * 4, (size_t)nq * 16);
    if (batch_size) *batch_size = count;
  } else {
    set_error("dvm_match_pool: " + err);
  }
  pool->gc.finish(li);
  return rc;
}
