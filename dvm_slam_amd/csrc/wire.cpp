// dvm_slam_amd/csrc/wire.cpp -- DVMW map wire format: layout, host-side assembly and validation (include/dvmslam_wire.h).
#include <cstring>
#include <map>
#include <string>

#include "../../include/dvmslam_wire.h"
#include "wire_kernels.h"

namespace dvm { void set_error(const std::string& s); }

static_assert(sizeof(dvm_wire_header) == 64 && sizeof(dvm_wire_keyframe) == 192 && sizeof(dvm_wire_mappoint) == 160 &&
                  sizeof(dvm_wire_link) == 24 && sizeof(dvm_wire_obs) == 24 && sizeof(dvm_uuid) == 16 && sizeof(dvm_keypoint) == 28,
              "wire record sizes");

extern "C" {

int dvm_wire_layout(const dvm_wire_header* h, dvm_wire_layout_t* out) {
  if (!h || !out) return DVM_ERR_INVALID;
  dvm::WireLayout L = dvm::wire_layout(dvm::wire_counts(*reinterpret_cast<const dvm::WireHeader*>(h)));
  for (int i = 0; i < DVM_WIRE_SECTIONS; i++) { out->offset[i] = L.offset[i]; out->bytes[i] = L.bytes[i]; }
  out->total_bytes = L.total;
  return DVM_OK;
}

static int check_ranges(const dvm_wire_header& h, const dvm_wire_keyframe* kfs, const dvm_wire_mappoint* mps, const int32_t* bow_ids) {
  auto inside = [](uint64_t off, uint64_t n, uint64_t total) { return off <= total && n <= total - off; };
  for (uint32_t i = 0; i < h.n_keyframes; i++) {
    const dvm_wire_keyframe& k = kfs[i];
    if (!inside(k.kp_off, k.n_kp, h.n_keypoints)) { dvm::set_error("wire: keyframe " + std::to_string(i) + " keypoint range outside its section"); return DVM_ERR_INVALID; }
    if (!inside(k.bow_off, k.n_bow, h.n_bow)) { dvm::set_error("wire: keyframe " + std::to_string(i) + " BoW range outside its section"); return DVM_ERR_INVALID; }
    if (!inside(k.fv_node_off, k.n_fv_nodes, h.n_fv_nodes) || k.fv_feat_off > h.n_fv_feats) { dvm::set_error("wire: keyframe " + std::to_string(i) + " feature-vector range outside its section"); return DVM_ERR_INVALID; }
    if (!inside(k.link_off, k.n_links, h.n_links)) { dvm::set_error("wire: keyframe " + std::to_string(i) + " link range outside its section"); return DVM_ERR_INVALID; }
    if (k.n_levels < 0 || k.n_levels > 64) { dvm::set_error("wire: keyframe " + std::to_string(i) + " n_levels"); return DVM_ERR_INVALID; }
    if (bow_ids)
      for (uint32_t w = 1; w < k.n_bow; w++)
        if (bow_ids[k.bow_off + w] <= bow_ids[k.bow_off + w - 1]) { dvm::set_error("wire: keyframe " + std::to_string(i) + " BoW ids not ascending"); return DVM_ERR_INVALID; }
  }
  for (uint32_t i = 0; i < h.n_mappoints; i++)
    if (!inside(mps[i].obs_off, mps[i].n_obs, h.n_obs)) { dvm::set_error("wire: map point " + std::to_string(i) + " observation range outside its section"); return DVM_ERR_INVALID; }
  return DVM_OK;
}

int dvm_wire_build(const dvm_wire_header* counts, const dvm_wire_keyframe* kfs, const dvm_wire_mappoint* mps,
                   const dvm_keypoint* kps, const uint8_t* desc, const dvm_uuid* kp_mappoint, const int32_t* bow_ids,
                   const double* bow_vals, const int32_t* fv_nodes, const int32_t* fv_feats, const dvm_wire_link* links,
                   const dvm_wire_obs* obs, int head_only, void* out, uint64_t out_bytes) {
  if (!counts || !out || (counts->n_keyframes && !kfs) || (counts->n_mappoints && !mps)) return DVM_ERR_INVALID;
  dvm_wire_layout_t L;
  dvm_wire_layout(counts, &L);
  const uint64_t need = head_only ? L.offset[3] : L.total_bytes;
  if (out_bytes < need) { dvm::set_error("wire: output buffer too small"); return DVM_ERR_CAPACITY; }
  const int rc = check_ranges(*counts, kfs, mps, bow_ids);
  if (rc != DVM_OK) return rc;
  // the feature lists of a keyframe must fit the feature section
  if (fv_nodes)
    for (uint32_t i = 0; i < counts->n_keyframes; i++) {
      uint64_t nf = 0;
      for (uint32_t k = 0; k < kfs[i].n_fv_nodes; k++) {
        const int32_t c = fv_nodes[2 * (kfs[i].fv_node_off + k) + 1];
        if (c < 0) { dvm::set_error("wire: negative feature count"); return DVM_ERR_INVALID; }
        nf += (uint64_t)c;
      }
      if (kfs[i].fv_feat_off + nf > counts->n_fv_feats) { dvm::set_error("wire: keyframe " + std::to_string(i) + " feature list outside its section"); return DVM_ERR_INVALID; }
    }
  uint8_t* o = static_cast<uint8_t*>(out);
  std::memset(o, 0, (size_t)need);
  dvm_wire_header h = *counts;
  h.magic = DVM_WIRE_MAGIC; h.version = DVM_WIRE_VERSION; h.total_bytes = L.total_bytes;
  h.reserved[0] = h.reserved[1] = 0;
  std::memcpy(o, &h, sizeof(h));
  if (counts->n_keyframes) std::memcpy(o + L.offset[1], kfs, (size_t)L.bytes[1]);
  if (counts->n_mappoints) std::memcpy(o + L.offset[2], mps, (size_t)L.bytes[2]);
  if (head_only) return DVM_OK;
  const void* src[DVM_WIRE_SECTIONS] = {nullptr, nullptr, nullptr, kps, desc, kp_mappoint, bow_ids, bow_vals, fv_nodes, fv_feats, links, obs};
  for (int s = 3; s < DVM_WIRE_SECTIONS; s++)
    if (src[s] && L.bytes[s]) std::memcpy(o + L.offset[s], src[s], (size_t)L.bytes[s]);
  return DVM_OK;
}

int dvm_wire_validate(const void* block, uint64_t bytes) {
  if (!block || bytes < sizeof(dvm_wire_header)) { dvm::set_error("wire: block shorter than its header"); return DVM_ERR_INVALID; }
  dvm_wire_header h;
  std::memcpy(&h, block, sizeof(h));
  if (h.magic != DVM_WIRE_MAGIC) { dvm::set_error("wire: bad magic"); return DVM_ERR_INVALID; }
  if (h.version != DVM_WIRE_VERSION) { dvm::set_error("wire: unsupported version " + std::to_string(h.version)); return DVM_ERR_INVALID; }
  dvm_wire_layout_t L;
  dvm_wire_layout(&h, &L);
  if (h.total_bytes != L.total_bytes || bytes < L.total_bytes) { dvm::set_error("wire: size does not match the counts in the header"); return DVM_ERR_INVALID; }
  const uint8_t* b = static_cast<const uint8_t*>(block);
  const dvm_wire_keyframe* kfs = reinterpret_cast<const dvm_wire_keyframe*>(b + L.offset[1]);
  const dvm_wire_mappoint* mps = reinterpret_cast<const dvm_wire_mappoint*>(b + L.offset[2]);
  int rc = check_ranges(h, kfs, mps, reinterpret_cast<const int32_t*>(b + L.offset[6]));
  if (rc != DVM_OK) return rc;
  const int32_t* fvn = reinterpret_cast<const int32_t*>(b + L.offset[8]);
  for (uint32_t i = 0; i < h.n_keyframes; i++) {
    uint64_t nf = 0;
    for (uint32_t k = 0; k < kfs[i].n_fv_nodes; k++) {
      const int32_t c = fvn[2 * (kfs[i].fv_node_off + k) + 1];
      if (c < 0) { dvm::set_error("wire: negative feature count"); return DVM_ERR_INVALID; }
      nf += (uint64_t)c;
    }
    if (kfs[i].fv_feat_off + nf > h.n_fv_feats) { dvm::set_error("wire: keyframe " + std::to_string(i) + " feature list outside its section"); return DVM_ERR_INVALID; }
    const int32_t* ff = reinterpret_cast<const int32_t*>(b + L.offset[9]) + kfs[i].fv_feat_off;
    for (uint64_t k = 0; k < nf; k++)
      if (ff[k] < 0 || (uint32_t)ff[k] >= kfs[i].n_kp) { dvm::set_error("wire: feature index beyond the keyframe's keypoints"); return DVM_ERR_INVALID; }
  }
  // observations: a keypoint index is never negative (index_right may be -1: monocular), and where the observing keyframe
  // travels in the same block the index must address one of ITS keypoints -- consumers index with it (agents.unpack_candidate)
  if (h.n_obs) {
    std::map<std::pair<uint64_t, uint64_t>, uint32_t> n_kp_of;
    auto key = [](const dvm_uuid& u) { uint64_t a, c; std::memcpy(&a, u.b, 8); std::memcpy(&c, u.b + 8, 8); return std::make_pair(a, c); };
    for (uint32_t i = 0; i < h.n_keyframes; i++) n_kp_of[key(kfs[i].uuid)] = kfs[i].n_kp;
    const dvm_wire_obs* obs = reinterpret_cast<const dvm_wire_obs*>(b + L.offset[11]);
    for (uint32_t k = 0; k < h.n_obs; k++) {
      if (obs[k].index < 0 || obs[k].index_right < -1) { dvm::set_error("wire: observation " + std::to_string(k) + " has a negative keypoint index"); return DVM_ERR_INVALID; }
      const auto it = n_kp_of.find(key(obs[k].kf_uuid));
      if (it != n_kp_of.end() && (uint32_t)obs[k].index >= it->second) { dvm::set_error("wire: observation " + std::to_string(k) + " indexes past its keyframe's keypoints"); return DVM_ERR_INVALID; }
    }
  }
  return DVM_OK;
}

int dvm_wire_gather_keypoints(void* d_block, int first_kf, int count, const dvm_keypoint* d_kps, int64_t kps_stride,
                              const uint8_t* d_desc, int64_t desc_stride, void* stream) {
  if (!d_block || first_kf < 0 || count < 0 || (count > 0 && (!d_kps || !d_desc))) return DVM_ERR_INVALID;
  if (count == 0) return DVM_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { dvm::set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  dvm::launch_wire_gather((hipStream_t)stream, static_cast<uint8_t*>(d_block), first_kf, count, reinterpret_cast<const uint32_t*>(d_kps), kps_stride,
                          d_desc, desc_stride);
  if (hipGetLastError() != hipSuccess) { dvm::set_error("wire gather launch failed"); return DVM_ERR_HIP; }
  return DVM_OK;
}

}  // extern "C"
