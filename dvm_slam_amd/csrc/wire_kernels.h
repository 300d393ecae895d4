// dvm_slam_amd/csrc/wire_kernels.h -- DVMW layout arithmetic shared by host and device + the gather launcher.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dvm {

constexpr int kWireSections = 12;
struct WireHeader {   // == dvm_wire_header
  uint32_t magic, version;
  uint64_t total_bytes;
  uint32_t n_keyframes, n_mappoints, n_keypoints, n_bow, n_fv_nodes, n_fv_feats, n_links, n_obs;
  uint32_t sender_agent, flags, reserved[2];
};
struct WireCounts { uint32_t n_keyframes, n_mappoints, n_keypoints, n_bow, n_fv_nodes, n_fv_feats, n_links, n_obs; };
struct WireLayout { uint64_t offset[kWireSections], bytes[kWireSections], total; };

__host__ __device__ inline WireCounts wire_counts(const WireHeader& h) {
  return {h.n_keyframes, h.n_mappoints, h.n_keypoints, h.n_bow, h.n_fv_nodes, h.n_fv_feats, h.n_links, h.n_obs};
}
__host__ __device__ inline WireLayout wire_layout(const WireCounts& c) {
  WireLayout L;
  const uint64_t b[kWireSections] = {64, 192ull * c.n_keyframes, 160ull * c.n_mappoints, 28ull * c.n_keypoints, 32ull * c.n_keypoints,
                                     16ull * c.n_keypoints, 4ull * c.n_bow, 8ull * c.n_bow, 8ull * c.n_fv_nodes, 4ull * c.n_fv_feats,
                                     24ull * c.n_links, 24ull * c.n_obs};
  uint64_t off = 0;
  for (int i = 0; i < kWireSections; i++) {
    L.offset[i] = off; L.bytes[i] = b[i];
    off = (off + b[i] + 63) & ~63ull;
  }
  L.total = off;
  return L;
}

void launch_wire_gather(hipStream_t s, uint8_t* d_block, int first_kf, int count, const uint32_t* d_kps, int64_t kps_stride,
                        const uint8_t* d_desc, int64_t desc_stride);
}  // namespace dvm
