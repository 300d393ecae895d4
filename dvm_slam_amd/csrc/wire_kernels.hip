// dvm_slam_amd/csrc/wire_kernels.hip -- device side of the DVMW sender (include/dvmslam_wire.h): the keypoints and
// descriptors of a batch of keyframes go from the extractor's result arrays straight into the pooled sections of the
// block in HBM, where RCCL picks the block up.  Pure byte traffic: 60 B per keypoint read and written once, dword /
// dwordx4 coalesced; grid = (chunks, keyframes).
#include <hip/hip_runtime.h>

#include "wire_kernels.h"

namespace dvm {

__global__ void __launch_bounds__(256) k_wire_gather(uint8_t* __restrict__ blk, int first_kf, const uint32_t* __restrict__ kps, int64_t kps_stride,
                                                     const uint8_t* __restrict__ desc, int64_t desc_stride) {
  const WireHeader h = *reinterpret_cast<const WireHeader*>(blk);
  const WireLayout L = wire_layout(wire_counts(h));
  const int j = blockIdx.y, kf = first_kf + j;
  if ((uint32_t)kf >= h.n_keyframes) return;
  const uint32_t* rec = reinterpret_cast<const uint32_t*>(blk + L.offset[1] + 192ull * kf);
  const uint32_t n_kp = rec[35], kp_off = rec[36];   // dvm_wire_keyframe::n_kp / kp_off at byte 140 / 144
  if ((uint64_t)kp_off + n_kp > h.n_keypoints) return;
  // keypoints: 7 dwords each
  const uint32_t* ks = kps + (int64_t)j * kps_stride * 7;
  uint32_t* kd = reinterpret_cast<uint32_t*>(blk + L.offset[3]) + (uint64_t)kp_off * 7;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_kp * 7; i += gridDim.x * 256) kd[i] = ks[i];
  // descriptors: 2 x 16 B each (source and destination are 32-byte aligned)
  const uint4* ds = reinterpret_cast<const uint4*>(desc + (int64_t)j * desc_stride);
  uint4* dd = reinterpret_cast<uint4*>(blk + L.offset[4]) + (uint64_t)kp_off * 2;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_kp * 2; i += gridDim.x * 256) dd[i] = ds[i];
}

void launch_wire_gather(hipStream_t s, uint8_t* d_block, int first_kf, int count, const uint32_t* d_kps, int64_t kps_stride,
                        const uint8_t* d_desc, int64_t desc_stride) {
  hipLaunchKernelGGL(k_wire_gather, dim3(8, count), dim3(256), 0, s, d_block, first_kf, d_kps, kps_stride, d_desc, desc_stride);
}

}  // namespace dvm
