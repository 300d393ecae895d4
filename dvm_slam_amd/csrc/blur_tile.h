// dvm_slam_amd/csrc/blur_tile.h -- one 64 x 64 tile of GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) in OpenCV's 8-bit fixed-point form
// (reference ORBextractor.cc:1105 -> cv::GaussianBlur), as a device function: k_blur7 (orb_kernels.hip) runs it for a batch, and the
// one-frame path runs it as extra workgroups of the octree launch (octree_kernel.hip: k_octree_blur), where it fills the chip the
// octree's eight latency-bound workgroups leave idle instead of being a launch of its own on the chain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orb_device.h"

namespace dvm {

typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
// v_dot2_u32_u16: a.lo * b.lo + a.hi * b.hi + c
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b), c, false);
}

constexpr int kRawPitch = 72;   // bytes: 64 + 6 halo, rounded to dwords (tile rows are dword aligned: x0 % 64 == 0)
constexpr int kHtPitch = (kBlurTH / 2 + 4) | 1;   // dwords per COLUMN of the transposed h-pass buffer (row pairs), odd -> conflict-free
// (64 x 32 tiles, tried for a smaller LDS footprint next to the concurrently running k_octree: blur 0.31 -> 0.36 ms)
// GaussianBlur 7x7 sigma 2, OpenCV's 8-bit fixed-point path: h-pass 8.8 (u16), v-pass 16.16 accumulate, +0.5, >> 16.
// The kernel is VALU-bound, so both passes run on the dot-product units:
//   h-pass: out(x) = v_dot4_u32_u8(bytes x-3..x, (g0,g1,g2,g3)) + v_dot4_u32_u8(bytes x+1..x+4, (g2,g1,g0,0));
//           one item = 4 columns x 2 rows, written as (row, row+1) u16 pairs into a column-major LDS buffer
//   v-pass: out(y) = sum of four v_dot2_u32_u16 over vertical pairs; one item = 4 columns x 4 rows, odd rows use
//           pairs re-aligned with v_alignbyte; rounding constant rides in the accumulator operand
// raw: (kBlurTH + 6) * kRawPitch bytes, hpt: kBlurTW * kHtPitch dwords of LDS (16-byte aligned); g0..g3: the 8.8 kernel's first four taps
__device__ __forceinline__ void blur_tile(const uint8_t* __restrict__ pyr, int pyr_frame_bytes, uint8_t* __restrict__ blur, int blur_frame_bytes,
                                          const TileDesc t, const PipelineDesc& PD, int f, uint32_t g0, uint32_t g1, uint32_t g2, uint32_t g3,
                                          uint8_t* raw, uint32_t* hpt) {
  const int tid = threadIdx.x;
  const LevelDesc& L = PD.lv[t.level];
  const int th = min(kBlurTH, L.h - t.y0);
  // raw tile: rows y0-3 .. y0+th+2, bordered columns (16 + x0) .. +71 as 18 aligned dwords per row
  const int colb = kEdge - 3 + t.x0;                       // multiple of 4
  const int ndw = min(kRawPitch / 4, (L.stride - colb) >> 2);  // stay inside the bordered row
  const uint32_t* g32 = reinterpret_cast<const uint32_t*>(pyr + (int64_t)f * pyr_frame_bytes + L.pyr_off +
                                                          (int64_t)(kEdge + t.y0 - 3) * L.stride + colb);
  uint32_t* r32 = reinterpret_cast<uint32_t*>(raw);
  {
    // (kBlurTH + 6) * 18 = 1260 dwords = 5 per thread: all loads go out before the first LDS store (one load + s_waitcnt
    // vmcnt(0) + store per loop iteration was five global round trips in a row)
    constexpr int kIt = ((kBlurTH + 6) * 18 + 255) / 256;
    uint32_t v[kIt];
    const int total = (th + 6) * 18;
#pragma unroll
    for (int k = 0; k < kIt; k++) {
      const int i = min(tid + 256 * k, total - 1);
      const int y = i / 18, x = i - 18 * y;
      v[k] = (x < ndw) ? g32[(int64_t)y * (L.stride >> 2) + x] : 0u;
    }
#pragma unroll
    for (int k = 0; k < kIt; k++)
      if (tid + 256 * k < total) r32[tid + 256 * k] = v[k];
  }
  __syncthreads();
  const uint32_t GA = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24), GB = g2 | (g1 << 8) | (g0 << 16);
  // h-pass
  const int npair = (th + 7) >> 1;
  for (int i = tid; i < npair * 16; i += 256) {
    const int yp = i >> 4, g = i & 15;
    const uint32_t* ra = r32 + (2 * yp) * 18 + g;
    const uint32_t a0 = ra[0], a1 = ra[1], a2 = ra[2], b0 = ra[18], b1 = ra[19], b2 = ra[20];
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t wa0 = k ? __builtin_amdgcn_alignbyte(a1, a0, k) : a0, wa1 = k ? __builtin_amdgcn_alignbyte(a2, a1, k) : a1;
      const uint32_t wb0 = k ? __builtin_amdgcn_alignbyte(b1, b0, k) : b0, wb1 = k ? __builtin_amdgcn_alignbyte(b2, b1, k) : b1;
      const uint32_t ha = __builtin_amdgcn_udot4(wa1, GB, __builtin_amdgcn_udot4(wa0, GA, 0u, false), false);
      const uint32_t hb = __builtin_amdgcn_udot4(wb1, GB, __builtin_amdgcn_udot4(wb0, GA, 0u, false), false);
      o[k] = ha | (hb << 16);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) hpt[(4 * g + k) * kHtPitch + yp] = o[k];
  }
  __syncthreads();
  // v-pass
  const uint32_t W01 = g0 | (g1 << 16), W23 = g2 | (g3 << 16), W21 = g2 | (g1 << 16), W0 = g0;
  uint8_t* dst = blur + (int64_t)f * blur_frame_bytes + L.blur_off + (int64_t)t.y0 * L.blur_stride + t.x0;
  for (int i = tid; i < ((th + 3) >> 2) * 16; i += 256) {
    const int gy = i >> 4, gx = i & 15;
    uint32_t acc[4][4];   // [row][column]
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint32_t* col = hpt + (4 * gx + c) * kHtPitch + 2 * gy;
      uint32_t P[6], Q[5];
#pragma unroll
      for (int j = 0; j < 6; j++) P[j] = col[j];
#pragma unroll
      for (int j = 0; j < 5; j++) Q[j] = __builtin_amdgcn_alignbyte(P[j + 1], P[j], 2);
      acc[0][c] = udot2(P[3], W0, udot2(P[2], W21, udot2(P[1], W23, udot2(P[0], W01, 32768u))));
      acc[1][c] = udot2(Q[3], W0, udot2(Q[2], W21, udot2(Q[1], W23, udot2(Q[0], W01, 32768u))));
      acc[2][c] = udot2(P[4], W0, udot2(P[3], W21, udot2(P[2], W23, udot2(P[1], W01, 32768u))));
      acc[3][c] = udot2(Q[4], W0, udot2(Q[3], W21, udot2(Q[2], W23, udot2(Q[1], W01, 32768u))));
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int y = 4 * gy + r;
      if (y < th) {
        // byte 2 of each accumulator = (acc >> 16) & 255 (the sum never reaches 256 << 16)
        const uint32_t lo = __builtin_amdgcn_perm(acc[r][1], acc[r][0], 0x0c0c0602u);
        const uint32_t hi = __builtin_amdgcn_perm(acc[r][3], acc[r][2], 0x06020c0cu);
        // the blurred image's pitch is a multiple of 64, so a full dword store never leaves the row
        *reinterpret_cast<uint32_t*>(dst + (int64_t)y * L.blur_stride + 4 * gx) = lo | hi;
      }
    }
  }
}

}  // namespace dvm
