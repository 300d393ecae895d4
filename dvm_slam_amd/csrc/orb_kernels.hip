// dvm_slam_amd/csrc/orb_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ORB front end.
//
// Stage map (reference file:line -> kernel), see DESIGN.md for layouts and rooflines:
//   ORBextractor.cc:957-976  ComputePyramid              -> k_pyr_level0, k_pyr_resize
//   ORBextractor.cc:634-692  per-cell cv::FAST x2 + NMS  -> k_fast_cells (lists concatenated per level by k_octree)
//   ORBextractor.cc:876-955  operator() output placement -> k_assemble
//   ORBextractor.cc:919-920  GaussianBlur 7x7 sigma 2    -> k_blur7
//   ORBextractor.cc:75-99    IC_Angle                    -> k_orient_desc (phase 1)
//   ORBextractor.cc:102-143  computeOrbDescriptor        -> k_orient_desc (phase 2)
// All integer results are bit-exact by construction; the few float ops (fastAtan2, rotation of the
// BRIEF pattern, keypoint scaling) are compiled with FP contraction OFF so every operation is one
// IEEE rounding, the same sequence the CPU oracle performs.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>

#include <algorithm>
#include <cstdlib>

#include "orb_device.h"
#include "orb_kernels.h"
#include "blur_tile.h"

#pragma clang fp contract(off)

namespace dvm {

__constant__ int c_pattern[1024] = {
#include "orb_pattern_31.inc"
};
// FAST-16 Bresenham circle, OpenCV order (dx,dy)
__constant__ int8_t c_circle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                       {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
// orientation disc: for pixel p of the 749-px disc, (u,v) offsets; filled by the host at init
// orientation disc as dot4 weights: item (row v+15, dword c) of the 31 x 32-byte patch rows -> {WU, W1}: byte k of
// WU = u+16 (u = 4c+k-15) inside the disc else 0, byte k of W1 = 1 inside else 0; filled by the host at init
__device__ uint2 c_disc_w[256];
__constant__ int c_gauss7[7];

// XCD-aware (block, frame) mapping for per-frame kernels launched as a 1-D grid of
// blocks_per_frame * 8 * ceil(batch/8) workgroups.  The dispatcher places workgroup b on XCD b % 8
// (observed, used for speed only): interleaving 8 frames keeps ALL workgroups of one frame on one XCD,
// so the frame's pyramid / blurred levels (~2 MB) are served by that XCD's 4 MB L2 instead of being
// fetched into all eight.  Returns false for the padding frames of a batch that is not a multiple of 8.
__device__ __forceinline__ bool xcd_frame_map(int blocks_per_frame, int batch, int& blk, int& f) {
  const int b = blockIdx.x;
  const int i = b >> 3;
  blk = i % blocks_per_frame;
  f = (i / blocks_per_frame) * 8 + (b & 7);
  return f < batch;
}
static inline int xcd_grid(int blocks_per_frame, int batch) { return blocks_per_frame * 8 * ((batch + 7) / 8); }

__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * (len - 1) - p;
  return p;
}

// ------------------------------------------------------------------------------------------ K1
// ComputePyramid (reference ORBextractor.cc:957-976) in three kinds of launches:
//   k_pyr_level0   level 0 = the input image (interior only)
//   k_pyr_resize   level l = cv::resize(level l-1, INTER_LINEAR), 8-bit fixed point, interior only;
//                  a workgroup stages the source rectangle of its 256x16 destination tile in LDS with
//                  aligned dword loads, then every destination pixel is 4 LDS byte reads
//   k_pyr_borders  copyMakeBorder(REFLECT_101, 19 px) of ALL levels in one launch (the border of a level
//                  depends only on that level's interior; no later stage of the mono path reads it except
//                  the blur, which needs 3 px)
// Destination dwords are aligned in BORDERED coordinates (the image starts at byte 19 of a row), so a
// dword at a row end may mix interior and border bytes: those are stored byte-wise.
constexpr int kRzTW = 256, kRzTH = 64;   // destination tile (bordered columns x interior rows); 16 rows for tiny batches (latency path)

__device__ __forceinline__ void store_px4(uint8_t* D, int X4, int w, uint32_t v) {
  const int d0 = X4 - kEdge;             // interior x of byte 0
  if (d0 >= 0 && d0 + 3 < w) {
    *reinterpret_cast<uint32_t*>(D + X4) = v;
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (d0 + k >= 0 && d0 + k < w) D[X4 + k] = (uint8_t)(v >> (8 * k));
  }
}

// 16 bytes of source row S starting at column x0 (any alignment, may start before / end after the row): 5 aligned
// dwords clamped to the row + v_alignbyte.  Bytes outside [0, cols) are unspecified.
__device__ __forceinline__ uint4 row_window16(const uint8_t* S, int cols, int x0) {
  const uintptr_t r0 = reinterpret_cast<uintptr_t>(S);
  const uintptr_t lo = r0 & ~(uintptr_t)3, hi = (r0 + cols - 1) & ~(uintptr_t)3;   // first / last dword of the row
  const intptr_t a = (intptr_t)r0 + x0;
  const uintptr_t q = (uintptr_t)(a & ~(intptr_t)3);
  const uint32_t sh = (uint32_t)(a & 3);
  uint32_t d[5];
#pragma unroll
  for (int j = 0; j < 5; j++) {
    intptr_t p = (intptr_t)q + 4 * j;
    p = p < (intptr_t)lo ? (intptr_t)lo : (p > (intptr_t)hi ? (intptr_t)hi : p);
    d[j] = *reinterpret_cast<const uint32_t*>(p);
  }
  return make_uint4(__builtin_amdgcn_alignbyte(d[1], d[0], sh), __builtin_amdgcn_alignbyte(d[2], d[1], sh),
                    __builtin_amdgcn_alignbyte(d[3], d[2], sh), __builtin_amdgcn_alignbyte(d[4], d[3], sh));
}
__device__ __forceinline__ uint32_t byte_mask_lt(int n) {   // bytes [0, n) set, n clamped to 0..4
  return n <= 0 ? 0u : (n >= 4 ? 0xFFFFFFFFu : ((1u << (8 * n)) - 1u));
}
// bordered rows that mirror interior row y (REFLECT_101, 19 px): at most one above and one below when h >= 20
__device__ __forceinline__ void mirror_rows(int y, int h, int& top, int& bot) {
  top = (y >= 1 && y <= kEdge) ? kEdge - y : -1;
  bot = (y >= h - 1 - kEdge && y <= h - 2) ? 2 * h + kEdge - 2 - y : -1;
}

// level 0 WITH its REFLECT_101 frame: thread = 16 destination bytes [X, X+16) (X % 16 == 0, bordered columns
// 0 .. w+37) of FOUR interior rows (y, y+4, y+8, y+12; all loads are issued before the first store).
// Interior bytes come from the source run at x = X-19 (row_window16); a group that overlaps the left / right frame
// also builds the byte-reversed run of the pixels it mirrors and merges the two per byte, so every cache line of
// the bordered row is written once, in full.  Rows 1..19 and h-20..h-2 are stored a second time into the top /
// bottom strip row that mirrors them.
__device__ __forceinline__ void pyr_level0_tile(const uint8_t* __restrict__ S, int rows, int cols, int sstride,
                                                uint8_t* __restrict__ D, const LevelDesc& L, int bx, int by) {
  const int X = (bx * 64 + (threadIdx.x & 63)) * 16;
  const int yb = by * 16 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row addresses / mirror rows in SGPRs
  if (X >= cols + 2 * kEdge) return;
  const bool left = X < kEdge, right = X + 16 > cols + kEdge;
  // mirrored run: left frame  x' = 19 - X - k  -> reversed window starting at 4 - X
  //               right frame x' = 2w + 17 - X - k -> reversed window starting at 2w + 2 - X
  const int xm = left ? 4 - X : 2 * cols + 2 - X;
  const int kb = left ? kEdge - X : cols + kEdge - X;   // left: bytes k < kb mirrored; right: bytes k >= kb mirrored
  uint4 v[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int y = min(yb + 4 * i, rows - 1);
    const uint8_t* Sr = S + (int64_t)y * sstride;
    v[i] = row_window16(Sr, cols, X - kEdge);
    if (left || right) {
      const uint4 m = row_window16(Sr, cols, xm);
      const uint32_t r0 = __builtin_amdgcn_perm(0u, m.w, 0x00010203u), r1 = __builtin_amdgcn_perm(0u, m.z, 0x00010203u);
      const uint32_t r2 = __builtin_amdgcn_perm(0u, m.y, 0x00010203u), r3 = __builtin_amdgcn_perm(0u, m.x, 0x00010203u);
      // mask = bytes that take the mirrored value
      uint32_t k0 = byte_mask_lt(kb), k1 = byte_mask_lt(kb - 4), k2 = byte_mask_lt(kb - 8), k3 = byte_mask_lt(kb - 12);
      if (!left) { k0 = ~k0; k1 = ~k1; k2 = ~k2; k3 = ~k3; }
      v[i].x = (v[i].x & ~k0) | (r0 & k0);
      v[i].y = (v[i].y & ~k1) | (r1 & k1);
      v[i].z = (v[i].z & ~k2) | (r2 & k2);
      v[i].w = (v[i].w & ~k3) | (r3 & k3);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int y = yb + 4 * i;
    if (y < rows) {
      *reinterpret_cast<uint4*>(D + (int64_t)(kEdge + y) * L.stride + X) = v[i];
      int top, bot;
      mirror_rows(y, rows, top, bot);
      if (top >= 0) *reinterpret_cast<uint4*>(D + (int64_t)top * L.stride + X) = v[i];
      if (bot >= 0) *reinterpret_cast<uint4*>(D + (int64_t)bot * L.stride + X) = v[i];
    }
  }
}

__global__ void __launch_bounds__(256) k_pyr_level0(const uint8_t* __restrict__ src, int rows, int cols, int sstride,
                                                    int64_t frame_stride, uint8_t* __restrict__ pyr,
                                                    int pyr_frame_bytes, LevelDesc L) {
  const int f = blockIdx.z;
  pyr_level0_tile(src + (int64_t)f * frame_stride, rows, cols, sstride, pyr + (int64_t)f * pyr_frame_bytes + L.pyr_off, L, blockIdx.x, blockIdx.y);
}

// level l = cv::resize(level l-1, INTER_LINEAR) WITH its REFLECT_101 frame.  A workgroup owns a 256 x 16 tile of
// BORDERED columns x interior rows; a frame column computes the pixel it mirrors (its source lies in the same LDS
// rectangle, at most 19 destination pixels further in), rows 1..19 / h-20..h-2 are stored twice (strip rows).
// TH = rows per tile: 64 for throughput (the column set-up and the tile load are amortised over 16 rows per thread), 16 when a
// handful of frames is all there is (one frame through the drop-in boundary: 21 workgroups of 16 serial rows each made every level
// a 13 us launch; four times as many workgroups of 4 rows are back in ~6 us).  Same arithmetic, same bytes.
extern __shared__ __attribute__((aligned(16))) uint8_t rz_smem[];
template <int TH>
__device__ __forceinline__ void pyr_resize_tile(uint8_t* __restrict__ pyr, int pyr_frame_bytes, const LevelDesc& P, const LevelDesc& L,
                                                const int32_t* __restrict__ tabs, int lds_pitch, int f, int bx, int by) {
  const int tid = threadIdx.x, tx = tid & 63;
  const int ty = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the row tables / weights / LDS row bases below go to SGPRs
  const int w = L.w, bw = L.w + 2 * kEdge;
  const int X0 = bx * kRzTW;               // first bordered column of the tile (dword aligned)
  const int y0 = by * TH;               // first interior row
  if (X0 >= bw) return;
  // interior columns whose sources the tile needs: its own, plus the ones its frame columns mirror
  const int lo = X0 - kEdge, hi = min(X0 + kRzTW - 1, bw - 1) - kEdge;
  int dxa = max(lo, 0), dxb = min(hi, w - 1);
  if (lo < 0) dxb = max(dxb, min(-lo, w - 1));
  if (hi > w - 1) dxa = min(dxa, max(2 * (w - 1) - hi, 0));
  const int32_t* xofs = tabs + L.tab_off;
  const int32_t* xal = xofs + L.w;   // (a0 | a1 << 16)
  const int32_t* yofs = xal + L.w;
  const int32_t* ybe = yofs + L.h;   // (b0 | b1 << 16)
  const int yb = min(y0 + TH - 1, L.h - 1);
  const int sya = min(max(yofs[y0], 0), P.h - 1), syb = min(max(yofs[yb] + 1, 0), P.h - 1);
  const int sxa = xofs[dxa], sxb = xofs[dxb] + 1;  // sx+1 may be the first border column of level l-1 (weight 0)
  const int ga = (kEdge + sxa) & ~3;               // bordered source column of LDS column 0
  const int ndw = ((kEdge + sxb - ga) >> 2) + 1;
  const uint8_t* Sg = pyr + (int64_t)f * pyr_frame_bytes + P.pyr_off + (int64_t)(kEdge + sya) * P.stride + ga;
  // this thread's column tables and this wave's row tables go out BEFORE the tile is fetched: they depend on the tables only,
  // and behind the barrier they were one more memory round trip on a kernel that is nothing but a chain of them at small batch
  const int X4 = X0 + 4 * tx;
  int sx[4];
  uint32_t al[4];   // (a0 | a1 << 16): the two 11-bit horizontal weights, ready for v_dot2_u32_u16
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int x = X4 + k - kEdge;
    const int xm = x < 0 ? -x : (x >= w ? 2 * (w - 1) - x : x);   // REFLECT_101 (one bounce: w >= 20)
    const int dxc = min(max(min(max(xm, dxa), dxb), 0), w - 1);   // pad bytes beyond the frame: any in-tile column
    sx[k] = kEdge + xofs[dxc] - ga;
    al[k] = (uint32_t)xal[dxc];
  }
  // the row tables of this wave's TH / 4 rows, all fetched before the first row is computed (read inside the loop, every row
  // began with two scalar loads and an s_waitcnt lgkmcnt(0))
  int sy_[TH / 4];
  uint32_t bb_[TH / 4];
#pragma unroll
  for (int rr = 0; rr < TH / 4; rr++) {
    const int dyc = min(y0 + ty + 4 * rr, L.h - 1);
    sy_[rr] = yofs[dyc];
    bb_[rr] = (uint32_t)ybe[dyc];
  }
  uint32_t* lds32 = reinterpret_cast<uint32_t*>(rz_smem);
  const int nrow = syb - sya + 1;
  {
    // EIGHT loads in flight per thread: written as one load + one LDS store per loop iteration, the compiler put an
    // s_waitcnt vmcnt(0) between them and a workgroup paid ~25 global round trips in a row for its ~6 400 source dwords
    // (a 256 x 64 tile's launch was 30 us of waiting: the seven resize launches of a 256-frame batch 241 us for 102 us of issue)
    const int total = nrow * ndw, pitch4 = lds_pitch >> 2, sstride4 = P.stride >> 2;
    const float inv = 1.0f / (float)ndw;
    const uint32_t* S32 = reinterpret_cast<const uint32_t*>(Sg);   // (ga and the row pitch are multiples of 4)
    for (int base = tid; base < total; base += 256 * 8) {
      uint32_t v[8];
      int at[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int i = min(base + 256 * k, total - 1);
        const int r = (int)(((float)i + 0.5f) * inv), c = i - r * ndw;     // exact for these small ints
        v[k] = S32[r * sstride4 + c];
        at[k] = r * pitch4 + c;
      }
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (base + 256 * k < total) lds32[at[k]] = v[k];
    }
  }
  __syncthreads();
  if (X4 >= bw) return;
  uint8_t* Dl = pyr + (int64_t)f * pyr_frame_bytes + L.pyr_off;
#pragma unroll
  for (int rr = 0; rr < TH / 4; rr++) {
    const int dy = y0 + ty + 4 * rr;
    if (dy >= L.h) break;
    const int sy = sy_[rr];
    const int r0 = min(max(sy, 0), P.h - 1) - sya, r1 = min(max(sy + 1, 0), P.h - 1) - sya;
    const uint32_t bb = bb_[rr];
    const uint32_t b0 = bb & 0xFFFFu, b1 = bb >> 16;
    const uint8_t* S0 = rz_smem + r0 * lds_pitch;
    const uint8_t* S1 = rz_smem + r1 * lds_pitch;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // (unaligned LDS u16 reads are slow on gfx950 -- measured 2x on the whole pyramid -- so two byte reads per row)
      const uint32_t s00 = S0[sx[k]], s01 = S0[sx[k] + 1], s10 = S1[sx[k]], s11 = S1[sx[k] + 1];
      const uint32_t h0 = udot2(s00 | (s01 << 16), al[k], 0u);
      const uint32_t h1 = udot2(s10 | (s11 << 16), al[k], 0u);
      // ((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2; a convex combination, never above 255
      const uint32_t r = ((__umul24(b0, h0 >> 4) >> 16) + (__umul24(b1, h1 >> 4) >> 16) + 2u) >> 2;
      v |= r << (8 * k);
    }
    // row pitch is a multiple of 64 >= bw: a full dword store never leaves the row
    *reinterpret_cast<uint32_t*>(Dl + (int64_t)(kEdge + dy) * L.stride + X4) = v;
    int top, bot;
    mirror_rows(dy, L.h, top, bot);
    if (top >= 0) *reinterpret_cast<uint32_t*>(Dl + (int64_t)top * L.stride + X4) = v;
    if (bot >= 0) *reinterpret_cast<uint32_t*>(Dl + (int64_t)bot * L.stride + X4) = v;
  }
}

template <int TH>
__global__ void __launch_bounds__(256) k_pyr_resize(uint8_t* __restrict__ pyr, int pyr_frame_bytes, LevelDesc P,
                                                    LevelDesc L, const int32_t* __restrict__ tabs, int lds_pitch) {
  pyr_resize_tile<TH>(pyr, pyr_frame_bytes, P, L, tabs, lds_pitch, blockIdx.z, blockIdx.x, blockIdx.y);
}

// FALLBACK for tiny levels (w < 40 or h < 20, where the mirror bounces more than once): the level kernels above
// already write the frame for every normal size, and this kernel is then not launched.
// REFLECT_101 frame of every level in ONE launch.  Per level, "bordered row" R in [0, h+38):
//   interior rows (19 <= R < h+19): source row = the row itself, only the 19 + 19 side columns are written
//   strip rows (top / bottom 19):   source row = the mirrored interior row; the whole row is written
// Work items:  blockIdx.y <  strip_blocks : 4 strip rows per workgroup, thread = one FULLY INTERIOR 16-byte group
//                                           (straight dwordx4 copy of the mirrored row)
//              above                       : 128 bordered rows per workgroup, two threads per row:
//   left  thread: D[X] = S[38 - X], X = 0..18 -> byte-reversed S[16..39] (5 v_perm); strip rows also copy S[20..31]
//   right thread: D[X] = S[2w + 36 - X], X = w+19..w+37 -> per destination dword an aligned pair of source dwords,
//                 v_alignbyte to the 4-byte run, byte reversal; the first dword keeps its interior bytes; strip rows
//                 also copy the interior dwords after the last full 16-byte group
// Sources are always INTERIOR pixels of a level, so nothing here depends on another border byte.
__global__ void __launch_bounds__(256) k_pyr_borders(uint8_t* __restrict__ pyr, PipelineDesc PD, int strip_blocks) {
  const int f = blockIdx.z, tid = threadIdx.x;
  if ((int)blockIdx.y < strip_blocks) {
    const int sr = blockIdx.y * 4 + (tid >> 6);
    if (sr >= 2 * kEdge * PD.nlevels) return;
    const LevelDesc& L = PD.lv[sr / (2 * kEdge)];
    const int r = sr % (2 * kEdge);
    const int row = r < kEdge ? r : L.h + r;          // bordered row: 0..18 or h+19..h+37
    const int X = 32 + (blockIdx.x * 64 + (tid & 63)) * 16;
    if (X + 16 > L.w + kEdge) return;
    uint8_t* base = pyr + (int64_t)f * PD.pyr_frame_bytes + L.pyr_off;
    const uint8_t* S = base + (int64_t)(kEdge + reflect101(row - kEdge, L.h)) * L.stride;
    *reinterpret_cast<uint4*>(base + (int64_t)row * L.stride + X) = *reinterpret_cast<const uint4*>(S + X);
    return;
  }
  if (blockIdx.x != 0) return;
  int g = ((int)blockIdx.y - strip_blocks) * 128 + (tid >> 1), lvl = 0;
  while (lvl < PD.nlevels && g >= PD.lv[lvl].h + 2 * kEdge) { g -= PD.lv[lvl].h + 2 * kEdge; lvl++; }
  if (lvl >= PD.nlevels) return;
  const LevelDesc& L = PD.lv[lvl];
  const int w = L.w;
  const bool strip = g < kEdge || g >= L.h + kEdge;
  uint8_t* base = pyr + (int64_t)f * PD.pyr_frame_bytes + L.pyr_off;
  uint8_t* D = base + (int64_t)g * L.stride;
  const uint8_t* S = base + (int64_t)(kEdge + reflect101(g - kEdge, L.h)) * L.stride;   // == D for interior rows
  uint32_t* D32 = reinterpret_cast<uint32_t*>(D);
  const uint32_t* S32 = reinterpret_cast<const uint32_t*>(S);
  if (w < 40) {   // tiny level: the mirror may bounce more than once -> generic byte path
    if (tid & 1) return;
    for (int X = 0; X < w + 2 * kEdge; X++)
      if (strip || X < kEdge || X >= w + kEdge) D[X] = S[kEdge + reflect101(X - kEdge, w)];
    return;
  }
  if ((tid & 1) == 0) {
    uint32_t r[6];
#pragma unroll
    for (int i = 0; i < 6; i++) r[i] = S32[4 + i];          // S[16..39]
    uint32_t o[5];
#pragma unroll
    for (int j = 0; j < 5; j++) o[j] = __builtin_amdgcn_perm(r[5 - j], r[4 - j], 0x03040506u);
    *reinterpret_cast<uint4*>(D) = make_uint4(o[0], o[1], o[2], o[3]);
    if (strip) *reinterpret_cast<uint4*>(D + 16) = make_uint4(o[4], r[1], r[2], r[3]);
    else D32[4] = o[4];
  } else {
    const int t = 2 * w + 36;
    const int Xd0 = (w + kEdge) & ~3, nd = ((w + 37) >> 2) - ((w + kEdge) >> 2) + 1;   // <= 6 dwords
    const int keep = (w + kEdge) - Xd0;                                               // interior bytes of dword 0
    uint32_t v[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const int Xd = Xd0 + 4 * min(j, nd - 1);
      const int s0 = t - Xd - 3;
      const uint32_t A = S32[s0 >> 2], B = S32[(s0 >> 2) + 1];
      v[j] = __builtin_amdgcn_perm(0u, __builtin_amdgcn_alignbyte(B, A, (uint32_t)(s0 & 3)), 0x00010203u);
    }
    if (keep) {
      const uint32_t m = (1u << (8 * keep)) - 1u;
      v[0] = (S32[Xd0 >> 2] & m) | (v[0] & ~m);
    }
    uint32_t c[3];
    const int Xc = 32 + ((w + kEdge - 32) & ~15);   // end of the last full interior 16-byte group (w >= 40)
    const int nc = (Xd0 - Xc) >> 2;                 // interior dwords a strip row still has to copy (0..3)
    if (strip) {
#pragma unroll
      for (int j = 0; j < 3; j++) c[j] = S32[(Xc >> 2) + min(j, max(nc - 1, 0))];
    }
#pragma unroll
    for (int j = 0; j < 6; j++)
      if (j < nd) D32[(Xd0 >> 2) + j] = v[j];
    if (strip) {
#pragma unroll
      for (int j = 0; j < 3; j++)
        if (j < nc) D32[(Xc >> 2) + j] = c[j];
    }
  }
}

// ------------------------------------------------------------------------------------------ K2
// FAST-9/16 strength max(A,B) of one pixel v with ring p[0..15]:
//   A = max over the 16 nine-pixel arcs of min(v - p_i) = v - min_arcs max_i p_i
//   B = max over arcs of min(p_i - v)                   = max_arcs min_i p_i - v
// so the whole sliding-window network runs on the raw ring bytes: window-9 min and max over the circular
// 16-vector on PACKED 16-bit lanes (v_pk_min_u16 / v_pk_max_u16: two ring positions per instruction).  PITCH != 0: compile-time LDS pitch, ring offsets become ds_read immediates.
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ushort2_t pk(int lo, int hi) {
  const uint32_t w = (uint32_t)lo | ((uint32_t)hi << 16);
  return __builtin_bit_cast(ushort2_t, w);
}
template <int PITCH>
__device__ __forceinline__ int fast_strength(const uint8_t* __restrict__ t, int pitch_dyn) {
  constexpr int cx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  constexpr int cy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
  const int pitch = PITCH ? PITCH : pitch_dyn;
  const int v = t[0];
  int p[16];
#pragma unroll
  for (int k = 0; k < 16; k++) p[k] = (int)t[cx[k] + cy[k] * pitch];
  // van Herk / Gil-Werman on the ring split into two blocks of 8, both blocks side by side in the halves of a packed
  // register: X[i] = (p[i], p[8+i]).  The 9-arc starting at k < 8 is p[k..7] (a suffix of block 0) plus p[8..8+k] (a prefix of
  // block 1); the arc starting at 8 + k is the suffix of block 1 plus the prefix of block 0.  So with packed prefix / suffix
  // scans (7 + 7 operations) arc k and arc 8 + k come out of ONE packed operation on (suffix[k], prefix[k] with its halves
  // swapped -- an op_sel modifier, no instruction): 29 packed operations per side instead of 40 for the doubling network.
  ushort2_t X[8];
#pragma unroll
  for (int i = 0; i < 8; i++) X[i] = pk(p[i], p[i + 8]);
  ushort2_t pmin[8], pmax[8], smin[8], smax[8];
  pmin[0] = pmax[0] = X[0];
  smin[7] = smax[7] = X[7];
#pragma unroll
  for (int i = 1; i < 8; i++) {
    pmin[i] = __builtin_elementwise_min(pmin[i - 1], X[i]);
    pmax[i] = __builtin_elementwise_max(pmax[i - 1], X[i]);
    smin[7 - i] = __builtin_elementwise_min(smin[8 - i], X[7 - i]);
    smax[7 - i] = __builtin_elementwise_max(smax[8 - i], X[7 - i]);
  }
  ushort2_t maxmin = pk(0, 0), minmax = pk(255, 255);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const ushort2_t lo9 = __builtin_elementwise_min(smin[k], __builtin_shufflevector(pmin[k], pmin[k], 1, 0));
    const ushort2_t hi9 = __builtin_elementwise_max(smax[k], __builtin_shufflevector(pmax[k], pmax[k], 1, 0));
    maxmin = __builtin_elementwise_max(maxmin, lo9);
    minmax = __builtin_elementwise_min(minmax, hi9);
  }
  const int A = v - min((int)minmax.x, (int)minmax.y), B = max((int)maxmin.x, (int)maxmin.y) - v;
  return max(A, B);
}

// One workgroup per (cell, frame).  Reproduces, for the cell's ROI,
//   FAST(roi, kps, iniThFAST, true); if (kps.empty()) FAST(roi, kps, minThFAST, true);
// using the identity (DESIGN.md "FAST as a score map"): with S(p) = max(A,B)-1 where max(A,B) > tlow
// (0 elsewhere and outside the 3-px ROI frame), cv::FAST(t) with NMS returns exactly
// { p : S(p) >= t and S(p) > S(q) for the 8 neighbours q }, in row-major order, response S(p).
// Output: candidates in the cell's slot range, row-major, packed (x-16, y-16, score).
//
// Structure (work shrinks at every step, wavefronts stay dense; the kernel is VALU-issue bound, so every
// step is written for instruction count):
//   1. ROI tile -> LDS with aligned dword loads; score map zeroed
//   A. every pixel: compass pre-test (two ADJACENT ring positions of {0,4,8,12} both darker than
//      v - tlow or both brighter than v + tlow -- necessary because any 9-arc of the 16-ring holds two
//      adjacent compass pixels), 4 pixels per thread on packed 16-bit lanes.  Each wave owns a contiguous
//      row-major quarter of the items and appends its survivors, in order, to its own list region;
//      the four regions concatenated are the row-major survivor list (indexed virtually, never copied)
//   B. per wave, its own survivors: full FAST strength -> score map; corners (score > 0) compacted in place
//   C. per wave, its own corners: 3x3 strict-maximum test, threshold bits, ordered compaction (one prefix over waves)
// LDS is dynamic and sized for the largest cell of the current image size (FastLds), so the BASELINE
// config needs ~13 KB per workgroup and 8 workgroups (32 waves) stay resident per CU.
struct FastLds {
  int tile_pitch, tile_bytes, score_bytes, plist_bytes, pscore_bytes;
  __host__ __device__ int total() const { return tile_bytes + score_bytes + plist_bytes + pscore_bytes; }
};
// compile-time pitches the kernel is instantiated for (0 = dynamic fallback)
__host__ __device__ inline int fast_pick_pitch(int max_rw) {
  const int need = (max_rw + 3 + 3) & ~3;
  // 64: rows are 16-byte aligned in LDS -> the tile is loaded with dwordx4 / ds_write_b128.  (Every 4th row then shares
  // its banks -- 2.3x the bank-conflict cycles of a 56-byte pitch -- but a conflict-free 80-byte pitch costs a workgroup
  // of occupancy and measured slower: 0.59 vs 0.565 ms.)
  return need <= 64 ? 64 : need;
}
__host__ __device__ inline FastLds fast_lds_layout(int max_rw, int max_rh) {
  FastLds l;
  l.tile_pitch = fast_pick_pitch(max_rw);
  l.tile_bytes = (l.tile_pitch * max_rh + 15) & ~15;
  const int ew = max_rw - 6 > 0 ? max_rw - 6 : 1, eh = max_rh - 6 > 0 ? max_rh - 6 : 1;
  l.score_bytes = ((((ew + 2 + 3) & ~3) * (eh + 2)) + 15) & ~15;
  l.plist_bytes = (((ew + 9) * eh + 16) * 2 + 15) & ~15;   // 4 wave regions of ceil(items/4)*4 entries
  l.pscore_bytes = 0;      // (corner scores are read back from the score map)
  return l;
}
// zero-extend two bytes of the 8-byte pool {a: bytes 4..7, b: bytes 0..3} into the halves of a dword
#define DVM_PERM2(a, b, i0, i1) __builtin_amdgcn_perm((a), (b), 0x0c000c00u | (uint32_t)(i0) | ((uint32_t)(i1) << 16))

// inclusive prefix sum over the 64 lanes of a wave (DPP row shifts, then the row carries)
__device__ __forceinline__ int wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);    // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);    // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);    // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);    // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return x;
}

#ifdef DVM_FAST_DEBUG
__device__ unsigned long long g_fast_dbg[8192 * 8];
__device__ unsigned int g_fast_hist[96];   // [0,32): survivors per wave / 8; [32,64): corners per wave / 4; [64,96): survivors per cell / 8
#define DVM_FSTAMP(i) do { if (threadIdx.x == 0) { const long long t_ = __builtin_readcyclecounter(); unsigned long long* g_ = g_fast_dbg + (blockIdx.x & 8191) * 8; if (i) g_[i] += (unsigned long long)(t_ - t_prev_); else g_[0] += 1ull; t_prev_ = t_; } } while (0)
#else
#define DVM_FSTAMP(i) do { } while (0)
#endif
constexpr int kFastFramesPerWG = 4;   // a workgroup walks the same cell of 4 frames: amortises dispatch + prologue (8: measured slower, 0.61 vs 0.57 ms -- longer tail)
template <int PITCH, int NW>
__device__ __forceinline__ void fast_cell(const uint8_t* __restrict__ pyr, int pyr_frame_bytes, const CellDesc& c,
                                          const PipelineDesc& PD, uint32_t* __restrict__ cand,
                                          int32_t* __restrict__ cell_count, int max_rw, int max_rh, int cell_id, int f) {
  extern __shared__ __attribute__((aligned(16))) uint8_t fast_smem[];
  const FastLds lay = fast_lds_layout(max_rw, max_rh);
  const int kTilePitch = PITCH ? PITCH : lay.tile_pitch;
  uint8_t* tile = fast_smem;
  uint8_t* score = tile + lay.tile_bytes;
  uint16_t* plist = reinterpret_cast<uint16_t*>(score + lay.score_bytes);
  constexpr int NT = 64 * NW;   // NW = 2 for small cells: fewer half-empty rounds and half the per-wave fixed cost
  __shared__ int s_tot[NW];

  const LevelDesc& L = PD.lv[c.level];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef DVM_FAST_DEBUG
  long long t_prev_ = 0;
#endif
  const int rw = c.rw, rh = c.rh;
  const int ew = rw - 6, eh = rh - 6;  // evaluated area (FAST skips a 3-px frame of the ROI)
  const int sp = (ew + 2 + 3) & ~3;    // score pitch (1-px zero ring), multiple of 4
  int32_t* my_count = cell_count + (int64_t)f * PD.ncells + cell_id;
  if (ew <= 0 || eh <= 0) {
    if (tid == 0) *my_count = 0;
    return;
  }
  DVM_FSTAMP(0);
  // ---- 1. tile load: aligned dwords covering each ROI row; LDS column = global column - (x_start & ~3)
  const int64_t row0 = (int64_t)f * pyr_frame_bytes + L.pyr_off + (int64_t)(kEdge + c.y0) * L.stride;
  const int xg = kEdge + c.x0;           // first ROI column inside the bordered row
  const int sh = xg & 3;                 // tile[y][sh + x] holds ROI pixel (x, y)
  const int dwr = (sh + rw + 3) >> 2;    // dwords per row
  const uint32_t* g32 = reinterpret_cast<const uint32_t*>(pyr + row0 + (xg - sh));  // stride is a multiple of 64
  uint32_t* t32 = reinterpret_cast<uint32_t*>(tile);
  const int pitch4 = kTilePitch >> 2;
  if (PITCH == 64) {
    // 16 bytes per thread: an (unaligned) dwordx4 from the pyramid row, one ds_write_b128; a row is at most 4 such
    // items and the last one may run past the ROI (never past the 64-byte LDS row; the pyramid buffer has slack)
    const int q = (dwr + 3) >> 2;
    int y = (int)(((float)tid + 0.5f) * (1.0f / (float)q)), x = tid - y * q;
    const int dy = NT / q, dx = NT - dy * q;
    while (y < rh) {
      uint4 v;
      __builtin_memcpy(&v, g32 + (int64_t)y * (L.stride >> 2) + 4 * x, 16);
      *reinterpret_cast<uint4*>(t32 + y * 16 + 4 * x) = v;
      x += dx; y += dy;
      if (x >= q) { x -= q; y++; }
    }
  } else {
    int y = (int)(((float)tid + 0.5f) * (1.0f / (float)dwr)), x = tid - y * dwr;   // exact for these small ints
    const int dy = NT / dwr, dx = NT - dy * dwr;
    while (y < rh) {
      t32[y * pitch4 + x] = g32[(int64_t)y * (L.stride >> 2) + x];
      x += dx; y += dy;
      if (x >= dwr) { x -= dwr; y++; }
    }
  }
  uint4* s128 = reinterpret_cast<uint4*>(score);           // (score starts 16-byte aligned and its allocation is rounded up to 16)
  for (int i = tid; i < (sp * (eh + 2) + 15) >> 4; i += NT) s128[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  DVM_FSTAMP(1);

  const uint8_t* T = tile + sh;
  // Two thresholds, like the reference: FAST(iniThFAST) first, the whole cell again at minThFAST only if that found nothing.
  // Pass 0 runs the pre-test and the strength network at t = iniTh: a strict local maximum p with S(p) >= iniTh beats every
  // neighbour whose score is below iniTh whether that score is known or counted as 0, so the maxima found among the
  // S >= iniTh pixels are exactly cv::FAST(iniTh, nms) -- and far fewer pixels survive the pre-test than at t = minTh
  // (the strength network, ~150 instructions per survivor, is where this kernel's time goes).  Pass 1 (rare: flat cells,
  // where few pixels survive anyway) repeats A-C at t = minTh; the scores already in the map stay valid (same values).
  int tlow = PD.ini_th;
  int total = 0;
  for (int pass = 0; pass < 2; pass++) {
  // ---- A. compass pre-test.  One item = one LDS dword column of one row = 4 pixels: north / south / centre
  // dwords plus west / east dwords give all five bytes of each pixel; v_perm_b32 zero-extends byte pairs to
  // packed u16, the test itself is 9 packed min/max per pixel pair:
  //   darker  pair (i,j) of adjacent compass points:  v - max(p_i, p_j) > t
  //   brighter:                                         min(p_i, p_j) - v > t
  //   pass  <=>  max(v - min_pairs max, max_pairs min - v) > t
  const int c0 = (sh + 3) >> 2, c1 = (sh + 3 + ew - 1) >> 2, ncol = c1 - c0 + 1;
  const int nitems = eh * ncol;
  const int Q = (nitems + NW - 1) / NW;     // items per wave
  const int roundsA = (Q + 63) >> 6;
  const int it_end = min(nitems, (wave + 1) * Q);
  uint16_t* mylist = plist + wave * Q * 4;
  int wcount = 0;                           // wave-uniform running length of this wave's list
  {
    int it = wave * Q + lane;
    int ey = (int)(((float)it + 0.5f) * (1.0f / (float)ncol)), cc = it - ey * ncol;
    const int dy = 64 / ncol, dc = 64 - dy * ncol;
    const ushort2_t T2 = pk(tlow, tlow);
    // valid pixels of the first / last dword column (evaluated x in [0, ew)), in the survivor-bit layout below
    const int exb0 = 4 * c0 - sh - 3;                       // evaluated x of byte 0 of column c0 (may be < 0)
    const int vlo = max(0, -exb0), vhi = min(4, ew - (exb0 + 4 * (ncol - 1)));
    uint32_t mask_first = 0, mask_last = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t bit = (k & 1 ? 0x80000000u : 0x00008000u) >> (k >> 1);
      if (k >= vlo) mask_first |= bit;
      if (k < vhi) mask_last |= bit;
    }
    if (ncol == 1) mask_first &= mask_last;
    for (int r = 0; r < roundsA; r++) {
      uint32_t pm = 0;
      if (it < it_end) {
        const int c = c0 + cc;
        const uint32_t* row = t32 + (ey + 3) * pitch4 + c;
        const uint32_t Cd = row[0], Wd = row[-1], Ed = row[1], Nd = row[3 * pitch4], Sd = row[-3 * pitch4];
        uint32_t q[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const ushort2_t C = __builtin_bit_cast(ushort2_t, DVM_PERM2(0u, Cd, 2 * h, 2 * h + 1));
          const ushort2_t N = __builtin_bit_cast(ushort2_t, DVM_PERM2(0u, Nd, 2 * h, 2 * h + 1));
          const ushort2_t S = __builtin_bit_cast(ushort2_t, DVM_PERM2(0u, Sd, 2 * h, 2 * h + 1));
          // west neighbour (x-3) of pixel k = byte k+1 of {Cd:Wd}; east (x+3) = byte k+3 of {Ed:Cd}
          const ushort2_t W = __builtin_bit_cast(ushort2_t, DVM_PERM2(Cd, Wd, 2 * h + 1, 2 * h + 2));
          const ushort2_t E = __builtin_bit_cast(ushort2_t, DVM_PERM2(Ed, Cd, 2 * h + 3, 2 * h + 4));
          // the four adjacent compass pairs (N,E) (E,S) (S,W) (W,N) are exactly {N,S} x {E,W}, so
          //   min over pairs of max(x, y) = max(min(N,S), min(E,W)),  max over pairs of min(x, y) = min(max(N,S), max(E,W))
          const ushort2_t nsl = __builtin_elementwise_min(N, S), nsh = __builtin_elementwise_max(N, S);
          const ushort2_t ewl = __builtin_elementwise_min(E, W), ewh = __builtin_elementwise_max(E, W);
          const ushort2_t mx = __builtin_elementwise_max(nsl, ewl);
          const ushort2_t mn = __builtin_elementwise_min(nsh, ewh);
          typedef short short2s __attribute__((ext_vector_type(2)));
          const short2s dk = __builtin_bit_cast(short2s, C) - __builtin_bit_cast(short2s, mx);   // darker margin
          const short2s br = __builtin_bit_cast(short2s, mn) - __builtin_bit_cast(short2s, C);   // brighter margin
          const short2s m = __builtin_elementwise_max(dk, br);
          q[h] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2s, T2) - m);  // sign bit of a half set <=> margin > t
        }
        // survivor bits stay where the sign bits are: pixel 0 -> bit 15, 1 -> bit 31, 2 -> bit 14, 3 -> bit 30
        const uint32_t vmask = cc == 0 ? mask_first : (cc == ncol - 1 ? mask_last : 0xC000C000u);
        pm = ((q[0] & 0x80008000u) | ((q[1] >> 1) & 0x40004000u)) & vmask;
      }
      // ordered append: lane-major, pixel order inside a lane = exclusive prefix sum of the lanes' survivor counts
      const int cnt = __popc(pm);
      const int incl = wave_incl_scan(cnt);
      if (pm) {
        int pos = wcount + incl - cnt;
        const int e0 = (ey << 7) + 4 * cc + exb0;
        if (pm & 0x00008000u) mylist[pos++] = (uint16_t)e0;
        if (pm & 0x80000000u) mylist[pos++] = (uint16_t)(e0 + 1);
        if (pm & 0x00004000u) mylist[pos++] = (uint16_t)(e0 + 2);
        if (pm & 0x40000000u) mylist[pos] = (uint16_t)(e0 + 3);
      }
      const int tot = __builtin_amdgcn_readlane(incl, 63);
      wcount += tot;
      it += 64; cc += dc; ey += dy;
      if (cc >= ncol) { cc -= ncol; ey++; }
    }
  }
  DVM_FSTAMP(2);
#ifdef DVM_FAST_HIST
  if (lane == 0 && pass == 0) atomicAdd(&g_fast_hist[min(wcount >> 3, 31)], 1u);
#endif
  // ---- B. per wave, no barrier: full strength of the wave's OWN survivors.  Corners (score > 0) are compacted in
  // place at the head of the wave's list -- the write position never passes the read position --; their scores live in
  // the score map; the four corner lists concatenated are still row-major.
  int ncorner = 0;   // wave-uniform
  for (int base = 0; base < wcount; base += 64) {
    const int k = base + lane;
    int pe = 0, sc = 0;
    if (k < wcount) {
      pe = mylist[k];
      const int ey = pe >> 7, ex = pe & 127;
      const int m = fast_strength<PITCH>(&T[(ey + 3) * kTilePitch + ex + 3], kTilePitch);
      sc = m > tlow ? m - 1 : 0;
      if (sc) score[(ey + 1) * sp + ex + 1] = (uint8_t)sc;
    }
    const unsigned long long bc = __ballot(sc > 0);
    if (sc > 0) {
      const int pos = ncorner + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bc, 0u));
      mylist[pos] = (uint16_t)pe;
    }
    ncorner += __popcll(bc);
  }
  DVM_FSTAMP(3);
#ifdef DVM_FAST_HIST
  if (lane == 0 && pass == 0) atomicAdd(&g_fast_hist[32 + min(ncorner >> 2, 31)], 1u);
#endif
  __syncthreads();   // the score map is complete
  DVM_FSTAMP(4);
  // ---- C. per wave: strict local maxima among its corners; bit0 = passes iniTh, bit1 = passes minTh
  const int rounds = (ncorner + 63) >> 6;   // <= 32: a wave owns at most 2048 pixels
  uint32_t flags_lo = 0, flags_hi = 0;      // 2 bits per round
  int kept = 0;                             // wave-uniform: strict maxima of this wave (ballots; was an LDS atomic per lane + a barrier of its own)
  for (int r = 0; r < rounds; r++) {
    const int k = r * 64 + lane;
    int fl = 0;
    if (k < ncorner) {
      const int pe = mylist[k];
      const uint8_t* s = &score[((pe >> 7) + 1) * sp + (pe & 127) + 1];
      const int sv = s[0];
      const int mx = max(max(max(s[-sp - 1], s[-sp]), max(s[-sp + 1], s[-1])),
                         max(max(s[1], s[sp - 1]), max(s[sp], s[sp + 1])));
      if (sv > mx) fl = 1;    // every corner of this pass has sv >= tlow + 0: score = max(A,B) - 1 >= tlow
    }
    kept += __popcll(__ballot(fl != 0));
    if (r < 16) flags_lo |= (uint32_t)fl << (2 * r);
    else flags_hi |= (uint32_t)fl << (2 * (r - 16));
  }
  if (lane == 0) s_tot[wave] = kept;
  DVM_FSTAMP(5);
  __syncthreads();
  int wg_kept = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) wg_kept += s_tot[w];
  if (wg_kept == 0) {   // workgroup-uniform: nothing at this threshold
    if (pass == 0 && PD.min_th < PD.ini_th) {
      tlow = PD.min_th;
      __syncthreads();      // everyone has read s_tot / the lists before they are rebuilt
      continue;
    }
    break;
  }
  const int bit = 1;
  // ordered compaction: wave lists in wave order, inside a list in list order = row-major
  int base = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) { if (w < wave) base += s_tot[w]; total += s_tot[w]; }
  uint32_t* out = cand + (int64_t)f * PD.cand_frame_slots + c.cand_base;
  for (int r = 0; r < rounds; r++) {
    const int fl = (r < 16) ? (flags_lo >> (2 * r)) : (flags_hi >> (2 * (r - 16)));
    const bool keep = (fl & bit) != 0;
    const unsigned long long m = __ballot(keep);
    const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (keep && pos < c.cand_cap) {
      const int k = r * 64 + lane;
      const int pe = mylist[k];
      // border-relative level coordinates: (roi origin + 3 + e) - 16
      out[pos] = pack_cand(c.x0 + 3 + (pe & 127) - (kEdge - 3), c.y0 + 3 + (pe >> 7) - (kEdge - 3), score[((pe >> 7) + 1) * sp + (pe & 127) + 1]);
    }
    base += __popcll(m);
  }
    break;
  }
  DVM_FSTAMP(6);
  if (tid == 0) *my_count = min(total, c.cand_cap);
}

// Launch geometry: 1-D grid of cell_num * 8 * ceil(batch / 32) workgroups; workgroup b handles cell
// cell_first + (b >> 3) % cell_num of frames fbase + 8 j (j < 4), fbase = (b >> 3) / cell_num * 32 + (b & 7): the eight
// frames of an XCD group stay on one XCD (see xcd_frame_map), and the cell descriptor is read once.
template <int PITCH, int NW>
__global__ void __launch_bounds__(64 * NW) k_fast_cells(const uint8_t* __restrict__ pyr, int pyr_frame_bytes,
                                                    const CellDesc* __restrict__ cells, PipelineDesc PD,
                                                    uint32_t* __restrict__ cand, int32_t* __restrict__ cell_count,
                                                    int max_rw, int max_rh, int batch, int cell_first, int cell_num) {
  const int b = blockIdx.x, i = b >> 3;
  const int cell_id = cell_first + i % cell_num;
  const int fbase = (i / cell_num) * (8 * kFastFramesPerWG) + (b & 7);
  const CellDesc c = cells[cell_id];
  for (int j = 0; j < kFastFramesPerWG; j++) {
    const int f = fbase + 8 * j;
    if (f >= batch) break;
    if (j) __syncthreads();   // the previous frame's LDS is dead
    fast_cell<PITCH, NW>(pyr, pyr_frame_bytes, c, PD, cand, cell_count, max_rw, max_rh, cell_id, f);
  }
}

// ---- ONE WAVE PER CELL (k_fast_cells1): the same stages for cells whose ROI fits a 48-byte LDS pitch.  A single wave owns the
// whole cell, so nothing is split per wave: no list regions, no per-wave round quantisation of the survivor lists, no barriers between
// the stages, no prefix over waves in the output.  To keep eight of these waves on a SIMD the workgroup's LDS stays under 5 KB:
//   * the survivor list holds ONE pre-test round (<= 256 entries): stage B consumes it right behind the round, in FULL rounds of 64 --
//     what does not fill a round stays in a register (one entry per lane) and goes first in the next one, so the strength network only
//     ever runs a partial round once per cell and pass;
//   * the corner list is capped (kFast1Corners).  A cell with more corners than that (noise) drops the list for this pass and stage C
//     walks the score map pixel by pixel instead: same set, same order, slower.
constexpr int kFast1List = 256, kFast1Corners = 192, kFast1Items = 128;
struct FastLds1 {
  int tile_bytes, score_bytes;
  __host__ __device__ int total() const { return tile_bytes + score_bytes + 2 * (kFast1List + kFast1Corners) + 4 * kFast1Items; }
};
__host__ __device__ inline FastLds1 fast1_lds_layout(int pitch, int max_rw, int max_rh) {
  FastLds1 l;
  l.tile_bytes = (pitch * max_rh + 15) & ~15;
  const int ew = max_rw - 6 > 0 ? max_rw - 6 : 1, eh = max_rh - 6 > 0 ? max_rh - 6 : 1;
  l.score_bytes = ((((ew + 2 + 3) & ~3) * (eh + 2)) + 15) & ~15;
  return l;
}
// pitch 48 holds a row of sh + rw <= 3 + rw bytes (sh = the ROI's misalignment in the pyramid row); wider cells take pitch 64
__host__ __device__ inline int fast1_pick_pitch(int max_rw) { return max_rw + 3 <= 48 ? 48 : (max_rw + 3 <= 64 ? 64 : 0); }

template <int PITCH>
__device__ __forceinline__ void fast_cell1(const uint8_t* __restrict__ pyr, int pyr_frame_bytes, const CellDesc& c,
                                           const PipelineDesc& PD, uint32_t* __restrict__ cand,
                                           int32_t* __restrict__ cell_count, int max_rw, int max_rh, int cell_id, int f) {
  extern __shared__ __attribute__((aligned(16))) uint8_t fast_smem[];
  const FastLds1 lay = fast1_lds_layout(PITCH, max_rw, max_rh);
  uint8_t* tile = fast_smem;
  uint8_t* score = tile + lay.tile_bytes;
  uint32_t* iring = reinterpret_cast<uint32_t*>(score + lay.score_bytes);
  uint16_t* list = reinterpret_cast<uint16_t*>(iring + kFast1Items);
  uint16_t* clist = list + kFast1List;

  const LevelDesc& L = PD.lv[c.level];
  const int lane = threadIdx.x;
  const int rw = c.rw, rh = c.rh;
  const int ew = rw - 6, eh = rh - 6;
  const int sp = (ew + 2 + 3) & ~3;
  int32_t* my_count = cell_count + (int64_t)f * PD.ncells + cell_id;
  if (ew <= 0 || eh <= 0) {
    if (lane == 0) *my_count = 0;
    return;
  }
  // ---- 1. tile load (as fast_cell: 16 bytes per lane and item, rows of PITCH / 16 items)
  const int64_t row0 = (int64_t)f * pyr_frame_bytes + L.pyr_off + (int64_t)(kEdge + c.y0) * L.stride;
  const int xg = kEdge + c.x0;
  const int sh = xg & 3;
  const int dwr = (sh + rw + 3) >> 2;
  const uint32_t* g32 = reinterpret_cast<const uint32_t*>(pyr + row0 + (xg - sh));
  uint32_t* t32 = reinterpret_cast<uint32_t*>(tile);
  constexpr int pitch4 = PITCH >> 2;
  {
    const int q = (dwr + 3) >> 2;
    int y = (int)(((float)lane + 0.5f) * (1.0f / (float)q)), x = lane - y * q;
    const int dy = 64 / q, dx = 64 - dy * q;
    while (y < rh) {
      uint4 v;
      __builtin_memcpy(&v, g32 + (int64_t)y * (L.stride >> 2) + 4 * x, 16);
      *reinterpret_cast<uint4*>(t32 + y * pitch4 + 4 * x) = v;
      x += dx; y += dy;
      if (x >= q) { x -= q; y++; }
    }
  }
  uint4* s128 = reinterpret_cast<uint4*>(score);
  for (int i = lane; i < (sp * (eh + 2) + 15) >> 4; i += 64) s128[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();

  const uint8_t* T = tile + sh;
  uint32_t* out = cand + (int64_t)f * PD.cand_frame_slots + c.cand_base;
  int tlow = PD.ini_th;
  int total = 0;
  for (int pass = 0; pass < 2; pass++) {
    const int c0 = (sh + 3) >> 2, c1 = (sh + 3 + ew - 1) >> 2, ncol = c1 - c0 + 1;
    const int nitems = eh * ncol;
    const int roundsA = (nitems + 63) >> 6;
    int nc = 0;              // corners in clist (wave-uniform)
    bool overflow = false;   // wave-uniform: clist dropped for this pass
    int npend = 0, pend = 0; // survivors waiting for a full round: lane i < npend holds one
    int ihead = 0, itail = 0, nitem = 0;   // the item ring (wave-uniform)
    // stage B on 64 (or, at the end, npend) survivors, one per lane
    auto strength_round = [&](int pe, bool valid) {
      int sc = 0;
      if (valid) {
        const int ey = pe >> 7, ex = pe & 127;
        const int m = fast_strength<PITCH>(&T[(ey + 3) * PITCH + ex + 3], PITCH);
        sc = m > tlow ? m - 1 : 0;
        if (sc) score[(ey + 1) * sp + ex + 1] = (uint8_t)sc;
      }
      const unsigned long long bc = __ballot(sc > 0);
      const int n = __popcll(bc);
      if (nc + n > kFast1Corners) overflow = true;
      if (!overflow) {
        if (sc > 0) clist[nc + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bc, 0u))] = (uint16_t)pe;
        nc += n;
      }
    };
    {
      int it = lane;
      int ey = (int)(((float)it + 0.5f) * (1.0f / (float)ncol)), cc = it - ey * ncol;
      const int dy = 64 / ncol, dc = 64 - dy * ncol;
      const int exb0 = 4 * c0 - sh - 3;
      const int vlo = max(0, -exb0), vhi = min(4, ew - (exb0 + 4 * (ncol - 1)));
      uint32_t mask_first = 0, mask_last = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (k >= vlo) mask_first |= 0x80u << (8 * k);
        if (k < vhi) mask_last |= 0x80u << (8 * k);
      }
      if (ncol == 1) mask_first &= mask_last;
      // A on four pixels per 32-bit operation.  The pre-test only has to let every corner through (stage B is exact), so it runs on
      // the pixels' upper six bits: v - p > t implies (v >> 2) - (p >> 2) >= qt = ceil((t - 2) / 4) -- at t = 20 that lets ~5 % more
      // pixels through than the exact comparison.  With q = p >> 2 in [0, 63] per byte, qt <= 64 and KQ = 128 - qt per byte:
      //   darker   qC - qp >= qt  <=>  bit 7 of (qC + KQ) - qp        (bytes stay in [1, 191]: no borrow into the neighbour)
      //   brighter qp - qC >= qt  <=>  bit 7 of qp + (KQ - qC)        ([1, 191]: no carry)
      // and "two adjacent compass points" = (N | S) & (E | W) on those bits, as in fast_cell.
      const uint32_t KQ = (uint32_t)(128 - min(64, (tlow + 1) >> 2)) * 0x01010101u;
      // survivors of one expansion (<= 64 items of <= 4 pixels) -> stage B in full rounds, the rest waits in `pend`
      auto expand = [&](int n) {
        uint32_t e = 0;
        if (lane < n) e = iring[(ihead + lane) & (kFast1Items - 1)];
        ihead += n;
        const uint32_t m = e & 0x80808080u;
        const int cnt = __popc(m);
        const int incl = wave_incl_scan(cnt);
        if (m) {
          int pos = incl - cnt;
          const int e0 = (int)((e >> 1) & 0x1F80u) + (int)((e & 15u) << 2) + exb0;   // ey << 7 (ey sits at bit 8), + 4 cc
          if (m & 0x00000080u) list[pos++] = (uint16_t)e0;
          if (m & 0x00008000u) list[pos++] = (uint16_t)(e0 + 1);
          if (m & 0x00800000u) list[pos++] = (uint16_t)(e0 + 2);
          if (m & 0x80000000u) list[pos] = (uint16_t)(e0 + 3);
        }
        int avail = __builtin_amdgcn_readlane(incl, 63), base = 0;
        while (npend + avail >= 64) {   // B, full rounds only: the waiting entries first (they come first in row-major order)
          const int pe = lane < npend ? pend : (int)list[base + lane - npend];
          strength_round(pe, true);
          base += 64 - npend; avail -= 64 - npend; npend = 0;
        }
        if (avail) {
          if (lane >= npend && lane < npend + avail) pend = list[base + lane - npend];
          npend += avail;
        }
      };
      for (int r = 0; r < roundsA; r++) {
        uint32_t m = 0;
        if (it < nitems) {
          const int cix = c0 + cc;
          const uint32_t* row = t32 + (ey + 3) * pitch4 + cix;
          const uint32_t Cd = row[0], Wd = row[-1], Ed = row[1], Nd = row[3 * pitch4], Sd = row[-3 * pitch4];
          const uint32_t Wb = __builtin_amdgcn_alignbyte(Cd, Wd, 1u);   // pixels x - 3 of the four
          const uint32_t Eb = __builtin_amdgcn_alignbyte(Ed, Cd, 3u);   // pixels x + 3
          const uint32_t qC = (Cd >> 2) & 0x3F3F3F3Fu, qN = (Nd >> 2) & 0x3F3F3F3Fu, qS = (Sd >> 2) & 0x3F3F3F3Fu;
          const uint32_t qW = (Wb >> 2) & 0x3F3F3F3Fu, qE = (Eb >> 2) & 0x3F3F3F3Fu;
          const uint32_t hi = qC + KQ, lo = KQ - qC;
          const uint32_t dark = ((hi - qN) | (hi - qS)) & ((hi - qE) | (hi - qW));
          const uint32_t bright = ((qN + lo) | (qS + lo)) & ((qE + lo) | (qW + lo));
          const uint32_t vmask = cc == 0 ? mask_first : (cc == ncol - 1 ? mask_last : 0x80808080u);
          m = (dark | bright) & vmask;
        }
        // items with a survivor, in item order, to the ring: flags (bits 7, 15, 23, 31) | ey << 8 | cc
        const unsigned long long bal = __ballot(m != 0);
        if (m) {
          const int pos = itail + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
          iring[pos & (kFast1Items - 1)] = m | ((uint32_t)ey << 8) | (uint32_t)cc;
        }
        const int n_new = __popcll(bal);
        itail += n_new; nitem += n_new;
        if (nitem >= 64) { expand(64); nitem -= 64; }
        it += 64; cc += dc; ey += dy;
        if (cc >= ncol) { cc -= ncol; ey++; }
      }
      if (nitem) expand(nitem);
    }
    if (npend) strength_round(pend, lane < npend);
    __syncthreads();   // (one wave: orders the score-map stores before stage C's loads)
    // ---- C. strict local maxima, written straight to the cell's slots in list order = row-major
    int kept = 0;
    if (!overflow) {
      for (int r = 0; r < nc; r += 64) {
        const int k = r + lane;
        bool keep = false;
        int pe = 0, sv = 0;
        if (k < nc) {
          pe = clist[k];
          const uint8_t* s = &score[((pe >> 7) + 1) * sp + (pe & 127) + 1];
          sv = s[0];
          const int mx = max(max(max(s[-sp - 1], s[-sp]), max(s[-sp + 1], s[-1])),
                             max(max(s[1], s[sp - 1]), max(s[sp], s[sp + 1])));
          keep = sv > mx;
        }
        const unsigned long long m = __ballot(keep);
        const int pos = kept + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (keep && pos < c.cand_cap)
          out[pos] = pack_cand(c.x0 + 3 + (pe & 127) - (kEdge - 3), c.y0 + 3 + (pe >> 7) - (kEdge - 3), sv);
        kept += __popcll(m);
      }
    } else {
      // the score map pixel by pixel (every non-zero score is a corner of this pass: see the pass comment in fast_cell)
      int ey = (int)(((float)lane + 0.5f) * (1.0f / (float)ew)), ex = lane - ey * ew;
      const int dy = 64 / ew, dx = 64 - dy * ew;
      for (int k = 0; k < ew * eh; k += 64) {
        bool keep = false;
        int sv = 0;
        if (ey < eh) {
          const uint8_t* s = &score[(ey + 1) * sp + ex + 1];
          sv = s[0];
          const int mx = max(max(max(s[-sp - 1], s[-sp]), max(s[-sp + 1], s[-1])),
                             max(max(s[1], s[sp - 1]), max(s[sp], s[sp + 1])));
          keep = sv > 0 && sv > mx;
        }
        const unsigned long long m = __ballot(keep);
        const int pos = kept + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (keep && pos < c.cand_cap) out[pos] = pack_cand(c.x0 + 3 + ex - (kEdge - 3), c.y0 + 3 + ey - (kEdge - 3), sv);
        kept += __popcll(m);
        ex += dx; ey += dy;
        if (ex >= ew) { ex -= ew; ey++; }
      }
    }
    total = kept;
    if (kept == 0 && pass == 0 && PD.min_th < PD.ini_th) {
      tlow = PD.min_th;
      __syncthreads();
      continue;
    }
    break;
  }
  if (lane == 0) *my_count = min(total, c.cand_cap);
}

// same launch geometry as k_fast_cells (workgroup = one wave)
template <int PITCH>
__global__ void __launch_bounds__(64, 8) __attribute__((amdgpu_num_sgpr(80))) k_fast_cells1(const uint8_t* __restrict__ pyr, int pyr_frame_bytes,
                                                    const CellDesc* __restrict__ cells, PipelineDesc PD,
                                                    uint32_t* __restrict__ cand, int32_t* __restrict__ cell_count,
                                                    int max_rw, int max_rh, int batch, int cell_first, int cell_num) {
  const int b = blockIdx.x, i = b >> 3;
  const int cell_id = cell_first + i % cell_num;
  const int fbase = (i / cell_num) * (8 * kFastFramesPerWG) + (b & 7);
  const CellDesc c = cells[cell_id];
  for (int j = 0; j < kFastFramesPerWG; j++) {
    const int f = fbase + 8 * j;
    if (f >= batch) break;
    if (j) __syncthreads();
    fast_cell1<PITCH>(pyr, pyr_frame_bytes, c, PD, cand, cell_count, max_rw, max_rh, cell_id, f);
  }
}

#ifdef DVM_FAST_DEBUG
extern "C" int dvm_debug_fast_stamps(unsigned long long* out, int reset) {
  static unsigned long long h[8192 * 8];
  int rc = (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fast_dbg), sizeof(h));
  for (int i = 0; i < 16; i++) out[i] = 0;
  {
    unsigned int hh[96];
    rc |= (int)hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_fast_hist), sizeof(hh));
    for (int i = 0; i < 96; i++) out[16 + i] = hh[i];
    if (reset) { for (auto& v : hh) v = 0; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fast_hist), hh, sizeof(hh)); }
  }
  for (int b = 0; b < 8192; b++) for (int i = 0; i < 8; i++) out[i] += h[b * 8 + i];
  if (reset) { for (auto& v : h) v = 0; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fast_dbg), h, sizeof(h)); }
  return rc;
}
#endif
// ------------------------------------------------------------------------------------ assemble
// operator() output placement (reference ORBextractor.cc:898-951).  One workgroup per frame.
// Walks the selected keypoints in (level, octree-list) order g = 0..N-1, scales pt by
// mvScaleFactor[level] (level>0), and places keypoints inside the lapping area from the back
// (stereoIndex--) and the others from the front (monoIndex++).
__global__ void __launch_bounds__(256) k_assemble(const uint32_t* __restrict__ sel, const int32_t* __restrict__ nsel,
                                                  PipelineDesc PD, int lap0, int lap1, dvm_keypoint_pod* __restrict__ kps,
                                                  KpAux* __restrict__ aux, int32_t* __restrict__ n_out,
                                                  int32_t* __restrict__ mono_out, HostMirror hm) {
  __shared__ int s_lvl_start[kMaxLevels + 1];
  __shared__ int s_wave[2][4];   // lapping keypoints per wave, double-buffered over the 256-keypoint steps: one barrier per step
  const int f = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  if (tid == 0) {
    int acc = 0;
    for (int l = 0; l < PD.nlevels; l++) {
      s_lvl_start[l] = acc;
      acc += min(nsel[f * PD.nlevels + l], PD.lv[l].sel_cap);
    }
    s_lvl_start[PD.nlevels] = min(acc, PD.kp_cap);
  }
  __syncthreads();
  const int N = s_lvl_start[PD.nlevels];
  const uint32_t* S = sel + (int64_t)f * PD.sel_frame_slots;
  dvm_keypoint_pod* K = kps + (int64_t)f * PD.kp_cap;
  KpAux* A = aux + (int64_t)f * PD.kp_cap;
  int carry = 0;   // lapping keypoints before this step (every thread keeps the same count)
  for (int g0 = 0, it = 0; g0 < N; g0 += 256, it++) {
    int g = g0 + tid;
    int lvl = 0, inlap = 0;
    float px = 0, py = 0, resp = 0;
    int cx = 0, cy = 0;
    if (g < N) {
      while (g >= s_lvl_start[lvl + 1]) lvl++;
      int x, y, s;
      unpack_cand(S[PD.lv[lvl].sel_off + (g - s_lvl_start[lvl])], x, y, s);
      cx = x + (kEdge - 3);
      cy = y + (kEdge - 3);
      px = (float)cx;
      py = (float)cy;
      if (lvl != 0) {
        px = px * PD.lv[lvl].scale;
        py = py * PD.lv[lvl].scale;
      }
      resp = (float)s;
      inlap = (px >= (float)lap0 && px <= (float)lap1) ? 1 : 0;
    }
    // lapping keypoints among g' < g: ballot within the wave, the four wave totals through LDS
    const unsigned long long b = __ballot(inlap);
    if (lane == 0) s_wave[it & 1][wv] = __popcll(b);
    __syncthreads();
    int before = __popcll(b & ((1ull << lane) - 1ull)), total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const int c = s_wave[it & 1][w];
      before += w < wv ? c : 0;
      total += c;
    }
    if (g < N) {
      int lap_before = carry + before;
      int pos = inlap ? (N - 1 - lap_before) : (g - lap_before);
      dvm_keypoint_pod kp;
      kp.x = px; kp.y = py;
      kp.size = (float)PD.lv[lvl].patch_size;
      kp.angle = -1.f;
      kp.response = resp;
      kp.octave = lvl;
      kp.class_id = -1;
      K[pos] = kp;
      if (hm.kps) hm.kps[(int64_t)f * PD.kp_cap + pos] = kp;   // latency path: the caller's copy, written over PCIe as it is produced
      KpAux a;
      a.level = (int16_t)lvl; a.pad = 0; a.cx = (int16_t)cx; a.cy = (int16_t)cy; a.out_pos = pos;
      A[g] = a;
    }
    carry += total;
  }
  if (tid == 0) {
    n_out[f] = N;
    mono_out[f] = N - carry;
    if (hm.n) { hm.n[f] = N; hm.mono[f] = N - carry; }
  }
}

// ------------------------------------------------------------------------------------------ K5
// GaussianBlur(7x7, sigma=2, BORDER_REFLECT_101) in OpenCV's 8-bit fixed-point form:
// out = (sum_j sum_i k[j] k[i] src + 32768) >> 16 with the 8.8 kernel [18,34,48,56,48,34,18].
// The pyramid's 19-px REFLECT_101 frame already holds the mirrored pixels, so the tile loader just
// reads the bordered buffer.  Tile = 64 x 32 outputs, LDS: raw (38 x 70) u8 + hpass (38 x 64) u16.
__global__ void __launch_bounds__(256) k_blur7(const uint8_t* __restrict__ pyr, int pyr_frame_bytes,
                                               uint8_t* __restrict__ blur, int blur_frame_bytes,
                                               const TileDesc* __restrict__ tiles, PipelineDesc PD,
                                               const int32_t* __restrict__ lvl_start, int batch) {
  __shared__ __attribute__((aligned(16))) uint8_t raw[(kBlurTH + 6) * kRawPitch];
  __shared__ __attribute__((aligned(16))) uint32_t hpt[kBlurTW * kHtPitch];
  int tile_id, f;
  if (!xcd_frame_map(PD.ntiles, batch, tile_id, f)) return;
  const TileDesc t = tiles[tile_id];
  // the reference skips levels without keypoints (:915-916); a level has keypoints iff it has FAST candidates
  // (the octree keeps >= 1 of n > 0), which is known before the octree runs -> the blur overlaps with it
  if (lvl_start) {   // (null: blur every level -- lets the launch precede FAST; an empty level's blur is simply unused)
    const int32_t* ls = lvl_start + (int64_t)f * (kMaxLevels + 1) + t.level;
    if (ls[1] == ls[0]) return;
  }
  blur_tile(pyr, pyr_frame_bytes, blur, blur_frame_bytes, t, PD, f, (uint32_t)c_gauss7[0], (uint32_t)c_gauss7[1], (uint32_t)c_gauss7[2], (uint32_t)c_gauss7[3], raw, hpt);
}

// -------------------------------------------------------------------------------------- K4 + K6
// cv::fastAtan2 scalar path (degrees), no FMA.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
  const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
  const float eps = (float)2.2204460492503131e-16;
  float ax = fabsf(x), ay = fabsf(y), a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// (float)cos / (float)sin of angle_deg * (float)(pi/180): DESIGN.md "sincos spec" (double Cody-Waite
// reduction by pi/2 + fdlibm kernel polynomials, Horner with separate mul/add).
__device__ __forceinline__ void sincos_deg(float angle_deg, float& cs, float& sn) {
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  float angf = angle_deg * factorPI;
  double x = (double)angf;
  double kd = rint(x * 0.63661977236758134308);
  int k = (int)kd;
  double r = (x - kd * 1.57079632679489655800e+00) - kd * 6.12323399573676603587e-17;
  double z = r * r;
  double ps = -1.66666666666666324348e-01 +
              z * (8.33333333332248946124e-03 +
                   z * (-1.98412698298579493134e-04 +
                        z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  double pc = 4.16666666666666019037e-02 +
              z * (-1.38888888888741095749e-03 +
                   z * (2.48015872894767294178e-05 +
                        z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  double s = r + r * (z * ps);
  double c = (1.0 - 0.5 * z) + (z * z) * pc;
  double co, si;
  switch (k & 3) {
    case 0: co = c; si = s; break;
    case 1: co = -s; si = c; break;
    case 2: co = -c; si = -s; break;
    default: co = s; si = -c; break;
  }
  cs = (float)co;
  sn = (float)si;
}

// wave64 integer sum: 4 DPP steps inside each row of 16 lanes, then 4 v_readlane + scalar adds (wave-uniform result)
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);   // row_mirror
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}

// One wave per keypoint (4 keypoints per workgroup).  Phase 1: intensity-centroid moments over the 749-px disc of the
// UN-blurred level (int32, exact, order-free), fastAtan2.  Phase 2: 256 BRIEF tests on the blurred level, pair
// p = 64*i + lane, packed by 4 wave ballots (bit p%8 of byte p/8).  The kernel used to be bound by the texture path (20 byte-gather instructions per wave, every
// lane on its own cache line); now every memory instruction of a wave covers whole row segments:
//   IC_Angle: the 31 x 31 patch is read as 31 rows x 8 UNALIGNED dwords (4 instructions); the moments are
//             v_dot4_u32_u8 against the disc weights:  m10 = sum (u+16) I - 16 sum I,  m01 = sum v I  (exact integers)
//   rBRIEF:   the 37 x 37 blurred patch (pattern radius 18.4) is staged in LDS as 37 rows x 10 dwords (6 instructions);
//             the 512 rotated samples are LDS byte reads; 256 tests = 4 ballots of 64 lanes
constexpr int kDescR = 18, kDescPitch = 40, kDescRows = 2 * kDescR + 1;
constexpr int kDescPerWave = 2;   // keypoints a wave handles one after the other (8 per workgroup).  4 made this kernel itself 12 % faster
                                  // (more loads in flight) but held 24 KB of LDS per workgroup, which the stages of the other pipeline lane
                                  // could not use: the 256-frame stream ran 222.7 k frames/s with 4, 227-228 k with 2, 222.4 k with 1
// A wave walks kDescPerWave keypoints: all their loads go out first (blurred patches -> LDS, IC rows -> registers), then the
// pairs of moments are reduced, then LANES 0..kDescPerWave-1 run fastAtan2 + the double-precision sincos for the wave's keypoints AT ONCE
// -- that part is ~100 instructions of wave-uniform floating point per keypoint when every keypoint has its own wave --, then
// the four descriptors.  The pattern points (as floats) and the disc weights stay in registers across the keypoints.
__global__ void __launch_bounds__(256) k_orient_desc(const uint8_t* __restrict__ pyr, int pyr_frame_bytes,
                                                     const uint8_t* __restrict__ blur, int blur_frame_bytes,
                                                     PipelineDesc PD, const KpAux* __restrict__ aux,
                                                     const int32_t* __restrict__ n_kp, dvm_keypoint_pod* __restrict__ kps,
                                                     uint8_t* __restrict__ desc, int batch, HostMirror hm) {
  __shared__ __attribute__((aligned(16))) uint8_t s_patch[4][kDescPerWave][kDescRows * kDescPitch];
  int blk, f;
  if (!xcd_frame_map((PD.kp_cap + 4 * kDescPerWave - 1) / (4 * kDescPerWave), batch, blk, f)) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g0 = (blk * 4 + wave) * kDescPerWave;
  const int nkf = n_kp[f];
  if (nkf - blk * 4 * kDescPerWave <= 0) return;           // the whole workgroup is past the frame's keypoints
  const int nk = max(min(kDescPerWave, nkf - g0), 0);      // wave-uniform; a wave without keypoints stays for the barriers below
  KpAux a[kDescPerWave];                                   // (and walks keypoint 0 of the workgroup: valid data, nothing stored)
#pragma unroll
  for (int q = 0; q < kDescPerWave; q++) a[q] = aux[(int64_t)f * PD.kp_cap + (nk > 0 ? g0 + min(q, nk - 1) : blk * 4 * kDescPerWave)];
  // ---- blurred patches -> LDS (flat addressing of the blurred level; addresses are clamped to the level so the unused
  // corner bytes of edge keypoints never leave the buffer)
  // (registers first, LDS afterwards: with the store inside the load loop the second keypoint's loads waited for the first one's)
  uint32_t pv[kDescPerWave][6];
#pragma unroll
  for (int q = 0; q < kDescPerWave; q++) {
    const LevelDesc& L = PD.lv[a[q].level];
    const uint8_t* bl = blur + (int64_t)f * blur_frame_bytes + L.blur_off;
    const int st = L.blur_stride;
    const int bmax = st * L.h - 4;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const int item = min(lane + 64 * i, kDescRows * 10 - 1);
      const int r = (item * 205) >> 11, c = item - 10 * r;   // item / 10 for item < 1024
      const int off = min(max((a[q].cy - kDescR + r) * st + a[q].cx - kDescR + 4 * c, 0), bmax);
      __builtin_memcpy(&pv[q][i], bl + off, 4);
    }
  }
#pragma unroll
  for (int q = 0; q < kDescPerWave; q++) {
    uint8_t* P = s_patch[wave][q];
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const int item = lane + 64 * i;
      const int r = (item * 205) >> 11, c = item - 10 * r;
      if (item < kDescRows * 10) *reinterpret_cast<uint32_t*>(P + r * kDescPitch + 4 * c) = pv[q][i];
    }
  }
  // ---- IC_Angle: the 31 x 31 patch as 31 rows x 8 unaligned dwords; moments = v_dot4_u32_u8 against the disc weights
  uint2 w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) w[i] = c_disc_w[lane + 64 * i];
  int m10[kDescPerWave], m01[kDescPerWave];
#pragma unroll
  for (int q = 0; q < kDescPerWave; q++) {
    const LevelDesc& L = PD.lv[a[q].level];
    const uint8_t* c0 = pyr + (int64_t)f * pyr_frame_bytes + L.pyr_off + (int64_t)(kEdge + a[q].cy) * L.stride + kEdge + a[q].cx;
    int s10 = 0, s01 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int item = lane + 64 * i;            // rows 0..30 (item 248..255: weights are zero, row clamped)
      const int r = min(item >> 3, 30), c = item & 7;
      uint32_t v;
      __builtin_memcpy(&v, c0 + (r - kHalfPatch) * L.stride + 4 * c - kHalfPatch, 4);
      const int s1 = (int)__builtin_amdgcn_udot4(v, w[i].y, 0u, false);
      const int su = (int)__builtin_amdgcn_udot4(v, w[i].x, 0u, false);
      s10 += su - 16 * s1;
      s01 += (r - kHalfPatch) * s1;
    }
    m10[q] = wave_sum_i32(s10);
    m01[q] = wave_sum_i32(s01);
  }
  // ---- angle and rotation: fastAtan2 + the double-precision sincos are ~150 instructions of one-lane floating point per
  // keypoint; every wave running them for its own two keypoints spent a third of the kernel's issue slots on two lanes.  The
  // moments of the workgroup's 4 x kDescPerWave keypoints meet in LDS and ONE wave computes all the angles, a lane each.
  __shared__ int s_mom[4 * kDescPerWave][2];
  __shared__ float s_rot[4 * kDescPerWave][3];
#pragma unroll
  for (int q = 0; q < kDescPerWave; q++)
    if (lane == q) { s_mom[wave * kDescPerWave + q][0] = m10[q]; s_mom[wave * kDescPerWave + q][1] = m01[q]; }
  __syncthreads();
  if (threadIdx.x < 4 * kDescPerWave) {
    float ang, ca_, sb_;
    ang = fast_atan2_deg((float)s_mom[threadIdx.x][1], (float)s_mom[threadIdx.x][0]);
    sincos_deg(ang, ca_, sb_);
    s_rot[threadIdx.x][0] = ang; s_rot[threadIdx.x][1] = ca_; s_rot[threadIdx.x][2] = sb_;
  }
  __syncthreads();
  float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int* pt = &c_pattern[(64 * i + lane) * 4];
    px0[i] = (float)pt[0]; py0[i] = (float)pt[1]; px1[i] = (float)pt[2]; py1[i] = (float)pt[3];
  }
#pragma unroll
  for (int q = 0; q < kDescPerWave; q++) {
    if (q >= nk) break;
    const float angle = s_rot[wave * kDescPerWave + q][0], ca = s_rot[wave * kDescPerWave + q][1], sb = s_rot[wave * kDescPerWave + q][2];
    const uint8_t* c1 = s_patch[wave][q] + kDescR * kDescPitch + kDescR;
    unsigned long long words[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int t0 = c1[__float2int_rn(px0[i] * sb + py0[i] * ca) * kDescPitch + __float2int_rn(px0[i] * ca - py0[i] * sb)];
      const int t1 = c1[__float2int_rn(px1[i] * sb + py1[i] * ca) * kDescPitch + __float2int_rn(px1[i] * ca - py1[i] * sb)];
      words[i] = __ballot(t0 < t1);
    }
    if (lane == 0) {
      kps[(int64_t)f * PD.kp_cap + a[q].out_pos].angle = angle;
      unsigned long long* d = reinterpret_cast<unsigned long long*>(desc + ((int64_t)f * PD.kp_cap + a[q].out_pos) * 32);
      d[0] = words[0]; d[1] = words[1]; d[2] = words[2]; d[3] = words[3];
      if (hm.kps) {
        hm.kps[(int64_t)f * PD.kp_cap + a[q].out_pos].angle = angle;
        unsigned long long* h = reinterpret_cast<unsigned long long*>(hm.desc + ((int64_t)f * PD.kp_cap + a[q].out_pos) * 32);
        h[0] = words[0]; h[1] = words[1]; h[2] = words[2]; h[3] = words[3];
      }
    }
  }
}

// ------------------------------------------------------------------------------------- launchers
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

bool raise_dynamic_lds(const void* fn, int bytes) {
  static std::mutex mtx;
  static std::map<std::pair<int, const void*>, int> cap;   // (device, kernel) -> bytes granted so far
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(mtx);
  int& have = cap[{dev, fn}];
  if (bytes <= have) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  have = bytes;
  return true;
}

void upload_constants(const int8_t* disc_u, const int8_t* disc_v, const int* gauss7) {
  uint2 w[256];
  for (int i = 0; i < 256; i++) w[i] = make_uint2(0u, 0u);
  for (int p = 0; p < kDiscPixels; p++) {
    const int u = disc_u[p], v = disc_v[p];
    const int item = (v + kHalfPatch) * 8 + ((u + kHalfPatch) >> 2), k = (u + kHalfPatch) & 3;
    w[item].x |= (uint32_t)(u + 16) << (8 * k);
    w[item].y |= 1u << (8 * k);
  }
  hipMemcpyToSymbol(HIP_SYMBOL(c_disc_w), w, sizeof(w));
  hipMemcpyToSymbol(HIP_SYMBOL(c_gauss7), gauss7, 7 * sizeof(int));
}

void launch_pyr_level0(hipStream_t s, const uint8_t* d_src, int rows, int cols, int sstride, int64_t frame_stride,
                       uint8_t* d_pyr, const PipelineDesc& PD, int batch) {
  const LevelDesc& L = PD.lv[0];
  dim3 grid(cdiv(cdiv(L.w + 2 * kEdge, 16), 64), cdiv(L.h, 16), batch);
  hipLaunchKernelGGL(k_pyr_level0, grid, dim3(256), 0, s, d_src, rows, cols, sstride, frame_stride, d_pyr,
                     PD.pyr_frame_bytes, L);
}
void launch_pyr_resize(hipStream_t s, uint8_t* d_pyr, const PipelineDesc& PD, int level, const int32_t* d_tabs, int batch) {
  const LevelDesc& L = PD.lv[level];
  const LevelDesc& P = PD.lv[level - 1];
  // LDS for the source rectangle of a 256 x 16 destination tile (scale = P/L, +margins)
  const double sxs = (double)P.w / L.w, sys = (double)P.h / L.h;
  const int pitch = (((int)(kRzTW * sxs) + 16) + 3) & ~3;
  if (batch <= 8) {   // latency path: short tiles
    const int nrows = (int)(16 * sys) + 4;
    dim3 grid(cdiv(L.w + 2 * kEdge, kRzTW), cdiv(L.h, 16), batch);
    hipLaunchKernelGGL(k_pyr_resize<16>, grid, dim3(256), (size_t)pitch * nrows, s, d_pyr, PD.pyr_frame_bytes, P, L, d_tabs, pitch);
    return;
  }
  const int nrows = (int)(kRzTH * sys) + 4;
  dim3 grid(cdiv(L.w + 2 * kEdge, kRzTW), cdiv(L.h, kRzTH), batch);
  hipLaunchKernelGGL(k_pyr_resize<kRzTH>, grid, dim3(256), (size_t)pitch * nrows, s, d_pyr, PD.pyr_frame_bytes, P, L, d_tabs, pitch);
}
void launch_pyr_borders(hipStream_t s, uint8_t* d_pyr, const PipelineDesc& PD, int batch) {
  int rows = 0, maxw = 0;
  for (int l = 0; l < PD.nlevels; l++) { rows += PD.lv[l].h + 2 * kEdge; maxw = std::max(maxw, PD.lv[l].w + 2 * kEdge); }
  const int strip_blocks = cdiv(2 * kEdge * PD.nlevels, 4);
  dim3 grid(std::max(1, cdiv(cdiv(maxw, 16), 64)), strip_blocks + cdiv(rows, 128), batch);
  hipLaunchKernelGGL(k_pyr_borders, grid, dim3(256), 0, s, d_pyr, PD, strip_blocks);
}
// the two-wave (four-wave) form, one launch over a cell range
static void launch_fast_wide(hipStream_t s, const uint8_t* d_pyr, const CellDesc* d_cells, const PipelineDesc& PD, uint32_t* d_cand,
                             int32_t* d_cell_count, int batch, int max_rw, int max_rh, int cell_first, int cell_num) {
  const FastLds lay = fast_lds_layout(max_rw, max_rh);
  const dim3 grid(cell_num * 8 * ((batch + 8 * kFastFramesPerWG - 1) / (8 * kFastFramesPerWG)));
  // two waves per cell while 32 survivor rounds of 128 cover the largest cell, else four
  const bool small = (max_rw - 6) * (max_rh - 6) <= 32 * 128;
  // (beyond the default 48 KB of dynamic LDS the limit is raised on the current device: per device, so per launch)
  static const int lds_pad = std::getenv("DVM_FAST_LDS_PAD") ? atoi(std::getenv("DVM_FAST_LDS_PAD")) : 0;   /* experiment */
#define DVM_FAST_LAUNCH(P, W)                                                                                              \
  do {                                                                                                                     \
    if (lay.total() > 48 * 1024)                                                                                           \
      raise_dynamic_lds(reinterpret_cast<const void*>(k_fast_cells<P, W>), lay.total());                                   \
    hipLaunchKernelGGL((k_fast_cells<P, W>), grid, dim3(64 * W), lay.total() + lds_pad, s, d_pyr, PD.pyr_frame_bytes, d_cells, PD,   \
                       d_cand, d_cell_count, max_rw, max_rh, batch, cell_first, cell_num);                                 \
  } while (0)
  // (one wave per cell was measured too: occupancy-bound, 0.76 ms vs 0.68 ms for two)
  if (lay.tile_pitch == 64) { if (small) DVM_FAST_LAUNCH(64, 2); else DVM_FAST_LAUNCH(64, 4); }
  else DVM_FAST_LAUNCH(0, 4);
#undef DVM_FAST_LAUNCH
}
// Levels whose cells fit k_fast_cells1's small LDS layout (one wave per cell, eight waves per SIMD) take that kernel; the others (the
// coarse levels' taller cells at 640 x 480, any level of a large-cell configuration) the two-wave form.  Consecutive levels of one kind
// share a launch.  h_cells = host copy of the cell table (nullptr: the wide form for everything).
void launch_fast(hipStream_t s, const uint8_t* d_pyr, const CellDesc* d_cells, const PipelineDesc& PD, uint32_t* d_cand,
                 int32_t* d_cell_count, int batch, int max_rw, int max_rh, int cell_first, int cell_num, const CellDesc* h_cells) {
  if (cell_num <= 0) return;
  // LDS is granted in steps of 1 280 bytes: <= 5 120 keeps 32 one-wave workgroups on a CU.  (Measured and not adopted: a second tier
  // <= 6 400 bytes, or a 64-byte pitch, for the taller cells of the coarse levels -- two more launches with short grids, 0.51 ms
  // against 0.39; and one polarity per list entry with half the strength network -- the wider expansion eats what stage B saves.)
  // DVM_FAST1_LDS=0: the wide form only (A/B)
  static const int lds1_max = std::getenv("DVM_FAST1_LDS") ? atoi(std::getenv("DVM_FAST1_LDS")) : 5120;
  static const int lds1_max64 = std::getenv("DVM_FAST1_LDS64") ? atoi(std::getenv("DVM_FAST1_LDS64")) : 0;
  if (!h_cells || lds1_max <= 0) { launch_fast_wide(s, d_pyr, d_cells, PD, d_cand, d_cell_count, batch, max_rw, max_rh, cell_first, cell_num); return; }
  const int end = cell_first + cell_num;
  int run_first = cell_first, run_kind = -1, run_rw = 0, run_rh = 0;
  auto flush = [&](int upto) {
    if (upto <= run_first || run_kind < 0) return;
    const int n = upto - run_first;
    if (run_kind == 0) { launch_fast_wide(s, d_pyr, d_cells, PD, d_cand, d_cell_count, batch, run_rw, run_rh, run_first, n); return; }
    const int pitch = run_kind;
    const int lds = fast1_lds_layout(pitch, run_rw, run_rh).total();
    const dim3 grid(n * 8 * ((batch + 8 * kFastFramesPerWG - 1) / (8 * kFastFramesPerWG)));
    if (pitch == 48)
      hipLaunchKernelGGL((k_fast_cells1<48>), grid, dim3(64), lds, s, d_pyr, PD.pyr_frame_bytes, d_cells, PD, d_cand, d_cell_count, run_rw, run_rh, batch, run_first, n);
    else
      hipLaunchKernelGGL((k_fast_cells1<64>), grid, dim3(64), lds, s, d_pyr, PD.pyr_frame_bytes, d_cells, PD, d_cand, d_cell_count, run_rw, run_rh, batch, run_first, n);
  };
  int c = cell_first;
  while (c < end) {
    const int lvl = h_cells[c].level;
    int e = c, rw = 8, rh = 8;
    for (; e < end && h_cells[e].level == lvl; e++) { rw = std::max<int>(rw, h_cells[e].rw); rh = std::max<int>(rh, h_cells[e].rh); }
    const int pitch = fast1_pick_pitch(rw);
    // (a list entry packs the evaluated x into 7 bits and y above it: the same limits as the wide form's)
    // kind: 0 = the wide form, else the one-wave form's LDS pitch (levels of different pitch do not share a launch: the narrow
    // layout is what keeps eight waves on a SIMD)
    const int kind = pitch != 0 && fast1_lds_layout(pitch, rw, rh).total() <= (pitch == 48 ? lds1_max : lds1_max64) ? pitch : 0;
    if (kind != 0 && kind == run_kind) {   // joining the run must not push the run's own layout over the limit
      const int jrw = std::max(run_rw, rw), jrh = std::max(run_rh, rh);
      if (fast1_lds_layout(kind, jrw, jrh).total() > (kind == 48 ? lds1_max : lds1_max64)) { flush(c); run_first = c; run_kind = -1; }
    }
    if (kind != run_kind) { flush(c); run_first = c; run_kind = kind; run_rw = rw; run_rh = rh; }
    else { run_rw = std::max(run_rw, rw); run_rh = std::max(run_rh, rh); }
    c = e;
  }
  flush(end);
}
void launch_assemble(hipStream_t s, const uint32_t* d_sel, const int32_t* d_nsel, const PipelineDesc& PD, int lap0,
                     int lap1, dvm_keypoint_pod* d_kps, KpAux* d_aux, int32_t* d_n, int32_t* d_mono, int batch, HostMirror hm) {
  hipLaunchKernelGGL(k_assemble, dim3(batch), dim3(256), 0, s, d_sel, d_nsel, PD, lap0, lap1, d_kps, d_aux, d_n, d_mono, hm);
}
void launch_blur(hipStream_t s, const uint8_t* d_pyr, uint8_t* d_blur, const TileDesc* d_tiles, const PipelineDesc& PD,
                 const int32_t* d_lvl_start, int batch) {
  hipLaunchKernelGGL(k_blur7, dim3(xcd_grid(PD.ntiles, batch)), dim3(256), 0, s, d_pyr, PD.pyr_frame_bytes, d_blur,
                     PD.blur_frame_bytes, d_tiles, PD, d_lvl_start, batch);
}
void launch_orient_desc(hipStream_t s, const uint8_t* d_pyr, const uint8_t* d_blur, const PipelineDesc& PD,
                        const KpAux* d_aux, const int32_t* d_n, dvm_keypoint_pod* d_kps, uint8_t* d_desc, int batch, HostMirror hm) {
  hipLaunchKernelGGL(k_orient_desc, dim3(xcd_grid(cdiv(PD.kp_cap, 4 * kDescPerWave), batch)), dim3(256), 0, s, d_pyr, PD.pyr_frame_bytes,
                     d_blur, PD.blur_frame_bytes, PD, d_aux, d_n, d_kps, d_desc, batch, hm);
}

}  // namespace dvm
