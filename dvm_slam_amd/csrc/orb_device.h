// dvm_slam_amd/csrc/orb_device.h -- shared host/device descriptors of the ORB pipeline (gfx950).
#pragma once
#include <stdint.h>

namespace dvm {

constexpr int kEdge = 19;        // EDGE_THRESHOLD, reference ORBextractor.cc:72
constexpr int kBlurTW = 64, kBlurTH = 64;  // k_blur7 output tile
constexpr int kHalfPatch = 15;   // HALF_PATCH_SIZE, :71
constexpr int kMaxLevels = 16;
constexpr int kMaxCellDim = 96;  // ROI side (wCell+6) upper bound: wCell < 70 for W=35 cells
constexpr int kDiscPixels = 749; // pixels of the radius-15 orientation disc (umax table)

// One pyramid level of one frame inside the per-frame pyramid block.
struct LevelDesc {
  int32_t w, h;          // interior size
  int32_t stride;        // row pitch of the bordered buffer (multiple of 64)
  int32_t pyr_off;       // byte offset of the bordered buffer inside the frame's pyramid block
  int32_t blur_off;      // byte offset of the blurred image (pitch = blur_stride) inside the blur block
  int32_t blur_stride;
  int32_t tab_off;       // int32 offset of this level's resize tables (xofs,xa,yofs,yb) ; level>0
  int32_t cand_off;      // first candidate slot of this level inside the frame's candidate block
  int32_t cand_cap;      // candidate slots of this level
  int32_t cell_first, cell_count;  // range in the cell table
  int32_t quota;         // mnFeaturesPerLevel
  int32_t sel_off;       // first slot of this level in the per-frame selected-keypoint block
  int32_t sel_cap;
  float scale;           // mvScaleFactor[level]
  int32_t patch_size;    // (int)(31 * scale)
};

// One FAST cell (reference ORBextractor.cc:634-692): ROI [x0,x0+rw) x [y0,y0+rh) in level pixels.
struct CellDesc {
  int16_t level, pad;
  int16_t x0, y0, rw, rh;
  int32_t cand_base;     // slot base inside the frame's cell-slotted candidate block
  int32_t cand_cap;
};

// One blur tile.
struct TileDesc {
  int16_t level, x0, y0, pad;
};

struct PipelineDesc {
  int32_t nlevels, ncells, ntiles;
  int32_t rows, cols;
  int32_t ini_th, min_th;
  int32_t pyr_frame_bytes, blur_frame_bytes, cand_frame_slots, sel_frame_slots, kp_cap;
  LevelDesc lv[kMaxLevels];
};

// candidate packing: x (12 bits) | y (12 bits) << 12 | score << 24 ; x,y relative to minBorder (=16)
__host__ __device__ inline uint32_t pack_cand(int x, int y, int s) {
  return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)s << 24);
}
__host__ __device__ inline void unpack_cand(uint32_t c, int& x, int& y, int& s) {
  x = (int)(c & 0xFFFu);
  y = (int)((c >> 12) & 0xFFFu);
  s = (int)(c >> 24);
}

// per-keypoint work item for the orientation / descriptor kernel
struct KpAux {
  int16_t level, pad;
  int16_t cx, cy;   // cvRound(pt) in level pixels
  int32_t out_pos;  // row in the output arrays
};

}  // namespace dvm
