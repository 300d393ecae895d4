// dvm_slam_amd/csrc/orb_kernels.h -- launchers of the ORB front-end kernels (orb_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orb_device.h"

namespace dvm {

// same layout as dvm_keypoint / cv::KeyPoint
struct dvm_keypoint_pod {
  float x, y, size, angle, response;
  int32_t octave, class_id;
};

// Latency path (a handful of frames per call): the caller's copies of the results in mapped pinned host memory, written by the
// kernels that produce them (posted PCIe writes) instead of by D2H copies queued behind them.  All null: HBM only.
struct HostMirror {
  dvm_keypoint_pod* kps = nullptr;
  uint8_t* desc = nullptr;
  int32_t* n = nullptr;
  int32_t* mono = nullptr;
};

void upload_constants(const int8_t* disc_u, const int8_t* disc_v, const int* gauss7);
void launch_pyr_level0(hipStream_t s, const uint8_t* d_src, int rows, int cols, int sstride, int64_t frame_stride,
                       uint8_t* d_pyr, const PipelineDesc& PD, int batch);
void launch_pyr_resize(hipStream_t s, uint8_t* d_pyr, const PipelineDesc& PD, int level, const int32_t* d_tabs, int batch);
void launch_pyr_borders(hipStream_t s, uint8_t* d_pyr, const PipelineDesc& PD, int batch);
void launch_fast(hipStream_t s, const uint8_t* d_pyr, const CellDesc* d_cells, const PipelineDesc& PD, uint32_t* d_cand,
                 int32_t* d_cell_count, int batch, int max_rw, int max_rh, int cell_first, int cell_num, const CellDesc* h_cells);
int octree_root_nodes(const LevelDesc& L);
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-(device, kernel) cap shared by every handle and host thread: only ever
// RAISE it, so that a handle with a smaller configuration cannot pull the cap under a launch another thread is about to make.
bool raise_dynamic_lds(const void* fn, int bytes);
bool octree_fits_device(const PipelineDesc& PD);
bool octree_prepare_device(const PipelineDesc& PD);   // raises the LDS limit for this configuration; false: cannot run it   // else: DistributeOctTree runs on the host for this configuration
void launch_octree(hipStream_t s, const uint32_t* d_cand, const int32_t* d_cell_count, const CellDesc* d_cells, uint32_t* d_dense,
                   int32_t* d_lvl_count, const PipelineDesc& PD, int32_t* d_nid, uint32_t* d_sel, int32_t* d_nsel, int32_t* d_err,
                   int batch, int level_first, int level_num, bool latency = false);
bool octree_blur_fits(const PipelineDesc& PD);
void launch_octree_blur(hipStream_t s, const uint32_t* d_cand, const int32_t* d_cell_count, const CellDesc* d_cells, uint32_t* d_dense,
                        int32_t* d_lvl_count, const PipelineDesc& PD, int32_t* d_nid, uint32_t* d_sel, int32_t* d_nsel, int32_t* d_err, int batch,
                        const uint8_t* d_pyr, uint8_t* d_blur, const TileDesc* d_tiles, const int* gauss7);
void launch_assemble(hipStream_t s, const uint32_t* d_sel, const int32_t* d_nsel, const PipelineDesc& PD, int lap0,
                     int lap1, dvm_keypoint_pod* d_kps, KpAux* d_aux, int32_t* d_n, int32_t* d_mono, int batch,
                     HostMirror hm = HostMirror());
void launch_blur(hipStream_t s, const uint8_t* d_pyr, uint8_t* d_blur, const TileDesc* d_tiles, const PipelineDesc& PD,
                 const int32_t* d_lvl_start, int batch);
void launch_orient_desc(hipStream_t s, const uint8_t* d_pyr, const uint8_t* d_blur, const PipelineDesc& PD,
                        const KpAux* d_aux, const int32_t* d_n, dvm_keypoint_pod* d_kps, uint8_t* d_desc, int batch,
                        HostMirror hm = HostMirror());

}  // namespace dvm
