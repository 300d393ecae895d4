// dvm_slam_amd/csrc/orb_pipeline.h -- host side of the ORB extractor (handle behind dvm_orb_*).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/dvmslam_hip.h"
#include "orb_device.h"
#include "orb_kernels.h"

namespace dvm {

void set_error(const std::string& msg);
const char* last_error_cstr();   // the calling thread's last error text
int hip_check(hipError_t e, const char* what);
#define DVM_HIP(call)                                  \
  do {                                                 \
    int _rc = ::dvm::hip_check((call), #call);         \
    if (_rc != DVM_OK) return _rc;                     \
  } while (0)

// HIP-event stopwatch: one (start, stop) pair per timed launch group on the owning stream.
class Profiler {
 public:
  bool enabled = false;
  bool only_fast = false;   // dvm_orb_profiling(h, 2): HIP events around the dominant kernel's bracket only -- every event record is a marker
                            // packet between two kernels of the batch (a few microseconds of idle device each; twelve per batch otherwise)
  void begin(hipStream_t s, const char* name);
  void end(hipStream_t s);
  void resolve();  // after a stream sync: fold finished pairs into the totals
  void reset();
  bool get(const std::string& name, double* ms, int64_t* launches);
  ~Profiler();

 private:
  struct Pending { std::string name; hipEvent_t a, b; hipStream_t stream; bool closed; bool live; };
  std::vector<Pending> pending_;
  std::vector<hipEvent_t> pool_;
  std::map<std::string, std::pair<double, int64_t>> totals_;
  hipEvent_t get_event();
};

// DistributeOctTree (reference ORBextractor.cc:419-610) on the host: array formulation, see .cpp
#ifdef DVM_DEBUG
void octree_select(const uint32_t* cand, int n, int minX, int maxX, int minY, int maxY, int N, std::vector<uint32_t>& out);
#endif

class OrbPipeline {
 public:
  OrbPipeline(const dvm_orb_params& p, int device, int max_batch);
  ~OrbPipeline();
  int init();  // device + stream + constants
  int extract_device(const uint8_t* d_imgs, int batch, int rows, int cols, int stride, int64_t frame_stride, int lap0,
                     int lap1);
  int extract_host(const uint8_t* imgs, int batch, int rows, int cols, int stride, int64_t frame_stride, int lap0,
                   int lap1);
  int ensure_stage(size_t need);
  hipStream_t copy_stream = nullptr;            // H2D of the staged ingest (dvm_orb_extract_staged)
  hipEvent_t ev_copied = nullptr, ev_stage_free = nullptr;
  bool copy_pending = false, stage_free_valid = false;
  int staging(int batch, int rows, int cols, uint8_t** host_ptr);
  int extract_staged(int batch, int rows, int cols, int lap0, int lap1);
  int extract_staged_sync_owner(int batch, int rows, int cols, int lap0, int lap1);   // few frames: read in place (no H2D copy)
  int sync();
  int download(int frame, dvm_keypoint* kps, uint8_t* desc, int cap, int* n, int* mono);

  dvm_orb_params params;
  int device, max_batch;
  hipStream_t stream = nullptr;
  // A batch is cut into `chunks` frame ranges that run as a software pipeline: chunk c's main chain on lane c & 1
  // (lane 0 = `stream`), its blur on that lane's side stream; chunk c+1 starts when chunk c has left the
  // throughput-bound stages (pyramid, FAST), so they overlap chunk c's latency-bound k_octree.
  static constexpr int kMaxChunks = 8;
  hipStream_t lane_main[2] = {nullptr, nullptr}, lane_side[2] = {nullptr, nullptr};
  // Optional FAST / octree pipelining over level groups [0,a) [a,b) [b,L) (DVM_GROUPS=a,b): the octree of a group runs
  // on lane_main[1] under the FAST cells of the next group.  Measured 1.56 vs 1.58 ms per 256-frame step (the octree is
  // VALU work too, only its latency hides), so the default stays ONE k_fast_cells launch per batch -- which is also what
  // the roofline line of bench.py and the rocprofv3 summary describe.
  int group_split[2] = {kMaxLevels, kMaxLevels};
  hipEvent_t ev_group[4] = {};   // FAST level group g done (0..2) / auxiliary-stream octrees done (3)
  hipEvent_t ev_start = nullptr, ev_compact[kMaxChunks] = {}, ev_fork[kMaxChunks] = {}, ev_join[kMaxChunks] = {}, ev_done = nullptr;
  int chunks = 1;            // DVM_CHUNKS=n; measured on MI355X at batch 256: 1 -> 1.82 ms, 2 -> 1.93 ms, 4 -> 2.11 ms per
                             // step (concurrent queues do not recover the k_octree idle time), so the default is off
  bool blur_early = true;    // launch the blur right after the pyramid (all levels) instead of after the candidate counts
  bool overlap_blur = true;  // DVM_SERIAL=1 puts the blur back on `stream`
  int side_priority = 0;     // lowest stream priority of the device (the blur's side stream)
  Profiler prof;
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> nfeat;
  int umax[16];

  PipelineDesc PD{};
  std::vector<CellDesc> cells;
  std::vector<TileDesc> tiles;
  int last_batch = 0;
  int max_cell_rw = 8, max_cell_rh = 8;  // largest FAST cell ROI of the current image size
  bool tiny_levels = false;  // some level is too small for the fused single-bounce frame -> k_pyr_borders fixes it up
  bool configured = false;

  // device memory
  uint8_t* d_pyr = nullptr;
  uint8_t* d_blur = nullptr;
  int32_t* d_tabs = nullptr;
  CellDesc* d_cells = nullptr;
  TileDesc* d_tiles = nullptr;
  uint32_t* d_cand = nullptr;        // cell-slotted
  uint32_t* d_dense = nullptr;       // per-frame vToDistributeKeys, level l at its own cand_off (written by k_octree)
  int32_t* d_cell_count = nullptr;
  int32_t* d_lvl_count = nullptr;    // [batch][kMaxLevels] candidates per level (written by k_octree)
  uint32_t* d_sel = nullptr;         // [batch][sel_frame_slots]
  int32_t* d_nsel = nullptr;         // [batch][nlevels]
  dvm_keypoint_pod* d_kps = nullptr; // [batch][kp_cap]
  uint8_t* d_desc = nullptr;         // [batch][kp_cap][32]
  KpAux* d_aux = nullptr;
  int32_t* d_n = nullptr;            // [batch]
  int32_t* d_mono = nullptr;
  int32_t* d_nid = nullptr;          // [batch][cand_frame_slots] octree scratch: node id per candidate
  int32_t* d_err = nullptr;          // octree capacity flag: device address of ...
  int32_t* h_err = nullptr;          // ... this word of mapped host memory
  bool host_octree = false;          // -DDVM_DEBUG builds only: DistributeOctTree on the host instead of k_octree (DVM_HOST_OCTREE=1);
  bool host_octree_forced = false;   // always false in a release library
  uint8_t* d_stage = nullptr;        // staging for host images
  size_t stage_bytes = 0;
  // pinned host mirrors
  int32_t* h_cell_count = nullptr;   // host octree fallback: per-cell counts
  uint32_t* h_dense = nullptr;
  uint32_t* h_sel = nullptr;
  int32_t* h_nsel = nullptr;
  int32_t* h_n = nullptr;
  int32_t* h_mono = nullptr;
  uint8_t* h_stage = nullptr;
  // Latency path -- a call of at most kLatencyBatch frames (Tracking hands over ONE): nothing is bandwidth-bound at that size, the
  // call is a chain of ~13 dependent launches plus the copies around it (0.229 ms host to host before, 0.157 ms now).  So
  // (1) level 0 reads the image from the pinned staging buffer over PCIe (no H2D copy in front of the chain);
  // (2) no side stream: the fork / join events around it cost more than the 9 us the blur takes (see (5));
  // (3) k_assemble / k_orient_desc also store counts, keypoints and descriptors into mapped host memory, so download() is a
  //     stream synchronisation and a memcpy;
  // (4) k_octree<true>: wave-synchronous rounds, the level's keys and node ids in LDS (octree_rounds_wave.inc);
  // (5) the blur's tiles ride in the octree's launch (k_octree_blur): they need the pyramid only and fill the chip the octree's eight
  //     latency-bound workgroups leave idle -- off the chain without a second stream.
  // DVM_LATENCY_PATH=0 / DVM_ZERO_COPY_IN=0: A-B switches.  Tried and not kept: the pyramid in one launch with inter-workgroup
  // flags (write-through stores + per-row-tile counters: 56 us against 45 us for the eight launches -- a cross-XCD hand-off costs
  // more than a kernel boundary); three pyramid levels per launch, a tile recomputing the rectangles of the levels between its
  // group's base and itself in LDS (bit-identical, 15 us per three-level group against 3 x 5 us: no gain at 256 or 1 024 threads);
  // the whole call as a captured hipGraph (0.183 ms against 0.180: replay is no cheaper than 17 eager calls);
  // level 0's FAST + octree on a second stream behind k_pyr_level0 (DVM_LAT_SPLIT=1, 0.187 ms: the host issues the second
  // chain's launches in front of the first one's, and in a graph the branches serialised: 0.34 ms).
  static constexpr int kLatencyBatch = 4;
  bool latency_path = true, zero_copy_in = true;
  bool lat_split = false;            // opt-in, see above
  bool oct_blur = true;              // (5) the blur as extra workgroups of the octree launch (k_octree_blur); DVM_OCT_BLUR=0: A-B switch
  int gauss7[7] = {};                // the blur's 8.8 fixed-point kernel (also in constant memory for k_blur7)
  hipStream_t lat_aux = nullptr;     // DVM_LAT_SPLIT=1 only: created by the first small call
  bool last_mirrored = false;        // the last batch's results are in h_kps_m / h_desc_m / h_n / h_mono
  dvm_keypoint_pod* h_kps_b = nullptr;   // download_batch: page-locked [frames][kp_cap] blocks, grown on demand
  uint8_t* h_desc_b = nullptr;
  size_t h_batch_cap = 0;
  int download_batch(int count, dvm_keypoint* const* kps, uint8_t* const* desc, const int* caps, int* n, int* mono);
  dvm_keypoint_pod* h_kps_m = nullptr;   // [kLatencyBatch][kp_cap], mapped
  uint8_t* h_desc_m = nullptr;           // [kLatencyBatch][kp_cap][32], mapped
  HostMirror mirror_dev;                 // their device addresses (and those of h_n / h_mono)
  uint8_t* stage_view = nullptr;         // device address of h_stage

 private:
  int configure(int rows, int cols);
  void free_all();
};

}  // namespace dvm
