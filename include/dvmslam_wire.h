/* include/dvmslam_wire.h -- "DVMW" map wire format: the keyframes and map points one agent ships to another, as one
 * flat block of fixed-size records + pooled SoA sections that can be assembled in, sent from and consumed in HBM.
 *
 * Replaces the Boost binary archive of ORB_SLAM3::KeyFrame / MapPoint that DVM-SLAM publishes on its
 * ROS topics (reference src/slam_system/orb_slam3/include/KeyFrame.h:57-194 and include/MapPoint.h:50-103 `serialize`,
 * src/Atlas.cc:325-346 pre/post-save, src/slam_system/src/orb_slam3_wrapper.cpp:212-384 sender, :386-455 receiver).
 * Every field the monocular, non-inertial path serialises has a slot here (pose, calibration, scale pyramid constants,
 * keypoints, descriptors, BoW / feature vectors, per-keypoint map point uuid, covisibility / spanning-tree / loop /
 * merge links by uuid, map point geometry, descriptor and observations by uuid); stereo, fisheye and IMU members are
 * not carried.  The Frame grid (mGrid) is not shipped: the receiver rebuilds it on the device with dvm_frame_build
 * straight from the keypoint section.
 *
 * Layout (all little-endian, every section starts on a 64-byte boundary, in this order):
 *   0 header            dvm_wire_header                       64 B
 *   1 keyframes         dvm_wire_keyframe  x n_keyframes      192 B each
 *   2 map points        dvm_wire_mappoint  x n_mappoints      160 B each
 *   3 keypoints         dvm_keypoint       x n_keypoints      pooled; keyframe i owns [kp_off, kp_off + n_kp)
 *   4 descriptors       32 B               x n_keypoints      same indexing
 *   5 keypoint -> map point uuid  dvm_uuid x n_keypoints      all zero = no map point
 *   6 bow word ids      int32              x n_bow            ascending per keyframe ([bow_off, bow_off + n_bow))
 *   7 bow values        double             x n_bow
 *   8 feature-vector nodes (node id, n features) int32 pairs x n_fv_nodes
 *   9 feature-vector features  int32       x n_fv_feats       concatenated in node order ([fv_feat_off, ...))
 *  10 links             dvm_wire_link      x n_links          covisible keyframes (weight), children, loop / merge edges
 *  11 observations      dvm_wire_obs       x n_obs            map point i owns [obs_off, obs_off + n_obs)
 */
#ifndef DVMSLAM_WIRE_H
#define DVMSLAM_WIRE_H
#include <stdint.h>

#include "dvmslam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DVM_WIRE_MAGIC 0x574D5644u /* "DVMW" */
#define DVM_WIRE_VERSION 1u
#define DVM_WIRE_SECTIONS 12

typedef struct { uint8_t b[16]; } dvm_uuid;

typedef struct {
  uint32_t magic, version;
  uint64_t total_bytes;
  uint32_t n_keyframes, n_mappoints, n_keypoints, n_bow, n_fv_nodes, n_fv_feats, n_links, n_obs;
  uint32_t sender_agent, flags;
  uint32_t reserved[2];
} dvm_wire_header; /* 64 B */

enum { DVM_WIRE_KF_BAD = 1, DVM_WIRE_KF_NOT_ERASE = 2, DVM_WIRE_KF_FIRST_CONNECTION = 4 };
typedef struct {
  dvm_uuid uuid, parent_uuid;       /* KeyFrame::uuid, mBackupParentUuid */
  uint64_t mn_id, frame_id;         /* mnId, mnFrameId */
  double timestamp;                 /* mTimeStamp */
  float tcw[3], qcw[4];             /* mTcw: translation, unit quaternion (x, y, z, w) */
  float fx, fy, cx, cy;
  float min_x, max_x, min_y, max_y; /* mnMinX .. mnMaxY */
  float scale_factor, log_scale_factor;
  int32_t n_levels, creator_agent, origin_map_id;
  uint32_t flags;
  uint32_t n_kp, kp_off;
  uint32_t n_bow, bow_off, n_fv_nodes, fv_node_off, fv_feat_off, n_links, link_off;
  uint32_t reserved[4];
} dvm_wire_keyframe; /* 192 B */

enum { DVM_WIRE_MP_BAD = 1 };
typedef struct {
  dvm_uuid uuid, ref_kf_uuid, replaced_uuid; /* MapPoint::uuid, mBackupRefKFUuid, mBackupReplacedUuid */
  uint64_t mn_id, first_kf_id;
  float pos[3], normal[3], min_distance, max_distance;
  uint8_t descriptor[32];
  int32_t creator_agent;
  uint32_t flags, n_obs, obs_off;
  uint32_t reserved[4];
} dvm_wire_mappoint; /* 160 B */

enum { DVM_WIRE_LINK_COVISIBLE = 0, DVM_WIRE_LINK_CHILD = 1, DVM_WIRE_LINK_LOOP = 2, DVM_WIRE_LINK_MERGE = 3 };
typedef struct { dvm_uuid uuid; int32_t weight, kind; } dvm_wire_link; /* 24 B */
typedef struct { dvm_uuid kf_uuid; int32_t index, index_right; } dvm_wire_obs; /* 24 B */

typedef struct { uint64_t offset[DVM_WIRE_SECTIONS], bytes[DVM_WIRE_SECTIONS], total_bytes; } dvm_wire_layout_t;

/* Section offsets / sizes implied by the counts of `h` (magic, version and total_bytes are not read). */
int dvm_wire_layout(const dvm_wire_header* h, dvm_wire_layout_t* out);

/* Assemble a block in host memory.  `counts` supplies the n_* fields, sender_agent and flags; section sources that are
 * NULL are left zero (e.g. keypoints / descriptors that dvm_wire_gather_keypoints fills on the device afterwards; with
 * `head_only` != 0 only sections 0-2 are written and out_bytes may be just their size).  Validates the per-record
 * ranges.  Returns DVM_OK, DVM_ERR_INVALID (bad range / count) or DVM_ERR_CAPACITY (out_bytes too small). */
int dvm_wire_build(const dvm_wire_header* counts, const dvm_wire_keyframe* kfs, const dvm_wire_mappoint* mps,
                   const dvm_keypoint* kps, const uint8_t* desc, const dvm_uuid* kp_mappoint, const int32_t* bow_ids,
                   const double* bow_vals, const int32_t* fv_nodes, const int32_t* fv_feats, const dvm_wire_link* links,
                   const dvm_wire_obs* obs, int head_only, void* out, uint64_t out_bytes);

/* Check a received block (host memory): magic, version, size, every record's ranges inside its section, ascending BoW
 * ids per keyframe.  DVM_OK or DVM_ERR_INVALID (dvm_last_error says what). */
int dvm_wire_validate(const void* block, uint64_t bytes);

/* Device side of the sender: the head of the block (sections 0-2, from dvm_wire_build(head_only)) must already be in
 * d_block; copies the keypoints and descriptors of keyframes [first_kf, first_kf + count) from the extractor's device
 * result arrays (frame j of the batch at d_kps + j * kps_stride elements / d_desc + j * desc_stride bytes, as
 * dvm_orb_result_device exposes them) into sections 3 and 4 -- n_kp and kp_off are read from the records in d_block.
 * Asynchronous on `stream`. */
int dvm_wire_gather_keypoints(void* d_block, int first_kf, int count, const dvm_keypoint* d_kps, int64_t kps_stride,
                              const uint8_t* d_desc, int64_t desc_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif
