/* dvmslam_host.h -- C interface of libdvmslam_host.so: the host-side mirrors of the reference's matcher / vocabulary /
 * keyframe-database FUNCTIONS (one batched device call through libdvmslam_hip.so + the reference's sequential bookkeeping
 * replayed on the host), for callers that cannot include the C++ classes of dvm_slam_amd/host/ (orb_matcher.h,
 * orb_vocabulary.h, keyframe_database.h -- what the C++ shims of the reference tree use directly).  Each entry point names the
 * reference function it stands for (paths relative to src/slam_system/orb_slam3/).  Plain pointers and sizes, int status
 * (>= 0: the reference function's return value; < 0: dvm_status, text from dvm_last_error()), nothing thrown.
 *
 * The view structs are the members of ORB_SLAM3::Frame / KeyFrame / MapPoint the functions read, by the reference's names;
 * map points appear as small integer ids (-1 = NULL). */
#ifndef DVMSLAM_HOST_H
#define DVMSLAM_HOST_H
#include "dvmslam_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* DBoW2::FeatureVector flattened: node ids ascending, the features of node k are feat[off[k] .. off[k+1]) */
typedef struct dvmh_feature_vector_view { int32_t n; const int32_t* node; const int32_t* off; const int32_t* feat; } dvmh_feature_vector_view;
/* the members of ORB_SLAM3::Frame the matcher touches (monocular; include/Frame.h:221-251) */
typedef struct dvmh_frame_view {
  int32_t N;
  const dvm_keypoint* mvKeysUn;
  const uint8_t* mDescriptors;           /* N x 32 */
  int32_t* mvpMapPoints;                 /* map point id per keypoint, -1 = NULL */
  const uint8_t* mvbOutlier;             /* may be NULL */
  dvm_se3f Tcw;                          /* GetPose() */
  float fx, fy, cx, cy;
  float mnMinX, mnMaxX, mnMinY, mnMaxY;
  const float* mvScaleFactors;
  int32_t nLevels;
  const dvm_device_frame* dev;           /* may be NULL.  The extractor's reference to these very keypoints + descriptors in HBM
                                            (dvm_orb_last_result): valid only while mvKeysUn == mvKeys (no distortion) -- the feature
                                            grid is then built from the device arrays instead of uploading them again */
} dvmh_frame_view;
/* the members of ORB_SLAM3::KeyFrame the matcher touches (monocular; include/KeyFrame.h) */
typedef struct dvmh_keyframe_view {
  int32_t N;
  const dvm_keypoint* mvKeysUn;
  const uint8_t* mDescriptors;
  int32_t* mvpMapPoints;                 /* GetMapPointMatches() */
  const uint8_t* mpBad;                  /* isBad() of that map point; may be NULL */
  dvmh_feature_vector_view mFeatVec;
  dvm_se3f Tcw, Twc;                     /* GetPose(), GetPoseInverse() */
  float fx, fy, cx, cy;
  float mnMinX, mnMaxX, mnMinY, mnMaxY;
  const float* mvScaleFactors;
  const float* mvLevelSigma2;
  const float* mvInvLevelSigma2;
  float mfLogScaleFactor;
  int32_t nLevels;
} dvmh_keyframe_view;
/* a list of map points as the projection searches read them (GetWorldPos, GetNormal, mfMin/MaxDistance, GetDescriptor, isBad) */
typedef struct dvmh_map_points_view {
  int32_t n;
  const int32_t* id;
  const uint8_t* bad;                    /* may be NULL */
  const float* pos;                      /* 3n */
  const float* normal;                   /* 3n */
  const float* min_dist;
  const float* max_dist;
  const uint8_t* desc;                   /* 32n */
} dvmh_map_points_view;
/* a last-frame map point of SearchByProjection(CurrentFrame, LastFrame): GetWorldPos, GetDescriptor, Observations() */
typedef struct dvmh_map_point { float pos[3]; uint8_t desc[32]; int32_t n_obs; } dvmh_map_point;
/* a local map point as Tracking::SearchLocalPoints leaves it: the mTrack* members Frame::isInFrustum wrote, isBad, descriptor, Observations() */
typedef struct dvmh_tracked_point {
  float mTrackProjX, mTrackProjY, mTrackDepth, mTrackViewCos;
  int32_t mnTrackScaleLevel;
  uint8_t mbTrackInView, bad, pad_[2];
  uint8_t desc[32];
  int32_t n_obs;
} dvmh_tracked_point;

/* ---- ORBmatcher (src/ORBmatcher.cc).  `requeried` (may be NULL): queries re-evaluated on the host because an earlier match claimed their keypoint */
/* SearchByProjection(CurrentFrame, LastFrame, th, bMono = true), :1553-1748.  K = fx fy cx cy, bounds = minX maxX minY maxY */
int dvmh_search_by_projection_frames(int device, int Nc, const dvm_keypoint* kps_c, const uint8_t* desc_c, int32_t* mp_c, const dvm_se3f* Tcw,
                                     const float* K, const float* bounds, const float* scale_factors, int nlevels, int Nl, const dvm_keypoint* kps_l,
                                     const int32_t* mp_l, const uint8_t* outlier_l, const dvmh_map_point* mps, float th, int check_ori, int* requeried);
/* Several agents on one GPU: route the grid build + window search of every SearchByProjection(CurrentFrame, LastFrame) of this process
 * through a shared search service (dvm_match_pool_create, include/dvmslam_hip.h); NULL: back to one staged call per thread.  Frames beyond
 * the pool's capacity, and frames still in HBM, keep the per-thread call.  The pool must outlive the calls. */
void dvmh_set_match_pool(dvm_match_pool* pool);
/* the same with the current frame's keypoints + descriptors still in HBM where the extractor left them (dvm_orb_last_result): the grid is
 * built from there when the reference is valid (*grid_from_device = 1), from the host arrays otherwise */
int dvmh_search_by_projection_frames_dev(int device, int Nc, const dvm_keypoint* kps_c, const uint8_t* desc_c, int32_t* mp_c, const dvm_se3f* Tcw,
                                         const float* K, const float* bounds, const float* scale_factors, int nlevels, int Nl, const dvm_keypoint* kps_l,
                                         const int32_t* mp_l, const uint8_t* outlier_l, const dvmh_map_point* mps, float th, int check_ori, int* requeried,
                                         const dvm_device_frame* dev_c, int* grid_from_device);
/* ---- One tracked frame as one device chain: Frame::Frame -> ExtractORB (src/Frame.cc:371-411) + Tracking::TrackWithMotionModel
 * (src/Tracking.cc:2584-2667) over dvm_track_begin / dvm_track_finish (include/dvmslam_hip.h).  The extraction is queued first; the
 * projection queries of LastFrame's map points are built on the host while it runs; grid, window search, claim replay, rotation check,
 * PoseOptimization and the outlier flags follow on the same stream; ONE synchronisation.  The doubled window of Tracking.cc:2616-2624
 * (fewer than 20 matches) and the rare host replay of the claims are handled here.  Results equal dvm_orb_extract +
 * dvmh_search_by_projection_frames + dvm_pose_optimize + the outlier drop, bit for bit.
 *   Tcw_pred = mVelocity * mLastFrame.GetPose() (what TrackWithMotionModel sets on CurrentFrame); K = fx fy cx cy; dist may be NULL
 *   out: kps / desc [cap] (mvKeys, mDescriptors), kps_un [cap] (mvKeysUn; may be NULL), mp_c [cap] = CurrentFrame.mvpMapPoints after the
 *   outlier drop (index into mps or -1), dropped [cap] = the map point whose match PoseOptimization rejected or -1 (the caller's
 *   pMP->mbTrackInView = false / mnLastFrameSeen bookkeeping, Tracking.cc:2645-2652) */
typedef struct {
  int32_t n, mono_index;       /* extraction */
  int32_t nmatches;            /* after the outlier drop (Tracking.cc:2653) */
  int32_t nmatches_search;     /* SearchByProjection's return value (of the doubled window if that ran) */
  int32_t nmatches_map;        /* nmatchesMap */
  int32_t n_inliers;           /* PoseOptimization's return value */
  int32_t wide_window;         /* 1: the doubled window was searched */
  int32_t replayed_on_host;    /* 1: the claims were replayed on the host (not produced any more: see n_requeried) */
  int32_t tracked;             /* 0: fewer than 20 matches even with the doubled window (TrackWithMotionModel returns false; pose untouched) */
  dvm_se3f Tcw;                /* the optimised pose as CurrentFrame.SetPose receives it */
  double pose[7];              /* the same as PoseOptimization's doubles (tx ty tz qx qy qz qw) */
  int32_t n_requeried;         /* queries whose window the device searched again at their turn (all four ranked candidates taken) */
  int32_t pad_;
} dvmh_track_result;
int dvmh_track_with_motion_model(dvm_tracker* t, dvm_orb* h, int device, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1,
                                 const dvm_se3f* Tcw_pred, const float* K, const float* bounds, const dvm_distortion* dist,
                                 const float* scale_factors, const float* inv_level_sigma2, int nlevels, int Nl, const dvm_keypoint* kps_l,
                                 const int32_t* mp_l, const uint8_t* outlier_l, const dvmh_map_point* mps, float th, int check_ori,
                                 dvm_keypoint* kps, uint8_t* desc, int cap, dvm_keypoint* kps_un, int32_t* mp_c, int32_t* dropped,
                                 dvmh_track_result* out);

/* The same for `count` frames at once -- the frames of several agents sharing the GPU at one camera tick (dvm_tracker_create_batch with
 * max_frames >= count; an extractor handle with max_batch >= count): ONE chain of batched launches behind ONE synchronisation, the frames'
 * queries built by a few host threads while the batch extraction runs.  imgs: frame b at imgs + b * frame_stride.  The frames share K,
 * bounds, the level tables, th and check_ori (one camera model per call); distortion is not applied (k1 != 0: the single call).
 * Frame b's outputs equal dvmh_track_with_motion_model on that frame alone, bit for bit. */
typedef struct {
  const dvm_se3f* Tcw_pred;                  /* mVelocity * mLastFrame.GetPose() of this agent */
  int32_t Nl; const dvm_keypoint* kps_l; const int32_t* mp_l; const uint8_t* outlier_l;   /* its LastFrame */
  const dvmh_map_point* mps;                 /* its map points */
} dvmh_track_in;
typedef struct {
  dvm_keypoint* kps; uint8_t* desc; int32_t cap; dvm_keypoint* kps_un;   /* [cap]; kps_un may be NULL */
  int32_t *mp_c, *dropped;                   /* [cap] as in the single call */
} dvmh_track_out;
/* imgs == NULL: the frames are already in the extractor's page-locked input buffer (dvm_orb_staging, tight rows) */
int dvmh_track_with_motion_model_batch(dvm_tracker* t, dvm_orb* h, int device, int count, const uint8_t* imgs, int rows, int cols, int stride,
                                       int64_t frame_stride, int lap0, int lap1, const float* K, const float* bounds, const float* scale_factors,
                                       const float* inv_level_sigma2, int nlevels, float th, int check_ori, const dvmh_track_in* in,
                                       const dvmh_track_out* outs, dvmh_track_result* res);

/* SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints), :44-205.  claimed_obs[j] != 0 <=> F.mvpMapPoints[j]->Observations() > 0 */
int dvmh_search_by_projection_points(int device, int N, const dvm_keypoint* kps, const uint8_t* desc, int32_t* mp, const uint8_t* claimed_obs,
                                     const float* bounds, const float* scale_factors, int nlevels, const dvmh_tracked_point* pts, int npts, float th,
                                     float nnratio, int far_points, float th_far, int* requeried);
/* SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), :605-707 */
int dvmh_search_for_initialization(int device, const dvmh_frame_view* F1, const dvmh_frame_view* F2, float* prev_matched, int32_t* matches12, int window,
                                   float nnratio, int check_ori);
/* SearchByBoW(pKF, F, vpMapPointMatches), :214-393 and SearchByBoW(pKF1, pKF2, vpMatches12), :709-834 */
int dvmh_search_by_bow_kf_frame(int device, const dvmh_keyframe_view* KF, const dvmh_frame_view* F, const dvmh_feature_vector_view* Ffv, float nnratio,
                                int check_ori, int32_t* matches, int* requeried);
int dvmh_search_by_bow_kf_kf(int device, const dvmh_keyframe_view* KF1, const dvmh_keyframe_view* KF2, float nnratio, int check_ori, int32_t* matches12,
                             int* requeried);
/* SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo = false, bCoarse), :836-1058; pairs: up to KF1->N (idx1, idx2) */
int dvmh_search_for_triangulation(int device, const dvmh_keyframe_view* KF1, const dvmh_keyframe_view* KF2, int coarse, int check_ori, int32_t* pairs);
/* the geometry it derives from the two poses (:841-862, CameraModels/Pinhole.cpp:106-110): R12 [9], t12 [3], epipole [2], F12 [9] */
void dvmh_triangulation_geometry(const dvmh_keyframe_view* KF1, const dvmh_keyframe_view* KF2, float* R12, float* t12, float* ep, float* F12);
/* Fuse(pKF, vpMapPoints, th), :1060-1234, search part: best_idx[i] = keypoint the i-th point would fuse into (-1: none) */
int dvmh_fuse(int device, const dvmh_keyframe_view* KF, const dvmh_map_points_view* P, const uint8_t* inKF, float th, int32_t* best_idx);
/* Fuse(pKF, Scw, vpPoints, th, vpReplacePoint), :1236-1345: KF->mvpMapPoints receives the added points, replace[i] the id to replace */
int dvmh_fuse_sim3(int device, dvmh_keyframe_view* KF, const dvm_sim3f* Scw, const dvmh_map_points_view* P, float th, int32_t* replace);
/* SearchByProjection(pKF, Scw, vpPoints, [vpPointsKFs,] vpMatched, [vpMatchedKF,] th, ratioHamming), :395-603; point_kf / matched_kf may be NULL */
int dvmh_search_by_projection_sim3(int device, const dvmh_keyframe_view* KF, const dvm_sim3f* Scw, const dvmh_map_points_view* P, const int32_t* point_kf,
                                   int32_t* matched, int32_t* matched_kf, int th, float ratio_hamming, int* requeried);
/* SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist), :1750-1860 (relocalisation); already: ascending ids */
int dvmh_search_by_projection_reloc(int device, dvmh_frame_view* Cur, const dvmh_keyframe_view* KF, const dvmh_map_points_view* P, const int32_t* already,
                                    int n_already, float th, int orb_dist, int check_ori, int* requeried);
/* SearchBySim3(pKF1, pKF2, vpMatches12, S12, th), :1347-1551; idx_in_kf2[i] = get<0>(vpMatches12[i]->GetIndexInKeyFrame(pKF2)) or NULL */
int dvmh_search_by_sim3(int device, const dvmh_keyframe_view* KF1, const dvmh_keyframe_view* KF2, const dvmh_map_points_view* P1,
                        const dvmh_map_points_view* P2, int32_t* matches12, const int32_t* idx_in_kf2, const dvm_sim3f* S12, float th);

/* ---- pose arithmetic in the reference's float operation order (Sophus 1.x over Eigen 3.4; csrc/pose_f32.h) */
void dvmh_pose_matrices(const dvm_se3f* Tcw, float* Rcw, float* tcw, float* Ow);   /* Frame::UpdatePoseMatrices, src/Frame.cc:553-559 */
void dvmh_se3_inverse(const dvm_se3f* T, dvm_se3f* out);
void dvmh_sim3_to_se3(const dvm_sim3f* S, dvm_se3f* Tcw, float* Ow);                /* ORBmatcher.cc:403-404 */
void dvmh_se3_apply(const dvm_se3f* T, const float* p, int n, float* out);
void dvmh_sim3_apply(const dvm_sim3f* S, const float* p, int n, float* out);
void dvmh_sim3_inverse(const dvm_sim3f* S, dvm_sim3f* out);
float dvmh_logf(float x);                                                            /* the logf MapPoint::PredictScale shares with the device */

/* ---- ORBVocabulary (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1025-1144): transform of n features on a vocabulary given as
 * flat arrays (children of node k = children[child_off[k] .. child_off[k+1]), node descriptors 32 B, leaf weights / word ids) */
int dvmh_vocab_transform(int device, int n_nodes, const int32_t* child_off, const int32_t* children, const uint8_t* desc, const double* weight,
                         const int32_t* word_id, int L, const uint8_t* features, int n, int levelsup, int32_t* bow_ids, double* bow_vals, int* n_bow,
                         int32_t* fv_nodes, int32_t* fv_off, int32_t* fv_feat, int* n_fv);
/* bool TemplatedVocabulary::loadFromTextFile(filename) (TemplatedVocabulary.h:1211-1286; ORBvoc.txt's format) into a vocabulary that lives
 * on the device; info4 = {k, L, nodes, words}.  NULL: the file does not parse.  dvmh_vocab_transform_loaded = transform() on it. */
typedef struct dvmh_vocab dvmh_vocab;
dvmh_vocab* dvmh_vocab_load_text(int device, const char* filename, int32_t* info4);
void dvmh_vocab_destroy(dvmh_vocab* v);
int dvmh_vocab_transform_loaded(dvmh_vocab* voc, const uint8_t* features, int n, int levelsup, int32_t* bow_ids, double* bow_vals, int* n_bow,
                                int32_t* fv_nodes, int32_t* fv_off, int32_t* fv_feat, int* n_fv);
double dvmh_bow_score(const int32_t* ids1, const double* vals1, int n1, const int32_t* ids2, const double* vals2, int n2);   /* L1 score, ScoringObject.cpp:23-63 */

/* ---- KeyFrameDatabase (src/KeyFrameDatabase.cc:43-70, 555-808); keyframes are slots, uuid 0 is reserved */
typedef struct dvmh_kfdb dvmh_kfdb;
dvmh_kfdb* dvmh_kfdb_create(int device);
void dvmh_kfdb_destroy(dvmh_kfdb* db);
int dvmh_kfdb_add(dvmh_kfdb* db, const int32_t* ids, const double* vals, int n, int32_t map_id, uint64_t uuid, int64_t mnId);   /* returns the slot */
void dvmh_kfdb_erase(dvmh_kfdb* db, int slot);
void dvmh_kfdb_set_bad(dvmh_kfdb* db, int slot, int bad);
void dvmh_kfdb_set_map_bad(dvmh_kfdb* db, int32_t map_id, int bad);
void dvmh_kfdb_set_map(dvmh_kfdb* db, int slot, int32_t map_id);                              /* KeyFrame::UpdateMap: LoopClosing::MergeLocal moves keyframes to the merged map (LoopClosing.cc:1558,1767) */
void dvmh_kfdb_set_neighbours(dvmh_kfdb* db, int slot, const int32_t* neigh, int n);        /* GetBestCovisibilityKeyFrames(10) */
void dvmh_kfdb_set_connected(dvmh_kfdb* db, int slot, const int32_t* conn, int n);          /* GetConnectedKeyFrames() */
void dvmh_kfdb_get_state(dvmh_kfdb* db, int slot, uint64_t* query, int32_t* words, float* score);
int dvmh_kfdb_merge_score(dvmh_kfdb* db, const int32_t* qids, const double* qvals, int nq, uint64_t keyFrameId, int32_t map_id, float* score,
                          int32_t* bestKeyFrame);                                           /* CalculateMergeScore, :688-786 */
int dvmh_kfdb_detect_merge_possibility(dvmh_kfdb* db, const int32_t* qids, const double* qvals, int nq, uint64_t uuid, int32_t map_id,
                                       int32_t* bestKeyFrame, float* score, float* baseline); /* DetectMergePossibility, :789-808 */
int dvmh_kfdb_detect_n_best(dvmh_kfdb* db, int slot, int nNum, int32_t* loop, int32_t* n_loop, int32_t* merge, int32_t* n_merge);
/* vector<KeyFrame*> KeyFrameDatabase::DetectRelocalizationCandidates(Frame* F, Map* pMap) (KeyFrameDatabase.cc:810-909): the frame is
 * its BowVector and mnId; `out` needs room for one entry per stored keyframe.  dvmh_kfdb_get_reloc_state: mnRelocQuery /
 * mnRelocWords / mRelocScore of a slot (the state the query leaves on the keyframes). */
int dvmh_kfdb_detect_reloc(dvmh_kfdb* db, const int32_t* qids, const double* qvals, int nq, uint64_t frame_id, int32_t map_id,
                           int32_t* out, int32_t* n_out);
void dvmh_kfdb_get_reloc_state(dvmh_kfdb* db, int slot, uint64_t* query, int32_t* words, float* score);   /* :555-669 */

#ifdef __cplusplus
}
#endif
#endif
