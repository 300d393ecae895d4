/*
 * oracle/oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.  The product
 * (dvm_slam_amd/csrc -> libdvmslam_hip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference (proroklab/DVM-SLAM, ORB-SLAM3 fork) cannot be built in this
 * image (needs OpenCV>=4.2, Eigen3, Boost, ROS 2, Pangolin -- none present, no network) and holds
 * no test / golden vector for this path.  The pixel arithmetic lives in
 * un-vendored OpenCV (cv::resize, cv::GaussianBlur, cv::FAST, cv::fastAtan2, cvRound), restated
 * here from the published OpenCV 4.x algorithms; everything else follows
 * the reference file:line cited at each function.  The oracle is pinned only against independent
 * brute-force numpy restatements (tests/test_oracle_*.py) and the committed tests/golden vectors.
 * (oracle/_ref -- `make _ref` -- holds the one piece of the reference that compiles from its own
 * files, DUtils::Random; it pins the RANSAC draws of the Sim3Solver tests and nothing else.)
 */
#ifndef DVM_ORACLE_H
#define DVM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* layout-identical to cv::KeyPoint (7 x 4 B) */
typedef struct {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} orc_keypoint;

typedef struct {
  int32_t nfeatures;
  float scale_factor;
  int32_t nlevels, ini_th_fast, min_th_fast;
} orc_orb_params;

typedef struct orc_orb orc_orb; /* opaque */

/* ---- ORB extractor (reference ORBextractor.cc) ---- */
orc_orb* orc_orb_create(const orc_orb_params* p);
void orc_orb_destroy(orc_orb* h);
/* constructor tables, ORBextractor.cc:282-339.  Arrays sized nlevels (umax: 16). */
void orc_orb_tables(const orc_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int32_t* nfeat_per_level, int32_t* umax16);
/* operator(), ORBextractor.cc:876-955.  Returns number of keypoints written (<= cap), or -1 for an
 * empty image; *mono_index receives the reference's return value. */
int orc_orb_extract(orc_orb* h, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1,
                    orc_keypoint* kps, uint8_t* desc, int cap, int* mono_index);
/* intermediates of the last extract (for stage-wise parity tests) */
int orc_orb_level_dims(const orc_orb* h, int level, int* rows, int* cols);
/* copies the level image WITHOUT border, tightly packed cols bytes per row */
int orc_orb_get_level(const orc_orb* h, int level, uint8_t* out);
/* copies the level image WITH its 19-px REFLECT_101 border, (cols+38) bytes per row */
int orc_orb_get_level_bordered(const orc_orb* h, int level, uint8_t* out);
int orc_orb_get_blurred(const orc_orb* h, int level, uint8_t* out);
/* per-level FAST candidates (vToDistributeKeys, border-relative coords) in reference order */
int orc_orb_get_candidates(const orc_orb* h, int level, int32_t* xs, int32_t* ys, int32_t* scores, int cap);
/* per-level keypoints after octree + orientation, level coordinates (before scaling) */
int orc_orb_get_level_keypoints(const orc_orb* h, int level, orc_keypoint* kps, int cap);

/* ---- stand-alone primitives (OpenCV restatements) ---- */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                          int dstride);
void orc_gaussian_blur7_s2_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
void orc_gaussian_kernel7_s2_q8(int32_t k[7]);
/* cv::FAST(roi, kps, threshold, nonmaxSuppression=true), TYPE_9_16.  Returns count. */
int orc_fast9_16(const uint8_t* roi, int w, int h, int stride, int threshold, int32_t* xs, int32_t* ys,
                 int32_t* scores, int cap);
float orc_fast_atan2(float y, float x);
int orc_cv_round(float v);
/* (float)cos / (float)sin of (angle_deg * (float)(pi/180)) -- shared-spec implementation, DESIGN.md */
void orc_sincos_deg(float angle_deg, float* c, float* s);
float orc_ic_angle(const uint8_t* img, int stride, int cx, int cy);
void orc_brief_descriptor(const uint8_t* blurred, int stride, int cx, int cy, float angle_deg, uint8_t out[32]);
/* DistributeOctTree, ORBextractor.cc:419-610.  Returns number of selected keypoints. */
int orc_distribute_octree(const int32_t* xs, const int32_t* ys, const int32_t* scores, int n, int minX, int maxX,
                          int minY, int maxY, int N, int32_t* out_idx, int cap);

/* ---- matching (reference ORBmatcher.cc / Frame.cc) ---- */
/* ORBmatcher::DescriptorDistance, ORBmatcher.cc:1900-1914 */
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b);
void orc_hamming_matrix(const uint8_t* A, int nA, const uint8_t* B, int nB, uint16_t* D);

/* Frame grid: Frame.cc:443-444,481-506,773-782 (PosInGrid) and :712-770 (GetFeaturesInArea) */
typedef struct orc_grid orc_grid;
orc_grid* orc_grid_create(const orc_keypoint* kps, int n, float minX, float maxX, float minY, float maxY);
void orc_grid_destroy(orc_grid* g);
int orc_grid_features_in_area(const orc_grid* g, float x, float y, float r, int minLevel, int maxLevel,
                              int32_t* out, int cap);

/* Windowed best / second-best search: the inner loop shared by ORBmatcher::SearchByProjection
 * (ORBmatcher.cc:70-115 and :1604-1639).  For query q (descriptor qdesc[q], predicted position
 * (qx,qy), radius qr, octave window [qmin,qmax]) scan GetFeaturesInArea candidates in reference
 * order, skipping train indices with skip[idx]!=0, strict '<' updates.  Outputs best idx (-1 if
 * none), best dist (256), second-best dist (256), levels of both (-1). */
void orc_match_window(const orc_grid* g, const uint8_t* tdesc, const uint8_t* skip, const uint8_t* qdesc,
                      const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                      const int32_t* qmax, int nq, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist,
                      int32_t* best_level, int32_t* second_level);

/* Frame::isInFrustum (Frame.cc:575-636), mono */
typedef struct { float Rcw[9], tcw[3], Ow[3], fx, fy, cx, cy, min_x, max_x, min_y, max_y, bf, log_scale_factor; int32_t n_levels; } orc_frustum_frame;
typedef struct { float proj_x, proj_y, proj_xr, depth, view_cos; int32_t level; int32_t in_view; } orc_track_point;
void orc_is_in_frustum(const orc_frustum_frame* F, const float* P, const float* normal, const float* min_dist,
                       const float* max_dist, int n, float viewing_cos_limit, orc_track_point* out);

/* ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono=true), whole function (ORBmatcher.cc:1553-1748):
   mp_c / mp_l hold map-point indices (-1 = NULL); mp_c is updated in place; returns nmatches. */
typedef struct { float pos[3]; uint8_t desc[32]; int32_t n_obs; } orc_map_point;
int orc_search_by_projection_frames(int Nc, const orc_keypoint* kps_c, const uint8_t* desc_c, int32_t* mp_c, const float* Tcw7,
                                    const float* K, const float* bounds, const float* scale_factors,
                                    int Nl, const orc_keypoint* kps_l, const int32_t* mp_l, const uint8_t* outlier_l,
                                    const orc_map_point* mps, float th, int check_ori);

/* ORBmatcher::SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints), mono (ORBmatcher.cc:44-205) */
typedef struct { float proj_x, proj_y, depth, view_cos; int32_t level; uint8_t in_view, bad, pad_[2]; uint8_t desc[32]; int32_t n_obs; } orc_tracked_point;
int orc_search_by_projection_points(int N, const orc_keypoint* kps, const uint8_t* desc, int32_t* mp, const uint8_t* claimed_obs,
                                    const float* bounds, const float* scale_factors, const orc_tracked_point* pts, int npts,
                                    float th, float nnratio, int far_points, float th_far);

/* Whole matcher functions, SURVEY.md 8(a) M4-M7 (monocular).  Map points are indices (-1 = NULL), bad_* = isBad() of the
   map point at that keypoint (may be NULL); FeatureVectors are CSR (node ids ascending). See match_oracle.cpp. */
int orc_search_for_initialization(int N1, const orc_keypoint* kps1, const uint8_t* desc1, int N2, const orc_keypoint* kps2,
                                  const uint8_t* desc2, const float* bounds, float* prev_matched, int32_t* matches12,
                                  int windowSize, float nnratio, int check_ori);
int orc_search_by_bow_kf_frame(const orc_keypoint* kps_kf, const uint8_t* desc_kf, const int32_t* mp_kf, const uint8_t* bad_kf,
                               const int32_t* fv_nodes_kf, const int32_t* fv_off_kf, const int32_t* fv_feat_kf, int nn_kf, int N_f,
                               const orc_keypoint* kps_f, const uint8_t* desc_f, const int32_t* fv_nodes_f, const int32_t* fv_off_f,
                               const int32_t* fv_feat_f, int nn_f, float nnratio, int check_ori, int32_t* matches);
int orc_search_by_bow_kf_kf(int N1, const orc_keypoint* kps1, const uint8_t* desc1, const int32_t* mp1, const uint8_t* bad1,
                            const int32_t* fv_nodes1, const int32_t* fv_off1, const int32_t* fv_feat1, int nn1, int N2,
                            const orc_keypoint* kps2, const uint8_t* desc2, const int32_t* mp2, const uint8_t* bad2,
                            const int32_t* fv_nodes2, const int32_t* fv_off2, const int32_t* fv_feat2, int nn2, float nnratio,
                            int check_ori, int32_t* matches12);
void orc_triangulation_geometry(const float* T1w7, const float* T2w7, const float* K1, const float* K2, float* R12, float* t12,
                                float* ep, float* F12);
int orc_search_for_triangulation(int N1, const orc_keypoint* kps1, const uint8_t* desc1, const int32_t* mp1, const int32_t* fv_nodes1,
                                 const int32_t* fv_off1, const int32_t* fv_feat1, int nn1, int N2, const orc_keypoint* kps2,
                                 const uint8_t* desc2, const int32_t* mp2, const int32_t* fv_nodes2, const int32_t* fv_off2,
                                 const int32_t* fv_feat2, int nn2, const float* F12, const float* ep, const float* scale_factors2,
                                 const float* level_sigma2_2, int coarse, int check_ori, int32_t* pairs);
void orc_project_search(int N, const orc_keypoint* kps, const uint8_t* desc, const float* bounds, const uint8_t* skip, const float* Tcw7,
                        const float* Ow, const float* K, int n, const float* P, const float* normal,
                        const float* min_dist, const float* max_dist, const uint8_t* pdesc, const uint8_t* valid, float th,
                        const float* scale_factors, float log_scale_factor, int n_levels, const float* gate_inv_sigma2, double gate,
                        int32_t* best_idx, int32_t* best_dist, float* proj);
int orc_fuse_sim3(int N, const orc_keypoint* kps, const uint8_t* desc, const float* bounds, int32_t* kf_mp, const uint8_t* kf_mp_bad,
                  const float* Scw7, const float* K, int n, const int32_t* point_id,
                  const uint8_t* point_bad, const float* P, const float* normal, const float* min_dist, const float* max_dist,
                  const uint8_t* pdesc, float th, const float* scale_factors, float log_scale_factor, int n_levels, int32_t* replace);
int orc_search_by_projection_sim3(int N, const orc_keypoint* kps, const uint8_t* desc, const float* bounds, int32_t* matched,
                                  const float* Scw7, const float* K, int n, const int32_t* point_id,
                                  const uint8_t* point_bad, const float* P, const float* normal, const float* min_dist,
                                  const float* max_dist, const uint8_t* pdesc, int th, float ratioHamming, const float* scale_factors,
                                  float log_scale_factor, int n_levels);

int orc_search_by_sim3(int N1, const orc_keypoint* kps1, const uint8_t* desc1, const int32_t* mp1, const uint8_t* bad1, const float* P1,
                       const float* min1, const float* max1, const uint8_t* mdesc1, const float* T1w7, int N2,
                       const orc_keypoint* kps2, const uint8_t* desc2, const int32_t* mp2, const uint8_t* bad2, const float* P2,
                       const float* min2, const float* max2, const uint8_t* mdesc2, const float* T2w7,
                       const float* bounds, const float* K, const float* S12_7, float th,
                       const float* scale_factors, float log_scale_factor, int n_levels, int32_t* matches12, const int32_t* idx_in_kf2);

int orc_search_by_projection_reloc(int Nc, const orc_keypoint* kps_c, const uint8_t* desc_c, int32_t* mp_c, const float* bounds,
                                   const float* Tcw7, const float* K, int Nk, const orc_keypoint* kps_k,
                                   const int32_t* mp_k, const uint8_t* bad_k, const float* P, const float* min_dist, const float* max_dist,
                                   const uint8_t* pdesc, const int32_t* already, int n_already, float th, int ORBdist,
                                   const float* scale_factors, float log_scale_factor, int n_levels, int check_ori);

/* Pose arithmetic of the reference (oracle/sophus_oracle.h: Sophus so3/se3/rxso3/sim3.hpp + Eigen 3.4.0 restated in f32).
 * A pose is 7 floats: quaternion coeffs (x, y, z, w) as Sophus stores them (unit for SE3f, |q|^2 = scale for Sim3f), then
 * translation.  orc_pose_matrices = Frame::UpdatePoseMatrices (Frame.cc:553-559). */
void orc_se3_inverse(const float* T7, float* out7);
void orc_pose_matrices(const float* Tcw7, float* Rcw, float* tcw, float* Ow);
void orc_sim3_to_se3(const float* S7, float* Tcw7, float* Ow);
void orc_sim3_inverse(const float* S7, float* out7);
void orc_se3_act(const float* T7, const float* p, int n, float* out);
void orc_sim3_act(const float* S7, const float* p, int n, float* out);
float orc_logf(float x);
/* Frame::UndistortKeyPoints / ComputeImageBounds (Frame.cc:791-848; cv::undistortPoints restated, see match_oracle.cpp).
   cam = {fx, fy, cx, cy, k1, k2, p1, p2, k3} */
void orc_undistort_points(const float* cam, const float* xy_in, int n, float* xy_out);
void orc_image_bounds(const float* cam, int cols, int rows, float* out4);

/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:384-453), batched over map points (CSR offsets into desc) */
void orc_distinctive_descriptors(const uint8_t* desc, const int32_t* off, int npts, int32_t* best_idx, int32_t* best_median);

/* DBoW2 TemplatedVocabulary::transform (TF_IDF, L1_NORM) on a CSR vocabulary tree + L1Scoring::score */
void orc_vocab_transform(int n_nodes, const int32_t* child_off, const int32_t* children, const uint8_t* node_desc,
                         const double* weight, const int32_t* word_id, int L, const uint8_t* features, int n, int levelsup,
                         int32_t* out_word, int32_t* out_node, double* out_weight, int32_t* bow_ids, double* bow_vals,
                         int32_t* n_bow, int32_t* fv_nodes, int32_t* fv_off, int32_t* fv_feat, int32_t* n_fv);
double orc_bow_score(const int32_t* ids1, const double* vals1, int n1, const int32_t* ids2, const double* vals2, int n2);

/* ---- KeyFrameDatabase place recognition (KeyFrameDatabase.cc:43-70,555-808; see kfdb_oracle.cpp).  Keyframes are slots. ---- */
typedef struct orc_kfdb orc_kfdb;
orc_kfdb* orc_kfdb_create(void);
void orc_kfdb_destroy(orc_kfdb* db);
int orc_kfdb_add(orc_kfdb* db, const int32_t* ids, const double* vals, int n, int32_t map_id, uint64_t uuid, int64_t mnId);
void orc_kfdb_erase(orc_kfdb* db, int slot);
void orc_kfdb_set_bad(orc_kfdb* db, int slot, int bad);
void orc_kfdb_set_map_bad(orc_kfdb* db, int32_t map_id, int bad);
void orc_kfdb_set_map(orc_kfdb* db, int slot, int32_t map_id);
void orc_kfdb_set_neighbours(orc_kfdb* db, int slot, const int32_t* neigh, int n);
void orc_kfdb_set_connected(orc_kfdb* db, int slot, const int32_t* conn, int n);
void orc_kfdb_get_state(const orc_kfdb* db, int slot, uint64_t* query, int32_t* words, float* score);
void orc_kfdb_merge_score(orc_kfdb* db, const int32_t* qids, const double* qvals, int nq, uint64_t keyFrameId, int32_t map_id,
                          float* score, int32_t* bestKeyFrame);
int orc_kfdb_detect_merge_possibility(orc_kfdb* db, const int32_t* qids, const double* qvals, int nq, uint64_t uuid, int32_t map_id,
                                      int32_t* bestKeyFrame, float* score_out, float* baseline_out);
void orc_kfdb_detect_n_best(orc_kfdb* db, int slot, int nNumCandidates, int32_t* loop, int32_t* n_loop, int32_t* merge, int32_t* n_merge);
/* DetectRelocalizationCandidates(Frame*, Map*) (KeyFrameDatabase.cc:810-909): out holds at most one entry per stored keyframe */
void orc_kfdb_detect_reloc(orc_kfdb* db, const int32_t* qids, const double* qvals, int nq, uint64_t frame_id, int32_t map_id,
                           int32_t* out, int32_t* n_out);
void orc_kfdb_get_reloc_state(const orc_kfdb* db, int slot, uint64_t* query, int32_t* words, float* score);

/* ---- bundle adjustment (reference Optimizer.cc + vendored g2o; see ba_oracle.cpp) ---- */
typedef struct { int32_t pose, point; double u, v, inv_sigma2; } orc_ba_edge;
typedef struct { double fx, fy, cx, cy, huber_delta; /* <= 0: no robust kernel */ } orc_ba_camera;
typedef struct {
  int32_t iterations, total_trials, stop_reason, pad;  /* stop: 0 ran out of iterations, 1 LM terminate, 2 Mur-Artal criterion */
  double chi2_initial, chi2_final, lambda_final;
  int32_t trials_per_iter[64];
  double chi2_per_iter[64], lambda_per_iter[64];
} orc_ba_stats;
/* poses [P][7] = (tx,ty,tz,qx,qy,qz,qw) world->camera, in/out; points [L][3] in/out.  Runs
 * optimizer.optimize(iterations) of the reference's BA graph.  edge_chi2 (may be NULL) receives each
 * edge's chi2() as g2o would report it after optimize().  Returns iterations performed. */
int orc_ba_optimize(double* poses, const uint8_t* fixed, int P, double* points, int L, const orc_ba_edge* edges, int E,
                    const orc_ba_camera* cam, int iterations, orc_ba_stats* st, double* edge_chi2);
/* a further optimize(n) on the same graph: estimates taken as they are (no normalisation of the input quaternions) */
int orc_ba_optimize_continue(double* poses, const uint8_t* fixed, int P, double* points, int L, const orc_ba_edge* edges, int E,
                             const orc_ba_camera* cam, int iterations, orc_ba_stats* st, double* edge_chi2);
/* oracle/f64_spec.h evaluated on n arguments: out = [sin | cos | cube] */
void orc_f64_spec(const double* x, int n, double* out);
void orc_ba_edge_chi2(const double* poses, const double* points, const orc_ba_edge* edges, int E,
                      const orc_ba_camera* cam, double* chi2, uint8_t* depth_positive);
int orc_pose_optimize(double* pose, const double* Xw, const double* obs, const double* inv_sigma2, int N,
                      const orc_ba_camera* cam, uint8_t* outlier);

/* Optimizer::OptimizeSim3 numerics (Optimizer.cc:1960-2212); S12 = (qx,qy,qz,qw,tx,ty,tz,s); K = (fx,fy,cx,cy) */
int orc_optimize_sim3(double* S12io, int fix_scale, const double* P1c, const double* P2c, const double* obs1,
                      const double* obs2, const double* w1, const double* w2, int N, const double* K1, const double* K2,
                      double th2, uint8_t* inlier);

/* Optimizer::OptimizeEssentialGraph numerics (Optimizer.cc:1389-1652): Sim3 pose graph, EdgeSim3 with g2o's numeric
   Jacobians, LM with lambda_init 1e-16.  S[n][8] = (q_xyzw, t, s) Siw in/out; ev[E][2] = (vertex 0 = i, vertex 1 = j);
   emeas[E][8] = Sji; stats[70] = iterations, trials, chi2_initial, chi2_final, lambda_final, stop_reason, chi2_per_iter[32], trials_per_iter[32] */
void orc_sim3_exp_log(const double* u, double* S8, double* log7);
int orc_pose_graph_optimize(double* S, const uint8_t* fixed, int n, const int32_t* ev, const double* emeas, int E, int fix_scale,
                            int iterations, double* stats);

/* Sim3Solver::ComputeSim3 + CheckInliers (src/Sim3Solver.cc:294-408) for H given minimal sets; T12 = [s, R row-major (9), t (3)] */
void orc_sim3_hypotheses(const float* P1c, const float* P2c, const float* max_err1, const float* max_err2, int N, const float* K1,
                         const float* K2, const int32_t* triples, int H, int fix_scale, float* T12, int32_t* n_inliers,
                         uint8_t* inlier_mask);

/* LocalMapping::CreateNewMapPoints, per-match geometry for one neighbour keyframe (LocalMapping.cc:598-741 mono pinhole branch,
 * GeometricTools.cc:48-67): parallax, triangulation, depth / reprojection / scale-consistency tests.  T*w: 3x4 row-major
 * (GetPose().matrix3x4()); pairs [n][2] = (index in keyframe 1, index in keyframe 2); status 0 = accepted (see ba_oracle.cpp). */
void orc_triangulate_matches(const float* K1, const float* K2, const float* T1w, const float* T2w, const float* Ow1, const float* Ow2,
                             const orc_keypoint* kps1, const orc_keypoint* kps2, const int32_t* pairs, int n, const float* sigma2_1,
                             const float* sigma2_2, const float* sf1, const float* sf2, float ratio_factor, double cos_parallax_max,
                             int far_points, float th_far, float* x3D_out, int32_t* status);

#ifdef __cplusplus
}
#endif
#endif
