/*
 * include/dvmslam_hip.h -- C ABI of libdvmslam_hip.so: the MI355X (gfx950) implementation of the
 * per-agent visual-SLAM hot path of proroklab/DVM-SLAM (ORB front end, Hamming matching, bundle
 * adjustment).  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Reference interface replaced (all under /root/reference/src/slam_system/orb_slam3/):
 *   dvm_orb_*      <- class ORBextractor            include/ORBextractor.h:47-91, src/ORBextractor.cc:282-976
 *   dvm_frame_*    <- Frame grid members            include/Frame.h:44-45,221-251, src/Frame.cc:443-506,712-782
 *   dvm_hamming_*  <- ORBmatcher::DescriptorDistance include/ORBmatcher.h:44, src/ORBmatcher.cc:1900-1914
 *   dvm_match_*    <- inner loops of ORBmatcher::SearchByProjection / SearchForInitialization
 *                                                    src/ORBmatcher.cc:44-205,605-707,1553-1748
 *   dvm_ba_*       <- Optimizer::{BundleAdjustment,LocalBundleAdjustment} + g2o BlockSolver_6_3/LM
 *                                                    src/Optimizer.cc:55-356,1030-1387
 *   dvm_pose_optimize <- Optimizer::PoseOptimization src/Optimizer.cc:744-1028
 *   dvm_pose_graph_optimize <- Optimizer::OptimizeEssentialGraph src/Optimizer.cc:1389-1652 (g2o part)
 *   dvm_sim3_hypotheses <- Sim3Solver::ComputeSim3 + CheckInliers src/Sim3Solver.cc:294-408
 *   dvm_distinctive_descriptors <- MapPoint::ComputeDistinctiveDescriptors src/MapPoint.cc:384-453
 *   dvm_vocab_transform <- DBoW2::TemplatedVocabulary::transform Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1098-1138
 * The reference has no FFI: these classes live inside static libORB_SLAM3.a.  INTEGRATION.md shows
 * the C++ shim classes (same names / signatures) a maintainer links instead.
 *
 * Conventions: every function returns DVM_OK (0) or a negative dvm_status; nothing throws across
 * the boundary.  A handle owns one HIP stream and all of its device memory; handles are not
 * thread-safe (one per calling thread, like the reference's ORBextractor instance per Tracking).
 * Pointers named d_* are DEVICE pointers (HBM), all others are host pointers.
 */
#ifndef DVMSLAM_HIP_H
#define DVMSLAM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  DVM_OK = 0,
  DVM_ERR_INVALID = -1,    /* bad argument */
  DVM_ERR_EMPTY = -2,      /* empty image: the reference's operator() returns -1 (ORBextractor.cc:879-880) */
  DVM_ERR_CAPACITY = -3,   /* caller buffer too small */
  DVM_ERR_HIP = -4,        /* HIP runtime error (see dvm_last_error) */
  DVM_ERR_NO_DEVICE = -5,  /* no gfx950 device visible: the library never falls back to a CPU path */
  DVM_ERR_STATE = -6       /* call sequence error */
} dvm_status;

const char* dvm_last_error(void);
const char* dvm_version(void);
/* number of visible HIP devices (0 when none); never fails */
int dvm_device_count(void);
/* makes `device` the calling THREAD's current HIP device.  Entry points that take a handle or a device argument select their GPU
 * themselves; the stateless ones without either (dvm_hamming_matrix, dvm_is_in_frustum, dvm_triangulate_matches,
 * dvm_undistort_keypoints, dvm_match_lists, dvm_match_triangulation, dvm_distinctive_descriptors, dvm_pose_optimize's host form ...)
 * run on the calling thread's current device -- 0 on a fresh thread.  An agent pinned to GPU k calls this once per thread (the shims
 * of dvm_slam_amd/host do it through dvm_host::use_device()).  DVM_ERR_NO_DEVICE / DVM_ERR_INVALID as for the handle constructors. */
int dvm_set_device(int device);

/* layout-identical to cv::KeyPoint (7 x 4 B): pt.x, pt.y, size, angle, response, octave, class_id */
typedef struct {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} dvm_keypoint;

/* ------------------------------------------------------------------------------ ORB extractor */
/* ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) */
typedef struct {
  int32_t nfeatures;
  float scale_factor;
  int32_t nlevels, ini_th_fast, min_th_fast;
} dvm_orb_params;

typedef struct dvm_orb dvm_orb;

/* max_batch: frames processed per launch set (>=1).  Device buffers are sized lazily for the first
 * image size seen and re-sized when it changes. */
int dvm_orb_create(const dvm_orb_params* p, int device, int max_batch, dvm_orb** out);
void dvm_orb_destroy(dvm_orb* h);
/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares and
 * mnFeaturesPerLevel; arrays of nlevels entries, any may be NULL */
int dvm_orb_tables(const dvm_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                   int32_t* nfeat_per_level);

/* operator()(image, mask, keypoints, descriptors, vLappingArea): one host image in, host results
 * out (synchronous).  *n = number of keypoints, *mono_index = the reference's return value.
 * desc receives n x 32 bytes.  Returns DVM_ERR_EMPTY for a null / zero-sized image. */
int dvm_orb_extract(dvm_orb* h, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1,
                    dvm_keypoint* kps, uint8_t* desc, int cap, int* n, int* mono_index);

/* A SHARED extractor for several agents' frames.  The reference runs one ORBextractor per agent, each called once per frame from that
 * agent's tracking thread (Tracking.cc:1423-1426; orb_slam3_wrapper.cpp runs one System per agent).  K such threads on one GPU issue K
 * chains of small launches that serialise in the runtime; dvm_orb_pool_extract is the same blocking call -- one image in, that
 * frame's keypoints and descriptors out, the same bytes as dvm_orb_extract -- but frames that arrive within `window_us` of each other
 * (and share size and lapping area) are extracted as ONE batch of up to max_batch frames: the caller that opened the batch waits for
 * the arrivals to pause, runs it, and every caller copies its own frame's results out.  Four batches are in flight (one collecting while
 * others run; max_batch 8 measured best: small overlapping batches beat large ones that keep the agents in lockstep).  Callable from any number of threads at once; window_us < 0: 20 us; a caller that has been alone for eight calls stops waiting.
 * batch_size (may be NULL): how many frames the call's batch held.  Destroy a pool only when no call on it is in flight (this holds for
 * dvm_pose_pool and dvm_match_pool too). */
typedef struct dvm_orb_pool dvm_orb_pool;
int dvm_orb_pool_create(const dvm_orb_params* p, int device, int max_batch, int window_us, dvm_orb_pool** out);
void dvm_orb_pool_destroy(dvm_orb_pool* pool);
int dvm_orb_pool_extract(dvm_orb_pool* pool, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1, dvm_keypoint* kps,
                         uint8_t* desc, int cap, int* n, int* mono_index, int* batch_size);

/* Batched, device-resident form: `batch` frames of rows x cols at d_imgs + f*frame_stride (bytes),
 * row pitch `stride`.  Asynchronous on the handle's stream; results stay in HBM until fetched. */
int dvm_orb_extract_batch_device(dvm_orb* h, const uint8_t* d_imgs, int batch, int rows, int cols, int stride,
                                 int64_t frame_stride, int lap0, int lap1);
/* Same, from host memory (one pinned staging copy + H2D on the handle's stream). */
int dvm_orb_extract_batch_host(dvm_orb* h, const uint8_t* imgs, int batch, int rows, int cols, int stride,
                               int64_t frame_stride, int lap0, int lap1);
/* blocks until the handle's stream is idle */
/* Pinned-staging ingest: dvm_orb_staging returns the handle's page-locked input buffer (batch x rows x cols bytes, rows
 * tight) for the caller -- camera driver, decoder -- to write frames into; dvm_orb_extract_staged queues the PCIe copy and
 * the extraction on the handle's stream and returns without waiting.  dvm_orb_staging waits for the previous copy out of
 * the buffer.  Two handles used alternately overlap one batch's transfer with the other's kernels. */
int dvm_orb_staging(dvm_orb* h, int batch, int rows, int cols, uint8_t** host_ptr);
int dvm_orb_extract_staged(dvm_orb* h, int batch, int rows, int cols, int lap0, int lap1);
int dvm_orb_sync(dvm_orb* h);
/* device views of frame f's results (valid until the next extract on this handle).  kps are in
 * the reference's output order; d_n points at the int32 keypoint count of the frame. */
int dvm_orb_result_device(dvm_orb* h, int frame, const dvm_keypoint** d_kps, const uint8_t** d_desc,
                          const int32_t** d_n, int* capacity);
/* A reference to the keypoints + descriptors a single-frame dvm_orb_extract has just produced, still in HBM: a plain value the
 * caller keeps beside the host copies (the Frame that owns mvKeys / mDescriptors -- Frame.cc:411 -- and every copy of that Frame) and
 * hands to the consumers of the same data (dvm_frame_build through dvmh_frame_view::dev: the grid of SearchByProjection is then
 * built from the device arrays, no second upload).  The arrays live until the handle extracts again or is destroyed;
 * dvm_device_frame_valid tells (0 / 1) whether a reference still names the handle's current result. */
typedef struct {
  const dvm_keypoint* d_kps;
  const uint8_t* d_desc;
  int32_t n, device;
  uint64_t handle_id, serial;
} dvm_device_frame;
int dvm_orb_last_result(dvm_orb* h, dvm_device_frame* out);
int dvm_device_frame_valid(const dvm_device_frame* ref);
/* asynchronous device-to-device copy (on the handle's stream) of frame f's keypoints, descriptors
 * and count into caller-owned device buffers (capacity as reported by dvm_orb_result_device) --
 * used to carry the last frame of a batch over to the next batch's frame-to-frame search */
int dvm_orb_copy_result(dvm_orb* h, int frame, dvm_keypoint* d_kps_dst, uint8_t* d_desc_dst, int32_t* d_n_dst);
/* device array of mvScaleFactor (nlevels floats) */
const float* dvm_orb_scale_factors_device(dvm_orb* h);
/* synchronises, then copies frame f's results to the host */
int dvm_orb_download(dvm_orb* h, int frame, dvm_keypoint* kps, uint8_t* desc, int cap, int* n, int* mono_index);
/* the first `count` frames of the last batch in one go (one synchronisation, two block copies through page-locked memory): kps[f] / desc[f]
 * with caps[f] entries, n[f], mono_index[f] */
int dvm_orb_download_batch(dvm_orb* h, int count, dvm_keypoint* const* kps, uint8_t* const* desc, const int* caps, int* n, int* mono_index);
/* mvImagePyramid[level] of frame f: device pointer to pixel (0,0) of the level; the 19-px
 * REFLECT_101 border is addressable at negative offsets exactly like the reference's ROI Mats */
int dvm_orb_pyramid(dvm_orb* h, int frame, int level, const uint8_t** d_ptr, int* rows, int* cols, int* stride);

/* stage-wise intermediates of frame f (parity tests; each synchronises) */
int dvm_orb_debug_level(dvm_orb* h, int frame, int level, int bordered, uint8_t* out /* tight rows */);
int dvm_orb_debug_blurred(dvm_orb* h, int frame, int level, uint8_t* out);
int dvm_orb_debug_candidates(dvm_orb* h, int frame, int level, int32_t* xs, int32_t* ys, int32_t* scores,
                             int cap, int* n);
int dvm_orb_debug_level_keypoints(dvm_orb* h, int frame, int level, dvm_keypoint* kps, int cap, int* n);

/* per-kernel HIP-event timing on the handle's stream.  names: "pyramid","fast","octree",
 * "assemble","blur","orient_desc","grid_sort","match" ...  Returns accumulated milliseconds and launch count since
 * the last reset. */
int dvm_orb_profiling(dvm_orb* h, int enable);
int dvm_orb_profile_get(dvm_orb* h, const char* name, double* total_ms, int64_t* launches);
int dvm_orb_profile_reset(dvm_orb* h);
/* the handle's hipStream_t (so callers can order their own work / events against it) */
void* dvm_orb_stream(dvm_orb* h);

/* ----------------------------------------------------------------------------------- matching */
/* DescriptorDistance for all pairs: D[i*nB+j] = popcount(A[i]^B[j]) over 256 bits (uint16).
 * on_device != 0: A, B, D are device pointers and the call is asynchronous on `stream`
 * (hipStream_t, may be NULL for the default stream); otherwise host pointers, synchronous. */
int dvm_hamming_matrix(const uint8_t* A, int nA, const uint8_t* B, int nB, uint16_t* D, int on_device, void* stream);

/* Frame feature grid (FRAME_GRID_COLS=64 x FRAME_GRID_ROWS=48) + windowed best/second-best search.
 * A dvm_frame holds `slots` frames' undistorted keypoints + descriptors in HBM, each ordered the
 * way Frame::GetFeaturesInArea enumerates them, so ties resolve exactly like the reference's loops.
 * capacity <= 8192 keypoints per slot. */
typedef struct dvm_frame dvm_frame;
int dvm_frame_create(int device, int capacity, int slots, dvm_frame** out);
void dvm_frame_destroy(dvm_frame* f);
/* (Re)build slot `slot` from n keypoints + descriptors; on_device selects pointer kind; d_n (device
 * int32*, may be NULL, only with on_device) overrides n with a device-side count.  Bounds are
 * Frame::mnMinX/mnMaxX/mnMinY/mnMaxY (0,cols,0,rows when undistorted) and apply to all slots.
 * Asynchronous on `stream` when on_device (host pointers: synchronous). */
int dvm_frame_build(dvm_frame* f, int slot, const dvm_keypoint* kps, const uint8_t* desc, int n, const int32_t* d_n,
                    float minX, float maxX, float minY, float maxY, int on_device, void* stream);
/* Build slots [first_slot, first_slot+count) from device arrays d_kps + i*kps_stride (elements),
 * d_desc + i*desc_stride (bytes), counts d_n[i] -- the layout dvm_orb_result_device exposes. */
int dvm_frame_build_batch(dvm_frame* f, int first_slot, int count, const dvm_keypoint* d_kps, int64_t kps_stride,
                          const uint8_t* d_desc, int64_t desc_stride, const int32_t* d_n, float minX, float maxX,
                          float minY, float maxY, void* stream);
/* Device-side counts (d_n of dvm_frame_build[_batch], the per-frame counts of dvm_match_frames_batch) larger than the handle's
 * capacity are clamped by the kernels; every such event is counted.  Call after synchronising the stream(s) the launches
 * went to: *count = events so far (sticky), returns DVM_ERR_CAPACITY if non-zero. */
int dvm_frame_overflows(dvm_frame* f, int32_t* count);

typedef struct {
  int32_t best_idx;     /* index into the slot's ORIGINAL keypoint order, -1 if no candidate */
  int32_t best_dist;    /* 256 if none */
  int32_t second_dist;  /* 256 if none */
  int16_t best_level, second_level; /* octaves of best / second best, -1 if none */
} dvm_match;

/* For each query q: scan Frame::GetFeaturesInArea(qx,qy,qr,qmin,qmax) of train slot `slot`
 * (skipping indices with skip[idx]!=0, skip may be NULL) and return best / second best by
 * DescriptorDistance with the reference's strict-'<' first-wins tie rule.  Query arrays: qdesc
 * nq x 32 B, qx/qy/qr float, qmin/qmax int32 (-1 = unbounded, as in the reference).  d_nq (device
 * int32*, may be NULL) overrides nq.  All pointers are device pointers when on_device != 0
 * (asynchronous on `stream`), else host pointers (synchronous). */
int dvm_match_window(const dvm_frame* train, int slot, const uint8_t* skip, const uint8_t* qdesc, const float* qx,
                     const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax, int nq,
                     const int32_t* d_nq, dvm_match* out, int on_device, void* stream);
/* The same search, also returning the runner-up's train index (second_idx[q], -1 if none; may be NULL): what the reference's
 * scan returns when the best candidate is excluded -- a caller replaying ORBmatcher.cc:1613-1664's claims in query order takes
 * it when the best candidate was claimed by an earlier query of the same call, instead of searching again. */
int dvm_match_window_top2(const dvm_frame* train, int slot, const uint8_t* skip, const uint8_t* qdesc, const float* qx,
                          const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax, int nq,
                          const int32_t* d_nq, dvm_match* out, int32_t* second_idx, int on_device, void* stream);
/* The same scan returning the FOUR best candidates of every query in the reference's order of preference (distance, then scan
 * position): ranked[4 q + c] = dist << 16 | train index, dist = 256 from the end of the list on (fewer than four candidates: the list
 * is complete).  For a caller replaying ORBmatcher.cc:1613-1664's claims in query order: it walks the list to the first keypoint no
 * earlier query of the call has taken, and searches again itself only when all four are gone. */
int dvm_match_window_ranked(const dvm_frame* train, int slot, const uint8_t* skip, const uint8_t* qdesc, const float* qx, const float* qy,
                            const float* qr, const int32_t* qmin, const int32_t* qmax, int nq, uint32_t* ranked, int on_device, void* stream);
/* Host-pointer convenience, latency path of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame) (ORBmatcher.cc:1553-1748):
 * dvm_frame_build(f, slot, kps, desc, n, ..., on_device = 0) followed by dvm_match_window_ranked(f, slot, ..., on_device = 0) as ONE
 * staged call -- one upload, the two kernels back to back on the calling thread's stream, one synchronisation.  Same results as
 * the two calls.  kps_on_device != 0: kps / desc are DEVICE arrays (dvm_orb_last_result: the frame as the extractor left it in HBM);
 * everything else stays a host pointer. */
int dvm_frame_build_match_window_ranked(dvm_frame* f, int slot, const dvm_keypoint* kps, const uint8_t* desc, int n, float minX,
                                        float maxX, float minY, float maxY, const uint8_t* skip, const uint8_t* qdesc,
                                        const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                                        const int32_t* qmax, int nq, uint32_t* ranked, int kps_on_device);
/* The shared form of the call above for several agents on one GPU (as dvm_orb_pool for the extractor): the same arguments (host pointers), the
 * same ranked lists, callable from any number of threads; calls that arrive within `window_us` of each other and share the image bounds run
 * as ONE launch pair over the batch's grid slots.  kp_cap / q_cap: keypoints / queries per frame the pool holds (a larger frame is refused
 * with DVM_ERR_CAPACITY: the caller takes dvm_frame_build_match_window_ranked).  batch_size (may be NULL): frames in the call's launch. */
typedef struct dvm_match_pool dvm_match_pool;
int dvm_match_pool_create(int device, int max_batch, int kp_cap, int q_cap, int window_us, dvm_match_pool** out);
void dvm_match_pool_destroy(dvm_match_pool* pool);
int dvm_match_pool_capacity(const dvm_match_pool* pool, int* kp_cap, int* q_cap);
int dvm_match_pool_build_match_ranked(dvm_match_pool* pool, const dvm_keypoint* kps, const uint8_t* desc, int n, float minX, float maxX, float minY,
                                      float maxY, const uint8_t* skip, const uint8_t* qdesc, const float* qx, const float* qy, const float* qr,
                                      const int32_t* qmin, const int32_t* qmax, int nq, uint32_t* ranked, int* batch_size);
/* The stream the host-pointer convenience calls of the CALLING THREAD run on (device `device`; created on first use).  A caller
 * that builds a grid with on_device = 1 and then searches it through a host-pointer call passes it as `stream`, so that the two are
 * one in-order chain (no synchronisation in between). */
void* dvm_thread_stream(int device);

/* Frame::UndistortKeyPoints (Frame.cc:791-818) and Frame::ComputeImageBounds (:820-848): cv::undistortPoints(pts, K, D,
 * noArray(), K) -- OpenCV's 5-iteration fixed-point inversion of the (k1, k2, p1, p2, k3) model, in double from the float
 * inputs.  kps_out[i] = kps_in[i] with the point replaced (in place allowed); k1 == 0 copies, as the reference's early return
 * does.  dvm_image_bounds: {mnMinX, mnMaxX, mnMinY, mnMaxY} from the four undistorted image corners ({0, cols, 0, rows} for
 * k1 == 0); the values dvm_frame_build and the matcher gates take.  Host pointers (synchronous) or device pointers. */
typedef struct { float fx, fy, cx, cy, k1, k2, p1, p2, k3; } dvm_distortion;
int dvm_undistort_keypoints(const dvm_distortion* cam, const dvm_keypoint* kps_in, dvm_keypoint* kps_out, int n, int on_device, void* stream);
int dvm_image_bounds(const dvm_distortion* cam, int cols, int rows, float bounds[4]);

/* Frame::isInFrustum (Frame.cc:575-636, mono branch) for n map points at once: projection with the frame's
 * Rcw/tcw (float; mRcw = mTcw.rotationMatrix(), Frame.cc:553-559 -- this function does use the matrix form, :585),
 * image bounds, distance inside [0.8*mfMinDistance, 1.2*mfMaxDistance], viewing cosine,
 * MapPoint::PredictScale.  Outputs the mbTrackInView / mTrackProj* / mnTrackScaleLevel / mTrackViewCos fields the
 * reference stores on the MapPoint.  Host pointers (synchronous) or device pointers (asynchronous on `stream`). */
typedef struct { float Rcw[9], tcw[3], Ow[3], fx, fy, cx, cy, min_x, max_x, min_y, max_y, bf, log_scale_factor; int32_t n_levels; } dvm_frustum_frame;
typedef struct { float proj_x, proj_y, proj_xr, depth, view_cos; int32_t level; int32_t in_view; } dvm_track_point;
int dvm_is_in_frustum(const dvm_frustum_frame* frame, const float* P, const float* normal, const float* min_dist,
                      const float* max_dist, int n, float viewing_cos_limit, dvm_track_point* out, int on_device, void* stream);

/* LocalMapping::CreateNewMapPoints, the geometry of ONE neighbour keyframe (LocalMapping.cc:598-741, monocular pinhole branch) for
 * the index pairs ORBmatcher::SearchForTriangulation returned: unprojectEig, parallax of the two rays (cos > 0 and, in double,
 * < cos_parallax_max: 0.9998, 0.9996 inertial), GeometricTools::Triangulate (GeometricTools.cc:48-67), z > 0 in both cameras,
 * reprojection error <= 5.991 * mvLevelSigma2[octave] in both, zero / far distance, the distance-ratio vs octave-ratio test with
 * ratio_factor = 1.5f * mfScaleFactor.  pairs[2 m], pairs[2 m + 1] = index into kps1 (the current keyframe's mvKeysUn) / kps2.
 * x3D[3 m ..] = the triangulated point (zeros when no triangulation was attempted), status[m]: 0 accepted -- the caller creates the
 * MapPoint --, 1 parallax, 2 homogeneous w == 0, 3 z1 <= 0, 4 z2 <= 0, 5 / 6 reprojection error in keyframe 1 / 2, 7 zero distance,
 * 8 far point, 9 scale consistency, -1 index or octave out of range.  float arithmetic in Eigen's evaluation order; the null
 * vector of the 4x4 system comes from a double Jacobi diagonalisation of A^T A instead of Eigen::JacobiSVD<Matrix4f> (tolerance
 * parity on x3D).  Host pointers (synchronous) or device pointers (asynchronous on `stream`). */
typedef struct {
  double cos_parallax_max;
  float K1[4], K2[4];       /* fx, fy, cx, cy */
  float T1w[12], T2w[12];   /* KeyFrame::GetPose().matrix3x4(), row-major */
  float Ow1[3], Ow2[3];     /* KeyFrame::GetCameraCenter() */
  float ratio_factor, th_far;
  int32_t far_points;       /* mbFarPoints */
  int32_t n_levels;         /* length of the sigma2 / scale-factor tables */
} dvm_tri_pair;
int dvm_triangulate_matches(const dvm_tri_pair* pair, const dvm_keypoint* kps1, int n1, const dvm_keypoint* kps2, int n2,
                            const int32_t* pairs, int n, const float* sigma2_1, const float* sigma2_2, const float* scale_factors_1,
                            const float* scale_factors_2, float* x3D, int32_t* status, int on_device, void* stream);

/* Best / second best over an explicit candidate list per query -- the inner loop of the
 * vocabulary-node restricted searches (SearchByBoW ORBmatcher.cc:262-300,760-800; SearchForTriangulation
 * :905-960; SearchBySim3; Fuse): query q scans train descriptors cand[off[q] .. off[q+1]) in that order
 * (entries < 0 are skipped), strict-'<' first-wins.  best_idx is the train index, levels are -1.
 * Host pointers (synchronous) or device pointers (asynchronous on `stream`). */
int dvm_match_lists(const uint8_t* tdesc, int nt, const uint8_t* qdesc, int nq, const int32_t* off, const int32_t* cand,
                    dvm_match* out, int on_device, void* stream);

/* Projection of n map points into a keyframe + windowed best-descriptor search -- the common body of
 * ORBmatcher::Fuse(KF, vpMapPoints, th) (ORBmatcher.cc:1060-1234, gate_inv_sigma2 = KF.mvInvLevelSigma2, gate = 5.99),
 * Fuse(KF, Scw, ...) (:1236-1345), SearchByProjection(KF, Scw, vpPoints, vpMatched, th, ratioHamming) x2 (:395-603),
 * both directions of SearchBySim3 (:1347-1551; cam->sim3_pair) (gate_inv_sigma2 = NULL).  Per point (skipped when valid[i] == 0; valid may be NULL):
 * p3Dc = Tcw * p in Sophus' QUATERNION form (Thirdparty/Sophus/sophus/so3.hpp:356-367, se3.hpp:319-324 -- the reference never
 * goes through a rotation matrix here, and the two forms differ in the last float ulp), depth >= 0, KeyFrame::IsInImage,
 * dist in [0.8*min_dist, 1.2*max_dist], PO.Pn >= 0.5*dist, level = MapPoint::PredictScale, radius = th *
 * scale_factors[level], candidates = KeyFrame::GetFeaturesInArea(u, v, radius) with octave in [level-1, level], minus
 * skip[idx] != 0 (skip may be NULL; `cap` bytes of the train frame).  out[i] = best / second best (strict '<', first
 * wins); proj[i] (may be NULL) = projection, radius and predicted level (-1: rejected before the search).
 * Host pointers (synchronous) or device pointers (asynchronous on `stream`). */
typedef struct { float q[4], t[3]; } dvm_se3f;   /* Sophus::SE3f: unit_quaternion().coeffs() = (x, y, z, w), translation() */
typedef struct { float q[4], t[3]; } dvm_sim3f;  /* Sophus::Sim3f: rxso3().quaternion().coeffs() (scale = |q|^2), translation() */
typedef struct {
  /* Tcw: KeyFrame::GetPose() / Frame::GetPose(); for the Sim3 variants SE3f(Scw.rotationMatrix(), Scw.translation() /
   * Scw.scale()) (ORBmatcher.cc:403,505,1245 -- dvm_host::Sim3ToSE3 derives it in the reference's order).
   * Ow: GetCameraCenter() resp. Tcw.inverse().translation(). */
  dvm_se3f Tcw;
  float Ow[3], fx, fy, cx, cy, min_x, max_x, min_y, max_y, log_scale_factor;
  int32_t n_levels;
  /* sim3_pair == 1 selects the ORBmatcher::SearchBySim3 form (ORBmatcher.cc:1380-1437): p' = S2 * (Tcw * p) with Tcw the
   * OTHER keyframe's pose and S2 = S21 resp. S12 (Sim3f action rxso3.hpp:265-273, sim3.hpp:226-229),
   * u = fx * (X * invz) + cx with invz = 1.0 / Z, distance = |p'|, no viewing-angle test; Ow is not used.
   * sim3_pair == 2 selects the relocalisation form, ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th,
   * ORBdist) (:1750-1860): no depth test, bounds inclusive at both ends, no viewing-angle test, octaves [level-1, level+1]. */
  int32_t sim3_pair;
  dvm_sim3f S2;
} dvm_kf_camera;
typedef struct { float u, v, radius; int32_t level; } dvm_projection;
int dvm_project_search(const dvm_frame* train, int slot, const uint8_t* skip, const dvm_kf_camera* cam, const float* P,
                       const float* normal, const float* min_dist, const float* max_dist, const uint8_t* desc,
                       const uint8_t* valid, int n, float th, const float* scale_factors, const float* gate_inv_sigma2,
                       double gate, dvm_match* out, dvm_projection* proj, int on_device, void* stream);

/* ORBmatcher::SearchForTriangulation inner loop (ORBmatcher.cc:905-998, monocular): query q = keypoint qidx[q] of KF1
 * scans the KF2 keypoints cand[off[q] .. off[q+1]) (entries < 0 skipped); a candidate is kept if dist <= 50, dist <= best
 * so far, it is not within sqrt(100*scale_factors2[octave]) px of the epipole `ep`, and (coarse != 0 or)
 * Pinhole::epipolarConstrain (CameraModels/Pinhole.cpp:104-127) holds with the fundamental matrix F12 (row-major 3x3,
 * float) and level_sigma2_2[octave].  best_idx[q] = KF2 index (-1 none), best_dist[q] (256 none); a tie goes to the LAST
 * candidate, as the reference's "dist > bestDist -> continue" does. */
int dvm_match_triangulation(const uint8_t* desc1, const dvm_keypoint* kps1, int n1, const int32_t* qidx, int nq,
                            const uint8_t* desc2, const dvm_keypoint* kps2, int n2, const int32_t* off, const int32_t* cand,
                            const float* F12, const float* ep, int coarse, const float* scale_factors2,
                            const float* level_sigma2_2, int nlevels, int32_t* best_idx, int32_t* best_dist, int on_device,
                            void* stream);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:384-453), batched: map point p owns the descriptors
 * desc[off[p] .. off[p+1]) (32 B each, its observations in the reference's iteration order); best_idx[p] = index
 * inside that range of the descriptor with the least median Hamming distance to the others (median =
 * sorted_row[0.5 * (N - 1)], first strictly smaller wins), best_median[p] = that median.  Empty range: -1;
 * more than 512 observations: -2 (the caller keeps its CPU path for those).
 * Host pointers (synchronous) or device pointers (asynchronous on `stream`). */
int dvm_distinctive_descriptors(const uint8_t* desc, const int32_t* off, int n_points, int32_t* best_idx, int32_t* best_median,
                                int on_device, void* stream);

/* DBoW2 vocabulary tree on the device + per-feature transform (reference Thirdparty/DBoW2/DBoW2/
 * TemplatedVocabulary.h:1098-1138, FORB::distance FORB.cpp:80-97).  Nodes 0..n_nodes-1, node 0 = root;
 * children[child_off[i] .. child_off[i+1]) in m_nodes[i].children order (empty = leaf); desc 32 B per node; weight and
 * word_id as in m_nodes[i] (word_id < 0 for inner nodes); L = m_L.  dvm_vocab_transform fills, for feature f,
 * word_id / weight of the leaf reached and node_id = the node on the path at level L - levelsup (0 if that level is
 * <= 0; -1 where the reference leaves *nid unset because the leaf lies above that level).  The BowVector /
 * FeatureVector bookkeeping on top of it is host code (dvm_slam_amd/host/orb_vocabulary.cpp). */
typedef struct dvm_vocab dvm_vocab;
int dvm_vocab_create(int device, int n_nodes, const int32_t* child_off, const int32_t* children, const uint8_t* desc,
                     const double* weight, const int32_t* word_id, int L, dvm_vocab** out);
void dvm_vocab_destroy(dvm_vocab* v);
int dvm_vocab_transform(const dvm_vocab* v, const uint8_t* features, int n, int levelsup, int32_t* word_id, int32_t* node_id,
                        double* weight, int on_device, void* stream);

/* BowVectors of the keyframes of a KeyFrameDatabase on the device + the per-query work of its place-recognition searches
 * (reference src/KeyFrameDatabase.cc:43-70 add / erase, :224-808 DetectCandidates / DetectNBestCandidates /
 * CalculateMergeScore): for every stored keyframe at once, the number of words it shares with the query BowVector
 * (= mnPlaceRecognitionWords after the inverted-file walk), the first shared word (with the insertion order this gives the
 * keyframe's position in lKFsSharingWords) and mpVoc->score(query, keyframe) (L1Scoring, cast to float as the reference
 * stores it).  Word ids ascending (std::map order).  The covisibility accumulation and candidate selection on top are
 * host code (dvm_slam_amd/host/keyframe_database.cpp). */
typedef struct dvm_bowdb dvm_bowdb;
int dvm_bowdb_create(int device, dvm_bowdb** out);
void dvm_bowdb_destroy(dvm_bowdb* db);
/* append a keyframe's BowVector; *slot receives its index (slots are never reused) */
int dvm_bowdb_add(dvm_bowdb* db, const int32_t* word_ids, const double* values, int n, int32_t* slot);
int dvm_bowdb_erase(dvm_bowdb* db, int32_t slot);
int dvm_bowdb_size(const dvm_bowdb* db);   /* slots handed out so far */
/* common[s] = shared words (-1: erased slot), first_word[s] = smallest shared word id (-1 none), score[s]; arrays of
 * dvm_bowdb_size() entries, host pointers, synchronous. */
int dvm_bowdb_query(dvm_bowdb* db, const int32_t* word_ids, const double* values, int n, int32_t* common, int32_t* first_word,
                    float* score);
/* out[4] = {slots ever added, live slots, words stored (live + erased, shrinks when the store is repacked), word capacity}.
 * Erased slots keep their number; their words are reclaimed by a repack once they outnumber the live words. */
int dvm_bowdb_stats(const dvm_bowdb* db, int64_t out[4]);

/* TrackWithMotionModel-style frame-to-frame search over a batch (ORBmatcher.cc:1596-1611, mono):
 * pair i (0 <= i < count) searches train slot first_slot+i for every keypoint of frame i-1 of the
 * device arrays (d_kps + (i-1)*kps_stride, ...); pair 0 takes its queries from the carry frame
 * (d_carry_*, may be NULL: pair 0 then yields 0 matches).  Window th*scale[octave], octaves
 * [o-1,o+1].  d_out + i*out_stride receives `cap` dvm_match per pair, d_nq_out[i] the query count. */
int dvm_match_frames_batch(const dvm_frame* train, int first_slot, int count, const dvm_keypoint* d_kps,
                           int64_t kps_stride, const uint8_t* d_desc, int64_t desc_stride, const int32_t* d_n,
                           const dvm_keypoint* d_carry_kps, const uint8_t* d_carry_desc, const int32_t* d_carry_n,
                           int cap, float th, const float* d_scale_factors, int nlevels, dvm_match* d_out,
                           int64_t out_stride, int32_t* d_nq_out, void* stream);

/* --------------------------------------------------------------------------- bundle adjustment */
/* Optimizer::BundleAdjustment / LocalBundleAdjustment (Optimizer.cc:55-356,1030-1387): SE3 camera
 * vertices (fixed or free), XYZ landmark vertices (all marginalised), one EdgeSE3ProjectXYZ per
 * observation with information I*inv_sigma2 and an optional Huber kernel, solved by g2o's
 * BlockSolver_6_3 + Levenberg recipe in FP64.  Cameras are (tx,ty,tz,qx,qy,qz,qw), world->camera. */
typedef struct { int32_t pose, point; double u, v, inv_sigma2; } dvm_ba_edge;
typedef struct { double fx, fy, cx, cy, huber_delta; /* <= 0: no robust kernel (bRobust=false) */ } dvm_ba_camera;
typedef struct {
  int32_t iterations, total_trials, stop_reason, kernel_us; /* stop: 0 iteration budget / stop flag, 1 LM terminate, 2 Mur-Artal criterion;
                                                          kernel_us: dvm_ba_optimize_windows_fast only -- the launch's duration (HIP events), 0 elsewhere */
  double chi2_initial, chi2_final, lambda_final;
  int32_t trials_per_iter[64];
  double chi2_per_iter[64], lambda_per_iter[64];
  double ms_structure, ms_optimize;                   /* host wall time: graph build / optimize() */
  int32_t spec_trials, spec_kept;                     /* trials whose successor was enqueued on the device's own accept decision,
                                                         and how many of those the host's decision confirmed bit for bit */
} dvm_ba_stats;
typedef struct dvm_ba dvm_ba;
int dvm_ba_create(int device, dvm_ba** out);
void dvm_ba_destroy(dvm_ba* h);
/* builds the graph (vertex order, incidence lists, reduced-camera block pattern) and uploads it */
int dvm_ba_set_problem(dvm_ba* h, const double* poses, const uint8_t* fixed, int P, const double* points, int L,
                       const dvm_ba_edge* edges, int E, const dvm_ba_camera* cam);
/* optimizer.optimize(iterations); stop_flag (may be NULL) is g2o's forceStopFlag: polled between
 * iterations and trials, may be written by another thread (LocalMapping.cc:305,359).
 * ACCURACY CONTRACT (tests/test_gpu_ba_weak.py, tests/test_gpu_config_size.py, tests/test_gpu_ba_window.py), against the CPU
 * restatement of g2o's recipe (oracle/ba_oracle.cpp; the reference itself cannot be run here -- "parity unpinned"):
 *   - problems with <= 6 free cameras run every sum in g2o's own sequential order: BIT-IDENTICAL states, chi2, lambda, trial sequence;
 *   - larger problems are solved with parallel (tile) summation orders: the LM accept / reject sequence is identical and poses and
 *     landmarks agree within 1e-6 -- EXCEPT on weakly constrained problems (two- or three-view landmarks, few hundred points: a gauge
 *     that is barely held), where the result of the recipe itself moves by more than 1e-6 when nothing but the ORDER of the edge list
 *     changes.  The reference adds its edges in heap-address order (std::map<KeyFrame*, ...>, Optimizer.cc:1108-1230), so its own result is
 *     one sample of that spread.  There the bound is 10 x the distance between two runs of the restatement on permuted edge lists,
 *     measured per problem by the test; observed: up to 4e-5 on an 87-keyframe / 273-landmark / 3-view problem whose own spread is 2e-5. */
int dvm_ba_optimize(dvm_ba* h, int iterations, const volatile uint8_t* stop_flag, dvm_ba_stats* stats);
int dvm_ba_get_result(dvm_ba* h, double* poses, double* points);
/* Two-round solves on one graph -- the welding bundle adjustment of a map merge (Optimizer.cc:3474-3519: optimize(5), then
 * e->setLevel(1) on the outliers, e->setRobustKernel(0) on every edge, initializeOptimization(0), optimize(10)) -- without a
 * second dvm_ba_set_problem: flags[e] bit 0 = the edge stays at level 0 (a level-1 edge is outside the active set: it
 * contributes nothing and dvm_ba_edge_chi2 keeps reporting the chi2 of its last evaluation, as g2o's e->chi2() does), bit 1 =
 * the edge keeps its robust kernel (huber_delta of the problem).  Takes effect with the next dvm_ba_optimize, which starts
 * from the estimates the previous one left (iteration 0 again: fresh computeLambdaInit).  NULL: every edge active and
 * robust again.  Not available on a landmark-sharded problem. */
#define DVM_BA_EDGE_ACTIVE 1
#define DVM_BA_EDGE_ROBUST 2
int dvm_ba_set_edge_flags(dvm_ba* h, const uint8_t* flags);
/* BASELINE.json config 5 (global BA sharded over the GPUs of a node; the reference solves it on one CPU thread,
 * Optimizer.cc:44-53 -> block_solver.hpp:381-439): rank r of `world` receives the WHOLE problem (so that the free-camera
 * order, the block pattern and the tile schedule are identical everywhere) but evaluates only the observations of the
 * landmarks it owns (point % world == rank): edge pass, Hll / Hpl, its share of Hpp / b and of the Schur complement.  Per LM
 * trial the partial reduced camera systems -- the structurally non-zero 64x64 tiles, rhs row included -- are summed over the
 * ranks through the caller's all-reduce (RCCL over xGMI), every rank factors the sum redundantly, back-substitutes its own
 * landmarks and moves all cameras; chi2 / scale are two more host scalars.  dvm_ba_optimize ends with one exchange of the
 * landmarks, so dvm_ba_get_result returns the full state on every rank.  dvm_ba_edge_chi2 then covers the LOCAL edges
 * (those with point % world == rank, in input order).
 * dvm_allreduce_fn: in-place reduction over all ranks of n doubles at `buf` -- device memory (on_host = 0; must be ordered
 * after the work already queued on `stream` and before work queued later) or host memory (on_host = 1); op 0 = sum,
 * 1 = max; returns 0 on success.  d_buf: device buffer of at least dvm_ba_allreduce_doubles() doubles owned by the caller
 * (e.g. a torch tensor, so that torch.distributed can reduce it). */
typedef int (*dvm_allreduce_fn)(void* ctx, void* buf, int64_t n, int on_host, int op, void* stream);
int dvm_ba_set_problem_sharded(dvm_ba* h, const double* poses, const uint8_t* fixed, int P, const double* points, int L,
                               const dvm_ba_edge* edges, int E, const dvm_ba_camera* cam, int rank, int world);
int dvm_ba_set_allreduce(dvm_ba* h, dvm_allreduce_fn fn, void* ctx, void* d_buf, int64_t cap_doubles);
int64_t dvm_ba_allreduce_doubles(const dvm_ba* h);
/* Measurement aids (bench.py).  dvm_ba_schedule_info: the symbolic tile factorisation of the current problem,
 * out[12] = {levels launched, tile columns, strips (trsm tiles), update targets, (target, contributor) products, of which on
 * diagonal targets, non-zero tiles, ldS, free cameras, non-zero 6x6 blocks, local edges, tiles per side} -- what the executed
 * FLOP count of one LM trial follows from.  dvm_ba_profile: enable (1) / disable (0) / read only (-1) HIP-event timing of the
 * four phases of a trial; ms4 = {linearise, Schur complement, tile Cholesky + back substitution, landmarks + update + chi2}
 * accumulated since the last enable, over *trials trials / *iters iterations. */
int dvm_ba_schedule_info(const dvm_ba* h, int64_t* out);
/* dvm_ba_solve_info: which form of the reduced solve (the replacement of g2o's LinearSolverEigen::solve,
 * Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:89-112, under BlockSolver::solve, g2o/core/block_solver.hpp:354-486) the current
 * problem runs: out[6] = {form: 0 = one launch (or two) per elimination-tree level, 1 = the flow form -- the whole factorisation + back
 * substitution as ONE persistent launch of tile tasks, chosen from 5 elimination levels on while the tasks are few per workgroup --, tile tasks of the flow
 * form, chains (= leaves of the elimination tree), workgroups launched, KEPT landmarks, camera tiles}.  The two forms sum in the same order
 * (same numbers).  Kept landmarks: when a few landmarks seen from places far apart on the trajectory are what makes the elimination tree
 * deep (>= 12 levels), up to 210 of them are not eliminated into the Schur complement but stay unknowns of the reduced system, in tiles of
 * their own behind the camera tiles -- the same linear system as g2o's, eliminated in another order (results agree within the 1e-6 of
 * the accuracy contract, not bit for bit).  DVM_BA_BORDER=0 in the environment switches that off. */
int dvm_ba_solve_info(const dvm_ba* h, int64_t* out);
int dvm_ba_profile(dvm_ba* h, int enable, double* ms4, int32_t* trials, int32_t* iters);
/* per-edge chi2() as g2o reports it after optimize(), and isDepthPositive() (outlier tests of
 * Optimizer.cc:1317-1354); either output may be NULL */
int dvm_ba_edge_chi2(dvm_ba* h, double* chi2, uint8_t* depth_positive);
void* dvm_ba_stream(dvm_ba* h);

/* K INDEPENDENT bundle adjustments in one launch: the LocalBundleAdjustment windows (Optimizer.cc:1030-1387) of several agents that
 * share this GPU (BASELINE.json config 4 with more agents than GPUs), or the GlobalBundleAdjustemnt of a freshly initialised
 * two-keyframe map (Tracking.cc:2330, Optimizer.cc:55-356).  Every window is what one dvm_ba_set_problem + dvm_ba_optimize(iterations)
 * + dvm_ba_get_result + dvm_ba_edge_chi2 sequence would be given and return; one workgroup per window runs the whole
 * optimizer.optimize(iterations) on the device.
 * Summation order: every sum of a window runs in the order of g2o's single-threaded code (edges in input order per Hessian block
 * and in chi2, landmarks in index order in the Schur complement, columns in order in the Cholesky factorisation) and sin / cos /
 * pow(x, 3) are the fixed double-precision sequences of csrc/f64_spec.h, so the result is BIT-IDENTICAL to the CPU restatement of
 * g2o's recipe -- which matters for windows with a weak gauge (one or two free cameras), whose result moves by 1e-3 and more when
 * only the order of the floating-point sums changes.
 * Limits: at most 30 free cameras per window (DVM_ERR_CAPACITY above; dvm_ba_optimize has no limit).  Host pointers, synchronous.
 * stop_flag (may be NULL) is polled by the kernel between iterations and trials, for all windows.  stats: K entries or NULL;
 * ms_optimize = wall time of the whole call (upload, launch, download), ms_structure = this window's share of the host set-up. */
typedef struct {
  int32_t n_poses, n_points, n_edges, iterations;
  const double* poses;         /* [n_poses][7]  (tx,ty,tz,qx,qy,qz,qw), world -> camera */
  const uint8_t* fixed;        /* [n_poses] */
  const double* points;        /* [n_points][3] */
  const dvm_ba_edge* edges;    /* [n_edges], pose / point are indices into THIS window's arrays */
  dvm_ba_camera cam;
  double* poses_out;           /* [n_poses][7] or NULL */
  double* points_out;          /* [n_points][3] or NULL */
  double* edge_chi2_out;       /* [n_edges] or NULL: e->chi2() after the last computeActiveErrors */
  uint8_t* depth_positive_out; /* [n_edges] or NULL: isDepthPositive() at the result */
} dvm_ba_window;
int dvm_ba_optimize_windows(int device, const dvm_ba_window* windows, int K, const volatile uint8_t* stop_flag, dvm_ba_stats* stats);
/* The same K windows, ONE launch, the whole Levenberg-Marquardt control on the device -- with every sum as a tree in a FIXED order instead of
 * g2o's sequential one, and a CLUSTER of G workgroups per window (G = the largest of 8 / 4 / 2 / 1 for which the launch's
 * 8 * ceil(K / 8) * G workgroups are resident at once; DVM_BA_CLUSTER forces it): the data-parallel phases are split over the cluster and
 * separated by agent-scope barriers, workgroup 0 solves the reduced system in its LDS.  One window: 2.5 ms (tile solver 1.1 ms, the
 * sequential-order kernel 15 ms); 32 windows: 2.8 ms of kernel, 4.2-4.8 ms per call from host arrays = 67-76 k LM iterations/s; 128 windows: 15-17 ms =
 * 75-84 k.  Deterministic and
 * independent of G and of the other windows of the call; equal to the CPU recipe to the general solver's contract -- same LM trial
 * sequence, poses / landmarks within 1e-6 on well-posed windows -- not bit for bit.  Up to 30 free cameras per window (DVM_ERR_CAPACITY
 * beyond: dvm_ba_optimize_batch).  A barrier that times out (a workgroup of a cluster not resident: other work holds compute units) makes the
 * call repeat the batch with G = 1, which needs no co-residency.  The calling thread keeps its windows' host tables for the next call
 * (~2 MB per window).  This is the call for the LocalBundleAdjustment windows of several agents sharing a GPU (Optimizer.cc:1030-1387,
 * LocalMapping.cc:172). */
int dvm_ba_optimize_windows_fast(int device, const dvm_ba_window* windows, int K, const volatile uint8_t* stop_flag, dvm_ba_stats* stats);
/* The shared form for several agents' LocalMapping threads (as dvm_orb_pool / dvm_pose_pool for the tracking threads): dvm_ba_pool_optimize
 * is dvm_ba_optimize_windows_fast for ONE window -- a blocking call from any number of threads; calls that arrive within `window_us`
 * (< 0: 300) of each other run as ONE launch.  The result of a window does not depend on the batch it rode in: every caller gets the bits
 * of a solo dvm_ba_optimize_windows_fast call.  batch_size (may be NULL): how many windows the call's launch held.  No stop flag. */
typedef struct dvm_ba_pool dvm_ba_pool;
int dvm_ba_pool_create(int device, int max_batch, int window_us, dvm_ba_pool** out);
void dvm_ba_pool_destroy(dvm_ba_pool* pool);
int dvm_ba_pool_optimize(dvm_ba_pool* pool, const dvm_ba_window* w, dvm_ba_stats* stats, int* batch_size);
/* The same K independent problems solved CONCURRENTLY on the general solver: up to `threads` (<= 0: 4) host threads of this call each
 * drive one solver handle from a process-wide pool and pull windows from a shared counter, so that the launch chains of different
 * windows interleave on the device (one window alone leaves it mostly idle).  Window k gets exactly what dvm_ba_set_problem +
 * dvm_ba_optimize(iterations) + dvm_ba_get_result + dvm_ba_edge_chi2 on a handle of its own give -- no limit on the number of free
 * cameras, parity with the CPU recipe as for dvm_ba_optimize (windows of at most 6 free cameras take the sequential-order kernel
 * there and are bit-identical).  This is the call for several agents' LocalBundleAdjustment windows per GPU
 * (Optimizer.cc:1030-1387, one LocalMapping thread per agent in the reference).  stats: K entries or NULL. */
int dvm_ba_optimize_batch(int device, const dvm_ba_window* windows, int K, int threads, const volatile uint8_t* stop_flag,
                          dvm_ba_stats* stats);
/* csrc/f64_spec.h evaluated on the device for n arguments: out = [sin(x) | cos(x) | x^3], 3 n doubles.  A test aid: the host build of
 * the spec, the device build and the oracle's restatement must agree bit for bit (tests/test_f64_spec.py, tests/test_gpu_ba_window.py). */
int dvm_f64_spec_eval(int device, const double* x, int n, double* out);

/* Optimizer::PoseOptimization (Optimizer.cc:744-1028, monocular edges): `batch` independent frames.
 * Frame f: pose (tx,ty,tz,qx,qy,qz,qw) at pose_in + 7f; n[f] 2D-3D matches stored with a common
 * `stride` (Xw [batch][stride][3], obs [batch][stride][2] = undistorted keypoint, inv_sigma2
 * [batch][stride] = mvInvLevelSigma2[octave]).  Outputs: optimised pose, mvbOutlier flags, and the
 * reference's return value nInitialCorrespondences - nBad.  Host pointers, synchronous; each frame is
 * one workgroup running the 4 x optimize(10) Levenberg rounds entirely on the device. */
int dvm_pose_optimize(int device, const double* pose_in, const double* Xw, const double* obs, const double* inv_sigma2,
                      const int32_t* n, int stride, int batch, const dvm_ba_camera* cam, double* pose_out,
                      uint8_t* outlier, int32_t* n_inliers);

/* The shared form for several agents on one GPU (as dvm_orb_pool for the extractor): dvm_pose_pool_optimize is dvm_pose_optimize for ONE
 * frame -- same arguments, same results bit for bit --, callable from any number of threads; calls that arrive within `window_us` of each
 * other and share the camera run as ONE launch of the batched kernel (a workgroup per frame).  A frame with more than 1 280
 * correspondences is handed to dvm_pose_optimize directly.  batch_size (may be NULL): how many frames the call's launch held. */
typedef struct dvm_pose_pool dvm_pose_pool;
int dvm_pose_pool_create(int device, int max_batch, int window_us, dvm_pose_pool** out);
void dvm_pose_pool_destroy(dvm_pose_pool* pool);
int dvm_pose_pool_optimize(dvm_pose_pool* pool, const double* pose_in, const double* Xw, const double* obs, const double* inv_sigma2, int n,
                           const dvm_ba_camera* cam, double* pose_out, uint8_t* outlier, int32_t* n_inliers, int* batch_size);

/* ---- The tracking step of one frame as ONE device chain (csrc/track.cpp): Frame::Frame -> ExtractORB (src/Frame.cc:371-411), then
 * Tracking::TrackWithMotionModel (src/Tracking.cc:2584-2667) = SearchByProjection(CurrentFrame, LastFrame) (src/ORBmatcher.cc:1553-1748)
 * -> Optimizer::PoseOptimization (src/Optimizer.cc:744-1028) -> outlier matches dropped (:2636-2660).
 *   dvm_track_begin   queues the extraction of the frame on the extractor's stream and returns;
 *   dvm_track_finish  queues [Frame::UndistortKeyPoints] -> AssignFeaturesToGrid -> the ranked window search of the caller's projection
 *                     queries -> claim replay + rotation histogram -> PoseOptimization -> outlier flags behind it, synchronises once and
 *                     returns the frame's keypoints / descriptors (what dvm_orb_extract returns) together with the tracking results.
 * The queries are what the loop header of ORBmatcher.cc:1573-1611 produces from LastFrame's map points and the predicted pose (the
 * host library's dvmh_track_with_motion_model builds them while the extraction runs).  CurrentFrame.mvpMapPoints is taken as cleared
 * (Tracking.cc:2603).  Results equal dvm_orb_extract + dvmh_search_by_projection_frames + dvm_pose_optimize bit for bit. */
typedef struct dvm_tracker dvm_tracker;
typedef struct {
  int32_t nq;
  const uint8_t* qdesc;            /* [nq][32] the map points' descriptors */
  const float *qx, *qy, *qr;       /* projection (u, v) and window radius th * mvScaleFactors[nLastOctave] */
  const int32_t *qmin, *qmax;      /* nLastOctave - 1, nLastOctave + 1 */
  const uint8_t* q_claims;         /* [nq] the map point has Observations() > 0 */
  const float* q_angle;            /* [nq] LastFrame.mvKeysUn[i].angle */
  const float* q_pos;              /* [nq][3] GetWorldPos() */
  float bounds[4];                 /* mnMinX mnMaxX mnMinY mnMaxY */
  const dvm_distortion* dist;      /* NULL or k1 == 0: mvKeysUn = mvKeys */
  const float* inv_level_sigma2;   /* mvInvLevelSigma2, nlevels entries */
  int32_t nlevels;
  dvm_ba_camera cam;               /* fx fy cx cy of PoseOptimization's edges */
  double pose_in[7];               /* the predicted Tcw (tx ty tz qx qy qz qw): mVelocity * mLastFrame.GetPose() */
  int32_t th_high, check_ori;      /* ORBmatcher::TH_HIGH (100), mbCheckOrientation */
  int32_t min_matches;             /* 20 (Tracking.cc:2616) */
} dvm_track_queries;
enum { DVM_TRACK_COMPLETE = 0, DVM_TRACK_FEW_MATCHES = 1, DVM_TRACK_REPLAY_ON_HOST = 2 };
typedef struct {
  int32_t n, mono_index;           /* the extraction's keypoint count and monoIndex */
  int32_t status;                  /* DVM_TRACK_COMPLETE; FEW_MATCHES: nmatches < min_matches, no pose (search again with the doubled window:
                                      dvm_track_finish with the wider queries); REPLAY_ON_HOST: kept for callers of older builds -- the device now
                                      searches such a query's window again itself (n_requeried), the status is no longer produced */
  int32_t nmatches;                /* SearchByProjection's return value */
  int32_t nmatches_before_rotation;
  int32_t n_edges, n_inliers;      /* PoseOptimization: nInitialCorrespondences and its return value */
  int32_t nmatches_map, nmatches_after;   /* Tracking.cc:2636-2660: nmatchesMap and nmatches after the outliers were dropped */
  int32_t n_requeried;             /* queries whose four ranked candidates were all taken by earlier queries: their windows were searched
                                      again on the device at their turn (ORBmatcher.cc:1613-1650 skips taken keypoints) */
  double pose[7];                  /* the optimised Tcw */
} dvm_track_result;
int dvm_tracker_create(int device, int max_keypoints, int max_queries, dvm_tracker** out);
/* The same for up to max_frames frames per call -- the frames of several agents sharing the GPU at one camera tick: ONE chain of batched
 * launches (extraction of the batch, max_frames grids, ranked searches, claim replays and PoseOptimizations side by side, a workgroup per
 * frame) behind ONE synchronisation.  The extractor handle needs max_batch >= count.  The frames of a call share camera, image bounds,
 * level table and matcher thresholds, and are not undistorted here (k1 != 0: the single-frame call).  Frame b's results equal
 * dvm_track_begin / dvm_track_finish on that frame alone, bit for bit. */
typedef struct {
  dvm_keypoint* kps; uint8_t* desc; int32_t cap;   /* [cap] the extraction (as dvm_orb_extract) */
  dvm_keypoint* kps_un;                            /* [cap] or NULL: mvKeysUn */
  int32_t* assign; uint8_t* outlier;               /* [cap] as dvm_track_finish */
  uint32_t* ranked;                                /* NULL, or [nq][4] (REPLAY_ON_HOST only) */
} dvm_track_frame_out;
int dvm_tracker_create_batch(int device, int max_frames, int max_keypoints, int max_queries, dvm_tracker** out);
int dvm_track_begin_batch(dvm_tracker* t, dvm_orb* h, const uint8_t* imgs, int count, int rows, int cols, int stride, int64_t frame_stride, int lap0, int lap1);
/* the same for `count` frames the caller has written into the extractor's page-locked input buffer (dvm_orb_staging): no host copy */
int dvm_track_begin_staged(dvm_tracker* t, dvm_orb* h, int count, int rows, int cols, int lap0, int lap1);
void dvm_tracker_destroy(dvm_tracker* t);
int dvm_track_begin(dvm_tracker* t, dvm_orb* h, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1);
/* kps / desc [cap]: the extraction; kps_un (may be NULL): mvKeysUn; assign [cap]: per keypoint the index of the query matched to it or -1
 * (BEFORE the outlier drop); outlier [cap]: mvbOutlier as PoseOptimization leaves it (the matches Tracking then drops); ranked (may be
 * NULL): [nq][4] candidate lists, filled for REPLAY_ON_HOST only.  May be called again after one dvm_track_begin (wider window). */
int dvm_track_finish(dvm_tracker* t, dvm_orb* h, const dvm_track_queries* q, dvm_keypoint* kps, uint8_t* desc, int cap, dvm_keypoint* kps_un,
                     int32_t* assign, uint8_t* outlier, uint32_t* ranked, dvm_track_result* res);
/* qs / outs / res: `count` entries, frame b of the dvm_track_begin_batch call.  A frame with fewer than min_matches matches reports
 * DVM_TRACK_FEW_MATCHES; the call may be repeated (all frames, the wider queries for those) after one begin. */
int dvm_track_finish_batch(dvm_tracker* t, dvm_orb* h, int count, const dvm_track_queries* qs, const dvm_track_frame_out* outs, dvm_track_result* res);

/* Optimizer::OptimizeSim3 (Optimizer.cc:1960-2212), numerics for N correspondences gathered by the caller:
 * P1c / P2c = the matched map points in their own key frame's camera frame (R1w*P+t1w, R2w*P+t2w), obs1 / obs2 =
 * undistorted keypoints in KF1 / KF2, w1 / w2 = mvInvLevelSigma2[octave], K1 / K2 = (fx,fy,cx,cy) of both pinhole
 * cameras, th2 = chi-square gate (Huber delta = sqrt(th2)).  S12 (in/out) = (qx,qy,qz,qw, tx,ty,tz, s) = g2o::Sim3.
 * Runs optimize(5) -> inlier test -> kernels off -> optimize(5|10) -> final test on the device (one workgroup).
 * inlier[N] = 1 for surviving pairs; *n_inliers = the reference's return value (0 if < 10 pairs survive round 1).
 * Host pointers, synchronous. */
int dvm_optimize_sim3(int device, double* S12, int fix_scale, const double* P1c, const double* P2c, const double* obs1,
                      const double* obs2, const double* w1, const double* w2, int N, const double* K1, const double* K2,
                      double th2, uint8_t* inlier, int32_t* n_inliers);

/* Optimizer::OptimizeEssentialGraph numerics (src/Optimizer.cc:1389-1652): Sim3 pose graph of n vertices
 * (S[n][8] = estimates Siw as (q_xyzw, t, s), in/out; fixed[v] != 0 keeps a vertex) and E EdgeSim3 (vertex 0 = vi,
 * vertex 1 = vj, measurement Sji = Sjw * Swi, information = identity).  g2o's numeric Jacobians, Levenberg-Marquardt
 * with lambda_init 1e-16, `iterations` = optimize(20) in the reference.  Which keyframes / edges enter the graph and the
 * SE3 / map-point correction afterwards (:1601-1648) stay with the caller.  Host pointers, synchronous. */
typedef struct { int32_t vi, vj; double Sji[8]; } dvm_pg_edge;
typedef struct {
  int32_t iterations, total_trials, stop_reason, levels;
  double chi2_initial, chi2_final, lambda_final, tile_fill, ms_structure, ms_optimize;
  double chi2_per_iter[32];   /* chi2 after each outer iteration (first 32) */
  int32_t trials_per_iter[32];
} dvm_pg_stats;
int dvm_pose_graph_optimize(int device, double* S, const uint8_t* fixed, int n, const dvm_pg_edge* edges, int E,
                            int fix_scale, int iterations, dvm_pg_stats* stats);

/* Sim3Solver::ComputeSim3 + CheckInliers (src/Sim3Solver.cc:294-408) for H RANSAC hypotheses in one launch.
 * P1c / P2c: the N matched map points in the two keyframes' camera frames (mvX3Dc1 / mvX3Dc2, float[3N]);
 * max_err1/2: mvnMaxError1/2 as the reference stores them, (float)(size_t)(9.210 * sigma2); K1 / K2: fx, fy, cx, cy;
 * triples: the minimal sets (3 correspondence indices each; the reference draws them with DUtils::Random, :171-181).
 * Outputs per hypothesis: T12[13] = {s12, R12 row-major, t12}, n_inliers, inlier_mask[N].  The sequential
 * best-hypothesis rule of iterate() (:154-185) stays with the caller.  Host pointers, synchronous. */
int dvm_sim3_hypotheses(int device, const float* P1c, const float* P2c, const float* max_err1, const float* max_err2, int N,
                        const float* K1, const float* K2, const int32_t* triples, int H, int fix_scale, float* T12,
                        int32_t* n_inliers, uint8_t* inlier_mask);

#ifdef __cplusplus
}
#endif
#endif
