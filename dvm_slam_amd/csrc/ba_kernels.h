// dvm_slam_amd/csrc/ba_kernels.h -- device view + launchers of the FP64 bundle-adjustment kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

namespace dvm {

constexpr int kSchurLmLandmarks = 32, kSchurLmRows = 256, kSchurLmPairs = 1536, kSchurLmRuns = 512;   // a chunk of k_schur_lm: W rows 37 KB + Dinv + its index lists = 50 KB of LDS, three workgroups per CU (750 workgroups for the 500-keyframe problem: one round)
constexpr int kEdgeLinStride = 16;  // doubles per edge and array (one 128-byte line each): e_lin = B[12] w wr0 wr1 + pad (what the camera
                                    // accumulation reads), e_linA = A[6] w wr0 wr1 + pad (what the landmark accumulation reads).  As ONE 192-byte row both
                                    // gathers dragged 2-3 lines per edge through the memory system: ~100 MB for 30 MB of operands
// first row of camera i (elimination order) in the tiled reduced system: 10 whole cameras per 64-row tile
__host__ __device__ inline int ba_row(int i) { return (i / 10) * 64 + (i % 10) * 6; }

// Structure-of-arrays problem state in HBM (DESIGN.md "BA layout").  Poses are (t, q_xyzw) doubles.
struct BaView {
  int32_t P, L, E, nfree, nblk, ldS;
  double fx, fy, cx, cy, delta;
  double* poses;            // [P][7]
  double* points;           // [L][3]
  const int32_t* pidx;      // [P] free index or -1
  const int32_t* free_pose; // [nfree] pose index
  const int32_t* e_pose;    // [E]
  const int32_t* e_point;   // [E]
  const double* e_obs;      // [E][2]
  const double* e_info;     // [E]
  const uint8_t* e_flags;   // [E] or null (= every edge active and robust).  bit 0: the edge is at level 0 (g2o: part of the active set; a level-1 edge
                            // contributes nothing and keeps the chi2 of its last evaluation), bit 1: it keeps its robust kernel (else delta = 0)
  double* e_chi2;           // [E] chi2 at the last evaluation (what g2o's e->chi2() reports)
  double* e_lin;            // [E][kEdgeLinStride]  pose half of the linearisation
  double* e_linA;           // [E][kEdgeLinStride]  landmark half
  double* e_W;              // [E][18]  Hpl block (pose 6 x point 3)
  const int32_t* pt_start;  // [L+1] CSR landmark -> edges (input order)
  const int32_t* pt_edges;  // [E]
  const int32_t* pt_fi;     // [E] free index (or -1) of the camera of edge pt_edges[i]: spares the landmark back substitution two dependent loads
  const int32_t* ps_start;  // [P+1] CSR camera -> edges
  const int32_t* ps_edges;  // [E]
  double *Hpp, *bp;         // [nfree][36], [nfree*6]
  double *Hll, *bl;         // [L][9], [L][3]
  double *Dinv, *db;        // [L][9], [L][3]
  const int32_t *blk_i1, *blk_i2, *blk_start;  // non-zero lower blocks of the reduced camera matrix
  const int32_t *pair_k1, *pair_k2;            // per block: (edge of i1, edge of i2) sharing a landmark
  const int32_t* pair_pt;    // [npairs] landmark of the pair (= e_point[pair_k1]): spares the gather kernel a dependent load
  int32_t schur_wide;        // 1: few blocks with long pair lists (a local-BA window) -- k_schur with 8 waves per block instead of 2
  // Landmark-chunk form of the Schur complement (k_schur_lm + k_schur_reduce; an experiment, built and used only under DVM_BA_SCHUR_LM=1):
  // the landmarks, sorted by their first camera, in chunks of <= kSchurLmLandmarks landmarks / <= kSchurLmRows free-camera rows.
  int32_t n_slc, n_slrun;    // chunks; runs (= partial 6x6 blocks) over all chunks
  const int32_t* sl_desc;    // [n_slc][8]: row offset, rows, landmark offset, landmarks, pair offset, pairs, run offset, runs
  const int32_t* sl_row_edge;   // per chunk row: the (local) edge whose W row it is
  const int32_t* sl_row_lm;     // per chunk row: its landmark's slot in the chunk
  const int32_t* sl_lm;         // per chunk landmark slot: the landmark
  const int32_t* sl_pairs;      // per chunk, sorted by block: row1 | row2 << 9 | landmark slot << 18
  const int32_t* sl_runs;       // per run: block, first pair (index inside the chunk); one sentinel per chunk
  double* sl_part;              // [n_slrun][36] partial blocks, summed per block in chunk order by k_schur_reduce
  const int32_t *bp_start, *bp_slots;   // per block: its runs (global run indices), ascending chunk
  double* S;                // [ldS][ldS] dense lower triangle + augmented rhs row
  double* Linv;             // [ldS/64][64*64] inverses of the factored diagonal blocks
  double* ytmp;             // [n_pad + 64] doubles, used as int32 words: [0] ticket, [1 + k] hand-off flag of tile column k
                            // of the one-launch back substitution; must be zeroed once after allocation
  double* x;                // [6*nfree + 3*L]
  double *partial, *partial2;
  // Reduced camera system in TILE space (ba_ordering.h): camera i (elimination order) owns rows
  // ba_row(i) = (i / 10) * 64 + (i % 10) * 6; rows 60..63 of a tile are identity padding; n_pad = 64 * camera tiles;
  // row n_pad carries bschur^T (augmented rhs), so ldS = n_pad + 64.
  int32_t n_pad;
  int32_t per_tile, dof;    // unknown blocks per 64-row tile and their size: (10, 6) cameras, (9, 7) Sim3 vertices
  double* xrow;             // [n_pad] solution in row space (back substitution reads its ancestors here)
  // level schedule of the tile Cholesky (columns of one elimination-tree height are independent):
  const int32_t* cols;      // columns by level                                  (host offsets h_level_off)
  const int32_t* strips;    // (row tile, column) pairs by level                 (h_strip_off)
  const int32_t* targets;   // (ti, tj, c0, c1) trailing tiles by level          (h_tgt_off)
  const int32_t* contrib;   // contributing columns of each target, ascending
  const int32_t* contrib_strip;  // per contribution: indices (into strips) of the two strips it multiplies
  int32_t* strip_flags;     // [4 * strips]: hand-off flag of every 16-row slice of a strip (k_chol_trsm_update); zeroed once after
                            // allocation, compared with the solve's sequence number like the back substitution's flags
  const int32_t* nz_tiles;  // (i, j) pairs of the structurally non-zero tiles (n_nz of them): cleared before every trial
  int32_t n_nz;
  const int32_t* colstrip_off;  // [ntiles+1] device
  const int32_t* colstrips;     // per column: its strip rows
  // KEPT LANDMARKS ("border", round 5).  A map whose reduced camera system is ruined by a handful of landmarks -- a few dozen points seen from
  // places far apart on the trajectory couple camera tiles that nothing else couples: 0.2 % of the landmarks turn the 7-level elimination
  // tree of a 500-keyframe loop into a 37-level chain -- keeps those landmarks OUT of the Schur complement: they stay unknowns of the
  // reduced system, 21 to a 64-row tile (3 x 21 rows + one of padding) behind the camera tiles and in front of the rhs row,
  //     [ S_E  W_K ] [xp  ]   [bp - sum_E W Dinv bl]        S_E: Schur complement over the ELIMINATED landmarks only,
  //     [ W_K' H_K ] [xl_K] = [bl_K                ]        H_K = Hll + lambda I of the kept ones, W_K their Hpl blocks
  // -- the same linear system as g2o's (block_solver.hpp:381-486 eliminates every landmark), another elimination order.  kept_slot[l] =
  // position of landmark l among the kept ones or -1; kept_list the inverse; ncamt = camera tiles (the kept tiles follow).
  const int32_t* kept_slot;     // [L] or null (nkept = 0)
  const int32_t* kept_list;     // [nkept]
  int32_t nkept, ncamt;
  // FLOW form of the solve (k_chol_flow, ba_ordering.h): tile tasks taken through a ticket, factorisation + back substitution in ONE launch
  const int32_t* flow_tasks;    // [n_flow_tasks][8]
  const int32_t* flow_contrib;  // [..][4] flattened contributor lists (ba_ordering.h)
  const int32_t* flow_col;      // [tiles][8] chain links and diagonal-tile modes (ba_ordering.h)
  const int32_t* colstrip_id;   // per colstrips entry: index of that strip
  int32_t* flow_flags;          // [2 x strips] a 32-row half of X published | [tiles] L^-1 of a column | [tiles] T' of a diagonal tile (PRE) | [2 x strips] a
                                // half of a chain strip gathered | xrow tagged | ticket; compared with the solve's sequence number, zeroed once after allocation
  int32_t n_flow_tasks, n_strips_total, n_tiles_total;
  int32_t flow;                 // 1: ba_launch_cholesky_solve uses k_chol_flow (set by dvm_ba_set_problem; 0 after one of its waits timed out)
  int32_t flow_wgs;             // persistent workgroups to launch (compute units of the device)
  const int32_t *h_level_off, *h_strip_off, *h_tgt_off;  // HOST arrays [nlevels+1]
  int32_t nlevels;
  int32_t pair_a, pair_b;   // tile columns of the top pair (ba_ordering.h: BaTileSchedule::pair_a / pair_b) and pair_ok = 1, or pair_ok = 0
  int32_t pair_ok;
  int32_t diag_in_level;    // levels of <= 256 slice workgroups factor their diagonal tiles inside k_chol_trsm_update<true> (0: DVM_BA_NO_DIAG_IN_LEVEL)
  int32_t n_root_raw;       // columns of the last launched level whose only strip is the rhs row: their panel solve (one 64x64
                            // matrix-vector product each) is done by the back substitution itself, no k_chol_trsm launch (0: launch it)
  const double* lambda;     // device scalar with the current LM damping (pose graph), or null: use lambda_v
  double lambda_v;          // LM damping by value (bundle adjustment: no H2D copy per trial)
  double damp_s;            // 1 normally.  Landmark-sharded BA (dvm_ba_set_problem_sharded): every rank builds a PARTIAL reduced
                            // system that is summed over ranks, so what is not a sum over edges -- lambda on the camera
                            // diagonal, the identity padding, the augmented corner, the cameras' lambda x^2 -- is contributed by
                            // rank 0 only (damp_s = 0 elsewhere)
  int32_t shard_rank, shard_world;  // landmark-sharded BA: this rank evaluates the landmarks l with l % shard_world == shard_rank (1 rank: all)
  double *poses_new, *points_new;   // trial state: k_update writes exp(dx) * poses -> poses_new, points + dx -> points_new; an
                                    // accepted trial swaps the pointers on the host, a rejected one leaves (poses, points) alone
                                    // -- g2o's push() / pop() / discardTop() without copies
};

// How a phase-ending kernel hands its scalar to the host without a D2H copy + stream synchronisation: the LAST workgroup to
// finish (arrival counter) sums the per-workgroup partials in the fixed order of k_reduce_sum and stores the result into
// page-locked host memory mapped into the device; with `publish` it then copies the Cholesky failure flag and releases
// `seq` (system scope), which the host is spinning on.
struct BaPublish {
  double* dev_vals;               // device memory [8]: every phase result lands here first (visible to later kernels)
  double* host_vals;              // mapped host memory [8]: the publishing workgroup copies dev_vals[0..5] + the failure flag
                                  // here itself, right before releasing `seq` -- one writer, one ordered stream of stores
  unsigned long long* host_seq;   // mapped host memory
  unsigned long long seq;
  unsigned int* counter;          // device arrival counter (reset by the last workgroup)
  const int* d_fail;              // device: Cholesky failure flag of this trial (copied to host_vals[6] on publish), may be null
  int slot;
  int publish;
  // Speculation on the trial being ACCEPTED (ba_solver.cpp): the workgroup that publishes the trial's chi2 also takes g2o's
  // decision -- rho from chi2 / the gain-ratio denominator (dev_vals[2]) / the failure flag, and the damping an accepted trial
  // hands to the next iteration -- into spec[0] (the next lambda, or -1: rejected) and host_vals[7]; the launches that follow
  // on the stream (k_accum's landmark inverses, the next trial's k_schur) read it instead of waiting for the host.
  double* spec;                   // device [2], or null: no decision on the device
  double cur_chi, lambda;         // the iteration's accepted chi2 and the trial's damping
  int spec_mode;                  // 0: the LM decision described above; 1: spec[0] = 1e-5 * (the reduced max |diag|), computeLambdaInit
  int n_bad;                      // iterations in a row without a 0.1 % improvement so far (the stopping rule of dvm_ba_optimize): an accepted
                                  // trial that completes the third one ends the optimisation -- nothing is enqueued behind it (spec[0] = -2)
};

// Sim3 pose graph (Optimizer::OptimizeEssentialGraph numerics): vertex states + EdgeSim3 list + block structure.
// The reduced system lives in a BaView used as a plain tile system (S, schedule, x): per_tile = 9, dof = 7.
struct PgView {
  int32_t n, E, nfree, fix_scale, nblk;
  double* S;                 // [n][8] vertex estimates Siw (q_xyzw, t, s)
  const int32_t* vidx;       // [n] position in the elimination order or -1 (fixed)
  const int32_t* free_v;     // [nfree] vertex id
  const int32_t* ev;         // [E][2] (vertex 0 = i, vertex 1 = j)
  const double* emeas;       // [E][8] measurement Sji
  double* e_err;             // [E][7]
  double* e_J;               // [E][2][49] numeric Jacobians w.r.t. vertex i / vertex j, row-major [error row][dof]
  const int32_t *blk_a, *blk_b, *blk_start;   // non-zero 7x7 blocks (pos a >= pos b) and their contribution lists
  const int32_t* blk_contrib;                 // edge << 2 | side_a << 1 | side_b
  const int32_t *v_start, *v_contrib;         // per free vertex: incident edges, edge << 1 | side
  double* bp;                // [7 * nfree] gradient b (compact), for computeScale
  double* partial;           // per-block partial sums
};
void pg_launch_edge_eval(hipStream_t s, const PgView& G, bool jac, double* d_scalars, int slot);
void pg_launch_build(hipStream_t s, const PgView& G, const BaView& T);
void pg_launch_update(hipStream_t s, const PgView& G, const BaView& T, double* d_scalars, int slot_scale);

// jac: linearise at (poses, points); !jac: chi2 only, at the trial state (poses_new, points_new)
void ba_launch_edge_eval(hipStream_t s, const BaView& V, bool jac, const BaPublish& pub);
// spec (device, BaPublish::spec) != null: landmarks also get Dinv / db for the damping an accepted trial continues with
// max_pub != null: the launch also reduces max |diag(Hpp), diag(Hll)| and publishes it (computeLambdaInit's input)
void ba_launch_accum(hipStream_t s, const BaView& V, const double* spec = nullptr, const BaPublish* max_pub = nullptr);
// the structurally non-zero tiles of S back to empty (what ba_launch_accum does on its way)
void ba_launch_clear_tiles(hipStream_t s, const BaView& V);
// the Schur complement alone, for a trial whose damping is decided on the device: V.lambda points at BaPublish::spec (a negative
// value there = the trial before was rejected: the launch does nothing); resets the Cholesky failure flag like the prologue
void ba_launch_schur_speculative(hipStream_t s, const BaView& V, int* d_fail);
void ba_launch_max_diag(hipStream_t s, const BaView& V, const BaPublish& pub);
void ba_launch_schur(hipStream_t s, const BaView& V, int* d_fail);
// solve_seq: a number > 0 that differs from call to call on this BaView (the back substitution's hand-off flags compare against it)
void ba_launch_cholesky_solve(hipStream_t s, const BaView& V, int* d_fail, int solve_seq);
// d_fail: the Cholesky failure flag of this trial -- after a failed solve x keeps the last good solution (g2o's _x), which is applied all the same
void ba_launch_backsub_update(hipStream_t s, const BaView& V, const BaPublish& pub, const int* d_fail);
void ba_launch_edge_depth(hipStream_t s, const BaView& V, uint8_t* d_out);
// sharded BA: the structurally non-zero tiles of S (augmented rhs row included) <-> a contiguous buffer of n_nz * 4096 doubles
void ba_launch_pack_tiles(hipStream_t s, const BaView& V, double* buf, bool unpack);
// sharded BA: diag(Hpp) (6 per free camera) <-> buffer; and max(buffer[0..6 nfree), diag Hll of the local landmarks) -> host
void ba_launch_hpp_diag(hipStream_t s, const BaView& V, double* buf);
void ba_launch_max_diag_sharded(hipStream_t s, const BaView& V, const double* hpp_diag, const BaPublish& pub);
// sharded BA: buf[3 l + k] = points[l] for the landmarks this rank owns (l % world == rank -- observed or not: an unobserved landmark
// travels unchanged), 0 elsewhere; and the inverse after the sum
void ba_launch_points_exchange(hipStream_t s, const BaView& V, double* buf, bool scatter);
void ba_launch_optimize_sim3(hipStream_t s, double* S12io, int fix_scale, const double* P1c, const double* P2c,
                             const double* obs1, const double* obs2, const double* w1, const double* w2, int N,
                             const double* K, double th2, uint8_t* inlier, int32_t* nin, double* chi_scratch, uint8_t* flag_scratch);
void ba_launch_sim3_hypotheses(hipStream_t s, const float* P1c, const float* P2c, const float* e1, const float* e2, int N,
                               const float* K, const int32_t* triples, int H, int fix_scale, float* T12, int32_t* nin, uint8_t* mask);
void ba_launch_pose_optimize(hipStream_t s, const double* pose_in, const double* Xw, const double* obs, const double* info,
                             const int32_t* n_per_frame, int stride, int batch, double fx, double fy, double cx, double cy,
                             double* pose_out, uint8_t* outlier, int32_t* n_inliers, double* chi_scratch);

}  // namespace dvm

// csrc/ba_window.hip: dvm_ba_optimize_windows with the choice of normalising the input quaternions (a fresh graph: SE3Quat's
// constructor does) or taking them as they are (the state a previous optimize() of the same graph left)
#include "../../include/dvmslam_hip.h"
int dvm_ba_optimize_windows_impl(int device, const dvm_ba_window* windows, int K, const volatile uint8_t* stop_flag, dvm_ba_stats* stats, bool normalize_input, bool fast, int cluster);
