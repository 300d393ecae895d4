// dvm_slam_amd/csrc/ba_ordering.h -- host-side ordering + symbolic analysis of the reduced camera system.
//
// g2o hands the reduced camera matrix to a sparse Cholesky with a fill-reducing ordering
// (reference Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:59, Eigen::SimplicialLDLT + AMD).  On the GPU the
// matrix is factored as dense 64x64 tiles; what matters there is not fill alone but the LENGTH OF THE DEPENDENCY
// CHAIN: a trajectory-ordered matrix is block-banded and its elimination tree is a path (one tile column after the
// other, 47 dependent steps at 500 keyframes).  This module
//   1. bands the camera graph (Cuthill-McKee), packs whole cameras into tiles (10 cameras = 60 of 64 rows),
//   2. orders the TILES by nested dissection (BFS level-structure separators), which turns the path into a
//      bushy elimination tree,
//   3. runs the symbolic tile factorisation and groups the tile columns by elimination-tree height: all columns of
//      one height are independent and are factored by ONE launch.
// Pure host code, no HIP: unit-tested on the CPU (tests/test_host_logic.py via tools/check_ba_ordering.cpp).
#pragma once
#include <cstdint>
#include <vector>

namespace dvm {

constexpr int kCamsPerTile = 10;   // BA: 6 * 10 = 60 rows of a 64-row tile; rows 60..63 are identity padding
constexpr int kSim3PerTile = 9;    // pose graph: 7 * 9 = 63 rows

// adj[a] = cameras sharing a landmark with camera a (any order, may contain duplicates / a itself).
// Returns pos[a] = position of camera a in the elimination order (a permutation of 0..n-1).
std::vector<int> ba_order_cameras(const std::vector<std::vector<int>>& adj, int per_tile = kCamsPerTile);

struct BaTileSchedule {
  int ntiles = 0;                      // tiles of the factor, the last one holds the augmented rhs row
  int nlevels = 0;
  std::vector<int32_t> level_off;      // [nlevels+1] -> cols
  std::vector<int32_t> cols;           // tile columns grouped by elimination-tree height (leaves first)
  std::vector<int32_t> strip_off;      // [nlevels+1] -> strips (pairs)
  std::vector<int32_t> strips;         // (row tile i, column k): L(i,k) != 0, i > k, k in the level
  std::vector<int32_t> tgt_off;        // [nlevels+1] -> targets (quads)
  std::vector<int32_t> targets;        // (ti, tj, c0, c1): trailing tile ti >= tj updated by contrib[c0..c1)
  std::vector<int32_t> contrib;        // column k of each contribution, ascending within a target (deterministic sum)
  std::vector<int32_t> contrib_strip;  // per contribution: indices (into strips) of the strips (ti, k) and (tj, k) it multiplies
  std::vector<int32_t> colstrip_off;   // [ntiles+1] -> colstrips
  std::vector<int32_t> colstrips;      // per column k: row tiles i > k with L(i,k) != 0 (back substitution)
  std::vector<int32_t> nz_tiles;       // (i, j), i >= j: every structurally non-zero tile of the factor (what a trial must clear)
  int root_level = -1;                 // the last level that gets launches (the level of the rhs tile alone behind it is skipped)
  int n_root_raw = 0;                  // its columns, if nothing but the rhs row hangs below them and they update nothing: their panel
                                       // solve is one matrix-vector product each, done by the back substitution (0: launched as usual)
  int pair_a = -1, pair_b = -1;        // the last two camera tile columns when they form a chain of their own: the root column b (only the rhs
                                       // row below it) and column a alone on the level before it with nothing but (b, a) and the rhs row below --
                                       // one workgroup factorises both and solves for their unknowns (k_chol_pair); -1: no such pair
  double fill = 1.0;                   // non-zero tiles / all lower tiles
  // FLOW form (k_chol_flow): factorisation + back substitution as ONE persistent launch of tile tasks taken through a ticket.
  //   SLICE (kind 1): a 32-row half of a strip tile (i, k) gathers its updates LEFT-LOOKING -- tgt = S, then per level that updates the
  //     tile acc = sum over that level's columns m of X(i,m) X(k,m)^T and tgt -= acc: the order and association of the level launches,
  //     same bits --, waits for L_k^-1 and publishes X = T L_k^-T in place;  TSLICE (kind 4): the same for a CHAIN strip, gather only;
  //   PRE (kind 2): a diagonal tile (p, p) updated by more than one level -- all levels but the last one, T' in place;
  //   DIAG (kind 0): a leaf's diagonal tile: factorisation, L^-1 published -- and then UP THE TREE along the chain (flow_col): the
  //     workgroup solves the chain strip (p, k) with the L_k^-1 it holds, publishes X(p,k), gathers p's other children, adds X X^T, factorises p ...
  // A ticketed task only waits for tasks BEFORE it in the list, except the chain, which waits for the TSLICE / PRE tasks of the columns
  // it climbs to: those are drawn by the other workgroups (the launcher keeps the chains -- one per leaf -- far below the workgroup count).
  std::vector<int32_t> flow_tasks;     // 8 ints per task: kind, ti, tj, half, c0, c1 (flow_contrib entries), own strip index or -1, PRE: the tile's mode
  std::vector<int32_t> flow_contrib;   // 4 ints per entry: column m, strip (ti, m), strip (tj, m), 1 = last contributor of its level (tgt -= acc behind it)
  std::vector<int32_t> flow_col;       // 8 ints per tile column: chain parent or -1, chain strip, mode of the column's own diagonal tile, the other
                                       // contributors of its last level as a range of flow_contrib, 0 0 0 (see ba_ordering.cpp)
  int flow_leaves = 0;                 // columns without contributors = ticketed DIAG tasks = chains
  std::vector<int32_t> colstrip_id;    // per `colstrips` entry: that strip's index in `strips`
};
// T[i][j] (i >= j) = structurally non-zero tile of the matrix in elimination order; the last tile row (rhs) is dense.
BaTileSchedule ba_tile_schedule(std::vector<std::vector<char>> T);

}  // namespace dvm
