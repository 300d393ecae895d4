// dvm_slam_amd/csrc/ba_window.hip -- dvm_ba_optimize_windows: K independent bundle adjustments ("windows"), ONE launch, one workgroup per
// window, the whole optimizer.optimize(n) -- Levenberg-Marquardt iterations, trials, stopping rules -- on the device.
//
// What it is for.
//  (1) Several agents sharing one GPU (BASELINE.json config 4 with more agents than GPUs): their LocalBundleAdjustment windows
//      (Optimizer.cc:1030-1387) are independent problems; K of them fill K compute units side by side instead of queueing behind one
//      another on the 40-launches-per-trial tile solver (ba_solver.cpp), which is laid out for one large map.
//  (2) Tiny problems -- GlobalBundleAdjustemnt(map, 20) on the TWO keyframes of a monocular initialisation (Tracking.cc:2330), local
//      windows of 3..5 keyframes right after it.  One free camera with free points leaves the scale gauge to the damping alone; the
//      result of such a problem moves by 1e-3 .. 1e-1 when nothing but the ORDER of the floating-point sums changes (tools/
//      ba_sensitivity.py: the CPU oracle against itself with its edge list permuted).  "Within 1e-6 of the reference" is then only
//      meaningful for an implementation that adds in the reference's order -- so this kernel does:
//
// Every sum runs in the order g2o's single-threaded code runs it, and every rounding is one IEEE operation (-ffp-contract=off, IEEE
// division and square root, the libm calls replaced by csrc/f64_spec.h on both sides):
//   * chi2 = sum of rho(e) over the edges in edge order (SparseOptimizer::activeRobustChi2), by ONE wave, sequentially;
//   * Hpp / bp of a camera, Hll / bl of a landmark: contributions in edge order (BaseBinaryEdge::constructQuadraticForm called
//     edge by edge, block_solver.hpp:502-560) -- one lane per matrix entry walks the vertex's edges in order;
//   * Schur complement (block_solver.hpp:381-439): landmark by landmark, Hschur(i1, i2) -= W1 Dinv W2^T in landmark order -- one lane
//     per entry walks the block's (edge, edge) pairs in that order; bschur likewise;
//   * the reduced system: Cholesky by rows with ascending-k dot products (the oracle's envelope form; a right-looking elimination
//     applies the same subtractions to every entry in the same order, so it runs in parallel and gives the same bits), forward and
//     backward substitution likewise; LDS-resident (packed lower triangle: up to 30 free cameras = 130 KB of the CU's 160 KB);
//   * landmark back substitution, oplus (SE3Quat::exp, se3quat.h:212-240), computeScale (sequential), the gain ratio and the damping
//     update of optimization_algorithm_levenberg.cpp:107-147, the stopping rules of Optimizer / g2o (:154-162).
// tests/test_gpu_ba_window.py holds the result to the CPU oracle BIT FOR BIT (poses, points, per-edge chi2, LM trial sequence, lambda).
//
// The price is speed per window -- the sequential sums are one lane's dependent FP64 additions (8 cycles each) -- which K windows side
// by side buy back; one large window is the tile solver's job (dvm_ba_optimize).
//
// dvm_ba_optimize_windows_fast (k_ba_window_cluster, second half of this file) is the same optimizer for MANY LocalBundleAdjustment
// windows per call (Optimizer.cc:1030-1387, one per agent): tree sums in a fixed order instead of the sequential ones, edges sorted
// camera-major with structure-of-arrays rows, a cluster of G = 1 / 2 / 4 / 8 workgroups per window separated by agent-scope phase
// barriers, the reduced system solved in workgroup 0's LDS.  Deterministic, independent of G; equal to the oracle to the general
// solver's contract (same LM trial sequence, 1e-6), not bit for bit (tests/test_gpu_ba_window_fast.py, tools/soak_r06.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dvmslam_hip.h"
#include "f64_spec.h"
#include "ba_kernels.h"
#include "group_commit.h"
#include "host_stage.h"
#include "orb_pipeline.h"   // set_error / hip_check / DVM_HIP

namespace dvm {

constexpr int kWinThreads = 512;
constexpr int kWinMaxFree = 30;          // 6 * 30 = 180 rows: packed lower triangle 130 320 B of LDS
constexpr int kRowB = 16, kRowA = 12, kRowW = 18, kRowD = 24, kRowH = 12;   // doubles per row of the per-edge / per-landmark arrays below
constexpr int kWinIntCap = 6 * kWinThreads;   // ints of one chunk's index block (six per thread in flight)

// device view of one window: pointers into the call's staging block.
// Per-edge rows live in the order their consumer walks them, so that a chunk of consecutive rows is one coalesced copy into LDS:
//   rowB [E][16]  edge order            B (2x6), w, wr0, wr1         -> Hpp / bp chains of the cameras (edge order)
//   rowA [E][12]  landmark-major (lpos) A (2x3), w, wr0, wr1         -> Hll / bl of a landmark: its rows are contiguous
//   rowW [F][18]  landmark-major over the F edges of FREE cameras (fpos): W = w B^T A
// (W Dinv and W Dinv bl of a trial never reach global memory: the Schur pass forms them in LDS from the staged W rows.)
struct BaWin {
  int32_t P, L, E, F, nfree, nact, nblk, iterations;
  int32_t C, C2, n_sc, n_hc;          // rows per Schur chunk / per Hessian chunk, number of chunks
  int32_t stage_doubles, pad0;        // LDS doubles of the staging area this window needs
  double fx, fy, cx, cy, delta;
  double *poses, *poses_t;            // [P][7] accepted / trial state
  double *pts, *pts_t;                // [L][3]
  double *out_poses, *out_pts;        // the accepted state at the end (what the host fetches)
  const int32_t *pidx, *lidx;         // [P] free index or -1; [L] active index or -1
  const int32_t *free_pose, *act_pt;  // [nfree], [nact]
  const int32_t *e_pose, *e_point;    // [E]
  const double *e_obs, *e_info;       // [E][2], [E]
  const int32_t *lpos, *fpos;         // [E] row of the edge in landmark-major order over all edges / over free-camera edges (-1)
  const int32_t *pt_start, *f_start;  // [nact + 1] a landmark's rows in rowA / in rowW
  const int32_t *f_cam;               // [F] free camera index of the row
  double *rowB, *rowA, *rowW;
  double *e_chi2, *e_rho;             // [E] chi2 / rho(chi2) of the last evaluation, edge order
  uint8_t* e_depth;                   // [E] isDepthPositive() at the final state
  // Hessian chunks (edge order, C2 rows each), one fixed-size int block per chunk:
  // [per camera the range of its rows in the list (nfree + 1) | the chunk's free-camera row slots, camera-major (C2)]
  const int32_t *hc_ints;             // [n_hc][nfree + 1 + C2]
  // Schur chunks (whole landmarks, <= C free rows, <= C landmarks): descriptor {first row, rows, int offset, int length, runs, pairs, first
  // landmark, landmarks}; int block = [per camera the range of its rows (nfree + 1) | row slots camera-major (rows) | landmark of every row,
  // relative (rows) | runs: i1 | i2 << 8 | first pair << 16, + sentinel | pairs: slot1 | slot2 << 16, sorted by block]
  const int32_t *sc_desc, *sc_ints;   // [n_sc][8]
  const int32_t *blk_ij;              // [nblk] i1 | i2 << 8
  double *Hpp, *bp, *HB, *DD, *x, *terms;   // [nfree][36], [6 nfree], [nact][12] = Hll (9) bl (3), [nact][12] = Dinv (9) Dinv bl (3), [6 nfree + 3 nact] x 2
  dvm_ba_stats* stats;
  unsigned long long* prof;           // [16] shader-clock cycles per phase, accumulated by thread 0 (DVM_BA_WINDOW_PROF=1), or null
  // ---- the fast (cluster) form only (k_ba_window_cluster: tree sums in a fixed order instead of g2o's sequential ones, nothing streamed through LDS).
  // Its edges are SORTED camera-major on the host -- free cameras first, a camera's edges by landmark, the fixed cameras' edges behind
  // them -- and the per-edge rows are structure-of-arrays with pitch Fp / Ep, so that a lane per edge reads and writes consecutive words.
  // e_pose / e_point / e_obs / e_info above are in that order; the first F edges are the free cameras' (the rows of W / T / B).
  int32_t Fp, Ep;                         // pitches: F and E rounded up to 64
  const int32_t *e_orig;                  // [E] the edge's index in the caller's list (chi2 / depth go back there)
  const int32_t *e_lm;                    // [E] active landmark of the edge
  const int32_t *cam_start;               // [nfree + 1] a free camera's edges
  const int32_t *pt_edges;                // [E] a landmark's edges (sorted indices, ascending: its free-camera rows come first), ranges pt_start
  const int32_t *bp_start;                // [nblk + 1] a block's (row of camera i1, row of camera i2) pairs, landmark order
  const int2* bp_pairs;
  const int32_t *sched_start, *sched_task;   // the Schur pass's tasks per wave of the cluster (8 G waves): task < nfree: rhs of that camera, else block task - nfree
  double *Bs, *Ws, *Ts, *Cs;              // [15][Fp] B (12) w wr0 wr1 | [18][Fp] W | [24][Fp] W Dinv (18), W Dinv bl (6) | [3][Fp] W^T x_p
  double *chi_s;                          // [E] chi2 of the last evaluation, sorted order
  // ---- the CLUSTER form (k_ba_window_cluster: G workgroups per window): what the workgroups hand each other through global memory
  unsigned int *cl_ctr, *cl_tmo;          // arrival counter of the window's barriers, time-out word (both zeroed before every launch)
  double *Sblk, *rhsg;                    // [nblk][36] the blocks of the reduced system as their owner waves leave them, [6 nfree]
  double *cl_part;                        // [4][8] partial sums of the eight fixed parts: chi2, scale terms, max |diag|
  double *cl_ctl;                         // [4] 0: the solve succeeded, 1: the stop word as the leader saw it
};

// ------------------------------------------------------------------------------------------------ small algebra (the oracle's sequences)
__device__ __forceinline__ void w_quat_to_R(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void w_R_to_quat(const double* R, double* q) {      // Eigen's quaternion-from-matrix
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
}
__device__ __forceinline__ void w_quat_normalize(double* q) {      // SE3Quat::normalizeRotation, se3quat.h:261-266
  if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
__device__ __forceinline__ void w_mat3_vec(const double* R, const double* v, double* o) {
  o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}
// T' = exp(u) * T, u = (omega, upsilon): SE3Quat::exp + operator* + normalizeRotation with the libm calls taken from f64_spec.h
__device__ void w_se3_oplus(const double* T, const double* u, double* Tn) {
  const double om0 = u[0], om1 = u[1], om2 = u[2];
  const double theta = sqrt(om0 * om0 + om1 * om1 + om2 * om2);
  const double O[9] = {0, -om2, om1, om2, 0, -om0, -om1, om0, 0};
  double O2[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) O2[3 * r + c] = O[3 * r] * O[c] + O[3 * r + 1] * O[3 + c] + O[3 * r + 2] * O[6 + c];
  double R[9], Vm[9];
  if (theta < 0.00001) {
#pragma unroll
    for (int k = 0; k < 9; k++) { R[k] = ((k % 4 == 0) ? 1.0 : 0.0) + O[k] + O2[k]; Vm[k] = R[k]; }
  } else {
    const double sn = f64_sin(theta), cs = f64_cos(theta);
    const double a = sn / theta, bb = (1 - cs) / (theta * theta), c = (theta - sn) / f64_cube(theta);
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const double I = (k % 4 == 0) ? 1.0 : 0.0;
      R[k] = I + a * O[k] + bb * O2[k];
      Vm[k] = I + bb * O[k] + c * O2[k];
    }
  }
  double dq[4], dt[3], Rd[9], rt[3], nq[4];
  w_R_to_quat(R, dq);
  w_quat_normalize(dq);
  w_mat3_vec(Vm, u + 3, dt);
  w_quat_to_R(dq, Rd);
  w_mat3_vec(Rd, T, rt);
  const double* q = T + 3;
  nq[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  nq[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  nq[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  nq[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  w_quat_normalize(nq);
  Tn[0] = dt[0] + rt[0]; Tn[1] = dt[1] + rt[1]; Tn[2] = dt[2] + rt[2];
  Tn[3] = nq[0]; Tn[4] = nq[1]; Tn[5] = nq[2]; Tn[6] = nq[3];
}
__device__ __forceinline__ void w_robustify(double e, double delta, double& rho0, double& rho1) {   // robust_kernel_impl.cpp:68-81
  if (delta <= 0 || e <= delta * delta) { rho0 = e; rho1 = 1.; }
  else { const double s = sqrt(e); rho0 = 2 * s * delta - delta * delta; rho1 = delta / s; }
}
__device__ __forceinline__ void w_inv3(const double* M, double* Inv) {
  const double a = M[0], b = M[1], c = M[2], d = M[3], e = M[4], f = M[5], g = M[6], h = M[7], i = M[8];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  const double id = 1.0 / det;
  Inv[0] = (e * i - f * h) * id; Inv[1] = (c * h - b * i) * id; Inv[2] = (b * f - c * e) * id;
  Inv[3] = (f * g - d * i) * id; Inv[4] = (a * i - c * g) * id; Inv[5] = (c * d - a * f) * id;
  Inv[6] = (d * h - e * g) * id; Inv[7] = (b * g - a * h) * id; Inv[8] = (a * e - b * d) * id;
}

// ------------------------------------------------------------------------------------------------ the sequential sum
// sum of v[0..n) in index order, s = ((0 + v0) + v1) + ..., by ONE wave: 64 values at a time go lane -> LDS, every lane then adds them
// with uniform-address (broadcast) reads -- the chain of dependent v_add_f64 is the cost (8 cycles per value), the next 64 values
// are in flight meanwhile.  Padding a short last chunk with +0.0 is exact (s + 0.0 == s; s is never -0.0: it starts as +0.0).
__device__ double wave_sequential_sum(const double* __restrict__ v, int n, double* __restrict__ buf /* 128 doubles of LDS, this wave's */) {
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  double mine = lane < n ? v[lane] : 0.0;
  for (int base = 0, c = 0; base < n; base += 64, c ^= 1) {
    double* b = buf + 64 * c;
    b[lane] = mine;
    const int nx = base + 64 + lane;
    mine = nx < n ? v[nx] : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // 16 values at a time into registers, the next 16 requested before the current 16 are added: left to itself the compiler waits
    // for every ds_read right in front of its add (~26 cycles per value instead of the add's own 8)
    double t[2][16];
#pragma unroll
    for (int j = 0; j < 16; j++) t[0][j] = b[j];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (g < 3) {
#pragma unroll
        for (int j = 0; j < 16; j++) t[(g + 1) & 1][j] = b[16 * (g + 1) + j];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 16; j++) s += t[g & 1][j];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  return s;
}

// ------------------------------------------------------------------------------------------------ edge pass
// JAC = false: computeActiveErrors -- chi2 and rho of every edge at the state (poses, pts).  JAC = true: additionally linearizeOplus +
// the edge's part of constructQuadraticForm: A (2x3), B (2x6), w = rho' Omega, wr = -Omega e rho', W = w B^T A, each into the row its
// consumer will stream.
template <bool JAC>
__device__ void win_edge_pass(const BaWin& W, const double* __restrict__ poses, const double* __restrict__ pts) {
  for (int k = threadIdx.x; k < W.E; k += kWinThreads) {
    const int p = W.e_pose[k], l = W.e_point[k];
    const double* T = poses + 7 * (size_t)p;
    const double* X = pts + 3 * (size_t)l;
    double R[9], Xc[3];
    w_quat_to_R(T + 3, R);
    w_mat3_vec(R, X, Xc);
    Xc[0] += T[0]; Xc[1] += T[1]; Xc[2] += T[2];
    const double x = Xc[0], y = Xc[1], z = Xc[2];
    const double info = W.e_info[k];
    const double e0 = W.e_obs[2 * k] - (W.fx * x / z + W.cx);
    const double e1 = W.e_obs[2 * k + 1] - (W.fy * y / z + W.cy);
    const double chi2 = e0 * info * e0 + e1 * info * e1;
    double r0, r1;
    w_robustify(chi2, W.delta, r0, r1);
    W.e_chi2[k] = chi2;
    W.e_rho[k] = r0;
    if (!JAC) continue;
    const double J[6] = {-(W.fx / z), 0, W.fx * x / (z * z), 0, -(W.fy / z), W.fy * y / (z * z)};
    double A[6], B[12];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) A[3 * r + c] = J[3 * r] * R[c] + J[3 * r + 1] * R[3 + c] + J[3 * r + 2] * R[6 + c];
    const double S[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int c = 0; c < 6; c++) B[6 * r + c] = J[3 * r] * S[c] + J[3 * r + 1] * S[6 + c] + J[3 * r + 2] * S[12 + c];
    const double w = r1 * info;
    const double wr0 = -info * e0 * r1, wr1 = -info * e1 * r1;
    double* oB = W.rowB + kRowB * (size_t)k;
    double* oA = W.rowA + kRowA * (size_t)W.lpos[k];
#pragma unroll
    for (int i = 0; i < 12; i++) oB[i] = B[i];
    oB[12] = w; oB[13] = wr0; oB[14] = wr1; oB[15] = 1.0;       // (slot 15: the "weight" of a bp chain step, see win_accumulate_cameras)
#pragma unroll
    for (int i = 0; i < 6; i++) oA[i] = A[i];
    oA[6] = w; oA[7] = wr0; oA[8] = wr1;
    const int fp = W.fpos[k];
    if (fp >= 0) {
      double* oW = W.rowW + kRowW * (size_t)fp;
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) oW[3 * a + b] = w * (B[a] * A[b] + B[6 + a] * A[3 + b]);
    }
  }
}

// A chunk of a streamed pass travels global memory -> registers -> LDS: every thread fetches its share of the NEXT chunk (fixed
// stride, a handful of values) before the current one is consumed, so the round trip to L2 runs under the chain steps instead of in
// front of them (a copy loop per chunk cost ~6 us of dependent latencies: descriptor -> rows -> LDS -> barrier).
// barrier for data exchanged through LDS only: __syncthreads() also waits for every global access in flight (s_waitcnt vmcnt(0)), i.e.
// for the NEXT chunk's rows that have just been requested -- which is the round trip the register pipeline exists to hide
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int ND, int NI>
struct ChunkRegs {
  double d[ND];
  int32_t i[NI];
  __device__ __forceinline__ void load_d(int slot0, const double* __restrict__ src, int n) {     // values slot0, slot0 + 1, ... of this thread
#pragma unroll
    for (int u = 0; u < ND; u++) if (u >= slot0) { const int idx = threadIdx.x + (u - slot0) * kWinThreads; d[u] = idx < n ? src[idx] : 0.0; }
  }
  __device__ __forceinline__ void load_i(const int32_t* __restrict__ src, int n) {
#pragma unroll
    for (int u = 0; u < NI; u++) { const int idx = threadIdx.x + u * kWinThreads; i[u] = idx < n ? src[idx] : 0; }
  }
};
template <int N0, int N1, int ND, int NI>
__device__ __forceinline__ void store_d(const ChunkRegs<ND, NI>& r, double* __restrict__ dst, int n) {   // registers N0 .. N1 - 1 -> dst
#pragma unroll
  for (int u = N0; u < N1; u++) { const int idx = threadIdx.x + (u - N0) * kWinThreads; if (idx < n) dst[idx] = r.d[u]; }
}
template <int ND, int NI>
__device__ __forceinline__ void store_i(const ChunkRegs<ND, NI>& r, int32_t* __restrict__ dst, int n) {
#pragma unroll
  for (int u = 0; u < NI; u++) { const int idx = threadIdx.x + u * kWinThreads; if (idx < n) dst[idx] = r.i[u]; }
}

// Hpp / bp of the free cameras: every entry is ONE lane's chain over the camera's edges in edge order (BaseBinaryEdge::
// constructQuadraticForm is called edge by edge).  The B rows stream through LDS C2 edges at a time, the chunk's rows listed per
// camera (host tables), so that a chain step is a few LDS reads.  21 lower (a, b) entries + 6 of bp = 27 lanes per camera, at most two
// (camera, entry) pairs per thread (27 * 30 <= 2 * 512).
__device__ void win_accumulate_cameras(const BaWin& W, double* stage) {
  const int tid = threadIdx.x, nlanes = 27 * W.nfree;
  double* rows = stage;                                               // [C2][16]
  int32_t* ints = reinterpret_cast<int32_t*>(rows + (size_t)W.C2 * kRowB);   // [nfree + 1] ranges | [<= C2] row slots, camera-major
  double acc[2] = {0.0, 0.0};
  int cam[2], ea[2], eb[2];
  // one form for both kinds of chain: acc += row[ow] * (row[o1] * row[o2] + row[o3] * row[o4]) -- an Hpp entry (a, b): w * (B_a B_b + B_6+a B_6+b);
  // a bp entry a: 1.0 * (B_a wr0 + B_6+a wr1), and 1.0 * x is x
  int o1[2], o2[2], o3[2], o4[2], ow[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int t = tid + u * kWinThreads;
    cam[u] = t < nlanes ? t / 27 : -1;
    const int e = t - 27 * (t / 27);
    int a, b;
    if (e < 21) { a = 0; int r = e; while (r > a) { r -= a + 1; a++; } b = r; }   // e = a (a + 1) / 2 + b, b <= a
    else { a = e - 21; b = -1; }                                                   // bp(a)
    ea[u] = a; eb[u] = b;
    o1[u] = a; o3[u] = 6 + a;
    if (b >= 0) { o2[u] = b; o4[u] = 6 + b; ow[u] = 12; } else { o2[u] = 13; o4[u] = 14; ow[u] = 15; }
  }
  if (W.n_hc > 0) {
    // chunk c = edges [c C2, c C2 + C2) and the int block at c * (nfree + 1 + C2): nothing to look up before its loads can go out
    const int istride = W.nfree + 1 + W.C2;
    ChunkRegs<8, 1> pre;
    pre.load_d(0, W.rowB, min(W.C2, W.E) * kRowB);
    pre.load_i(W.hc_ints, istride);
    for (int c = 0; c < W.n_hc; c++) {
      const int nr = min(W.C2, W.E - c * W.C2);
      lds_barrier();                                                  // the previous chunk has been consumed
      store_d<0, 8>(pre, rows, nr * kRowB);
      store_i(pre, ints, istride);
      if (c + 1 < W.n_hc) {
        const int k1 = (c + 1) * W.C2;
        pre.load_d(0, W.rowB + kRowB * (size_t)k1, min(W.C2, W.E - k1) * kRowB);
        pre.load_i(W.hc_ints + (size_t)(c + 1) * istride, istride);
      }
      lds_barrier();
      const int32_t* range = ints;
      const int32_t* list = ints + W.nfree + 1;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (cam[u] < 0) continue;
        double sacc = acc[u];
        const int q1 = range[cam[u] + 1];
        // four steps at a time: their operands are fetched together, then the four dependent additions; a step beyond the camera's
        // range adds +0.0, which changes nothing (the accumulator is never -0.0)
        for (int q = range[cam[u]]; q < q1; q += 4) {
          double term[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const bool in = q + j < q1;
            const double* B = rows + kRowB * (in ? list[q + j] : 0);
            const double v = B[ow[u]] * (B[o1[u]] * B[o2[u]] + B[o3[u]] * B[o4[u]]);
            term[j] = in ? v : 0.0;
          }
#pragma unroll
          for (int j = 0; j < 4; j++) sacc += term[j];
        }
        acc[u] = sacc;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 2; u++) {
    if (cam[u] < 0) continue;
    const int i = cam[u], a = ea[u], b = eb[u];
    if (b >= 0) { W.Hpp[36 * (size_t)i + 6 * a + b] = acc[u]; W.Hpp[36 * (size_t)i + 6 * b + a] = acc[u]; }   // (a product commutes: the mirrored entry has the same bits)
    else W.bp[6 * (size_t)i + a] = acc[u];
  }
}

// Hll / bl of the active landmarks: a thread per landmark, its rows (edge order) are contiguous in rowA
__device__ void win_accumulate_landmarks(const BaWin& W) {
  for (int li = threadIdx.x; li < W.nact; li += kWinThreads) {
    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int q = W.pt_start[li]; q < W.pt_start[li + 1]; q++) {
      const double* A = W.rowA + kRowA * (size_t)q;
      const double w = A[6], wr0 = A[7], wr1 = A[8];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        g[a] += A[a] * wr0 + A[3 + a] * wr1;
#pragma unroll
        for (int b = 0; b < 3; b++) h[3 * a + b] += w * (A[a] * A[b] + A[3 + a] * A[3 + b]);
      }
    }
    double* o = W.HB + kRowH * (size_t)li;
#pragma unroll
    for (int i = 0; i < 9; i++) o[i] = h[i];
#pragma unroll
    for (int i = 0; i < 3; i++) o[9 + i] = g[i];
  }
}

// packed lower triangle: row i starts at i (i + 1) / 2
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// Cholesky of the reduced system in LDS (packed lower triangle), in place; diag receives L_kk.  Entry (i, j) receives its subtractions
// L(i, k) L(j, k) in ascending k, as the row-wise dot products of the envelope factorisation apply them; L(i, j) = s / L(j, j) by IEEE
// division, L(j, j) = sqrt(s).  Six columns (one camera) at a time -- two barriers per camera instead of two per column: every thread
// factorises the 6 x 6 diagonal block for itself (the same 21 words, the same operations: the same bits in every thread), then its
// rows' six entries of the panel, column by column; the trailing entries then take the six columns' subtractions in ascending k.
// Every entry sees the operations of the column-by-column form in the same order: identical bits.  The panel also goes to Lp
// (n x 7 doubles: an odd pitch in 8-byte words keeps the sixteen rows a half-wave reads apart on different LDS banks; read out of the
// packed triangle, whose rows start at i (i + 1) / 2, the trailing update spent most of its time in bank conflicts).
// (The sequential-order kernel's solve.  The cluster form, whose contract is a tolerance, has its own: win_cholesky_rhs below.)
__device__ bool win_cholesky(double* __restrict__ S, double* __restrict__ diag, double* __restrict__ Lp, int n) {
  const int tid = threadIdx.x;
  constexpr int NT = kWinThreads;
  bool ok = true;
  for (int k0 = 0; k0 < n && ok; k0 += 6) {
    double Ld[21], dg[6];    // the diagonal block's factor (lower, packed) and its L_kk
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
      for (int c = 0; c <= a; c++) {
        double v = S[tri(k0 + a, k0 + c)];
#pragma unroll
        for (int k = 0; k < c; k++) v -= Ld[a * (a + 1) / 2 + k] * Ld[c * (c + 1) / 2 + k];
        if (a == c) {
          if (!(v > 0)) ok = false;              // uniform: every thread computes the same word
          dg[a] = sqrt(v);
          Ld[a * (a + 1) / 2 + a] = v;
        } else {
          Ld[a * (a + 1) / 2 + c] = v / dg[c];
        }
      }
    }
    if (!ok) break;
    // (no barrier here: the panel reads rows below the block and writes them and Lp, which the previous step's update has finished with
    //  behind its closing barrier; the block's own factor is written once everybody has read the block -- behind the panel's barrier)
    for (int i = k0 + 6 + tid; i < n; i += NT) {
      double* row = S + tri(i, k0);
      double l[6];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double v = row[c];
#pragma unroll
        for (int k = 0; k < c; k++) v -= l[k] * Ld[c * (c + 1) / 2 + k];
        l[c] = v / dg[c];
      }
#pragma unroll
      for (int c = 0; c < 6; c++) { row[c] = l[c]; Lp[7 * i + c] = l[c]; }
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
      for (int a = 0; a < 6; a++) {
        diag[k0 + a] = dg[a];
#pragma unroll
        for (int c = 0; c < a; c++) S[tri(k0 + a, k0 + c)] = Ld[a * (a + 1) / 2 + c];
      }
    }
    const int tx = tid & 15, ty = tid >> 4;
    for (int i = k0 + 6 + ty; i < n; i += NT / 16) {
      const double* ri = Lp + 7 * i;
      const double li0 = ri[0], li1 = ri[1], li2 = ri[2], li3 = ri[3], li4 = ri[4], li5 = ri[5];
      double* Si = S + tri(i, 0);
      for (int j = k0 + 6 + tx; j <= i; j += 16) {
        const double* rj = Lp + 7 * j;
        double v = Si[j];
        v -= li0 * rj[0]; v -= li1 * rj[1]; v -= li2 * rj[2]; v -= li3 * rj[3]; v -= li4 * rj[4]; v -= li5 * rj[5];
        Si[j] = v;
      }
    }
    __syncthreads();
  }
  return ok;
}
// forward substitution: y(i) = (b(i) - sum_{j < i} L(i, j) y(j)) / L(i, i), the subtractions in ascending j; then backward:
// x(i) /= L(i, i); x(j) -= L(i, j) x(i) for j < i, i descending.  ONE wave (the caller's wave 0), a camera's six unknowns per step:
// every lane forms the six values for itself, then applies them to its rows in the order of the one-at-a-time form.
__device__ void win_substitute(const double* __restrict__ S, const double* __restrict__ diag, double* __restrict__ rhs, int n) {
  const int lane = threadIdx.x & 63;
  for (int k0 = 0; k0 < n; k0 += 6) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double y[6];
#pragma unroll
    for (int a = 0; a < 6; a++) {
      double v = rhs[k0 + a];
#pragma unroll
      for (int c = 0; c < a; c++) v -= S[tri(k0 + a, k0 + c)] * y[c];
      y[a] = v / diag[k0 + a];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 6) { double yv = y[0]; yv = lane == 1 ? y[1] : yv; yv = lane == 2 ? y[2] : yv; yv = lane == 3 ? y[3] : yv; yv = lane == 4 ? y[4] : yv; yv = lane == 5 ? y[5] : yv; rhs[k0 + lane] = yv; }
    for (int r = k0 + 6 + lane; r < n; r += 64) {
      const double* rr = S + tri(r, k0);
      double v = rhs[r];
#pragma unroll
      for (int a = 0; a < 6; a++) v -= rr[a] * y[a];
      rhs[r] = v;
    }
  }
  for (int k0 = n - 6; k0 >= 0; k0 -= 6) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double x6[6];
#pragma unroll
    for (int a = 5; a >= 0; a--) {
      double v = rhs[k0 + a];
#pragma unroll
      for (int c = 5; c > a; c--) v -= S[tri(k0 + c, k0 + a)] * x6[c];
      x6[a] = v / diag[k0 + a];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 6) { double xv = x6[0]; xv = lane == 1 ? x6[1] : xv; xv = lane == 2 ? x6[2] : xv; xv = lane == 3 ? x6[3] : xv; xv = lane == 4 ? x6[4] : xv; xv = lane == 5 ? x6[5] : xv; rhs[k0 + lane] = xv; }
    for (int j = lane; j < k0; j += 64) {
      double v = rhs[j];
#pragma unroll
      for (int a = 5; a >= 0; a--) v -= S[tri(k0 + a, j)] * x6[a];
      rhs[j] = v;
    }
  }
}

// The cluster form's solve (its contract is a tolerance, not the column-by-column bits): the same six-columns-a-step factorisation with
//   * the right-hand side as row n of the matrix -- it sits behind the packed triangle, i.e. AT tri(n, 0): its panel entries ARE the forward
//     substitution's y, its trailing entries take the same subtractions as any other row -- no separate forward pass (one wave, 20
//     dependent steps: 16 us of a 310 us iteration);
//   * the 6 x 6 diagonal block factorised only by the waves that own panel rows (every thread of a wave for itself, as before) and by wave 0
//     -- eight waves doing it side by side shared four SIMDs: 1 800 cycles a step instead of 600 --, fused multiply-adds throughout;
//   * the trailing update on the FP64 matrix cores: A(i, j) -= sum_k L(i, k) L(j, k), k = 6 padded to 8, as two v_mfma_f64_16x16x4 per
//     16 x 16 tile of the lower triangle (rows from the step's first trailing row), tiles dealt round-robin to the eight waves.  The panel
//     copy Lp (pitch 9 doubles, columns 6 and 7 zero) supplies both operands.  A tile strictly below the diagonal whose sixteen rows all
//     exist takes the short path (one multiplication for its four row offsets, no masks); diagonal tiles and the last tile row the masked
//     one.  One entry at a time (six dependent multiply-subtract pairs behind seven LDS reads) the update was 1.5 us a step -- 30 of the
//     factorisation's 51 us.
// rinv[k] = 1 / L_kk and the diagonal block's own factor (rows k0 .. k0 + 5 of Lp, dead since their panel step) are what the backward
// substitution reads.  flag: an LDS word (the positivity verdict).
typedef double win_d4 __attribute__((ext_vector_type(4)));
constexpr int kLpPitch = 9;
__device__ __forceinline__ double w_rsqrt_fma(double v) {
  double y = __builtin_amdgcn_rsq(v);
  y = __builtin_fma(0.5 * y, __builtin_fma(-v * y, y, 1.0), y);
  y = __builtin_fma(0.5 * y, __builtin_fma(-v * y, y, 1.0), y);
  return y;
}
__device__ bool win_cholesky_rhs(double* __restrict__ S, double* __restrict__ rinv, double* __restrict__ Lp, int n, int* __restrict__ flag,
                                 unsigned long long* prof) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (uniform: the tile loop below branches on it)
  constexpr int NW = kWinThreads / 64, LPP = kLpPitch;
  // (phase clocks of thread 0, kept in registers and added to prof[12..15] at the end: a read-modify-write of global memory per lap would put
  //  its latency into the phase that follows)
  unsigned long long tp = (prof && tid == 0) ? __builtin_amdgcn_s_memtime() : 0ull, pacc[4] = {0, 0, 0, 0};
  auto lapc = [&](int slot) { if (prof && tid == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_amdgcn_s_memtime(); pacc[slot - 12] += t - tp; tp = t; } };
  if (tid == 0) *flag = 1;
  __syncthreads();
  for (int k0 = 0; k0 < n; k0 += 6) {
    const int base = k0 + 6, rows = n + 1 - base;              // trailing rows base .. n (row n = the right-hand side)
    if (wave == 0 || 64 * wave < rows) {
      double Ld[21], rdg[6];
      bool ok = true;
#pragma unroll
      for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int c = 0; c <= a; c++) {
          double v = S[tri(k0 + a, k0 + c)];
#pragma unroll
          for (int k = 0; k < c; k++) v = __builtin_fma(-Ld[a * (a + 1) / 2 + k], Ld[c * (c + 1) / 2 + k], v);
          if (a == c) {
            if (!(v > 0)) ok = false;
            rdg[a] = w_rsqrt_fma(v > 0 ? v : 1.0);
            Ld[a * (a + 1) / 2 + a] = v;
          } else Ld[a * (a + 1) / 2 + c] = v * rdg[c];
        }
      }
      lapc(12);
      const int i = base + tid;
      if (i <= n) {
        double* row = S + tri(i, k0);
        double l[6];
#pragma unroll
        for (int c = 0; c < 6; c++) {
          double v = row[c];
#pragma unroll
          for (int k = 0; k < c; k++) v = __builtin_fma(-l[k], Ld[c * (c + 1) / 2 + k], v);
          l[c] = v * rdg[c];
        }
        double* lp = Lp + LPP * i;
#pragma unroll
        for (int c = 0; c < 6; c++) { row[c] = l[c]; lp[c] = l[c]; }
        lp[6] = 0.0; lp[7] = 0.0;
      }
      if (tid == 0) {
        if (!ok) *flag = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
          rinv[k0 + a] = rdg[a];
#pragma unroll
          for (int c = 0; c < a; c++) Lp[LPP * (k0 + a) + c] = Ld[a * (a + 1) / 2 + c];
        }
      }
      lapc(13);
    }
    __syncthreads();
    lapc(14);
    if (*flag == 0) break;                       // (uniform: read behind the barrier; nobody writes it before the next one)
    // ---- trailing update, 16 x 16 tiles of rows / columns base + 16 t; tile t = ti (ti + 1) / 2 + tj goes to wave t mod 8
    const int T = (rows + 15) >> 4, ntiles = T * (T + 1) / 2;
    const int lr = lane & 15, lq = lane >> 4;
    int ti = 0, tj = wave;
    while (tj > ti) { tj -= ti + 1; ti++; }
    for (int t = wave; t < ntiles; t += NW) {
      const int i0 = base + 16 * ti, j0 = base + 16 * tj;
      const int j = j0 + lr, ifirst = i0 + lq;
      if (ti > tj && i0 + 15 <= n) {
        const double* ra = Lp + LPP * (i0 + lr) + lq;
        const double* rb = Lp + LPP * (j0 + lr) + lq;
        const double a0 = ra[0], b0 = rb[0], a1 = ra[4], b1 = rb[4];
        double* e0 = S + tri(ifirst, j);                   // rows ifirst + 4 r: tri(i + 4, j) = tri(i, j) + 4 i + 10
        double* e1 = e0 + 4 * ifirst + 10;
        double* e2 = e1 + 4 * ifirst + 26;
        double* e3 = e2 + 4 * ifirst + 42;
        const double c0 = *e0, c1 = *e1, c2 = *e2, c3 = *e3;
        win_d4 acc = {0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
        *e0 = c0 - acc[0]; *e1 = c1 - acc[1]; *e2 = c2 - acc[2]; *e3 = c3 - acc[3];
      } else {
        const double* ra = Lp + LPP * min(i0 + lr, n) + lq;
        const double* rb = Lp + LPP * min(j0 + lr, n) + lq;
        const double a0 = ra[0], b0 = rb[0], a1 = ra[4], b1 = rb[4];
        double* e[4];
        double cur[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = ifirst + 4 * r;
          e[r] = (i > n || j > i || j >= n) ? nullptr : S + tri(i, j);
          cur[r] = e[r] ? *e[r] : 0.0;
        }
        win_d4 acc = {0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) if (e[r]) *e[r] = cur[r] - acc[r];
      }
      tj += NW;
      while (tj > ti) { tj -= ti + 1; ti++; }
    }
    lapc(15);
    __syncthreads();
    lapc(14);
  }
  if (prof && tid == 0) for (int i = 0; i < 4; i++) prof[12 + i] += pacc[i];
  return *flag != 0;
}
// backward substitution behind win_cholesky_rhs (rhs = row n of S holds y): x(i) = y(i) / L(i, i); y(j) -= L(i, j) x(i) for j < i, i
// descending, a camera's six unknowns per step.  ONE wave.
__device__ void win_backsubstitute(const double* __restrict__ S, const double* __restrict__ rinv, const double* __restrict__ Lp, double* __restrict__ rhs, int n) {
  const int lane = threadIdx.x & 63;
  for (int k0 = n - 6; k0 >= 0; k0 -= 6) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double x6[6];
#pragma unroll
    for (int a = 5; a >= 0; a--) {
      double v = rhs[k0 + a];
#pragma unroll
      for (int c = 5; c > a; c--) v = __builtin_fma(-Lp[kLpPitch * (k0 + c) + a], x6[c], v);        // (the block's own factor: win_cholesky_rhs left it in Lp)
      x6[a] = v * rinv[k0 + a];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 6) { double xv = x6[0]; xv = lane == 1 ? x6[1] : xv; xv = lane == 2 ? x6[2] : xv; xv = lane == 3 ? x6[3] : xv; xv = lane == 4 ? x6[4] : xv; xv = lane == 5 ? x6[5] : xv; rhs[k0 + lane] = xv; }
    for (int j = lane; j < k0; j += 64) {
      double v = rhs[j];
#pragma unroll
      for (int a = 5; a >= 0; a--) v = __builtin_fma(-S[tri(k0 + a, j)], x6[a], v);
      rhs[j] = v;
    }
  }
}

// solve(lambda), first half (block_solver.hpp:381-439): Dinv = (Hll + lambda I)^-1 per landmark, then landmark by landmark, per landmark
// every (edge, edge) pair of free cameras: Hschur(i1, i2) -= (W1 Dinv) W2^T, bschur(i1) -= W1 Dinv bl.  S and rhs live in LDS and start
// as Hpp + lambda I / bp.  The landmarks stream through LDS in chunks of whole landmarks: their Hll | bl and their W rows arrive by the
// register pipeline above, Dinv / Dinv bl are formed in place (and stored for the back substitution), W Dinv / W Dinv bl are formed
// in LDS -- they never exist in global memory --, and the chunk's pairs, listed by block, are applied: the 36 entries of a block's run
// are 36 work items; a block appears in ONE run per chunk and the chunks follow each other behind barriers, so every entry of S
// receives its subtractions in landmark order -- the order of g2o's loop.  rhs(6 i + a) likewise walks camera i's rows of the chunk.
__device__ void win_schur(const BaWin& W, double* S, double* rhs, double* stage, double lambda) {
  const int tid = threadIdx.x, n = 6 * W.nfree;
  double* __restrict__ Wb = stage;                                    // [C][18]
  double* __restrict__ Db = Wb + (size_t)W.C * kRowW;                 // [C][24]
  double* __restrict__ Hl = Db + (size_t)W.C * kRowD;                 // [C][12]  Hll | bl  ->  Dinv | Dinv bl
  int32_t* ints = reinterpret_cast<int32_t*>(Hl + (size_t)W.C * kRowH);
  for (int i = tid; i < n * (n + 1) / 2; i += kWinThreads) S[i] = 0.0;
  __syncthreads();
  for (int t = tid; t < 21 * W.nfree; t += kWinThreads) {
    const int i = t / 21, e = t - 21 * i;
    int a = 0, r = e; while (r > a) { r -= a + 1; a++; }
    const int b = r;
    S[tri(6 * i + a, 6 * i + b)] = W.Hpp[36 * (size_t)i + 6 * a + b] + (a == b ? lambda : 0.0);
  }
  for (int t = tid; t < n; t += kWinThreads) rhs[t] = W.bp[t];
  if (W.n_sc == 0) { __syncthreads(); return; }
  struct Desc { int r0, nr, ioff, ni, nu, np, la, nlm; };
  const Desc* descs = reinterpret_cast<const Desc*>(W.sc_desc);
  ChunkRegs<8, 6> pre;               // W rows: C * 18 <= 5 per thread; Hll | bl: C * 12 <= 3 per thread; ints <= 6 per thread
  Desc dsc = descs[0], nxt = descs[W.n_sc > 1 ? 1 : 0];      // the descriptor of chunk c + 1 is in registers a chunk before its loads go out
  pre.load_d(0, W.rowW + kRowW * (size_t)dsc.r0, dsc.nr * kRowW);
  pre.load_d(5, W.HB + kRowH * (size_t)dsc.la, dsc.nlm * kRowH);
  pre.load_i(W.sc_ints + dsc.ioff, dsc.ni);
  unsigned long long tp = __builtin_amdgcn_s_memtime();
  auto lap2 = [&](int slot) { if (W.prof && tid == 0) { const unsigned long long t = __builtin_amdgcn_s_memtime(); W.prof[slot] += t - tp; tp = t; } };
  for (int c = 0; c < W.n_sc; c++) {
    const Desc cur = dsc;
    lds_barrier();
    lap2(11);
    store_d<0, 5>(pre, Wb, cur.nr * kRowW);
    store_d<5, 8>(pre, Hl, cur.nlm * kRowH);
    store_i(pre, ints, cur.ni);
    if (c + 1 < W.n_sc) {
      dsc = nxt;
      pre.load_d(0, W.rowW + kRowW * (size_t)dsc.r0, dsc.nr * kRowW);
      pre.load_d(5, W.HB + kRowH * (size_t)dsc.la, dsc.nlm * kRowH);
      pre.load_i(W.sc_ints + dsc.ioff, dsc.ni);
      if (c + 2 < W.n_sc) nxt = descs[c + 2];
    }
    lds_barrier();
    lap2(12);
    const int32_t* crange = ints;
    const int32_t* crows = crange + W.nfree + 1;
    const int32_t* rowlm = crows + cur.nr;
    const int32_t* runs = rowlm + cur.nr;
    const int32_t* pairs = runs + cur.nu;
    if (tid < cur.nlm) {             // Dinv, Dinv bl of the chunk's landmarks, in place
      double* h = Hl + kRowH * tid;
      double D[9], Di[9], d3[3];
#pragma unroll
      for (int i = 0; i < 9; i++) D[i] = h[i];
      const double g[3] = {h[9], h[10], h[11]};
      D[0] += lambda; D[4] += lambda; D[8] += lambda;
      w_inv3(D, Di);
      w_mat3_vec(Di, g, d3);
      double* o = W.DD + kRowH * (size_t)(cur.la + tid);
#pragma unroll
      for (int i = 0; i < 9; i++) { h[i] = Di[i]; o[i] = Di[i]; }
#pragma unroll
      for (int i = 0; i < 3; i++) { h[9 + i] = d3[i]; o[9 + i] = d3[i]; }
    }
    lds_barrier();
    lap2(13);
    // W Dinv (18) and W Dinv bl (6) of every row: three entries of a thread at a time, operands first (each alone is two dependent LDS
    // round trips for five flops)
    for (int e0 = tid; e0 < cur.nr * kRowD; e0 += 3 * kWinThreads) {
      double w0[3], w1[3], w2[3], h0[3], h1[3], h2[3];
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int e = min(e0 + u * kWinThreads, cur.nr * kRowD - 1);
        const int row = e / kRowD, j = e - kRowD * row;
        const double* h = Hl + kRowH * rowlm[row];
        const int a = j < 18 ? j / 3 : j - 18;
        const double* W1 = Wb + kRowW * row + 3 * a;
        w0[u] = W1[0]; w1[u] = W1[1]; w2[u] = W1[2];
        if (j < 18) { const int b = j - 3 * a; h0[u] = h[b]; h1[u] = h[3 + b]; h2[u] = h[6 + b]; }
        else { h0[u] = h[9]; h1[u] = h[10]; h2[u] = h[11]; }
      }
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int e = e0 + u * kWinThreads;
        if (e < cur.nr * kRowD) Db[e] = w0[u] * h0[u] + w1[u] * h1[u] + w2[u] * h2[u];
      }
    }
    lds_barrier();
    lap2(14);
    for (int t = tid; t < n; t += kWinThreads) {
      const int i = t / 6, a = t - 6 * i;
      double sacc = rhs[t];
      const int q1 = crange[i + 1];
      for (int q = crange[i]; q < q1; q += 4) {                       // (four steps' operands together; a step beyond the range subtracts +0.0)
        double term[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const bool in = q + j < q1; const double v = Db[kRowD * (in ? crows[q + j] : 0) + 18 + a]; term[j] = in ? v : 0.0; }
#pragma unroll
        for (int j = 0; j < 4; j++) sacc -= term[j];
      }
      rhs[t] = sacc;
    }
    lap2(3);
    if (6 * (cur.nu - 1) >= kWinThreads / 2) {
      // many short runs (a window of many cameras): a work item = row a of a block's run -- its three W Dinv values meet all six
      // columns of W2 (b <= a on a diagonal block)
      for (int t = tid; t < 6 * (cur.nu - 1); t += kWinThreads) {
        const int ru = t / 6, a = t - 6 * ru;
        const int rw = runs[ru], i1 = rw & 255, i2 = (rw >> 8) & 255, q0 = rw >> 16, q1 = runs[ru + 1] >> 16;
        const int nb = i1 == i2 ? a + 1 : 6;
        double* Srow = S + tri(6 * i1 + a, 6 * i2);
        double sv[6];
#pragma unroll
        for (int b = 0; b < 6; b++) sv[b] = b < nb ? Srow[b] : 0.0;
        for (int q = q0; q < q1; q++) {
          const int pr = pairs[q];
          const double* WD = Db + kRowD * (pr & 0xffff) + 3 * a;
          const double* W2 = Wb + kRowW * (pr >> 16);
          const double d0 = WD[0], d1 = WD[1], d2 = WD[2];
#pragma unroll
          for (int b = 0; b < 6; b++) sv[b] -= d0 * W2[3 * b] + d1 * W2[3 * b + 1] + d2 * W2[3 * b + 2];
        }
#pragma unroll
        for (int b = 0; b < 6; b++) if (b < nb) Srow[b] = sv[b];
      }
    } else {
      // few long runs (two or three cameras): a work item = one entry (a, b) of a block's run, its pairs two at a time
      for (int t = tid; t < 36 * (cur.nu - 1); t += kWinThreads) {
        const int ru = t / 36, ab = t - 36 * ru, a = ab / 6, b = ab - 6 * a;
        const int rw = runs[ru], i1 = rw & 255, i2 = (rw >> 8) & 255, q0 = rw >> 16, q1 = runs[ru + 1] >> 16;
        if (i1 == i2 && b > a) continue;
        const int idx = tri(6 * i1 + a, 6 * i2 + b);
        double sacc = S[idx];
        for (int q = q0; q < q1; q += 2) {
          double term[2];
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const bool in = q + j < q1;
            const int pr = in ? pairs[q + j] : 0;
            const double* WD = Db + kRowD * (pr & 0xffff) + 3 * a;
            const double* W2 = Wb + kRowW * (pr >> 16) + 3 * b;
            const double v = WD[0] * W2[0] + WD[1] * W2[1] + WD[2] * W2[2];
            term[j] = in ? v : 0.0;
          }
          sacc -= term[0];
          sacc -= term[1];
        }
        S[idx] = sacc;
      }
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------ tree reductions (cluster form)
// Sums as trees in a FIXED order (deterministic, but not g2o's sequential order): partial sums per lane meet in four DPP steps and one
// LDS hop.  Used by the cluster form below (k_ba_window_cluster); the sequential-order kernel (k_ba_window) does not use them.
template <int CTRL>
__device__ __forceinline__ double w_dpp_add(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return v + __hiloint2double(hi, lo);
}
// the sum over a DPP row (16 lanes), in all of its lanes: quad swaps, half-row mirror, row mirror
__device__ __forceinline__ double w_row_sum(double v) {
  v = w_dpp_add<0xB1>(v); v = w_dpp_add<0x4E>(v); v = w_dpp_add<0x141>(v); v = w_dpp_add<0x140>(v);
  return v;
}
// acc[0..N) of the 64 lanes -> lane i < N returns the wave's sum of acc[i] ((row0 + row1) + (row2 + row3)); wbuf: 4 * N doubles of LDS, this wave's
template <int N>
__device__ __forceinline__ double w_wave_reduce(double (&acc)[N], double* __restrict__ wbuf) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < N; i++) acc[i] = w_row_sum(acc[i]);
  __builtin_amdgcn_wave_barrier();      // (the previous use of wbuf has been read)
  if ((lane & 15) == 0) {
#pragma unroll
    for (int i = 0; i < N; i++) wbuf[(lane >> 4) * N + i] = acc[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  double r = 0.0;
  if (lane < N) r = (wbuf[lane] + wbuf[N + lane]) + (wbuf[2 * N + lane] + wbuf[3 * N + lane]);
  return r;
}
// ------------------------------------------------------------------------------------------------ the kernel
__global__ void __launch_bounds__(kWinThreads) k_ba_window(const BaWin* __restrict__ wins, const volatile int* __restrict__ stop) {
  extern __shared__ double lds[];
  constexpr int NT = kWinThreads;
  const BaWin& W = wins[blockIdx.x];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int n = 6 * W.nfree, nl = 3 * W.nact;
  double* S = lds;                                  // packed lower triangle of the reduced camera system, n (n + 1) / 2
  double* rhs = S + (size_t)n * (n + 1) / 2;        // bschur -> x_p                                          [n]
  double* diag = rhs + n;                           // L_kk                                                   [n]
  double* ctl = diag + n;                           // control words shared by the workgroup                  [64]
  double* seqbuf = ctl + 64 + 128 * (wave & 1);     // wave_sequential_sum's buffer (waves 0 and 1 use it)    [2][128]
  double* stage = ctl + 64 + 256;                   // streaming area of the chunked passes                   [stage_doubles]
  dvm_ba_stats* const st = W.stats;                 // written by thread 0 only
  if (tid == 0) {
    st->iterations = st->total_trials = st->stop_reason = st->kernel_us = 0;
    st->chi2_initial = st->chi2_final = st->lambda_final = 0;
    for (int i = 0; i < 64; i++) { st->trials_per_iter[i] = 0; st->chi2_per_iter[i] = 0; st->lambda_per_iter[i] = 0; }
    st->ms_structure = st->ms_optimize = 0; st->spec_trials = st->spec_kept = 0;
  }
  // g2o's buildStructure reallocates _x: "the last successful solve" starts as zeros
  for (int i = tid; i < n + nl; i += NT) W.x[i] = 0.0;
  // the trial state starts as a copy (fixed cameras and unobserved landmarks never change)
  for (int i = tid; i < 7 * W.P; i += NT) W.poses_t[i] = W.poses[i];
  for (int i = tid; i < 3 * W.L; i += NT) W.pts_t[i] = W.pts[i];
  double* poses = W.poses; double* poses_t = W.poses_t; double* pts = W.pts; double* pts_t = W.pts_t;
  __syncthreads();

  double lambda = -1, ni = 2, currentChi = 0, chi_last = 0;
  int nBad = 0, it_done = 0, trials_total = 0, stop_reason = 0;
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
  auto lap = [&](int slot) {        // phase timing: thread 0 between barriers
    if (W.prof && tid == 0) { const unsigned long long t = __builtin_amdgcn_s_memtime(); W.prof[slot] += t - tprev; tprev = t; }
  };
  for (int it = 0; it < W.iterations; it++) {
    if (tid == 0) ctl[0] = (stop && *stop) ? 1.0 : 0.0;
    __syncthreads();
    if (ctl[0] != 0.0) break;
    // computeActiveErrors + robust chi2 + buildSystem at the accepted state.  (From the second iteration on g2o recomputes the chi2
    // of the state the last accepted trial has just evaluated: same state, same sums, same bits -- only the Jacobians are new.)
    lap(15);
    win_edge_pass<true>(W, poses, pts);
    __syncthreads();
    lap(0);
    if (it == 0) {
      if (wave == 0) { const double c = wave_sequential_sum(W.e_rho, W.E, seqbuf); if (tid == 0) ctl[1] = c; }
    }
    win_accumulate_landmarks(W);
    __syncthreads();
    lap(1);
    win_accumulate_cameras(W, stage);
    __syncthreads();
    lap(2);
    if (it == 0) {
      currentChi = ctl[1];
      if (tid == 0) st->chi2_initial = currentChi;
      // computeLambdaInit: tau * max |diagonal| over all active vertices (a maximum has no order)
      double mx = 0;
      for (int i = tid; i < n; i += NT) mx = fmax(mx, fabs(W.Hpp[36 * (size_t)(i / 6) + 7 * (i % 6)]));
      for (int i = tid; i < nl; i += NT) mx = fmax(mx, fabs(W.HB[kRowH * (size_t)(i / 3) + 4 * (i % 3)]));
      for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
      __syncthreads();
      if ((tid & 63) == 0) ctl[8 + wave] = mx;
      __syncthreads();
      mx = ctl[8];
      for (int w2 = 1; w2 < NT / 64; w2++) mx = fmax(mx, ctl[8 + w2]);
      lambda = 1e-5 * mx;
      ni = 2; nBad = 0;
    }
    const double iniChi = currentChi;
    double tempChi = currentChi, rho = 0;
    int qmax = 0;
    bool stopped = false;
    do {
      // ---- solve(lambda): Dinv, Schur complement, reduced right-hand side (one streamed pass)
      lap(3);
      win_schur(W, S, rhs, stage, lambda);
      lap(4);
      // ---- Cholesky + substitution of the reduced system (win_cholesky / win_substitute: the column-by-column operation order)
      const bool ok = win_cholesky(S, diag, ctl + 64, n);       // (Lp: the streaming area is idle during the solve)
      lap(5);
      if (ok) {
        if (wave == 0) win_substitute(S, diag, rhs, n);
        __syncthreads();
        lap(6);
        for (int i = tid; i < n; i += NT) W.x[i] = rhs[i];
        // xl = Dinv (bl - W^T xp): per landmark, its free-camera rows in order, per row the six camera components in order
        for (int li = tid; li < W.nact; li += NT) {
          const double* hb = W.HB + kRowH * (size_t)li;
          double c0 = hb[9], c1 = hb[10], c2 = hb[11];
          for (int q = W.f_start[li]; q < W.f_start[li + 1]; q++) {
            const int i = W.f_cam[q];
            const double* Wk = W.rowW + kRowW * (size_t)q;
#pragma unroll
            for (int a = 0; a < 6; a++) {
              const double xa = rhs[6 * i + a];
              c0 -= Wk[3 * a] * xa; c1 -= Wk[3 * a + 1] * xa; c2 -= Wk[3 * a + 2] * xa;
            }
          }
          const double c[3] = {c0, c1, c2};
          double xl[3];
          w_mat3_vec(W.DD + kRowH * (size_t)li, c, xl);
          W.x[n + 3 * (size_t)li] = xl[0]; W.x[n + 3 * (size_t)li + 1] = xl[1]; W.x[n + 3 * (size_t)li + 2] = xl[2];
        }
      }
      __syncthreads();
      lap(7);
      // ---- the update is applied and the errors evaluated whether or not the solve succeeded (g2o: x then still holds the last
      // successful solve, optimization_algorithm_levenberg.cpp:107-127); computeScale's terms x_j (lambda x_j + b_j)
      for (int i = tid; i < W.nfree; i += NT) {
        const int p = W.free_pose[i];
        w_se3_oplus(poses + 7 * (size_t)p, W.x + 6 * (size_t)i, poses_t + 7 * (size_t)p);
      }
      for (int t = tid; t < nl; t += NT) {
        const int l = W.act_pt[t / 3];
        pts_t[3 * (size_t)l + t % 3] = pts[3 * (size_t)l + t % 3] + W.x[n + t];
      }
      for (int j = tid; j < n; j += NT) { const double xj = W.x[j]; W.terms[j] = xj * (lambda * xj + W.bp[j]); }
      for (int j = tid; j < nl; j += NT) { const double xj = W.x[n + j]; W.terms[n + j] = xj * (lambda * xj + W.HB[kRowH * (size_t)(j / 3) + 9 + j % 3]); }
      __syncthreads();
      lap(8);
      win_edge_pass<false>(W, poses_t, pts_t);
      __syncthreads();
      lap(9);
      if (wave == 0) { const double c = wave_sequential_sum(W.e_rho, W.E, seqbuf); if (tid == 0) ctl[1] = c; }
      if (wave == 1) { const double c = wave_sequential_sum(W.terms, n + nl, seqbuf); if (tid == 64) ctl[2] = c; }
      __syncthreads();
      lap(10);
      // ---- the decision, taken by every thread on the same words (optimization_algorithm_levenberg.cpp:113-147)
      tempChi = ok ? ctl[1] : 1.7976931348623157e308;
      rho = currentChi - tempChi;
      const double scale = ctl[2] + 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - f64_cube(2 * rho - 1);
        alpha = (2. / 3. < alpha) ? 2. / 3. : alpha;            // std::min(alpha, 2/3) and std::max(1/3, alpha) as the C++ library defines
        lambda *= (1. / 3. < alpha) ? alpha : 1. / 3.;          // them (a NaN alpha passes the first and loses the second; fmin / fmax differ)
        ni = 2;
        currentChi = tempChi;
        double* sw = poses; poses = poses_t; poses_t = sw;      // discardTop(): the trial state becomes the state
        sw = pts; pts = pts_t; pts_t = sw;
        // (the other buffer's fixed cameras / inactive landmarks are the same values: both started as copies of the input)
      } else {
        lambda *= ni;
        ni *= 2;                                               // pop(): the state stays
      }
      qmax++;
      trials_total++;
      if (tid == 0) ctl[0] = (stop && *stop) ? 1.0 : 0.0;
      __syncthreads();
      stopped = ctl[0] != 0.0;
      __syncthreads();
    } while (rho < 0 && qmax < 10 && !stopped);
    it_done++;
    chi_last = currentChi;
    if (tid == 0 && it < 64) { st->trials_per_iter[it] = qmax; st->chi2_per_iter[it] = currentChi; st->lambda_per_iter[it] = lambda; }
    if (qmax == 10 || rho == 0) { stop_reason = 1; break; }
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) { stop_reason = 2; break; }
  }
  __syncthreads();
  // No iteration ran (iterations == 0, or the stop word was already up at the first poll): no edge pass has written e_chi2 / e_rho, and the
  // caller downloads them (Optimizer_shim's LocalBundleAdjustment erases observations on chi2 > 5.991).  Evaluate the edges at the
  // unchanged input state, so that what leaves is the chi2 OF the state that leaves -- never the previous tenant of the buffer.
  if (it_done == 0) {
    win_edge_pass<false>(W, poses, pts);
    __syncthreads();
    if (wave == 0) { const double c = wave_sequential_sum(W.e_rho, W.E, seqbuf); if (tid == 0) { st->chi2_initial = c; chi_last = c; } }
    __syncthreads();
  }
  // results: the accepted state (the buffers may have been swapped any number of times), depth signs at that state
  for (int i = tid; i < 7 * W.P; i += NT) W.out_poses[i] = poses[i];
  for (int i = tid; i < 3 * W.L; i += NT) W.out_pts[i] = pts[i];
  for (int k = tid; k < W.E; k += NT) {
    const double* T = poses + 7 * (size_t)W.e_pose[k];
    const double* X = pts + 3 * (size_t)W.e_point[k];
    double R[9], Xc[3];
    w_quat_to_R(T + 3, R);
    w_mat3_vec(R, X, Xc);
    W.e_depth[k] = (Xc[2] + T[2]) > 0.0 ? 1 : 0;
  }
  if (tid == 0) {
    st->iterations = it_done; st->total_trials = trials_total; st->chi2_final = chi_last; st->lambda_final = lambda; st->stop_reason = stop_reason;
  }
}

// ------------------------------------------------------------------------------------------------ the CLUSTER form
// dvm_ba_optimize_windows_fast: the optimizer with tree sums in a fixed order, on G workgroups per window (G = 1, 2, 4, 8).  One CU streams a window's ~23 MB per iteration through ONE 64 B/clk
// memory pipe with eight waves' worth of requests in flight (~40 GB/s): with K <= 32 windows on 256 CUs the other seven eighths of the
// chip idle.  Here every data-parallel phase is split over the cluster -- element ranges in EIGHT fixed parts (part p belongs to
// workgroup p % G: the partial sums and therefore the results do not depend on G), cameras / blocks round-robin over the cluster's
// waves -- and the phases are separated by cluster barriers (Guideline 16, counter form: every wave drains, lane 0 releases at agent
// scope, arrives on the window's counter, polls it relaxed, acquires).  The reduced system is solved by workgroup 0 (S in its LDS);
// every workgroup takes the LM decisions redundantly on the same words.  Correctness does not depend on where the workgroups run;
// the block -> window map only tries to keep a window's workgroups on one XCD (observed: block b runs on XCD b % 8).  Every spin is
// bounded: a barrier that times out (a workgroup of the cluster is not resident) raises cl_tmo and the whole cluster leaves; the
// host then solves the batch with G = 1, which needs no co-residency.
constexpr int kParts = 8;
struct ClusterCtx { int g, G; unsigned int epoch; bool dead; };

__device__ __forceinline__ bool cluster_barrier(const BaWin& W, ClusterCtx& C, int* s_flag) {
  if (C.G == 1) { __syncthreads(); return true; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every wave: its stores have left
  __syncthreads();
  C.epoch++;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (restated: the compiler may drop the wait behind buffer_wbl2)
    __hip_atomic_fetch_add(W.cl_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int target = C.epoch * (unsigned int)C.G;
    int ok = 1;
    for (unsigned int spins = 0; __hip_atomic_load(W.cl_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; spins++) {
      __builtin_amdgcn_s_sleep(2);
      if ((spins & 63u) == 63u && __hip_atomic_load(W.cl_tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
      if (spins > (1u << 20)) { __hip_atomic_store(W.cl_tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *s_flag = ok;
  }
  __syncthreads();
  const bool ok = *s_flag != 0;
  __syncthreads();
  if (!ok) C.dead = true;
  return ok;
}
// the elements [lo, hi) of part p of n (fixed eight parts, multiples of 64)
__device__ __forceinline__ void part_range(int n, int p, int& lo, int& hi) {
  const int chunk = (((n + kParts - 1) / kParts) + 63) & ~63;
  lo = min(n, p * chunk); hi = min(n, lo + chunk);
}
// the partial sums of this workgroup's parts of v[0..n) -> W.cl_part[slot * 8 + part]
__device__ void cluster_partial_sums(const BaWin& W, const ClusterCtx& C, const double* __restrict__ v, int n, int slot, double* red) {
  __syncthreads();                    // (v's entries of this workgroup's parts were written by its own threads just before)
  for (int p = C.g; p < kParts; p += C.G) {
    int lo, hi;
    part_range(n, p, lo, hi);
    double s = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += kWinThreads) s += v[i];
    s = w_row_sum(s);
    __syncthreads();
    if ((threadIdx.x & 15) == 0) red[threadIdx.x >> 4] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < kWinThreads / 16; i++) t += red[i];
      W.cl_part[slot * kParts + p] = t;
    }
  }
}
__device__ __forceinline__ double cluster_total(const BaWin& W, int slot) {
  double t = 0.0;
#pragma unroll
  for (int p = 0; p < kParts; p++) t += W.cl_part[slot * kParts + p];
  return t;
}
// Dinv and Dinv bl of a landmark from Hll | bl and the damping: formed where they are needed (the T rows, the back substitution) --
// the same operations on the same words give the same bits, and a phase (and its barrier) less
__device__ __forceinline__ void w_dinv(const double* __restrict__ h, double lambda, double* Di, double* d3) {
  double D[9];
#pragma unroll
  for (int i = 0; i < 9; i++) D[i] = h[i];
  const double g3[3] = {h[9], h[10], h[11]};
  D[0] += lambda; D[4] += lambda; D[8] += lambda;
  w_inv3(D, Di);
  w_mat3_vec(Di, g3, d3);
}

template <bool JAC>
__device__ void cl_edge_pass(const BaWin& W, const ClusterCtx& C, const double* __restrict__ poses, const double* __restrict__ pts) {
  const size_t Fp = (size_t)W.Fp;
  for (int p = C.g; p < kParts; p += C.G) {
    int lo, hi;
    part_range(W.E, p, lo, hi);
    for (int k = lo + threadIdx.x; k < hi; k += kWinThreads) {
      const int pi = W.e_pose[k], l = W.e_point[k];
      const double* T = poses + 7 * (size_t)pi;
      const double* X = pts + 3 * (size_t)l;
      double R[9], Xc[3];
      w_quat_to_R(T + 3, R);
      w_mat3_vec(R, X, Xc);
      Xc[0] += T[0]; Xc[1] += T[1]; Xc[2] += T[2];
      const double x = Xc[0], y = Xc[1], z = Xc[2];
      const double info = W.e_info[k];
      const double e0 = W.e_obs[2 * k] - (W.fx * x / z + W.cx);
      const double e1 = W.e_obs[2 * k + 1] - (W.fy * y / z + W.cy);
      const double chi2 = e0 * info * e0 + e1 * info * e1;
      double r0, r1;
      w_robustify(chi2, W.delta, r0, r1);
      W.chi_s[k] = chi2;
      W.e_rho[k] = r0;
      if (!JAC) continue;
      const double J[6] = {-(W.fx / z), 0, W.fx * x / (z * z), 0, -(W.fy / z), W.fy * y / (z * z)};
      double A[6], B[12];
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) A[3 * r + c] = J[3 * r] * R[c] + J[3 * r + 1] * R[3 + c] + J[3 * r + 2] * R[6 + c];
      const double w = r1 * info;
      const double wr0 = -info * e0 * r1, wr1 = -info * e1 * r1;
      double2* oA = reinterpret_cast<double2*>(W.rowA + kRowA * (size_t)k);
      oA[0] = make_double2(A[0], A[1]); oA[1] = make_double2(A[2], A[3]); oA[2] = make_double2(A[4], A[5]);
      oA[3] = make_double2(w, wr0); oA[4] = make_double2(wr1, 0.0);
      if (k < W.F) {
        const double S[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int c = 0; c < 6; c++) B[6 * r + c] = J[3 * r] * S[c] + J[3 * r + 1] * S[6 + c] + J[3 * r + 2] * S[12 + c];
#pragma unroll
        for (int i = 0; i < 12; i++) W.Bs[i * Fp + k] = B[i];
        W.Bs[12 * Fp + k] = w; W.Bs[13 * Fp + k] = wr0; W.Bs[14 * Fp + k] = wr1;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
          for (int b2 = 0; b2 < 3; b2++) W.Ws[(3 * a + b2) * Fp + k] = w * (B[a] * A[b2] + B[6 + a] * A[3 + b2]);
      }
    }
  }
}
// Hll / bl of this workgroup's landmarks, Hpp / bp of its waves' cameras; mx = max |diagonal| over what it formed
__device__ double cl_accumulate(const BaWin& W, const ClusterCtx& C, double* wred) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t Fp = (size_t)W.Fp;
  double mx = 0.0;
  for (int p = C.g; p < kParts; p += C.G) {
    int lo, hi;
    part_range(W.nact, p, lo, hi);
    for (int li = lo + threadIdx.x; li < hi; li += kWinThreads) {
      double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g3[3] = {0, 0, 0};
      for (int q = W.pt_start[li]; q < W.pt_start[li + 1]; q++) {
        const double2* Ap = reinterpret_cast<const double2*>(W.rowA + kRowA * (size_t)W.pt_edges[q]);
        const double2 a01 = Ap[0], a23 = Ap[1], a45 = Ap[2], ww = Ap[3], w1 = Ap[4];
        const double A[6] = {a01.x, a01.y, a23.x, a23.y, a45.x, a45.y};
        const double w = ww.x, wr0 = ww.y, wr1 = w1.x;
#pragma unroll
        for (int a = 0; a < 3; a++) {
          g3[a] += A[a] * wr0 + A[3 + a] * wr1;
#pragma unroll
          for (int b2 = 0; b2 < 3; b2++) h[3 * a + b2] += w * (A[a] * A[b2] + A[3 + a] * A[3 + b2]);
        }
      }
      double* o = W.HB + kRowH * (size_t)li;
#pragma unroll
      for (int i = 0; i < 9; i++) o[i] = h[i];
#pragma unroll
      for (int i = 0; i < 3; i++) o[9 + i] = g3[i];
      mx = fmax(mx, fmax(fabs(h[0]), fmax(fabs(h[4]), fabs(h[8]))));
    }
  }
  double* wbuf = wred + wave * 4 * 36;
  for (int i = C.g * (kWinThreads / 64) + wave; i < W.nfree; i += C.G * (kWinThreads / 64)) {
    double acc[27];
#pragma unroll
    for (int t = 0; t < 27; t++) acc[t] = 0.0;
    const int q1 = W.cam_start[i + 1];
    for (int q = W.cam_start[i] + lane; q < q1; q += 64) {
      double B[15];
#pragma unroll
      for (int j = 0; j < 15; j++) B[j] = W.Bs[j * Fp + q];
      const double w = B[12], wr0 = B[13], wr1 = B[14];
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int b2 = 0; b2 <= a; b2++) acc[t++] += w * (B[a] * B[b2] + B[6 + a] * B[6 + b2]);
      }
#pragma unroll
      for (int a = 0; a < 6; a++) acc[21 + a] += B[a] * wr0 + B[6 + a] * wr1;
    }
    const double tot = w_wave_reduce<27>(acc, wbuf);
    if (lane < 21) {
      int a = 0, r = lane; while (r > a) { r -= a + 1; a++; }
      const int b2 = r;
      W.Hpp[36 * (size_t)i + 6 * a + b2] = tot; W.Hpp[36 * (size_t)i + 6 * b2 + a] = tot;
      if (a == b2) mx = fmax(mx, fabs(tot));
    } else if (lane < 27) W.bp[6 * (size_t)i + (lane - 21)] = tot;
  }
  return mx;
}
// W Dinv | W Dinv bl of this workgroup's rows
__device__ void cl_t_rows(const BaWin& W, const ClusterCtx& C, double lambda) {
  const size_t Fp = (size_t)W.Fp;
  for (int p = C.g; p < kParts; p += C.G) {
    int lo, hi;
    part_range(W.F, p, lo, hi);
    for (int r = lo + threadIdx.x; r < hi; r += kWinThreads) {
      const double2* hp = reinterpret_cast<const double2*>(W.HB + kRowH * (size_t)W.e_lm[r]);
      double w[18], h[12], Di[9], d3[3];
#pragma unroll
      for (int j = 0; j < 6; j++) { const double2 v = hp[j]; h[2 * j] = v.x; h[2 * j + 1] = v.y; }
#pragma unroll
      for (int j = 0; j < 18; j++) w[j] = W.Ws[j * Fp + r];
      w_dinv(h, lambda, Di, d3);
#pragma unroll
      for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int b2 = 0; b2 < 3; b2++) W.Ts[(3 * a + b2) * Fp + r] = w[3 * a] * Di[b2] + w[3 * a + 1] * Di[3 + b2] + w[3 * a + 2] * Di[6 + b2];
        W.Ts[(18 + a) * Fp + r] = w[3 * a] * d3[0] + w[3 * a + 1] * d3[1] + w[3 * a + 2] * d3[2];
      }
    }
  }
}
// the reduced right-hand side and the blocks of the reduced system, each formed completely by ONE wave of the cluster
__device__ void cl_schur(const BaWin& W, const ClusterCtx& C, double* wred, double lambda) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t Fp = (size_t)W.Fp;
  double* wbuf = wred + wave * 4 * 36;
  // Which wave forms what is the host's schedule (build_window_fast: longest task first onto the least loaded of the cluster's 8 G waves): a
  // diagonal block has ~8 rounds of 64 pairs, an off-diagonal one 1-2, and dealt round-robin the waves' loads differed by a factor of two.
  // Every right-hand side and every block is still formed completely by ONE wave, in its own pair order: the same bits whatever the schedule.
  const int gw = C.g * (kWinThreads / 64) + wave;
  for (int idx = W.sched_start[gw]; idx < W.sched_start[gw + 1]; idx++) {
    const int task = W.sched_task[idx];
    if (task >= W.nfree) continue;
    const int i = task;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const int q1 = W.cam_start[i + 1];
    for (int q = W.cam_start[i] + lane; q < q1; q += 64) {
#pragma unroll
      for (int a = 0; a < 6; a++) acc[a] += W.Ts[(18 + a) * Fp + q];
    }
    const double tot = w_wave_reduce<6>(acc, wbuf);
    if (lane < 6) W.rhsg[6 * i + lane] = W.bp[6 * (size_t)i + lane] - tot;
  }
  for (int idx = W.sched_start[gw]; idx < W.sched_start[gw + 1]; idx++) {
    const int bk = W.sched_task[idx] - W.nfree;
    if (bk < 0) continue;
    double acc[36];
#pragma unroll
    for (int t = 0; t < 36; t++) acc[t] = 0.0;
    const int q1 = W.bp_start[bk + 1];
    for (int q = W.bp_start[bk] + lane; q < q1; q += 64) {
      const int2 pr = W.bp_pairs[q];
      double t1[18], w2[18];
      // (gathered out of the column-major arrays, 36 eight-byte loads per pair.  Row-major copies of W Dinv and W -- a pair's operands in three
      //  cache lines each instead of eighteen -- were measured: the pass went from 54 to 70 us, the pass that writes the rows from 19 to 31)
#pragma unroll
      for (int j = 0; j < 18; j++) { t1[j] = W.Ts[j * Fp + pr.x]; w2[j] = W.Ws[j * Fp + pr.y]; }
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b2 = 0; b2 < 6; b2++) acc[6 * a + b2] += t1[3 * a] * w2[3 * b2] + t1[3 * a + 1] * w2[3 * b2 + 1] + t1[3 * a + 2] * w2[3 * b2 + 2];
    }
    const double tot = w_wave_reduce<36>(acc, wbuf);
    const int ij = W.blk_ij[bk], i1 = ij & 255, i2 = (ij >> 8) & 255;
    if (lane < 36) {
      const int a = lane / 6, b2 = lane - 6 * a;
      const double base = i1 == i2 ? W.Hpp[36 * (size_t)i1 + 6 * a + b2] + (a == b2 ? lambda : 0.0) : 0.0;
      W.Sblk[36 * (size_t)bk + lane] = base - tot;
    }
  }
}

__global__ void __launch_bounds__(kWinThreads) k_ba_window_cluster(const BaWin* __restrict__ wins, const volatile int* __restrict__ stop, int K, int G) {
  extern __shared__ double lds[];
  constexpr int NT = kWinThreads;
  // K < 0: the placement test's map (tests/test_gpu_ba_window_fast.py) -- a window's workgroups are CONSECUTIVE blocks, i.e. spread over all
  // eight XCDs; the results must not depend on it (they do not: the barriers release and acquire at agent scope)
  const bool scatter = K < 0;
  if (scatter) K = -K;
  const int bid = blockIdx.x, slot = bid >> 3;
  const int wi = scatter ? bid / G : (slot / G) * 8 + (bid & 7);
  if (wi >= K) return;
  const BaWin& W = wins[wi];
  ClusterCtx C{scatter ? bid % G : slot % G, G, 0u, false};
  const bool leader = C.g == 0;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int n = 6 * W.nfree, nl = 3 * W.nact;
  double* S = lds;                                  // (workgroup 0 of the cluster only) packed lower triangle, n (n + 1) / 2
  double* rhs = S + (size_t)n * (n + 1) / 2;
  double* diag = rhs + n;
  double* ctl = diag + n;                           // [64]
  double* red = ctl + 64;                           // [256] partial sums of the tree reductions
  double* wred = ctl + 64 + 256;                    // the waves' reduction buffers
  int* const s_flag_p = reinterpret_cast<int*>(ctl + 60);   // the barrier's verdict (no static LDS beside the 160 KB dynamic block)
#define s_flag (*s_flag_p)
  dvm_ba_stats* const st = W.stats;
  // DVM_BA_WINDOW_PROF: shader-clock cycles per phase of the cluster's workgroup 0 (slot 10: waiting in the barriers), by its thread 0
  unsigned long long tprev = (W.prof && leader && tid == 0) ? __builtin_amdgcn_s_memtime() : 0ull;
  auto lap = [&](int slot) { if (W.prof && leader && tid == 0) { const unsigned long long t = __builtin_amdgcn_s_memtime(); W.prof[slot] += t - tprev; tprev = t; } };
#define CL_BARRIER() do { lap(ph); if (!cluster_barrier(W, C, &s_flag)) return; lap(10); } while (0)
  int ph = 11;
  if (leader && tid == 0) {
    st->iterations = st->total_trials = st->stop_reason = st->kernel_us = 0;
    st->chi2_initial = st->chi2_final = st->lambda_final = 0;
    for (int i = 0; i < 64; i++) { st->trials_per_iter[i] = 0; st->chi2_per_iter[i] = 0; st->lambda_per_iter[i] = 0; }
    st->ms_structure = st->ms_optimize = 0; st->spec_trials = st->spec_kept = 0;
    W.cl_ctl[1] = (stop && *stop) ? 1.0 : 0.0;
  }
  for (int p = C.g; p < kParts; p += G) {
    int lo, hi;
    part_range(n + nl, p, lo, hi);
    for (int i = lo + tid; i < hi; i += NT) W.x[i] = 0.0;
    part_range(7 * W.P, p, lo, hi);
    for (int i = lo + tid; i < hi; i += NT) W.poses_t[i] = W.poses[i];
    part_range(3 * W.L, p, lo, hi);
    for (int i = lo + tid; i < hi; i += NT) W.pts_t[i] = W.pts[i];
  }
  double* poses = W.poses; double* poses_t = W.poses_t; double* pts = W.pts; double* pts_t = W.pts_t;
  CL_BARRIER();

  double lambda = -1, ni = 2, currentChi = 0, chi_last = 0;
  int nBad = 0, it_done = 0, trials_total = 0, stop_reason = 0;
  bool stopped = W.cl_ctl[1] != 0.0;
  for (int it = 0; it < W.iterations && !stopped; it++) {
    ph = 0;
    // (Evaluating a trial WITH its Jacobians into a second set of rows, so that an accepted trial hands the next iteration its linearisation --
    //  as the tile solver's loop and k_pose_optimize do -- was measured: 2.92 -> 3.27 ms for 32 windows.  The pass it saves is 15 us, but
    //  two sets of rows are 260 MB for 32 windows and no longer stay in the 256 MB memory-side cache: the accumulation behind it went from
    //  27 to 48 us, the W Dinv pass from 18 to 25.)
    cl_edge_pass<true>(W, C, poses, pts);
    if (it == 0) cluster_partial_sums(W, C, W.e_rho, W.E, 0, red);
    CL_BARRIER();
    ph = 1;
    {
      double mx = cl_accumulate(W, C, wred);
      if (it == 0) {      // computeLambdaInit's max |diagonal|: this workgroup's share -> its parts' slots (a maximum has no order)
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
        __syncthreads();
        if ((tid & 63) == 0) red[wave] = mx;
        __syncthreads();
        if (tid == 0) {
          double m = red[0];
          for (int w2 = 1; w2 < NT / 64; w2++) m = fmax(m, red[w2]);
          for (int p = 0; p < kParts; p++) if (p % G == C.g) W.cl_part[2 * kParts + p] = m;
        }
      }
    }
    CL_BARRIER();
    if (it == 0) {
      currentChi = cluster_total(W, 0);
      if (leader && tid == 0) st->chi2_initial = currentChi;
      double mx = 0;
      for (int p = 0; p < kParts; p++) mx = fmax(mx, W.cl_part[2 * kParts + p]);
      lambda = 1e-5 * mx;
      ni = 2; nBad = 0;
    }
    const double iniChi = currentChi;
    double tempChi = currentChi, rho = 0;
    int qmax = 0;
    do {
      ph = 2;
      cl_t_rows(W, C, lambda);
      CL_BARRIER();
      ph = 3;
      cl_schur(W, C, wred, lambda);
      CL_BARRIER();
      ph = 6;
      if (leader) {
        for (int i = tid; i < n * (n + 1) / 2; i += NT) S[i] = 0.0;
        __syncthreads();
        for (int t = tid; t < 36 * W.nblk; t += NT) {
          const int bk = t / 36, e = t - 36 * bk, a = e / 6, b2 = e - 6 * a;
          const int ij = W.blk_ij[bk], i1 = ij & 255, i2 = (ij >> 8) & 255;
          if (i1 != i2 || b2 <= a) S[tri(6 * i1 + a, 6 * i2 + b2)] = W.Sblk[t];
        }
        for (int t = tid; t < n; t += NT) rhs[t] = W.rhsg[t];
        __syncthreads();
        lap(4);
        // (rhs IS row n of the packed triangle; diag: 1 / L_kk; Lp: the reduction buffers are idle during the solve)
        const bool ok = win_cholesky_rhs(S, diag, ctl + 64, n, reinterpret_cast<int*>(ctl + 61), W.prof);
        lap(5);
        if (ok) {
          if (wave == 0) win_backsubstitute(S, diag, ctl + 64, rhs, n);
          __syncthreads();
          for (int i = tid; i < n; i += NT) W.x[i] = rhs[i];
        }
        if (tid == 0) { W.cl_ctl[0] = ok ? 1.0 : 0.0; W.cl_ctl[1] = (stop && *stop) ? 1.0 : 0.0; }
      }
      CL_BARRIER();
      ph = 7;
      const bool ok = W.cl_ctl[0] != 0.0;
      const size_t Fp = (size_t)W.Fp;
      if (ok) {      // W^T x_p of this workgroup's rows
        for (int p = C.g; p < kParts; p += G) {
          int lo, hi;
          part_range(W.F, p, lo, hi);
          for (int r = lo + tid; r < hi; r += NT) {
            const double* xp = W.x + 6 * W.f_cam[r];
            double c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
              const double xa = xp[a];
              c0 += W.Ws[(3 * a) * Fp + r] * xa; c1 += W.Ws[(3 * a + 1) * Fp + r] * xa; c2 += W.Ws[(3 * a + 2) * Fp + r] * xa;
            }
            W.Cs[r] = c0; W.Cs[Fp + r] = c1; W.Cs[2 * Fp + r] = c2;
          }
        }
      }
      CL_BARRIER();
      ph = 8;
      // landmarks of this workgroup: xl = Dinv (bl - sum of their rows' W^T x_p), the trial point, the scale terms; its cameras: oplus
      for (int p = C.g; p < kParts; p += G) {
        int lo, hi;
        part_range(W.nact, p, lo, hi);
        for (int li = lo + tid; li < hi; li += NT) {
          const double* hb = W.HB + kRowH * (size_t)li;
          double xl[3];
          if (ok) {
            double h[12], Di[9], d3[3];
#pragma unroll
            for (int j = 0; j < 12; j++) h[j] = hb[j];
            w_dinv(h, lambda, Di, d3);
            double c0 = h[9], c1 = h[10], c2 = h[11];
            // (the rows' W^T x_p come from the phase above: formed here, by the landmark's thread -- one phase and one barrier less -- the
            //  scattered W loads made this phase 46-54 us instead of 18 + 18)
            for (int q = W.pt_start[li]; q < W.pt_start[li + 1]; q++) {
              const int r = W.pt_edges[q];
              if (r >= W.F) break;
              c0 -= W.Cs[r]; c1 -= W.Cs[Fp + r]; c2 -= W.Cs[2 * Fp + r];
            }
            const double c[3] = {c0, c1, c2};
            w_mat3_vec(Di, c, xl);
#pragma unroll
            for (int a = 0; a < 3; a++) W.x[n + 3 * (size_t)li + a] = xl[a];
          } else {
#pragma unroll
            for (int a = 0; a < 3; a++) xl[a] = W.x[n + 3 * (size_t)li + a];     // (a failed solve: the last successful one's x)
          }
          const int l = W.act_pt[li];
#pragma unroll
          for (int a = 0; a < 3; a++) {
            pts_t[3 * (size_t)l + a] = pts[3 * (size_t)l + a] + xl[a];
            W.terms[n + 3 * li + a] = xl[a] * (lambda * xl[a] + hb[9 + a]);
          }
        }
        part_range(W.nfree, p, lo, hi);
        for (int i = lo + tid; i < hi; i += NT) {
          const int pp = W.free_pose[i];
          w_se3_oplus(poses + 7 * (size_t)pp, W.x + 6 * (size_t)i, poses_t + 7 * (size_t)pp);
#pragma unroll
          for (int a = 0; a < 6; a++) { const double xj = W.x[6 * i + a]; W.terms[6 * i + a] = xj * (lambda * xj + W.bp[6 * i + a]); }
        }
      }
      CL_BARRIER();
      ph = 9;
      cl_edge_pass<false>(W, C, poses_t, pts_t);
      cluster_partial_sums(W, C, W.e_rho, W.E, 0, red);
      cluster_partial_sums(W, C, W.terms, n + nl, 1, red);
      CL_BARRIER();
      ph = 11;
      // ---- the decision, taken by every thread of every workgroup on the same words (optimization_algorithm_levenberg.cpp:113-147)
      tempChi = ok ? cluster_total(W, 0) : 1.7976931348623157e308;
      rho = currentChi - tempChi;
      const double scale = cluster_total(W, 1) + 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - f64_cube(2 * rho - 1);
        alpha = (2. / 3. < alpha) ? 2. / 3. : alpha;
        lambda *= (1. / 3. < alpha) ? alpha : 1. / 3.;
        ni = 2;
        currentChi = tempChi;
        double* sw = poses; poses = poses_t; poses_t = sw;
        sw = pts; pts = pts_t; pts_t = sw;
      } else {
        lambda *= ni;
        ni *= 2;
      }
      qmax++;
      trials_total++;
      stopped = W.cl_ctl[1] != 0.0;
    } while (rho < 0 && qmax < 10 && !stopped);
    it_done++;
    chi_last = currentChi;
    if (leader && tid == 0 && it < 64) { st->trials_per_iter[it] = qmax; st->chi2_per_iter[it] = currentChi; st->lambda_per_iter[it] = lambda; }
    if (qmax == 10 || rho == 0) { stop_reason = 1; break; }
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) { stop_reason = 2; break; }
  }
  if (it_done == 0) {     // no iteration ran: the edges are evaluated at the unchanged input state (see k_ba_window)
    cl_edge_pass<false>(W, C, poses, pts);
    cluster_partial_sums(W, C, W.e_rho, W.E, 0, red);
    CL_BARRIER();
    chi_last = cluster_total(W, 0);
    if (leader && tid == 0) st->chi2_initial = chi_last;
  }
  // results: the accepted state, chi2 / depth signs back in the caller's edge order (every workgroup its parts; all inputs are final:
  // the last barrier is behind every write of the state and of chi_s)
  for (int p = C.g; p < kParts; p += G) {
    int lo, hi;
    part_range(7 * W.P, p, lo, hi);
    for (int i = lo + tid; i < hi; i += NT) W.out_poses[i] = poses[i];
    part_range(3 * W.L, p, lo, hi);
    for (int i = lo + tid; i < hi; i += NT) W.out_pts[i] = pts[i];
    part_range(W.E, p, lo, hi);
    for (int k = lo + tid; k < hi; k += NT) {
      const double* T = poses + 7 * (size_t)W.e_pose[k];
      const double* X = pts + 3 * (size_t)W.e_point[k];
      double R[9], Xc[3];
      w_quat_to_R(T + 3, R);
      w_mat3_vec(R, X, Xc);
      const int ko = W.e_orig[k];
      W.e_depth[ko] = (Xc[2] + T[2]) > 0.0 ? 1 : 0;
      W.e_chi2[ko] = W.chi_s[k];
    }
  }
  if (leader && tid == 0) {
    st->iterations = it_done; st->total_trials = trials_total; st->chi2_final = chi_last; st->lambda_final = lambda; st->stop_reason = stop_reason;
  }
  lap(11);
#undef CL_BARRIER
#undef s_flag
}

// evaluates csrc/f64_spec.h on the device (tests: the device build against the host build and the oracle's restatement)
__global__ void k_f64_spec(const double* __restrict__ x, int n, double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = f64_sin(x[i]); out[n + i] = f64_cos(x[i]); out[2 * n + i] = f64_cube(x[i]);
}

}  // namespace dvm

using namespace dvm;

namespace {

// host-side structure of one window: g2o's vertex ordering and incidence lists (sparse_optimizer.cpp:161-185, block_solver.hpp:143-295)
// and the chunk tables the kernel streams by
struct WinBuild {
  int P = 0, L = 0, E = 0, F = 0, nfree = 0, nact = 0, nblk = 0, C = 128, C2 = 256, n_sc = 0, n_hc = 0, stage_doubles = 0;
  std::vector<double> poses;                 // normalised quaternions (SE3Quat's constructor)
  std::vector<int32_t> pidx, lidx, free_pose, act_pt, e_pose, e_point, lpos, fpos, pt_start, f_start, f_cam;
  std::vector<int32_t> hc_ints, sc_desc, sc_ints, blk_ij;
  std::vector<double> e_obs, e_info;
  // the fast form's tables (the e_* arrays above are then in the camera-major order)
  int Fp = 0, Ep = 0;
  std::vector<int32_t> e_orig, e_lm, cam_start, pt_edges, bp_start, bp_pairs;   // bp_pairs: (row1, row2) interleaved
  // scratch of build_window_fast, kept with the object: a pooled WinBuild touches no fresh pages from its second use on (32 windows
  // built by 16 threads spent 2.3 ms of a 9 ms call in the page faults of their ~2 MB of vectors each)
  std::vector<int32_t> t_by_lm, t_ep, t_ept, t_cnt, t_blk_of, t_cr, t_fill;
  std::vector<double> t_eo, t_ei;
  // the fast form's tables packed into page-locked memory by the builder's own thread, while they are in its caches (pack_fast); the block
  // is one input of the call's staging, sent from where it lies
  int cluster_waves = 8;              // 8 G: set by the call before the build (the Schur pass's schedule is per wave of the cluster)
  std::vector<int32_t> sched_start, sched_task;
  PinnedArena arena;
  struct ArenaOff { ptrdiff_t poses, pidx, lidx, free_pose, act_pt, e_pose, e_point, e_obs, e_info, pt_start, f_cam, blk_ij, e_orig, e_lm, cam_start, pt_edges, bp_start, bp_pairs, sched_start, sched_task; } ao;
  ptrdiff_t shared_off = -1;          // >= 0: the tables lie at this offset of the call's shared block, not in `arena`
  size_t packed_bytes = 0;
  int pack_fast(PinnedArena* shared) {
    auto pad = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
    auto b4 = [&](const std::vector<int32_t>& v) { return pad(v.size() * 4); };
    auto b8 = [&](const std::vector<double>& v) { return pad(v.size() * 8); };
    const size_t total = b8(poses) + b4(pidx) + b4(lidx) + b4(free_pose) + b4(act_pt) + b4(e_pose) + b4(e_point) + b8(e_obs) + b8(e_info) + b4(pt_start) + b4(f_cam) +
                         b4(blk_ij) + b4(e_orig) + b4(e_lm) + b4(cam_start) + b4(pt_edges) + b4(bp_start) + b4(bp_pairs) + b4(sched_start) + b4(sched_task);
    shared_off = shared ? shared->claim(total) : -1;
    if (shared_off < 0) { const int rc = arena.reserve(total); if (rc != DVM_OK) return rc; }
    else arena.used = 0;
    // (a span of the shared block is filled through a view of it: base at the span, the same bump logic)
    uint8_t* const base = shared_off >= 0 ? shared->base + shared_off : arena.base;
    size_t used = 0;
    auto put = [&](const void* src, size_t bytes) -> ptrdiff_t {
      if (!bytes) return -1;
      const size_t off = used;
      std::memcpy(base + off, src, bytes);
      used = off + pad(bytes);
      return (ptrdiff_t)off;
    };
    auto p4 = [&](const std::vector<int32_t>& v) { return put(v.data(), v.size() * 4); };
    auto p8 = [&](const std::vector<double>& v) { return put(v.data(), v.size() * 8); };
    ao.poses = p8(poses); ao.pidx = p4(pidx); ao.lidx = p4(lidx); ao.free_pose = p4(free_pose); ao.act_pt = p4(act_pt); ao.e_pose = p4(e_pose); ao.e_point = p4(e_point);
    ao.e_obs = p8(e_obs); ao.e_info = p8(e_info); ao.pt_start = p4(pt_start); ao.f_cam = p4(f_cam); ao.blk_ij = p4(blk_ij); ao.e_orig = p4(e_orig); ao.e_lm = p4(e_lm);
    ao.cam_start = p4(cam_start); ao.pt_edges = p4(pt_edges); ao.bp_start = p4(bp_start); ao.bp_pairs = p4(bp_pairs);
    ao.sched_start = p4(sched_start); ao.sched_task = p4(sched_task);
    if (shared_off < 0) arena.used = used;
    packed_bytes = used;
    return DVM_OK;
  }
  void reset() {
    P = L = E = F = nfree = nact = nblk = 0; C = 128; C2 = 256; n_sc = n_hc = stage_doubles = 0; Fp = Ep = 0;
    for (auto* v : {&pidx, &lidx, &free_pose, &act_pt, &e_pose, &e_point, &lpos, &fpos, &pt_start, &f_start, &f_cam, &hc_ints, &sc_desc, &sc_ints, &blk_ij,
                    &e_orig, &e_lm, &cam_start, &pt_edges, &bp_start, &bp_pairs}) v->clear();
    poses.clear(); e_obs.clear(); e_info.clear();
  }
};

// The fast form's structure: edges sorted camera-major (free cameras in free-index order, a camera's edges by landmark, then the fixed
// cameras' edges), per landmark its edges in that order, per block of the reduced system its (row, row) pairs in landmark order
// (block_solver.hpp:381-439: edge k1 (outer) x edge k2 (inner), lower blocks; (k1, k1) on the diagonal).
int build_window_fast(WinBuild& b) {
  const int E = b.E, nf = b.nfree, P = b.P;
  // rank of an edge's camera: free index, or nf + pose for a fixed camera; two stable counting sorts: by landmark, then by rank
  std::vector<int32_t>& by_lm = b.t_by_lm; std::vector<int32_t>& order = b.e_orig;
  by_lm.resize(E); order.resize(E);
  {
    std::vector<int32_t>& cnt = b.t_cnt;
    cnt.assign(b.nact + 1, 0);
    for (int k = 0; k < E; k++) cnt[b.lidx[b.e_point[k]] + 1]++;
    for (int i = 0; i < b.nact; i++) cnt[i + 1] += cnt[i];
    for (int k = 0; k < E; k++) by_lm[cnt[b.lidx[b.e_point[k]]]++] = k;
    std::vector<int32_t>& cr = b.t_cr;
    cr.assign(nf + P + 1, 0);
    auto rank = [&](int k) { const int i = b.pidx[b.e_pose[k]]; return i >= 0 ? i : nf + b.e_pose[k]; };
    for (int k = 0; k < E; k++) cr[rank(k) + 1]++;
    for (int i = 0; i < nf + P; i++) cr[i + 1] += cr[i];
    b.cam_start.assign(cr.begin(), cr.begin() + nf + 1);
    for (int j = 0; j < E; j++) { const int k = by_lm[j]; order[cr[rank(k)]++] = k; }
  }
  b.F = nf ? b.cam_start[nf] : 0;
  b.Fp = (b.F + 63) & ~63; b.Ep = (E + 63) & ~63;
  std::vector<int32_t>& ep = b.t_ep; std::vector<int32_t>& ept = b.t_ept;
  std::vector<double>& eo = b.t_eo; std::vector<double>& ei = b.t_ei;
  ep.resize(E); ept.resize(E); eo.resize(2 * (size_t)E); ei.resize(E);
  b.e_lm.resize(E);
  for (int j = 0; j < E; j++) {
    const int k = order[j];
    ep[j] = b.e_pose[k]; ept[j] = b.e_point[k]; eo[2 * (size_t)j] = b.e_obs[2 * (size_t)k]; eo[2 * (size_t)j + 1] = b.e_obs[2 * (size_t)k + 1]; ei[j] = b.e_info[k];
    b.e_lm[j] = b.lidx[ept[j]];
  }
  b.e_pose.swap(ep); b.e_point.swap(ept); b.e_obs.swap(eo); b.e_info.swap(ei);
  b.f_cam.resize(b.F);
  for (int i = 0; i < nf; i++) for (int j = b.cam_start[i]; j < b.cam_start[i + 1]; j++) b.f_cam[j] = i;
  // a landmark's edges, ascending sorted index (its free rows first)
  b.pt_start.assign(b.nact + 1, 0);
  for (int j = 0; j < E; j++) b.pt_start[b.e_lm[j] + 1]++;
  for (int i = 0; i < b.nact; i++) b.pt_start[i + 1] += b.pt_start[i];
  b.pt_edges.resize(E);
  { std::vector<int32_t>& fill = b.t_fill;
    fill.assign(b.pt_start.begin(), b.pt_start.end() - 1);
    for (int j = 0; j < E; j++) b.pt_edges[fill[b.e_lm[j]]++] = j; }
  // blocks and their pairs.  Within a landmark the free rows ascend with the camera: (q1, q2) with camera(q2) <= camera(q1) is q2 <= q1,
  // minus the off-diagonal pairs of two observations by the SAME camera (as the sequential-order form leaves them out)
  std::vector<int32_t>& blk_of = b.t_blk_of; std::vector<int32_t>& cnt = b.t_cnt;
  blk_of.assign((size_t)nf * nf, -1); cnt.clear();
  auto each_pair = [&](auto&& fn) {
    for (int li = 0; li < b.nact; li++) {
      const int q0 = b.pt_start[li];
      int qe = q0;
      while (qe < b.pt_start[li + 1] && b.pt_edges[qe] < b.F) qe++;
      for (int a1 = q0; a1 < qe; a1++)
        for (int a2 = q0; a2 <= a1; a2++) {
          const int r1 = b.pt_edges[a1], r2 = b.pt_edges[a2], i1 = b.f_cam[r1], i2 = b.f_cam[r2];
          if (i1 == i2 && r1 != r2) continue;
          fn(i1, i2, r1, r2);
        }
    }
  };
  each_pair([&](int i1, int i2, int, int) {
    int32_t& id = blk_of[(size_t)i1 * nf + i2];
    if (id < 0) { id = (int32_t)b.blk_ij.size(); b.blk_ij.push_back(i1 | (i2 << 8)); cnt.push_back(0); }
    cnt[id]++;
  });
  b.nblk = (int)b.blk_ij.size();
  b.bp_start.assign(b.nblk + 1, 0);
  for (int j = 0; j < b.nblk; j++) b.bp_start[j + 1] = b.bp_start[j] + cnt[j];
  b.bp_pairs.resize(2 * (size_t)b.bp_start[b.nblk]);
  { std::vector<int32_t>& fill = b.t_fill;
    fill.assign(b.bp_start.begin(), b.bp_start.end() - 1);
    each_pair([&](int i1, int i2, int r1, int r2) { const int at = fill[blk_of[(size_t)i1 * nf + i2]]++; b.bp_pairs[2 * (size_t)at] = r1; b.bp_pairs[2 * (size_t)at + 1] = r2; }); }
  // the Schur pass's schedule: tasks = the nf right-hand sides (a camera's rows, six sums) and the blocks (a block's pairs, 36 sums), each
  // done by one wave; cost ~ its rounds of 64 rows / pairs times the loads of a round, plus the reduction; longest first onto the least
  // loaded wave (ties: the lowest wave)
  {
    const int GW = std::max(b.cluster_waves, 1), nt = nf + b.nblk;
    std::vector<std::pair<int32_t, int32_t>> task(nt);          // (cost, id)
    // (weights per round 6 / 36 = the loads of a round, 25 / 100 for a task's fixed part: kernel 2.92 -> 2.78 ms for 32 windows; other
    //  weights -- diagonal blocks' coalesced rounds at 12 ... 80, the fixed part at 50 ... 300 -- measured within +-0.05 ms of it or worse)
    for (int i = 0; i < nf; i++) task[i] = {6 * ((b.cam_start[i + 1] - b.cam_start[i] + 63) / 64) + 25, i};
    for (int j = 0; j < b.nblk; j++) task[nf + j] = {36 * ((b.bp_start[j + 1] - b.bp_start[j] + 63) / 64) + 100, nf + j};
    std::stable_sort(task.begin(), task.end(), [](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) { return x.first > y.first; });
    std::vector<int64_t> load(GW, 0);
    std::vector<int32_t> owner(nt, 0), count(GW + 1, 0);
    for (int t = 0; t < nt; t++) {
      int w = 0;
      for (int k = 1; k < GW; k++) if (load[k] < load[w]) w = k;
      load[w] += task[t].first; owner[t] = w; count[w + 1]++;
    }
    for (int k = 0; k < GW; k++) count[k + 1] += count[k];
    b.sched_start.assign(count.begin(), count.end());
    b.sched_task.assign(std::max(nt, 1), 0);
    std::vector<int32_t> fill(count.begin(), count.end() - 1);
    for (int t = 0; t < nt; t++) b.sched_task[fill[owner[t]]++] = task[t].second;
  }
  b.n_hc = b.n_sc = 0;
  // the waves' reduction buffers; during the solve the same area (from ctl + 64 on: 256 doubles of tree partials first) holds the panel copy
  // of win_cholesky_rhs, (n + 1) rows of kLpPitch doubles
  b.stage_doubles = std::max((kWinThreads / 64) * 4 * 36, kLpPitch * (6 * nf + 1) - 256 + 8);
  return DVM_OK;
}

int build_window(const dvm_ba_window& w, WinBuild& b, bool normalize, bool fast = false) {
  const int P = w.n_poses, L = w.n_points, E = w.n_edges;
  if (P < 0 || L < 0 || E < 0 || (P && (!w.poses || !w.fixed)) || (L && !w.points) || (E && !w.edges)) { set_error("dvm_ba_optimize_windows: null array"); return DVM_ERR_INVALID; }
  b.reset();
  b.P = P; b.L = L; b.E = E;
  b.poses.assign(w.poses, w.poses + 7 * (size_t)P);
  for (int p = 0; normalize && p < P; p++) {   // quat_normalize as the oracle / g2o::SE3Quat(q, t) does it: sign, then divide by the norm
    double* q = &b.poses[7 * (size_t)p + 3];
    if (q[3] < 0) for (int i = 0; i < 4; i++) q[i] = -q[i];
    const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= nrm;
  }
  std::vector<uint8_t> pose_used(P, 0), pt_used(L, 0);
  b.e_pose.resize(E); b.e_point.resize(E); b.e_obs.resize(2 * (size_t)E); b.e_info.resize(E);
  for (int k = 0; k < E; k++) {
    const dvm_ba_edge& e = w.edges[k];
    if (e.pose < 0 || e.pose >= P || e.point < 0 || e.point >= L) { set_error("dvm_ba_optimize_windows: edge index out of range"); return DVM_ERR_INVALID; }
    b.e_pose[k] = e.pose; b.e_point[k] = e.point; b.e_obs[2 * (size_t)k] = e.u; b.e_obs[2 * (size_t)k + 1] = e.v; b.e_info[k] = e.inv_sigma2;
    pose_used[e.pose] = 1; pt_used[e.point] = 1;
  }
  b.pidx.assign(P, -1); b.lidx.assign(L, -1);
  for (int p = 0; p < P; p++) if (!w.fixed[p] && pose_used[p]) { b.pidx[p] = b.nfree++; b.free_pose.push_back(p); }
  for (int l = 0; l < L; l++) if (pt_used[l]) { b.lidx[l] = b.nact++; b.act_pt.push_back(l); }
  if (b.nfree > kWinMaxFree) { set_error("dvm_ba_optimize_windows: more than 30 free cameras in one window (use dvm_ba_optimize)"); return DVM_ERR_CAPACITY; }
  if (b.nfree > 24) { b.C = 32; b.C2 = 64; }              // the packed reduced system takes up to 130 KB of the 160: smaller streaming chunks (else 128 / 256)
  if (fast) return build_window_fast(b);
  const int nf = b.nfree;
  // landmark-major order of the edges (a landmark's edges in input order): rowA's order; its free-camera subsequence: rowW's order
  b.pt_start.assign(b.nact + 1, 0);
  for (int k = 0; k < E; k++) b.pt_start[b.lidx[b.e_point[k]] + 1]++;
  for (int i = 0; i < b.nact; i++) b.pt_start[i + 1] += b.pt_start[i];
  std::vector<int32_t> pt_edges(E);
  b.lpos.resize(E);
  {
    std::vector<int32_t> pc(b.pt_start.begin(), b.pt_start.end() - 1);
    for (int k = 0; k < E; k++) { const int q = pc[b.lidx[b.e_point[k]]]++; pt_edges[q] = k; b.lpos[k] = q; }
  }
  b.fpos.assign(E, -1); b.f_start.assign(b.nact + 1, 0);
  std::vector<int32_t> f_edge;
  int maxdeg = 0;
  for (int li = 0; li < b.nact; li++) {
    for (int q = b.pt_start[li]; q < b.pt_start[li + 1]; q++) {
      const int k = pt_edges[q], i = b.pidx[b.e_pose[k]];
      if (i < 0) continue;
      b.fpos[k] = (int32_t)f_edge.size(); f_edge.push_back(k); b.f_cam.push_back(i);
    }
    b.f_start[li + 1] = (int32_t)f_edge.size();
    maxdeg = std::max(maxdeg, b.f_start[li + 1] - b.f_start[li]);
  }
  b.F = (int)f_edge.size();
  if (maxdeg > b.C) { set_error("dvm_ba_optimize_windows: a landmark with more free-camera observations than a streaming chunk holds (duplicate observations?)"); return DVM_ERR_CAPACITY; }
  // Hessian chunks: C2 consecutive edges; the free-camera ones among them listed per camera (ascending edge = the camera's own order)
  b.n_hc = nf ? (E + b.C2 - 1) / b.C2 : 0;
  b.hc_ints.assign((size_t)b.n_hc * (nf + 1 + b.C2), 0);
  for (int c = 0; c < b.n_hc; c++) {
    const int k0 = c * b.C2, k1 = std::min(E, k0 + b.C2);
    int32_t* rg = &b.hc_ints[(size_t)c * (nf + 1 + b.C2)];       // [range (nf + 1) | row slots, camera-major (<= C2)]
    for (int k = k0; k < k1; k++) { const int i = b.pidx[b.e_pose[k]]; if (i >= 0) rg[i + 1]++; }
    for (int i = 0; i < nf; i++) rg[i + 1] += rg[i];
    std::vector<int32_t> fill(rg, rg + nf);
    for (int k = k0; k < k1; k++) { const int i = b.pidx[b.e_pose[k]]; if (i >= 0) rg[nf + 1 + fill[i]++] = k - k0; }
  }
  // Schur chunks: whole landmarks -- at most C of them, at most C free rows, an int block of at most kWinIntCap --; per chunk the rows by
  // camera, the landmark of every row, and the (edge, edge) pairs of block_solver.hpp:381-439 -- landmark by landmark, per landmark edge k1
  // (outer) x edge k2 (inner), lower blocks only, (k1, k1) on the diagonal -- sorted by block (stable: a block's pairs stay in landmark order)
  std::vector<int32_t> blk_of((size_t)nf * nf, -1);
  int max_ints = 0;
  for (int li = 0; li < b.nact;) {
    const int r0 = b.f_start[li];
    int lj = li, npairs = 0;
    while (lj < b.nact && lj - li < b.C && b.f_start[lj + 1] - r0 <= b.C) {
      const int deg = b.f_start[lj + 1] - b.f_start[lj];
      const int np_new = npairs + deg * (deg + 1) / 2 + deg * (deg - 1) / 2 * 0;    // lower blocks incl. the diagonal: one pair per (k1, k2) with i2 <= i1
      const int rows_new = b.f_start[lj + 1] - r0;
      if (lj > li && nf + 1 + 2 * rows_new + 2 * (np_new + deg * deg) + 1 > kWinIntCap) break;   // (generous: duplicates of a camera add pairs)
      npairs = np_new; lj++;
    }
    if (lj == li) { set_error("dvm_ba_optimize_windows: a landmark does not fit a streaming chunk"); return DVM_ERR_CAPACITY; }
    const int r1 = b.f_start[lj], nr = r1 - r0;
    const int32_t ioff = (int32_t)b.sc_ints.size();
    b.sc_ints.resize(ioff + nf + 1 + 2 * nr, 0);
    int32_t* rg = &b.sc_ints[ioff];
    for (int r = r0; r < r1; r++) rg[b.f_cam[r] + 1]++;
    for (int i = 0; i < nf; i++) rg[i + 1] += rg[i];
    {
      std::vector<int32_t> fill(rg, rg + nf);
      for (int r = r0; r < r1; r++) b.sc_ints[ioff + nf + 1 + fill[b.f_cam[r]]++] = r - r0;
    }
    for (int l2 = li; l2 < lj; l2++)
      for (int r = b.f_start[l2]; r < b.f_start[l2 + 1]; r++) b.sc_ints[ioff + nf + 1 + nr + (r - r0)] = l2 - li;
    std::vector<std::pair<int32_t, int32_t>> pr;       // (block, slot1 | slot2 << 16)
    for (int l2 = li; l2 < lj; l2++)
      for (int q1 = b.f_start[l2]; q1 < b.f_start[l2 + 1]; q1++)
        for (int q2 = b.f_start[l2]; q2 < b.f_start[l2 + 1]; q2++) {
          const int i1 = b.f_cam[q1], i2 = b.f_cam[q2];
          if (i2 > i1 || (i2 == i1 && q2 != q1)) continue;
          int32_t& id = blk_of[(size_t)i1 * nf + i2];
          if (id < 0) { id = (int32_t)b.blk_ij.size(); b.blk_ij.push_back(i1 | (i2 << 8)); }
          pr.push_back({id, (q1 - r0) | ((q2 - r0) << 16)});
        }
    std::stable_sort(pr.begin(), pr.end(), [](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) { return x.first < y.first; });
    int nruns = 0;
    for (size_t j = 0; j < pr.size(); j++)
      if (j == 0 || pr[j].first != pr[j - 1].first) { b.sc_ints.push_back(b.blk_ij[pr[j].first] | ((int32_t)j << 16)); nruns++; }   // (i1 | i2 << 8 | first pair << 16)
    b.sc_ints.push_back((int32_t)pr.size() << 16);       // sentinel: where the last run ends
    for (const auto& q : pr) b.sc_ints.push_back(q.second);
    const int ni = (int)b.sc_ints.size() - ioff;
    if (ni > kWinIntCap || pr.size() >= 65536) { set_error("dvm_ba_optimize_windows: the index block of a chunk exceeds its capacity"); return DVM_ERR_CAPACITY; }
    max_ints = std::max(max_ints, ni);
    const int32_t d[8] = {r0, nr, ioff, ni, nruns + 1, (int32_t)pr.size(), li, lj - li};
    b.sc_desc.insert(b.sc_desc.end(), d, d + 8);
    li = lj;
  }
  b.n_sc = (int)b.sc_desc.size() / 8;
  b.nblk = (int)b.blk_ij.size();
  const size_t hess = (size_t)b.C2 * kRowB + ((size_t)b.C2 + nf + 1 + 1) / 2;
  const size_t schur = (size_t)b.C * (kRowW + kRowD + kRowH) + ((size_t)max_ints + 1) / 2;
  b.stage_doubles = (int)std::max(hess, schur) + 2;
  return DVM_OK;
}

struct StopWord {                      // a word of page-locked host memory the kernel polls; the host copies the caller's flag into it
  int* h = nullptr; int* d = nullptr;
  ~StopWord() { if (h) hipHostFree(h); }
  int ensure() {
    if (h) return DVM_OK;
    int rc = hip_check(hipHostMalloc(reinterpret_cast<void**>(&h), 64, hipHostMallocMapped), "hipHostMalloc(stop word)");
    if (rc != DVM_OK) { h = nullptr; return rc; }
    return hip_check(hipHostGetDevicePointer(reinterpret_cast<void**>(&d), h, 0), "hipHostGetDevicePointer");
  }
};

}  // namespace

int dvm_ba_optimize_windows_impl(int device, const dvm_ba_window* windows, int K, const volatile uint8_t* stop_flag, dvm_ba_stats* stats, bool normalize_input, bool fast, int cluster) {
  if (K < 0 || (K && !windows)) { set_error("dvm_ba_optimize_windows: null windows"); return DVM_ERR_INVALID; }
  if (K == 0) return DVM_OK;
  int rc = dvm_set_device(device);
  if (rc != DVM_OK) return rc;
  const auto t0 = std::chrono::steady_clock::now();
  // the cluster size: the largest of 8 / 4 / 2 / 1 workgroups per window whose grid (8 * ceil(K / 8) * G workgroups, one per CU: the
  // reduced system's LDS) is resident at once; forced by `cluster` (the G = 1 repeat after a barrier time-out) or DVM_BA_CLUSTER.  Chosen
  // before the builds: the windows' Schur schedules are per wave of the cluster.
  int G = 1;
  if (fast) {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    const char* env_c = std::getenv("DVM_BA_CLUSTER");      // (read at every call: the tests walk the cluster sizes)
    const int env_g = env_c ? atoi(env_c) : 0;
    const int want = cluster > 0 ? cluster : env_g;
    const int groups = 8 * ((K + 7) / 8);
    for (int g = 8; g >= 1; g >>= 1) if ((want > 0 && g == want) || (want <= 0 && groups * g <= cus)) { G = g; break; }
  }
  // (the fast form keeps its windows' host tables in a per-thread pool: see WinBuild's scratch members)
  static thread_local std::vector<WinBuild> build_pool;
  std::vector<WinBuild> local_builds;
  if (fast) { if ((int)build_pool.size() < K) build_pool.resize(K); } else local_builds.resize(K);
  std::vector<WinBuild>& B = fast ? build_pool : local_builds;
  if (fast) for (int k = 0; k < K; k++) B[k].cluster_waves = (kWinThreads / 64) * G;
  // the fast form's tables of ALL windows go into one page-locked block (one DMA): sized before the builds from a generous estimate --
  // ~66 bytes per edge are typical --; a window that finds it full packs into a block of its own
  static thread_local PinnedArena shared_tables_tls;
  // ... and travel to a device block of the same size, every window's span the moment its builder has packed it (from the builder's thread,
  // on one of four upload streams: the copies of the first windows run under the builds of the last, and side by side -- one 32 MB copy
  // behind the builds took 1.4 ms, 23 GB/s); the launch's stream waits for the four streams' events
  struct TableUpload {
    hipStream_t s[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    uint8_t* dev = nullptr; size_t dcap = 0; int device = -1;
    void release() {
      for (int i = 0; i < 4; i++) { if (s[i]) { hipStreamSynchronize(s[i]); hipStreamDestroy(s[i]); } if (e[i]) hipEventDestroy(e[i]); s[i] = nullptr; e[i] = nullptr; }
      if (dev) hipFree(dev);
      dev = nullptr; dcap = 0;
    }
    int ensure(int dev_id, size_t bytes) {
      if (dev_id != device) { release(); device = dev_id; }
      for (int i = 0; i < 4; i++) {
        if (!s[i]) { int rc = hip_check(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking), "stream"); if (rc != DVM_OK) return rc; }
        if (!e[i]) { int rc = hip_check(hipEventCreateWithFlags(&e[i], hipEventDisableTiming), "event"); if (rc != DVM_OK) return rc; }
      }
      if (bytes > dcap) {
        if (dev) hipFree(dev);
        dev = nullptr; dcap = 0;
        int rc = hip_check(hipMalloc(reinterpret_cast<void**>(&dev), bytes), "hipMalloc(window tables)");
        if (rc != DVM_OK) return rc;
        dcap = bytes;
      }
      return DVM_OK;
    }
    ~TableUpload() { release(); }
  };
  static thread_local TableUpload tup_tls;
  // (the builders' threads must see THIS thread's block and streams: a thread_local named inside their lambda would be their own)
  PinnedArena& shared_tables = shared_tables_tls;
  TableUpload& tup = tup_tls;
  if (fast) {
    size_t est = 0;
    for (int k = 0; k < K; k++) est += 80 * (size_t)std::max(windows[k].n_poses, 0) + 16 * (size_t)std::max(windows[k].n_points, 0) + 96 * (size_t)std::max(windows[k].n_edges, 0) + (64 << 10);
    if ((rc = shared_tables.reserve(est)) != DVM_OK) return rc;
    if ((rc = tup.ensure(device, shared_tables.cap)) != DVM_OK) return rc;
  }
  const bool private_tables = fast && std::getenv("DVM_BA_TEST_PRIVATE_TABLES") != nullptr;   // (test hook: every second window packs into a block of its own, as one that finds the shared block full does)
  auto send_tables = [&](int k) -> int {        // window k's span of the shared block -> the device block, asynchronously
    const WinBuild& b = B[k];
    if (b.shared_off < 0 || !b.packed_bytes) return (int)DVM_OK;
    return hip_check(hipMemcpyAsync(tup.dev + b.shared_off, shared_tables.base + b.shared_off, b.packed_bytes, hipMemcpyHostToDevice, tup.s[k & 3]), "upload(window tables)");
  };
  if (fast && K > 1) {
    // the windows' index tables are independent: built by up to 16 pooled host threads (0.2 ms each; 32 of them one after the other would cost
    // more than the launch that solves them)
    std::vector<int> rcs(K, DVM_OK);
    std::vector<std::string> errs(K);
    static const int bt = std::getenv("DVM_BA_BUILD_THREADS") ? std::max(1, atoi(std::getenv("DVM_BA_BUILD_THREADS"))) : 32;
    HostPool::get().run((size_t)K, bt, [&](size_t k) {
      rcs[k] = build_window(windows[k], B[k], normalize_input, true);
      if (rcs[k] == DVM_OK) { hipSetDevice(device); rcs[k] = B[k].pack_fast((private_tables && (k & 1)) ? nullptr : &shared_tables); }
      if (rcs[k] == DVM_OK) rcs[k] = send_tables((int)k);     // (a pooled thread: the device of the call, for the arena's first allocation)
      if (rcs[k] != DVM_OK) errs[k] = last_error_cstr();
    });
    for (int k = 0; k < K; k++) if (rcs[k] != DVM_OK) { set_error(errs[k]); return rcs[k]; }
  } else {
    for (int k = 0; k < K; k++) {
      if ((rc = build_window(windows[k], B[k], normalize_input, fast)) != DVM_OK) return rc;
      if (fast && ((rc = B[k].pack_fast(&shared_tables)) != DVM_OK || (rc = send_tables(k)) != DVM_OK)) return rc;
    }
  }
  const auto t1 = std::chrono::steady_clock::now();
  if (fast && std::getenv("DVM_BA_WINDOW_TIMING")) {     // (measurement only: how long the tables' copies run on behind the builds)
    for (int i = 0; i < 4; i++) hipStreamSynchronize(tup.s[i]);
    std::fprintf(stderr, "tables: %.1f MB, copies done %.2f ms behind the builds\n", shared_tables.top() / 1048576.0, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  }
  thread_local StopWord sw;
  if ((rc = sw.ensure()) != DVM_OK) return rc;
  *sw.h = (stop_flag && *stop_flag) ? 1 : 0;

  Stage st;
  struct Slots { int poses, pts, pidx, lidx, free_pose, act_pt, e_pose, e_point, e_obs, e_info, lpos, fpos, pt_start, f_start, f_cam, hc_ints, sc_desc, sc_ints,
                     blk_ij, out_poses, out_pts, echi, edepth, stats, poses_t, pts_t, rowB, rowA, rowW, erho, Hpp, bp, HB, DD, x, terms, prof,
                     e_orig, e_lm, cam_start, pt_edges, bp_start, bp_pairs, Bs, Ws, Ts, Cs, chi_s, cl_sync, Sblk, rhsg, cl_part, cl_ctl; };
  std::vector<Slots> sl(K);
  static const bool want_prof = std::getenv("DVM_BA_WINDOW_PROF") != nullptr;
  std::vector<std::vector<unsigned long long>> prof_out(K, std::vector<unsigned long long>(16, 0));
  std::vector<unsigned long long> prof_zero(16, 0);
  auto I32 = [&](const std::vector<int32_t>& v) { return st.in(v.empty() ? nullptr : v.data(), v.size() * 4); };
  auto F64 = [&](const std::vector<double>& v) { return st.in(v.empty() ? nullptr : v.data(), v.size() * 8); };
  std::vector<int> arena_slot(K, -1);
  if (fast) for (int i = 0; i < 4; i++) DVM_HIP(hipEventRecord(tup.e[i], tup.s[i]));
  for (int k = 0; k < K; k++) {                     // inputs: the state and g2o's graph structure as index arrays
    const WinBuild& b = B[k]; Slots& s = sl[k];
    if (fast) {     // the window's tables: already page-locked (WinBuild::pack_fast); the landmarks from the caller's array
      if (b.shared_off < 0) arena_slot[k] = st.in_pinned(b.arena.used ? b.arena.base : nullptr, b.arena.used);
      s.pts = st.in(b.L ? windows[k].points : nullptr, 24 * (size_t)b.L);
      continue;
    }
    s.poses = F64(b.poses);
    s.pts = st.in(b.L ? windows[k].points : nullptr, 24 * (size_t)b.L);
    s.pidx = I32(b.pidx); s.lidx = I32(b.lidx); s.free_pose = I32(b.free_pose); s.act_pt = I32(b.act_pt); s.e_pose = I32(b.e_pose); s.e_point = I32(b.e_point);
    s.e_obs = F64(b.e_obs); s.e_info = F64(b.e_info);
    s.lpos = I32(b.lpos); s.fpos = I32(b.fpos); s.pt_start = I32(b.pt_start); s.f_start = I32(b.f_start); s.f_cam = I32(b.f_cam);
    s.hc_ints = I32(b.hc_ints); s.sc_desc = I32(b.sc_desc); s.sc_ints = I32(b.sc_ints); s.blk_ij = I32(b.blk_ij);
    s.e_orig = I32(b.e_orig); s.e_lm = I32(b.e_lm); s.cam_start = I32(b.cam_start); s.pt_edges = I32(b.pt_edges);
    s.bp_start = I32(b.bp_start); s.bp_pairs = I32(b.bp_pairs);
  }
  std::vector<uint32_t> sync_zero(16, 0u);        // the cluster form's arrival counter [0] and time-out word [8], zero at every launch
  std::vector<std::vector<uint32_t>> sync_back(K, std::vector<uint32_t>(16, 0u));
  if (fast) for (int k = 0; k < K; k++) sl[k].cl_sync = st.add(sync_zero.data(), sync_back[k].data(), 64);
  std::vector<BaWin> views(K);
  const int views_slot = st.in(views.data(), sizeof(BaWin) * (size_t)K);   // filled in below, once layout() has placed everything
  struct Outs { dvm_ba_stats st; };
  std::vector<Outs> outs(K);
  for (int k = 0; k < K; k++) {                     // outputs (one contiguous span to fetch), straight into the caller's arrays where it gave any
    const WinBuild& b = B[k]; Slots& s = sl[k]; Outs& o = outs[k];
    const dvm_ba_window& w = windows[k];
    s.out_poses = (w.poses_out && b.P) ? st.out(w.poses_out, 56 * (size_t)b.P) : st.scratch(56 * (size_t)b.P);
    s.out_pts = (w.points_out && b.L) ? st.out(w.points_out, 24 * (size_t)b.L) : st.scratch(24 * (size_t)b.L);
    s.echi = (w.edge_chi2_out && b.E) ? st.out(w.edge_chi2_out, 8 * (size_t)b.E) : st.scratch(8 * (size_t)b.E);
    s.edepth = (w.depth_positive_out && b.E) ? st.out(w.depth_positive_out, (size_t)b.E) : st.scratch((size_t)b.E);
    s.stats = st.out(&o.st, sizeof(dvm_ba_stats));
    s.prof = want_prof ? st.add(prof_zero.data(), prof_out[k].data(), 128) : -1;
  }
  for (int k = 0; k < K; k++) {                     // working memory
    const WinBuild& b = B[k]; Slots& s = sl[k];
    const size_t E = b.E, F = b.F, n = 6 * (size_t)b.nfree, nl = 3 * (size_t)b.nact;
    s.poses_t = st.scratch(56 * (size_t)b.P); s.pts_t = st.scratch(24 * (size_t)b.L);
    s.rowB = fast ? -1 : st.scratch(8 * kRowB * E); s.rowA = st.scratch(8 * kRowA * E); s.rowW = fast ? -1 : st.scratch(8 * kRowW * F);
    s.erho = st.scratch(8 * E);
    s.Hpp = st.scratch(288 * (size_t)b.nfree); s.bp = st.scratch(8 * n); s.HB = st.scratch(8 * kRowH * (size_t)b.nact); s.DD = st.scratch(8 * kRowH * (size_t)b.nact);
    s.x = st.scratch(8 * (n + nl)); s.terms = st.scratch(8 * (n + nl));
    if (fast) {
      const size_t Fp = (size_t)b.Fp;
      s.Bs = st.scratch(8 * 15 * Fp); s.Ws = st.scratch(8 * 18 * Fp); s.Ts = st.scratch(8 * 24 * Fp); s.Cs = st.scratch(8 * 3 * Fp); s.chi_s = st.scratch(8 * E);
      s.Sblk = st.scratch(8 * 36 * (size_t)std::max(b.nblk, 1)); s.rhsg = st.scratch(8 * std::max<size_t>(n, 1)); s.cl_part = st.scratch(8 * 4 * 8); s.cl_ctl = st.scratch(8 * 4);
    }
  }
  const auto t1a = std::chrono::steady_clock::now();
  if ((rc = st.layout()) != DVM_OK) return rc;
  const auto t1b = std::chrono::steady_clock::now();
  size_t lds_doubles = 0;
  for (int k = 0; k < K; k++) {
    const WinBuild& b = B[k]; const Slots& s = sl[k]; BaWin& v = views[k];
    std::memset(&v, 0, sizeof(v));
    v.P = b.P; v.L = b.L; v.E = b.E; v.F = b.F; v.nfree = b.nfree; v.nact = b.nact; v.nblk = b.nblk; v.iterations = windows[k].iterations;
    v.C = b.C; v.C2 = b.C2; v.n_sc = b.n_sc; v.n_hc = b.n_hc; v.stage_doubles = b.stage_doubles;
    v.fx = windows[k].cam.fx; v.fy = windows[k].cam.fy; v.cx = windows[k].cam.cx; v.cy = windows[k].cam.cy; v.delta = windows[k].cam.huber_delta;
    v.pts = st.ptr<double>(s.pts); v.poses_t = st.ptr<double>(s.poses_t); v.pts_t = st.ptr<double>(s.pts_t);
    v.out_poses = st.ptr<double>(s.out_poses); v.out_pts = st.ptr<double>(s.out_pts);
    uint8_t* const ab = !fast ? nullptr : b.shared_off >= 0 ? tup.dev + b.shared_off : st.ptr<uint8_t>(arena_slot[k]);   // the window's tables on the device
    auto A4 = [&](ptrdiff_t off) { return off < 0 ? nullptr : reinterpret_cast<int32_t*>(ab + off); };
    auto A8 = [&](ptrdiff_t off) { return off < 0 ? nullptr : reinterpret_cast<double*>(ab + off); };
    if (fast) {
      v.poses = A8(b.ao.poses); v.pidx = A4(b.ao.pidx); v.lidx = A4(b.ao.lidx); v.free_pose = A4(b.ao.free_pose); v.act_pt = A4(b.ao.act_pt);
      v.e_pose = A4(b.ao.e_pose); v.e_point = A4(b.ao.e_point); v.e_obs = A8(b.ao.e_obs); v.e_info = A8(b.ao.e_info);
      v.pt_start = A4(b.ao.pt_start); v.f_cam = A4(b.ao.f_cam);
    } else {
    v.poses = st.ptr<double>(s.poses);
    v.pidx = st.ptr<int32_t>(s.pidx); v.lidx = st.ptr<int32_t>(s.lidx); v.free_pose = st.ptr<int32_t>(s.free_pose); v.act_pt = st.ptr<int32_t>(s.act_pt);
    v.e_pose = st.ptr<int32_t>(s.e_pose); v.e_point = st.ptr<int32_t>(s.e_point); v.e_obs = st.ptr<double>(s.e_obs); v.e_info = st.ptr<double>(s.e_info);
    v.lpos = st.ptr<int32_t>(s.lpos); v.fpos = st.ptr<int32_t>(s.fpos); v.pt_start = st.ptr<int32_t>(s.pt_start); v.f_start = st.ptr<int32_t>(s.f_start);
    v.f_cam = st.ptr<int32_t>(s.f_cam);
    }
    v.rowB = fast ? nullptr : st.ptr<double>(s.rowB); v.rowA = st.ptr<double>(s.rowA); v.rowW = fast ? nullptr : st.ptr<double>(s.rowW);
    v.e_chi2 = st.ptr<double>(s.echi); v.e_rho = st.ptr<double>(s.erho); v.e_depth = st.ptr<uint8_t>(s.edepth);
    if (!fast) { v.hc_ints = st.ptr<int32_t>(s.hc_ints); v.sc_desc = st.ptr<int32_t>(s.sc_desc); v.sc_ints = st.ptr<int32_t>(s.sc_ints); v.blk_ij = st.ptr<int32_t>(s.blk_ij); }
    if (fast) {
      v.Fp = b.Fp; v.Ep = b.Ep;
      v.blk_ij = A4(b.ao.blk_ij);
      v.e_orig = A4(b.ao.e_orig); v.e_lm = A4(b.ao.e_lm); v.cam_start = A4(b.ao.cam_start); v.pt_edges = A4(b.ao.pt_edges);
      v.bp_start = A4(b.ao.bp_start); v.bp_pairs = reinterpret_cast<const int2*>(A4(b.ao.bp_pairs));
      v.sched_start = A4(b.ao.sched_start); v.sched_task = A4(b.ao.sched_task);
      v.Bs = st.ptr<double>(s.Bs); v.Ws = st.ptr<double>(s.Ws); v.Ts = st.ptr<double>(s.Ts); v.Cs = st.ptr<double>(s.Cs); v.chi_s = st.ptr<double>(s.chi_s);
      v.cl_ctr = st.ptr<unsigned int>(s.cl_sync); v.cl_tmo = v.cl_ctr + 8;
      v.Sblk = st.ptr<double>(s.Sblk); v.rhsg = st.ptr<double>(s.rhsg); v.cl_part = st.ptr<double>(s.cl_part); v.cl_ctl = st.ptr<double>(s.cl_ctl);
    }
    v.Hpp = st.ptr<double>(s.Hpp); v.bp = st.ptr<double>(s.bp); v.HB = st.ptr<double>(s.HB); v.DD = st.ptr<double>(s.DD);
    v.x = st.ptr<double>(s.x); v.terms = st.ptr<double>(s.terms);
    v.stats = st.ptr<dvm_ba_stats>(s.stats);
    v.prof = s.prof >= 0 ? st.ptr<unsigned long long>(s.prof) : nullptr;
    const size_t n = 6 * (size_t)b.nfree;
    lds_doubles = std::max(lds_doubles, n * (n + 1) / 2 + 2 * n + 64 + 256 + (size_t)b.stage_doubles);
  }
  const auto t1c = std::chrono::steady_clock::now();
  if ((rc = st.upload()) != DVM_OK) return rc;
  if (fast) for (int i = 0; i < 4; i++) DVM_HIP(hipStreamWaitEvent(st.stream(), tup.e[i], 0));     // the tables have arrived before the launch starts
  const auto t1d = std::chrono::steady_clock::now();
  // dynamic LDS: the packed reduced system of the largest window, rhs / diagonal, control words and the streaming area
  const size_t lds_bytes = sizeof(double) * lds_doubles;
  if (lds_bytes > 160 * 1024) { set_error("dvm_ba_optimize_windows: a window needs more than 160 KB of LDS"); return DVM_ERR_CAPACITY; }
  if (!fast) DVM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_window), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // on the staging stream of the calling thread (upload -> kernel -> download is one in-order chain there; the legacy NULL stream would
  // also order this launch against every other thread's staging stream: dvm_ba_optimize_batch's workers serialised on it)
  struct KernelEvents { hipEvent_t a = nullptr, b = nullptr; ~KernelEvents() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); } };
  thread_local KernelEvents kev;     // the launch's duration -> dvm_ba_stats::kernel_us (the fast form's roofline figure in bench_legs.lba_fast)
  if (fast) {
    const int groups = 8 * ((K + 7) / 8);
    DVM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_window_cluster), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (!kev.a) { DVM_HIP(hipEventCreate(&kev.a)); DVM_HIP(hipEventCreate(&kev.b)); }
    hipEventRecord(kev.a, st.stream());
    const bool scatter = std::getenv("DVM_BA_CLUSTER_SCATTER") != nullptr;     // (placement test: see the kernel)
    hipLaunchKernelGGL(k_ba_window_cluster, dim3(groups * G), dim3(kWinThreads), lds_bytes, st.stream(), st.ptr<BaWin>(views_slot), sw.d, scatter ? -K : K, G);
    hipEventRecord(kev.b, st.stream());
  }
  else hipLaunchKernelGGL(k_ba_window, dim3(K), dim3(kWinThreads), lds_bytes, st.stream(), st.ptr<BaWin>(views_slot), sw.d);
  DVM_HIP(hipGetLastError());
  if (stop_flag) {                     // g2o's forceStopFlag: written by another thread while the optimisation runs (LocalMapping.cc:305,359)
    hipEvent_t ev;
    DVM_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    DVM_HIP(hipEventRecord(ev, st.stream()));
    while (hipEventQuery(ev) == hipErrorNotReady) *sw.h = *stop_flag ? 1 : 0;
    hipEventDestroy(ev);
  }
  const auto t1e = std::chrono::steady_clock::now();
  if ((rc = st.download()) != DVM_OK) return rc;
  if (std::getenv("DVM_BA_WINDOW_TIMING")) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "windows K=%d G=%d: build %.2f  register %.2f  layout %.2f  views %.2f  upload(memcpy + enqueue) %.2f  launch %.2f  wait + download %.2f ms\n", K, G,
                 ms(t0, t1), ms(t1, t1a), ms(t1a, t1b), ms(t1b, t1c), ms(t1c, t1d), ms(t1d, t1e), ms(t1e, std::chrono::steady_clock::now()));
  }
  if (fast && G > 1) {
    bool timed_out = std::getenv("DVM_BA_TEST_TIMEOUT") != nullptr;      // (test hook: take the repeat path without a real time-out)
    for (int k = 0; k < K; k++) timed_out = timed_out || sync_back[k][8] != 0u;
    if (timed_out)     // a cluster's workgroups were not all resident (other work held compute units): solve the batch again, one workgroup per window
      return dvm_ba_optimize_windows_impl(device, windows, K, stop_flag, stats, normalize_input, true, 1);
  }
  const auto t2 = std::chrono::steady_clock::now();
  if (want_prof) {
    static const char* names_seq[16] = {"edge pass + Jacobians", "Hll / bl", "Hpp / bp (streamed)", "schur: rhs chain", "Schur (streamed)", "Cholesky", "forward / backward", "landmark back-sub",
                                    "oplus + scale terms", "edge pass chi2", "sequential sums", "schur: runs + wait", "schur: store + prefetch", "schur: Dinv", "schur: W Dinv", "decision + rest"};
    const char* const* names = names_seq;
    static const char* cnames[16] = {"edge pass + Jacobians", "Hll / bl, Hpp / bp", "W Dinv rows", "Schur blocks + rhs", "assemble S in LDS", "Cholesky", "substitution + x", "W^T x rows",
                                     "landmarks + oplus", "edge pass chi2 + sums", "IN BARRIERS (workgroup 0)", "decision + rest", "  chol: diagonal block", "  chol: panel", "  chol: barriers", "  chol: trailing update"};
    if (fast) names = cnames;
    for (int i = 0; i < 16; i++) if (prof_out[0][i]) std::fprintf(stderr, "window 0: %-24s %10.1f us\n", names[i], prof_out[0][i] / 2400.0);   // s_memtime ticks at the shader clock (MI355X_MICROARCH.md): us at 2.4 GHz
  }
  for (int k = 0; k < K; k++) {
    if (stats) {
      stats[k] = outs[k].st;
      if (fast && kev.a) { float ms = 0; if (hipEventElapsedTime(&ms, kev.a, kev.b) == hipSuccess) stats[k].kernel_us = (int32_t)(ms * 1e3f + 0.5f); }
      stats[k].ms_structure = std::chrono::duration<double, std::milli>(t1 - t0).count() / K;
      stats[k].ms_optimize = std::chrono::duration<double, std::milli>(t2 - t1).count();     // the whole batch: upload, the one launch, download
    }
  }
  return DVM_OK;
}

extern "C" {

int dvm_ba_optimize_windows(int device, const dvm_ba_window* windows, int K, const volatile uint8_t* stop_flag, dvm_ba_stats* stats) {
  return dvm_ba_optimize_windows_impl(device, windows, K, stop_flag, stats, true, false, 0);
}
int dvm_ba_optimize_windows_fast(int device, const dvm_ba_window* windows, int K, const volatile uint8_t* stop_flag, dvm_ba_stats* stats) {
  // DVM_BA_SPLIT: a batch of 64 or more windows in two halves, the second on a persistent helper thread with its own staging buffers and
  // stream; the two launches (G workgroups per window each) share the chip.  Every window's result is independent of its batch and of G, so
  // the split changes nothing but the time.  It was the default while the host side of such a call (tables, 60+ MB through a staging copy)
  // was as long as its kernel (128 windows: 23 -> 18 ms); with the tables sent from the builders' threads the one launch is the faster and
  // the steadier form (128 windows: 17.0-17.2 ms against 15.5-19.5).
  const bool split = std::getenv("DVM_BA_SPLIT") != nullptr;       // (read at every call: the tests take both paths)
  if (K >= 64 && split) {
    if (K < 0 || !windows) { set_error("dvm_ba_optimize_windows: null windows"); return DVM_ERR_INVALID; }
    HelperThread& H = HelperThread::get();
    std::unique_lock<std::mutex> user(H.use, std::try_to_lock);
    if (user.owns_lock()) {
      const int K0 = (K + 1) / 2, K1 = K - K0;
      // the cluster size of BOTH launches from the whole batch: their workgroups must all be resident together (two half grids that each
      // filled the chip would starve each other's barriers)
      int cus = 0, G = 1;
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
      const char* env_c = std::getenv("DVM_BA_CLUSTER");
      const int want = env_c ? atoi(env_c) : 0;
      const int groups = 8 * ((K0 + 7) / 8) + 8 * ((K1 + 7) / 8);
      for (int g = 8; g >= 1; g >>= 1) if ((want > 0 && g == want) || (want <= 0 && groups * g <= cus)) { G = g; break; }
      int rc1 = DVM_OK;
      std::string err1;
      H.submit([&, K0, K1, G] {
        rc1 = dvm_ba_optimize_windows_impl(device, windows + K0, K1, stop_flag, stats ? stats + K0 : nullptr, true, true, G);
        if (rc1 != DVM_OK) err1 = last_error_cstr();
      });
      const int rc0 = dvm_ba_optimize_windows_impl(device, windows, K0, stop_flag, stats, true, true, G);
      H.wait();
      if (rc0 != DVM_OK) return rc0;
      if (rc1 != DVM_OK) { set_error(err1); return rc1; }
      return DVM_OK;
    }
  }
  return dvm_ba_optimize_windows_impl(device, windows, K, stop_flag, stats, true, true, 0);
}

// ---- dvm_ba_pool_*: the LocalBundleAdjustment calls of several agents' LocalMapping threads (one per agent in the reference:
// LocalMapping.cc:172 -> Optimizer.cc:1030-1387), each a BLOCKING call with its own window, batched behind the boundary into one launch
// of the cluster form (group commit, as dvm_orb_pool / dvm_match_pool / dvm_pose_pool do it for the tracking threads' per-frame calls).
// A window's result does not depend on the batch it rode in (k_ba_window_cluster): every caller gets the bits of a solo
// dvm_ba_optimize_windows_fast call.
struct dvm_ba_pool {
  int device = 0;
  GroupCommit gc;
  std::vector<dvm_ba_window> win[GroupCommit::kLanes];
  std::vector<dvm_ba_stats> st[GroupCommit::kLanes];
};
int dvm_ba_pool_create(int device, int max_batch, int window_us, dvm_ba_pool** out) {
  if (!out || max_batch < 1 || max_batch > 256) return DVM_ERR_INVALID;
  *out = nullptr;
  int rc = dvm_set_device(device);
  if (rc != DVM_OK) return rc;
  dvm_ba_pool* p = new (std::nothrow) dvm_ba_pool();
  if (!p) return DVM_ERR_INVALID;
  p->device = device;
  p->gc.max_batch = max_batch;
  p->gc.window_us = window_us >= 0 ? window_us : 300;     // a LocalBundleAdjustment call takes milliseconds: 0.3 ms of collecting costs little
  for (int l = 0; l < GroupCommit::kLanes; l++) { p->win[l].resize((size_t)max_batch); p->st[l].resize((size_t)max_batch); }
  *out = p;
  return DVM_OK;
}
void dvm_ba_pool_destroy(dvm_ba_pool* pool) { delete pool; }
int dvm_ba_pool_optimize(dvm_ba_pool* pool, const dvm_ba_window* w, dvm_ba_stats* stats, int* batch_size) {
  if (!pool || !w) return DVM_ERR_INVALID;
  if (w->n_poses < 1 || w->n_points < 0 || w->n_edges < 0 || w->iterations < 0 || !w->poses || !w->fixed || (w->n_points && !w->points) || (w->n_edges && !w->edges)) {
    set_error("dvm_ba_pool_optimize: the window is incomplete");
    return DVM_ERR_INVALID;
  }
  const int64_t key[4] = {0, 0, 0, 0};
  int li = 0, slot = 0;
  int rc = pool->gc.join(key, [](int) { return 0; }, li, slot);
  if (rc != 0) return rc;
  pool->win[li][(size_t)slot] = *w;                       // (the arrays stay the caller's: it is blocked here until the batch is through)
  if (pool->gc.arrive(li, slot)) {
    const int count = pool->gc.batch_count(li);
    // the batch runs on the library's helper thread, whichever caller leads it: ONE set of page-locked / device staging buffers and
    // window tables for the service instead of one per agent thread (they are thread-local: ~5 MB per window of the largest batch)
    int brc = DVM_OK;
    std::string berr;
    {
      HelperThread& H = HelperThread::get();
      std::lock_guard<std::mutex> user(H.use);
      H.submit([&] {
        brc = dvm_set_device(pool->device);
        if (brc == DVM_OK) brc = dvm_ba_optimize_windows_fast(pool->device, pool->win[li].data(), count, nullptr, pool->st[li].data());
        if (brc != DVM_OK) berr = last_error_cstr();
      });
      H.wait();
    }
    pool->gc.publish(li, brc, berr);
  }
  std::string err;
  int count = 0;
  rc = pool->gc.result(li, &err, &count);
  if (rc == DVM_OK && stats) *stats = pool->st[li][(size_t)slot];
  if (batch_size) *batch_size = count;
  pool->gc.finish(li);
  if (rc != DVM_OK) set_error("dvm_ba_pool_optimize: " + err);
  return rc;
}

int dvm_f64_spec_eval(int device, const double* x, int n, double* out) {
  if (n < 0 || (n && (!x || !out))) return DVM_ERR_INVALID;
  if (n == 0) return DVM_OK;
  int rc = dvm_set_device(device);
  if (rc != DVM_OK) return rc;
  Stage st;
  const int ix = st.in(x, 8 * (size_t)n), io = st.out(out, 24 * (size_t)n);
  if ((rc = st.upload()) != DVM_OK) return rc;
  hipLaunchKernelGGL(k_f64_spec, dim3((n + 255) / 256), dim3(256), 0, st.stream(), st.ptr<double>(ix), n, st.ptr<double>(io));
  DVM_HIP(hipGetLastError());
  return st.download();
}

}  // extern "C"
