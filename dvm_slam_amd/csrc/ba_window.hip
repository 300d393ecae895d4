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
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>

#include "../../include/dvmslam_hip.h"
#include "f64_spec.h"
#include "ba_kernels.h"
#include "host_stage.h"
#include "orb_pipeline.h"   // set_error / hip_check / DVM_HIP

namespace dvm {

constexpr int kWinThreads = 512;
constexpr int kWinMaxFree = 30;          // 6 * 30 = 180 rows: packed lower triangle 130 320 B of LDS
constexpr int kWinLdsMisc = 2 * 180 + 64 + 8 * 128;   // doubles besides S: rhs / diag, control words, the waves' sequential-sum buffers

// device view of one window: pointers into the call's staging block
struct BaWin {
  int32_t P, L, E, nfree, nact, nblk, iterations, pad0;
  double fx, fy, cx, cy, delta;
  double *poses, *poses_t;            // [P][7] accepted / trial state
  double *pts, *pts_t;                // [L][3]
  double *out_poses, *out_pts;        // the accepted state at the end (what the host fetches)
  const int32_t *pidx, *lidx;         // [P] free index or -1; [L] active index or -1
  const int32_t *free_pose, *act_pt;  // [nfree], [nact]
  const int32_t *e_pose, *e_point;    // [E]
  const double *e_obs, *e_info;       // [E][2], [E]
  double *e_A, *e_B, *e_w, *e_W, *e_WD, *e_Wdb;   // [E][6], [E][12], [E][4] (w, wr0, wr1, -), [E][18], [E][18], [E][6]
  double *e_chi2, *e_rho;             // [E] chi2 / rho(chi2) of the last evaluation
  uint8_t* e_depth;                   // [E] isDepthPositive() at the final state
  const int32_t *cam_start, *cam_edges;     // free camera -> its edges, ascending edge index            [nfree + 1], [..]
  const int32_t *camp_edges;                // the same lists ordered by (landmark, edge): the Schur loop's order
  const int32_t *pt_start, *pt_edges;       // active landmark -> its edges, ascending                   [nact + 1], [..]
  const int32_t *blk_i1, *blk_i2, *blk_start, *pair_k1, *pair_k2;   // non-zero lower blocks of the reduced system and their pair lists
  double *Hpp, *bp, *Hll, *bl, *Dinv, *db, *x, *terms;   // [nfree][36], [6 nfree], [nact][9], [3 nact], [nact][9], [3 nact], [6 nfree + 3 nact] x 2
  dvm_ba_stats* stats;
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
#pragma unroll
    for (int j = 0; j < 64; j++) s += b[j];
  }
  return s;
}

// ------------------------------------------------------------------------------------------------ edge pass
// JAC = false: computeActiveErrors -- chi2 and rho of every edge at the state (poses, pts).  JAC = true: additionally linearizeOplus +
// the edge's part of constructQuadraticForm: A (2x3), B (2x6), w = rho' Omega, wr = -Omega e rho', W = w B^T A.
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
    double* oA = W.e_A + 6 * (size_t)k;
    double* oB = W.e_B + 12 * (size_t)k;
    double* oW = W.e_W + 18 * (size_t)k;
#pragma unroll
    for (int i = 0; i < 6; i++) oA[i] = A[i];
#pragma unroll
    for (int i = 0; i < 12; i++) oB[i] = B[i];
    W.e_w[4 * (size_t)k] = w; W.e_w[4 * (size_t)k + 1] = wr0; W.e_w[4 * (size_t)k + 2] = wr1;
    if (W.pidx[p] >= 0) {
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) oW[3 * a + b] = w * (B[a] * A[b] + B[6 + a] * A[3 + b]);
    } else {
#pragma unroll
      for (int i = 0; i < 18; i++) oW[i] = 0.0;
    }
  }
}

// Hpp / bp of the free cameras and Hll / bl of the active landmarks: every entry is one lane's chain over the vertex's edges in
// edge order.  Camera entries: 21 lower (a, b) pairs + 6 of bp = 27 lanes per camera; landmark: one thread does its 6 + 3 chains.
__device__ void win_accumulate(const BaWin& W) {
  const int ncam_lanes = 27 * W.nfree;
  for (int t = threadIdx.x; t < ncam_lanes; t += kWinThreads) {
    const int i = t / 27, e = t - 27 * i;
    int a, b;
    if (e < 21) { a = 0; int r = e; while (r > a) { r -= a + 1; a++; } b = r; }   // e = a (a + 1) / 2 + b, b <= a
    else { a = e - 21; b = 0; }
    const int s0 = W.cam_start[i], s1 = W.cam_start[i + 1];
    double acc = 0.0;
    if (e < 21) {
      for (int q = s0; q < s1; q++) {
        const int k = W.cam_edges[q];
        const double* B = W.e_B + 12 * (size_t)k;
        const double w = W.e_w[4 * (size_t)k];
        acc += w * (B[a] * B[b] + B[6 + a] * B[6 + b]);
      }
      W.Hpp[36 * (size_t)i + 6 * a + b] = acc;
      W.Hpp[36 * (size_t)i + 6 * b + a] = acc;       // (a product commutes: the mirrored entry has the same bits)
    } else {
      for (int q = s0; q < s1; q++) {
        const int k = W.cam_edges[q];
        const double* B = W.e_B + 12 * (size_t)k;
        const double wr0 = W.e_w[4 * (size_t)k + 1], wr1 = W.e_w[4 * (size_t)k + 2];
        acc += B[a] * wr0 + B[6 + a] * wr1;
      }
      W.bp[6 * (size_t)i + a] = acc;
    }
  }
  for (int li = threadIdx.x; li < W.nact; li += kWinThreads) {
    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int q = W.pt_start[li]; q < W.pt_start[li + 1]; q++) {
      const int k = W.pt_edges[q];
      const double* A = W.e_A + 6 * (size_t)k;
      const double w = W.e_w[4 * (size_t)k], wr0 = W.e_w[4 * (size_t)k + 1], wr1 = W.e_w[4 * (size_t)k + 2];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        g[a] += A[a] * wr0 + A[3 + a] * wr1;
#pragma unroll
        for (int b = 0; b < 3; b++) h[3 * a + b] += w * (A[a] * A[b] + A[3 + a] * A[3 + b]);
      }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) W.Hll[9 * (size_t)li + i] = h[i];
#pragma unroll
    for (int i = 0; i < 3; i++) W.bl[3 * (size_t)li + i] = g[i];
  }
}

// packed lower triangle: row i starts at i (i + 1) / 2
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// ------------------------------------------------------------------------------------------------ the kernel
__global__ void __launch_bounds__(kWinThreads) k_ba_window(const BaWin* __restrict__ wins, const volatile int* __restrict__ stop) {
  extern __shared__ double lds[];
  const BaWin& W = wins[blockIdx.x];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int n = 6 * W.nfree, nl = 3 * W.nact;
  double* S = lds;                                  // packed lower triangle of the reduced camera system, n (n + 1) / 2
  double* rhs = S + (size_t)n * (n + 1) / 2;        // bschur -> x_p                                          [n]
  double* diag = rhs + n;                           // L_kk                                                   [n]
  double* ctl = diag + n;                           // control words shared by the workgroup                  [64]
  double* seqbuf = ctl + 64 + 128 * wave;           // this wave's buffer of wave_sequential_sum              [128]
  dvm_ba_stats* const st = W.stats;                 // written by thread 0 only
  if (tid == 0) {
    st->iterations = st->total_trials = st->stop_reason = st->pad = 0;
    st->chi2_initial = st->chi2_final = st->lambda_final = 0;
    for (int i = 0; i < 64; i++) { st->trials_per_iter[i] = 0; st->chi2_per_iter[i] = 0; st->lambda_per_iter[i] = 0; }
    st->ms_structure = st->ms_optimize = 0; st->spec_trials = st->spec_kept = 0;
  }
  // g2o's buildStructure reallocates _x: "the last successful solve" starts as zeros
  for (int i = tid; i < n + nl; i += kWinThreads) W.x[i] = 0.0;
  // the trial state starts as a copy (fixed cameras and unobserved landmarks never change)
  for (int i = tid; i < 7 * W.P; i += kWinThreads) W.poses_t[i] = W.poses[i];
  for (int i = tid; i < 3 * W.L; i += kWinThreads) W.pts_t[i] = W.pts[i];
  double* poses = W.poses; double* poses_t = W.poses_t; double* pts = W.pts; double* pts_t = W.pts_t;
  __syncthreads();

  double lambda = -1, ni = 2, currentChi = 0, chi_last = 0;
  int nBad = 0, it_done = 0, trials_total = 0, stop_reason = 0;
  for (int it = 0; it < W.iterations; it++) {
    if (tid == 0) ctl[0] = (stop && *stop) ? 1.0 : 0.0;
    __syncthreads();
    if (ctl[0] != 0.0) break;
    // computeActiveErrors + robust chi2 + buildSystem at the accepted state.  (From the second iteration on g2o recomputes the chi2
    // of the state the last accepted trial has just evaluated: same state, same sums, same bits -- only the Jacobians are new.)
    win_edge_pass<true>(W, poses, pts);
    __syncthreads();
    if (it == 0) {
      if (wave == 0) { const double c = wave_sequential_sum(W.e_rho, W.E, seqbuf); if (tid == 0) ctl[1] = c; }
    }
    win_accumulate(W);
    __syncthreads();
    if (it == 0) {
      currentChi = ctl[1];
      if (tid == 0) st->chi2_initial = currentChi;
      // computeLambdaInit: tau * max |diagonal| over all active vertices (a maximum has no order)
      double mx = 0;
      for (int i = tid; i < n; i += kWinThreads) mx = fmax(mx, fabs(W.Hpp[36 * (size_t)(i / 6) + 7 * (i % 6)]));
      for (int i = tid; i < nl; i += kWinThreads) mx = fmax(mx, fabs(W.Hll[9 * (size_t)(i / 3) + 4 * (i % 3)]));
      for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
      __syncthreads();
      if ((tid & 63) == 0) ctl[8 + wave] = mx;
      __syncthreads();
      mx = ctl[8];
      for (int w2 = 1; w2 < kWinThreads / 64; w2++) mx = fmax(mx, ctl[8 + w2]);
      lambda = 1e-5 * mx;
      ni = 2; nBad = 0;
    }
    const double iniChi = currentChi;
    double tempChi = currentChi, rho = 0;
    int qmax = 0;
    bool stopped = false;
    do {
      // ---- solve(lambda): damped landmark inverses, W Dinv, W Dinv bl per edge
      for (int li = tid; li < W.nact; li += kWinThreads) {
        double D[9], Di[9], d3[3];
#pragma unroll
        for (int i = 0; i < 9; i++) D[i] = W.Hll[9 * (size_t)li + i];
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        w_inv3(D, Di);
        w_mat3_vec(Di, W.bl + 3 * (size_t)li, d3);
#pragma unroll
        for (int i = 0; i < 9; i++) W.Dinv[9 * (size_t)li + i] = Di[i];
#pragma unroll
        for (int i = 0; i < 3; i++) W.db[3 * (size_t)li + i] = d3[i];
      }
      for (int i = tid; i < n * (n + 1) / 2; i += kWinThreads) S[i] = 0.0;
      __syncthreads();
      for (int k = tid; k < W.E; k += kWinThreads) {
        if (W.pidx[W.e_pose[k]] < 0) continue;
        const int li = W.lidx[W.e_point[k]];
        const double* W1 = W.e_W + 18 * (size_t)k;
        const double* Di = W.Dinv + 9 * (size_t)li;
        const double* d3 = W.db + 3 * (size_t)li;
        double* WD = W.e_WD + 18 * (size_t)k;
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
          for (int b = 0; b < 3; b++) WD[3 * a + b] = W1[3 * a] * Di[b] + W1[3 * a + 1] * Di[3 + b] + W1[3 * a + 2] * Di[6 + b];
          W.e_Wdb[6 * (size_t)k + a] = W1[3 * a] * d3[0] + W1[3 * a + 1] * d3[1] + W1[3 * a + 2] * d3[2];
        }
      }
      __syncthreads();
      // bschur(i) = bp(i) - sum over the camera's edges in (landmark, edge) order of W Dinv bl
      for (int t = tid; t < n; t += kWinThreads) {
        const int i = t / 6, a = t - 6 * i;
        double acc = W.bp[t];
        for (int q = W.cam_start[i]; q < W.cam_start[i + 1]; q++) acc -= W.e_Wdb[6 * (size_t)W.camp_edges[q] + a];
        rhs[t] = acc;
      }
      // Hschur(i1, i2) = [Hpp + lambda I] - sum over the block's pairs in landmark order of (W1 Dinv) W2^T
      for (int t = tid; t < 36 * W.nblk; t += kWinThreads) {
        const int blk = t / 36, ab = t - 36 * blk, a = ab / 6, b = ab - 6 * a;
        const int i1 = W.blk_i1[blk], i2 = W.blk_i2[blk];
        if (i1 == i2 && b > a) continue;
        double acc = 0.0;
        if (i1 == i2) acc = W.Hpp[36 * (size_t)i1 + 6 * a + b] + (a == b ? lambda : 0.0);
        for (int q = W.blk_start[blk]; q < W.blk_start[blk + 1]; q++) {
          const double* WD = W.e_WD + 18 * (size_t)W.pair_k1[q] + 3 * a;
          const double* W2 = W.e_W + 18 * (size_t)W.pair_k2[q] + 3 * b;
          acc -= WD[0] * W2[0] + WD[1] * W2[1] + WD[2] * W2[2];
        }
        S[tri(6 * i1 + a, 6 * i2 + b)] = acc;
      }
      __syncthreads();
      // ---- Cholesky: entry (i, j) receives its subtractions L(i, k) L(j, k) in ascending k, as the row-wise dot products of the
      // envelope factorisation apply them; L(i, j) = s / L(j, j) by IEEE division, L(j, j) = sqrt(s)
      bool ok = true;
      for (int k = 0; k < n; k++) {
        const double d = S[tri(k, k)];
        if (!(d > 0)) { ok = false; break; }             // uniform: every thread reads the same word
        const double lkk = sqrt(d);
        for (int i = k + 1 + tid; i < n; i += kWinThreads) S[tri(i, k)] = S[tri(i, k)] / lkk;
        if (tid == 0) diag[k] = lkk;
        __syncthreads();
        const int tx = tid & 15, ty = tid >> 4;
        for (int i = k + 1 + ty; i < n; i += kWinThreads / 16) {
          const double lik = S[tri(i, k)];
          for (int j = k + 1 + tx; j <= i; j += 16) S[tri(i, j)] -= lik * S[tri(j, k)];
        }
        __syncthreads();
      }
      if (ok) {
        // forward substitution: y(i) = (b(i) - sum_{j < i} L(i, j) y(j)) / L(i, i), the subtractions in ascending j; then backward:
        // x(i) /= L(i, i); x(j) -= L(i, j) x(i) for j < i, i descending.  One wave, its lanes own rows lane, lane + 64, lane + 128.
        if (wave == 0) {
          const int lane = tid;
          for (int i = 0; i < n; i++) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const double yi = rhs[i] / diag[i];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) rhs[i] = yi;
            for (int r = i + 1 + lane; r < n; r += 64) rhs[r] -= S[tri(r, i)] * yi;
          }
          for (int i = n - 1; i >= 0; i--) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const double xi = rhs[i] / diag[i];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) rhs[i] = xi;
            for (int j = lane; j < i; j += 64) rhs[j] -= S[tri(i, j)] * xi;
          }
        }
        __syncthreads();
        for (int i = tid; i < n; i += kWinThreads) W.x[i] = rhs[i];
        // xl = Dinv (bl - W^T xp): per landmark, its edges in order, per edge the six camera components in order
        for (int li = tid; li < W.nact; li += kWinThreads) {
          double c0 = W.bl[3 * (size_t)li], c1 = W.bl[3 * (size_t)li + 1], c2 = W.bl[3 * (size_t)li + 2];
          for (int q = W.pt_start[li]; q < W.pt_start[li + 1]; q++) {
            const int k = W.pt_edges[q];
            const int i = W.pidx[W.e_pose[k]];
            if (i < 0) continue;
            const double* Wk = W.e_W + 18 * (size_t)k;
#pragma unroll
            for (int a = 0; a < 6; a++) {
              const double xa = rhs[6 * i + a];
              c0 -= Wk[3 * a] * xa; c1 -= Wk[3 * a + 1] * xa; c2 -= Wk[3 * a + 2] * xa;
            }
          }
          const double c[3] = {c0, c1, c2};
          double xl[3];
          w_mat3_vec(W.Dinv + 9 * (size_t)li, c, xl);
          W.x[n + 3 * (size_t)li] = xl[0]; W.x[n + 3 * (size_t)li + 1] = xl[1]; W.x[n + 3 * (size_t)li + 2] = xl[2];
        }
      }
      __syncthreads();
      // ---- the update is applied and the errors evaluated whether or not the solve succeeded (g2o: x then still holds the last
      // successful solve, optimization_algorithm_levenberg.cpp:107-127); computeScale's terms x_j (lambda x_j + b_j)
      for (int i = tid; i < W.nfree; i += kWinThreads) {
        const int p = W.free_pose[i];
        w_se3_oplus(poses + 7 * (size_t)p, W.x + 6 * (size_t)i, poses_t + 7 * (size_t)p);
      }
      for (int t = tid; t < nl; t += kWinThreads) {
        const int l = W.act_pt[t / 3];
        pts_t[3 * (size_t)l + t % 3] = pts[3 * (size_t)l + t % 3] + W.x[n + t];
      }
      for (int j = tid; j < n; j += kWinThreads) { const double xj = W.x[j]; W.terms[j] = xj * (lambda * xj + W.bp[j]); }
      for (int j = tid; j < nl; j += kWinThreads) { const double xj = W.x[n + j]; W.terms[n + j] = xj * (lambda * xj + W.bl[j]); }
      __syncthreads();
      win_edge_pass<false>(W, poses_t, pts_t);
      __syncthreads();
      if (wave == 0) { const double c = wave_sequential_sum(W.e_rho, W.E, seqbuf); if (tid == 0) ctl[1] = c; }
      if (wave == 1) { const double c = wave_sequential_sum(W.terms, n + nl, seqbuf); if (tid == 64) ctl[2] = c; }
      __syncthreads();
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
  // results: the accepted state (the buffers may have been swapped any number of times), depth signs at that state
  for (int i = tid; i < 7 * W.P; i += kWinThreads) W.out_poses[i] = poses[i];
  for (int i = tid; i < 3 * W.L; i += kWinThreads) W.out_pts[i] = pts[i];
  for (int k = tid; k < W.E; k += kWinThreads) {
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
struct WinBuild {
  int P = 0, L = 0, E = 0, nfree = 0, nact = 0, nblk = 0;
  std::vector<double> poses;                 // normalised quaternions (SE3Quat's constructor)
  std::vector<int32_t> pidx, lidx, free_pose, act_pt, e_pose, e_point, cam_start, cam_edges, camp_edges, pt_start, pt_edges;
  std::vector<int32_t> blk_i1, blk_i2, blk_start, pair_k1, pair_k2;
  std::vector<double> e_obs, e_info;
};

int build_window(const dvm_ba_window& w, WinBuild& b, bool normalize) {
  const int P = w.n_poses, L = w.n_points, E = w.n_edges;
  if (P < 0 || L < 0 || E < 0 || (P && (!w.poses || !w.fixed)) || (L && !w.points) || (E && !w.edges)) { set_error("dvm_ba_optimize_windows: null array"); return DVM_ERR_INVALID; }
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
  // incidence lists in edge order
  b.cam_start.assign(b.nfree + 1, 0); b.pt_start.assign(b.nact + 1, 0);
  for (int k = 0; k < E; k++) {
    if (b.pidx[b.e_pose[k]] >= 0) b.cam_start[b.pidx[b.e_pose[k]] + 1]++;
    b.pt_start[b.lidx[b.e_point[k]] + 1]++;
  }
  for (int i = 0; i < b.nfree; i++) b.cam_start[i + 1] += b.cam_start[i];
  for (int i = 0; i < b.nact; i++) b.pt_start[i + 1] += b.pt_start[i];
  b.cam_edges.resize(b.cam_start[b.nfree]); b.pt_edges.resize(b.pt_start[b.nact]);
  {
    std::vector<int32_t> cc(b.cam_start.begin(), b.cam_start.end() - 1), pc(b.pt_start.begin(), b.pt_start.end() - 1);
    for (int k = 0; k < E; k++) {
      const int i = b.pidx[b.e_pose[k]];
      if (i >= 0) b.cam_edges[cc[i]++] = k;
      b.pt_edges[pc[b.lidx[b.e_point[k]]]++] = k;
    }
  }
  // the Schur loop's order (block_solver.hpp:381-439): landmark by landmark, a landmark's edges in order, per edge the edges again.
  // camp_edges: a camera's edges as that loop meets them; pairs: per non-zero lower block (i1 >= i2) its (k1, k2) in that order.
  b.camp_edges.resize(b.cam_edges.size());
  std::vector<int32_t> blk_of((size_t)b.nfree * b.nfree, -1);
  std::vector<std::vector<int32_t>> pairs;
  {
    std::vector<int32_t> cc(b.cam_start.begin(), b.cam_start.end() - 1);
    for (int li = 0; li < b.nact; li++) {
      for (int q1 = b.pt_start[li]; q1 < b.pt_start[li + 1]; q1++) {
        const int k1 = b.pt_edges[q1], i1 = b.pidx[b.e_pose[k1]];
        if (i1 < 0) continue;
        b.camp_edges[cc[i1]++] = k1;
        for (int q2 = b.pt_start[li]; q2 < b.pt_start[li + 1]; q2++) {
          const int k2 = b.pt_edges[q2], i2 = b.pidx[b.e_pose[k2]];
          if (i2 < 0 || i2 > i1) continue;
          if (i2 == i1 && k2 != k1) continue;
          int32_t& id = blk_of[(size_t)i1 * b.nfree + i2];
          if (id < 0) { id = (int32_t)pairs.size(); pairs.emplace_back(); b.blk_i1.push_back(i1); b.blk_i2.push_back(i2); }
          pairs[id].push_back(k1); pairs[id].push_back(k2);
        }
      }
    }
  }
  // every free camera has a diagonal block even without a landmark of its own among the free ones (it always has: it is "used")
  b.nblk = (int)pairs.size();
  b.blk_start.assign(b.nblk + 1, 0);
  for (int i = 0; i < b.nblk; i++) b.blk_start[i + 1] = b.blk_start[i] + (int32_t)pairs[i].size() / 2;
  b.pair_k1.resize(b.blk_start[b.nblk]); b.pair_k2.resize(b.blk_start[b.nblk]);
  for (int i = 0; i < b.nblk; i++)
    for (size_t j = 0; j < pairs[i].size() / 2; j++) { b.pair_k1[b.blk_start[i] + j] = pairs[i][2 * j]; b.pair_k2[b.blk_start[i] + j] = pairs[i][2 * j + 1]; }
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

int dvm_ba_optimize_windows_impl(int device, const dvm_ba_window* windows, int K, const volatile uint8_t* stop_flag, dvm_ba_stats* stats, bool normalize_input) {
  if (K < 0 || (K && !windows)) { set_error("dvm_ba_optimize_windows: null windows"); return DVM_ERR_INVALID; }
  if (K == 0) return DVM_OK;
  int rc = dvm_set_device(device);
  if (rc != DVM_OK) return rc;
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<WinBuild> B(K);
  for (int k = 0; k < K; k++) if ((rc = build_window(windows[k], B[k], normalize_input)) != DVM_OK) return rc;
  const auto t1 = std::chrono::steady_clock::now();
  thread_local StopWord sw;
  if ((rc = sw.ensure()) != DVM_OK) return rc;
  *sw.h = (stop_flag && *stop_flag) ? 1 : 0;

  Stage st;
  struct Slots { int poses, pts, pidx, lidx, free_pose, act_pt, e_pose, e_point, e_obs, e_info, cam_start, cam_edges, camp_edges, pt_start, pt_edges,
                     blk_i1, blk_i2, blk_start, pair_k1, pair_k2, out_poses, out_pts, echi, edepth, stats, poses_t, pts_t, eA, eB, ew, eW, eWD, eWdb, erho, Hpp, bp, Hll,
                     bl, Dinv, db, x, terms; };
  std::vector<Slots> sl(K);
  auto I32 = [&](const std::vector<int32_t>& v) { return st.in(v.empty() ? nullptr : v.data(), v.size() * 4); };
  auto F64 = [&](const std::vector<double>& v) { return st.in(v.empty() ? nullptr : v.data(), v.size() * 8); };
  for (int k = 0; k < K; k++) {                     // inputs: the state and g2o's graph structure as index arrays
    const WinBuild& b = B[k]; Slots& s = sl[k];
    s.poses = F64(b.poses);
    s.pts = st.in(b.L ? windows[k].points : nullptr, 24 * (size_t)b.L);
    s.pidx = I32(b.pidx); s.lidx = I32(b.lidx); s.free_pose = I32(b.free_pose); s.act_pt = I32(b.act_pt); s.e_pose = I32(b.e_pose); s.e_point = I32(b.e_point);
    s.e_obs = F64(b.e_obs); s.e_info = F64(b.e_info);
    s.cam_start = I32(b.cam_start); s.cam_edges = I32(b.cam_edges); s.camp_edges = I32(b.camp_edges); s.pt_start = I32(b.pt_start); s.pt_edges = I32(b.pt_edges);
    s.blk_i1 = I32(b.blk_i1); s.blk_i2 = I32(b.blk_i2); s.blk_start = I32(b.blk_start); s.pair_k1 = I32(b.pair_k1); s.pair_k2 = I32(b.pair_k2);
  }
  std::vector<BaWin> views(K);
  const int views_slot = st.in(views.data(), sizeof(BaWin) * (size_t)K);   // filled in below, once layout() has placed everything
  struct Outs { std::vector<double> poses, pts, chi2; std::vector<uint8_t> depth; dvm_ba_stats st; };
  std::vector<Outs> outs(K);
  for (int k = 0; k < K; k++) {                     // outputs (one contiguous span to fetch)
    const WinBuild& b = B[k]; Slots& s = sl[k]; Outs& o = outs[k];
    o.poses.resize(7 * (size_t)b.P); o.pts.resize(3 * (size_t)b.L); o.chi2.resize(b.E); o.depth.resize(b.E);
    s.out_poses = st.out(o.poses.data(), 56 * (size_t)b.P);
    s.out_pts = st.out(o.pts.data(), 24 * (size_t)b.L);
    s.echi = st.out(o.chi2.data(), 8 * (size_t)b.E);
    s.edepth = st.out(o.depth.data(), (size_t)b.E);
    s.stats = st.out(&o.st, sizeof(dvm_ba_stats));
  }
  for (int k = 0; k < K; k++) {                     // working memory
    const WinBuild& b = B[k]; Slots& s = sl[k];
    const size_t E = b.E, n = 6 * (size_t)b.nfree, nl = 3 * (size_t)b.nact;
    s.poses_t = st.scratch(56 * (size_t)b.P); s.pts_t = st.scratch(24 * (size_t)b.L);
    s.eA = st.scratch(48 * E); s.eB = st.scratch(96 * E); s.ew = st.scratch(32 * E); s.eW = st.scratch(144 * E); s.eWD = st.scratch(144 * E); s.eWdb = st.scratch(48 * E);
    s.erho = st.scratch(8 * E);
    s.Hpp = st.scratch(288 * (size_t)b.nfree); s.bp = st.scratch(8 * n); s.Hll = st.scratch(72 * (size_t)b.nact); s.bl = st.scratch(8 * nl);
    s.Dinv = st.scratch(72 * (size_t)b.nact); s.db = st.scratch(8 * nl); s.x = st.scratch(8 * (n + nl)); s.terms = st.scratch(8 * (n + nl));
  }
  if ((rc = st.layout()) != DVM_OK) return rc;
  for (int k = 0; k < K; k++) {
    const WinBuild& b = B[k]; const Slots& s = sl[k]; BaWin& v = views[k];
    std::memset(&v, 0, sizeof(v));
    v.P = b.P; v.L = b.L; v.E = b.E; v.nfree = b.nfree; v.nact = b.nact; v.nblk = b.nblk; v.iterations = windows[k].iterations;
    v.fx = windows[k].cam.fx; v.fy = windows[k].cam.fy; v.cx = windows[k].cam.cx; v.cy = windows[k].cam.cy; v.delta = windows[k].cam.huber_delta;
    v.poses = st.ptr<double>(s.poses); v.pts = st.ptr<double>(s.pts); v.poses_t = st.ptr<double>(s.poses_t); v.pts_t = st.ptr<double>(s.pts_t);
    v.out_poses = st.ptr<double>(s.out_poses); v.out_pts = st.ptr<double>(s.out_pts);
    v.pidx = st.ptr<int32_t>(s.pidx); v.lidx = st.ptr<int32_t>(s.lidx); v.free_pose = st.ptr<int32_t>(s.free_pose); v.act_pt = st.ptr<int32_t>(s.act_pt);
    v.e_pose = st.ptr<int32_t>(s.e_pose); v.e_point = st.ptr<int32_t>(s.e_point); v.e_obs = st.ptr<double>(s.e_obs); v.e_info = st.ptr<double>(s.e_info);
    v.e_A = st.ptr<double>(s.eA); v.e_B = st.ptr<double>(s.eB); v.e_w = st.ptr<double>(s.ew); v.e_W = st.ptr<double>(s.eW); v.e_WD = st.ptr<double>(s.eWD);
    v.e_Wdb = st.ptr<double>(s.eWdb); v.e_chi2 = st.ptr<double>(s.echi); v.e_rho = st.ptr<double>(s.erho); v.e_depth = st.ptr<uint8_t>(s.edepth);
    v.cam_start = st.ptr<int32_t>(s.cam_start); v.cam_edges = st.ptr<int32_t>(s.cam_edges); v.camp_edges = st.ptr<int32_t>(s.camp_edges);
    v.pt_start = st.ptr<int32_t>(s.pt_start); v.pt_edges = st.ptr<int32_t>(s.pt_edges);
    v.blk_i1 = st.ptr<int32_t>(s.blk_i1); v.blk_i2 = st.ptr<int32_t>(s.blk_i2); v.blk_start = st.ptr<int32_t>(s.blk_start);
    v.pair_k1 = st.ptr<int32_t>(s.pair_k1); v.pair_k2 = st.ptr<int32_t>(s.pair_k2);
    v.Hpp = st.ptr<double>(s.Hpp); v.bp = st.ptr<double>(s.bp); v.Hll = st.ptr<double>(s.Hll); v.bl = st.ptr<double>(s.bl); v.Dinv = st.ptr<double>(s.Dinv);
    v.db = st.ptr<double>(s.db); v.x = st.ptr<double>(s.x); v.terms = st.ptr<double>(s.terms);
    v.stats = st.ptr<dvm_ba_stats>(s.stats);
  }
  if ((rc = st.upload()) != DVM_OK) return rc;
  // dynamic LDS: the packed reduced system of the largest window + the fixed part; the attribute is raised to the kernel's maximum once per call
  int max_n = 0;
  for (int k = 0; k < K; k++) max_n = std::max(max_n, 6 * B[k].nfree);
  const size_t lds_bytes = sizeof(double) * ((size_t)max_n * (max_n + 1) / 2 + kWinLdsMisc);
  DVM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_window), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(double) * ((size_t)(6 * kWinMaxFree) * (6 * kWinMaxFree + 1) / 2 + kWinLdsMisc))));
  hipLaunchKernelGGL(k_ba_window, dim3(K), dim3(kWinThreads), lds_bytes, 0, st.ptr<BaWin>(views_slot), sw.d);
  DVM_HIP(hipGetLastError());
  if (stop_flag) {                     // g2o's forceStopFlag: written by another thread while the optimisation runs (LocalMapping.cc:305,359)
    hipEvent_t ev;
    DVM_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    DVM_HIP(hipEventRecord(ev, 0));
    while (hipEventQuery(ev) == hipErrorNotReady) *sw.h = *stop_flag ? 1 : 0;
    hipEventDestroy(ev);
  }
  if ((rc = st.download()) != DVM_OK) return rc;
  const auto t2 = std::chrono::steady_clock::now();
  for (int k = 0; k < K; k++) {
    const dvm_ba_window& w = windows[k];
    if (w.poses_out && B[k].P) std::memcpy(w.poses_out, outs[k].poses.data(), 56 * (size_t)B[k].P);
    if (w.points_out && B[k].L) std::memcpy(w.points_out, outs[k].pts.data(), 24 * (size_t)B[k].L);
    if (w.edge_chi2_out && B[k].E) std::memcpy(w.edge_chi2_out, outs[k].chi2.data(), 8 * (size_t)B[k].E);
    if (w.depth_positive_out && B[k].E) std::memcpy(w.depth_positive_out, outs[k].depth.data(), (size_t)B[k].E);
    if (stats) {
      stats[k] = outs[k].st;
      stats[k].ms_structure = std::chrono::duration<double, std::milli>(t1 - t0).count() / K;
      stats[k].ms_optimize = std::chrono::duration<double, std::milli>(t2 - t1).count();     // the whole batch: upload, the one launch, download
    }
  }
  return DVM_OK;
}

extern "C" {

int dvm_ba_optimize_windows(int device, const dvm_ba_window* windows, int K, const volatile uint8_t* stop_flag, dvm_ba_stats* stats) {
  return dvm_ba_optimize_windows_impl(device, windows, K, stop_flag, stats, true);
}

int dvm_f64_spec_eval(int device, const double* x, int n, double* out) {
  if (n < 0 || (n && (!x || !out))) return DVM_ERR_INVALID;
  if (n == 0) return DVM_OK;
  int rc = dvm_set_device(device);
  if (rc != DVM_OK) return rc;
  Stage st;
  const int ix = st.in(x, 8 * (size_t)n), io = st.out(out, 24 * (size_t)n);
  if ((rc = st.upload()) != DVM_OK) return rc;
  hipLaunchKernelGGL(k_f64_spec, dim3((n + 255) / 256), dim3(256), 0, 0, st.ptr<double>(ix), n, st.ptr<double>(io));
  DVM_HIP(hipGetLastError());
  return st.download();
}

}  // extern "C"
