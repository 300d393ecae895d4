// dvm_slam_amd/csrc/ba_kernels.hip -- FP64 bundle-adjustment kernels for gfx950.
//
// One Levenberg-Marquardt iteration of the reference's BA (g2o BlockSolver_6_3 + Levenberg,
// reference Thirdparty/g2o/g2o/core/{block_solver.hpp,optimization_algorithm_levenberg.cpp},
// src/Optimizer.cc:55-356,1030-1387, src/OptimizableTypes.cpp:136-155, src/CameraModels/Pinhole.cpp):
//   K8  k_edge_eval            per-edge residual, Huber weight, 2x3 / 2x6 Jacobians, Hpl block
//       k_point_accum          Hll (3x3), bl per landmark       (gather over the landmark's edges)
//       k_pose_accum           Hpp (6x6), bp per camera         (one wave per camera, fixed-order reduce)
//   K9  k_dinv, k_schur_blocks Dinv = (Hll+lambda I)^-1 ; Hschur block = Hpp - sum W Dinv W^T
//       k_schur_rhs            bschur = bp - sum W Dinv bl  (stored as the augmented row of S)
//   K10 k_chol_diag / k_chol_trsm / k_chol_update   blocked dense Cholesky on v_mfma_f64_16x16x4
//       k_chol_backsolve       L^T x = y
//   K11 k_point_backsub, k_update      xl = Dinv (bl - W^T xp); T <- exp(d) T, X <- X + d
// All accumulations are gathers with a fixed order: results are run-to-run deterministic.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "ba_kernels.h"
#include "f64_spec.h"
#include "jacobi4.h"

namespace dvm {

// ------------------------------------------------------------------------------- small algebra
__device__ __forceinline__ void quat_to_R(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
               tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void R_to_quat(const double* R, double* q) {
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
__device__ __forceinline__ void quat_normalize(double* q) {
  if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
__device__ __forceinline__ void mat3_vec(const double* R, const double* v, double* o) {
  o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}

// T <- exp(u) * T, u = (omega, upsilon): SE3Quat::exp + operator* + normalizeRotation
// (reference Thirdparty/g2o/g2o/types/se3quat.h:212-266, types_six_dof_expmap.h:71-74)
__device__ __forceinline__ void se3_oplus(double* T, const double* u) {
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
    // sin / cos / pow(theta, 3) of SE3Quat::exp: the double-precision spec the oracle evaluates too (csrc/f64_spec.h), not the device libm
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
  R_to_quat(R, dq);
  quat_normalize(dq);
  mat3_vec(Vm, u + 3, dt);
  quat_to_R(dq, Rd);
  mat3_vec(Rd, T, rt);
  const double* q = T + 3;
  nq[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  nq[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  nq[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  nq[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  quat_normalize(nq);
  T[0] = dt[0] + rt[0]; T[1] = dt[1] + rt[1]; T[2] = dt[2] + rt[2];
  T[3] = nq[0]; T[4] = nq[1]; T[5] = nq[2]; T[6] = nq[3];
}

// Huber (robust_kernel_impl.cpp:68-81); delta <= 0 means "no robust kernel"
__device__ __forceinline__ void robustify(double e, double delta, double& rho0, double& rho1) {
  if (delta <= 0 || e <= delta * delta) { rho0 = e; rho1 = 1.; }
  else { const double s = sqrt(e); rho0 = 2 * s * delta - delta * delta; rho1 = delta / s; }
}

__device__ __forceinline__ double ba_lambda(const BaView& V) { return V.lambda ? *V.lambda : V.lambda_v; }

// Block sum of `v` -> partial[blockIdx.x]; the last workgroup to arrive then reduces all partials exactly like
// k_reduce_sum (thread-strided sums, then the same tree: identical bits) and hands the result to the host (BaPublish).
template <bool MAX>
__device__ __forceinline__ void block_reduce_publish(double v, double* __restrict__ partial, const BaPublish& pub, int nb_part = 0) {
  __shared__ double s_red[256];
  __shared__ int s_last;
  const int tid = threadIdx.x, nb = nb_part ? nb_part : (int)gridDim.x;   // workgroups [0, nb) take part
  s_red[tid] = v;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) s_red[tid] = MAX ? fmax(s_red[tid], s_red[tid + off]) : s_red[tid] + s_red[tid + off];
    __syncthreads();
  }
  if (tid == 0) {
    // 8-byte agent-scope (write-through) store of the partial, drained before the arrival is counted: whoever sees the
    // count can read the partial with an agent-scope load.  No release fence: that would write back this XCD's whole L2
    // (the edge pass has just dirtied megabytes) once per workgroup -- measured 23 -> 58 us on k_edge_eval.
    __hip_atomic_store(partial + blockIdx.x, s_red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = __hip_atomic_fetch_add(pub.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nb - 1);
  }
  __syncthreads();
  if (!s_last) return;
  double acc = 0;
  for (int i = tid; i < nb; i += 256) {
    const double p = __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc = MAX ? fmax(acc, p) : acc + p;
  }
  s_red[tid] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) s_red[tid] = MAX ? fmax(s_red[tid], s_red[tid + off]) : s_red[tid] + s_red[tid + off];
    __syncthreads();
  }
  if (tid == 0) {
    __hip_atomic_store(pub.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(pub.dev_vals + pub.slot, s_red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (pub.publish) {
      // results of earlier kernels of this phase (other slots) are in dev_vals: kernel boundaries made them visible.  ALL of them
      // (and the failure flag) are fetched before the first store: alternating agent-scope loads and system-scope stores was six
      // dependent round trips on one thread -- 4-5 us at the very end of the kernel that publishes chi2 in every trial
      double dv[6];
#pragma unroll
      for (int i = 0; i < 6; i++) dv[i] = pub.dev_vals[i];
      const int f = pub.d_fail ? *pub.d_fail : 0;
#pragma unroll
      for (int i = 0; i < 6; i++) dv[i] = (i == pub.slot) ? s_red[0] : dv[i];
#pragma unroll
      for (int i = 0; i < 6; i++) __hip_atomic_store(pub.host_vals + i, dv[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(pub.host_vals + 6, (double)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (pub.spec) {
        // OptimizationAlgorithmLevenberg::solve's accept / reject and lambda update (optimization_algorithm_levenberg.cpp:
        // 113-131) exactly as dvm_ba_optimize takes them on the host, which checks the result bit for bit before it relies on
        // anything derived from it.  pow(t, 3) is formed with the rounding errors of both products carried along (the
        // correctly rounded cube, which is what libm returns for almost every t).
        const double tmp = f == 0 ? s_red[0] : 1.7976931348623157e308;
        const double scale = dv[2] + 1e-3;                     // computeScale() runs on whatever x holds, failed solve or not
        const double rho = (pub.cur_chi - tmp) / scale;
        double next = -1.0;
        if (pub.spec_mode == 1) next = 1e-5 * s_red[0];   // computeLambdaInit (_tau = 1e-5) on the max |diag| this workgroup has just reduced
        else if (rho > 0 && isfinite(tmp)) {
          const double t = 2 * rho - 1;
          const double t2 = t * t, e2 = __builtin_fma(t, t, -t2);
          const double t3 = t2 * t, e3 = __builtin_fma(t2, t, -t3);
          const double cube = t3 + (e3 + e2 * t);
          double alpha = 1. - cube;
          alpha = fmin(alpha, 2. / 3.);
          next = pub.lambda * fmax(1. / 3., alpha);
          if ((pub.cur_chi - tmp) * 1e3 < pub.cur_chi && pub.n_bad + 1 >= 3) next = -2.0;   // accepted, and the optimisation stops here
        }
        __hip_atomic_store(pub.spec, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pub.host_vals + 7, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __hip_atomic_store(pub.host_seq, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Sums of N per-thread doubles over a 256-thread workgroup, THROUGH LDS: every thread parks its values (pitch N + 1), thread
// (q, i) adds the 64 threads of wave q for value i in thread order, then the four wave sums are added -- a fixed order.  The
// alternative, xor-butterflies of __shfl_xor, is ds_bpermute_b32 twice per double and step: 24 cycles of the CU's LDS unit
// each (tools/valu_issue2.hip), 336 of them for 28 values -- 13 us per reduction with four waves sharing the unit.
// park: 256 * (N + 1) doubles, part: 4 * N doubles, out: N doubles (all in LDS; out is valid for every thread on return).
template <int N>
__device__ __forceinline__ void block_sum_lds(const double* v, double* park, double* part, double* out) {
  const int tid = threadIdx.x;
  double* mine = park + (size_t)tid * (N + 1);
#pragma unroll
  for (int i = 0; i < N; i++) mine[i] = v[i];
  __syncthreads();
  if (tid < 4 * N) {
    const int q = tid / N, i = tid - q * N;
    const double* col = park + (size_t)(64 * q) * (N + 1) + i;
    double s = 0;
#pragma unroll 16
    for (int l = 0; l < 64; l++) s += col[(size_t)l * (N + 1)];
    part[q * N + i] = s;
  }
  __syncthreads();
  if (tid < N) out[tid] = (part[tid] + part[N + tid]) + (part[2 * N + tid] + part[3 * N + tid]);
  __syncthreads();
}
// the same for one wave (64 lanes), no workgroup barrier: lane i < N returns the sum of value i over the lanes, in lane order
template <int N>
__device__ __forceinline__ double wave_sum_lds(const double* v, double* park /* 64 * (N + 1) doubles of this wave */) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_wave_barrier();
  double* mine = park + (size_t)lane * (N + 1);
#pragma unroll
  for (int i = 0; i < N; i++) mine[i] = v[i];
  __builtin_amdgcn_wave_barrier();
  double s = 0;
  if (lane < N) {
#pragma unroll 16
    for (int l = 0; l < 64; l++) s += park[(size_t)l * (N + 1) + lane];
  }
  __builtin_amdgcn_wave_barrier();
  return s;
}

// ------------------------------------------------------------------------------------------ K8
// JAC=false: residual / chi2 only (computeActiveErrors + activeRobustChi2 terms).
// JAC=true : additionally Jacobians A (2x3, point), B (2x6, pose), weights and Hpl block W (6x3).
// The Jacobian rows leave through LDS: a thread's rows are 128 B (e_lin: camera half) + 128 B (e_linA: landmark half) + 144 B (W),
// and per-thread row stores touch 64 different cache lines per instruction (1 344 line writes per wave for 172 lines of data when
// the halves still shared a 192-byte row: ~10 of the kernel's 28 us).  A wave's 64 rows are one contiguous 8 KB / 8 KB / 9 KB image
// in global memory: every lane parks its row in LDS (row pitch 18 doubles: 16-byte writes without bank conflicts), three times in
// turn, and the wave copies the image out with 16-byte stores, 1 KB per instruction.
constexpr int kLinPitch = 18;
template <bool JAC>
__global__ void __launch_bounds__(256) k_edge_eval(BaView V, BaPublish pub) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  __shared__ double2 s_rows[JAC ? 4 * 64 * kLinPitch / 2 : 1];
  double* park = reinterpret_cast<double*>(s_rows) + (size_t)(threadIdx.x >> 6) * 64 * kLinPitch;   // this wave's rows
  const int lane = threadIdx.x & 63;
  double Wv[18], Av[9];
  double rho0 = 0;
  if (k < V.E) {
    const int p = V.e_pose[k], l = V.e_point[k];
    const double* T = (JAC ? V.poses : V.poses_new) + 7 * (size_t)p;     // chi2-only evaluations look at the TRIAL state
    const double* X = (JAC ? V.points : V.points_new) + 3 * (size_t)l;
    double R[9], Xc[3];
    quat_to_R(T + 3, R);
    mat3_vec(R, X, Xc);
    Xc[0] += T[0]; Xc[1] += T[1]; Xc[2] += T[2];
    const double x = Xc[0], y = Xc[1], z = Xc[2];
    const double info = V.e_info[k];
    const double e0 = V.e_obs[2 * (size_t)k] - (V.fx * x / z + V.cx);
    const double e1 = V.e_obs[2 * (size_t)k + 1] - (V.fy * y / z + V.cy);
    const double chi2 = e0 * info * e0 + e1 * info * e1;
    const unsigned fl = V.e_flags ? V.e_flags[k] : 3u;
    const bool active = (fl & 1u) != 0;                 // g2o: level 0.  A level-1 edge is outside the active set: no error evaluation
    if (active) V.e_chi2[k] = chi2;                     // (its chi2() stays what it was), nothing in chi2 / H / b
    double rho1;
    robustify(chi2, (fl & 2u) ? V.delta : 0.0, rho0, rho1);
    if (!active) { rho0 = 0; rho1 = 0; }
    if (JAC) {
      const double J[6] = {-(V.fx / z), 0, V.fx * x / (z * z), 0, -(V.fy / z), V.fy * y / (z * z)};
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
      if (!active) {                                    // exact zeros whatever the geometry (a point on the camera plane: inf / NaN rows)
#pragma unroll
        for (int i = 0; i < 6; i++) A[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 12; i++) B[i] = 0.0;
      }
      const double w = active ? rho1 * info : 0.0;
      const double wr0 = active ? -info * e0 * rho1 : 0.0, wr1 = active ? -info * e1 * rho1 : 0.0;
      double2* mine = reinterpret_cast<double2*>(park + (size_t)lane * kLinPitch);
#pragma unroll
      for (int i = 0; i < 6; i++) mine[i] = make_double2(B[2 * i], B[2 * i + 1]);
      mine[6] = make_double2(w, wr0); mine[7] = make_double2(wr1, 0.0);
#pragma unroll
      for (int i = 0; i < 6; i++) Av[i] = A[i];
      Av[6] = w; Av[7] = wr0; Av[8] = wr1;
      const bool pose_free = V.pidx[p] >= 0;
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) Wv[3 * a + b] = pose_free ? w * (B[a] * A[b] + B[6 + a] * A[3 + b]) : 0.0;
    }
  }
  if (JAC) {
    const int first = blockIdx.x * 256 + (threadIdx.x & ~63);      // the wave's first edge
    const int nrow = min(64, V.E - first);                          // (<= 0 for a wave beyond the last edge)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double2* img = reinterpret_cast<double2*>(V.e_lin + (size_t)first * kEdgeLinStride);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int g = 64 * i + lane, row = g >> 3, c = g & 7;      // 16-byte chunk g of the image = chunk c of row `row`
      if (row < nrow) img[g] = *reinterpret_cast<const double2*>(park + (size_t)row * kLinPitch + 2 * c);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (k < V.E) {                                   // the landmark half the same way
      double2* mine = reinterpret_cast<double2*>(park + (size_t)lane * kLinPitch);
#pragma unroll
      for (int i = 0; i < 4; i++) mine[i] = make_double2(Av[2 * i], Av[2 * i + 1]);
      mine[4] = make_double2(Av[8], 0.0); mine[5] = make_double2(0.0, 0.0); mine[6] = make_double2(0.0, 0.0); mine[7] = make_double2(0.0, 0.0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double2* imga = reinterpret_cast<double2*>(V.e_linA + (size_t)first * kEdgeLinStride);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int g = 64 * i + lane, row = g >> 3, c = g & 7;
      if (row < nrow) imga[g] = *reinterpret_cast<const double2*>(park + (size_t)row * kLinPitch + 2 * c);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (k < V.E) {
      double2* mine = reinterpret_cast<double2*>(park + (size_t)lane * 18);
#pragma unroll
      for (int i = 0; i < 9; i++) mine[i] = make_double2(Wv[2 * i], Wv[2 * i + 1]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double2* imgw = reinterpret_cast<double2*>(V.e_W + (size_t)first * 18);
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int g = 64 * i + lane, row = g / 9;
      if (row < nrow) imgw[g] = *reinterpret_cast<const double2*>(park + 2 * (size_t)g);      // pitch 18 = the image itself
    }
  }
  block_reduce_publish<false>(rho0, V.partial, pub);   // fixed-order sum of rho0 over all edges -> host
}

// Sum `n` partials in a fixed order into out[slot]; single workgroup.
__global__ void __launch_bounds__(256) k_reduce_sum(const double* __restrict__ partial, int n, double* __restrict__ out, int slot) {
  __shared__ double s[256];
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[slot] = s[0];
}

// (Hll + lambda I)^-1 and its product with bl for one landmark
__device__ __forceinline__ void dinv_apply(const double* H, const double* bl, double lambda, double* __restrict__ D, double* __restrict__ db) {
  const double a = H[0] + lambda, b = H[1], c = H[2], d = H[3], e = H[4] + lambda, f = H[5], g = H[6], h = H[7], i = H[8] + lambda;
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  const double id = 1.0 / det;
  double o[9];
  o[0] = (e * i - f * h) * id; o[1] = (c * h - b * i) * id; o[2] = (b * f - c * e) * id;
  o[3] = (f * g - d * i) * id; o[4] = (a * i - c * g) * id; o[5] = (c * d - a * f) * id;
  o[6] = (d * h - e * g) * id; o[7] = (b * g - a * h) * id; o[8] = (a * e - b * d) * id;
#pragma unroll
  for (int k = 0; k < 9; k++) D[k] = o[k];
  db[0] = o[0] * bl[0] + o[1] * bl[1] + o[2] * bl[2];
  db[1] = o[3] * bl[0] + o[4] * bl[1] + o[5] * bl[2];
  db[2] = o[6] * bl[0] + o[7] * bl[1] + o[8] * bl[2];
}

// a KEPT landmark (BaView::kept_slot) is not eliminated: it contributes nothing to the Schur complement or its right-hand side
__device__ __forceinline__ void dinv_kept(const BaView& V, int l) {
#pragma unroll
  for (int k = 0; k < 9; k++) V.Dinv[9 * (size_t)l + k] = 0.0;
  V.db[3 * (size_t)l] = V.db[3 * (size_t)l + 1] = V.db[3 * (size_t)l + 2] = 0.0;
}
// Hll (3x3) and bl per landmark: thread per landmark, edges in input order.
__device__ __forceinline__ double point_accum_body(const BaView& V, int block, const double* __restrict__ spec) {   // returns max |Hll diagonal|
  const int l = block * 256 + threadIdx.x;
  if (l >= V.L) return 0.0;
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  const int e_end = V.pt_start[l + 1];
  for (int i0 = V.pt_start[l]; i0 < e_end; i0 += 4) {      // four edges in flight (index -> row: two dependent round trips each)
    int k[4];
#pragma unroll
    for (int u = 0; u < 4; u++) k[u] = (i0 + u < e_end) ? V.pt_edges[i0 + u] : -1;
    double Av[4][6], wv[4][3];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const double* lin = V.e_linA + (size_t)max(k[u], 0) * kEdgeLinStride;
#pragma unroll
      for (int j = 0; j < 6; j++) Av[u][j] = lin[j];
      wv[u][0] = lin[6]; wv[u][1] = lin[7]; wv[u][2] = lin[8];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (k[u] < 0) continue;
      const double* lin = Av[u];
      const double w = wv[u][0], wr0 = wv[u][1], wr1 = wv[u][2];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        b[a] += lin[a] * wr0 + lin[3 + a] * wr1;
#pragma unroll
        for (int c = 0; c < 3; c++) H[3 * a + c] += w * (lin[a] * lin[c] + lin[3 + a] * lin[3 + c]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 9; i++) V.Hll[9 * (size_t)l + i] = H[i];
#pragma unroll
  for (int i = 0; i < 3; i++) V.bl[3 * (size_t)l + i] = b[i];
  // the trial this state belongs to was accepted on the device (spec[0] = the next damping): the next trial's prologue work --
  // (Hll + lambda I)^-1 and Dinv bl, dinv_landmark's arithmetic on the same values -- is done here, from registers
  if (spec) {
    const double lambda = *spec;
    if (lambda >= 0 && V.pt_start[l + 1] != V.pt_start[l]) {
      if (V.nkept && V.kept_slot[l] >= 0) dinv_kept(V, l);
      else dinv_apply(H, b, lambda, V.Dinv + 9 * (size_t)l, V.db + 3 * (size_t)l);
    }
  }
  return V.pt_start[l + 1] > V.pt_start[l] ? fmax(fabs(H[0]), fmax(fabs(H[4]), fabs(H[8]))) : 0.0;   // (a landmark nobody observes is no vertex)
}

// Hpp (6x6) and bp per free camera: one wave per camera, lanes stride over the camera's edges, then a
// fixed-order xor-butterfly reduction (identical on every run).
// SPLIT (BaView::schur_wide, a local-BA window: a few cameras with hundreds of edges each on an otherwise empty chip): ONE
// camera per workgroup, its edges dealt over the four waves, the waves' sums added in wave order.
template <bool SPLIT>
__device__ __forceinline__ double pose_accum_body(const BaView& V, int block) {   // returns this thread's |Hpp diagonal entry| (or 0)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = SPLIT ? block : block * 4 + wave;
  if (fi >= V.nfree) return 0.0;
  const int p = V.free_pose[fi];
  double H[21], b[6];
#pragma unroll
  for (int i = 0; i < 21; i++) H[i] = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) b[i] = 0;
  // four edges per lane in flight: index -> row is a dependent pair of global round trips, and a camera of the BASELINE problem
  // gives a lane five edges -- one after the other that was ten round trips (the kernel's whole 15 us); same summation order
  const int e_end = V.ps_start[p + 1];
  const int step = SPLIT ? 256 : 64;
  for (int i0 = V.ps_start[p] + lane + (SPLIT ? 64 * wave : 0); i0 < e_end; i0 += 4 * step) {
    int k[4];
#pragma unroll
    for (int u = 0; u < 4; u++) k[u] = (i0 + step * u < e_end) ? V.ps_edges[i0 + step * u] : -1;
    double Bv[4][12], wv[4][3];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const double* lin = V.e_lin + (size_t)max(k[u], 0) * kEdgeLinStride;
#pragma unroll
      for (int j = 0; j < 12; j++) Bv[u][j] = lin[j];
      wv[u][0] = lin[12]; wv[u][1] = lin[13]; wv[u][2] = lin[14];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (k[u] < 0) continue;
      const double* B = Bv[u];
      const double w = wv[u][0], wr0 = wv[u][1], wr1 = wv[u][2];
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; a++) {
        b[a] += B[a] * wr0 + B[6 + a] * wr1;
#pragma unroll
        for (int c = 0; c <= a; c++) H[t++] += w * (B[a] * B[c] + B[6 + a] * B[6 + c]);
      }
    }
  }
  // lane t < 21 ends with H[t], lanes 21..26 with b (sum over the lanes through LDS, lane order: see wave_sum_lds)
  __shared__ double s_park[4][64 * 28];
  double v27[27];
#pragma unroll
  for (int i = 0; i < 21; i++) v27[i] = H[i];
#pragma unroll
  for (int i = 0; i < 6; i++) v27[21 + i] = b[i];
  double tot = wave_sum_lds<27>(v27, s_park[wave]);
  if (SPLIT) {
    __shared__ double s_w[4][27];
    if (lane < 27) s_w[wave][lane] = tot;
    __syncthreads();
    if (wave != 0) return 0.0;
    if (lane < 27) tot = ((s_w[0][lane] + s_w[1][lane]) + s_w[2][lane]) + s_w[3][lane];
  }
  if (lane < 21) {
    const int a = lane >= 15 ? 5 : lane >= 10 ? 4 : lane >= 6 ? 3 : lane >= 3 ? 2 : lane >= 1 ? 1 : 0;
    const int c = lane - a * (a + 1) / 2;
    double* out = V.Hpp + 36 * (size_t)fi;
    out[6 * a + c] = tot;
    out[6 * c + a] = tot;
    return a == c ? fabs(tot) : 0.0;
  } else if (lane < 27) {
    V.bp[6 * (size_t)fi + (lane - 21)] = tot;
  }
  return 0.0;
}

// one structurally non-zero tile of the reduced system back to "empty": zeros, identity on the padding rows of a diagonal tile
__device__ __forceinline__ void clear_tile(const BaView& V, int t) {
  const int ti = V.nz_tiles[2 * t], tj = V.nz_tiles[2 * t + 1];
  double* base = V.S + (size_t)ti * 64 * V.ldS + tj * 64;
  for (int i = threadIdx.x; i < 64 * 32; i += 256) {
    const int r = i >> 5, c2 = i & 31;
    reinterpret_cast<double2*>(base + (size_t)r * V.ldS)[c2] = make_double2(0.0, 0.0);
  }
  if (ti == tj && ti * 64 < V.n_pad) {
    __syncthreads();
    const int w = threadIdx.x;
    if (V.nkept && ti >= V.ncamt) {      // a tile of kept landmarks: 3 rows each, 21 to the tile
      if (w < 64 && (w >= 63 || (ti - V.ncamt) * 21 + w / 3 >= V.nkept)) base[(size_t)w * V.ldS + w] = V.damp_s;
    } else if (w < 64 && (w >= V.per_tile * V.dof || ti * V.per_tile + w / V.dof >= V.nfree)) base[(size_t)w * V.ldS + w] = V.damp_s;
  }
}

// Both accumulations in ONE launch (they are independent of each other): workgroups [0, nb_pose) take the cameras -- the
// longer job, so it starts first --, the next nb_point the landmarks (two dependent 14 us launches before), and the rest
// empty the reduced system for the trial that follows: every trial is preceded by a linearisation, the factor of the last
// trial is dead by then, and this launch runs while the host decides -- the trial's own first launch shrinks to the 79
// landmark workgroups.
// with_max: the launch also delivers max |diag| over Hpp and Hll (computeLambdaInit; k_max_diag as a launch of its own before).
__global__ void __launch_bounds__(256) k_accum(BaView V, int nb_pose, int nb_point, const double* __restrict__ spec, BaPublish pub, int with_max) {
  if (spec && *spec == -2.0) return;   // the device-side decision: accepted, and the optimisation ends with it -- nobody reads this linearisation
  double m;
  if ((int)blockIdx.x < nb_pose) m = V.schur_wide ? pose_accum_body<true>(V, blockIdx.x) : pose_accum_body<false>(V, blockIdx.x);
  else if ((int)blockIdx.x < nb_pose + nb_point) m = point_accum_body(V, blockIdx.x - nb_pose, spec);
  else { clear_tile(V, blockIdx.x - nb_pose - nb_point); return; }
  if (with_max) block_reduce_publish<true>(m, V.partial2, pub, nb_pose + nb_point);
}

// max |diag| over Hpp and Hll (computeLambdaInit) -> host, grid-wide with the last workgroup finishing the reduction.
__global__ void __launch_bounds__(256) k_max_diag(BaView V, BaPublish pub) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double m = 0;
  if (i < V.nfree * 6) m = fabs(V.Hpp[36 * (size_t)(i / 6) + 7 * (i % 6)]);
  if (i < V.L * 3 && V.pt_start[i / 3 + 1] > V.pt_start[i / 3]) m = fmax(m, fabs(V.Hll[9 * (size_t)(i / 3) + 4 * (i % 3)]));
  block_reduce_publish<true>(m, V.partial2, pub);
}

// ------------------------------------------------------------------------------------------ K9
__device__ __forceinline__ void dinv_landmark(const BaView& V, int l) {
  const double lambda = ba_lambda(V);
  if (l >= V.L) return;
  if (V.pt_start[l + 1] == V.pt_start[l]) return;  // landmark without observation: not a vertex of the graph
  if (V.nkept && V.kept_slot[l] >= 0) { dinv_kept(V, l); return; }
  dinv_apply(V.Hll + 9 * (size_t)l, V.bl + 3 * (size_t)l, lambda, V.Dinv + 9 * (size_t)l, V.db + 3 * (size_t)l);
}
__global__ void __launch_bounds__(256) k_dinv(BaView V) { dinv_landmark(V, blockIdx.x * 256 + threadIdx.x); }

// One wave per non-zero lower block (i1 >= i2) of the reduced camera matrix:
//   S[i1,i2] = Hpp[i1] (+lambda I) if i1 == i2  -  sum over co-observed landmarks of W1 Dinv W2^T
// `pairs` lists (edge of pose i1, edge of pose i2) per block (symbolic structure built once on host).
// Gather form: the pairs of a block reference W rows (144 B) and Dinv blocks (72 B) scattered over tens of MB.  Read
// lane-per-pair, every load instruction touches 64 different cache lines and the kernel runs at the texture addresser's
// line rate (measured 75 us, whatever the occupancy).  So a wave stages 32 pairs at a time through LDS with CHUNK-PARALLEL
// loads -- consecutive lanes read consecutive 16-byte chunks of the same row: ~7 rows per instruction instead of 64 -- and
// then computes from LDS, two lanes per pair (lane half hf owns rows 3 hf .. 3 hf + 2 of the 6x6 product).  Summation order:
// pair slots p, p + 32, ... per lane, then the xor-butterfly over the 32 slots -- fixed, identical on every run.
constexpr int kSchurStagePairs = 32, kSchurRow = 45;   // doubles per staged pair: W1 (18) | W2 (18) | Dinv (9); odd dword-pair stride: conflict-free
template <int NW>   // waves of the workgroup = waves sharing one block
__device__ __forceinline__ void schur_blocks_body(const BaView& V, int block) {
  __shared__ double s_stage[NW][kSchurStagePairs * kSchurRow];
  __shared__ int32_t s_idx[NW][3][kSchurStagePairs];
  const double lambda = ba_lambda(V);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = block;                                   // one block per WORKGROUP: its stages are dealt over the four waves
  const int i1 = V.blk_i1[blk], i2 = V.blk_i2[blk];
  const int hf = lane & 1, slot = lane >> 1;
  double* stage = s_stage[wave];
  int32_t (*idx)[kSchurStagePairs] = s_idx[wave];
  double acc[18];
#pragma unroll
  for (int i = 0; i < 18; i++) acc[i] = 0;
  // Software pipeline over the stages of the block (a wave walks its stages serially, so every dependent global round trip
  // inside a stage is paid n_stages times: measured 2.1 us per stage, 12 stages for the largest blocks): the (edge, edge, point)
  // indices are fetched two stages ahead, the rows one stage ahead into registers; a stage then costs max(compute, one load
  // latency) instead of three latencies plus the compute.
  constexpr int kIt = (kSchurStagePairs * 9 + 63) / 64;
  const int t_beg = V.blk_start[blk], t_end = V.blk_start[blk + 1];
  const int nst = (t_end - t_beg + kSchurStagePairs - 1) / kSchurStagePairs;
  auto load_idx = [&](int st, int& k1, int& k2, int& pt) {
    const int t = t_beg + st * kSchurStagePairs + lane;
    k1 = k2 = pt = 0;
    if (st < nst && lane < kSchurStagePairs && t < t_end) { k1 = V.pair_k1[t]; k2 = V.pair_k2[t]; pt = V.pair_pt[t]; }
  };
  double2 rw1[kIt], rw2[kIt];
  double rd[kIt];
  auto load_rows = [&](int st) {     // reads the indices of stage `st` from LDS
    const int np = min(kSchurStagePairs, t_end - (t_beg + st * kSchurStagePairs));
#pragma unroll
    for (int i = 0; i < kIt; i++) {
      const int c = i * 64 + lane;
      const int pr = (c * 57) >> 9, part = c - 9 * pr;     // c / 9 for c < 512
      if (pr < np) {
        rw1[i] = *reinterpret_cast<const double2*>(V.e_W + (size_t)idx[0][pr] * 18 + 2 * part);
        rw2[i] = *reinterpret_cast<const double2*>(V.e_W + (size_t)idx[1][pr] * 18 + 2 * part);
        rd[i] = V.Dinv[9 * (size_t)idx[2][pr] + part];
      }
    }
  };
  int a1, a2, a3, b1, b2, b3;
  load_idx(wave, a1, a2, a3);
  load_idx(wave + NW, b1, b2, b3);
  if (lane < kSchurStagePairs) { idx[0][lane] = a1; idx[1][lane] = a2; idx[2][lane] = a3; }
  __builtin_amdgcn_wave_barrier();
  if (wave < nst) load_rows(wave);
  for (int st = wave; st < nst; st += NW) {
    const int np = min(kSchurStagePairs, t_end - (t_beg + st * kSchurStagePairs));
    __builtin_amdgcn_wave_barrier();                       // stage st - 1 has been consumed (DS operations of a wave execute in order)
#pragma unroll
    for (int i = 0; i < kIt; i++) {
      const int c = i * 64 + lane;
      const int pr = (c * 57) >> 9, part = c - 9 * pr;
      if (pr < np) {
        double* row = stage + pr * kSchurRow;
        row[2 * part] = rw1[i].x; row[2 * part + 1] = rw1[i].y;
        row[18 + 2 * part] = rw2[i].x; row[18 + 2 * part + 1] = rw2[i].y;
        row[36 + part] = rd[i];
      }
    }
    if (st + NW < nst) {
      if (lane < kSchurStagePairs) { idx[0][lane] = b1; idx[1][lane] = b2; idx[2][lane] = b3; }
      __builtin_amdgcn_wave_barrier();
      load_rows(st + NW);                                  // in flight while stage st is computed
      load_idx(st + 2 * NW, b1, b2, b3);
    }
    __builtin_amdgcn_wave_barrier();
    if (slot < np) {
      const double* row = stage + slot * kSchurRow;
      const double* W1 = row + 9 * hf;                     // rows 3 hf .. 3 hf + 2 of W1 (6 x 3, row-major)
      const double* W2 = row + 18;
      const double* D = row + 36;
      double WD[9];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) WD[3 * a + b] = W1[3 * a] * D[b] + W1[3 * a + 1] * D[3 + b] + W1[3 * a + 2] * D[6 + b];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 6; b++) acc[6 * a + b] += WD[3 * a] * W2[3 * b] + WD[3 * a + 1] * W2[3 * b + 1] + WD[3 * a + 2] * W2[3 * b + 2];
    }
  }
  // Sum over the 32 pair slots THROUGH LDS: every lane parks its 18 partial sums in the (now idle) stage buffer, then lane
  // (hf, a, b) adds the 32 slots of its output in slot order.  ds_bpermute_b32 -- what __shfl_xor compiles to -- occupies the
  // CU's LDS unit for 24 cycles per instruction (tools/valu_issue2.hip): the 180 of them a double-precision butterfly over
  // 18 values needs were ~30 us of this kernel; 18 ds_write_b64 + 32 ds_read_b64 are ~200 cycles.
  __builtin_amdgcn_wave_barrier();
  {
    static_assert(2 * kSchurStagePairs * 19 <= kSchurStagePairs * kSchurRow, "the parking area reuses the stage buffer");
    double* red = stage + (size_t)lane * 19;                // 19-double pitch: odd dword-pair stride, conflict-free both ways
    if (lane < 2 * kSchurStagePairs) {
#pragma unroll
      for (int i = 0; i < 18; i++) red[i] = acc[i];
    }
  }
  if (NW > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
  if (threadIdx.x < 36) {
    const int ohf = lane / 18, oi = lane - 18 * ohf;        // output (row 3 ohf + oi / 6, column oi % 6)
    double v = 0;
    for (int w = 0; w < NW; w++) {                          // waves in order, slots in order: a fixed summation order
      const double* st_w = s_stage[w];
#pragma unroll 8
      for (int sl = 0; sl < kSchurStagePairs; sl++) v += st_w[(size_t)(2 * sl + ohf) * 19 + oi];
    }
    const int ra = 3 * ohf + oi / 6, cb = oi % 6;
    v = -v;
    if (i1 == i2) v += V.Hpp[36 * (size_t)i1 + 6 * ra + cb] + (ra == cb ? lambda * V.damp_s : 0.0);
    V.S[(size_t)(ba_row(i1) + ra) * V.ldS + ba_row(i2) + cb] = v;
  }
}

// bschur[i] = bp[i] - sum_{edges k of camera i} W_k (Dinv bl)_{point(k)}  -> augmented row n of S.
// SPLIT = false: one wave per camera (NW cameras per workgroup) -- the global problem has hundreds of cameras, and the rhs hides
// under the block workgroups.  SPLIT = true (BaView::schur_wide, a local-BA window): ONE camera per workgroup, its edges dealt
// over the NW waves (wave w takes edges w*64 + lane, then + 64 NW, ...): twenty cameras of 500 edges each were twenty waves
// walking three dependent global round trips per 256 edges, 27 us -- longer than the blocks.  Fixed summation order either way:
// per lane in edge order, xor butterfly over the lanes, then (SPLIT) the waves in order.
template <int NW, bool SPLIT>
__device__ __forceinline__ void schur_rhs_body(const BaView& V, int block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = SPLIT ? block : block * NW + wave;
  if (fi >= V.nfree) return;
  const int p = V.free_pose[fi];
  double acc[6] = {0, 0, 0, 0, 0, 0};
  // four edges per lane in flight: edge index -> (W row, landmark index -> Dinv bl) is a chain of three global round trips; same
  // summation order as one after the other
  const int e_end = V.ps_start[p + 1];
  const int first = V.ps_start[p] + lane + (SPLIT ? 64 * wave : 0), step = SPLIT ? 64 * NW : 64;
  for (int i0 = first; i0 < e_end; i0 += 4 * step) {
    int k[4], pt[4];
#pragma unroll
    for (int u = 0; u < 4; u++) k[u] = (i0 + step * u < e_end) ? V.ps_edges[i0 + step * u] : -1;
#pragma unroll
    for (int u = 0; u < 4; u++) pt[u] = V.e_point[max(k[u], 0)];
    double Wv[4][18], dv[4][3];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const double* W = V.e_W + (size_t)max(k[u], 0) * 18;
      const double* db = V.db + 3 * (size_t)pt[u];
#pragma unroll
      for (int j = 0; j < 18; j++) Wv[u][j] = W[j];
      dv[u][0] = db[0]; dv[u][1] = db[1]; dv[u][2] = db[2];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (k[u] < 0) continue;
#pragma unroll
      for (int a = 0; a < 6; a++) acc[a] += Wv[u][3 * a] * dv[u][0] + Wv[u][3 * a + 1] * dv[u][1] + Wv[u][3 * a + 2] * dv[u][2];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int a = 0; a < 6; a++) acc[a] += __shfl_xor(acc[a], off);
  if (SPLIT) {
    __shared__ double s_w[NW][6];
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 6; a++) s_w[wave][a] = acc[a];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      double t = 0;
      for (int w = 0; w < NW; w++) t += s_w[w][threadIdx.x];
      V.S[(size_t)V.n_pad * V.ldS + ba_row(fi) + threadIdx.x] = V.bp[6 * (size_t)fi + threadIdx.x] - t;
    }
    if (fi == 0 && threadIdx.x == 0) V.S[(size_t)V.n_pad * V.ldS + V.n_pad] = 1e200 * V.damp_s;  // augmented corner: keeps the last pivot positive
    return;
  }
  if (lane == 0) {
    for (int a = 0; a < 6; a++) V.S[(size_t)V.n_pad * V.ldS + ba_row(fi) + a] = V.bp[6 * (size_t)fi + a] - acc[a];
    if (fi == 0) V.S[(size_t)V.n_pad * V.ldS + V.n_pad] = 1e200 * V.damp_s;  // augmented corner: keeps the last pivot positive
  }
}

// The three rows of kept landmark j (BaView::kept_*): Hll + lambda I on the diagonal, the transposed Hpl block of every observation
// by a free camera left of it (dvm_ba_set_problem keeps no landmark with two observations from one camera: every block is written
// once), bl in the rhs row.  The tiles were emptied for this trial like every other non-zero tile.
__device__ __forceinline__ void schur_kept_body(const BaView& V, int j) {
  if (j >= V.nkept) return;
  const int l = V.kept_list[j];
  const int row0 = 64 * (V.ncamt + j / 21) + 3 * (j % 21);
  const double lambda = ba_lambda(V);
  double* const Sr = V.S + (size_t)row0 * V.ldS;
  const int i0 = V.pt_start[l], i1 = V.pt_start[l + 1];
  for (int i = i0 + (int)threadIdx.x; i < i1; i += (int)blockDim.x) {
    const int fi = V.pt_fi[i];
    if (fi < 0) continue;
    const double* W = V.e_W + (size_t)V.pt_edges[i] * 18;       // 6 (camera) x 3 (landmark)
    const int c0 = ba_row(fi);
#pragma unroll
    for (int bb = 0; bb < 3; bb++)
#pragma unroll
      for (int a = 0; a < 6; a++) Sr[(size_t)bb * V.ldS + c0 + a] = W[3 * a + bb];
  }
  if (threadIdx.x < 9) {
    const int a = threadIdx.x / 3, c = threadIdx.x % 3;
    if (c <= a) Sr[(size_t)a * V.ldS + row0 + c] = V.Hll[9 * (size_t)l + 3 * a + c] + (a == c ? lambda * V.damp_s : 0.0);
  } else if (threadIdx.x < 12) {
    V.S[(size_t)V.n_pad * V.ldS + row0 + (threadIdx.x - 9)] = V.bl[3 * (size_t)l + (threadIdx.x - 9)];
  }
}

// The reduced system and its right-hand side in ONE launch (independent of each other; both only need Dinv / db of the
// prologue): workgroups [0, nb_blk) build the 6x6 blocks, the rest the rhs row.
// Waves per block: 2 for the global problem (thousands of blocks of ~180 pairs: 12 KB of LDS per wave, every block resident at
// once), 8 for a local-BA window (BaView::schur_wide: ~100 blocks of ~500 pairs -- with two waves a block was eight dependent
// stages of gathers per wave on a chip that is otherwise empty: 28 us for 45 000 pairs).  The stages of a block are dealt
// round-robin over its waves and the waves' partial sums are added in wave order: deterministic for either width.
constexpr int kSchurWaves = 2, kSchurWavesWide = 8;
template <int kNW>
__global__ void __launch_bounds__(64 * kNW) k_schur(BaView V, int nb_blk, int nb_chunk, int nb_rhs, int* __restrict__ fail_reset) {
  constexpr int kSchurWaves = kNW;
  // speculative launch (ba_launch_schur_speculative): the damping comes from the device-side decision; a negative value = the
  // trial before this one was rejected, the host will start the next one itself.  There is no prologue launch in front of a
  // speculative one: the Cholesky failure flag is reset here.
  if (fail_reset) {
    if (*V.lambda < 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) *fail_reset = 0;
  }
  // The rhs workgroups come FIRST: a camera's rhs is one wave walking ~320 edges (17 us on its own); dispatched behind
  // thousands of block workgroups it would start late and set the kernel's tail.
  if ((int)blockIdx.x < nb_rhs) { schur_rhs_body<kSchurWaves, kNW == kSchurWavesWide>(V, blockIdx.x); return; }
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, each with its own 4 MB L2.  Workgroup b takes block
  // group (b % 8) * nb_chunk + b / 8, so an XCD works through a CONTIGUOUS run of the (camera-sorted) block list: the W rows
  // of its ~60 cameras (46 KB each) stay in that L2.
  const int b = (int)blockIdx.x - nb_rhs;
  if (b >= 8 * nb_chunk) { schur_kept_body(V, b - 8 * nb_chunk); return; }      // the rows of a kept landmark (behind the block workgroups)
  const int g = (b & 7) * nb_chunk + (b >> 3);
  if (g < nb_blk) schur_blocks_body<kSchurWaves>(V, g);
}

// ---- the landmark-chunk form (BaView::sl_*).  Workgroup = one chunk of landmarks (<= 64 landmarks, <= 512 free-camera rows): its W rows go
// to LDS once -- a row is then used by every pair it is in, where the per-block form fetched it once per pair --, each landmark's Dinv too; the
// chunk's (edge, edge) pairs are listed by block, and a work item is row a of one block's run: W1 Dinv (three values per pair, formed on the
// fly) against the six rows of W2, summed over the run's pairs in landmark order into one partial 6x6 block.  k_schur_reduce adds a block's
// partials in chunk order, negates, adds Hpp + lambda I on the diagonal and writes the block into tile space.  Both orders are fixed.
__global__ void __launch_bounds__(256) k_schur_lm(BaView V, int nb_rhs, int* __restrict__ fail_reset) {
  if (fail_reset) {
    if (*V.lambda < 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) *fail_reset = 0;
  }
  if ((int)blockIdx.x < nb_rhs) {      // (a local-BA window: one camera per workgroup, its ~500 edges dealt over the four waves)
    if (V.schur_wide) schur_rhs_body<4, true>(V, blockIdx.x); else schur_rhs_body<4, false>(V, blockIdx.x);
    return;
  }
  extern __shared__ double sl_lds[];
  const int c = (int)blockIdx.x - nb_rhs, tid = threadIdx.x;
  const int32_t* d = V.sl_desc + 8 * (size_t)c;
  const int row_off = d[0], nrows = d[1], lm_off = d[2], nlm = d[3], pair_off = d[4], run_off = d[6], nruns = d[7];
  const int npairs = d[5];
  double* Wb = sl_lds;                                  // [nrows][18]
  double* Di = Wb + (size_t)kSchurLmRows * 18;          // [nlm][9]
  int32_t* pairs = reinterpret_cast<int32_t*>(Di + kSchurLmLandmarks * 9);   // [npairs]  (in LDS: a pair step must not be a round trip to L2)
  int32_t* runs = pairs + kSchurLmPairs;                // [nruns + 1][2]
  // W rows: nine 16-byte pieces per row, consecutive lanes take consecutive pieces.  All of a thread's loads go out before the first LDS
  // store: a copy loop (index load -> row load -> ds_write per iteration) paid two round trips to L2 fourteen times over, 26 of the 65 us
  // this kernel took
  {
    constexpr int kIt = (kSchurLmRows * 9 + 255) / 256;
    int32_t ek[kIt];
#pragma unroll
    for (int u = 0; u < kIt; u++) { const int i = tid + 256 * u; ek[u] = i < nrows * 9 ? V.sl_row_edge[row_off + i / 9] : -1; }
    double2 wv[kIt];
#pragma unroll
    for (int u = 0; u < kIt; u++) {
      const int i = tid + 256 * u;
      if (ek[u] >= 0) wv[u] = *reinterpret_cast<const double2*>(V.e_W + (size_t)ek[u] * 18 + 2 * (i % 9));
    }
    constexpr int kPt = (kSchurLmPairs + 255) / 256, kLt = (kSchurLmLandmarks * 9 + 255) / 256, kUt = (2 * (kSchurLmRuns + 1) + 255) / 256;
    int32_t pv[kPt], lv[kLt], uv[kUt];
#pragma unroll
    for (int u = 0; u < kPt; u++) { const int i = tid + 256 * u; pv[u] = i < npairs ? V.sl_pairs[pair_off + i] : 0; }
#pragma unroll
    for (int u = 0; u < kLt; u++) { const int i = tid + 256 * u; lv[u] = i < nlm * 9 ? V.sl_lm[lm_off + i / 9] : -1; }
#pragma unroll
    for (int u = 0; u < kUt; u++) { const int i = tid + 256 * u; uv[u] = i < 2 * (nruns + 1) ? V.sl_runs[2 * (size_t)run_off + i] : 0; }
    double dv[kLt];
#pragma unroll
    for (int u = 0; u < kLt; u++) { const int i = tid + 256 * u; if (lv[u] >= 0) dv[u] = V.Dinv[9 * (size_t)lv[u] + i % 9]; }
#pragma unroll
    for (int u = 0; u < kIt; u++) {
      const int i = tid + 256 * u;
      if (ek[u] >= 0) { const int r = i / 9, part = i - 9 * r; Wb[18 * r + 2 * part] = wv[u].x; Wb[18 * r + 2 * part + 1] = wv[u].y; }
    }
#pragma unroll
    for (int u = 0; u < kPt; u++) { const int i = tid + 256 * u; if (i < npairs) pairs[i] = pv[u]; }
#pragma unroll
    for (int u = 0; u < kLt; u++) { const int i = tid + 256 * u; if (lv[u] >= 0) Di[i] = dv[u]; }
#pragma unroll
    for (int u = 0; u < kUt; u++) { const int i = tid + 256 * u; if (i < 2 * (nruns + 1)) runs[i] = uv[u]; }
  }
  __syncthreads();
  double* part0 = V.sl_part + 36 * (size_t)(run_off - c);          // (run_off counts one sentinel per earlier chunk)
  for (int t = tid; t < 6 * nruns; t += 256) {
    const int ru = t / 6, a = t - 6 * ru;
    const int q0 = runs[2 * ru + 1], q1 = runs[2 * ru + 3];
    double sv[6] = {0, 0, 0, 0, 0, 0};
    for (int q = q0; q < q1; q++) {
      const int pr = pairs[q], r1 = pr & 511, r2 = (pr >> 9) & 511;
      const double* W1 = Wb + 18 * r1 + 3 * a;
      const double* D = Di + 9 * (pr >> 18);
      const double* W2 = Wb + 18 * r2;
      const double w0 = W1[0], w1 = W1[1], w2 = W1[2];
      const double d0 = w0 * D[0] + w1 * D[3] + w2 * D[6], d1 = w0 * D[1] + w1 * D[4] + w2 * D[7], d2 = w0 * D[2] + w1 * D[5] + w2 * D[8];
#pragma unroll
      for (int b = 0; b < 6; b++) sv[b] += d0 * W2[3 * b] + d1 * W2[3 * b + 1] + d2 * W2[3 * b + 2];
    }
    double* o = part0 + 36 * (size_t)ru + 6 * a;
#pragma unroll
    for (int b = 0; b < 6; b++) o[b] = sv[b];
  }
}
__global__ void __launch_bounds__(256) k_schur_reduce(BaView V) {
  if (V.lambda && *V.lambda < 0) return;                // (speculative launch behind a rejected trial)
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 36 * V.nblk) return;
  const int blk = t / 36, e = t - 36 * blk, ra = e / 6, cb = e - 6 * ra;
  double v = 0;
  const int q0 = V.bp_start[blk], q1 = V.bp_start[blk + 1];
  for (int q = q0; q < q1; q += 8) {                    // eight partials at a time: their slots, then their values, then the eight additions in order
    int32_t sl[8];
#pragma unroll
    for (int j = 0; j < 8; j++) sl[j] = q + j < q1 ? V.bp_slots[q + j] : -1;
    double pv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) pv[j] = sl[j] >= 0 ? V.sl_part[36 * (size_t)sl[j] + e] : 0.0;
#pragma unroll
    for (int j = 0; j < 8; j++) v += pv[j];             // (+0.0 beyond the list: exact)
  }
  v = -v;
  const int i1 = V.blk_i1[blk], i2 = V.blk_i2[blk];
  if (i1 == i2) v += V.Hpp[36 * (size_t)i1 + e] + (ra == cb ? ba_lambda(V) * V.damp_s : 0.0);
  V.S[(size_t)(ba_row(i1) + ra) * V.ldS + ba_row(i2) + cb] = v;
}

// clears the structurally non-zero tiles of S (a trial rebuilds them); everything else is never touched and stays zero
// from the allocation-time memset: 197 of 1 326 tiles = 6 MB instead of 85 MB at 500 keyframes
__global__ void __launch_bounds__(256) k_zero_tiles(double* __restrict__ S, int ldS, const int32_t* __restrict__ nz) {
  const int ti = nz[2 * blockIdx.x], tj = nz[2 * blockIdx.x + 1];
  double* base = S + (size_t)ti * 64 * ldS + tj * 64;
  for (int i = threadIdx.x; i < 64 * 32; i += 256) {
    const int r = i >> 5, c2 = i & 31;
    reinterpret_cast<double2*>(base + (size_t)r * ldS)[c2] = make_double2(0.0, 0.0);
  }
}

// identity on the padding rows of the tiled system (rows 60..63 of every tile, cameras beyond nfree in the last tile)
__global__ void __launch_bounds__(256) k_pad_identity(BaView V) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= V.n_pad) return;
  const int w = r & 63;
  if (w >= V.per_tile * V.dof || (r >> 6) * V.per_tile + w / V.dof >= V.nfree) V.S[(size_t)r * V.ldS + r] = 1.0;
}

// Start of a BA trial: the damped landmark blocks are inverted (k_dinv); the Cholesky failure flag is reset on the way.  (The
// reduced system was emptied by the linearisation's launch, k_accum.)
__global__ void __launch_bounds__(256) k_trial_prologue(BaView V, int* __restrict__ fail) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *fail = 0;
  dinv_landmark(V, blockIdx.x * 256 + threadIdx.x);
}

// ----------------------------------------------------------------------------------------- K10
// Tile Cholesky of the lower triangle of S (row-major, leading dim ldS), NB = 64, over n1 = n_pad + 1 rows: the
// extra row carries bschur^T, so after the factorisation row n_pad holds y^T with L y = bschur (forward
// substitution for free).  The tile columns are processed LEVEL BY LEVEL of the elimination tree (ba_ordering.h):
// per level one k_chol_diag launch (a workgroup per column: block Cholesky + its inverse), one k_chol_trsm launch
// (all strips of the level as GEMMs with the inverses) and one k_chol_update launch (a workgroup per trailing tile,
// summing its contributing columns in a fixed order: deterministic, no atomics).
constexpr int NB = 64;

__device__ __forceinline__ double bcast_lane(double v, int src_lane) {  // src_lane must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2f64 lds_v2f64;
constexpr int LP = NB + 1;  // LDS pitch (doubles) of a 64x64 block

// Diagonal step kb: Cholesky factor L of the 64x64 block AND its inverse, one workgroup (4 waves).
// The block is processed as four 16-column panels:
//   A. wave 0: the panel (diagonal 16x16 sub-block and all rows below it) row-per-lane in registers, column
//      broadcasts by v_readlane -- the only strictly sequential part (4 x 16 columns instead of 64); the rows below
//      the diagonal are solved by the same instruction stream, and so is the sub-block's INVERSE: the lanes that have
//      no row left (16 / 32 / 48 of them in panels 1 / 2 / 3) carry the rows of the identity, which the panel turns
//      into the columns of the inverse
//   C. trailing sub-blocks:      A_ij -= X_i * X_j^T               (v_mfma_f64_16x16x4_f64, waves 0..2)
// then Linv by block forward substitution, Linv_ij = -Linv_ii * sum_k L_ik Linv_kj, again on MFMA: the
// C/D register layout of the f64 MFMA (row = (lane>>4) + 4*reg, col = lane&15) is exactly its B-operand
// layout for k-step = reg, so the running sum feeds the next product without touching LDS.
// The assembly runs beside the panels on the other waves, ordered by the panels' barriers alone: sub-block 0 (the one panel
// without idle lanes) is inverted by wave 3 beside panel 1, block (1, 0) follows beside panel 2, blocks (2, 0), (2, 1) and the
// three sums of block row 3 beside panel 3.  The tail after the last panel is four MFMAs per block of row 3, one block per
// wave, and the 16-byte store of L^-1.
#ifdef DVM_CHOL_DEBUG
__device__ long long g_chol_dbg[32];
#define DVM_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_chol_dbg[i] = __builtin_readcyclecounter(); } while (0)
#define DVM_STAMPW(i, w) do { if (blockIdx.x == 0 && threadIdx.x == 64 * (w)) g_chol_dbg[i] = __builtin_readcyclecounter(); } while (0)
#else
#define DVM_STAMP(i) do { } while (0)
#define DVM_STAMPW(i, w) do { } while (0)
#endif
typedef __attribute__((address_space(3))) double lds_f64;
// LDS of one tile factorisation (laid out by the caller, see k_chol_diag): Bm = the tile (rows beyond the matrix: identity), Li = where
// L^-1 goes; Pcol / Iv / Id are scratch (Id must hold the 16x16 identity).
struct DiagLds {
  lds_f64 (*Pcol)[NB];
  lds_f64 (*Iv)[16][17];
  lds_f64* Id;
  lds_f64* Bm;
  lds_f64* Li;
};
// One 64x64 tile in LDS: on return (all threads, behind a barrier) Li holds L^-1 -- every 16x16 block at or below the diagonal;
// the blocks above it are NOT written -- and Bm the blocks of L below the diagonal blocks.  256 threads.
// Hook: what waves 1..3 do for the CALLER beside the last panel, while wave 0 is in the column chain (k_chol_flow: the chain strip of the
// NEXT step travels global -> LDS under the factorisation): hook.begin() before, hook.end() behind their own work there.  NoDiagHook: no code.
struct NoDiagHook { static constexpr bool on = false; __device__ void begin() {} __device__ void end() {} };
template <class Hook>
__device__ __forceinline__ void chol_diag_tile(const DiagLds& D, int* __restrict__ fail, Hook& hook) {
  lds_f64 (*const Pcol)[NB] = D.Pcol;
  lds_f64 (*const Iv)[16][17] = D.Iv;
  lds_f64* const Id = D.Id;
  lds_f64* const Bm = D.Bm;
  lds_f64* const Li = D.Li;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  DVM_STAMP(1);
  const int lr = lane & 15, lq = lane >> 4;
  // inverse of the 16x16 diagonal sub-block bb by one wave: column `lane` of the inverse by forward substitution, L_it straight
  // from LDS (uniform address = broadcast read).  Only sub-block 0 needs it (beside panel 1): the inverses of sub-blocks 1..3
  // fall out of wave 0's own panel, see "identity rows" below.
  auto invert_block = [&](int bb) {
    const int ob = 16 * bb;
    // 1 / L_ii from the L_ii the panel left in the tile (d * rsqrt(d)): one division, lane i's, instead of a store per column there
    const double rl = 1.0 / Bm[(ob + (lane & 15)) * LP + ob + (lane & 15)];
    double y[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      double s0 = (lane == i) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
      for (int t = 0; t < i; t++) {
        const double l = Bm[(ob + i) * LP + ob + t];
        if (t & 1) s1 = __builtin_fma(-l, y[t], s1); else s0 = __builtin_fma(-l, y[t], s0);
      }
      y[i] = (s0 + s1) * bcast_lane(rl, i);
    }
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const double v = (lane <= i) ? y[i] : 0.0;
        Iv[bb][lane][i] = v;
        Li[(ob + i) * LP + ob + lane] = v;
      }
    }
  };
  // (The factor of the diagonal tile itself is NOT written back: nothing reads it -- the strips below are solved with L^-1, the
  //  back substitution uses L^-1 and the strips.  S keeps the tile as the Schur complement left it.)
  // L^-1 block (i, j), i > j:  -Iv_i * sum_{k = j .. i - 1} L_ik Linv_kj  (blocks (k, j), k < i, must be in Li already), one wave;
  // the C/D register layout of the f64 MFMA is its B-operand layout for k-step = reg, so the running sum feeds the product
  // with Iv_i without touching LDS
  auto linv_sum = [&](int i, int j) {
    double4_t t = {0, 0, 0, 0};
    for (int k = j; k < i; k++) {
#pragma unroll
      for (int kk = 0; kk < 16; kk += 4)
        t = __builtin_amdgcn_mfma_f64_16x16x4f64(Bm[(16 * i + lr) * LP + 16 * k + kk + lq],
                                                 k == j ? Iv[j][lr][kk + lq] : Li[(16 * k + kk + lq) * LP + 16 * j + lr], t, 0, 0, 0);
    }
    return t;
  };
  auto linv_finish = [&](int i, int j, double4_t t) {
    double4_t r4 = {0, 0, 0, 0};
#pragma unroll
    for (int st = 0; st < 4; st++) r4 = __builtin_amdgcn_mfma_f64_16x16x4f64(Iv[i][4 * st + lq][lr], t[st], r4, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) Li[(16 * i + lq + 4 * r) * LP + 16 * j + lr] = -r4[r];
  };
  auto linv_block = [&](int i, int j) { linv_finish(i, j, linv_sum(i, j)); };
  // the inverse of diagonal sub-block bb, which wave 0's panel left in Iv, into its place in Li (for the final store), one wave
  auto copy_inverse = [&](int bb) {
#pragma unroll
    for (int q = 0; q < 4; q++) Li[(16 * bb + lq + 4 * q) * LP + 16 * bb + lr] = Iv[bb][lr][lq + 4 * q];
  };
  // trailing update of sub-block (i, j) with the columns of panel bs:  A_ij -= X_i X_j^T, one wave
  auto trail = [&](int bs, int i, int j) {
    const int oi = 16 * i, oj = 16 * j, os = 16 * bs;
    double4_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; k += 4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Bm[(oi + lr) * LP + os + k + lq], Bm[(oj + lr) * LP + os + k + lq], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) Bm[(oi + lq + 4 * r) * LP + oj + lr] -= acc[r];
  };
  // block row 3 of L^-1: its three sums are formed beside the last panel (one per wave 1..3), only the product with Iv_3 -- four
  // MFMAs -- is left for the tail
  double4_t t3a = {0, 0, 0, 0};
  for (int b = 0; b < 4; b++) {
    const int o = 16 * b, nrows = NB - o;
    DVM_STAMP(2 + 3 * b);
    if (Hook::on && b == 3 && wave != 0) hook.begin();
    if (wave == 0) {
      // ---- A: the whole 16-column PANEL (diagonal sub-block + every row below it) in registers, lane = panel row;
      // the rows below the diagonal sub-block ride along in the same instructions (no separate triangular solve).
      // Right-looking: a finished column j updates every later column c of every row, a_c -= L_ij * L_cj.  The factor
      // L_cj is the same for all lanes: column j is published to LDS once and read back with uniform-address (broadcast)
      // reads -- except for c = j + 1, the next pivot column, which gets it by v_readlane.  The 16 columns are ONE
      // straight-line stream written by tools/gen_chol_panel.py (chol_panel.inc): this wave owns its SIMD and issues in order,
      // nothing overlaps, so the panel costs the sum of its instructions' issue times (5.3 cycles an FP64 operation, 8.3
      // when it needs the previous one, ~20 the rsq, ~30 a store behind an exec-mask round trip) -- the stream carries
      // nothing the factorisation does not need (no sqrt for L_jj, which nobody reads; one positivity test per panel; the
      // 1 / L_jj store from every lane) and a filler between any two dependent chain steps.  Cycle stamps per panel,
      // 2.4 GHz: 7 400 with every L_cj by v_readlane, 5 850 as the rolled loop the compiler scheduled (366 / column),
      // 3 850-4 150 now (250 / column); chain alone (no updates, no LDS) 2 600.
      double a[16];
      // IDENTITY ROWS: panels 1..3 have 16, 32, 48 lanes without a row.  Sixteen of them carry the rows of the identity: the
      // panel turns a row a_i below the diagonal sub-block into a_i L^-T, so the row e_i comes out as row i of L^-T = column i of
      // the sub-block's inverse -- by the very instructions that factorise the panel, none added: the lane's source and
      // destination pointers differ, that is all.  (Before: a second wave inverted the sub-block from LDS, 2 800 cycles beside
      // the next panel or, for the last one, 1 300 cycles behind this wave.)
      const int idl = lane - nrows;                          // 0..15 in panels 1..3: this lane carries e_idl
      const bool has_row = lane < nrows, is_id = b >= 1 && idl >= 0 && idl < 16;
      const lds_f64* const src = has_row ? Bm + (o + lane) * LP + o : Id + 16 * (idl & 15);
#pragma unroll
      for (int c = 0; c < 16; c++) a[c] = ((has_row || is_id) && (lane >= 16 || c <= lane)) ? src[c] : 0.0;
      bool bad = false;
      double d0 = bcast_lane(a[0], 0);          // pivot a_00, wave-uniform
#include "chol_panel.inc"
      // rows of L back to Bm; identity lanes: Iv_b[column idl][row c] = (L^-1)_{c, idl} (exact zeros above the diagonal, c < idl)
      lds_f64* const dst = has_row ? Bm + (o + lane) * LP + o : &Iv[b][idl & 15][0];
      if (has_row || is_id) {
#pragma unroll
        for (int c = 0; c < 16; c++) dst[c] = (lane < 16 && c > lane) ? 0.0 : a[c];
      }
      // (write-through: inside k_chol_flow other workgroups -- other XCDs -- read the flag before the launch ends: a plain store stayed in this
      //  XCD's L2 and the back substitution wrote x of a failed solve, now and then)
      if (bad && lane == 0) __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (b >= 1) {
      // beside the panel, no flags (everything is ordered by the panels' barriers): first the trailing updates of the PREVIOUS
      // panel that the current one does not need (sub-blocks right of its block column, see C below), then the pieces of L^-1
      // whose inputs are final: panel 1 -- wave 3 inverts sub-block 0; panel 2 -- wave 1 assembles block (1, 0); panel 3 --
      // blocks (2, 0), (2, 1) and the sums of block row 3, one per wave
      if (b == 1) {
        if (wave == 1) trail(0, 2, 2); else if (wave == 2) { trail(0, 3, 2); trail(0, 3, 3); } else invert_block(0);
      } else if (b == 2) {
        if (wave == 1) linv_block(1, 0);
        else if (wave == 2) copy_inverse(1);
        else trail(1, 3, 3);
      } else {
        if (wave == 3) { linv_block(2, 0); t3a = linv_sum(3, 0); }
        else if (wave == 1) { linv_block(2, 1); t3a = linv_sum(3, 1); }
        else { t3a = linv_sum(3, 2); copy_inverse(2); }
      }
    }
    if (Hook::on && b == 3 && wave != 0) hook.end();
    DVM_STAMP(3 + 3 * b);
    if (b == 3) { DVM_STAMPW(18, 1); DVM_STAMPW(19, 2); DVM_STAMPW(20, 3); }
    __syncthreads();
    DVM_STAMP(4 + 3 * b);
    if (b < 3) {
      // ---- C: trailing sub-blocks.  Only the next panel's block column (i, b + 1), i > b, is on the critical path: one wave
      // each, one round; the sub-blocks right of it are updated beside the next panel (A above) -- same operands, same order per
      // sub-block (panel b's contribution lands before panel b + 1's, a barrier in between), same bits.
      if (wave >= 1 && b + wave < 4) trail(b, b + wave, b + 1);
      __syncthreads();
    }
  }
  // tail: only L^-1 block row 3 is left (Iv_3 and block rows 0..2 were finished beside the last panel)
  DVM_STAMP(14);
  if (wave >= 1) linv_finish(3, wave == 3 ? 0 : wave, t3a);
  else copy_inverse(3);
  DVM_STAMP(15);
  __syncthreads();
}

__device__ __forceinline__ void chol_diag_tile(const DiagLds& D, int* __restrict__ fail) { NoDiagHook h; chol_diag_tile(D, fail, h); }

__global__ void __launch_bounds__(256) k_chol_diag(double* __restrict__ S, int ldS, int n1, const int32_t* __restrict__ cols,
                                                  int* __restrict__ fail, double* __restrict__ Linv_all) {
  DVM_STAMP(0);
  const int kb = cols[blockIdx.x];
  // One LDS block, laid out by hand: everything that is read with CONSTANT (wave-uniform) addresses sits in the first 64 KB,
  // where a ds instruction's 16-bit offset field reaches it -- left to the compiler, Li / Iv / Pcol landed above 64 KB and every
  // such address became a v_mov of a literal parked in an AGPR (hundreds of them in the kernel's prologue).
  __shared__ __attribute__((aligned(16))) double smem[16 * NB + 4 * 16 * 17 + 16 * 16 + 2 * NB * LP];
  lds_f64* const lds = (lds_f64*)smem;
  lds_f64 (*const Pcol)[NB] = (lds_f64 (*)[NB])lds;                                   // the panel's finished columns, one row per column: broadcast source for the updates
  lds_f64 (*const Iv)[16][17] = (lds_f64 (*)[16][17])(lds + 16 * NB);                // inverses of the diagonal sub-blocks, TRANSPOSED: Iv[b][column][row]
  lds_f64* const Id = lds + 16 * NB + 4 * 16 * 17;                            // 16x16 identity: the rows the idle lanes of a panel carry
  lds_f64* const Bm = Id + 16 * 16;
  lds_f64* const Li = Bm + NB * LP;
  const int tid = threadIdx.x;
  const int k0 = kb * NB;
  const int kw = min(NB, n1 - k0);
  {
    // the whole tile with 8 independent 16-byte loads per thread, issued back to back (a load per loop iteration behind a
    // branch serialised 16 global round trips: ~13 of the kernel's 27 us); rows beyond the matrix are clamped and replaced
    double2 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int r = 8 * i + (tid >> 5), c = 2 * (tid & 31);
      v[i] = *reinterpret_cast<const double2*>(S + (size_t)(k0 + min(r, kw - 1)) * ldS + k0 + c);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int r = 8 * i + (tid >> 5), c = 2 * (tid & 31);
      const bool in = r < kw;
      // rows beyond the matrix: identity.  (What lands ABOVE the diagonal is never read: the panel masks the upper triangle of
      // its diagonal sub-block, everything else works on blocks at or below the diagonal.)
      Bm[r * LP + c] = in ? v[i].x : (r == c ? 1.0 : 0.0);
      Bm[r * LP + c + 1] = in ? v[i].y : (r == c + 1 ? 1.0 : 0.0);
    }
  }
  Id[tid] = (tid >> 4) == (tid & 15) ? 1.0 : 0.0;
  __syncthreads();
  const DiagLds D = {Pcol, Iv, Id, Bm, Li};
  chol_diag_tile(D, fail);
  DVM_STAMP(16);
  double* Lo = Linv_all + (size_t)kb * NB * NB;
#pragma unroll
  for (int k = 0; k < 8; k++) {            // 16-byte stores: a wave writes 1 KB per instruction
    const int r = 8 * k + (tid >> 5), c = 2 * (tid & 31);
    const bool lower = (c >> 4) <= (r >> 4);     // the 16x16 blocks above the diagonal were never written in LDS: zeros from here
    const double2 v = lower ? make_double2(Li[r * LP + c], Li[r * LP + c + 1]) : make_double2(0.0, 0.0);
    *reinterpret_cast<double2*>(Lo + r * NB + c) = v;
  }
  DVM_STAMP(17);
}
// x of tile column kb, row t (value v) into the compact solution vector: a camera tile holds per_tile unknowns of dof rows; a tile of kept
// landmarks (kb >= ncamt, BaView::kept_*) 21 landmarks of 3 rows, whose x lives behind the cameras' at 3 * landmark
__device__ __forceinline__ void write_x(double* __restrict__ x, int kb, int t, double v, int nfree, int per_tile, int dof, int ncamt, int nkept,
                                        const int32_t* __restrict__ kept_list) {
  if (nkept > 0 && kb >= ncamt) {
    const int j = (kb - ncamt) * 21 + t / 3;
    if (t < 63 && j < nkept) x[dof * (size_t)nfree + 3 * (size_t)kept_list[j] + t % 3] = v;
    return;
  }
  const int cam = kb * per_tile + t / dof;
  if (t < per_tile * dof && cam < nfree) x[dof * (size_t)cam + t % dof] = v;
}
// The TOP PAIR of the elimination order in one workgroup (BaTileSchedule::pair_a / pair_b): column A, whose only tiles below
// the diagonal are (B, A) and the rhs row, and the root column B.  As separate launches this is diag(A) -> panel solve + update
// -> diag(B) -> two hops of the back substitution: five dependent kernels whose tiles travel through memory between them
// (~34 us of an iteration's solve at 500 keyframes, and the WHOLE solve of a local-BA window of <= 20 free keyframes).  Here
// the three tiles and the two rhs pieces are loaded once, everything else happens in LDS:
//   L_A, L_A^-1 (chol_diag_tile) ; X = (B,A) L_A^-T (MFMA, only the k-blocks L_A^-1 has) ; (B,B) -= X X^T (MFMA, lower blocks) ;
//   L_B, L_B^-1 ; y_A = L_A^-1 r_A, y_B = L_B^-1 (r_B - X y_A), x_B = L_B^-T y_B, x_A = L_A^-T (y_A - X^T x_B).
// Four tile buffers (BmA -> X, raw (B,A) -> L_B^-1, BmB, L_A^-1) = 133 KB of the CU's 160 KB.  Nothing but x leaves: no one
// else reads these columns' factor (their descendants need x_A / x_B, which go to xrow like any other column's).
__global__ void __launch_bounds__(256) k_chol_pair(const double* __restrict__ S, int ldS, int n_pad, int kbA, int kbB, int nfree, int per_tile,
                                                  int dof, double* __restrict__ xrow, double* __restrict__ x, int* __restrict__ fail,
                                                  int ncamt, int nkept, const int32_t* __restrict__ kept_list) {
  __shared__ __attribute__((aligned(16))) double smem[16 * NB + 4 * 16 * 17 + 16 * 16 + 4 * NB * LP + 10 * NB];
  lds_f64* const lds = (lds_f64*)smem;
  lds_f64 (*const Pcol)[NB] = (lds_f64 (*)[NB])lds;
  lds_f64 (*const Iv)[16][17] = (lds_f64 (*)[16][17])(lds + 16 * NB);
  lds_f64* const Id = lds + 16 * NB + 4 * 16 * 17;
  lds_f64* const T1 = Id + 16 * 16;          // tile (A,A): L_A; then X = L(B,A)
  lds_f64* const T2 = T1 + NB * LP;          // tile (B,A) as the Schur complement left it; then L_B^-1
  lds_f64* const T3 = T2 + NB * LP;          // tile (B,B)
  lds_f64* const T4 = T3 + NB * LP;          // L_A^-1
  lds_f64* const vr0 = T4 + NB * LP;         // r_A -> y_A -> (y_A - X^T x_B)
  lds_f64* const vr1 = vr0 + NB;             // r_B -> y_B
  lds_f64* const vx0 = vr1 + NB;
  lds_f64* const vx1 = vx0 + NB;
  lds_f64* const vt = vx1 + NB;              // a product's result before it is combined
  lds_f64* const part = vt + NB;             // [4][NB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int a0 = kbA * NB, b0 = kbB * NB;
  {
    double2 va[8], vb[8], vc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int r = 8 * i + (tid >> 5), c = 2 * (tid & 31);
      va[i] = *reinterpret_cast<const double2*>(S + (size_t)(a0 + r) * ldS + a0 + c);
      vb[i] = *reinterpret_cast<const double2*>(S + (size_t)(b0 + r) * ldS + a0 + c);
      vc[i] = *reinterpret_cast<const double2*>(S + (size_t)(b0 + r) * ldS + b0 + c);
    }
    double rv = 0;
    if (tid < 2 * NB) rv = S[(size_t)n_pad * ldS + (tid < NB ? a0 + tid : b0 + tid - NB)];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int r = 8 * i + (tid >> 5), c = 2 * (tid & 31);
      T1[r * LP + c] = va[i].x; T1[r * LP + c + 1] = va[i].y;
      T2[r * LP + c] = vb[i].x; T2[r * LP + c + 1] = vb[i].y;
      T3[r * LP + c] = vc[i].x; T3[r * LP + c + 1] = vc[i].y;
    }
    if (tid < NB) vr0[tid] = rv; else if (tid < 2 * NB) vr1[tid - NB] = rv;
  }
  Id[tid] = (tid >> 4) == (tid & 15) ? 1.0 : 0.0;
  __syncthreads();
  // the 16x16 blocks above the diagonal of an inverse are never written by chol_diag_tile: zeroed, so that what follows can
  // treat it as a full matrix
  auto zero_upper = [&](lds_f64* M) {
    for (int i = tid; i < 6 * 256; i += 256) {
      const int blk = i >> 8, e = i & 255;
      const int bi = blk < 3 ? 0 : blk < 5 ? 1 : 2, bj = blk < 3 ? blk + 1 : blk < 5 ? blk - 1 : 3;
      M[(16 * bi + (e >> 4)) * LP + 16 * bj + (e & 15)] = 0.0;
    }
  };
  // out[c] = sum_k M[c][k] v[k]   (TR: sum_k M[k][c] v[k]), in a fixed order: 16 terms per thread, then the four quarters
  auto matvec = [&](const lds_f64* M, bool TR, const lds_f64* v, lds_f64* out) {
    const int c = tid & 63, q = tid >> 6;
    double sacc = 0;
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
      const int k = 16 * q + kk;
      sacc += (TR ? M[k * LP + c] : M[c * LP + k]) * v[k];
    }
    part[q * NB + c] = sacc;
    __syncthreads();
    if (tid < NB) out[tid] = (part[tid] + part[NB + tid]) + (part[2 * NB + tid] + part[3 * NB + tid]);
    __syncthreads();
  };
  const DiagLds DA = {Pcol, Iv, Id, T1, T4};
  chol_diag_tile(DA, fail);
  zero_upper(T4);
  __syncthreads();
  // X = (B,A) L_A^-T: wave w takes block row w; column block bj only needs the k-blocks 0..bj (L_A^-1 is lower triangular)
  {
    double4_t xb[4];
#pragma unroll
    for (int bj = 0; bj < 4; bj++) {
      double4_t acc = {0, 0, 0, 0};
      for (int k = 0; k < 16 * (bj + 1); k += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T2[(16 * wave + lr) * LP + k + lq], T4[(16 * bj + lr) * LP + k + lq], acc, 0, 0, 0);
      xb[bj] = acc;
    }
#pragma unroll
    for (int bj = 0; bj < 4; bj++)
#pragma unroll
      for (int r = 0; r < 4; r++) T1[(16 * wave + lq + 4 * r) * LP + 16 * bj + lr] = xb[bj][r];
  }
  __syncthreads();
  matvec(T4, false, vr0, vr0);                 // y_A = L_A^-1 r_A   (in place: every thread has read v before the first barrier)
  matvec(T1, false, vr0, vt);                  // X y_A
  if (tid < NB) vr1[tid] -= vt[tid];
  // (B,B) -= X X^T, the ten blocks at or below the diagonal dealt over the waves
  {
    const int bis[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3}, bjs[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};
    for (int t = wave; t < 10; t += 4) {
      const int bi = bis[t], bj = bjs[t];
      double4_t acc = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < NB; k += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T1[(16 * bi + lr) * LP + k + lq], T1[(16 * bj + lr) * LP + k + lq], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) T3[(16 * bi + lq + 4 * r) * LP + 16 * bj + lr] -= acc[r];
    }
  }
  __syncthreads();
  const DiagLds DB = {Pcol, Iv, Id, T3, T2};
  chol_diag_tile(DB, fail);
  zero_upper(T2);
  __syncthreads();
  matvec(T2, false, vr1, vr1);                 // y_B
  matvec(T2, true, vr1, vx1);                  // x_B = L_B^-T y_B
  matvec(T1, true, vx1, vt);                   // X^T x_B
  if (tid < NB) vr0[tid] -= vt[tid];
  __syncthreads();
  matvec(T4, true, vr0, vx0);                  // x_A = L_A^-T (y_A - X^T x_B)
  // g2o's linear solver leaves _x untouched when the factorisation fails (see k_chol_backsolve)
  const bool good = __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
  if (tid < 2 * NB) {
    const int t = tid & 63, kb = tid < NB ? kbA : kbB;
    const double v = tid < NB ? vx0[t] : vx1[t];
    __hip_atomic_store(xrow + kb * NB + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (good) write_x(x, kb, t, v, nfree, per_tile, dof, ncamt, nkept, kept_list);
  }
}

#ifdef DVM_CHOL_DEBUG
extern "C" int dvm_debug_chol_stamps(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chol_dbg), sizeof(long long) * 32); }
#endif


// Panel solve of step kb: strip i (64 rows below the diagonal block) becomes X = A_ik * Linv_kk^T
// (a 64x64x64 product on v_mfma_f64_16x16x4_f64) -- the triangular solve as a GEMM, IN PLACE.  One workgroup per 16-ROW
// slice of the strip (grid = 4 x strips), each wave one 16x16 block of it: an f64 MFMA occupies its SIMD for 64 cycles, so a
// whole tile on one workgroup is 1.7 us of matrix pipe alone.  The split is by rows only: a workgroup reads and overwrites
// nothing but its own rows (a split by columns would let one workgroup overwrite A entries another is still reading).
constexpr int QP = 66;   // LDS pitch (doubles): conflict-free for the MFMA operand reads
// (xrow_tag != null on the first solve launch of a factorisation: workgroup 0 also fills the back substitution's hand-off
// slots with kXTag -- see k_chol_backsolve)
constexpr unsigned long long kXTag = 0x7FF8DEADBEEF0002ull;   // a NaN payload no arithmetic produces
__device__ __forceinline__ void tag_xrow(double* __restrict__ xrow, int n) {
  for (int i = threadIdx.x; i < n; i += 256) xrow[i] = __longlong_as_double((long long)kXTag);
}
__global__ void __launch_bounds__(256) k_chol_trsm(double* __restrict__ S, int ldS, int n1,
                                                   const double* __restrict__ Linv_all, const int32_t* __restrict__ strips,
                                                   double* __restrict__ xrow_tag, int n_tag) {
  __shared__ double Ai[16 * QP];
  __shared__ double Li[NB * QP];
  const int tid = threadIdx.x;
  if (xrow_tag && blockIdx.x == 0) tag_xrow(xrow_tag, n_tag);
  const int st = blockIdx.x >> 2, qi = (blockIdx.x & 3) * 16;
  const int kb = strips[2 * st + 1];
  const int k0 = kb * NB;
  const int r0 = strips[2 * st] * NB;  // tile row of a structurally non-zero strip of column kb
  const int rw = min(NB, n1 - r0);
  if (qi >= rw) return;
  const double* Lk = Linv_all + (size_t)kb * NB * NB;
  {
    double2 l[8];
#pragma unroll
    for (int i = 0; i < 8; i++) l[i] = *reinterpret_cast<const double2*>(Lk + (8 * i + (tid >> 5)) * NB + 2 * (tid & 31));
    const int r = tid >> 4, c = 4 * (tid & 15);
    const double* src = S + (size_t)(r0 + min(qi + r, rw - 1)) * ldS + k0 + c;
    const double2 a0 = *reinterpret_cast<const double2*>(src), a1 = *reinterpret_cast<const double2*>(src + 2);
    const bool in = qi + r < rw;
    Ai[r * QP + c] = in ? a0.x : 0.0; Ai[r * QP + c + 1] = in ? a0.y : 0.0;
    Ai[r * QP + c + 2] = in ? a1.x : 0.0; Ai[r * QP + c + 3] = in ? a1.y : 0.0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      Li[(8 * i + (tid >> 5)) * QP + 2 * (tid & 31)] = l[i].x;
      Li[(8 * i + (tid >> 5)) * QP + 2 * (tid & 31) + 1] = l[i].y;
    }
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int wj = wave * 16;
  const int lr = lane & 15, lk = lane >> 4;
  double4_t acc = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < NB; k += 4)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ai[lr * QP + k + lk], Li[(wj + lr) * QP + k + lk], acc, 0, 0, 0);
  const int kw = min(NB, n1 - k0);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = qi + lk + 4 * r, col = wj + lr;
    if (row < rw && col < kw) S[(size_t)(r0 + row) * ldS + k0 + col] = acc[r];
  }
}

// Trailing update A_ij -= sum_k A_ik A_jk^T for one 32x32 QUADRANT of a 64x64 tile (ti >= tj) and the columns k of the
// current level that reach it (contrib[c0..c1), ascending: fixed summation order).  grid = 4 x targets; 4 waves, each one
// 16x16 block of the quadrant (v_mfma_f64_16x16x4_f64: A operand: lane l holds A[l&15][l>>4]; B operand: B[l>>4][l&15];
// C/D: 4 doubles per lane, col = l&15, row = (l>>4) + 4*reg).  The half strips of contributor c+1 travel global ->
// registers while contributor c is on the matrix pipe.
__global__ void __launch_bounds__(256) k_chol_update(double* __restrict__ S, int ldS, int n1,
                                                     const int32_t* __restrict__ targets, const int32_t* __restrict__ contrib) {
  __shared__ double Ai[32 * QP];
  __shared__ double Aj[32 * QP];
  const int tg = blockIdx.x >> 2, qi = (blockIdx.x & 2) * 16, qj = (blockIdx.x & 1) * 32;
  const int ti = targets[4 * tg], tj = targets[4 * tg + 1];
  const int c0 = targets[4 * tg + 2], c1 = targets[4 * tg + 3];
  if (ti == tj && qj > qi) return;   // strictly upper quadrant of a diagonal tile
  const int i0 = ti * NB + qi, j0 = tj * NB + qj;
  const int iw = min(32, n1 - i0), jw = min(32, n1 - j0);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wi = (wave >> 1) * 16, wj = (wave & 1) * 16;
  double4_t acc = {0, 0, 0, 0};
  const int lr = lane & 15, lk = lane >> 4;
  // thread t owns elements (row 4k + t / 64, column t % 64), k = 0..7: every load instruction covers 4 full rows
  const int pr = tid >> 6, pc = tid & 63;
  // the target's own entries are fetched first: they are only needed at the very end, where the load used to be one more
  // dependent round trip behind the last product
  double tgt[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = wi + lk + 4 * r, col = wj + lr;
    tgt[r] = (row < iw && col < jw && (i0 + row) >= (j0 + col)) ? S[(size_t)(i0 + row) * ldS + j0 + col] : 0.0;
  }
  // Three contributors in flight: a contributor's half strips travel global -> registers while the two before it are
  // staged / on the matrix pipe.  With one in flight every contributor cost a full L2 round trip (~2.2 us against 0.43 us of
  // MFMA): the leaf level, where a separator tile collects up to ten columns, took 22 us.
  double ra0[8], rb0[8], ra1[8], rb1[8], ra2[8], rb2[8];
  auto fetch = [&](int c, double* ra, double* rb) {
    const int k0 = contrib[c] * NB;
    const double* ga = S + (size_t)(i0 + pr) * ldS + k0 + pc;
    const double* gb = S + (size_t)(j0 + pr) * ldS + k0 + pc;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      ra[k] = (4 * k + pr < iw) ? ga[(size_t)4 * k * ldS] : 0.0;
      rb[k] = (4 * k + pr < jw) ? gb[(size_t)4 * k * ldS] : 0.0;
    }
  };
  auto step = [&](int c, double* ra, double* rb) {
    if (c > c0) __syncthreads();          // the previous contributor's MFMAs have read Ai / Aj
#pragma unroll
    for (int k = 0; k < 8; k++) { Ai[(4 * k + pr) * QP + pc] = ra[k]; Aj[(4 * k + pr) * QP + pc] = rb[k]; }
    __syncthreads();
    if (c + 3 < c1) fetch(c + 3, ra, rb);
#pragma unroll
    for (int k = 0; k < NB; k += 4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ai[(wi + lr) * QP + k + lk], Aj[(wj + lr) * QP + k + lk], acc, 0, 0, 0);
  };
  fetch(c0, ra0, rb0);
  if (c0 + 1 < c1) fetch(c0 + 1, ra1, rb1);
  if (c0 + 2 < c1) fetch(c0 + 2, ra2, rb2);
  for (int c = c0; c < c1; c += 3) {
    step(c, ra0, rb0);
    if (c + 1 < c1) step(c + 1, ra1, rb1);
    if (c + 2 < c1) step(c + 2, ra2, rb2);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = wi + lk + 4 * r, col = wj + lr;
    if (row < iw && col < jw && (i0 + row) >= (j0 + col)) S[(size_t)(i0 + row) * ldS + j0 + col] = tgt[r] - acc[r];
  }
}

// k_chol_trsm and k_chol_update of one level in ONE launch.  Workgroups [0, n_trsm) are the panel solve's 16-row slices,
// the others the update's 32x32 quadrants; a quadrant waits for the slices of the strips it multiplies instead of a kernel
// boundary (4.8 + 5.5 us per level as two launches, each mostly boundary + load latency: ~1.7 us of boundary and one round
// trip less per level).  Hand-off (MI355X_MICROARCH.md, valid forms): the slice is written with 8-byte agent-scope
// (write-through) stores, drained (s_waitcnt vmcnt(0)) before its flag is raised to the solve's sequence number; the reader
// polls the flags with relaxed agent-scope loads on a few lanes and reads the strips with agent-scope loads (no L1, and the
// XCD's L2 cannot hold these lines yet: nothing in this launch reads a strip before its flag).  No deadlock: the launcher
// only uses this kernel when the whole grid is resident at once; a wait is bounded anyway and flags the trial as failed.
//
// kDiag: the level's k_chol_diag launch is folded in as well.  Every slice workgroup factors its column's diagonal tile ITSELF
// (chol_diag_tile in its own LDS: the 12 workgroups of a column with three strips do the same work and get the same bits) while
// its 16 strip rows are already on their way from memory, and multiplies by the L^-1 it finds in LDS: the level loses a kernel
// boundary and the L^-1 store -> load round trip (~2.5 us of ~19); the column's first slice workgroup stores L^-1 for the back
// substitution afterwards, off the critical path.  86 KB of LDS: one workgroup per CU, so the launcher uses it for levels of
// <= 256 workgroups (slices + quadrants, or the slices alone with the update as a second launch).
template <bool kDiag>
__global__ void __launch_bounds__(256) k_chol_trsm_update(double* __restrict__ S, int ldS, int n1, double* __restrict__ Linv_all,
                                                          const int32_t* __restrict__ strips, int strip_base, int n_trsm,
                                                          const int32_t* __restrict__ targets, const int32_t* __restrict__ contrib,
                                                          const int32_t* __restrict__ contrib_strip, int32_t* __restrict__ flags,
                                                          int gen, int gen_pub, int* __restrict__ fail, double* __restrict__ xrow_tag, int n_tag) {
  // trsm: Ai (16 rows) + Li (64 rows); update: Ai + Aj (32 rows each); kDiag: the layout of k_chol_diag (Ai over Pcol / Iv)
  __shared__ __attribute__((aligned(16))) double smem[kDiag ? 16 * NB + 4 * 16 * 17 + 16 * 16 + 2 * NB * LP : 16 * QP + NB * QP];
  const int tid = threadIdx.x;
  if (xrow_tag && blockIdx.x == 0) tag_xrow(xrow_tag, n_tag);
  if (kDiag && (int)blockIdx.x < n_trsm) {
    const int st = blockIdx.x >> 2, sl = blockIdx.x & 3, qi = sl * 16;
    const int kb = strips[2 * st + 1];
    const int k0 = kb * NB;
    const int r0 = strips[2 * st] * NB;
    const int rw = min(NB, n1 - r0);
    if (qi < rw) {
      lds_f64* const lds = (lds_f64*)smem;
      lds_f64 (*const Pcol)[NB] = (lds_f64 (*)[NB])lds;
      lds_f64 (*const Iv)[16][17] = (lds_f64 (*)[16][17])(lds + 16 * NB);
      lds_f64* const Id = lds + 16 * NB + 4 * 16 * 17;
      lds_f64* const Bm = Id + 16 * 16;
      lds_f64* const Li = Bm + NB * LP;
      const int kw = min(NB, n1 - k0);
      // this slice's 16 strip rows: in flight during the whole factorisation
      const int sr = tid >> 4, sc = 4 * (tid & 15);
      const double* src = S + (size_t)(r0 + min(qi + sr, rw - 1)) * ldS + k0 + sc;
      const double2 a0 = *reinterpret_cast<const double2*>(src), a1 = *reinterpret_cast<const double2*>(src + 2);
      {
        double2 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int r = 8 * i + (tid >> 5), c = 2 * (tid & 31);
          v[i] = *reinterpret_cast<const double2*>(S + (size_t)(k0 + min(r, kw - 1)) * ldS + k0 + c);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int r = 8 * i + (tid >> 5), c = 2 * (tid & 31);
          const bool in = r < kw;
          Bm[r * LP + c] = in ? v[i].x : (r == c ? 1.0 : 0.0);
          Bm[r * LP + c + 1] = in ? v[i].y : (r == c + 1 ? 1.0 : 0.0);
        }
      }
      Id[tid] = (tid >> 4) == (tid & 15) ? 1.0 : 0.0;
      __syncthreads();
      const DiagLds D = {Pcol, Iv, Id, Bm, Li};
      chol_diag_tile(D, fail);                    // (ends behind a barrier: Pcol / Iv are free)
      lds_f64* const Ai = lds;
      {
        const bool in = qi + sr < rw;
        Ai[sr * QP + sc] = in ? a0.x : 0.0; Ai[sr * QP + sc + 1] = in ? a0.y : 0.0;
        Ai[sr * QP + sc + 2] = in ? a1.x : 0.0; Ai[sr * QP + sc + 3] = in ? a1.y : 0.0;
      }
      __syncthreads();
      const int wave = tid >> 6, lane = tid & 63;
      const int wj = wave * 16;
      const int lr = lane & 15, lk = lane >> 4;
      double4_t acc = {0, 0, 0, 0};
      // X = A L^-T: column block `wave` of X needs the k-blocks 0..wave of L^-1 (the blocks above its diagonal were never written in
      // LDS; in the stored L^-1 they are zeros, so the products left out here are the exact zeros the two-launch form adds)
      for (int k = 0; k < 16 * (wave + 1); k += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ai[lr * QP + k + lk], Li[(wj + lr) * LP + k + lk], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = qi + lk + 4 * r, col = wj + lr;
        if (row < rw && col < kw) __hip_atomic_store(S + (size_t)(r0 + row) * ldS + k0 + col, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + 4 * (strip_base + st) + sl, gen_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // L^-1 for the back substitution: the first slice of the column's first strip (a level lists a column's strips together)
      if (sl == 0 && (st == 0 || strips[2 * (st - 1) + 1] != kb)) {
        double* Lo = Linv_all + (size_t)kb * NB * NB;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int r = 8 * k + (tid >> 5), c = 2 * (tid & 31);
          const bool lower = (c >> 4) <= (r >> 4);
          const double2 v = lower ? make_double2(Li[r * LP + c], Li[r * LP + c + 1]) : make_double2(0.0, 0.0);
          *reinterpret_cast<double2*>(Lo + r * NB + c) = v;
        }
      }
      return;
    }
    if (tid == 0) __hip_atomic_store(flags + 4 * (strip_base + st) + sl, gen_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if ((int)blockIdx.x < n_trsm) {
    double* Ai = smem;
    double* Li = smem + 16 * QP;
    const int st = blockIdx.x >> 2, sl = blockIdx.x & 3, qi = sl * 16;
    const int kb = strips[2 * st + 1];
    const int k0 = kb * NB;
    const int r0 = strips[2 * st] * NB;
    const int rw = min(NB, n1 - r0);
    if (qi < rw) {
      const double* Lk = Linv_all + (size_t)kb * NB * NB;
      {
        double2 l[8];
#pragma unroll
        for (int i = 0; i < 8; i++) l[i] = *reinterpret_cast<const double2*>(Lk + (8 * i + (tid >> 5)) * NB + 2 * (tid & 31));
        const int r = tid >> 4, c = 4 * (tid & 15);
        const double* src = S + (size_t)(r0 + min(qi + r, rw - 1)) * ldS + k0 + c;
        const double2 a0 = *reinterpret_cast<const double2*>(src), a1 = *reinterpret_cast<const double2*>(src + 2);
        const bool in = qi + r < rw;
        Ai[r * QP + c] = in ? a0.x : 0.0; Ai[r * QP + c + 1] = in ? a0.y : 0.0;
        Ai[r * QP + c + 2] = in ? a1.x : 0.0; Ai[r * QP + c + 3] = in ? a1.y : 0.0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          Li[(8 * i + (tid >> 5)) * QP + 2 * (tid & 31)] = l[i].x;
          Li[(8 * i + (tid >> 5)) * QP + 2 * (tid & 31) + 1] = l[i].y;
        }
      }
      __syncthreads();
      const int wave = tid >> 6, lane = tid & 63;
      const int wj = wave * 16;
      const int lr = lane & 15, lk = lane >> 4;
      double4_t acc = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < NB; k += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ai[lr * QP + k + lk], Li[(wj + lr) * QP + k + lk], acc, 0, 0, 0);
      const int kw = min(NB, n1 - k0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = qi + lk + 4 * r, col = wj + lr;
        if (row < rw && col < kw) __hip_atomic_store(S + (size_t)(r0 + row) * ldS + k0 + col, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's part of the slice has reached the coherence point
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + 4 * (strip_base + st) + sl, gen_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // gen_pub == gen (debug switch: see the launcher)
    return;
  }
  double* Ai = smem;
  double* Aj = smem + 32 * QP;
  const int bx = blockIdx.x - n_trsm;
  const int tg = bx >> 2, qi = (bx & 2) * 16, qj = (bx & 1) * 32;
  const int ti = targets[4 * tg], tj = targets[4 * tg + 1];
  const int c0 = targets[4 * tg + 2], c1 = targets[4 * tg + 3];
  if (ti == tj && qj > qi) return;   // strictly upper quadrant of a diagonal tile
  const int i0 = ti * NB + qi, j0 = tj * NB + qj;
  const int iw = min(32, n1 - i0), jw = min(32, n1 - j0);
  const int wave = tid >> 6, lane = tid & 63;
  const int wi = (wave >> 1) * 16, wj = (wave & 1) * 16;
  double4_t acc = {0, 0, 0, 0};
  const int lr = lane & 15, lk = lane >> 4;
  const int pr = tid >> 6, pc = tid & 63;
  double tgt[4];                       // the target's own entries: nothing in this launch writes them, fetched before the wait
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = wi + lk + 4 * r, col = wj + lr;
    tgt[r] = (row < iw && col < jw && (i0 + row) >= (j0 + col)) ? S[(size_t)(i0 + row) * ldS + j0 + col] : 0.0;
  }
  // the slices this quadrant multiplies: rows [qi, qi + 32) of strip (ti, k) and rows [qj, qj + 32) of strip (tj, k) for every
  // contributing column k -- four flags per contributor, one lane each
  for (int f = tid; f < 4 * (c1 - c0); f += 256) {
    const int c = c0 + (f >> 2), side = (f >> 1) & 1, half = f & 1;
    const int32_t* flag = flags + 4 * contrib_strip[2 * c + side] + ((side ? qj : qi) >> 4) + half;
    for (int spins = 0; __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen; spins++) {
      __builtin_amdgcn_s_sleep(1);
      if (spins > (1 << 18)) { *fail = 2; break; }          // never hang the device: give up, the trial is rejected
    }
  }
  __syncthreads();
  double ra0[8], rb0[8], ra1[8], rb1[8], ra2[8], rb2[8];
  auto fetch = [&](int c, double* ra, double* rb) {
    const int k0 = contrib[c] * NB;
    const double* ga = S + (size_t)(i0 + pr) * ldS + k0 + pc;
    const double* gb = S + (size_t)(j0 + pr) * ldS + k0 + pc;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      ra[k] = (4 * k + pr < iw) ? __hip_atomic_load(ga + (size_t)4 * k * ldS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      rb[k] = (4 * k + pr < jw) ? __hip_atomic_load(gb + (size_t)4 * k * ldS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
  };
  auto step = [&](int c, double* ra, double* rb) {
    if (c > c0) __syncthreads();          // the previous contributor's MFMAs have read Ai / Aj
#pragma unroll
    for (int k = 0; k < 8; k++) { Ai[(4 * k + pr) * QP + pc] = ra[k]; Aj[(4 * k + pr) * QP + pc] = rb[k]; }
    __syncthreads();
    if (c + 3 < c1) fetch(c + 3, ra, rb);
#pragma unroll
    for (int k = 0; k < NB; k += 4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ai[(wi + lr) * QP + k + lk], Aj[(wj + lr) * QP + k + lk], acc, 0, 0, 0);
  };
  fetch(c0, ra0, rb0);
  if (c0 + 1 < c1) fetch(c0 + 1, ra1, rb1);
  if (c0 + 2 < c1) fetch(c0 + 2, ra2, rb2);
  for (int c = c0; c < c1; c += 3) {
    step(c, ra0, rb0);
    if (c + 1 < c1) step(c + 1, ra1, rb1);
    if (c + 2 < c1) step(c + 2, ra2, rb2);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = wi + lk + 4 * r, col = wj + lr;
    if (row < iw && col < jw && (i0 + row) >= (j0 + col)) S[(size_t)(i0 + row) * ldS + j0 + col] = tgt[r] - acc[r];
  }
}

// Backward substitution L^T x = y (y = row n_pad of S) in PULL form, ALL tile columns in one launch:
//   x_k = Linv_kk^T (y_k - sum_{i in struct(k)} L(i,k)^T x_i);  the x_i belong to ancestors of k in the elimination tree.
// One workgroup per tile column.  A workgroup takes a ticket and processes the ticket-th column in root-first order, so
// every column it has to wait for is held by a workgroup that is already running (no deadlock whatever the dispatch
// order); x_i travels through `xrow` as 64 data-tagged words: the slots are filled with kXTag by the first solve launch of the
// factorisation, the producer overwrites them with 8-byte agent-scope (write-through) stores and every consumer lane waits
// for ITS word to change -- no flag, no drain in front of a flag (the back substitution of the BASELINE problem: 25.8 -> 20.7 us).
// Eight dependent launches of ~5 us each before (one per level) -> one launch with a ~2 us hop per level.
// The result goes to row space (xrow, for the descendants) and, compacted to dof doubles per unknown, to x.
__global__ void __launch_bounds__(256) k_chol_backsolve(const double* __restrict__ S, int ldS, int n_pad, int nfree, int per_tile, int dof,
                                                        const int32_t* __restrict__ cols, int ncols, const double* __restrict__ y,
                                                        double* __restrict__ xrow, double* __restrict__ x,
                                                        const double* __restrict__ Linv_all,
                                                        const int32_t* __restrict__ colstrip_off,
                                                        const int32_t* __restrict__ colstrips, int32_t* __restrict__ sync, int gen,
                                                        int* __restrict__ fail, int n_raw, int ncamt, int nkept, const int32_t* __restrict__ kept_list) {
  __shared__ double yk[NB];
  __shared__ double xi[NB];
  __shared__ double part[4][NB];
  __shared__ int s_col, s_ticket;
  const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == ncols - 1) __hip_atomic_store(sync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every ticket is taken
    s_col = cols[ncols - 1 - t];     // `cols` lists leaves first
    s_ticket = t;
  }
  __syncthreads();
  const int kb = s_col;
  const int k0 = kb * NB;
  if (k0 >= n_pad) return;                       // the rhs tile itself (never in the list; defensive)
  if (tid < NB) yk[tid] = y[k0 + tid];
  // Everything that does not depend on an ancestor's x is fetched BEFORE waiting for it: this column's Linv tile and the
  // strip tile of the first dependency (then always the next strip's tile while the current one is waited for / applied);
  // the hop per level shrinks from flag + three dependent tile loads to flag + 64 doubles of x.
  const double* Lk = Linv_all + (size_t)kb * NB * NB;
  double lk[16];
#pragma unroll
  for (int r = 0; r < 16; r++) lk[r] = Lk[(16 * q + r) * NB + c];
  if (s_ticket < n_raw) {
    // a column of the last level: nothing but the rhs row hangs below it, and the factorisation left that row unsolved (the
    // launcher skipped the level's k_chol_trsm: 5 us for one 64 x 64 matrix-vector product per column) -- y_k = Linv_kk r here
    __shared__ double Pm[NB][NB + 1];
    __syncthreads();                               // yk (= r) is in LDS
#pragma unroll
    for (int r = 0; r < 16; r++) Pm[16 * q + r][c] = lk[r] * yk[c];
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) s += Pm[c][16 * q + j];
    part[q][c] = s;
    __syncthreads();
    if (tid < NB) yk[tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    __syncthreads();
  }
  const int s_beg = colstrip_off[kb], s_end = colstrip_off[kb + 1];
  double tl[16];
  auto fetch_tile = [&](int s) {
    const int i0 = colstrips[s] * NB;
    if (i0 >= n_pad) return;
#pragma unroll
    for (int r = 0; r < 16; r++) tl[r] = S[(size_t)(i0 + 16 * q + r) * ldS + k0 + c];
  };
  // (round 5: the ancestors nearest the ROOT first -- their x is the first to arrive; in ascending order the column's parent, whose x
  //  arrives last, blocked every other product behind it: ~16 us a hop on a column with 40 strips.  A changed summation order.)
  if (s_beg < s_end) fetch_tile(s_end - 1);
  for (int s = s_end - 1; s >= s_beg; s--) {
    const int it = colstrips[s], i0 = it * NB;
    if (i0 >= n_pad) { if (s > s_beg) fetch_tile(s - 1); continue; }   // the rhs row is not an unknown
    if (tid < NB) {   // every lane waits for its own word of x_i: the data is its own flag
      double v = __hip_atomic_load(xrow + i0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int spins = 0; (unsigned long long)__double_as_longlong(v) == kXTag; spins++) {
        __builtin_amdgcn_s_sleep(1);
        if (spins > (1 << 22)) { *fail = 2; break; }     // never hang the device: give up, the trial is rejected
        v = __hip_atomic_load(xrow + i0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      xi[tid] = v;
    }
    __syncthreads();
    double u = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) u += tl[r] * xi[16 * q + r];
    if (s > s_beg) fetch_tile(s - 1);
    part[q][c] = u;
    __syncthreads();
    if (tid < NB) yk[tid] -= (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
  }
  __syncthreads();
  // (Linv^T y)[c] = sum_r Linv[r][c] y[r]  (Linv is zero above the diagonal); 4 row-quarters per column
  double s = 0;
#pragma unroll
  for (int r = 0; r < 16; r++) s += lk[r] * yk[16 * q + r];
  part[q][c] = s;
  __syncthreads();
  if (tid < NB) {   // one wave
    const double v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    __hip_atomic_store(xrow + k0 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // replaces the tag: the word is the hand-off
    // g2o's linear solver leaves _x untouched when the factorisation fails (linear_solver_eigen.h:89-112): x keeps the last
    // successful solve's values, which the LM loop then applies all the same (optimization_algorithm_levenberg.cpp:111-127)
    if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) write_x(x, kb, tid, v, nfree, per_tile, dof, ncamt, nkept, kept_list);
  }
}

// ----------------------------------------------------------------------------------------- flow
// The WHOLE reduced solve -- tile factorisation and back substitution -- as ONE persistent launch (round 5).  The level launches above pay a
// kernel boundary (or two) per elimination-tree level and run a level's trailing update to completion before the next diagonal tile
// starts; on a reduced system whose tree is a long chain (a loop-closed map: 37 levels at 500 keyframes) that is 43 us a level.  Here
// every structurally non-zero tile of a camera column is a TASK (BaTileSchedule::flow_tasks), taken by the workgroups through a ticket:
//   diagonal tile (k, k): gathers its updates LEFT-LOOKING -- tgt = S(k,k), then per level that updates it acc = sum over that level's
//     columns m of X(k,m) X(k,m)^T and tgt -= acc, the order and association of the level launches: same bits --, factorises
//     (chol_diag_tile) and publishes L_k^-1;
//   strip tile (i, k): gathers its updates the same way, waits for L_k^-1 and publishes X(i,k) = T L_k^-T in place;
//   then one task per camera column of the back substitution (k_chol_backsolve's body, root first).
// A task waits only for tasks BEFORE it in the list (a tile's contributors are columns of lower levels, its own diagonal tile is listed
// first, the back substitution comes last), so the workgroups that hold tickets always make progress: no residency assumption, no
// deadlock whatever else runs on the chip.  Hand-off: the payload is stored write-through (sc1: 16-byte buffer stores, or 8-byte
// agent-scope stores straight from the MFMA accumulators), drained by every storing wave, then ONE lane raises the tile's flag to the
// solve's sequence number; the consumer polls that word relaxed and reads the payload with sc1 loads (MI355X_MICROARCH.md, valid forms).
// What this buys: no kernel boundaries, and the trailing update of a column overlaps the factorisation of the next diagonal tile --
// the chain per level is [last contribution -> factorise -> L^-1 -> first strip], not [everything of the level].
#ifdef DVM_FLOW_DEBUG
// per task: [0] start, [1] gather done, [2] L^-1 there (strip) / tile in LDS (diagonal), [3] computed, [4] published, [5] wall-clock ticks
// thread 0 spent in flow_wait, [6] workgroup, [7] XCC id.  wall_clock64: 100 MHz.
__device__ long long g_flow_dbg[4096 * 8];
#define DVM_FSTMP(t, i) do { if (threadIdx.x == 0 && (t) < 4096) g_flow_dbg[(t) * 8 + (i)] = (long long)wall_clock64(); } while (0)
extern "C" int dvm_debug_flow_stamps(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_flow_dbg), sizeof(long long) * (size_t)n); }
#else
#define DVM_FSTMP(t, i) do { } while (0)
#endif
// k_chol_flow's hook into the factorisation (chol_diag_tile): the gathered chain strip T(p, k) into LDS under the last panel
struct FlowTHook {
  static constexpr bool on = true;
  const int32_t* flags;        // the two "half gathered" words of the chain strip (null: no chain parent)
  int gen;
  __amdgpu_buffer_rsrc_t rS;
  int toff, ldS;               // byte offset of the tile in S
  double* Xb;                  // LDS, pitch QP
  int* s_flag;                 // LDS word: 1 = the tile is (being) loaded
  int* fv;                     // the look's register (per thread)
  typedef unsigned int v4u32_ __attribute__((ext_vector_type(4)));
  // begin (waves 1..3, top of the last panel): wave 3 issues the loads of the strip's two flags
  __device__ __forceinline__ void begin() {
    *fv = gen;
    if (flags && threadIdx.x >= 192 && threadIdx.x < 194) *fv = __hip_atomic_load(flags + (threadIdx.x - 192), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // end (behind their own work there): wave 3 says whether both halves are gathered (LDS word: 2 = not decided yet, 0 / 1); the three
  // waves then move the tile, 2048 chunks of 16 bytes over 192 threads, global -> registers -> LDS
  __device__ __forceinline__ void end() {
    if (threadIdx.x >= 192) {
      const bool ok = __all(*fv == gen) && flags != nullptr;
      if (threadIdx.x == 192) *reinterpret_cast<volatile int*>(s_flag) = ok ? 1 : 0;
    }
    int f;
    while ((f = *reinterpret_cast<volatile int*>(s_flag)) == 2) __builtin_amdgcn_s_sleep(1);
    if (f == 0) return;
    const int q0 = (int)threadIdx.x - 64;
    v4u32_ v[11];
#pragma unroll
    for (int j = 0; j < 11; j++) {
      const int q = min(q0 + 192 * j, 2047);
      v[j] = __builtin_amdgcn_raw_buffer_load_b128(rS, toff + (int)(((size_t)(q >> 5) * ldS + 2 * (q & 31)) * 8), 0, 16);
    }
#pragma unroll
    for (int j = 0; j < 11; j++) {
      const int q = q0 + 192 * j;
      if (q < 2048) {
        double2 d;
        d.x = __hiloint2double((int)v[j].y, (int)v[j].x); d.y = __hiloint2double((int)v[j].w, (int)v[j].z);
        *reinterpret_cast<double2*>(Xb + (q >> 5) * 66 + 2 * (q & 31)) = d;
      }
    }
  }
};
struct FlowArgs {
  double* S; int ldS, n1, n_pad;
  double* Linv_all;
  const int32_t *tasks, *fc, *colinfo;
  int n_factor, n_back;
  int32_t* flags; int nstrips, ntiles;
  int gen, gen_pub; int* fail;            // gen_pub == gen (test switch DVM_BA_DEBUG_BREAK_FLOW: X flags nobody waits for)
  const int32_t* cols; double* xrow; double* x;
  const int32_t *colstrip_off, *colstrips, *colstrip_id;
  int nfree, per_tile, dof;
  int ncamt, nkept; const int32_t* kept_list;
};
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
// bounded wait of ONE lane for a word to reach the solve's sequence number; a timeout marks the trial (fail = 2: the host repeats it with
// the level launches) and lets everybody run out: once the mark is up nobody waits any more
__device__ __forceinline__ void flow_wait_raw(const int32_t* flag, int gen, int* __restrict__ fail) {
  for (int spins = 0; __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen; spins++) {
    __builtin_amdgcn_s_sleep(1);
    if ((spins & 255) == 255 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2) return;
    if (spins > (1 << 19)) { __hip_atomic_store(fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
  }
}
#ifdef DVM_FLOW_DEBUG
__device__ __forceinline__ void flow_wait_dbg(const int32_t* flag, int gen, int* __restrict__ fail, long long* wacc) {
  const long long w0 = (long long)wall_clock64();
  flow_wait_raw(flag, gen, fail);
  *wacc += (long long)wall_clock64() - w0;
}
#define flow_wait(f, g, x) flow_wait_dbg(f, g, x, &s_wacc)
#else
#define flow_wait(f, g, x) flow_wait_raw(f, g, x)
#endif
__device__ __forceinline__ double2 u4_to_d2(v4u32 v) {
  return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
}
__device__ __forceinline__ v4u32 d2_to_u4(double a, double b) {
  return v4u32{(unsigned)__double2loint(a), (unsigned)__double2hiint(a), (unsigned)__double2loint(b), (unsigned)__double2hiint(b)};
}
// One contributor's products for the NS blocks a wave owns: acc[s] += A_s B_s^T over the 64 columns of the strips in LDS.  Branch-free and
// unrolled by 16 columns with the operands of a chunk read before its MFMAs, so that the LDS latency of chunk n + 1 sits under the matrix
// pipe of chunk n (with a test per slot inside the loop the compiler read, waited and multiplied one block at a time: ~2.5x the pipe time).
template <int NS, bool SHARE_B>
__device__ __forceinline__ void flow_mma(const double* __restrict__ Ai, const double* __restrict__ Bp, const int (&arow)[3], const int (&brow)[3], int lr, int lk,
                                         double4_t (&acc)[3]) {
  const double* pa[NS];
  const double* pb[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) { pa[s] = Ai + (arow[s] + lr) * QP + lk; pb[s] = Bp + (brow[s] + lr) * QP + lk; }
#pragma unroll
  for (int kk = 0; kk < NB; kk += 16) {
    double a[NS][4], b[NS][4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int s = 0; s < NS; s++) {
        a[s][j] = pa[s][kk + 4 * j];
        if (!SHARE_B || s == 0) b[s][j] = pb[s][kk + 4 * j];
      }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int s = 0; s < NS; s++) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][j], b[SHARE_B ? 0 : s][j], acc[s], 0, 0, 0);
  }
}
__global__ void __launch_bounds__(256) k_chol_flow(FlowArgs A) {
  constexpr int kCholLds = 16 * NB + 4 * 16 * 17 + 16 * 16 + 2 * NB * LP;      // the factorisation's block (also the gather buffers Ai / Aj)
  __shared__ __attribute__((aligned(16))) double smem[kCholLds + NB * QP];     // + the chain strip T -> X (119 KB: one workgroup per CU either way)
  __shared__ int s_ticket, s_ready[2];
#ifdef DVM_FLOW_DEBUG
  __shared__ long long s_wacc;
#endif
  static_assert(2 * NB * QP <= 16 * NB + 4 * 16 * 17 + 16 * 16 + 2 * NB * LP, "the gather buffers live inside the factorisation's LDS block");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int ldS = A.ldS;
  const int F_L = 2 * A.nstrips, F_P = F_L + A.ntiles, F_T = F_P + A.ntiles;      // flag spaces: X halves | L^-1 | PRE | gathered halves of chain strips
  int32_t* const tagged = A.flags + F_T + 2 * A.nstrips;
  int32_t* const ticket = tagged + 1;
  const int ntotal = A.n_factor + A.n_back;
  // buffer descriptors (wave-uniform inputs only): S and the L^-1 tiles, for the 16-byte sc1 loads / stores
  const auto rS = __builtin_amdgcn_make_buffer_rsrc(A.S, 0, (int)min((size_t)ldS * ldS * sizeof(double), (size_t)0x7FFFFFF0), 0x00020000);
  const auto rL = __builtin_amdgcn_make_buffer_rsrc(A.Linv_all, 0, (int)((size_t)A.ntiles * NB * NB * sizeof(double)), 0x00020000);
  double* const Ai = smem;
  double* const Aj = smem + NB * QP;
  const int crow = tid >> 5, ccol = 2 * (tid & 31);      // 16-byte chunk of a 64-double row: instruction i covers rows 8 i + crow
  for (;;) {
    __syncthreads();                       // the previous task is through with the LDS block and s_ticket
    if (tid == 0) {
      const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == ntotal + (int)gridDim.x - 1) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the last draw of the launch: every workgroup draws one ticket beyond the list
      s_ticket = t;
    }
    __syncthreads();
    const int t = s_ticket;
    if (t >= ntotal) return;
#ifdef DVM_FLOW_DEBUG
    if (tid == 0) s_wacc = 0;
    if (tid == 0 && t < 4096) g_flow_dbg[t * 8 + 7] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;   // (XCC_ID register: best effort)
#endif
    DVM_FSTMP(t, 0);
    if (t == 0) {
      // the back substitution's hand-off slots: tags (write-through), published before anybody may poll them
      for (int i = tid; i < A.n_pad; i += 256) __hip_atomic_store(reinterpret_cast<unsigned long long*>(A.xrow) + i, kXTag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(tagged, A.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t < A.n_factor) {
      // ------------------------------------------------------------------------------------------ a tile (or half tile) of the factor
      const int32_t* T = A.tasks + 8 * (size_t)t;
      const int kind = T[0], ti = T[1], tj = T[2], half = T[3], c0 = T[4], c1 = T[5], self = T[6];
      const bool slice = kind == 1 || kind == 4;
      const int i0 = ti * NB + (slice ? 32 * half : 0), j0 = tj * NB;
      const int rw = min(slice ? 32 : NB, A.n1 - i0);      // rows of the region that exist (the rhs row: one)
#ifdef DVM_FLOW_DEBUG
      if (tid == 0 && t < 4096) g_flow_dbg[t * 8 + 6] = (long long)blockIdx.x | ((long long)kind << 12) | ((long long)ti << 16) | ((long long)tj << 32) | ((long long)(c1 - c0) << 48);
#endif
      // the 16x16 blocks this wave owns (slot s): a slice -- column block `wave` of its two block rows; a diagonal tile -- its ten lower
      // blocks dealt 3 3 2 2.  arow / brow: first row of the block's A / B operand in the LDS buffers (= the block's position in the region)
      int arow[3], brow[3];
      bool son[3];
#pragma unroll
      for (int s = 0; s < 3; s++) {
        if (slice) { arow[s] = 16 * s; brow[s] = 16 * wave; son[s] = s < 2 && 16 * s < rw; }
        else {
          const int b = wave < 2 ? 3 * wave + s : 6 + 2 * (wave - 2) + s;          // index into the row-major list of lower blocks
          son[s] = wave < 2 || s < 2;
          const int bi = b < 1 ? 0 : b < 3 ? 1 : b < 6 ? 2 : 3;
          arow[s] = 16 * bi; brow[s] = 16 * (b - bi * (bi + 1) / 2);
        }
      }
      double4_t tgt[3], acc[3];
#pragma unroll
      for (int s = 0; s < 3; s++) {
        acc[s] = double4_t{0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = arow[s] + lk + 4 * r, col = brow[s] + lr;
          const double* p = A.S + (size_t)(i0 + min(row, rw - 1)) * ldS + j0 + col;
          tgt[s][r] = (son[s] && row < rw) ? *p : 0.0;
        }
      }
      // ---- gather: three contributors in flight (global -> registers), one in LDS under the matrix pipe.  A contributor's three flags
      // (its A half, both halves of B) are looked at by three lanes at once, and the look for contributor c + 3 is issued before step
      // c's products and read behind them: a poll is a ~1 us trip to the coherence point, which three in a row on one lane in front of
      // a barrier put on every step (4.4 us a contributor measured, 1.3 us of it matrix pipe).
      v4u32 g0[12], g1[12], g2[12];
      bool l0 = false, l1 = false, l2 = false;
      int rc = 0;
      auto flag_of = [&](int c, int f) -> const int32_t* {      // f = 0: the A operand's half; 1, 2: the halves of B
        const int32_t* e = A.fc + 4 * (size_t)c;
        return f == 0 ? A.flags + 2 * e[1] + (slice ? half : 0) : A.flags + 2 * e[2] + (f - 1);
      };
      auto wait_block = [&](int c) {
        if (tid < 3) flow_wait(flag_of(c, tid), A.gen, A.fail);
        __syncthreads();
      };
      auto fetch = [&](int c, v4u32 (&g)[12]) {
        const int m0 = A.fc[4 * (size_t)c] * NB;
        if (slice) {
#pragma unroll
          for (int i = 0; i < 4; i++) g[i] = __builtin_amdgcn_raw_buffer_load_b128(rS, (int)(((size_t)(i0 + min(8 * i + crow, rw - 1)) * ldS + m0 + ccol) * 8), 0, 16);
#pragma unroll
          for (int i = 0; i < 8; i++) g[4 + i] = __builtin_amdgcn_raw_buffer_load_b128(rS, (int)(((size_t)(j0 + 8 * i + crow) * ldS + m0 + ccol) * 8), 0, 16);
        } else {
#pragma unroll
          for (int i = 0; i < 8; i++) g[i] = __builtin_amdgcn_raw_buffer_load_b128(rS, (int)(((size_t)(i0 + 8 * i + crow) * ldS + m0 + ccol) * 8), 0, 16);
        }
      };
      // chk: the step before this one looked at the flags of contributor c + 2 (result in s_ready): its loads go into that step's registers (gp)
      auto step = [&](int c, v4u32 (&g)[12], bool& l, v4u32 (&gp)[12], bool& lp, bool chk) {
        if (!l) { wait_block(c); fetch(c, g); }
        __syncthreads();                   // the product before this one has read Ai / Aj (and s_ready is written)
        if (chk) { lp = s_ready[(rc - 1) & 1] != 0; if (lp) fetch(c + 2, gp); }
        if (slice) {
#pragma unroll
          for (int i = 0; i < 4; i++) *reinterpret_cast<double2*>(Ai + (8 * i + crow) * QP + ccol) = 8 * i + crow < rw ? u4_to_d2(g[i]) : make_double2(0.0, 0.0);
#pragma unroll
          for (int i = 0; i < 8; i++) *reinterpret_cast<double2*>(Aj + (8 * i + crow) * QP + ccol) = u4_to_d2(g[4 + i]);
        } else {
#pragma unroll
          for (int i = 0; i < 8; i++) *reinterpret_cast<double2*>(Ai + (8 * i + crow) * QP + ccol) = u4_to_d2(g[i]);
        }
        const bool look = c + 3 < c1;
        int fv = A.gen;
        if (look && tid < 3) fv = __hip_atomic_load(flag_of(c + 3, tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        // (wave-uniform choice of the straight-line form: a slice owns two blocks with one B operand -- the rhs row: one --, a diagonal
        //  tile three or two blocks a wave)
        if (slice) { if (son[1]) flow_mma<2, true>(Ai, Aj, arow, brow, lr, lk, acc); else flow_mma<1, true>(Ai, Aj, arow, brow, lr, lk, acc); }
        else if (son[2]) flow_mma<3, false>(Ai, Ai, arow, brow, lr, lk, acc);
        else flow_mma<2, false>(Ai, Ai, arow, brow, lr, lk, acc);
        if (look) {
          if (wave == 0) { const bool ok = __all(fv == A.gen); if (tid == 0) s_ready[rc & 1] = ok ? 1 : 0; }
          rc++;
        }
        if (A.fc[4 * (size_t)c + 3]) {     // the level's last contributor: the level launch subtracts its sum here
#pragma unroll
          for (int s = 0; s < 3; s++) { tgt[s] -= acc[s]; acc[s] = double4_t{0, 0, 0, 0}; }
        }
      };
      {
        // the first three contributors: nine flags, nine lanes, one look
        int fv = A.gen;
        if (tid < 9 && c0 + tid / 3 < c1) fv = __hip_atomic_load(flag_of(c0 + tid / 3, tid % 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave == 0) {
          const unsigned long long bad = __ballot(fv != A.gen);
          if (tid == 0) { s_ready[0] = (int)(bad & 0x1FF); }
        }
        __syncthreads();
        const int bad = s_ready[0];
        __syncthreads();
        l0 = c0 < c1 && (bad & 7) == 0; l1 = c0 + 1 < c1 && (bad & 0x38) == 0; l2 = c0 + 2 < c1 && (bad & 0x1C0) == 0;
        if (l0) fetch(c0, g0);
        if (l1) fetch(c0 + 1, g1);
        if (l2) fetch(c0 + 2, g2);
      }
      {
        bool chk = false;
        for (int c = c0; c < c1; c += 3) {
          step(c, g0, l0, g2, l2, chk); chk = c + 3 < c1;
          if (c + 1 < c1) { step(c + 1, g1, l1, g0, l0, chk); chk = c + 4 < c1; }
          if (c + 2 < c1) { step(c + 2, g2, l2, g1, l1, chk); chk = c + 5 < c1; }
        }
      }
      if (c0 < c1) __syncthreads();        // everybody is through with Ai / Aj
      DVM_FSTMP(t, 1);
      if (kind == 4) {
        // ---- a half of a chain strip: T in place (the chain's workgroup solves it), flag
#pragma unroll
        for (int s = 0; s < 2; s++)
          if (son[s]) {
#pragma unroll
            for (int r = 0; r < 4; r++) Ai[(arow[s] + lk + 4 * r) * QP + brow[s] + lr] = tgt[s][r];
          }
        __syncthreads();
        DVM_FSTMP(t, 2);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int r = 8 * i + crow;
          if (r < rw) __builtin_amdgcn_raw_buffer_store_b128(d2_to_u4(Ai[r * QP + ccol], Ai[r * QP + ccol + 1]), rS, (int)(((size_t)(i0 + r) * ldS + j0 + ccol) * 8), 0, 16);
        }
        DVM_FSTMP(t, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(A.flags + F_T + 2 * self + half, A.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (slice) {
        // ---- X = T L^-T (block (bi, bj) needs the k-blocks 0..bj of L^-1: it is lower triangular), published in place, flag
#pragma unroll
        for (int s = 0; s < 2; s++)
          if (son[s]) {
#pragma unroll
            for (int r = 0; r < 4; r++) Ai[(arow[s] + lk + 4 * r) * QP + brow[s] + lr] = tgt[s][r];
          }
        if (tid == 0) flow_wait(A.flags + F_L + tj, A.gen, A.fail);
        __syncthreads();
        DVM_FSTMP(t, 2);
#pragma unroll
        for (int i = 0; i < 8; i++) g0[i] = __builtin_amdgcn_raw_buffer_load_b128(rL, (int)(((size_t)tj * NB * NB + (8 * i + crow) * NB + ccol) * 8), 0, 16);
#pragma unroll
        for (int i = 0; i < 8; i++) *reinterpret_cast<double2*>(Aj + (8 * i + crow) * QP + ccol) = u4_to_d2(g0[i]);
        __syncthreads();
        // the eight blocks cost 4 (bj + 1) MFMAs each: dealt so that every wave issues 20 -- waves 0, 1: column blocks 3 and 0 of block
        // row `wave`; waves 2, 3: column blocks 2 and 1 of block row `wave - 2`
        const int xbi = wave & 1, xbj0 = wave < 2 ? 3 : 2, xbj1 = wave < 2 ? 0 : 1;
        double4_t x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
        if (16 * xbi < rw) {
          for (int kk = 0; kk < 16 * (xbj0 + 1); kk += 4)
            x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ai[(16 * xbi + lr) * QP + kk + lk], Aj[(16 * xbj0 + lr) * QP + kk + lk], x0, 0, 0, 0);
          for (int kk = 0; kk < 16 * (xbj1 + 1); kk += 4)
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ai[(16 * xbi + lr) * QP + kk + lk], Aj[(16 * xbj1 + lr) * QP + kk + lk], x1, 0, 0, 0);
        }
        __syncthreads();                   // T has been read: X takes its place (staging for the 16-byte stores)
        if (16 * xbi < rw) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            Ai[(16 * xbi + lk + 4 * r) * QP + 16 * xbj0 + lr] = x0[r];
            Ai[(16 * xbi + lk + 4 * r) * QP + 16 * xbj1 + lr] = x1[r];
          }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int r = 8 * i + crow;
          if (r < rw) __builtin_amdgcn_raw_buffer_store_b128(d2_to_u4(Ai[r * QP + ccol], Ai[r * QP + ccol + 1]), rS, (int)(((size_t)(i0 + r) * ldS + j0 + ccol) * 8), 0, 16);
        }
        DVM_FSTMP(t, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains before the flag
        __syncthreads();
        if (tid == 0) __hip_atomic_store(A.flags + 2 * self + half, A.gen_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (kind == 2) {
        // ---- PRE: the sum of the levels before the last one back in place (lower blocks), flag
#pragma unroll
        for (int s = 0; s < 3; s++)
          if (son[s]) {
#pragma unroll
            for (int r = 0; r < 4; r++) Ai[(arow[s] + lk + 4 * r) * QP + brow[s] + lr] = tgt[s][r];
          }
        __syncthreads();
        DVM_FSTMP(t, 2);
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int r = 8 * i + crow;
          if ((ccol >> 4) <= (r >> 4)) __builtin_amdgcn_raw_buffer_store_b128(d2_to_u4(Ai[r * QP + ccol], Ai[r * QP + ccol + 1]), rS, (int)(((size_t)(i0 + r) * ldS + j0 + ccol) * 8), 0, 16);
        }
        DVM_FSTMP(t, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(A.flags + F_P + tj, A.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        // ---- DIAG: a leaf's diagonal tile (no contributor: tgt = S), and then UP THE TREE along the chain: factorise k; solve the chain
        // strip (p, k) with the L_k^-1 that is in LDS; publish L_k^-1 and X(p, k); take what PRE left of (p, p), add X X^T -- the last
        // contributor of p's last level --, subtract; k = p.  (The factor of a diagonal tile itself is not written back, as in k_chol_diag.)
        lds_f64* const lds = (lds_f64*)smem;
        lds_f64 (*const Pcol)[NB] = (lds_f64 (*)[NB])lds;
        lds_f64 (*const Iv)[16][17] = (lds_f64 (*)[16][17])(lds + 16 * NB);
        lds_f64* const Id = lds + 16 * NB + 4 * 16 * 17;
        lds_f64* const Bm = Id + 16 * 16;
        lds_f64* const Li = Bm + NB * LP;
        double* const Xb = smem + kCholLds;   // T -> X of the chain strip, pitch QP
        int k = tj;
        for (;;) {
          const int p = A.colinfo[8 * k], cstrip = A.colinfo[8 * k + 1];
#pragma unroll
          for (int s = 0; s < 3; s++)
            if (son[s]) {
#pragma unroll
              for (int r = 0; r < 4; r++) Bm[(arow[s] + lk + 4 * r) * LP + brow[s] + lr] = tgt[s][r];
            }
          Id[tid] = (tid >> 4) == (tid & 15) ? 1.0 : 0.0;
          if (tid == 0) s_ready[0] = 2;     // (the hook's word: not decided yet)
          __syncthreads();
          v4u32 tq[8];
          const int toff = p >= 0 ? (int)(((size_t)p * NB * ldS + (size_t)k * NB) * 8) : 0;
          DVM_FSTMP(3500 + k, 0);
          // the chain strip T(p, k): its two halves are looked at from INSIDE the factorisation (wave 3, beside the last panel), and if
          // they are there the tile travels into Xb under that panel and the tail
          int hook_fv = 0;
          FlowTHook hook;
          hook.flags = p >= 0 ? A.flags + F_T + 2 * cstrip : nullptr; hook.gen = A.gen; hook.rS = rS; hook.toff = toff; hook.ldS = ldS;
          hook.Xb = Xb; hook.s_flag = &s_ready[0]; hook.fv = &hook_fv;
          const DiagLds D = {Pcol, Iv, Id, Bm, Li};
          chol_diag_tile(D, A.fail, hook);
          const bool tpre = s_ready[0] == 1;
#ifdef DVM_FLOW_DEBUG
          if (tid == 0) g_flow_dbg[(3500 + k) * 8 + 6] = tpre ? 1 : 0;
#endif
          DVM_FSTMP(3500 + k, 1);
          const bool fold_root = p < 0 && A.colinfo[8 * k + 5] != 0;
          if (!fold_root) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
              const int r = 8 * q + crow;
              const bool lower = (ccol >> 4) <= (r >> 4);      // the blocks above the diagonal were never written in LDS: zeros from here
              __builtin_amdgcn_raw_buffer_store_b128(d2_to_u4(lower ? Li[r * LP + ccol] : 0.0, lower ? Li[r * LP + ccol + 1] : 0.0), rL, (int)(((size_t)k * NB * NB + r * NB + ccol) * 8), 0, 16);
            }
          }
          if (fold_root) {
            // a ROOT column (only the rhs row hangs below it): nobody else needs its factor.  y = L^-1 r and x = L^-T y right here, from
            // the L^-1 in LDS -- the sums of k_chol_backsolve's n_raw branch and of its last step, term for term (the blocks of L^-1 above
            // the diagonal, never written in LDS, are the zeros the stored tile has)
            double* const yk = Xb;
            double* const part = Xb + NB;
            if (tid == 0) flow_wait(A.flags + F_T + 2 * cstrip, A.gen, A.fail);
            __syncthreads();
            const int c = tid & 63, q = tid >> 6, k0 = k * NB;
            if (tid < NB) yk[tid] = __hip_atomic_load(A.S + (size_t)A.n_pad * ldS + k0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            double sy = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) sy += (q <= (c >> 4) ? Li[c * LP + 16 * q + j] : 0.0) * yk[16 * q + j];
            part[q * NB + c] = sy;
            __syncthreads();
            if (tid < NB) yk[tid] = (part[tid] + part[NB + tid]) + (part[2 * NB + tid] + part[3 * NB + tid]);
            __syncthreads();
            double sx = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) sx += ((c >> 4) <= q ? Li[(16 * q + r) * LP + c] : 0.0) * yk[16 * q + r];
            __syncthreads();
            part[q * NB + c] = sx;
            __syncthreads();
            if (tid < NB) {
              const double v = (part[tid] + part[NB + tid]) + (part[2 * NB + tid] + part[3 * NB + tid]);
              __hip_atomic_store(A.xrow + k0 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (__hip_atomic_load(A.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) write_x(A.x, k, tid, v, A.nfree, A.per_tile, A.dof, A.ncamt, A.nkept, A.kept_list);
            }
            DVM_FSTMP(3500 + k, 4);
            break;
          }
          // L_k^-1 is what this column's other strips wait for: out at once (drain, flag), whatever the chain waits for next
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (tid == 0) __hip_atomic_store(A.flags + F_L + k, A.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (p < 0) break;                // the chain ends here (the root, or a column that is not its parent's last contributor)
          if (!tpre) {
            if (tid < 2) flow_wait(A.flags + F_T + 2 * cstrip + tid, A.gen, A.fail);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; i++) tq[i] = __builtin_amdgcn_raw_buffer_load_b128(rS, toff + (int)(((size_t)(8 * i + crow) * ldS + ccol) * 8), 0, 16);
#pragma unroll
            for (int i = 0; i < 8; i++) *reinterpret_cast<double2*>(Xb + (8 * i + crow) * QP + ccol) = u4_to_d2(tq[i]);
          }
          const bool cont = A.colinfo[8 * k + 6] != 0;          // k is p's chain child: this workgroup goes on to p
          const int pmode = A.colinfo[8 * p + 2], lc0 = A.colinfo[8 * p + 3], lc1 = A.colinfo[8 * p + 4];
          if (cont && pmode >= 2 && tid == 0) flow_wait(A.flags + F_P + p, A.gen, A.fail);      // (long there: PRE runs levels ahead)
          __syncthreads();
          // what PRE left of (p, p): on its way while the strip is solved
          const int pi0 = p * NB;
#pragma unroll
          for (int s = 0; s < 3; s++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int row = arow[s] + lk + 4 * r, col = brow[s] + lr;
              const double* q = A.S + (size_t)(pi0 + row) * ldS + pi0 + col;
              tgt[s][r] = (!son[s] || !cont) ? 0.0 : pmode >= 2 ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
              acc[s][r] = 0.0;
            }
          }
          DVM_FSTMP(3500 + k, 2);
          // X = T L_k^-T: wave w takes block row w (4 + 8 + 12 + 16 MFMAs: L^-1 is lower triangular)
          double4_t xr[4];
#pragma unroll
          for (int bj = 0; bj < 4; bj++) {
            double4_t x4 = {0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < 16 * (bj + 1); kk += 4)
              x4 = __builtin_amdgcn_mfma_f64_16x16x4f64(Xb[(16 * wave + lr) * QP + kk + lk], Li[(16 * bj + lr) * LP + kk + lk], x4, 0, 0, 0);
            xr[bj] = x4;
          }
          __syncthreads();                 // T has been read: X takes its place
#pragma unroll
          for (int bj = 0; bj < 4; bj++)
#pragma unroll
            for (int r = 0; r < 4; r++) Xb[(16 * wave + lk + 4 * r) * QP + 16 * bj + lr] = xr[bj][r];
          __syncthreads();
#pragma unroll
          for (int i = 0; i < 8; i++) {
            const int r = 8 * i + crow;
            __builtin_amdgcn_raw_buffer_store_b128(d2_to_u4(Xb[r * QP + ccol], Xb[r * QP + ccol + 1]), rS, toff + (int)(((size_t)r * ldS + ccol) * 8), 0, 16);
          }
          DVM_FSTMP(3500 + k, 3);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // X has left (every storing wave): its flags go up before anything else --
          __syncthreads();                                     // the next chain strip's gather is waiting for them
          if (tid < 2) __hip_atomic_store(A.flags + 2 * cstrip + tid, A.gen_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!cont) break;     // not the parent's chain child: the parent belongs to a sibling's workgroup, which gathers this strip
          // p's other children (the contributors of its last level before the chain child, ascending): their strips (p, m) come from
          // memory like any gathered contributor, one at a time -- there are one or two
          for (int c = lc0; c < lc1; c++) {
            const int32_t* e = A.fc + 4 * (size_t)c;
            if (tid < 2) flow_wait(A.flags + 2 * e[1] + tid, A.gen, A.fail);
            __syncthreads();               // (also: the product before this one has read Ai)
#pragma unroll
            for (int i = 0; i < 8; i++) tq[i] = __builtin_amdgcn_raw_buffer_load_b128(rS, (int)(((size_t)(pi0 + 8 * i + crow) * ldS + (size_t)e[0] * NB + ccol) * 8), 0, 16);
#pragma unroll
            for (int i = 0; i < 8; i++) *reinterpret_cast<double2*>(Ai + (8 * i + crow) * QP + ccol) = u4_to_d2(tq[i]);
            __syncthreads();
            if (son[2]) flow_mma<3, false>(Ai, Ai, arow, brow, lr, lk, acc);
            else flow_mma<2, false>(Ai, Ai, arow, brow, lr, lk, acc);
          }
          // the chain child's product, the last of p's last level: acc += X X^T, tgt -= acc
          if (son[2]) flow_mma<3, false>(Xb, Xb, arow, brow, lr, lk, acc);
          else flow_mma<2, false>(Xb, Xb, arow, brow, lr, lk, acc);
#pragma unroll
          for (int s = 0; s < 3; s++) { tgt[s] -= acc[s]; acc[s] = double4_t{0, 0, 0, 0}; }
          __syncthreads();                 // Xb / Ai have been read: Bm may be written
          DVM_FSTMP(3500 + k, 4);
          k = p;
        }
      }
      DVM_FSTMP(t, 4);
#ifdef DVM_FLOW_DEBUG
      if (tid == 0 && t < 4096) g_flow_dbg[t * 8 + 5] = s_wacc;
#endif
      continue;
    }
    // -------------------------------------------------------------------------------------------- back substitution of one column
    {
      double* const yk = smem;                 // [NB]
      double* const xi = smem + NB;            // [NB]
      double (*const part)[NB] = (double (*)[NB])(smem + 2 * NB);   // [4][NB]
      const int c = tid & 63, q = tid >> 6;
      const int kb = A.cols[A.n_back - 1 - (t - A.n_factor)];      // `cols` lists leaves first
      const int k0 = kb * NB;
      if (A.colinfo[8 * kb + 5]) continue;      // a root column: solved by the chain's workgroup the moment it was factored
      const int s_beg = A.colstrip_off[kb], s_end = A.colstrip_off[kb + 1];
#ifdef DVM_FLOW_DEBUG
      if (tid == 0 && t < 4096) g_flow_dbg[t * 8 + 6] = (long long)blockIdx.x | (3ll << 12) | ((long long)kb << 16);
#endif
      // one lane per flag: the column's L^-1, the tags, and both halves of each of its strips (the rhs row: one half)
      if (tid == 0) flow_wait(tagged, A.gen, A.fail);
      if (tid == 1) flow_wait(A.flags + F_L + kb, A.gen, A.fail);
      for (int f = tid; f < 2 * (s_end - s_beg); f += 256) {
        const int s = s_beg + (f >> 1);
        if ((f & 1) == 0 || A.colstrips[s] * NB < A.n_pad) flow_wait(A.flags + 2 * A.colstrip_id[s] + (f & 1), A.gen, A.fail);
      }
      __syncthreads();
      DVM_FSTMP(t, 1);
      auto ld = [](const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
      if (tid < NB) yk[tid] = ld(A.S + (size_t)A.n_pad * ldS + k0 + tid);
      const double* Lk = A.Linv_all + (size_t)kb * NB * NB;
      double lkk[16];
#pragma unroll
      for (int r = 0; r < 16; r++) lkk[r] = ld(Lk + (16 * q + r) * NB + c);
      double tl[16];
      auto fetch_tile = [&](int s) {
        const int r0 = A.colstrips[s] * NB;
        if (r0 >= A.n_pad) return;
#pragma unroll
        for (int r = 0; r < 16; r++) tl[r] = ld(A.S + (size_t)(r0 + 16 * q + r) * ldS + k0 + c);
      };
      // ancestors nearest the root first: their x is the first to arrive, the parent's the last (see k_chol_backsolve)
      if (s_beg < s_end) fetch_tile(s_end - 1);
      for (int s = s_end - 1; s >= s_beg; s--) {
        const int r0 = A.colstrips[s] * NB;
        if (r0 >= A.n_pad) { if (s > s_beg) fetch_tile(s - 1); continue; }   // the rhs row is not an unknown
        if (tid < NB) {   // every lane waits for its own word of x_i: the data is its own flag
          double v = ld(A.xrow + r0 + tid);
          for (int spins = 0; (unsigned long long)__double_as_longlong(v) == kXTag; spins++) {
            __builtin_amdgcn_s_sleep(1);
            if (spins > (1 << 19) || ((spins & 255) == 255 && __hip_atomic_load(A.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) {
              __hip_atomic_store(A.fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
            v = ld(A.xrow + r0 + tid);
          }
          xi[tid] = v;
        }
        __syncthreads();
        double u = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) u += tl[r] * xi[16 * q + r];
        if (s > s_beg) fetch_tile(s - 1);
        part[q][c] = u;
        __syncthreads();
        if (tid < NB) yk[tid] -= (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
      }
      __syncthreads();
      double sum = 0;
#pragma unroll
      for (int r = 0; r < 16; r++) sum += lkk[r] * yk[16 * q + r];
      part[q][c] = sum;
      __syncthreads();
      if (tid < NB) {
        const double v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
        __hip_atomic_store(A.xrow + k0 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // replaces the tag: the word is the hand-off
        // (g2o's linear solver leaves _x untouched when the factorisation fails: see k_chol_backsolve)
        if (__hip_atomic_load(A.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) write_x(A.x, kb, tid, v, A.nfree, A.per_tile, A.dof, A.ncamt, A.nkept, A.kept_list);
      }
      DVM_FSTMP(t, 4);
    }
  }
}

// ----------------------------------------------------------------------------------------- K11
// xl = Dinv (bl - sum_k W_k^T xp[pose(k)])   (block_solver.hpp:459-483); thread per landmark.
// Eight lanes per landmark (a landmark of the BASELINE problem has eight observations): lane j takes edges j, j + 8, ...;
// the partial sums meet by three DPP steps inside the group of eight (quad swaps, then the half-row mirror) -- a fixed
// order.  A thread per landmark walked its edges one dependent gather after the other, on 79 workgroups (19 us).
__device__ __forceinline__ double dpp_xor_add(double v, int ctrl_is) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if (ctrl_is == 0) { lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false); }        // quad_perm [1,0,3,2]
  else if (ctrl_is == 1) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, false); }   // quad_perm [2,3,0,1]
  else { lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, false); }                  // row_half_mirror
  return v + __hiloint2double(hi, lo);
}
// Landmark back substitution, oplus of cameras and landmarks into the TRIAL state and the terms of computeScale =
// sum x (lambda x + b) in ONE launch (oplus: se3quat.h:212-240, types_six_dof_expmap.h:71-74): workgroups [0, nb_pose)
// update the cameras (thread per camera), the rest the landmarks; every workgroup feeds the grid-wide reduction.
__global__ void __launch_bounds__(256) k_point_backsub(BaView V, BaPublish pub, int nb_pose, const int* __restrict__ fail) {
  const double lambda = ba_lambda(V);
  double sc = 0;
  const bool failed = fail && *fail != 0;   // the reduced solve failed: x (cameras: untouched by the back substitution) and xl keep the last good values
  if ((int)blockIdx.x < nb_pose) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < V.nfree) {
      const int p = V.free_pose[i];
      const double* u = V.x + 6 * (size_t)i;
      const double* b = V.bp + 6 * (size_t)i;
#pragma unroll
      for (int a = 0; a < 6; a++) sc += u[a] * (lambda * V.damp_s * u[a] + b[a]);
      double T[7];
#pragma unroll
      for (int a = 0; a < 7; a++) T[a] = V.poses[7 * (size_t)p + a];
      se3_oplus(T, u);
#pragma unroll
      for (int a = 0; a < 7; a++) V.poses_new[7 * (size_t)p + a] = T[a];
    }
  } else {
    const int g = ((int)blockIdx.x - nb_pose) * 256 + threadIdx.x;
    const int l = g >> 3, j = g & 7;
    const bool live = l < V.L;
    double t[3] = {0, 0, 0};
    int i0 = 0, i1 = 0;
    if (live) { i0 = V.pt_start[l]; i1 = V.pt_start[l + 1]; }
    for (int i = i0 + j; i < i1; i += 8) {
      const int k = V.pt_edges[i];
      const int fi = V.pt_fi[i];
      if (fi < 0) continue;
      const double* W = V.e_W + (size_t)k * 18;
      const double* xp = V.x + 6 * (size_t)fi;
#pragma unroll
      for (int b = 0; b < 3; b++)
#pragma unroll
        for (int a = 0; a < 6; a++) t[b] += W[3 * a + b] * xp[a];
    }
#pragma unroll
    for (int b = 0; b < 3; b++) {
      t[b] = dpp_xor_add(t[b], 0);
      t[b] = dpp_xor_add(t[b], 1);
      t[b] = dpp_xor_add(t[b], 2);
    }
    if (live && j == 0) {
      const int n = 6 * V.nfree;
      double* xl = V.x + n + 3 * (size_t)l;
      if (i1 == i0) {
        xl[0] = xl[1] = xl[2] = 0;
      } else {
        const double* bl = V.bl + 3 * (size_t)l;
        const double c[3] = {bl[0] - t[0], bl[1] - t[1], bl[2] - t[2]};
        const double* D = V.Dinv + 9 * (size_t)l;
        double u[3];
        u[0] = D[0] * c[0] + D[1] * c[1] + D[2] * c[2];
        u[1] = D[3] * c[0] + D[4] * c[1] + D[5] * c[2];
        u[2] = D[6] * c[0] + D[7] * c[1] + D[8] * c[2];
        // (block_solver.hpp:445-446 returns before the landmark part); a kept landmark is an unknown of the reduced system: the back
        // substitution has written its x already
        if (failed || (V.nkept && V.kept_slot[l] >= 0)) { u[0] = xl[0]; u[1] = xl[1]; u[2] = xl[2]; }
        const double* X = V.points + 3 * (size_t)l;
        double* Xn = V.points_new + 3 * (size_t)l;
#pragma unroll
        for (int a = 0; a < 3; a++) { xl[a] = u[a]; sc += u[a] * (lambda * u[a] + bl[a]); Xn[a] = X[a] + u[a]; }
      }
    }
  }
  block_reduce_publish<false>(sc, V.partial2, pub);
}

// per-edge depth sign at the current state (EdgeSE3ProjectXYZ::isDepthPositive)
__global__ void __launch_bounds__(256) k_edge_depth(BaView V, uint8_t* __restrict__ out) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= V.E) return;
  const double* T = V.poses + 7 * (size_t)V.e_pose[k];
  const double* X = V.points + 3 * (size_t)V.e_point[k];
  double R[9];
  quat_to_R(T + 3, R);
  const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + T[2];
  out[k] = z > 0.0;
}

// ---------------------------------------------------------------------------------------- B1
// Optimizer::PoseOptimization (reference src/Optimizer.cc:744-1028, mono edges
// EdgeSE3ProjectXYZOnlyPose, src/OptimizableTypes.cpp:51-63): one camera, N unary reprojection edges,
// 4 rounds x optimize(10) of g2o's Levenberg with a dense 6x6 solve, re-classifying inliers after every
// round (chi2 > 5.991 as float), Huber removed after round 2.  ONE workgroup runs the whole thing for
// one frame -- about 40 LM iterations with no host round trip; frames are batched over the grid.
struct PoseAccum { double v[28]; };  // 21 upper-H + 6 b + 1 chi
// One value over the workgroup in a fixed order: xor-butterfly inside each wave, then the four wave sums in wave order.
// (What the chi2-only evaluation of a trial needs: running the 28-value reduction for it cost 2 us per LM trial.)
__device__ __forceinline__ double block_sum_one(double v, double* part4) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) part4[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = (part4[0] + part4[1]) + (part4[2] + part4[3]);
  __syncthreads();
  return r;
}
// The kernel is a LATENCY path -- Tracking calls PoseOptimization two or three times per frame, one frame at a time, and a
// workgroup runs ~40 dependent LM iterations -- so everything that repeats per iteration is kept off the memory system and off
// the serial thread: a thread's correspondences (up to kPoseEdgesPerThread x 256 per frame; more fall back to global reads) live
// in registers across all iterations together with their level flag and last chi2; the Jacobian pass reduces its 28 sums in one
// pass through LDS; a trial's chi2-only pass reduces ONE value; the 6x6 solve forms 1 / L_ii once per row (v_rsq_f64 + two
// Newton steps, as the tile Cholesky does) instead of 27 double-precision divisions and 6 square roots on a single lane.
// (Measured, one frame of 300 matches: 363 us per call before, of which ~2.8 us per LM trial were the divisions.)
constexpr int kPoseEdgesPerThread = 5;
#ifdef DVM_POSE_PROF   // make EXTRA=-DDVM_POSE_PROF: per-phase clocks of workgroup 0 (dvm_debug_pose_prof), measurement builds only
__device__ unsigned long long g_pose_prof[16];
#define POSE_T(k) do { if (tid == 0 && blockIdx.x == 0) { const unsigned long long now_ = wall_clock64(); g_pose_prof[k] += now_ - pose_last_; pose_last_ = now_; } } while (0)
#else
#define POSE_T(k) do {} while (0)
#endif
__global__ void __launch_bounds__(256) k_pose_optimize(const double* __restrict__ pose_in, const double* __restrict__ Xw,
                                                       const double* __restrict__ obs, const double* __restrict__ info,
                                                       const int32_t* __restrict__ n_per_frame, int stride, double fx,
                                                       double fy, double cx, double cy, double* __restrict__ pose_out,
                                                       uint8_t* __restrict__ outlier, int32_t* __restrict__ n_inliers,
                                                       double* __restrict__ chi_scratch) {
  __shared__ double s_park[256 * 29];
  __shared__ double s_part[4 * 28];
  __shared__ double s_sum[28];
  __shared__ double s_T[7], s_Tbak[7], s_T0[7];
  __shared__ double s_lambda, s_ni, s_cur, s_ini, s_rho;
  __shared__ int s_ctl, s_qmax, s_nbad, s_nact, s_lin;
  const int f = blockIdx.x, tid = threadIdx.x;
#ifdef DVM_POSE_PROF
  unsigned long long pose_last_ = wall_clock64();
#endif
  const int N = n_per_frame[f];
  const double* X = Xw + (size_t)f * stride * 3;
  const double* O = obs + (size_t)f * stride * 2;
  const double* W = info + (size_t)f * stride;
  uint8_t* outl = outlier + (size_t)f * stride;
  double* last_chi = chi_scratch + (size_t)f * stride;  // e->chi2() as g2o reports it (last evaluation): edges beyond the register-resident ones
  const double delta = (double)sqrtf(5.991f);
  const float chi2Mono = 5.991f;
  constexpr int EPT = kPoseEdgesPerThread;
  if (tid < 7) {
    double v = pose_in[7 * (size_t)f + tid];
    s_T0[tid] = v;
  }
  __syncthreads();
  if (tid == 0) quat_normalize(&s_T0[3]);
  // this thread's correspondences: registers for the first EPT, global memory beyond
  double eX[EPT][3], eO[EPT][2], eW[EPT], eChi[EPT];
  bool eOut[EPT];       // level(1) == outlier flag in the reference's bookkeeping
#pragma unroll
  for (int e = 0; e < EPT; e++) {
    const int i = tid + 256 * e;
    const bool in = i < N;
    const int ii = in ? i : 0;
    eX[e][0] = X[3 * ii]; eX[e][1] = X[3 * ii + 1]; eX[e][2] = X[3 * ii + 2];
    eO[e][0] = O[2 * ii]; eO[e][1] = O[2 * ii + 1];
    eW[e] = W[ii];
    eChi[e] = 0; eOut[e] = false;
  }
  for (int i = tid + 256 * EPT; i < N; i += 256) { outl[i] = 0; last_chi[i] = 0; }
  __syncthreads();
  if (N < 3) {  // nInitialCorrespondences < 3: return 0, pose untouched (Optimizer.cc:904-905)
    if (tid < 7) pose_out[7 * (size_t)f + tid] = pose_in[7 * (size_t)f + tid];
    for (int i = tid; i < N; i += 256) outl[i] = 0;
    if (tid == 0) n_inliers[f] = 0;
    return;
  }
  // one active edge at pose (R, T): chi2 (always; returned), H / b terms (jac)
  auto edge = [&](const double* R, const double* T, const double* Xp, double o0, double o1, double w0, bool jac, bool robust_on, PoseAccum& a) -> double {
    double Xc[3];
    mat3_vec(R, Xp, Xc);
    Xc[0] += T[0]; Xc[1] += T[1]; Xc[2] += T[2];
    const double x = Xc[0], y = Xc[1], z = Xc[2];
    const double e0 = o0 - (fx * x / z + cx), e1 = o1 - (fy * y / z + cy);
    const double chi2 = e0 * w0 * e0 + e1 * w0 * e1;
    double r0, r1;
    robustify(chi2, robust_on ? delta : 0.0, r0, r1);
    a.v[27] += r0;
    if (jac) {
      const double J[6] = {-(fx / z), 0, fx * x / (z * z), 0, -(fy / z), fy * y / (z * z)};
      const double S[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
      double B[12];
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) B[6 * r + c] = J[3 * r] * S[c] + J[3 * r + 1] * S[6 + c] + J[3 * r + 2] * S[12 + c];
      const double w = r1 * w0, wr0 = -w0 * e0 * r1, wr1 = -w0 * e1 * r1;
      int t = 0;
#pragma unroll
      for (int p = 0; p < 6; p++) {
        a.v[21 + p] += B[p] * wr0 + B[6 + p] * wr1;
#pragma unroll
        for (int q = 0; q <= p; q++) a.v[t++] += w * (B[p] * B[q] + B[6 + p] * B[6 + q]);
      }
    }
    return chi2;
  };
  // evaluates the active edges at pose T: chi (always), H / b (jac); updates the edges' last chi2.  Result in s_sum ([27] = chi2).
  auto eval = [&](const double* T, bool jac, bool robust_on) {
    PoseAccum a;
#pragma unroll
    for (int i = 0; i < 28; i++) a.v[i] = 0;
    double R[9];
    quat_to_R(T + 3, R);
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      if (tid + 256 * e < N && !eOut[e]) eChi[e] = edge(R, T, eX[e], eO[e][0], eO[e][1], eW[e], jac, robust_on, a);
    }
    for (int i = tid + 256 * EPT; i < N; i += 256) {
      if (outl[i]) continue;
      last_chi[i] = edge(R, T, X + 3 * i, O[2 * i], O[2 * i + 1], W[i], jac, robust_on, a);
    }
    if (jac) block_sum_lds<28>(a.v, s_park, s_part, s_sum);
    else {
      const double c = block_sum_one(a.v[27], s_part);
      if (tid == 0) s_sum[27] = c;
      __syncthreads();
    }
  };

  bool robust_on = true;
  for (int round = 0; round < 4; round++) {
    if (tid < 7) s_T[tid] = s_T0[tid];  // vSE3->setEstimate(pFrame->GetPose()) every round
    if (tid == 0) { s_nact = 0; s_ctl = 0; s_nbad = 0; s_lin = 0; }
    __syncthreads();
    int my = 0;
#pragma unroll
    for (int e = 0; e < EPT; e++) my += (tid + 256 * e < N && !eOut[e]) ? 1 : 0;
    for (int i = tid + 256 * EPT; i < N; i += 256) my += outl[i] ? 0 : 1;
    if (my) atomicAdd(&s_nact, my);
    __syncthreads();
    const int nact = s_nact;
    POSE_T(0);
    for (int it = 0; it < 10 && nact > 0; it++) {
      // Speculative linearisation (as the tile solver's LM loop does it): an accepted trial has evaluated its state WITH the Jacobians, so
      // the iteration that follows finds H, b and chi2 of its state in s_sum already -- one pass per accepted trial instead of two
      // (chi2 only, then the same edges again with Jacobians); a rejected trial's sums are simply overwritten.
      const bool have_lin = s_lin != 0;
      __syncthreads();
      if (tid == 0) s_lin = 0;
      if (!have_lin) eval(s_T, true, robust_on);
      POSE_T(1);
      if (tid == 0) {
        s_cur = s_sum[27]; s_ini = s_sum[27];
        if (it == 0) {
          double mx = 0;
          int t = 0;
          for (int p = 0; p < 6; p++) for (int q = 0; q <= p; q++) { if (p == q) mx = fmax(mx, fabs(s_sum[t])); t++; }
          s_lambda = 1e-5 * mx; s_ni = 2; s_nbad = 0;
        }
        s_qmax = 0;
      }
      __syncthreads();
      double Hs[21], bs[6];
      if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 21; i++) Hs[i] = s_sum[i];
#pragma unroll
        for (int i = 0; i < 6; i++) bs[i] = s_sum[21 + i];
      }
      POSE_T(2);
      while (true) {
        double xs[6];
        bool ok = true;
        if (tid == 0) {
          for (int i = 0; i < 7; i++) s_Tbak[i] = s_T[i];
          // dense 6x6 Cholesky of (H + lambda I), lower-packed Hs[p(p+1)/2 + q]; ri[j] = 1 / L_jj
          double Lm[21], ri[6];
#pragma unroll
          for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) {
              double sacc = Hs[i * (i + 1) / 2 + j] + (i == j ? s_lambda : 0.0);
#pragma unroll
              for (int k = 0; k < j; k++) sacc -= Lm[i * (i + 1) / 2 + k] * Lm[j * (j + 1) / 2 + k];
              if (i == j) {
                if (!(sacc > 0)) ok = false;
                const double dd = sacc > 0 ? sacc : 1.0;
                double y = __builtin_amdgcn_rsq(dd);
                y = __builtin_fma(0.5 * y, __builtin_fma(-dd * y, y, 1.0), y);
                y = __builtin_fma(0.5 * y, __builtin_fma(-dd * y, y, 1.0), y);
                double sq = dd * y;
                sq = __builtin_fma(0.5 * y, __builtin_fma(-sq, sq, dd), sq);
                Lm[i * (i + 1) / 2 + i] = sq; ri[i] = y;
              } else Lm[i * (i + 1) / 2 + j] = sacc * ri[j];
            }
          if (ok) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
              double sacc = bs[i];
#pragma unroll
              for (int k = 0; k < i; k++) sacc -= Lm[i * (i + 1) / 2 + k] * xs[k];
              xs[i] = sacc * ri[i];
            }
#pragma unroll
            for (int i = 5; i >= 0; i--) {
              double sacc = xs[i];
#pragma unroll
              for (int k = i + 1; k < 6; k++) sacc -= Lm[k * (k + 1) / 2 + i] * xs[k];
              xs[i] = sacc * ri[i];
            }
            se3_oplus(s_T, xs);
          }
          s_ctl = ok ? 1 : 0;
        }
        __syncthreads();
        POSE_T(3);
        const bool okb = s_ctl != 0;
        const bool spec = it + 1 < 10;        // (the last iteration of a round: nobody would use the linearisation)
        if (okb) eval(s_T, spec, robust_on);
        POSE_T(4);
        if (tid == 0) {
          const double tempChi = okb ? s_sum[27] : 1.7976931348623157e308;
          double rho = s_cur - tempChi;
          double scale = 0;
          if (okb) for (int j = 0; j < 6; j++) scale += xs[j] * (s_lambda * xs[j] + bs[j]);
          scale += 1e-3;
          rho /= scale;
          if (rho > 0 && isfinite(tempChi)) {
            double alpha = 1. - f64_cube(2 * rho - 1);   // pow(2 rho - 1, 3) as the shared double-precision spec forms it (f64_spec.h)
            alpha = fmin(alpha, 2. / 3.);
            s_lambda *= fmax(1. / 3., alpha);
            s_ni = 2;
            s_cur = tempChi;
            if (spec) s_lin = 1;                   // s_sum holds this state's linearisation
          } else {
            s_lambda *= s_ni; s_ni *= 2;
            for (int i = 0; i < 7; i++) s_T[i] = s_Tbak[i];
          }
          s_qmax++;
          s_rho = rho;
          s_ctl = (rho < 0 && s_qmax < 10) ? 1 : 0;  // continue the trial loop?
        }
        __syncthreads();
        POSE_T(5);
        if (!s_ctl) break;
        __syncthreads();
      }
      if (tid == 0) {
        int stop = 0;
        if (s_qmax == 10 || s_rho == 0) stop = 1;
        else {
          if ((s_ini - s_cur) * 1e3 < s_ini) s_nbad++; else s_nbad = 0;
          if (s_nbad >= 3) stop = 1;
        }
        s_ctl = stop;
      }
      __syncthreads();
      const int stop = s_ctl;
      __syncthreads();
      POSE_T(6);
#ifdef DVM_POSE_PROF
      if (tid == 0 && blockIdx.x == 0) g_pose_prof[15]++;
#endif
      if (stop) break;
    }
    // classification (Optimizer.cc:923-948): outliers recompute their error, inliers report the last evaluation
    {
      double R[9];
      quat_to_R(s_T + 3, R);
      auto fresh = [&](const double* Xp, double o0, double o1, double w0) {
        double Xc[3];
        mat3_vec(R, Xp, Xc);
        Xc[0] += s_T[0]; Xc[1] += s_T[1]; Xc[2] += s_T[2];
        const double e0 = o0 - (fx * Xc[0] / Xc[2] + cx), e1 = o1 - (fy * Xc[1] / Xc[2] + cy);
        return e0 * w0 * e0 + e1 * w0 * e1;
      };
#pragma unroll
      for (int e = 0; e < EPT; e++) {
        if (tid + 256 * e < N) {
          if (eOut[e]) eChi[e] = fresh(eX[e], eO[e][0], eO[e][1], eW[e]);
          eOut[e] = (float)eChi[e] > chi2Mono;
        }
      }
      for (int i = tid + 256 * EPT; i < N; i += 256) {
        if (outl[i]) last_chi[i] = fresh(X + 3 * i, O[2 * i], O[2 * i + 1], W[i]);
        outl[i] = (float)last_chi[i] > chi2Mono ? 1 : 0;
      }
    }
    if (round == 2) robust_on = false;
    __syncthreads();
    POSE_T(7);
    if (N < 10) break;  // optimizer.edges().size() < 10
  }
  if (tid == 0) s_nact = 0;
  __syncthreads();
  int bad = 0;
#pragma unroll
  for (int e = 0; e < EPT; e++) {
    const int i = tid + 256 * e;
    if (i < N) { outl[i] = eOut[e] ? 1 : 0; bad += eOut[e] ? 1 : 0; }
  }
  for (int i = tid + 256 * EPT; i < N; i += 256) bad += outl[i];
  if (bad) atomicAdd(&s_nact, bad);
  __syncthreads();
  if (tid < 7) pose_out[7 * (size_t)f + tid] = s_T[tid];
  if (tid == 0) n_inliers[f] = N - s_nact;
}

#ifdef DVM_POSE_PROF
extern "C" int dvm_debug_pose_prof(unsigned long long* out, int reset) {
  unsigned long long h[16];
  int rc = (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pose_prof), sizeof(h));
  for (int i = 0; i < 16; i++) out[i] = h[i];
  if (reset) { for (auto& v : h) v = 0; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pose_prof), h, sizeof(h)); }
  return rc;
}
#endif
void ba_launch_pose_optimize(hipStream_t s, const double* pose_in, const double* Xw, const double* obs, const double* info,
                             const int32_t* n_per_frame, int stride, int batch, double fx, double fy, double cx, double cy,
                             double* pose_out, uint8_t* outlier, int32_t* n_inliers, double* chi_scratch) {
  hipLaunchKernelGGL(k_pose_optimize, dim3(batch), dim3(256), 0, s, pose_in, Xw, obs, info, n_per_frame, stride, fx, fy, cx, cy,
                     pose_out, outlier, n_inliers, chi_scratch);
}

// ---------------------------------------------------------------------------------------- B9
// Optimizer::OptimizeSim3 (reference src/Optimizer.cc:1960-2212): one 7-DoF g2o::Sim3 vertex, two
// reprojection edges per correspondence (EdgeSim3ProjectXYZ, EdgeInverseSim3ProjectXYZ) whose Jacobians
// g2o takes NUMERICALLY (central differences, delta 1e-9, base_binary_edge.hpp:131-205) because the
// analytic linearizeOplus is commented out (include/OptimizableTypes.h:186,205); dense 7x7 Levenberg,
// optimize(5), inlier test chi2 <= th2, robust kernel off, optimize(5 or 10), final inlier count.
// One workgroup runs everything; the 14 perturbed Sim3 states (and inverses) are shared by all edges.
struct Sim3d { double q[4]; double t[3]; double s; };
__device__ void sim3_exp(const double* u, Sim3d& S) {  // g2o::Sim3(const Vector7d&), sim3.h:62-125
  const double om0 = u[0], om1 = u[1], om2 = u[2], sigma = u[6];
  const double theta = sqrt(om0 * om0 + om1 * om1 + om2 * om2);
  const double O[9] = {0, -om2, om1, om2, 0, -om0, -om1, om0, 0};
  double O2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
  S.s = exp(sigma);
  const double eps = 0.00001;
  double A, B, C, R[9];
  if (fabs(sigma) < eps) {
    C = 1;
    if (theta < eps) { A = 0.5; B = 1. / 6.; for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + O[i] + O2[i]; }
    else {
      const double th2 = theta * theta;
      A = (1 - cos(theta)) / th2; B = (theta - sin(theta)) / (th2 * theta);
      for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + sin(theta) / theta * O[i] + (1 - cos(theta)) / (theta * theta) * O2[i];
    }
  } else {
    C = (S.s - 1) / sigma;
    if (theta < eps) {
      const double s2 = sigma * sigma;
      A = ((sigma - 1) * S.s + 1) / s2; B = ((0.5 * s2 - sigma + 1) * S.s) / (s2 * sigma);
      for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + O[i] + O2[i];
    } else {
      for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + sin(theta) / theta * O[i] + (1 - cos(theta)) / (theta * theta) * O2[i];
      const double a = S.s * sin(theta), b = S.s * cos(theta), th2 = theta * theta, s2 = sigma * sigma, c = th2 + s2;
      A = (a * sigma + (1 - b) * theta) / (theta * c);
      B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / th2;
    }
  }
  R_to_quat(R, S.q);
  double W[9];
  for (int i = 0; i < 9; i++) W[i] = A * O[i] + B * O2[i] + C * ((i % 4 == 0) ? 1.0 : 0.0);
  mat3_vec(W, u + 3, S.t);
}
__device__ void sim3_mul(const Sim3d& a, const Sim3d& b, Sim3d& o) {
  const double* p = a.q; const double* q = b.q;
  o.q[3] = p[3] * q[3] - p[0] * q[0] - p[1] * q[1] - p[2] * q[2];
  o.q[0] = p[3] * q[0] + p[0] * q[3] + p[1] * q[2] - p[2] * q[1];
  o.q[1] = p[3] * q[1] + p[1] * q[3] + p[2] * q[0] - p[0] * q[2];
  o.q[2] = p[3] * q[2] + p[2] * q[3] + p[0] * q[1] - p[1] * q[0];
  double R[9], rt[3];
  quat_to_R(a.q, R);
  mat3_vec(R, b.t, rt);
  for (int i = 0; i < 3; i++) o.t[i] = a.s * rt[i] + a.t[i];
  o.s = a.s * b.s;
}
__device__ void sim3_inv(const Sim3d& a, Sim3d& o) {
  o.q[0] = -a.q[0]; o.q[1] = -a.q[1]; o.q[2] = -a.q[2]; o.q[3] = a.q[3];
  double R[9];
  const double v[3] = {(-1. / a.s) * a.t[0], (-1. / a.s) * a.t[1], (-1. / a.s) * a.t[2]};
  quat_to_R(o.q, R);
  mat3_vec(R, v, o.t);
  o.s = 1. / a.s;
}
struct Sim3M { double R[9]; double t[3]; double s; };  // map-ready form: x -> s R x + t
__device__ __forceinline__ void sim3_to_map(const Sim3d& a, Sim3M& m) {
  quat_to_R(a.q, m.R);
  m.t[0] = a.t[0]; m.t[1] = a.t[1]; m.t[2] = a.t[2]; m.s = a.s;
}
__device__ __forceinline__ void sim3_proj(const Sim3M& m, const double* x, const double* K, double& u, double& v) {
  double rx[3];
  mat3_vec(m.R, x, rx);
  const double X = m.s * rx[0] + m.t[0], Y = m.s * rx[1] + m.t[1], Z = m.s * rx[2] + m.t[2];
  u = K[0] * X / Z + K[2];
  v = K[1] * Y / Z + K[3];
}

__global__ void __launch_bounds__(256) k_optimize_sim3(double* __restrict__ S12io, int fix_scale, const double* __restrict__ P1c,
                                                       const double* __restrict__ P2c, const double* __restrict__ obs1,
                                                       const double* __restrict__ obs2, const double* __restrict__ w1,
                                                       const double* __restrict__ w2, int N, const double* __restrict__ Kio,
                                                       double th2, uint8_t* __restrict__ inlier, int32_t* __restrict__ nin_out,
                                                       double* __restrict__ chi_scratch, uint8_t* __restrict__ flag_scratch) {
  __shared__ double s_park[256 * 37];   // the 36 sums of a Jacobian pass in ONE pass through LDS
  __shared__ double s_part[4 * 36];
  __shared__ double s_sum[36];
  __shared__ Sim3d s_S, s_bak;
  __shared__ Sim3M s_M[30];   // [0] S, [1] S^-1, [2+2d] S+d, [3+2d] (S+d)^-1, [16+2d] S-d, [17+2d] (S-d)^-1
  __shared__ double s_K[8];
  __shared__ double s_lambda, s_ni, s_cur, s_ini, s_rho;
  __shared__ int s_ctl, s_qmax, s_nbad, s_cnt;
  const int tid = threadIdx.x;
  double* chi12 = chi_scratch;
  double* chi21 = chi_scratch + N;
  uint8_t* alive = flag_scratch;
  uint8_t* robust = flag_scratch + N;
  if (tid < 8) s_K[tid] = Kio[tid];
  if (tid == 0) {
    for (int i = 0; i < 4; i++) s_S.q[i] = S12io[i];
    for (int i = 0; i < 3; i++) s_S.t[i] = S12io[4 + i];
    s_S.s = S12io[7];
  }
  for (int i = tid; i < N; i += 256) { alive[i] = 1; robust[i] = 1; inlier[i] = 0; chi12[i] = 0; chi21[i] = 0; }
  __syncthreads();
  const double deltaHuber = (double)sqrtf((float)th2);

  // refreshes s_M[0..1] (jac=false) or all 30 maps (jac=true) from s_S
  auto prepare = [&](bool jac) {
    if (tid == 0) { sim3_to_map(s_S, s_M[0]); Sim3d Si; sim3_inv(s_S, Si); sim3_to_map(Si, s_M[1]); }
    if (jac && tid >= 64 && tid < 78) {
      const int k = tid - 64, d = k >> 1, sgn = k & 1;
      double u[7] = {0, 0, 0, 0, 0, 0, 0};
      u[d] = sgn ? -1e-9 : 1e-9;
      if (fix_scale) u[6] = 0;
      Sim3d E, Sx, Sxi;
      sim3_exp(u, E);
      sim3_mul(E, s_S, Sx);
      sim3_inv(Sx, Sxi);
      sim3_to_map(Sx, s_M[(sgn ? 16 : 2) + 2 * d]);
      sim3_to_map(Sxi, s_M[(sgn ? 17 : 3) + 2 * d]);
    }
    __syncthreads();
  };
  auto eval = [&](bool jac) {
    prepare(jac);
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = 0;
    for (int i = tid; i < N; i += 256) {
      if (!alive[i]) continue;
      const double* x1 = P1c + 3 * i; const double* x2 = P2c + 3 * i;
      double u, v;
      sim3_proj(s_M[0], x2, s_K, u, v);
      const double a0 = obs1[2 * i] - u, a1 = obs1[2 * i + 1] - v;
      sim3_proj(s_M[1], x1, s_K + 4, u, v);
      const double b0 = obs2[2 * i] - u, b1 = obs2[2 * i + 1] - v;
      const double c12 = w1[i] * (a0 * a0 + a1 * a1), c21 = w2[i] * (b0 * b0 + b1 * b1);
      chi12[i] = c12; chi21[i] = c21;
      const double dl = robust[i] ? deltaHuber : 0.0;
      double r0a, r1a, r0b, r1b;
      robustify(c12, dl, r0a, r1a);
      robustify(c21, dl, r0b, r1b);
      acc[35] += r0a;
      acc[35] += r0b;
      if (jac) {
        double J12[14], J21[14];
#pragma unroll
        for (int d = 0; d < 7; d++) {
          double up, vp, um, vm;
          sim3_proj(s_M[2 + 2 * d], x2, s_K, up, vp); sim3_proj(s_M[16 + 2 * d], x2, s_K, um, vm);
          // e(+d) - e(-d) = (obs - proj+) - (obs - proj-)
          J12[d] = 5e8 * ((obs1[2 * i] - up) - (obs1[2 * i] - um)); J12[7 + d] = 5e8 * ((obs1[2 * i + 1] - vp) - (obs1[2 * i + 1] - vm));
          sim3_proj(s_M[3 + 2 * d], x1, s_K + 4, up, vp); sim3_proj(s_M[17 + 2 * d], x1, s_K + 4, um, vm);
          J21[d] = 5e8 * ((obs2[2 * i] - up) - (obs2[2 * i] - um)); J21[7 + d] = 5e8 * ((obs2[2 * i + 1] - vp) - (obs2[2 * i + 1] - vm));
        }
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
          const double* J = pass ? J21 : J12;
          const double e0 = pass ? b0 : a0, e1 = pass ? b1 : a1, w0 = pass ? w2[i] : w1[i], r1 = pass ? r1b : r1a;
          const double w = r1 * w0, wr0 = -w0 * e0 * r1, wr1 = -w0 * e1 * r1;
          int t = 0;
#pragma unroll
          for (int p = 0; p < 7; p++) {
            acc[28 + p] += J[p] * wr0 + J[7 + p] * wr1;
#pragma unroll
            for (int q = 0; q <= p; q++) acc[t++] += w * (J[p] * J[q] + J[7 + p] * J[7 + q]);
          }
        }
      }
    }
    if (jac) block_sum_lds<36>(acc, s_park, s_part, s_sum);
    else {          // a trial's chi2: one value (the full reduction here cost ~2 us per LM trial)
      const double c = block_sum_one(acc[35], s_part);
      if (tid == 0) s_sum[35] = c;
      __syncthreads();
    }
  };
  auto optimize = [&](int iters) {
    for (int it = 0; it < iters; it++) {
      eval(true);
      double Hs[28], bs[7], xs[7];
      if (tid == 0) {
        s_cur = s_sum[35]; s_ini = s_sum[35];
        for (int i = 0; i < 28; i++) Hs[i] = s_sum[i];
        for (int i = 0; i < 7; i++) bs[i] = s_sum[28 + i];
        if (it == 0) {
          double mx = 0;
          for (int p = 0; p < 7; p++) mx = fmax(mx, fabs(Hs[p * (p + 1) / 2 + p]));
          s_lambda = 1e-5 * mx; s_ni = 2; s_nbad = 0;
        }
        s_qmax = 0;
      }
      __syncthreads();
      while (true) {
        if (tid == 0) {
          s_bak = s_S;
          // dense 7x7 Cholesky; ri[j] = 1 / L_jj by v_rsq_f64 + two Newton steps: no double-precision division or square root on
          // this single lane (35 of them before: ~3 us per LM trial)
          double Lm[28], ri[7];
          bool ok = true;
#pragma unroll
          for (int i = 0; i < 7; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) {
              double sacc = Hs[i * (i + 1) / 2 + j] + (i == j ? s_lambda : 0.0);
#pragma unroll
              for (int k = 0; k < j; k++) sacc -= Lm[i * (i + 1) / 2 + k] * Lm[j * (j + 1) / 2 + k];
              if (i == j) {
                if (!(sacc > 0)) ok = false;
                const double dd = sacc > 0 ? sacc : 1.0;
                double y = __builtin_amdgcn_rsq(dd);
                y = __builtin_fma(0.5 * y, __builtin_fma(-dd * y, y, 1.0), y);
                y = __builtin_fma(0.5 * y, __builtin_fma(-dd * y, y, 1.0), y);
                double sq = dd * y;
                sq = __builtin_fma(0.5 * y, __builtin_fma(-sq, sq, dd), sq);
                Lm[i * (i + 1) / 2 + i] = sq; ri[i] = y;
              } else Lm[i * (i + 1) / 2 + j] = sacc * ri[j];
            }
          if (ok) {
#pragma unroll
            for (int i = 0; i < 7; i++) {
              double sacc = bs[i];
#pragma unroll
              for (int k = 0; k < i; k++) sacc -= Lm[i * (i + 1) / 2 + k] * xs[k];
              xs[i] = sacc * ri[i];
            }
#pragma unroll
            for (int i = 6; i >= 0; i--) {
              double sacc = xs[i];
#pragma unroll
              for (int k = i + 1; k < 7; k++) sacc -= Lm[k * (k + 1) / 2 + i] * xs[k];
              xs[i] = sacc * ri[i];
            }
            double u[7];
            for (int i = 0; i < 7; i++) u[i] = xs[i];
            if (fix_scale) u[6] = 0;
            Sim3d E, Sn;
            sim3_exp(u, E);
            sim3_mul(E, s_S, Sn);
            s_S = Sn;
          }
          s_ctl = ok ? 1 : 0;
        }
        __syncthreads();
        const bool okb = s_ctl != 0;
        if (okb) eval(false);
        if (tid == 0) {
          const double tempChi = okb ? s_sum[35] : 1.7976931348623157e308;
          double rho = s_cur - tempChi;
          double scale = 0;
          if (okb) for (int j = 0; j < 7; j++) scale += xs[j] * (s_lambda * xs[j] + bs[j]);
          scale += 1e-3;
          rho /= scale;
          if (rho > 0 && isfinite(tempChi)) {
            double alpha = 1. - f64_cube(2 * rho - 1);   // pow(2 rho - 1, 3) as the shared double-precision spec forms it (f64_spec.h)
            alpha = fmin(alpha, 2. / 3.);
            s_lambda *= fmax(1. / 3., alpha);
            s_ni = 2;
            s_cur = tempChi;
          } else {
            s_lambda *= s_ni; s_ni *= 2;
            s_S = s_bak;
          }
          s_qmax++;
          s_rho = rho;
          s_ctl = (rho < 0 && s_qmax < 10) ? 1 : 0;
        }
        __syncthreads();
        const int again = s_ctl;
        __syncthreads();
        if (!again) break;
      }
      if (tid == 0) {
        int stop = 0;
        if (s_qmax == 10 || s_rho == 0) stop = 1;
        else {
          if ((s_ini - s_cur) * 1e3 < s_ini) s_nbad++; else s_nbad = 0;
          if (s_nbad >= 3) stop = 1;
        }
        s_ctl = stop;
      }
      __syncthreads();
      const int stop = s_ctl;
      __syncthreads();
      if (stop) break;
    }
  };

  optimize(5);
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  int bad = 0;
  for (int i = tid; i < N; i += 256) {
    if (chi12[i] > th2 || chi21[i] > th2) { alive[i] = 0; bad++; } else robust[i] = 0;
  }
  if (bad) atomicAdd(&s_cnt, bad);
  __syncthreads();
  const int nBad = s_cnt;
  __syncthreads();
  if (N - nBad < 10) {
    if (tid == 0) *nin_out = 0;
    return;
  }
  optimize(nBad > 0 ? 10 : 5);
  prepare(false);
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  int in = 0;
  for (int i = tid; i < N; i += 256) {
    if (!alive[i]) continue;
    double u, v;
    sim3_proj(s_M[0], P2c + 3 * i, s_K, u, v);
    const double a0 = obs1[2 * i] - u, a1 = obs1[2 * i + 1] - v;
    sim3_proj(s_M[1], P1c + 3 * i, s_K + 4, u, v);
    const double b0 = obs2[2 * i] - u, b1 = obs2[2 * i + 1] - v;
    const double c12 = w1[i] * (a0 * a0 + a1 * a1), c21 = w2[i] * (b0 * b0 + b1 * b1);
    if (!(c12 > th2 || c21 > th2)) { inlier[i] = 1; in++; }
  }
  if (in) atomicAdd(&s_cnt, in);
  __syncthreads();
  if (tid == 0) {
    *nin_out = s_cnt;
    for (int i = 0; i < 4; i++) S12io[i] = s_S.q[i];
    for (int i = 0; i < 3; i++) S12io[4 + i] = s_S.t[i];
    S12io[7] = s_S.s;
  }
}

void ba_launch_optimize_sim3(hipStream_t s, double* S12io, int fix_scale, const double* P1c, const double* P2c,
                             const double* obs1, const double* obs2, const double* w1, const double* w2, int N,
                             const double* K, double th2, uint8_t* inlier, int32_t* nin, double* chi_scratch, uint8_t* flag_scratch) {
  hipLaunchKernelGGL(k_optimize_sim3, dim3(1), dim3(256), 0, s, S12io, fix_scale, P1c, P2c, obs1, obs2, w1, w2, N, K, th2, inlier,
                     nin, chi_scratch, flag_scratch);
}

// ------------------------------------------------------------------------------------------ Sim3Solver
// Sim3Solver::ComputeSim3 (Horn 1987 closed form, reference src/Sim3Solver.cc:294-385) + CheckInliers (:387-408) for
// a batch of RANSAC hypotheses, one wavefront each: the 3-point solve is wave-uniform (every lane computes it, no
// communication), the N correspondences are strided over the lanes, inliers counted by ballots.  The minimal sets are
// input (the reference draws them with DUtils::Random).  float / double split as in the reference except the 4x4
// eigen-decomposition: cyclic Jacobi in double ("Horn spec", same as the oracle) instead of Eigen::EigenSolver<float>.
// (jacobi4_dev: jacobi4.h)

__global__ void __launch_bounds__(64) k_sim3_hypotheses(const float* __restrict__ P1c, const float* __restrict__ P2c,
                                                        const float* __restrict__ max_err1, const float* __restrict__ max_err2,
                                                        int N, const float* __restrict__ K, const int32_t* __restrict__ triples,
                                                        int H, int fix_scale, float* __restrict__ T12,
                                                        int32_t* __restrict__ n_inliers, uint8_t* __restrict__ mask) {
  const int h = blockIdx.x, lane = threadIdx.x;
  if (h >= H) return;
  float P1[3][3], P2[3][3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const int idx = triples[3 * h + c];
#pragma unroll
    for (int r = 0; r < 3; r++) { P1[r][c] = P1c[3 * idx + r]; P2[r][c] = P2c[3 * idx + r]; }
  }
  float O1[3], O2[3], Pr1[3][3], Pr2[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    O1[r] = ((P1[r][0] + P1[r][1]) + P1[r][2]) / 3.0f; O2[r] = ((P2[r][0] + P2[r][1]) + P2[r][2]) / 3.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) { Pr1[r][c] = P1[r][c] - O1[r]; Pr2[r][c] = P2[r][c] - O2[r]; }
  }
  float M[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) M[r][c] = (Pr2[r][0] * Pr1[c][0] + Pr2[r][1] * Pr1[c][1]) + Pr2[r][2] * Pr1[c][2];
  const float N11 = M[0][0] + M[1][1] + M[2][2], N12 = M[1][2] - M[2][1], N13 = M[2][0] - M[0][2], N14 = M[0][1] - M[1][0];
  const float N22 = M[0][0] - M[1][1] - M[2][2], N23 = M[0][1] + M[1][0], N24 = M[2][0] + M[0][2];
  const float N33 = -M[0][0] + M[1][1] - M[2][2], N34 = M[1][2] + M[2][1], N44 = -M[0][0] - M[1][1] + M[2][2];
  double A[4][4] = {{N11, N12, N13, N14}, {N12, N22, N23, N24}, {N13, N23, N33, N34}, {N14, N24, N34, N44}}, V[4][4];
  jacobi4_dev(A, V);
  int mi = 0;
#pragma unroll
  for (int k = 1; k < 4; k++) if (A[k][k] > A[mi][mi]) mi = k;
  double q0 = V[0][0], vx = V[1][0], vy = V[2][0], vz = V[3][0];
#pragma unroll
  for (int k = 1; k < 4; k++) if (mi == k) { q0 = V[0][k]; vx = V[1][k]; vy = V[2][k]; vz = V[3][k]; }
  const double vn = sqrt(vx * vx + vy * vy + vz * vz);
  const double ang = atan2(vn, q0);
  float R[3][3];
  {
    double ax = 0, ay = 0, az = 0;
    if (vn > 0) { ax = vx / vn; ay = vy / vn; az = vz / vn; }
    const double w = cos(ang), sh = sin(ang), x = sh * ax, y = sh * ay, z = sh * az;
    R[0][0] = (float)(1 - 2 * (y * y + z * z)); R[0][1] = (float)(2 * (x * y - z * w)); R[0][2] = (float)(2 * (x * z + y * w));
    R[1][0] = (float)(2 * (x * y + z * w)); R[1][1] = (float)(1 - 2 * (x * x + z * z)); R[1][2] = (float)(2 * (y * z - x * w));
    R[2][0] = (float)(2 * (x * z - y * w)); R[2][1] = (float)(2 * (y * z + x * w)); R[2][2] = (float)(1 - 2 * (x * x + y * y));
  }
  float P3[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) P3[r][c] = (R[r][0] * Pr2[0][c] + R[r][1] * Pr2[1][c]) + R[r][2] * Pr2[2][c];
  float sc = 1.0f;
  if (!fix_scale) {
    float nom = 0, den = 0;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int r = 0; r < 3; r++) { nom += Pr1[r][c] * P3[r][c]; den += P3[r][c] * P3[r][c]; }
    sc = (float)((double)nom / (double)den);
  }
  float t[3], sR[3][3], sRi[3][3], ti[3];
#pragma unroll
  for (int r = 0; r < 3; r++) t[r] = O1[r] - ((sc * R[r][0]) * O2[0] + (sc * R[r][1]) * O2[1] + (sc * R[r][2]) * O2[2]);
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) { sR[r][c] = sc * R[r][c]; sRi[r][c] = (float)((1.0 / sc) * R[c][r]); }
#pragma unroll
  for (int r = 0; r < 3; r++) ti[r] = (-sRi[r][0] * t[0] + -sRi[r][1] * t[1]) + -sRi[r][2] * t[2];
  if (lane == 0) {
    float* out = T12 + 13 * (size_t)h;
    out[0] = sc;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) out[1 + 3 * r + c] = R[r][c];
#pragma unroll
    for (int r = 0; r < 3; r++) out[10 + r] = t[r];
  }
  const float fx1 = K[0], fy1 = K[1], cx1 = K[2], cy1 = K[3], fx2 = K[4], fy2 = K[5], cx2 = K[6], cy2 = K[7];
  int nin = 0;
  for (int base = 0; base < N; base += 64) {
    const int i = base + lane;
    bool in = false;
    if (i < N) {
      const float X1[3] = {P1c[3 * i], P1c[3 * i + 1], P1c[3 * i + 2]}, X2[3] = {P2c[3 * i], P2c[3 * i + 1], P2c[3 * i + 2]};
      float a[3], b[3];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        a[r] = ((sR[r][0] * X2[0] + sR[r][1] * X2[1]) + sR[r][2] * X2[2]) + t[r];
        b[r] = ((sRi[r][0] * X1[0] + sRi[r][1] * X1[1]) + sRi[r][2] * X1[2]) + ti[r];
      }
      const float p1x = fx1 * X1[0] / X1[2] + cx1, p1y = fy1 * X1[1] / X1[2] + cy1;   // FromCameraToImage
      const float p2x = fx2 * X2[0] / X2[2] + cx2, p2y = fy2 * X2[1] / X2[2] + cy2;
      const float u1 = fx1 * a[0] / a[2] + cx1, v1 = fy1 * a[1] / a[2] + cy1;
      const float u2 = fx2 * b[0] / b[2] + cx2, v2 = fy2 * b[1] / b[2] + cy2;
      const float d1x = p1x - u1, d1y = p1y - v1, d2x = u2 - p2x, d2y = v2 - p2y;
      const float err1 = d1x * d1x + d1y * d1y, err2 = d2x * d2x + d2y * d2y;
      in = err1 < max_err1[i] && err2 < max_err2[i];
      mask[(size_t)h * N + i] = in ? 1 : 0;
    }
    nin += __popcll(__ballot(in));
  }
  if (lane == 0) n_inliers[h] = nin;
}

void ba_launch_sim3_hypotheses(hipStream_t s, const float* P1c, const float* P2c, const float* e1, const float* e2, int N,
                               const float* K, const int32_t* triples, int H, int fix_scale, float* T12, int32_t* nin, uint8_t* mask) {
  if (H > 0) hipLaunchKernelGGL(k_sim3_hypotheses, dim3(H), dim3(64), 0, s, P1c, P2c, e1, e2, N, K, triples, H, fix_scale, T12, nin, mask);
}

// ---------------------------------------------------------------------------------- essential graph
// Optimizer::OptimizeEssentialGraph numerics (reference src/Optimizer.cc:1389-1652): VertexSim3Expmap + EdgeSim3
// (types_seven_dof_expmap.h:93-117), numeric Jacobians as g2o takes them (base_binary_edge.hpp:131-205).
__device__ void sim3_log(const Sim3d& S, double* res) {   // g2o Sim3::log, sim3.h:128-197
  const double sigma = log(S.s);
  double R[9];
  quat_to_R(S.q, R);
  const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  double omega[3];
  const double eps = 0.00001;
  double A, B, C;
  if (fabs(sigma) < eps) {
    C = 1;
    if (d > 1 - eps) { for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i]; A = 1. / 2.; B = 1. / 6.; }
    else {
      const double theta = acos(d), theta2 = theta * theta;
      for (int i = 0; i < 3; i++) omega[i] = theta / (2 * sqrt(1 - d * d)) * dR[i];
      A = (1 - cos(theta)) / theta2; B = (theta - sin(theta)) / (theta2 * theta);
    }
  } else {
    C = (S.s - 1) / sigma;
    if (d > 1 - eps) {
      const double sigma2 = sigma * sigma;
      for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i];
      A = ((sigma - 1) * S.s + 1) / sigma2; B = ((0.5 * sigma2 - sigma + 1) * S.s) / (sigma2 * sigma);
    } else {
      const double theta = acos(d);
      for (int i = 0; i < 3; i++) omega[i] = theta / (2 * sqrt(1 - d * d)) * dR[i];
      const double theta2 = theta * theta, a = S.s * sin(theta), b = S.s * cos(theta), c = theta2 + sigma * sigma;
      A = (a * sigma + (1 - b) * theta) / (theta * c);
      B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / theta2;
    }
  }
  const double O[9] = {0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0};
  double W[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const double o2 = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
      W[3 * i + j] = A * O[3 * i + j] + B * o2 + C * (i == j ? 1.0 : 0.0);
    }
  const double c00 = W[4] * W[8] - W[5] * W[7], c01 = W[5] * W[6] - W[3] * W[8], c02 = W[3] * W[7] - W[4] * W[6];
  const double det = W[0] * c00 + W[1] * c01 + W[2] * c02;
  const double inv[9] = {c00, W[2] * W[7] - W[1] * W[8], W[1] * W[5] - W[2] * W[4],
                         c01, W[0] * W[8] - W[2] * W[6], W[2] * W[3] - W[0] * W[5],
                         c02, W[1] * W[6] - W[0] * W[7], W[0] * W[4] - W[1] * W[3]};
  for (int i = 0; i < 3; i++) res[i] = omega[i];
  for (int i = 0; i < 3; i++) res[3 + i] = (inv[3 * i] * S.t[0] + inv[3 * i + 1] * S.t[1] + inv[3 * i + 2] * S.t[2]) / det;
  res[6] = sigma;
}
__device__ __forceinline__ void sim3_load(const double* p, Sim3d& S) {
  S.q[0] = p[0]; S.q[1] = p[1]; S.q[2] = p[2]; S.q[3] = p[3]; S.t[0] = p[4]; S.t[1] = p[5]; S.t[2] = p[6]; S.s = p[7];
}
__device__ void pg_edge_error(const Sim3d& C, const Sim3d& Si, const Sim3d& Sj, double* e) {   // log(C * v1 * v2^-1)
  Sim3d Sji, t1, t2;
  sim3_inv(Sj, Sji);
  sim3_mul(C, Si, t1);
  sim3_mul(t1, Sji, t2);
  sim3_log(t2, e);
}
// thread per edge: error, chi2 partial; JAC: the two 7x7 numeric Jacobians (14 columns x 2 perturbed error evaluations)
template <bool JAC>
__global__ void __launch_bounds__(256) k_pg_edge(PgView G) {
  __shared__ double red[256];
  const int k = blockIdx.x * 256 + threadIdx.x;
  double chi = 0;
  if (k < G.E) {
    const int vi = G.ev[2 * k], vj = G.ev[2 * k + 1];
    Sim3d C, Si, Sj;
    sim3_load(G.emeas + 8 * (size_t)k, C); sim3_load(G.S + 8 * (size_t)vi, Si); sim3_load(G.S + 8 * (size_t)vj, Sj);
    double e[7];
    pg_edge_error(C, Si, Sj, e);
    for (int a = 0; a < 7; a++) { G.e_err[7 * (size_t)k + a] = e[a]; chi += e[a] * e[a]; }
    if (JAC) {
      for (int side = 0; side < 2; side++) {
        const int v = side ? vj : vi;
        double* J = G.e_J + (size_t)k * 98 + 49 * side;
        if (G.vidx[v] < 0) { for (int a = 0; a < 49; a++) J[a] = 0; continue; }
        const Sim3d& X = side ? Sj : Si;
        for (int d = 0; d < 7; d++) {
          double e1[7], e2[7];
          for (int sgn = 0; sgn < 2; sgn++) {
            double u[7] = {0, 0, 0, 0, 0, 0, 0};
            u[d] = sgn ? -1e-9 : 1e-9;
            if (G.fix_scale) u[6] = 0;
            Sim3d Ex, Xp;
            sim3_exp(u, Ex);
            sim3_mul(Ex, X, Xp);
            pg_edge_error(C, side ? Si : Xp, side ? Xp : Sj, sgn ? e2 : e1);
          }
          for (int a = 0; a < 7; a++) J[7 * a + d] = (1.0 / (2 * 1e-9)) * (e1[a] - e2[a]);
        }
      }
    }
  }
  red[threadIdx.x] = chi;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) G.partial[blockIdx.x] = red[0];
}
// 64 threads per non-zero 7x7 block (a >= b): H_ab = sum over its contributions J_a^T J_b (+ lambda on the diagonal),
// fixed order, written into the tile-space matrix
__global__ void __launch_bounds__(256) k_pg_blocks(PgView G, BaView T) {
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), t = threadIdx.x & 63;
  if (blk >= G.nblk || t >= 49) return;
  const int r = t / 7, c = t % 7;
  const int a = G.blk_a[blk], b = G.blk_b[blk];
  double acc = 0;
  for (int i = G.blk_start[blk]; i < G.blk_start[blk + 1]; i++) {
    const int w = G.blk_contrib[i];
    const double* Ja = G.e_J + (size_t)(w >> 2) * 98 + 49 * ((w >> 1) & 1);
    const double* Jb = G.e_J + (size_t)(w >> 2) * 98 + 49 * (w & 1);
    double h = 0;
#pragma unroll
    for (int k = 0; k < 7; k++) h += Ja[7 * k + r] * Jb[7 * k + c];
    acc += h;
  }
  if (a == b && r == c) acc += *T.lambda;
  const int ra = (a / T.per_tile) * 64 + (a % T.per_tile) * T.dof, rb = (b / T.per_tile) * 64 + (b % T.per_tile) * T.dof;
  T.S[(size_t)(ra + r) * T.ldS + rb + c] = acc;
}
// thread per free vertex: b_v = - sum J_v^T e  -> compact copy (computeScale) and the augmented rhs row
__global__ void __launch_bounds__(256) k_pg_rhs(PgView G, BaView T) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= G.nfree) return;
  double b[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = G.v_start[p]; i < G.v_start[p + 1]; i++) {
    const int w = G.v_contrib[i];
    const double* J = G.e_J + (size_t)(w >> 1) * 98 + 49 * (w & 1);
    const double* e = G.e_err + 7 * (size_t)(w >> 1);
    for (int r = 0; r < 7; r++) {
      double g = 0;
#pragma unroll
      for (int a = 0; a < 7; a++) g += J[7 * a + r] * (-e[a]);
      b[r] += g;
    }
  }
  const int row = (p / T.per_tile) * 64 + (p % T.per_tile) * T.dof;
  for (int r = 0; r < 7; r++) { G.bp[7 * (size_t)p + r] = b[r]; T.S[(size_t)T.n_pad * T.ldS + row + r] = b[r]; }
  if (p == 0) T.S[(size_t)T.n_pad * T.ldS + T.n_pad] = 1e200;
}
// thread per free vertex: oplus (S <- Sim3(x) * S) and the computeScale partial sum x^T (lambda x + b)
__global__ void __launch_bounds__(256) k_pg_update(PgView G, BaView T) {
  __shared__ double red[256];
  const int p = blockIdx.x * 256 + threadIdx.x;
  double sc = 0;
  if (p < G.nfree) {
    const double lambda = *T.lambda;
    double u[7];
    for (int r = 0; r < 7; r++) { u[r] = T.x[7 * (size_t)p + r]; sc += u[r] * (lambda * u[r] + G.bp[7 * (size_t)p + r]); }
    if (G.fix_scale) u[6] = 0;
    double* Sp = G.S + 8 * (size_t)G.free_v[p];
    Sim3d X, Ex, Xn;
    sim3_load(Sp, X);
    sim3_exp(u, Ex);
    sim3_mul(Ex, X, Xn);
    Sp[0] = Xn.q[0]; Sp[1] = Xn.q[1]; Sp[2] = Xn.q[2]; Sp[3] = Xn.q[3]; Sp[4] = Xn.t[0]; Sp[5] = Xn.t[1]; Sp[6] = Xn.t[2]; Sp[7] = Xn.s;
  }
  red[threadIdx.x] = sc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) G.partial[blockIdx.x] = red[0];
}

// ------------------------------------------------------------------------------------- launchers
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

void ba_launch_edge_eval(hipStream_t s, const BaView& V, bool jac, const BaPublish& pub) {
  const int nb = cdiv(V.E, 256);
  if (jac) hipLaunchKernelGGL(k_edge_eval<true>, dim3(nb), dim3(256), 0, s, V, pub);
  else hipLaunchKernelGGL(k_edge_eval<false>, dim3(nb), dim3(256), 0, s, V, pub);
}
void ba_launch_accum(hipStream_t s, const BaView& V, const double* spec, const BaPublish* max_pub) {
  const int nb_pose = V.nfree > 0 ? (V.schur_wide ? V.nfree : cdiv(V.nfree, 4)) : 0;   // (wide: a workgroup per camera)
  const int nb_point = cdiv(V.L, 256);
  BaPublish none;
  std::memset(&none, 0, sizeof(none));
  hipLaunchKernelGGL(k_accum, dim3(nb_pose + nb_point + (V.nfree > 0 ? V.n_nz : 0)), dim3(256), 0, s, V, nb_pose, nb_point, spec,
                     max_pub ? *max_pub : none, max_pub ? 1 : 0);
}
void ba_launch_max_diag(hipStream_t s, const BaView& V, const BaPublish& pub) {
  hipLaunchKernelGGL(k_max_diag, dim3(cdiv(std::max(3 * V.L, 6 * V.nfree), 256)), dim3(256), 0, s, V, pub);
}
static void launch_schur_kernel(hipStream_t s, const BaView& V, int* fail_reset) {
  if (V.n_slc > 0) {     // the landmark-chunk form (built by dvm_ba_set_problem only under DVM_BA_SCHUR_LM=1: measured slower)
    const int nb_rhs = V.schur_wide ? V.nfree : cdiv(V.nfree, 4);
    const size_t lds = sizeof(double) * ((size_t)kSchurLmRows * 18 + kSchurLmLandmarks * 9) + sizeof(int32_t) * (kSchurLmPairs + 2 * (kSchurLmRuns + 1));
    static bool raised = false;
    if (!raised) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_schur_lm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); raised = true; }
    hipLaunchKernelGGL(k_schur_lm, dim3(nb_rhs + V.n_slc), dim3(256), lds, s, V, nb_rhs, fail_reset);
    hipLaunchKernelGGL(k_schur_reduce, dim3(cdiv(36 * V.nblk, 256)), dim3(256), 0, s, V);
    return;
  }
  const int nb_blk = V.nblk, nb_chunk = cdiv(nb_blk, 8);   // one workgroup per 6x6 block
  const int nw = V.schur_wide ? kSchurWavesWide : kSchurWaves;
  const int nb_rhs = ((V.schur_wide ? V.nfree : cdiv(V.nfree, nw)) + 7) & ~7;   // (wide: a workgroup per camera) a multiple of 8 keeps the XCD phase of the block workgroups
  if (V.schur_wide) hipLaunchKernelGGL(k_schur<kSchurWavesWide>, dim3(nb_rhs + 8 * nb_chunk + V.nkept), dim3(64 * kSchurWavesWide), 0, s, V, nb_blk, nb_chunk, nb_rhs, fail_reset);
  else hipLaunchKernelGGL(k_schur<kSchurWaves>, dim3(nb_rhs + 8 * nb_chunk + V.nkept), dim3(64 * kSchurWaves), 0, s, V, nb_blk, nb_chunk, nb_rhs, fail_reset);
}
void ba_launch_schur(hipStream_t s, const BaView& V, int* d_fail) {
  const int nb_dinv = cdiv(V.L, 256);
  hipLaunchKernelGGL(k_trial_prologue, dim3(nb_dinv), dim3(256), 0, s, V, d_fail);
  if (V.nfree == 0) return;
  launch_schur_kernel(s, V, nullptr);
}
void ba_launch_schur_speculative(hipStream_t s, const BaView& V, int* d_fail) {
  if (V.nfree == 0 || !V.lambda) return;
  launch_schur_kernel(s, V, d_fail);
}
__global__ void __launch_bounds__(256) k_clear_tiles(BaView V) { clear_tile(V, blockIdx.x); }
void ba_launch_clear_tiles(hipStream_t s, const BaView& V) {
  if (V.nfree > 0 && V.n_nz > 0) hipLaunchKernelGGL(k_clear_tiles, dim3(V.n_nz), dim3(256), 0, s, V);
}
constexpr int kDiagLevelMaxWGs = 256;    // k_chol_trsm_update<true>: 86 KB of LDS, one workgroup per CU
constexpr int kFusedLevelMaxWGs = 512;   // 2 workgroups per CU on 256 CUs (3 fit: 41 KB of LDS each); the leaf level of the BASELINE problem (536) measured the same fused or not
void ba_launch_cholesky_solve(hipStream_t s, const BaView& V, int* d_fail, int solve_seq) {
  if (V.nfree == 0) return;
  const int n1 = V.n_pad + 1;
  if (V.flow && V.flow_tasks && V.flow_flags) {     // the whole solve as one persistent launch of tile tasks (k_chol_flow)
    FlowArgs A;
    A.S = V.S; A.ldS = V.ldS; A.n1 = n1; A.n_pad = V.n_pad; A.Linv_all = V.Linv;
    A.tasks = V.flow_tasks; A.fc = V.flow_contrib; A.colinfo = V.flow_col;
    A.n_factor = V.n_flow_tasks; A.n_back = V.h_level_off[V.nlevels - 1];
    A.flags = V.flow_flags; A.nstrips = V.n_strips_total; A.ntiles = V.n_tiles_total;
    // test switch: the strips publish a sequence number nobody waits for, so every wait on them gives up (bounded), the trial is marked
    // (fail = 2) and the caller repeats it with one launch per phase (tests/test_gpu_ba_flow.py)
    static const bool break_flow = std::getenv("DVM_BA_DEBUG_BREAK_FLOW") != nullptr;
    A.gen = solve_seq; A.gen_pub = break_flow ? -1 : solve_seq; A.fail = d_fail;
    A.cols = V.cols; A.xrow = V.xrow; A.x = V.x; A.colstrip_off = V.colstrip_off; A.colstrips = V.colstrips; A.colstrip_id = V.colstrip_id;
    A.nfree = V.nfree; A.per_tile = V.per_tile; A.dof = V.dof; A.ncamt = V.ncamt; A.nkept = V.nkept; A.kept_list = V.kept_list;
    const int wgs = std::max(1, std::min(A.n_factor + A.n_back, V.flow_wgs > 0 ? V.flow_wgs : 256));
    hipLaunchKernelGGL(k_chol_flow, dim3(wgs), dim3(256), 0, s, A);
    return;
  }
  // test switch: the slices publish a sequence number nobody waits for, so every wait times out and the caller's retry path
  // (one launch per phase) has to produce the result (tests/test_gpu_ba.py)
  static const bool break_handoff = std::getenv("DVM_BA_DEBUG_BREAK_HANDOFF") != nullptr;
  const bool diag_in_level = V.diag_in_level != 0;   // (A/B switch DVM_BA_NO_DIAG_IN_LEVEL, read by dvm_ba_set_problem)
  double* tag = V.xrow;   // the first launch that solves strips also re-arms the back substitution's hand-off slots (no strips: no waits)
  int root_level = V.nlevels - 1;          // the last level that is launched (see the break below), where n_root_raw applies
  if (root_level >= 1 && V.h_level_off[root_level + 1] - V.h_level_off[root_level] == 1 && V.h_strip_off[root_level + 1] == V.h_strip_off[root_level] &&
      V.h_tgt_off[root_level + 1] == V.h_tgt_off[root_level]) root_level--;
  // the top pair of the elimination order (two single-column levels: root_level - 1 and root_level) goes to k_chol_pair
  const bool pair = V.pair_ok && root_level >= 1;
  for (int h = 0; h < V.nlevels; h++) {
    if (pair && h == root_level - 1) break;
    const int nc = V.h_level_off[h + 1] - V.h_level_off[h], ns = V.h_strip_off[h + 1] - V.h_strip_off[h];
    const int nt = V.h_tgt_off[h + 1] - V.h_tgt_off[h];
    // the root of the elimination tree is the tile of the augmented rhs row: its "factorisation" (one scalar) is never
    // used -- row n_pad already holds y = L^-1 b once the last camera level is done -- so that level is not launched
    if (h == V.nlevels - 1 && nc == 1 && ns == 0 && nt == 0) break;
    const bool raw_root = ns > 0 && nt == 0 && V.n_root_raw > 0 && h == root_level;   // the back substitution solves these rhs strips itself
    // the whole level -- factorisation of the diagonal tiles included -- as one launch (k_chol_trsm_update<true>): one workgroup
    // per CU, so only while slices + quadrants fit the chip at once; else the slices alone that way and the update behind them
    const bool all = nt > 0 && 4 * (ns + nt) <= kDiagLevelMaxWGs;
    if (diag_in_level && !raw_root && ns > 0 && V.contrib_strip && V.strip_flags && 4 * ns <= kDiagLevelMaxWGs) {
      hipLaunchKernelGGL(k_chol_trsm_update<true>, dim3(4 * (ns + (all ? nt : 0))), dim3(256), 0, s, V.S, V.ldS, n1, V.Linv,
                         V.strips + 2 * (size_t)V.h_strip_off[h], V.h_strip_off[h], 4 * ns, V.targets + 4 * (size_t)V.h_tgt_off[h], V.contrib,
                         V.contrib_strip, V.strip_flags, solve_seq, break_handoff ? -1 : solve_seq, d_fail, tag, V.n_pad);
      tag = nullptr;
      if (nt > 0 && !all) hipLaunchKernelGGL(k_chol_update, dim3(4 * nt), dim3(256), 0, s, V.S, V.ldS, n1, V.targets + 4 * (size_t)V.h_tgt_off[h], V.contrib);
      continue;
    }
    hipLaunchKernelGGL(k_chol_diag, dim3(nc), dim3(256), 0, s, V.S, V.ldS, n1, V.cols + V.h_level_off[h], d_fail, V.Linv);
    // solve + update of the level as ONE launch when every workgroup of it is resident at once (a quadrant then never waits
    // for a slice that has no compute unit to run on): 41 KB of LDS per workgroup = 3 per CU
    if (ns > 0 && nt > 0 && 4 * (ns + nt) <= kFusedLevelMaxWGs && V.contrib_strip && V.strip_flags) {
      hipLaunchKernelGGL(k_chol_trsm_update<false>, dim3(4 * (ns + nt)), dim3(256), 0, s, V.S, V.ldS, n1, V.Linv, V.strips + 2 * (size_t)V.h_strip_off[h],
                         V.h_strip_off[h], 4 * ns, V.targets + 4 * (size_t)V.h_tgt_off[h], V.contrib, V.contrib_strip, V.strip_flags, solve_seq,
                         break_handoff ? -1 : solve_seq, d_fail, tag, V.n_pad);
      tag = nullptr;
      continue;
    }
    if (raw_root) continue;
    if (ns > 0) { hipLaunchKernelGGL(k_chol_trsm, dim3(4 * ns), dim3(256), 0, s, V.S, V.ldS, n1, V.Linv, V.strips + 2 * (size_t)V.h_strip_off[h], tag, V.n_pad); tag = nullptr; }
    if (nt > 0) hipLaunchKernelGGL(k_chol_update, dim3(4 * nt), dim3(256), 0, s, V.S, V.ldS, n1, V.targets + 4 * (size_t)V.h_tgt_off[h], V.contrib);
  }
  // y = L^-1 b is row n_pad of S (the augmented rhs row): the back substitution reads it in place; one launch for all
  // camera tile columns (levels 0 .. nlevels - 2; level nlevels - 1 is the rhs tile alone: not an unknown)
  int ncols = V.h_level_off[V.nlevels - 1];
  int n_raw = V.n_root_raw;
  if (pair) {
    // (the slots of xrow that the remaining columns wait on were tagged by the first strip launch above; these two are written for good)
    hipLaunchKernelGGL(k_chol_pair, dim3(1), dim3(256), 0, s, V.S, V.ldS, V.n_pad, V.pair_a, V.pair_b, V.nfree, V.per_tile, V.dof, V.xrow, V.x, d_fail, V.ncamt, V.nkept, V.kept_list);
    ncols -= 2;          // `cols` lists the levels in order: the pair's columns are its last two entries
    n_raw = 0;
  }
  if (ncols > 0)
    hipLaunchKernelGGL(k_chol_backsolve, dim3(ncols), dim3(256), 0, s, V.S, V.ldS, V.n_pad, V.nfree, V.per_tile, V.dof, V.cols, ncols,
                       V.S + (size_t)V.n_pad * V.ldS, V.xrow, V.x, V.Linv, V.colstrip_off, V.colstrips, reinterpret_cast<int32_t*>(V.ytmp),
                       solve_seq, d_fail, n_raw, V.ncamt, V.nkept, V.kept_list);
}
void ba_launch_backsub_update(hipStream_t s, const BaView& V, const BaPublish& pub, const int* d_fail) {
  const int nb_pose = cdiv(std::max(V.nfree, 1), 256);
  hipLaunchKernelGGL(k_point_backsub, dim3(nb_pose + cdiv(8 * V.L, 256)), dim3(256), 0, s, V, pub, nb_pose, d_fail);
}
__global__ void __launch_bounds__(256) k_pack_tiles(BaView V, double* __restrict__ buf, int unpack) {
  const int ti = V.nz_tiles[2 * blockIdx.x], tj = V.nz_tiles[2 * blockIdx.x + 1];
  double* base = V.S + (size_t)ti * 64 * V.ldS + tj * 64;
  double* b = buf + (size_t)blockIdx.x * 4096;
  for (int i = threadIdx.x; i < 64 * 32; i += 256) {
    const int r = i >> 5, c2 = i & 31;
    double2* g = reinterpret_cast<double2*>(base + (size_t)r * V.ldS) + c2;
    double2* p = reinterpret_cast<double2*>(b + r * 64) + c2;
    if (unpack) *g = *p; else *p = *g;
  }
}
__global__ void __launch_bounds__(256) k_hpp_diag(BaView V, double* __restrict__ buf) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < V.nfree * 6) buf[i] = V.Hpp[36 * (size_t)(i / 6) + 7 * (i % 6)];
}
__global__ void __launch_bounds__(256) k_max_diag_sharded(BaView V, const double* __restrict__ hpp_diag, BaPublish pub) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double m = 0;
  if (i < V.nfree * 6) m = fabs(hpp_diag[i]);
  if (i < V.L * 3 && V.pt_start[i / 3 + 1] > V.pt_start[i / 3]) m = fmax(m, fabs(V.Hll[9 * (size_t)(i / 3) + 4 * (i % 3)]));
  block_reduce_publish<true>(m, V.partial2, pub);
}
__global__ void __launch_bounds__(256) k_points_exchange(BaView V, double* __restrict__ buf, int scatter) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= V.L) return;
  const bool own = l % V.shard_world == V.shard_rank;   // by index, not by "has local edges": a landmark nobody observes keeps its value
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (scatter) V.points[3 * (size_t)l + k] = buf[3 * (size_t)l + k];
    else buf[3 * (size_t)l + k] = own ? V.points[3 * (size_t)l + k] : 0.0;
  }
}
void ba_launch_pack_tiles(hipStream_t s, const BaView& V, double* buf, bool unpack) {
  hipLaunchKernelGGL(k_pack_tiles, dim3(V.n_nz), dim3(256), 0, s, V, buf, unpack ? 1 : 0);
}
void ba_launch_hpp_diag(hipStream_t s, const BaView& V, double* buf) {
  hipLaunchKernelGGL(k_hpp_diag, dim3(cdiv(6 * V.nfree, 256)), dim3(256), 0, s, V, buf);
}
void ba_launch_max_diag_sharded(hipStream_t s, const BaView& V, const double* hpp_diag, const BaPublish& pub) {
  hipLaunchKernelGGL(k_max_diag_sharded, dim3(cdiv(std::max(3 * V.L, 6 * V.nfree), 256)), dim3(256), 0, s, V, hpp_diag, pub);
}
void ba_launch_points_exchange(hipStream_t s, const BaView& V, double* buf, bool scatter) {
  hipLaunchKernelGGL(k_points_exchange, dim3(cdiv(V.L, 256)), dim3(256), 0, s, V, buf, scatter ? 1 : 0);
}
void ba_launch_edge_depth(hipStream_t s, const BaView& V, uint8_t* d_out) {
  hipLaunchKernelGGL(k_edge_depth, dim3(cdiv(V.E, 256)), dim3(256), 0, s, V, d_out);
}

void pg_launch_edge_eval(hipStream_t s, const PgView& G, bool jac, double* d_scalars, int slot) {
  const int nb = cdiv(G.E, 256);
  if (jac) hipLaunchKernelGGL(k_pg_edge<true>, dim3(nb), dim3(256), 0, s, G);
  else hipLaunchKernelGGL(k_pg_edge<false>, dim3(nb), dim3(256), 0, s, G);
  hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(256), 0, s, G.partial, nb, d_scalars, slot);
}
void pg_launch_build(hipStream_t s, const PgView& G, const BaView& T) {
  hipLaunchKernelGGL(k_zero_tiles, dim3(T.n_nz), dim3(256), 0, s, T.S, T.ldS, T.nz_tiles);
  hipLaunchKernelGGL(k_pg_blocks, dim3(cdiv(G.nblk, 4)), dim3(256), 0, s, G, T);
  hipLaunchKernelGGL(k_pg_rhs, dim3(cdiv(G.nfree, 256)), dim3(256), 0, s, G, T);
  hipLaunchKernelGGL(k_pad_identity, dim3(cdiv(T.n_pad, 256)), dim3(256), 0, s, T);
}
void pg_launch_update(hipStream_t s, const PgView& G, const BaView& T, double* d_scalars, int slot_scale) {
  const int nb = cdiv(G.nfree, 256);
  hipLaunchKernelGGL(k_pg_update, dim3(nb), dim3(256), 0, s, G, T);
  hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(256), 0, s, G.partial, nb, d_scalars, slot_scale);
}

}  // namespace dvm
