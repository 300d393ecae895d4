// oracle/ba_oracle.cpp -- CPU restatement of the reference bundle-adjustment numerics (double).
//
// TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  PARITY UNPINNED: the reference's optimizer is
// vendored g2o + un-vendored Eigen3 (absent here, so the reference cannot be compiled) and no
// reference test holds a BA result.  This file restates the recipe; it is pinned against an
// independent numpy/scipy Gauss-Newton/LM implementation in tests/test_oracle_ba.py.
//
// Reference files followed (under /root/reference/src/slam_system/orb_slam3/):
//   src/Optimizer.cc:55-356      BundleAdjustment        (graph: SE3 vertices, XYZ vertices marginalised,
//   src/Optimizer.cc:1030-1387   LocalBundleAdjustment    EdgeSE3ProjectXYZ, info = I*invSigma2, Huber)
//   src/Optimizer.cc:744-1028    PoseOptimization        -> orc_pose_optimize
//   src/OptimizableTypes.cpp:51-63,136-155 / include/OptimizableTypes.h:98-103   error + Jacobians
//   src/CameraModels/Pinhole.cpp:38-44,69-79              project / projectJac
//   Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:59-188          LM driver
//   Thirdparty/g2o/g2o/core/block_solver.hpp:354-486,502-604                     buildSystem/Schur/solve
//   Thirdparty/g2o/g2o/core/base_binary_edge.hpp:55-116, base_unary_edge.hpp     constructQuadraticForm
//   Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:68-81                         Huber
//   Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:61-110,349-427                  errors, chi2, optimize, update
//   Thirdparty/g2o/g2o/types/se3quat.h:210-266, types_six_dof_expmap.h:71-74     SE3Quat exp/map/oplus
// The reduced camera system is solved by an envelope (skyline) Cholesky in natural pose order: a
// sparse direct solve like the reference's Eigen SimplicialLDLT (same solution up to round-off).
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "f64_spec.h"

namespace {

struct Pose { double t[3]; double q[4]; };  // q = (x,y,z,w)

void quat_to_R(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
               tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
void R_to_quat(const double R[9], double q[4]) {  // Eigen's quaternion-from-matrix
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
}
void quat_normalize(double q[4]) {  // SE3Quat::normalizeRotation, se3quat.h:261-266
  if (q[3] < 0) for (int i = 0; i < 4; i++) q[i] = -q[i];
  double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}
void quat_mul(const double a[4], const double b[4], double o[4]) {
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
void mat3_vec(const double R[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; i++) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
// T <- exp(update) * T, update = (omega, upsilon); se3quat.h:212-240 and operator*
void pose_oplus(Pose& T, const double u[6]) {
  const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
  const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double O2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
  double R[9], V[9];
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) R[i] = I[i] + O[i] + O2[i];
    std::memcpy(V, R, sizeof(R));
  } else {
    // se3quat.h:229-234 calls libm's sin / cos / pow(theta, 3); no two libms agree on those bits, so the oracle evaluates the shared
    // double-precision spec instead (oracle/f64_spec.h: fdlibm kernels, correctly rounded cube; within 1 ulp of glibc)
    const double sn = orc_spec::sin(theta), cs = orc_spec::cos(theta);
    const double a = sn / theta, b = (1 - cs) / (theta * theta), c = (theta - sn) / orc_spec::cube(theta);
    for (int i = 0; i < 9; i++) {
      R[i] = I[i] + a * O[i] + b * O2[i];
      V[i] = I[i] + b * O[i] + c * O2[i];
    }
  }
  double dq[4], dt[3];
  R_to_quat(R, dq);
  quat_normalize(dq);
  mat3_vec(V, up, dt);
  // (dq,dt) * (q,t): r = dq*q ; t = dt + dq*t ; normalize
  double Rd[9], rt[3], nq[4];
  quat_to_R(dq, Rd);
  mat3_vec(Rd, T.t, rt);
  quat_mul(dq, T.q, nq);
  for (int i = 0; i < 3; i++) T.t[i] = dt[i] + rt[i];
  std::memcpy(T.q, nq, sizeof(nq));
  quat_normalize(T.q);
}

struct Cam { double fx, fy, cx, cy, delta; };

// error, chi2, Jacobians of one projection edge (OptimizableTypes.cpp:136-155, Pinhole.cpp:38-79)
struct EdgeLin { double e[2]; double A[6]; double B[12]; double chi2; double Xc[3]; };
void linearize(const Pose& T, const double X[3], const double obs[2], double info, const Cam& c, EdgeLin& L, bool jac) {
  double R[9];
  quat_to_R(T.q, R);
  double Xc[3];
  mat3_vec(R, X, Xc);
  for (int i = 0; i < 3; i++) Xc[i] += T.t[i];
  std::memcpy(L.Xc, Xc, sizeof(Xc));
  const double x = Xc[0], y = Xc[1], z = Xc[2];
  L.e[0] = obs[0] - (c.fx * x / z + c.cx);
  L.e[1] = obs[1] - (c.fy * y / z + c.cy);
  L.chi2 = L.e[0] * info * L.e[0] + L.e[1] * info * L.e[1];
  if (!jac) return;
  // -projectJac
  const double J[6] = {-(c.fx / z), 0, c.fx * x / (z * z), 0, -(c.fy / z), c.fy * y / (z * z)};
  for (int r = 0; r < 2; r++)
    for (int k = 0; k < 3; k++) L.A[3 * r + k] = J[3 * r] * R[k] + J[3 * r + 1] * R[3 + k] + J[3 * r + 2] * R[6 + k];
  const double S[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
  for (int r = 0; r < 2; r++)
    for (int k = 0; k < 6; k++) L.B[6 * r + k] = J[3 * r] * S[k] + J[3 * r + 1] * S[6 + k] + J[3 * r + 2] * S[12 + k];
}
// RobustKernelHuber::robustify, robust_kernel_impl.cpp:68-81; delta <= 0 -> no kernel
inline void robustify(double e, double delta, double& rho0, double& rho1) {
  if (delta <= 0 || e <= delta * delta) { rho0 = e; rho1 = 1.; }
  else { double s = std::sqrt(e); rho0 = 2 * s * delta - delta * delta; rho1 = delta / s; }
}

bool inv3(const double M[9], double Inv[9]) {
  const double a = M[0], b = M[1], c = M[2], d = M[3], e = M[4], f = M[5], g = M[6], h = M[7], i = M[8];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  const double id = 1.0 / det;
  Inv[0] = (e * i - f * h) * id; Inv[1] = (c * h - b * i) * id; Inv[2] = (b * f - c * e) * id;
  Inv[3] = (f * g - d * i) * id; Inv[4] = (a * i - c * g) * id; Inv[5] = (c * d - a * f) * id;
  Inv[6] = (d * h - e * g) * id; Inv[7] = (b * g - a * h) * id; Inv[8] = (a * e - b * d) * id;
  return std::isfinite(id);
}

// Envelope (skyline) Cholesky of a symmetric matrix given by rows i -> columns [first[i], i].
struct Skyline {
  int n = 0;
  std::vector<int> first;
  std::vector<size_t> off;   // offset of row i's first stored entry
  std::vector<double> a;
  double& at(int i, int j) { return a[off[i] + (j - first[i])]; }
  void setup(int n_, const std::vector<int>& f) {
    n = n_; first = f; off.assign(n + 1, 0);
    for (int i = 0; i < n; i++) off[i + 1] = off[i] + (size_t)(i - first[i] + 1);
    a.assign(off[n], 0.0);
  }
  bool factor() {
    for (int i = 0; i < n; i++) {
      for (int j = first[i]; j <= i; j++) {
        const int k0 = std::max(first[i], first[j]);
        double s = at(i, j);
        const double* ri = &a[off[i] + (k0 - first[i])];
        const double* rj = &a[off[j] + (k0 - first[j])];
        for (int k = 0; k < j - k0; k++) s -= ri[k] * rj[k];
        if (j < i) at(i, j) = s / at(j, j);
        else {
          if (!(s > 0)) return false;
          at(i, i) = std::sqrt(s);
        }
      }
    }
    return true;
  }
  void solve(double* x) {  // in place
    for (int i = 0; i < n; i++) {
      double s = x[i];
      for (int j = first[i]; j < i; j++) s -= at(i, j) * x[j];
      x[i] = s / at(i, i);
    }
    for (int i = n - 1; i >= 0; i--) {
      x[i] /= at(i, i);
      for (int j = first[i]; j < i; j++) x[j] -= at(i, j) * x[i];
    }
  }
};

struct BA {
  int P, L, E;
  std::vector<Pose> poses;
  std::vector<uint8_t> fixed;
  std::vector<double> pts;
  std::vector<orc_ba_edge> edges;
  Cam cam;
  std::vector<int> pidx;  // pose -> free index or -1
  int nfree = 0;
  std::vector<uint8_t> pt_active;
  std::vector<int> lidx;  // landmark -> active index
  int nact = 0;
  // linear system
  std::vector<double> Hpp, bp, Hll, bl, W;  // W: per edge 6x3 (pose x point)
  std::vector<double> x;                     // [6*nfree + 3*nact]
  Skyline S;
  std::vector<std::vector<int>> pt_edges;
  std::vector<double> last_chi2;  // e->chi2() as g2o would report it: value at the LAST computeActiveErrors

  double robust_chi2() {
    double chi = 0;
    EdgeLin Lz;
    last_chi2.resize(E);
    for (int k = 0; k < E; k++) {
      const auto& e = edges[k];
      const double obs[2] = {e.u, e.v};
      linearize(poses[e.pose], &pts[3 * e.point], obs, e.inv_sigma2, cam, Lz, false);
      last_chi2[k] = Lz.chi2;
      double r0, r1;
      robustify(Lz.chi2, cam.delta, r0, r1);
      chi += r0;
    }
    return chi;
  }
  void build_system() {
    std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(bp.begin(), bp.end(), 0.0);
    std::fill(Hll.begin(), Hll.end(), 0.0); std::fill(bl.begin(), bl.end(), 0.0);
    EdgeLin Lz;
    for (int k = 0; k < E; k++) {
      const auto& e = edges[k];
      const double obs[2] = {e.u, e.v};
      linearize(poses[e.pose], &pts[3 * e.point], obs, e.inv_sigma2, cam, Lz, true);
      double r0, r1;
      robustify(Lz.chi2, cam.delta, r0, r1);
      const double w = r1 * e.inv_sigma2;                         // weightedOmega = rho' * Omega
      const double wr[2] = {-e.inv_sigma2 * Lz.e[0] * r1, -e.inv_sigma2 * Lz.e[1] * r1};  // omega_r * rho'
      const int li = lidx[e.point], pi = pidx[e.pose];
      double* hl = &Hll[9 * li];
      for (int a = 0; a < 3; a++) {
        bl[3 * li + a] += Lz.A[a] * wr[0] + Lz.A[3 + a] * wr[1];
        for (int b = 0; b < 3; b++) hl[3 * a + b] += w * (Lz.A[a] * Lz.A[b] + Lz.A[3 + a] * Lz.A[3 + b]);
      }
      double* Wk = &W[18 * k];
      if (pi >= 0) {
        double* hp = &Hpp[36 * pi];
        for (int a = 0; a < 6; a++) {
          bp[6 * pi + a] += Lz.B[a] * wr[0] + Lz.B[6 + a] * wr[1];
          for (int b = 0; b < 6; b++) hp[6 * a + b] += w * (Lz.B[a] * Lz.B[b] + Lz.B[6 + a] * Lz.B[6 + b]);
          for (int b = 0; b < 3; b++) Wk[3 * a + b] = w * (Lz.B[a] * Lz.A[b] + Lz.B[6 + a] * Lz.A[3 + b]);
        }
      } else std::fill(Wk, Wk + 18, 0.0);
    }
  }
  // Schur complement + solve + back-substitution with damping lambda; fills x.  block_solver.hpp:354-486
  bool solve(double lambda) {
    const int n = 6 * nfree;
    std::fill(S.a.begin(), S.a.end(), 0.0);
    std::vector<double> bs(bp.begin(), bp.begin() + n);
    for (int i = 0; i < nfree; i++)
      for (int a = 0; a < 6; a++)
        for (int b = 0; b <= a; b++) S.at(6 * i + a, 6 * i + b) = Hpp[36 * i + 6 * a + b] + (a == b ? lambda : 0.0);
    std::vector<double> Dinv((size_t)9 * nact);
    for (int l = 0; l < L; l++) {
      if (!pt_active[l]) continue;
      const int li = lidx[l];
      double D[9];
      std::memcpy(D, &Hll[9 * li], sizeof(D));
      D[0] += lambda; D[4] += lambda; D[8] += lambda;
      double* Di = &Dinv[9 * li];
      inv3(D, Di);
      double db[3];
      mat3_vec(Di, &bl[3 * li], db);
      const std::vector<int>& ev = pt_edges[l];
      for (int k1 : ev) {
        const int i1 = pidx[edges[k1].pose];
        if (i1 < 0) continue;
        const double* W1 = &W[18 * k1];
        double WD[18];
        for (int a = 0; a < 6; a++)
          for (int b = 0; b < 3; b++) WD[3 * a + b] = W1[3 * a] * Di[b] + W1[3 * a + 1] * Di[3 + b] + W1[3 * a + 2] * Di[6 + b];
        for (int a = 0; a < 6; a++) bs[6 * i1 + a] -= W1[3 * a] * db[0] + W1[3 * a + 1] * db[1] + W1[3 * a + 2] * db[2];
        for (int k2 : ev) {
          const int i2 = pidx[edges[k2].pose];
          if (i2 < 0 || i2 > i1) continue;  // lower triangle (i1 >= i2)
          if (i2 == i1 && k2 != k1) continue;  // one edge per (pose, point) pair by construction
          const double* W2 = &W[18 * k2];
          for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) {
              if (i1 == i2 && b > a) continue;
              S.at(6 * i1 + a, 6 * i2 + b) -= WD[3 * a] * W2[3 * b] + WD[3 * a + 1] * W2[3 * b + 1] + WD[3 * a + 2] * W2[3 * b + 2];
            }
        }
      }
    }
    if (!S.factor()) return false;
    S.solve(bs.data());
    std::copy(bs.begin(), bs.end(), x.begin());
    // xl = Dinv (bl - W^T xp)
    for (int l = 0; l < L; l++) {
      if (!pt_active[l]) continue;
      const int li = lidx[l];
      double c[3] = {bl[3 * li], bl[3 * li + 1], bl[3 * li + 2]};
      for (int k : pt_edges[l]) {
        const int i = pidx[edges[k].pose];
        if (i < 0) continue;
        const double* Wk = &W[18 * k];
        for (int b = 0; b < 3; b++)
          for (int a = 0; a < 6; a++) c[b] -= Wk[3 * a + b] * x[6 * i + a];
      }
      mat3_vec(&Dinv[9 * li], c, &x[n + 3 * li]);
    }
    return true;
  }
  void apply_update() {
    for (int p = 0; p < P; p++)
      if (pidx[p] >= 0) pose_oplus(poses[p], &x[6 * pidx[p]]);
    const int n = 6 * nfree;
    for (int l = 0; l < L; l++)
      if (pt_active[l]) for (int a = 0; a < 3; a++) pts[3 * l + a] += x[n + 3 * lidx[l] + a];
  }
};

}  // namespace

extern "C" {

static int ba_optimize_impl(double* poses, const uint8_t* fixed, int P, double* points, int L, const orc_ba_edge* edges, int E,
                            const orc_ba_camera* cam, int iterations, orc_ba_stats* st, double* edge_chi2, bool normalize_input);
int orc_ba_optimize(double* poses, const uint8_t* fixed, int P, double* points, int L, const orc_ba_edge* edges, int E,
                    const orc_ba_camera* cam, int iterations, orc_ba_stats* st, double* edge_chi2) {
  return ba_optimize_impl(poses, fixed, P, points, L, edges, E, cam, iterations, st, edge_chi2, true);
}
// a further optimizer.optimize(n) on the SAME graph (Optimizer.cc:1306-1311, :3474-3519: g2o keeps the vertices between the
// calls): the estimates are taken as the previous call left them, without SE3Quat's constructor normalising them again
int orc_ba_optimize_continue(double* poses, const uint8_t* fixed, int P, double* points, int L, const orc_ba_edge* edges, int E,
                             const orc_ba_camera* cam, int iterations, orc_ba_stats* st, double* edge_chi2) {
  return ba_optimize_impl(poses, fixed, P, points, L, edges, E, cam, iterations, st, edge_chi2, false);
}
static int ba_optimize_impl(double* poses, const uint8_t* fixed, int P, double* points, int L, const orc_ba_edge* edges, int E,
                            const orc_ba_camera* cam, int iterations, orc_ba_stats* st, double* edge_chi2, bool normalize_input) {
  BA ba;
  ba.P = P; ba.L = L; ba.E = E;
  ba.poses.resize(P);
  for (int p = 0; p < P; p++) {
    std::memcpy(ba.poses[p].t, poses + 7 * p, 3 * sizeof(double));
    std::memcpy(ba.poses[p].q, poses + 7 * p + 3, 4 * sizeof(double));
    if (normalize_input) quat_normalize(ba.poses[p].q);
  }
  ba.fixed.assign(fixed, fixed + P);
  ba.pts.assign(points, points + 3 * (size_t)L);
  ba.edges.assign(edges, edges + E);
  ba.cam = Cam{cam->fx, cam->fy, cam->cx, cam->cy, cam->huber_delta};
  ba.pidx.assign(P, -1);
  std::vector<uint8_t> pose_used(P, 0);
  ba.pt_active.assign(L, 0);
  ba.pt_edges.assign(L, {});
  for (int k = 0; k < E; k++) {
    pose_used[edges[k].pose] = 1;
    ba.pt_active[edges[k].point] = 1;
    ba.pt_edges[edges[k].point].push_back(k);
  }
  for (int p = 0; p < P; p++)
    if (!fixed[p] && pose_used[p]) ba.pidx[p] = ba.nfree++;
  ba.lidx.assign(L, -1);
  for (int l = 0; l < L; l++)
    if (ba.pt_active[l]) ba.lidx[l] = ba.nact++;
  const int n = 6 * ba.nfree;
  ba.Hpp.assign((size_t)36 * ba.nfree, 0); ba.bp.assign((size_t)n, 0);
  ba.Hll.assign((size_t)9 * ba.nact, 0); ba.bl.assign((size_t)3 * ba.nact, 0);
  ba.W.assign((size_t)18 * E, 0);
  ba.x.assign((size_t)n + 3 * ba.nact, 0);
  // envelope of the reduced camera matrix: first co-observing free pose of every pose
  std::vector<int> minblk(ba.nfree);
  for (int i = 0; i < ba.nfree; i++) minblk[i] = i;
  for (int l = 0; l < L; l++) {
    int lo = 1 << 30;
    for (int k : ba.pt_edges[l]) if (ba.pidx[edges[k].pose] >= 0) lo = std::min(lo, ba.pidx[edges[k].pose]);
    for (int k : ba.pt_edges[l]) { int i = ba.pidx[edges[k].pose]; if (i >= 0) minblk[i] = std::min(minblk[i], lo); }
  }
  std::vector<int> first(n);
  for (int i = 0; i < ba.nfree; i++) for (int a = 0; a < 6; a++) first[6 * i + a] = 6 * minblk[i];
  ba.S.setup(n, first);

  if (st) { std::memset(st, 0, sizeof(*st)); }
  double lambda = -1, ni = 2;
  int nBad = 0, it_done = 0, trials_total = 0, stop = 0;
  double chi_last = 0;
  for (int it = 0; it < iterations; it++) {
    double currentChi = ba.robust_chi2();
    double tempChi = currentChi;
    const double iniChi = currentChi;
    if (it == 0 && st) st->chi2_initial = currentChi;
    ba.build_system();
    if (it == 0) {  // computeLambdaInit: tau * max diagonal over all active vertices
      double mx = 0;
      for (int i = 0; i < ba.nfree; i++) for (int a = 0; a < 6; a++) mx = std::max(mx, std::fabs(ba.Hpp[36 * i + 7 * a]));
      for (int i = 0; i < ba.nact; i++) for (int a = 0; a < 3; a++) mx = std::max(mx, std::fabs(ba.Hll[9 * i + 4 * a]));
      lambda = 1e-5 * mx;
      ni = 2; nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      std::vector<Pose> poses_bak = ba.poses;
      std::vector<double> pts_bak = ba.pts;
      // optimization_algorithm_levenberg.cpp:107-127, line by line: the update is applied and the errors are evaluated WHETHER OR
      // NOT the linear solve succeeded -- after a failure _solver->x() still holds the last successful solve (the linear solver
      // returns before writing it, linear_solver_eigen.h:89-112; zeros here before the first success, where g2o's freshly
      // allocated _x is undefined) --, then tempChi is overridden with max() and divided by computeScale() of that x.  max() is
      // finite: with a negative scale the step would be accepted.
      const bool ok = ba.solve(lambda);
      ba.apply_update();
      tempChi = ba.robust_chi2();
      if (!ok) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = 0;
      for (int j = 0; j < n; j++) scale += ba.x[j] * (lambda * ba.x[j] + ba.bp[j]);
      for (int j = 0; j < 3 * ba.nact; j++) scale += ba.x[n + j] * (lambda * ba.x[n + j] + ba.bl[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - orc_spec::cube(2 * rho - 1);   // pow(2 rho - 1, 3), optimization_algorithm_levenberg.cpp:131 (f64_spec.h: the correctly rounded cube)
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        ba.poses = poses_bak;
        ba.pts = pts_bak;
      }
      qmax++;
      trials_total++;
    } while (rho < 0 && qmax < 10);
    it_done++;
    chi_last = currentChi;
    if (st && it < 64) { st->trials_per_iter[it] = qmax; st->chi2_per_iter[it] = currentChi; st->lambda_per_iter[it] = lambda; }
    if (qmax == 10 || rho == 0) { stop = 1; break; }
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) { stop = 2; break; }
  }
  for (int p = 0; p < P; p++) {
    std::memcpy(poses + 7 * p, ba.poses[p].t, 3 * sizeof(double));
    std::memcpy(poses + 7 * p + 3, ba.poses[p].q, 4 * sizeof(double));
  }
  std::memcpy(points, ba.pts.data(), sizeof(double) * 3 * (size_t)L);
  if (edge_chi2 && (int)ba.last_chi2.size() == E) std::memcpy(edge_chi2, ba.last_chi2.data(), sizeof(double) * (size_t)E);
  if (st) {
    st->iterations = it_done; st->total_trials = trials_total; st->chi2_final = chi_last; st->lambda_final = lambda;
    st->stop_reason = stop;
  }
  return it_done;
}

// the double-precision spec, for tests/test_f64_spec.py: out[0..n) = sin, out[n..2n) = cos, out[2n..3n) = cube
void orc_f64_spec(const double* x, int n, double* out) {
  for (int i = 0; i < n; i++) { out[i] = orc_spec::sin(x[i]); out[n + i] = orc_spec::cos(x[i]); out[2 * n + i] = orc_spec::cube(x[i]); }
}

// per-edge raw chi2 and depth sign at the given state (Optimizer.cc:1317-1354 outlier tests)
void orc_ba_edge_chi2(const double* poses, const double* points, const orc_ba_edge* edges, int E, const orc_ba_camera* cam,
                      double* chi2, uint8_t* depth_positive) {
  Cam c{cam->fx, cam->fy, cam->cx, cam->cy, cam->huber_delta};
  EdgeLin Lz;
  for (int k = 0; k < E; k++) {
    Pose T;
    std::memcpy(T.t, poses + 7 * edges[k].pose, 3 * sizeof(double));
    std::memcpy(T.q, poses + 7 * edges[k].pose + 3, 4 * sizeof(double));
    const double obs[2] = {edges[k].u, edges[k].v};
    linearize(T, points + 3 * (size_t)edges[k].point, obs, edges[k].inv_sigma2, c, Lz, false);
    chi2[k] = Lz.chi2;
    if (depth_positive) depth_positive[k] = Lz.Xc[2] > 0;
  }
}

// Optimizer::PoseOptimization (Optimizer.cc:744-1028), mono edges only.  pose: (t, q) double in/out
// (the reference reads/writes a float Sophus pose; callers cast).  Xw [N][3] double, obs [N][2],
// inv_sigma2 [N].  outlier[N] receives mvbOutlier.  Returns nInitialCorrespondences - nBad.
int orc_pose_optimize(double* pose, const double* Xw, const double* obs, const double* inv_sigma2, int N,
                      const orc_ba_camera* cam, uint8_t* outlier) {
  if (N < 3) return 0;
  const double delta = (double)(float)std::sqrt(5.991);  // const float deltaMono = sqrt(5.991)
  const float chi2Mono = 5.991f;
  Cam c{cam->fx, cam->fy, cam->cx, cam->cy, delta};
  Pose T0;
  std::memcpy(T0.t, pose, 3 * sizeof(double));
  std::memcpy(T0.q, pose + 3, 4 * sizeof(double));
  quat_normalize(T0.q);
  std::vector<uint8_t> level(N, 0), robust(N, 1);
  std::vector<double> last_chi2(N, 0.0);  // edge->chi2(): value at the last computeActiveErrors it took part in
  std::fill(outlier, outlier + N, 0);
  Pose T = T0;
  int nBad = 0;
  auto edge_err = [&](const Pose& P_, int i, EdgeLin& Lz, bool jac) {
    // unary edge: same projection; B (2x6) is the pose Jacobian (OptimizableTypes.cpp:51-63)
    linearize(P_, Xw + 3 * i, obs + 2 * i, inv_sigma2[i], c, Lz, jac);
  };
  for (int round = 0; round < 4; round++) {
    T = T0;  // vSE3->setEstimate(pFrame->GetPose()) every round
    auto chi_active = [&](const Pose& P_) {
      double chi = 0;
      EdgeLin Lz;
      for (int i = 0; i < N; i++) {
        if (level[i]) continue;
        edge_err(P_, i, Lz, false);
        last_chi2[i] = Lz.chi2;
        double r0, r1;
        robustify(Lz.chi2, robust[i] ? delta : 0.0, r0, r1);
        chi += r0;
      }
      return chi;
    };
    double lambda = -1, ni = 2;
    int nBadIt = 0;
    int nactive = 0;
    for (int i = 0; i < N; i++) nactive += !level[i];
    for (int it = 0; it < 10 && nactive > 0; it++) {
      double currentChi = chi_active(T), tempChi = currentChi;
      const double iniChi = currentChi;
      double H[36] = {0}, b[6] = {0};
      EdgeLin Lz;
      for (int i = 0; i < N; i++) {
        if (level[i]) continue;
        edge_err(T, i, Lz, true);
        double r0, r1;
        robustify(Lz.chi2, robust[i] ? delta : 0.0, r0, r1);
        const double w = r1 * inv_sigma2[i];
        const double wr[2] = {-inv_sigma2[i] * Lz.e[0] * r1, -inv_sigma2[i] * Lz.e[1] * r1};
        for (int a = 0; a < 6; a++) {
          b[a] += Lz.B[a] * wr[0] + Lz.B[6 + a] * wr[1];
          for (int bb = 0; bb < 6; bb++) H[6 * a + bb] += w * (Lz.B[a] * Lz.B[bb] + Lz.B[6 + a] * Lz.B[6 + bb]);
        }
      }
      if (it == 0) {
        double mx = 0;
        for (int a = 0; a < 6; a++) mx = std::max(mx, std::fabs(H[7 * a]));
        lambda = 1e-5 * mx; ni = 2; nBadIt = 0;
      }
      double rho = 0;
      int qmax = 0;
      do {
        Pose bak = T;
        // dense 6x6 Cholesky solve of (H + lambda I) x = b
        double Lm[36], x[6];
        bool ok = true;
        for (int i = 0; i < 6 && ok; i++)
          for (int j = 0; j <= i; j++) {
            double s = H[6 * i + j] + (i == j ? lambda : 0.0);
            for (int k = 0; k < j; k++) s -= Lm[6 * i + k] * Lm[6 * j + k];
            if (i == j) { if (!(s > 0)) { ok = false; break; } Lm[6 * i + i] = std::sqrt(s); }
            else Lm[6 * i + j] = s / Lm[6 * j + j];
          }
        if (ok) {
          for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= Lm[6 * i + k] * x[k]; x[i] = s / Lm[6 * i + i]; }
          for (int i = 5; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < 6; k++) s -= Lm[6 * k + i] * x[k]; x[i] = s / Lm[6 * i + i]; }
          pose_oplus(T, x);
        }
        tempChi = ok ? chi_active(T) : std::numeric_limits<double>::max();
        rho = currentChi - tempChi;
        double scale = 0;
        if (ok) for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - orc_spec::cube(2 * rho - 1);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
        } else {
          lambda *= ni; ni *= 2; T = bak;
        }
        qmax++;
      } while (rho < 0 && qmax < 10);
      if (qmax == 10 || rho == 0) break;
      if ((iniChi - currentChi) * 1e3 < iniChi) nBadIt++; else nBadIt = 0;
      if (nBadIt >= 3) break;
    }
    nBad = 0;
    EdgeLin Lz;
    for (int i = 0; i < N; i++) {
      if (outlier[i]) { edge_err(T, i, Lz, false); last_chi2[i] = Lz.chi2; }  // :929-931 recompute for outliers only
      const float chi2 = (float)last_chi2[i];
      if (chi2 > chi2Mono) { outlier[i] = 1; level[i] = 1; nBad++; }
      else { outlier[i] = 0; level[i] = 0; }
      if (round == 2) robust[i] = 0;
    }
    if (N < 10) break;  // optimizer.edges().size() < 10
  }
  std::memcpy(pose, T.t, 3 * sizeof(double));
  std::memcpy(pose + 3, T.q, 4 * sizeof(double));
  return N - nBad;
}

// ------------------------------------------------------------------------------------- Sim3
// g2o::Sim3 (Thirdparty/g2o/g2o/types/sim3.h:47-230): exp, product, inverse, map.
struct Sim3d { double q[4]; double t[3]; double s; };
static void sim3_exp(const double u[7], Sim3d& S) {
  const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]}, sigma = u[6];
  const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double O2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  S.s = std::exp(sigma);
  const double eps = 0.00001;
  double A, B, C, R[9];
  if (std::fabs(sigma) < eps) {
    C = 1;
    if (theta < eps) { A = 0.5; B = 1. / 6.; for (int i = 0; i < 9; i++) R[i] = I[i] + O[i] + O2[i]; }
    else {
      const double th2 = theta * theta;
      A = (1 - std::cos(theta)) / th2; B = (theta - std::sin(theta)) / (th2 * theta);
      for (int i = 0; i < 9; i++) R[i] = I[i] + std::sin(theta) / theta * O[i] + (1 - std::cos(theta)) / (theta * theta) * O2[i];
    }
  } else {
    C = (S.s - 1) / sigma;
    if (theta < eps) {
      const double s2 = sigma * sigma;
      A = ((sigma - 1) * S.s + 1) / s2; B = ((0.5 * s2 - sigma + 1) * S.s) / (s2 * sigma);
      for (int i = 0; i < 9; i++) R[i] = I[i] + O[i] + O2[i];
    } else {
      for (int i = 0; i < 9; i++) R[i] = I[i] + std::sin(theta) / theta * O[i] + (1 - std::cos(theta)) / (theta * theta) * O2[i];
      const double a = S.s * std::sin(theta), b = S.s * std::cos(theta), th2 = theta * theta, s2 = sigma * sigma, c = th2 + s2;
      A = (a * sigma + (1 - b) * theta) / (theta * c);
      B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / th2;
    }
  }
  R_to_quat(R, S.q);
  double W[9];
  for (int i = 0; i < 9; i++) W[i] = A * O[i] + B * O2[i] + C * I[i];
  mat3_vec(W, up, S.t);
}
static void sim3_mul(const Sim3d& a, const Sim3d& b, Sim3d& o) {
  quat_mul(a.q, b.q, o.q);
  double R[9], rt[3];
  quat_to_R(a.q, R);
  mat3_vec(R, b.t, rt);
  for (int i = 0; i < 3; i++) o.t[i] = a.s * rt[i] + a.t[i];
  o.s = a.s * b.s;
}
static void sim3_inv(const Sim3d& a, Sim3d& o) {
  o.q[0] = -a.q[0]; o.q[1] = -a.q[1]; o.q[2] = -a.q[2]; o.q[3] = a.q[3];
  double R[9], v[3] = {(-1. / a.s) * a.t[0], (-1. / a.s) * a.t[1], (-1. / a.s) * a.t[2]};
  quat_to_R(o.q, R);
  mat3_vec(R, v, o.t);
  o.s = 1. / a.s;
}
static void sim3_map(const Sim3d& a, const double x[3], double o[3]) {
  double R[9], rx[3];
  quat_to_R(a.q, R);
  mat3_vec(R, x, rx);
  for (int i = 0; i < 3; i++) o[i] = a.s * rx[i] + a.t[i];
}

// Optimizer::OptimizeSim3 numerics (Optimizer.cc:1960-2212) for N correspondences already gathered by the
// caller: P1c/P2c = map points in their own key frame's camera frame, obs1/obs2 = undistorted keypoints,
// w1/w2 = mvInvLevelSigma2, pinhole intrinsics of both cameras.  S12 in/out = (qx,qy,qz,qw, tx,ty,tz, s).
// Edges: e12 = obs1 - proj1(S12 * P2c), e21 = obs2 - proj2(S12^-1 * P1c); NUMERIC Jacobians (g2o
// BaseBinaryEdge::linearizeOplus, delta 1e-9, base_binary_edge.hpp:131-205); dense 7x7 LM.
// inlier[N] receives 1 for pairs that survive; returns nIn (0 when fewer than 10 pairs survive round 1).
int orc_optimize_sim3(double* S12io, int fix_scale, const double* P1c, const double* P2c, const double* obs1,
                      const double* obs2, const double* w1, const double* w2, int N, const double* K1, const double* K2,
                      double th2, uint8_t* inlier) {
  Sim3d S;
  std::memcpy(S.q, S12io, 4 * sizeof(double)); std::memcpy(S.t, S12io + 4, 3 * sizeof(double)); S.s = S12io[7];
  const double deltaHuber = (double)(float)std::sqrt((float)th2);  // const float deltaHuber = sqrt(th2)
  std::vector<uint8_t> alive(N, 1), robust(N, 1);
  std::vector<double> chi12(N, 0), chi21(N, 0);
  auto proj = [](const double* K, const double X[3], double uv[2]) { uv[0] = K[0] * X[0] / X[2] + K[2]; uv[1] = K[1] * X[1] / X[2] + K[3]; };
  auto errors = [&](const Sim3d& Sx, int i, double e12[2], double e21[2]) {
    Sim3d Si; sim3_inv(Sx, Si);
    double X[3], uv[2];
    sim3_map(Sx, P2c + 3 * i, X); proj(K1, X, uv); e12[0] = obs1[2 * i] - uv[0]; e12[1] = obs1[2 * i + 1] - uv[1];
    sim3_map(Si, P1c + 3 * i, X); proj(K2, X, uv); e21[0] = obs2[2 * i] - uv[0]; e21[1] = obs2[2 * i + 1] - uv[1];
  };
  auto chi_all = [&](const Sim3d& Sx) {
    double chi = 0;
    for (int i = 0; i < N; i++) {
      if (!alive[i]) continue;
      double a[2], b[2]; errors(Sx, i, a, b);
      chi12[i] = w1[i] * (a[0] * a[0] + a[1] * a[1]); chi21[i] = w2[i] * (b[0] * b[0] + b[1] * b[1]);
      double r0, r1;
      robustify(chi12[i], robust[i] ? deltaHuber : 0.0, r0, r1); chi += r0;
      robustify(chi21[i], robust[i] ? deltaHuber : 0.0, r0, r1); chi += r0;
    }
    return chi;
  };
  auto optimize = [&](int iters) {
    double lambda = -1, ni = 2; int nBad = 0;
    for (int it = 0; it < iters; it++) {
      double currentChi = chi_all(S), tempChi = currentChi; const double iniChi = currentChi;
      // perturbed estimates for the numeric Jacobian: Sim3(+-delta e_d) * S
      Sim3d Sp[7], Sm[7];
      for (int d = 0; d < 7; d++) {
        double u[7] = {0, 0, 0, 0, 0, 0, 0};
        u[d] = 1e-9; if (fix_scale) u[6] = 0; Sim3d E; sim3_exp(u, E); sim3_mul(E, S, Sp[d]);
        u[d] = -1e-9; if (fix_scale) u[6] = 0; sim3_exp(u, E); sim3_mul(E, S, Sm[d]);
      }
      double H[49] = {0}, b[7] = {0};
      for (int i = 0; i < N; i++) {
        if (!alive[i]) continue;
        double J12[14], J21[14];  // 2x7 row-major
        for (int d = 0; d < 7; d++) {
          double ap[2], bp[2], am[2], bm[2];
          errors(Sp[d], i, ap, bp); errors(Sm[d], i, am, bm);
          for (int r = 0; r < 2; r++) { J12[7 * r + d] = 5e8 * (ap[r] - am[r]); J21[7 * r + d] = 5e8 * (bp[r] - bm[r]); }
        }
        double e12[2], e21[2]; errors(S, i, e12, e21);
        for (int pass = 0; pass < 2; pass++) {
          const double* J = pass ? J21 : J12; const double* e = pass ? e21 : e12; const double w0 = pass ? w2[i] : w1[i];
          const double chi2 = w0 * (e[0] * e[0] + e[1] * e[1]);
          double r0, r1; robustify(chi2, robust[i] ? deltaHuber : 0.0, r0, r1);
          const double w = r1 * w0, wr0 = -w0 * e[0] * r1, wr1 = -w0 * e[1] * r1;
          for (int a = 0; a < 7; a++) {
            b[a] += J[a] * wr0 + J[7 + a] * wr1;
            for (int c = 0; c < 7; c++) H[7 * a + c] += w * (J[a] * J[c] + J[7 + a] * J[7 + c]);
          }
        }
      }
      if (it == 0) { double mx = 0; for (int a = 0; a < 7; a++) mx = std::max(mx, std::fabs(H[8 * a])); lambda = 1e-5 * mx; ni = 2; nBad = 0; }
      double rho = 0; int qmax = 0;
      do {
        const Sim3d bak = S;
        double Lm[49], x[7]; bool ok = true;
        for (int i = 0; i < 7 && ok; i++)
          for (int j = 0; j <= i; j++) {
            double sacc = H[7 * i + j] + (i == j ? lambda : 0.0);
            for (int k = 0; k < j; k++) sacc -= Lm[7 * i + k] * Lm[7 * j + k];
            if (i == j) { if (!(sacc > 0)) { ok = false; break; } Lm[7 * i + i] = std::sqrt(sacc); } else Lm[7 * i + j] = sacc / Lm[7 * j + j];
          }
        if (ok) {
          for (int i = 0; i < 7; i++) { double sacc = b[i]; for (int k = 0; k < i; k++) sacc -= Lm[7 * i + k] * x[k]; x[i] = sacc / Lm[7 * i + i]; }
          for (int i = 6; i >= 0; i--) { double sacc = x[i]; for (int k = i + 1; k < 7; k++) sacc -= Lm[7 * k + i] * x[k]; x[i] = sacc / Lm[7 * i + i]; }
          double u[7]; std::memcpy(u, x, sizeof(u)); if (fix_scale) u[6] = 0;
          Sim3d E; sim3_exp(u, E); Sim3d Sn; sim3_mul(E, S, Sn); S = Sn;
        }
        tempChi = ok ? chi_all(S) : std::numeric_limits<double>::max();
        rho = currentChi - tempChi;
        double scale = 0; if (ok) for (int j = 0; j < 7; j++) scale += x[j] * (lambda * x[j] + b[j]);
        scale += 1e-3; rho /= scale;
        if (rho > 0 && std::isfinite(tempChi)) { double alpha = 1. - orc_spec::cube(2 * rho - 1); alpha = std::min(alpha, 2. / 3.); lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi; }
        else { lambda *= ni; ni *= 2; S = bak; }
        qmax++;
      } while (rho < 0 && qmax < 10);
      if (qmax == 10 || rho == 0) break;
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      if (nBad >= 3) break;
    }
  };
  optimize(5);
  int nBad = 0;
  for (int i = 0; i < N; i++) {
    if (chi12[i] > th2 || chi21[i] > th2) { alive[i] = 0; nBad++; } else robust[i] = 0;
  }
  for (int i = 0; i < N; i++) inlier[i] = 0;
  if (N - nBad < 10) return 0;
  optimize(nBad > 0 ? 10 : 5);
  int nIn = 0;
  for (int i = 0; i < N; i++) {
    if (!alive[i]) continue;
    double a[2], b[2]; errors(S, i, a, b);
    const double c12 = w1[i] * (a[0] * a[0] + a[1] * a[1]), c21 = w2[i] * (b[0] * b[0] + b[1] * b[1]);
    if (!(c12 > th2 || c21 > th2)) { inlier[i] = 1; nIn++; }
  }
  std::memcpy(S12io, S.q, 4 * sizeof(double)); std::memcpy(S12io + 4, S.t, 3 * sizeof(double)); S12io[7] = S.s;
  return nIn;
}

// ---------------------------------------------------------------------------------------------------------------
// Optimizer::OptimizeEssentialGraph numerics (reference src/Optimizer.cc:1389-1652): a pose graph of VertexSim3Expmap
// (estimate Siw, oplus: S <- Sim3(update) * S, scale frozen by _fix_scale) and EdgeSim3 (error = log(Sji * Siw * Sjw^-1),
// information = identity, no robust kernel; Thirdparty/g2o/g2o/types/types_seven_dof_expmap.h:93-117) whose Jacobians
// g2o takes numerically (base_binary_edge.hpp:131-205, central differences, delta = 1e-9), solved by
// OptimizationAlgorithmLevenberg with setUserLambdaInit(1e-16) and optimize(20).  Graph assembly (which keyframes,
// which spanning-tree / loop / covisibility edges, their measurements Sji) stays with the caller.
// Sim3::log follows sim3.h:128-197; W.lu().solve(t) is restated as the adjugate inverse (3x3).
static void sim3_log(const Sim3d& S, double res[7]) {
  const double sigma = std::log(S.s);
  double R[9];
  quat_to_R(S.q, R);
  const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};   // deltaR
  double omega[3];
  const double eps = 0.00001;
  double A, B, C;
  if (std::fabs(sigma) < eps) {
    C = 1;
    if (d > 1 - eps) { for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i]; A = 1. / 2.; B = 1. / 6.; }
    else {
      const double theta = std::acos(d), theta2 = theta * theta;
      for (int i = 0; i < 3; i++) omega[i] = theta / (2 * std::sqrt(1 - d * d)) * dR[i];
      A = (1 - std::cos(theta)) / theta2; B = (theta - std::sin(theta)) / (theta2 * theta);
    }
  } else {
    C = (S.s - 1) / sigma;
    if (d > 1 - eps) {
      const double sigma2 = sigma * sigma;
      for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i];
      A = ((sigma - 1) * S.s + 1) / sigma2; B = ((0.5 * sigma2 - sigma + 1) * S.s) / (sigma2 * sigma);
    } else {
      const double theta = std::acos(d);
      for (int i = 0; i < 3; i++) omega[i] = theta / (2 * std::sqrt(1 - d * d)) * dR[i];
      const double theta2 = theta * theta, a = S.s * std::sin(theta), b = S.s * std::cos(theta), c = theta2 + sigma * sigma;
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
  // upsilon = W^-1 t
  const double c00 = W[4] * W[8] - W[5] * W[7], c01 = W[5] * W[6] - W[3] * W[8], c02 = W[3] * W[7] - W[4] * W[6];
  const double det = W[0] * c00 + W[1] * c01 + W[2] * c02;
  const double inv[9] = {c00, W[2] * W[7] - W[1] * W[8], W[1] * W[5] - W[2] * W[4],
                         c01, W[0] * W[8] - W[2] * W[6], W[2] * W[3] - W[0] * W[5],
                         c02, W[1] * W[6] - W[0] * W[7], W[0] * W[4] - W[1] * W[3]};
  for (int i = 0; i < 3; i++) res[i] = omega[i];
  for (int i = 0; i < 3; i++) res[3 + i] = (inv[3 * i] * S.t[0] + inv[3 * i + 1] * S.t[1] + inv[3 * i + 2] * S.t[2]) / det;
  res[6] = sigma;
}

// EdgeSim3::computeError: log(C * v1 * v2^-1), v1 = vertex 0 (i), v2 = vertex 1 (j)
static void pg_edge_error(const Sim3d& C, const Sim3d& Si, const Sim3d& Sj, double e[7]) {
  Sim3d Sji, t1, t2;
  sim3_inv(Sj, Sji);
  sim3_mul(C, Si, t1);
  sim3_mul(t1, Sji, t2);
  sim3_log(t2, e);
}

// test hook: g2o::Sim3(update) and Sim3::log of it (pins sim3_exp / sim3_log against scipy's expm in tests/)
void orc_sim3_exp_log(const double* u, double* S8, double* log7) {
  Sim3d S;
  sim3_exp(u, S);
  std::memcpy(S8, S.q, 32); std::memcpy(S8 + 4, S.t, 24); S8[7] = S.s;
  sim3_log(S, log7);
}

int orc_pose_graph_optimize(double* Sio, const uint8_t* fixed, int n, const int32_t* ev, const double* emeas, int E, int fix_scale,
                            int iterations, double* stats /*[6 + 64]: iterations, trials, chi2_initial, chi2_final, lambda_final, stop, chi2_per_iter[32], trials_per_iter[32]*/) {
  std::vector<Sim3d> S(n);
  for (int v = 0; v < n; v++) { std::memcpy(S[v].q, Sio + 8 * v, 32); std::memcpy(S[v].t, Sio + 8 * v + 4, 24); S[v].s = Sio[8 * v + 7]; }
  std::vector<Sim3d> C(E);
  for (int k = 0; k < E; k++) { std::memcpy(C[k].q, emeas + 8 * k, 32); std::memcpy(C[k].t, emeas + 8 * k + 4, 24); C[k].s = emeas[8 * k + 7]; }
  std::vector<int> idx(n, -1);
  int m = 0;
  for (int v = 0; v < n; v++) if (!fixed[v]) idx[v] = m++;
  const int dim = 7 * m;
  auto chi_all = [&]() { double chi = 0; for (int k = 0; k < E; k++) { double e[7]; pg_edge_error(C[k], S[ev[2 * k]], S[ev[2 * k + 1]], e); for (int a = 0; a < 7; a++) chi += e[a] * e[a]; } return chi; };
  auto oplus = [&](Sim3d& X, const double* u) {
    double uu[7]; for (int a = 0; a < 7; a++) uu[a] = u[a];
    if (fix_scale) uu[6] = 0;
    Sim3d Ex, Sn; sim3_exp(uu, Ex); sim3_mul(Ex, X, Sn); X = Sn;
  };
  double lambda = 1e-16, ni = 2;   // setUserLambdaInit(1e-16)
  int nBad = 0, it_done = 0, trials = 0, stop = 0;
  double chi_last = 0, chi_init = 0;
  std::vector<double> H((size_t)dim * dim), Hl, b(dim), x(dim);
  std::vector<int> first(dim);   // envelope of the lower triangle
  for (int r = 0; r < dim; r++) first[r] = r - r % 7;
  for (int k = 0; k < E; k++) {
    const int a = idx[ev[2 * k]], c = idx[ev[2 * k + 1]];
    if (a < 0 || c < 0) continue;
    const int hi = std::max(a, c), lo = std::min(a, c);
    for (int r = 7 * hi; r < 7 * hi + 7; r++) first[r] = std::min(first[r], 7 * lo);
  }
  for (int it = 0; it < iterations; it++) {
    double currentChi = chi_all(), tempChi = currentChi;
    const double iniChi = currentChi;
    if (it == 0) chi_init = currentChi;
    std::fill(H.begin(), H.end(), 0.0); std::fill(b.begin(), b.end(), 0.0);
    for (int k = 0; k < E; k++) {   // linearizeOplus (numeric) + constructQuadraticForm
      const int vi = ev[2 * k], vj = ev[2 * k + 1];
      double e[7], J[2][49];
      pg_edge_error(C[k], S[vi], S[vj], e);
      for (int side = 0; side < 2; side++) {
        const int v = side ? vj : vi;
        if (fixed[v]) continue;
        for (int d = 0; d < 7; d++) {
          double u[7] = {0, 0, 0, 0, 0, 0, 0}, e1[7], e2[7];
          Sim3d bak = S[v];
          u[d] = 1e-9; oplus(S[v], u); pg_edge_error(C[k], S[vi], S[vj], e1); S[v] = bak;
          u[d] = -1e-9; oplus(S[v], u); pg_edge_error(C[k], S[vi], S[vj], e2); S[v] = bak;
          for (int a = 0; a < 7; a++) J[side][7 * a + d] = (1.0 / (2 * 1e-9)) * (e1[a] - e2[a]);
        }
      }
      for (int sa = 0; sa < 2; sa++) {
        const int va = sa ? vj : vi;
        if (fixed[va]) continue;
        const int ia = 7 * idx[va];
        for (int r = 0; r < 7; r++) { double g = 0; for (int a = 0; a < 7; a++) g += J[sa][7 * a + r] * (-e[a]); b[ia + r] += g; }
        for (int sb = 0; sb < 2; sb++) {
          const int vb = sb ? vj : vi;
          if (fixed[vb]) continue;
          const int ib = 7 * idx[vb];
          for (int r = 0; r < 7; r++)
            for (int c = 0; c < 7; c++) { double hsum = 0; for (int a = 0; a < 7; a++) hsum += J[sa][7 * a + r] * J[sb][7 * a + c]; H[(size_t)(ia + r) * dim + ib + c] += hsum; }
        }
      }
    }
    double rho = 0;
    int qmax = 0;
    do {
      std::vector<Sim3d> bak = S;   // push()
      Hl = H;
      for (int r = 0; r < dim; r++) Hl[(size_t)r * dim + r] += lambda;
      // Cholesky + solve on the ENVELOPE of H (row r starts at its first structurally non-zero column): the skipped
      // products are exact zeros of the dense factorisation, every kept sum runs over ascending k as before -> same bits
      bool ok = true;
      for (int r = 0; r < dim && ok; r++) {
        double* Lr = &Hl[(size_t)r * dim];
        for (int j = first[r]; j < r; j++) {
          const double* Lj = &Hl[(size_t)j * dim];
          double v = Lr[j];
          for (int k = std::max(first[r], first[j]); k < j; k++) v -= Lr[k] * Lj[k];
          Lr[j] = v / Lj[j];
        }
        double dj = Lr[r];
        for (int k = first[r]; k < r; k++) dj -= Lr[k] * Lr[k];
        if (!(dj > 0)) { ok = false; break; }
        Lr[r] = std::sqrt(dj);
      }
      if (ok) {
        for (int r = 0; r < dim; r++) { double v = b[r]; for (int k = first[r]; k < r; k++) v -= Hl[(size_t)r * dim + k] * x[k]; x[r] = v / Hl[(size_t)r * dim + r]; }
        for (int r = dim - 1; r >= 0; r--) { double v = x[r]; for (int k = r + 1; k < dim; k++) if (first[k] <= r) v -= Hl[(size_t)k * dim + r] * x[k]; x[r] = v / Hl[(size_t)r * dim + r]; }
        for (int v = 0; v < n; v++) if (!fixed[v]) oplus(S[v], &x[7 * idx[v]]);
      }
      tempChi = ok ? chi_all() : std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = 0;
      if (ok) for (int r = 0; r < dim; r++) scale += x[r] * (lambda * x[r] + b[r]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - orc_spec::cube(2 * rho - 1);
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni; ni *= 2;
        S = bak;   // pop()
      }
      qmax++; trials++;
    } while (rho < 0 && qmax < 10);
    if (stats && it < 32) { stats[6 + it] = currentChi; stats[38 + it] = qmax; }
    it_done++;
    chi_last = currentChi;
    if (qmax == 10 || rho == 0) { stop = 1; break; }
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) { stop = 2; break; }
  }
  for (int v = 0; v < n; v++) { std::memcpy(Sio + 8 * v, S[v].q, 32); std::memcpy(Sio + 8 * v + 4, S[v].t, 24); Sio[8 * v + 7] = S[v].s; }
  if (stats) { stats[0] = it_done; stats[1] = trials; stats[2] = chi_init; stats[3] = chi_last; stats[4] = lambda; stats[5] = stop; }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Sim3Solver (reference src/Sim3Solver.cc): ComputeSim3 (Horn 1987, :294-385) for a batch of RANSAC hypotheses, each
// followed by CheckInliers (:387-408) with Project (:421-435) / Pinhole::project.  The minimal sets are INPUT
// (`triples`: the reference draws them with DUtils::Random, :171-181).  Arithmetic types follow the reference: points,
// centroids, M, N, R, s, t and the reprojection errors are float; only the eigen-decomposition differs in method:
// the reference runs Eigen::EigenSolver<Matrix4f>; here (and on the device) the symmetric N is diagonalised in double
// by cyclic Jacobi rotations ("Horn spec").  max_err* = (float)(size_t)(9.210 * sigma2) as the reference stores them.
namespace {
void jacobi4(double A[4][4], double V[4][4]) {
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 16; sweep++) {
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 4; k++) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
        for (int k = 0; k < 4; k++) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
        for (int k = 0; k < 4; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
      }
  }
}
}  // namespace

void orc_sim3_hypotheses(const float* P1c, const float* P2c, const float* max_err1, const float* max_err2, int N, const float* K1,
                         const float* K2, const int32_t* triples, int H, int fix_scale, float* T12 /*[H][12]: s R(9) t(3)... see below*/,
                         int32_t* n_inliers, uint8_t* inlier_mask /*[H][N]*/) {
  std::vector<float> p1im1(2 * (size_t)N), p2im2(2 * (size_t)N);
  for (int i = 0; i < N; i++) {   // FromCameraToImage (:437-446)
    p1im1[2 * i] = K1[0] * P1c[3 * i] / P1c[3 * i + 2] + K1[2]; p1im1[2 * i + 1] = K1[1] * P1c[3 * i + 1] / P1c[3 * i + 2] + K1[3];
    p2im2[2 * i] = K2[0] * P2c[3 * i] / P2c[3 * i + 2] + K2[2]; p2im2[2 * i + 1] = K2[1] * P2c[3 * i + 1] / P2c[3 * i + 2] + K2[3];
  }
  for (int h = 0; h < H; h++) {
    float P1[3][3], P2[3][3];   // [row][col], column i = point i
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) { P1[r][c] = P1c[3 * triples[3 * h + c] + r]; P2[r][c] = P2c[3 * triples[3 * h + c] + r]; }
    float O1[3], O2[3], Pr1[3][3], Pr2[3][3];
    for (int r = 0; r < 3; r++) {
      O1[r] = ((P1[r][0] + P1[r][1]) + P1[r][2]) / 3.0f; O2[r] = ((P2[r][0] + P2[r][1]) + P2[r][2]) / 3.0f;
      for (int c = 0; c < 3; c++) { Pr1[r][c] = P1[r][c] - O1[r]; Pr2[r][c] = P2[r][c] - O2[r]; }
    }
    float M[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[r][c] = (Pr2[r][0] * Pr1[c][0] + Pr2[r][1] * Pr1[c][1]) + Pr2[r][2] * Pr1[c][2];
    const float N11 = M[0][0] + M[1][1] + M[2][2], N12 = M[1][2] - M[2][1], N13 = M[2][0] - M[0][2], N14 = M[0][1] - M[1][0];
    const float N22 = M[0][0] - M[1][1] - M[2][2], N23 = M[0][1] + M[1][0], N24 = M[2][0] + M[0][2];
    const float N33 = -M[0][0] + M[1][1] - M[2][2], N34 = M[1][2] + M[2][1], N44 = -M[0][0] - M[1][1] + M[2][2];
    double A[4][4] = {{N11, N12, N13, N14}, {N12, N22, N23, N24}, {N13, N23, N33, N34}, {N14, N24, N34, N44}}, V[4][4];
    jacobi4(A, V);
    int mi = 0;
    for (int k = 1; k < 4; k++) if (A[k][k] > A[mi][mi]) mi = k;
    const double q0 = V[0][mi], vx = V[1][mi], vy = V[2][mi], vz = V[3][mi];
    const double vn = std::sqrt(vx * vx + vy * vy + vz * vz);
    const double ang = std::atan2(vn, q0);
    float R[3][3];
    {  // vec = 2*ang*vec/|vec|; R = SO3::exp(vec)  (Rodrigues via the unit quaternion (cos ang, sin ang * axis))
      double ax = 0, ay = 0, az = 0;
      if (vn > 0) { ax = vx / vn; ay = vy / vn; az = vz / vn; }
      const double w = std::cos(ang), sh = std::sin(ang), x = sh * ax, y = sh * ay, z = sh * az;
      R[0][0] = (float)(1 - 2 * (y * y + z * z)); R[0][1] = (float)(2 * (x * y - z * w)); R[0][2] = (float)(2 * (x * z + y * w));
      R[1][0] = (float)(2 * (x * y + z * w)); R[1][1] = (float)(1 - 2 * (x * x + z * z)); R[1][2] = (float)(2 * (y * z - x * w));
      R[2][0] = (float)(2 * (x * z - y * w)); R[2][1] = (float)(2 * (y * z + x * w)); R[2][2] = (float)(1 - 2 * (x * x + y * y));
    }
    float P3[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) P3[r][c] = (R[r][0] * Pr2[0][c] + R[r][1] * Pr2[1][c]) + R[r][2] * Pr2[2][c];
    float sc = 1.0f;
    if (!fix_scale) {
      float nom = 0, den = 0;
      for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) { nom += Pr1[r][c] * P3[r][c]; den += P3[r][c] * P3[r][c]; }   // column-major sum
      sc = (float)((double)nom / (double)den);
    }
    float t[3];
    for (int r = 0; r < 3; r++) t[r] = O1[r] - ((sc * R[r][0]) * O2[0] + (sc * R[r][1]) * O2[1] + (sc * R[r][2]) * O2[2]);
    float sR[3][3], sRi[3][3], ti[3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { sR[r][c] = sc * R[r][c]; sRi[r][c] = (float)((1.0 / sc) * R[c][r]); }
    for (int r = 0; r < 3; r++) ti[r] = (-sRi[r][0] * t[0] + -sRi[r][1] * t[1]) + -sRi[r][2] * t[2];
    float* out = T12 + 13 * (size_t)h;
    out[0] = sc;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out[1 + 3 * r + c] = R[r][c];
    for (int r = 0; r < 3; r++) out[10 + r] = t[r];
    int nin = 0;
    for (int i = 0; i < N; i++) {   // CheckInliers
      float a[3], b[3];
      for (int r = 0; r < 3; r++) {
        a[r] = ((sR[r][0] * P2c[3 * i] + sR[r][1] * P2c[3 * i + 1]) + sR[r][2] * P2c[3 * i + 2]) + t[r];      // X2 in camera 1
        b[r] = ((sRi[r][0] * P1c[3 * i] + sRi[r][1] * P1c[3 * i + 1]) + sRi[r][2] * P1c[3 * i + 2]) + ti[r];   // X1 in camera 2
      }
      const float u1 = K1[0] * a[0] / a[2] + K1[2], v1 = K1[1] * a[1] / a[2] + K1[3];
      const float u2 = K2[0] * b[0] / b[2] + K2[2], v2 = K2[1] * b[1] / b[2] + K2[3];
      const float d1x = p1im1[2 * i] - u1, d1y = p1im1[2 * i + 1] - v1, d2x = u2 - p2im2[2 * i], d2y = v2 - p2im2[2 * i + 1];
      const float err1 = d1x * d1x + d1y * d1y, err2 = d2x * d2x + d2y * d2y;
      const bool in = err1 < max_err1[i] && err2 < max_err2[i];
      inlier_mask[(size_t)h * N + i] = in ? 1 : 0;
      nin += in ? 1 : 0;
    }
    n_inliers[h] = nin;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LocalMapping::CreateNewMapPoints, the geometry of ONE neighbour keyframe (reference src/LocalMapping.cc:598-741, monocular
// pinhole branch) for the matches SearchForTriangulation returned: parallax of the two rays, GeometricTools::Triangulate
// (src/GeometricTools.cc:48-67), positive depth in both cameras, reprojection error against 5.991 * sigma2 in both, the
// distance-ratio / octave-ratio consistency test.  float arithmetic in Eigen's evaluation order (3-term sums a0 + (a1 + a2),
// v / s as divisions, Pinhole::project as (fx * x) / z + cx); comparisons against double literals in double, as C++ promotes
// them.  The one step that differs in METHOD: the reference takes the right singular vector of the smallest singular value of
// the 4x4 float matrix A from Eigen::JacobiSVD<Matrix4f>; here (and on the device) it is the eigenvector of the smallest
// eigenvalue of A^T A, formed and diagonalised in double by cyclic Jacobi rotations and rounded to float -- tolerance parity on
// x3D (the same vector up to float SVD error), identical decisions away from the thresholds.
// status: 0 accepted; 1 parallax (cos <= 0 or >= cos_parallax_max: "no stereo and very low parallax"); 2 homogeneous w == 0;
// 3 z1 <= 0; 4 z2 <= 0; 5 reprojection error in keyframe 1; 6 in keyframe 2; 7 zero distance; 8 far point; 9 scale consistency.
void orc_triangulate_matches(const float* K1, const float* K2, const float* T1w /*3x4 row-major: KeyFrame::GetPose().matrix3x4()*/,
                             const float* T2w, const float* Ow1, const float* Ow2, const orc_keypoint* kps1, const orc_keypoint* kps2,
                             const int32_t* pairs /*[n][2]*/, int n, const float* sigma2_1, const float* sigma2_2, const float* sf1,
                             const float* sf2, float ratio_factor, double cos_parallax_max, int far_points, float th_far,
                             float* x3D_out /*[n][3]*/, int32_t* status) {
  auto sum3 = [](float a, float b, float c) { return a + (b + c); };
  for (int m = 0; m < n; m++) {
    const orc_keypoint& kp1 = kps1[pairs[2 * m]];
    const orc_keypoint& kp2 = kps2[pairs[2 * m + 1]];
    float* X = x3D_out + 3 * (size_t)m;
    X[0] = X[1] = X[2] = 0.0f;
    // unprojectEig (Pinhole.cpp:61-63)
    const float xn1[3] = {(kp1.x - K1[2]) / K1[0], (kp1.y - K1[3]) / K1[1], 1.0f};
    const float xn2[3] = {(kp2.x - K2[2]) / K2[0], (kp2.y - K2[3]) / K2[1], 1.0f};
    // ray = Rwc * xn, Rwc = Rcw^T (:632-633)
    float r1[3], r2[3];
    for (int i = 0; i < 3; i++) {
      r1[i] = sum3(T1w[0 * 4 + i] * xn1[0], T1w[1 * 4 + i] * xn1[1], T1w[2 * 4 + i] * xn1[2]);
      r2[i] = sum3(T2w[0 * 4 + i] * xn2[0], T2w[1 * 4 + i] * xn2[1], T2w[2 * 4 + i] * xn2[2]);
    }
    const float n1 = std::sqrt(sum3(r1[0] * r1[0], r1[1] * r1[1], r1[2] * r1[2])), n2 = std::sqrt(sum3(r2[0] * r2[0], r2[1] * r2[1], r2[2] * r2[2]));
    const float cosParallaxRays = sum3(r1[0] * r2[0], r1[1] * r2[1], r1[2] * r2[2]) / (n1 * n2);
    // mono: cosParallaxStereo = cosParallaxRays + 1 (:636-646), no stereo alternative (:663-675)
    const float cosParallaxStereo = cosParallaxRays + 1;
    if (!(cosParallaxRays < cosParallaxStereo && cosParallaxRays > 0 && (double)cosParallaxRays < cos_parallax_max)) { status[m] = 1; continue; }
    // GeometricTools::Triangulate
    float A[4][4];
    for (int k = 0; k < 4; k++) {
      A[0][k] = xn1[0] * T1w[8 + k] - T1w[0 + k];
      A[1][k] = xn1[1] * T1w[8 + k] - T1w[4 + k];
      A[2][k] = xn2[0] * T2w[8 + k] - T2w[0 + k];
      A[3][k] = xn2[1] * T2w[8 + k] - T2w[4 + k];
    }
    double B[4][4], V[4][4];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        double acc = 0.0;
        for (int k = 0; k < 4; k++) acc += (double)A[k][i] * (double)A[k][j];
        B[i][j] = acc;
      }
    jacobi4(B, V);
    int mi = 0;
    for (int k = 1; k < 4; k++) if (B[k][k] < B[mi][mi]) mi = k;
    const float vh[4] = {(float)V[0][mi], (float)V[1][mi], (float)V[2][mi], (float)V[3][mi]};
    if (vh[3] == 0) { status[m] = 2; continue; }
    const float x3D[3] = {vh[0] / vh[3], vh[1] / vh[3], vh[2] / vh[3]};
    X[0] = x3D[0]; X[1] = x3D[1]; X[2] = x3D[2];
    const float z1 = sum3(T1w[8] * x3D[0], T1w[9] * x3D[1], T1w[10] * x3D[2]) + T1w[11];
    if (z1 <= 0) { status[m] = 3; continue; }
    const float z2 = sum3(T2w[8] * x3D[0], T2w[9] * x3D[1], T2w[10] * x3D[2]) + T2w[11];
    if (z2 <= 0) { status[m] = 4; continue; }
    const float sigmaSquare1 = sigma2_1[kp1.octave];
    const float x1 = sum3(T1w[0] * x3D[0], T1w[1] * x3D[1], T1w[2] * x3D[2]) + T1w[3];
    const float y1 = sum3(T1w[4] * x3D[0], T1w[5] * x3D[1], T1w[6] * x3D[2]) + T1w[7];
    {
      const float u = K1[0] * x1 / z1 + K1[2], v = K1[1] * y1 / z1 + K1[3];
      const float errX1 = u - kp1.x, errY1 = v - kp1.y;
      if ((errX1 * errX1 + errY1 * errY1) > 5.991 * sigmaSquare1) { status[m] = 5; continue; }
    }
    const float sigmaSquare2 = sigma2_2[kp2.octave];
    const float x2 = sum3(T2w[0] * x3D[0], T2w[1] * x3D[1], T2w[2] * x3D[2]) + T2w[3];
    const float y2 = sum3(T2w[4] * x3D[0], T2w[5] * x3D[1], T2w[6] * x3D[2]) + T2w[7];
    {
      const float u = K2[0] * x2 / z2 + K2[2], v = K2[1] * y2 / z2 + K2[3];
      const float errX2 = u - kp2.x, errY2 = v - kp2.y;
      if ((errX2 * errX2 + errY2 * errY2) > 5.991 * sigmaSquare2) { status[m] = 6; continue; }
    }
    const float d1[3] = {x3D[0] - Ow1[0], x3D[1] - Ow1[1], x3D[2] - Ow1[2]}, d2[3] = {x3D[0] - Ow2[0], x3D[1] - Ow2[1], x3D[2] - Ow2[2]};
    const float dist1 = std::sqrt(sum3(d1[0] * d1[0], d1[1] * d1[1], d1[2] * d1[2])), dist2 = std::sqrt(sum3(d2[0] * d2[0], d2[1] * d2[1], d2[2] * d2[2]));
    if (dist1 == 0 || dist2 == 0) { status[m] = 7; continue; }
    if (far_points && (dist1 >= th_far || dist2 >= th_far)) { status[m] = 8; continue; }
    const float ratioDist = dist2 / dist1;
    const float ratioOctave = sf1[kp1.octave] / sf2[kp2.octave];
    if (ratioDist * ratio_factor < ratioOctave || ratioDist > ratioOctave * ratio_factor) { status[m] = 9; continue; }
    status[m] = 0;
  }
}

}  // extern "C"
