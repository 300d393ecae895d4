// dvm_slam_amd/csrc/pg_solver.cpp -- host driver of the Sim3 pose-graph optimisation (dvm_pose_graph_optimize).
//
// Mirrors Optimizer::OptimizeEssentialGraph (reference src/Optimizer.cc:1389-1652) between "optimizer.addEdge" and
// "SE3 Pose Recovering": VertexSim3Expmap / EdgeSim3 with identity information, g2o's numeric Jacobians,
// OptimizationAlgorithmLevenberg with setUserLambdaInit(1e-16), optimize(20).  The sparse 7x7-block system goes through
// the same tile Cholesky as the bundle adjustment (nested-dissection order, elimination-tree level schedule; 9 Sim3
// vertices per 64-row tile); the LM control flow runs here, every numerical step is a HIP kernel (ba_kernels.hip).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <vector>

#include "../../include/dvmslam_hip.h"
#include "ba_kernels.h"
#include "ba_ordering.h"
#include "orb_pipeline.h"  // set_error / hip_check / DVM_HIP

using namespace dvm;

namespace {
struct DevPool {
  std::vector<void*> ptrs;
  int rc = DVM_OK;
  template <typename T>
  T* alloc(size_t n) {
    void* p = nullptr;
    if (rc == DVM_OK) rc = hip_check(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)), "hipMalloc(pg)");
    if (p) ptrs.push_back(p);
    return static_cast<T*>(p);
  }
  template <typename T>
  T* upload(const std::vector<T>& v) {
    T* p = alloc<T>(v.size());
    if (rc == DVM_OK && !v.empty()) rc = hip_check(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "upload(pg)");
    return p;
  }
  ~DevPool() { for (void* p : ptrs) hipFree(p); }
};
}  // namespace

extern "C" int dvm_pose_graph_optimize(int device, double* S, const uint8_t* fixed, int n, const dvm_pg_edge* edges, int E,
                                       int fix_scale, int iterations, dvm_pg_stats* st) {
  if (!S || !fixed || !edges || n < 1 || E < 1 || iterations < 0) { set_error("dvm_pose_graph_optimize: bad arguments"); return DVM_ERR_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libdvmslam_hip has no CPU path)"); return DVM_ERR_NO_DEVICE; }
  if (device < 0 || device >= ndev) return DVM_ERR_INVALID;
  DVM_HIP(hipSetDevice(device));
  if (st) std::memset(st, 0, sizeof(*st));
  const auto t_begin = std::chrono::steady_clock::now();
  // ---- structure: free vertices (g2o orders them by id; the fill-reducing order is applied here, ba_ordering.h)
  std::vector<int32_t> nat_of(n, -1), nat_v;
  for (int v = 0; v < n; v++) if (!fixed[v]) { nat_of[v] = (int32_t)nat_v.size(); nat_v.push_back(v); }
  const int nfree = (int)nat_v.size();
  std::vector<int32_t> ev(2 * (size_t)E);
  std::vector<double> emeas(8 * (size_t)E);
  std::vector<std::vector<int>> adj(nfree);
  for (int k = 0; k < E; k++) {
    const int vi = edges[k].vi, vj = edges[k].vj;
    if (vi < 0 || vi >= n || vj < 0 || vj >= n || vi == vj) { set_error("pose graph: edge vertex out of range"); return DVM_ERR_INVALID; }
    ev[2 * k] = vi; ev[2 * k + 1] = vj;
    std::memcpy(&emeas[8 * (size_t)k], edges[k].Sji, 64);
    if (nat_of[vi] >= 0 && nat_of[vj] >= 0) { adj[nat_of[vi]].push_back(nat_of[vj]); adj[nat_of[vj]].push_back(nat_of[vi]); }
  }
  if (nfree == 0) return DVM_OK;
  const std::vector<int> pos = ba_order_cameras(adj, kSim3PerTile);
  std::vector<int32_t> vidx(n, -1), free_v(nfree);
  for (int a = 0; a < nfree; a++) { vidx[nat_v[a]] = pos[a]; free_v[pos[a]] = nat_v[a]; }
  std::map<std::pair<int, int>, std::vector<int32_t>> blocks;
  std::vector<std::vector<int32_t>> vcon(nfree);
  for (int k = 0; k < E; k++) {
    const int pi = vidx[ev[2 * k]], pj = vidx[ev[2 * k + 1]];
    if (pi >= 0) { blocks[{pi, pi}].push_back((k << 2) | 0); vcon[pi].push_back((k << 1) | 0); }
    if (pj >= 0) { blocks[{pj, pj}].push_back((k << 2) | 3); vcon[pj].push_back((k << 1) | 1); }
    if (pi >= 0 && pj >= 0) {
      if (pi > pj) blocks[{pi, pj}].push_back((k << 2) | 1);   // a = i-side (0), b = j-side (1)
      else blocks[{pj, pi}].push_back((k << 2) | 2);           // a = j-side (1), b = i-side (0)
    }
  }
  for (int p = 0; p < nfree; p++) blocks[{p, p}];   // vertices without edges still own a (lambda-only) diagonal block
  std::vector<int32_t> blk_a, blk_b, blk_start{0}, blk_contrib, v_start{0}, v_contrib;
  for (auto& kv : blocks) {
    blk_a.push_back(kv.first.first); blk_b.push_back(kv.first.second);
    blk_contrib.insert(blk_contrib.end(), kv.second.begin(), kv.second.end());
    blk_start.push_back((int32_t)blk_contrib.size());
  }
  for (int p = 0; p < nfree; p++) { v_contrib.insert(v_contrib.end(), vcon[p].begin(), vcon[p].end()); v_start.push_back((int32_t)v_contrib.size()); }
  const int ntv = (nfree + kSim3PerTile - 1) / kSim3PerTile, nkb = ntv + 1;
  std::vector<std::vector<char>> Tp(nkb, std::vector<char>(nkb, 0));
  for (size_t b = 0; b < blk_a.size(); b++) {
    const int tr = blk_a[b] / kSim3PerTile, tc = blk_b[b] / kSim3PerTile;
    Tp[std::max(tr, tc)][std::min(tr, tc)] = 1;
  }
  const BaTileSchedule SC = ba_tile_schedule(Tp);

  DevPool D;
  PgView G{};
  BaView T{};
  G.n = n; G.E = E; G.nfree = nfree; G.fix_scale = fix_scale ? 1 : 0; G.nblk = (int)blk_a.size();
  G.S = D.alloc<double>(8 * (size_t)n);
  double* d_bak = D.alloc<double>(8 * (size_t)n);
  G.vidx = D.upload(vidx); G.free_v = D.upload(free_v); G.ev = D.upload(ev); G.emeas = D.upload(emeas);
  G.e_err = D.alloc<double>(7 * (size_t)E); G.e_J = D.alloc<double>(98 * (size_t)E);
  G.blk_a = D.upload(blk_a); G.blk_b = D.upload(blk_b); G.blk_start = D.upload(blk_start); G.blk_contrib = D.upload(blk_contrib);
  G.v_start = D.upload(v_start); G.v_contrib = D.upload(v_contrib);
  G.bp = D.alloc<double>(7 * (size_t)nfree);
  G.partial = D.alloc<double>((size_t)std::max((E + 255) / 256, (nfree + 255) / 256) + 1);
  T.nfree = nfree; T.per_tile = kSim3PerTile; T.dof = 7; T.n_pad = 64 * ntv; T.ldS = 64 * nkb; T.nlevels = SC.nlevels;
  T.S = D.alloc<double>((size_t)T.ldS * T.ldS); T.Linv = D.alloc<double>((size_t)nkb * 64 * 64);
  T.ytmp = D.alloc<double>((size_t)T.n_pad + 64); T.xrow = D.alloc<double>((size_t)T.n_pad + 64); T.x = D.alloc<double>(7 * (size_t)nfree + 8);
  T.nz_tiles = D.upload(SC.nz_tiles); T.n_nz = (int)(SC.nz_tiles.size() / 2);
  T.cols = D.upload(SC.cols); T.strips = D.upload(SC.strips); T.targets = D.upload(SC.targets); T.contrib = D.upload(SC.contrib);
  T.contrib_strip = nullptr; T.strip_flags = nullptr;   // the pose graph keeps one launch per phase (k_chol_trsm_update needs the caller's retry path)
  T.colstrip_off = D.upload(SC.colstrip_off); T.colstrips = D.upload(SC.colstrips);
  T.h_level_off = SC.level_off.data(); T.h_strip_off = SC.strip_off.data(); T.h_tgt_off = SC.tgt_off.data();
  double* d_scalars = D.alloc<double>(8);
  int* d_fail = D.alloc<int>(1);
  T.lambda = d_scalars + 7;
  if (D.rc != DVM_OK) return D.rc;
  DVM_HIP(hipMemset(T.S, 0, (size_t)T.ldS * T.ldS * sizeof(double)));   // once: trials clear only the non-zero tiles
  DVM_HIP(hipMemset(T.ytmp, 0, ((size_t)T.n_pad + 64) * sizeof(double)));  // ticket + hand-off flags of the back substitution
  int solve_seq = 0;
  // estimates: unit quaternions with w >= 0 like g2o::Sim3's constructor (sim3.h:56-60 normalises r)
  std::vector<double> Sn(S, S + 8 * (size_t)n);
  for (int v = 0; v < n; v++) {
    double* q = &Sn[8 * (size_t)v];
    const double nn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (!(nn > 0)) { set_error("pose graph: zero quaternion"); return DVM_ERR_INVALID; }
    for (int i = 0; i < 4; i++) q[i] /= nn;
  }
  DVM_HIP(hipMemcpy(G.S, Sn.data(), Sn.size() * sizeof(double), hipMemcpyHostToDevice));
  if (st) st->ms_structure = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();

  const auto t0 = std::chrono::steady_clock::now();
  hipStream_t s = nullptr;   // one-shot call: the null stream
  enum { S_CHI = 0, S_TMPCHI = 1, S_SCALE = 2 };
  double hs[8] = {0};
  int hfail = 0;
  auto read = [&]() -> int {
    int rc = hip_check(hipMemcpy(hs, d_scalars, 3 * sizeof(double), hipMemcpyDeviceToHost), "read scalars");
    if (rc == DVM_OK) rc = hip_check(hipMemcpy(&hfail, d_fail, sizeof(int), hipMemcpyDeviceToHost), "read fail flag");
    return rc;
  };
  double lambda = 1e-16, ni = 2;   // setUserLambdaInit(1e-16), Optimizer.cc:1401
  int nBad = 0, it_done = 0, trials = 0, stop = 0;
  double chi_last = 0;
  for (int it = 0; it < iterations; it++) {
    pg_launch_edge_eval(s, G, true, d_scalars, S_CHI);   // computeActiveErrors + linearizeOplus
    int rc = read();
    if (rc != DVM_OK) return rc;
    double currentChi = hs[S_CHI], tempChi = currentChi;
    const double iniChi = currentChi;
    if (it == 0 && st) st->chi2_initial = currentChi;
    double rho = 0;
    int qmax = 0;
    do {
      DVM_HIP(hipMemcpy(d_scalars + 7, &lambda, sizeof(double), hipMemcpyHostToDevice));
      DVM_HIP(hipMemcpyAsync(d_bak, G.S, 8 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));   // push()
      DVM_HIP(hipMemsetAsync(d_fail, 0, sizeof(int), s));
      pg_launch_build(s, G, T);
      ba_launch_cholesky_solve(s, T, d_fail, ++solve_seq);
      pg_launch_update(s, G, T, d_scalars, S_SCALE);
      pg_launch_edge_eval(s, G, false, d_scalars, S_TMPCHI);
      rc = read();
      if (rc != DVM_OK) return rc;
      const bool ok2 = (hfail == 0);
      tempChi = ok2 ? hs[S_TMPCHI] : std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = ok2 ? hs[S_SCALE] : 0.0;
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow(2 * rho - 1, 3);
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni; ni *= 2;
        DVM_HIP(hipMemcpyAsync(G.S, d_bak, 8 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));   // pop()
      }
      qmax++; trials++;
    } while (rho < 0 && qmax < 10);
    if (st && it < 32) { st->chi2_per_iter[it] = currentChi; st->trials_per_iter[it] = qmax; }
    it_done++;
    chi_last = currentChi;
    if (qmax == 10 || rho == 0) { stop = 1; break; }
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) { stop = 2; break; }
  }
  DVM_HIP(hipDeviceSynchronize());
  DVM_HIP(hipMemcpy(S, G.S, 8 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  if (st) {
    st->iterations = it_done; st->total_trials = trials; st->stop_reason = stop; st->chi2_final = chi_last; st->lambda_final = lambda;
    st->levels = SC.nlevels; st->tile_fill = SC.fill;
    st->ms_optimize = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  return DVM_OK;
}
