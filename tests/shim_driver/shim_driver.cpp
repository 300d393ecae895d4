// tests/shim_driver/shim_driver.cpp -- TEST INFRASTRUCTURE: links the drop-in shims of dvm_slam_amd/host/ (the code a
// maintainer compiles into the reference tree) against the behaving mock ORB_SLAM3 classes of tests/stubs/ and exposes a flat
// C interface, so that tests/test_gpu_shims_run.py can build a synthetic map, RUN ORB_SLAM3::Optimizer::* / ORBmatcher::* /
// ORBextractor::operator() / Frame::isInFrustum exactly as Tracking / LocalMapping / LoopClosing would call them, and read the
// map back.  Everything numerical happens in libdvmslam_hip.so / libdvmslam_host.so; this file only moves data in and out of
// the mock objects.  Indices handed over this interface are positions in the world's keyframe / map point / frame tables.
#include <cstdint>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <vector>

#include "ORBextractor_shim.h"
#include "ORBmatcher_shim.h"
#include "Optimizer_shim.h"
#include "Sim3Solver_shim.h"
#include "ORBVocabulary_shim.h"
#include "KeyFrameDatabase_shim.h"
#include "MapPoint_shim.h"
#include "LocalMapping_shim.h"
#include "Frame_grid_shim.h"

using namespace ORB_SLAM3;

float Frame::fx, Frame::fy, Frame::cx, Frame::cy, Frame::invfx, Frame::invfy;
float Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv;
float Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY;

namespace {
struct World {
  std::vector<std::unique_ptr<Map>> maps;
  std::vector<std::unique_ptr<KeyFrame>> kfs;
  std::vector<std::unique_ptr<MapPoint>> mps;
  std::vector<std::unique_ptr<Frame>> frames;
  GeometricCamera cam;
  std::unique_ptr<KeyFrameDatabase> kfdb;
  std::unique_ptr<ORBextractor> extractor;           // Tracking's mpORBextractorLeft: one instance, used frame after frame
  bool last_grid_from_device = false;
  std::string error;
  int kf_index(KeyFrame* k) const { for (size_t i = 0; i < kfs.size(); i++) if (kfs[i].get() == k) return (int)i; return -1; }
  int mp_index(MapPoint* p) const { if (!p) return -1; for (size_t i = 0; i < mps.size(); i++) if (mps[i].get() == p) return (int)i; return -2; }
};
// pose7 = (qx, qy, qz, qw, tx, ty, tz): the members of a Sophus::SE3f as stored
Sophus::SE3f se3_of(const float* p) {
  Eigen::Vector3f t;
  t(0) = p[4]; t(1) = p[5]; t(2) = p[6];
  return Sophus::SE3f::raw(Eigen::Quaternionf(p[3], p[0], p[1], p[2]), t);
}
void se3_to(const Sophus::SE3f& T, float* p) {
  for (int i = 0; i < 4; i++) p[i] = T.unit_quaternion().coeffs()(i);
  for (int i = 0; i < 3; i++) p[4 + i] = T.translation()(i);
}
Sophus::Sim3f sim3_of(const float* p) {   // (qx, qy, qz, qw with |q|^2 = scale, tx, ty, tz)
  Eigen::Vector3f t;
  t(0) = p[4]; t(1) = p[5]; t(2) = p[6];
  return Sophus::Sim3f(Sophus::RxSO3f(Eigen::Quaternionf(p[3], p[0], p[1], p[2])), t);
}
template <class F> int guarded(World* w, F&& f) {
  try { return f(); }
  catch (const std::exception& e) { w->error = e.what(); return -1000; }
}
std::vector<KeyFrame*> kf_list(World* w, const int32_t* idx, int n) { std::vector<KeyFrame*> v; for (int i = 0; i < n; i++) v.push_back(w->kfs[idx[i]].get()); return v; }
std::vector<MapPoint*> mp_list(World* w, const int32_t* idx, int n) { std::vector<MapPoint*> v; for (int i = 0; i < n; i++) v.push_back(idx[i] < 0 ? nullptr : w->mps[idx[i]].get()); return v; }
}  // namespace

extern "C" {

World* sw_create() { return new World; }
void sw_destroy(World* w) { delete w; }
const char* sw_error(World* w) { return w->error.c_str(); }
void sw_set_device(int d) { dvm_host::set_device(d); }

int sw_add_map(World* w, unsigned long init_kf_id) {
  w->maps.emplace_back(new Map);
  w->maps.back()->mock_init_kf_id = init_kf_id;
  return (int)w->maps.size() - 1;
}
void sw_map_set_origin(World* w, int map, int kf) { w->maps[map]->mock_origin = w->kfs[kf].get(); }
int sw_map_change_index(World* w, int map) { return w->maps[map]->mock_change_index; }
int sw_map_opt_fixed(World* w, int map, unsigned long* opt, unsigned long* fixed, int cap, int32_t* n_fixed) {
  int n = 0, m = 0;
  for (unsigned long id : w->maps[map]->msOptKFs) if (n < cap) opt[n++] = id;
  for (unsigned long id : w->maps[map]->msFixedKFs) if (m < cap) fixed[m++] = id;
  *n_fixed = m;
  return n;
}

// K4 = fx fy cx cy; tables = nlevels x (scale factor, level sigma2, inverse level sigma2) as three arrays; bounds = minX minY maxX maxY
int sw_add_keyframe(World* w, int map, unsigned long id, const float* pose7, const float* pose7_inv, const float* K4, int N, const dvm_keypoint* kps,
                    const uint8_t* desc, const float* scale, const float* sigma2, const float* inv_sigma2, int nlevels, float log_scale,
                    const int32_t* bounds, int bad) {
  KeyFrameInit s;
  s.id = id; s.fx = K4[0]; s.fy = K4[1]; s.cx = K4[2]; s.cy = K4[3];
  s.keysUn.resize(N);
  static_assert(sizeof(cv::KeyPoint) == sizeof(dvm_keypoint), "cv::KeyPoint layout");
  if (N) std::memcpy(static_cast<void*>(s.keysUn.data()), kps, sizeof(dvm_keypoint) * (size_t)N);
  s.descriptors.create(std::max(N, 1), 32, CV_8U);
  if (desc && N) std::memcpy(s.descriptors.data, desc, 32 * (size_t)N); else std::memset(s.descriptors.data, 0, 32 * (size_t)std::max(N, 1));
  s.scaleFactors.assign(scale, scale + nlevels); s.levelSigma2.assign(sigma2, sigma2 + nlevels); s.invLevelSigma2.assign(inv_sigma2, inv_sigma2 + nlevels);
  s.logScaleFactor = log_scale;
  if (bounds) { s.minX = bounds[0]; s.minY = bounds[1]; s.maxX = bounds[2]; s.maxY = bounds[3]; }
  Map* m = w->maps[map].get();
  w->kfs.emplace_back(new KeyFrame(s, m));
  KeyFrame* kf = w->kfs.back().get();
  if (pose7_inv) kf->mock_pose(se3_of(pose7), se3_of(pose7_inv));
  else { kf->SetPose(se3_of(pose7)); kf->mock_set_pose = 0; }
  kf->mbBad = bad != 0;
  for (int i = 0; i < 4; i++) w->cam.p_[i] = K4[i];      // Pinhole::mvParameters = (fx, fy, cx, cy)
  kf->mpCamera = &w->cam;
  m->mock_kfs.push_back(kf);
  m->mock_max_kf_id = std::max(m->mock_max_kf_id, id);
  return (int)w->kfs.size() - 1;
}
void sw_kf_set_feature_vector(World* w, int kf, int n, const int32_t* node, const int32_t* off, const int32_t* feat) {
  DBoW2::FeatureVector& fv = w->kfs[kf]->mFeatVec;
  fv.clear();
  for (int k = 0; k < n; k++) {
    std::vector<unsigned int>& v = fv[(DBoW2::NodeId)node[k]];
    for (int i = off[k]; i < off[k + 1]; i++) v.push_back((unsigned)feat[i]);
  }
}
int sw_add_mappoint(World* w, int map, unsigned long id, const float* xyz, const float* normal, float min_dist, float max_dist, const uint8_t* desc, int bad) {
  Eigen::Vector3f X;
  for (int k = 0; k < 3; k++) X(k) = xyz[k];
  w->mps.emplace_back(new MapPoint(id, X, w->maps[map].get()));
  MapPoint* p = w->mps.back().get();
  if (normal) for (int k = 0; k < 3; k++) p->mNormalVector(k) = normal[k];
  p->mfMinDistance = min_dist; p->mfMaxDistance = max_dist;
  if (desc) std::memcpy(MapPoint::MockAccess::descriptor(p).data, desc, 32);
  MapPoint::MockAccess::bad(p) = bad != 0;
  w->maps[map]->mock_mps.push_back(p);
  return (int)w->mps.size() - 1;
}
// the keyframe observes the point at keypoint idx (MapPoint::AddObservation + KeyFrame::AddMapPoint, as LocalMapping does)
void sw_observe(World* w, int kf, int mp, int idx) {
  MapPoint* p = w->mps[mp].get();
  KeyFrame* k = w->kfs[kf].get();
  p->AddObservation(k, idx);
  k->AddMapPoint(p, idx);
  if (!p->mpRefKF) p->mpRefKF = k;
}
void sw_kf_set_match(World* w, int kf, int idx, int mp) { w->kfs[kf]->mvpMapPoints[idx] = mp < 0 ? nullptr : w->mps[mp].get(); }   // a match without an observation
void sw_mp_set_obs_count(World* w, int mp, int n) { w->mps[mp]->nObs = n; }
void sw_mp_set_ref(World* w, int mp, int kf) { w->mps[mp]->mpRefKF = kf < 0 ? nullptr : w->kfs[kf].get(); }
void sw_mp_set_corrected(World* w, int mp, unsigned long by_kf, unsigned long reference) { w->mps[mp]->mnCorrectedByKF = by_kf; w->mps[mp]->mnCorrectedReference = reference; }
void sw_set_covisible(World* w, int kf, const int32_t* others, const int32_t* weights, int n) {
  KeyFrame* k = w->kfs[kf].get();
  k->mvpOrderedConnectedKeyFrames = kf_list(w, others, n);
  k->mvOrderedWeights.assign(weights, weights + n);
}
void sw_set_parent(World* w, int kf, int parent) {
  w->kfs[kf]->mpParent = w->kfs[parent].get();
  w->kfs[parent]->mspChildrens.insert(w->kfs[kf].get());
}
void sw_add_loop_edge(World* w, int a, int b) { w->kfs[a]->mspLoopEdges.insert(w->kfs[b].get()); w->kfs[b]->mspLoopEdges.insert(w->kfs[a].get()); }
void sw_kf_set_bef_merge(World* w, int kf, const float* Tcw7, const float* Twc7) { w->kfs[kf]->mTcwBefMerge = se3_of(Tcw7); w->kfs[kf]->mTwcBefMerge = se3_of(Twc7); }

// ---- read back
void sw_get_kf(World* w, int kf, float* pose7, float* gba7, float* bef_merge7, int32_t* info /* SetPose calls, mnBAGlobalForKF */) {
  KeyFrame* k = w->kfs[kf].get();
  if (pose7) se3_to(k->GetPose(), pose7);
  if (gba7) se3_to(k->mTcwGBA, gba7);
  if (bef_merge7) se3_to(k->mTcwBefMerge, bef_merge7);
  if (info) { info[0] = k->mock_set_pose; info[1] = (int32_t)k->mnBAGlobalForKF; }
}
void sw_get_mp(World* w, int mp, float* xyz, float* gba_xyz, int32_t* info /* bad, SetWorldPos calls, UpdateNormalAndDepth calls, nObs, mnBAGlobalForKF, replaced-by */) {
  MapPoint* p = w->mps[mp].get();
  if (xyz) for (int k = 0; k < 3; k++) xyz[k] = p->mWorldPos(k);
  if (gba_xyz) for (int k = 0; k < 3; k++) gba_xyz[k] = p->mPosGBA(k);
  if (info) { info[0] = MapPoint::MockAccess::bad(p); info[1] = p->mock_set_pos; info[2] = p->mock_update_normal; info[3] = p->nObs; info[4] = (int32_t)p->mnBAGlobalForKF; info[5] = w->mp_index(p->mpReplaced); }
}
int sw_get_kf_matches(World* w, int kf, int32_t* out) {
  KeyFrame* k = w->kfs[kf].get();
  for (int i = 0; i < k->N; i++) out[i] = w->mp_index(k->mvpMapPoints[i]);
  return k->N;
}
int sw_get_mp_observations(World* w, int mp, int32_t* kf_out, int32_t* idx_out, int cap) {
  int n = 0;
  for (const auto& o : MapPoint::MockAccess::observations(w->mps[mp].get())) {
    if (n < cap) { kf_out[n] = w->kf_index(o.first); idx_out[n] = std::get<0>(o.second); }
    n++;
  }
  return n;
}

// ---- Optimizer
// register / drop the shared dvm_ba_pool of LocalBundleAdjustment (several agents on one GPU: Optimizer_shim.h)
int sw_set_local_ba_pool(int on) {
  static dvm_ba_pool* pool = nullptr;
  if (on && !pool && dvm_ba_pool_create(dvm_host::device(), 8, 100, &pool) != 0) return -1;
  ORB_SLAM3::dvm_optimizer_detail::set_local_ba_pool(on ? pool : nullptr);
  return 0;
}
int sw_local_ba(World* w, int kf, int map, uint8_t* stop, int32_t* counts4) {
  return guarded(w, [&] {
    int a = -1, b = -1, c = -1, d = -1;
    Optimizer::LocalBundleAdjustment(w->kfs[kf].get(), reinterpret_cast<bool*>(stop), w->maps[map].get(), a, b, c, d);
    counts4[0] = a; counts4[1] = b; counts4[2] = c; counts4[3] = d;
    return 0;
  });
}
int sw_global_ba(World* w, int map, int iterations, unsigned long loop_kf, int robust) {
  return guarded(w, [&] { Optimizer::GlobalBundleAdjustemnt(w->maps[map].get(), iterations, nullptr, loop_kf, robust != 0); return 0; });
}
int sw_welding_ba(World* w, int main_kf, const int32_t* adjust, int n_adjust, const int32_t* fixed, int n_fixed, uint8_t* stop) {
  return guarded(w, [&] {
    Optimizer::LocalBundleAdjustment(w->kfs[main_kf].get(), kf_list(w, adjust, n_adjust), kf_list(w, fixed, n_fixed), reinterpret_cast<bool*>(stop));
    return 0;
  });
}
int sw_essential_graph_merge(World* w, int cur_kf, const int32_t* fixed, int n_fixed, const int32_t* fixed_corrected, int n_fc, const int32_t* non_fixed,
                             int n_nf, const int32_t* mps, int n_mps) {
  return guarded(w, [&] {
    std::vector<KeyFrame*> a = kf_list(w, fixed, n_fixed), b = kf_list(w, fixed_corrected, n_fc), c = kf_list(w, non_fixed, n_nf);
    std::vector<MapPoint*> p = mp_list(w, mps, n_mps);
    Optimizer::OptimizeEssentialGraph(w->kfs[cur_kf].get(), a, b, c, p);
    return 0;
  });
}
// Sim3 entries: 8 doubles (qx qy qz qw tx ty tz s) per listed keyframe; connections: (kf, kf) pairs
int sw_essential_graph_loop(World* w, int map, int loop_kf, int cur_kf, const int32_t* nc_kf, const double* nc_sim3, int n_nc, const int32_t* c_kf,
                            const double* c_sim3, int n_c, const int32_t* conn_pairs, int n_conn, int fix_scale) {
  return guarded(w, [&] {
    const auto sim3 = [](const double* s) {
      Eigen::Vector3d t;
      t(0) = s[4]; t(1) = s[5]; t(2) = s[6];
      return g2o::Sim3(Eigen::Quaterniond(s[3], s[0], s[1], s[2]), t, s[7]);
    };
    LoopClosing::KeyFrameAndPose nc, c;
    for (int i = 0; i < n_nc; i++) nc[w->kfs[nc_kf[i]].get()] = sim3(nc_sim3 + 8 * i);
    for (int i = 0; i < n_c; i++) c[w->kfs[c_kf[i]].get()] = sim3(c_sim3 + 8 * i);
    std::map<KeyFrame*, std::set<KeyFrame*>> conn;
    for (int i = 0; i < n_conn; i++) conn[w->kfs[conn_pairs[2 * i]].get()].insert(w->kfs[conn_pairs[2 * i + 1]].get());
    const bool fs = fix_scale != 0;
    Optimizer::OptimizeEssentialGraph(w->maps[map].get(), w->kfs[loop_kf].get(), w->kfs[cur_kf].get(), nc, c, conn, fs);
    return 0;
  });
}
// matches1: per keypoint of kf1 the matched map point of kf2 (index, -1 none), in / out; S12: 8 doubles in / out
int sw_optimize_sim3(World* w, int kf1, int kf2, int32_t* matches1, double* S12, float th2, int fix_scale, int all_points) {
  return guarded(w, [&] {
    KeyFrame* k1 = w->kfs[kf1].get();
    std::vector<MapPoint*> m = mp_list(w, matches1, k1->N);
    Eigen::Vector3d t;
    t(0) = S12[4]; t(1) = S12[5]; t(2) = S12[6];
    g2o::Sim3 S(Eigen::Quaterniond(S12[3], S12[0], S12[1], S12[2]), t, S12[7]);
    Eigen::Matrix<double, 7, 7> H;
    const int n = Optimizer::OptimizeSim3(k1, w->kfs[kf2].get(), m, S, th2, fix_scale != 0, H, all_points != 0);
    for (int i = 0; i < k1->N; i++) matches1[i] = w->mp_index(m[i]);
    S12[0] = S.rotation().x(); S12[1] = S.rotation().y(); S12[2] = S.rotation().z(); S12[3] = S.rotation().w();
    for (int k = 0; k < 3; k++) S12[4 + k] = S.translation()(k);
    S12[7] = S.scale();
    return n;
  });
}

// 1 = DUtils::Random comes from oracle/_ref/libdutils_ref.so (the reference's Random.cpp), 0 = from the stub header
int sw_dutils_is_reference() {
#ifdef DVM_REF_DUTILS
  return 1;
#else
  return 0;
#endif
}

// Sim3Solver (LoopClosing.cc: `Sim3Solver solver(pKF, pKFi, vpMatches, fixScale, vpMatchedKF); solver.SetRansacParameters(0.99, min, max);
// while (!converged && !noMore) T = solver.iterate(nPerCall, noMore, inliers, nInliers, converged);`) after srand(seed).
// matches12: per keypoint of kf1 the matched map point (index, -1 none).  out16: the returned T12 (row-major 4x4); est: the getters
// {R (9), t (3), s}; info: {calls, converged, noMore, nInliers, N}
int sw_sim3_solver(World* w, int kf1, int kf2, const int32_t* matches12, int fix_scale, int min_inliers, int max_its, int per_call, unsigned seed,
                   float* out16, float* est13, uint8_t* inliers, int32_t* info) {
  return guarded(w, [&] {
    KeyFrame* k1 = w->kfs[kf1].get();
    KeyFrame* k2 = w->kfs[kf2].get();
    std::vector<MapPoint*> m = mp_list(w, matches12, k1->N);
    srand(seed);
    Sim3Solver solver(k1, k2, m, fix_scale != 0);
    solver.SetRansacParameters(0.99, min_inliers, max_its);
    bool noMore = false, conv = false;
    std::vector<bool> vin;
    int nin = 0, calls = 0;
    Eigen::Matrix4f T = Eigen::Matrix4f::Identity();
    while (!conv && !noMore) { T = solver.iterate(per_call, noMore, vin, nin, conv); calls++; }
    for (int i = 0; i < 16; i++) out16[i] = T(i / 4, i % 4);
    const Eigen::Matrix3f R = solver.GetEstimatedRotation();
    const Eigen::Vector3f t = solver.GetEstimatedTranslation();
    for (int i = 0; i < 9; i++) est13[i] = R(i / 3, i % 3);
    for (int i = 0; i < 3; i++) est13[9 + i] = t(i);
    est13[12] = solver.GetEstimatedScale();
    for (int i = 0; i < k1->N; i++) inliers[i] = i < (int)vin.size() && vin[i];
    info[0] = calls; info[1] = conv; info[2] = noMore; info[3] = nin;
    return 0;
  });
}

// ORBVocabulary: loadFromTextFile, then what Frame::ComputeBoW / KeyFrame::ComputeBoW do (Frame.cc:783-789): the descriptor rows as a
// vector<cv::Mat>, transform(.., 4) -> the two maps flattened for the caller; score of the vector against itself's half
int sw_vocab_compute_bow(World* w, const char* path, const uint8_t* desc, int n, int levelsup, int32_t* bow_ids, double* bow_vals, int32_t* n_bow,
                         int32_t* fv_nodes, int32_t* fv_off, int32_t* fv_feat, int32_t* n_fv, int32_t* voc_size, double* self_score) {
  return guarded(w, [&] {
    ORBVocabulary voc;
    if (!voc.loadFromTextFile(path)) return -2;
    *voc_size = (int32_t)voc.size();
    cv::Mat D;
    D.create(std::max(n, 1), 32, CV_8U);
    if (n) std::memcpy(D.data, desc, 32 * (size_t)n);
    std::vector<cv::Mat> vCurrentDesc;
    for (int i = 0; i < n; i++) vCurrentDesc.push_back(D.row(i));
    DBoW2::BowVector bv;
    DBoW2::FeatureVector fv;
    voc.transform(vCurrentDesc, bv, fv, levelsup);
    int k = 0;
    for (const auto& e : bv) { bow_ids[k] = (int32_t)e.first; bow_vals[k] = e.second; k++; }
    *n_bow = k;
    int m = 0, t = 0;
    fv_off[0] = 0;
    for (const auto& e : fv) { fv_nodes[m] = (int32_t)e.first; for (unsigned i : e.second) fv_feat[t++] = (int32_t)i; fv_off[++m] = t; }
    *n_fv = m;
    *self_score = voc.score(bv, bv);
    return 0;
  });
}

// MapPoint::ComputeDistinctiveDescriptors: one by one (the member) or all of `mps` in one call (the batched form); out = 32 B per point
int sw_compute_distinctive(World* w, const int32_t* mps, int n, int batched, uint8_t* out) {
  return guarded(w, [&] {
    std::vector<MapPoint*> v = mp_list(w, mps, n);
    if (batched) MapPoint::ComputeDistinctiveDescriptorsBatch(v);
    else for (MapPoint* p : v) p->ComputeDistinctiveDescriptors();
    for (int i = 0; i < n; i++) { const cv::Mat d = v[i]->GetDescriptor(); std::memcpy(out + 32 * (size_t)i, d.data, 32); }
    return 0;
  });
}

// ---- KeyFrameDatabase (the class, on KeyFrame* / Map* / Frame*)
void sw_kf_set_bow(World* w, int kf, const int32_t* ids, const double* vals, int n) {
  DBoW2::BowVector& b = w->kfs[kf]->mBowVec;
  b.clear();
  for (int i = 0; i < n; i++) b[(DBoW2::WordId)ids[i]] = vals[i];
}
void sw_kf_set_uuid(World* w, int kf, const uint8_t* u16) { std::memcpy(w->kfs[kf]->uuid.data, u16, 16); }
void sw_kf_set_connected(World* w, int kf, const int32_t* others, int n) {
  w->kfs[kf]->mock_connected.clear();
  for (int i = 0; i < n; i++) w->kfs[kf]->mock_connected.insert(w->kfs[others[i]].get());
}
void sw_kf_set_bad(World* w, int kf, int bad) { w->kfs[kf]->mbBad = bad != 0; }
void sw_kf_update_map(World* w, int kf, int map) { w->kfs[kf]->UpdateMap(w->maps[map].get()); }    // KeyFrame::UpdateMap, as LoopClosing::MergeLocal calls it
void sw_map_set_bad(World* w, int map, int bad) { w->maps[map]->mock_bad = bad != 0; }
// test probe: the class keeps its mirror protected; the per-keyframe query state (mnPlaceRecognitionQuery / Words / Score and the relocalisation
// triple, which the reference keeps in public KeyFrame members) is read through a derived class
struct KfdbProbe : KeyFrameDatabase {
  bool state(KeyFrame* k, int reloc, uint64_t* q, int32_t* wd, float* sc) {
    const auto it = slot_of_.find(k);
    if (it == slot_of_.end()) return false;
    const dvm_host::KeyFrameDatabase::State s = reloc ? db_->GetRelocState(it->second) : db_->GetState(it->second);
    *q = s.query; *wd = s.words; *sc = s.score;
    return true;
  }
};
int sw_kfdb_create(World* w) { return guarded(w, [&] { w->kfdb.reset(new KfdbProbe()); return 0; }); }
int sw_kfdb_get_state(World* w, int kf, int reloc, uint64_t* query, int32_t* words, float* score) {
  return static_cast<KfdbProbe*>(w->kfdb.get())->state(w->kfs[kf].get(), reloc, query, words, score) ? 1 : 0;
}
int sw_kfdb_add(World* w, int kf) { return guarded(w, [&] { w->kfdb->add(w->kfs[kf].get()); return 0; }); }
int sw_kfdb_erase(World* w, int kf) { return guarded(w, [&] { w->kfdb->erase(w->kfs[kf].get()); return 0; }); }
int sw_kfdb_clear_map(World* w, int map) { return guarded(w, [&] { w->kfdb->clearMap(w->maps[map].get()); return 0; }); }
// returns found (0 / 1); best_kf = index of the keyframe whose uuid comes back (-1: nil uuid)
int sw_kfdb_detect_merge_possibility(World* w, const int32_t* ids, const double* vals, int n, const uint8_t* u16, int map, int32_t* best_kf) {
  return guarded(w, [&] {
    DBoW2::BowVector b;
    for (int i = 0; i < n; i++) b[(DBoW2::WordId)ids[i]] = vals[i];
    boost::uuids::uuid u;
    std::memcpy(u.data, u16, 16);
    const std::pair<bool, boost::uuids::uuid> r = w->kfdb->DetectMergePossibility(b, u, w->maps[map].get());
    *best_kf = w->kf_index(w->kfdb->ConvertUuidToKeyFrame(r.second));
    return r.first ? 1 : 0;
  });
}
int sw_kfdb_merge_score(World* w, const int32_t* ids, const double* vals, int n, const uint8_t* u16, int map, float* score, int32_t* best_kf) {
  return guarded(w, [&] {
    DBoW2::BowVector b;
    for (int i = 0; i < n; i++) b[(DBoW2::WordId)ids[i]] = vals[i];
    boost::uuids::uuid u;
    std::memcpy(u.data, u16, 16);
    KeyFrame* best = nullptr;
    w->kfdb->CalculateMergeScore(b, u, w->maps[map].get(), *score, best);
    *best_kf = w->kf_index(best);
    return 0;
  });
}
int sw_kfdb_detect_n_best(World* w, int kf, int n_num, int32_t* loop, int32_t* n_loop, int32_t* merge, int32_t* n_merge) {
  return guarded(w, [&] {
    std::vector<KeyFrame*> l, m;
    w->kfdb->DetectNBestCandidates(w->kfs[kf].get(), l, m, n_num);
    *n_loop = (int32_t)l.size(); *n_merge = (int32_t)m.size();
    for (size_t i = 0; i < l.size(); i++) loop[i] = w->kf_index(l[i]);
    for (size_t i = 0; i < m.size(); i++) merge[i] = w->kf_index(m[i]);
    return 0;
  });
}
int sw_kfdb_detect_reloc(World* w, const int32_t* ids, const double* vals, int n, unsigned long frame_id, int map, int32_t* out, int32_t* n_out) {
  return guarded(w, [&] {
    Frame F;
    for (int i = 0; i < n; i++) F.mBowVec[(DBoW2::WordId)ids[i]] = vals[i];
    F.mnId = frame_id;
    const std::vector<KeyFrame*> c = w->kfdb->DetectRelocalizationCandidates(&F, w->maps[map].get());
    *n_out = (int32_t)c.size();
    for (size_t i = 0; i < c.size(); i++) out[i] = w->kf_index(c[i]);
    return 0;
  });
}

// ---- frames
int sw_add_frame(World* w, const float* pose7, const float* K4, int N, const dvm_keypoint* kps, const uint8_t* desc, const float* scale, const float* sigma2,
                 const float* inv_sigma2, int nlevels, float log_scale, const float* bounds /* minX maxX minY maxY */) {
  w->frames.emplace_back(new Frame);
  Frame& F = *w->frames.back();
  F.N = N;
  F.mvKeysUn.resize(N);
  if (N) std::memcpy(static_cast<void*>(F.mvKeysUn.data()), kps, sizeof(dvm_keypoint) * (size_t)N);
  F.mvKeys = F.mvKeysUn;
  F.mvuRight.assign(N, -1.0f); F.mvDepth.assign(N, -1.0f);
  F.mDescriptors.create(std::max(N, 1), 32, CV_8U);
  if (desc && N) std::memcpy(F.mDescriptors.data, desc, 32 * (size_t)N);
  F.mvpMapPoints.assign(N, static_cast<MapPoint*>(nullptr));
  F.mvbOutlier.assign(N, false);
  Frame::fx = K4[0]; Frame::fy = K4[1]; Frame::cx = K4[2]; Frame::cy = K4[3]; Frame::invfx = 1.0f / K4[0]; Frame::invfy = 1.0f / K4[1];
  F.mnScaleLevels = nlevels; F.mfScaleFactor = nlevels > 1 ? scale[1] : 1.2f; F.mfLogScaleFactor = log_scale;
  F.mvScaleFactors.assign(scale, scale + nlevels); F.mvLevelSigma2.assign(sigma2, sigma2 + nlevels); F.mvInvLevelSigma2.assign(inv_sigma2, inv_sigma2 + nlevels);
  F.mvInvScaleFactors.resize(nlevels);
  for (int l = 0; l < nlevels; l++) F.mvInvScaleFactors[l] = 1.0f / scale[l];
  if (bounds) { Frame::mnMinX = bounds[0]; Frame::mnMaxX = bounds[1]; Frame::mnMinY = bounds[2]; Frame::mnMaxY = bounds[3]; }
  for (int i = 0; i < 4; i++) w->cam.p_[i] = K4[i];
  F.mpCamera = &w->cam;
  F.SetPose(se3_of(pose7));
  F.mock_set_pose = 0;
  return (int)w->frames.size() - 1;
}
// the frame's pose members exactly as given (mRcw row-major, mtcw, mOw): tests hand over the matrices the oracle was given
void sw_frame_set_pose_matrices(World* w, int f, const float* R9, const float* t3, const float* Ow3) {
  Frame& F = *w->frames[f];
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) F.mRcw(r, c) = R9[3 * r + c]; F.mtcw(r) = t3[r]; F.mOw(r) = Ow3[r]; }
}
void sw_frame_set_matches(World* w, int f, const int32_t* mp, const uint8_t* outlier) {
  Frame& F = *w->frames[f];
  for (int i = 0; i < F.N; i++) { F.mvpMapPoints[i] = mp[i] < 0 ? nullptr : w->mps[mp[i]].get(); if (outlier) F.mvbOutlier[i] = outlier[i] != 0; }
}
void sw_frame_set_feature_vector(World* w, int f, int n, const int32_t* node, const int32_t* off, const int32_t* feat) {
  DBoW2::FeatureVector& fv = w->frames[f]->mFeatVec;
  fv.clear();
  for (int k = 0; k < n; k++) {
    std::vector<unsigned int>& v = fv[(DBoW2::NodeId)node[k]];
    for (int i = off[k]; i < off[k + 1]; i++) v.push_back((unsigned)feat[i]);
  }
}
int sw_get_frame(World* w, int f, float* pose7, int32_t* mp, uint8_t* outlier) {
  Frame& F = *w->frames[f];
  if (pose7) se3_to(F.GetPose(), pose7);
  for (int i = 0; i < F.N; i++) { if (mp) mp[i] = w->mp_index(F.mvpMapPoints[i]); if (outlier) outlier[i] = F.mvbOutlier[i]; }
  return F.mock_set_pose;
}
int sw_pose_optimization(World* w, int f) {
  return guarded(w, [&] { return Optimizer::PoseOptimization(w->frames[f].get()); });
}
// Frame::isInFrustum point by point (the reference's own call pattern, Tracking.cc:3041-3103) or batched; track: 6 floats + 2 ints per point
int sw_is_in_frustum(World* w, int f, const int32_t* mps, int n, float cos_limit, int batched, float* track_f /* n x 5 */, int32_t* track_i /* n x 2 */) {
  return guarded(w, [&] {
    std::vector<MapPoint*> p = mp_list(w, mps, n);
    int nin = 0;
    if (batched) nin = Frame_isInFrustumBatch(*w->frames[f], p, cos_limit);
    else for (MapPoint* q : p) nin += w->frames[f]->isInFrustum(q, cos_limit) ? 1 : 0;
    for (int i = 0; i < n; i++) {
      track_f[5 * i] = p[i]->mTrackProjX; track_f[5 * i + 1] = p[i]->mTrackProjY; track_f[5 * i + 2] = p[i]->mTrackDepth;
      track_f[5 * i + 3] = p[i]->mTrackViewCos; track_f[5 * i + 4] = p[i]->mTrackProjXR;
      track_i[2 * i] = p[i]->mbTrackInView; track_i[2 * i + 1] = p[i]->mnTrackScaleLevel;
    }
    return nin;
  });
}
void sw_mp_set_track(World* w, int mp, float px, float py, float depth, float view_cos, int level, int in_view) {
  MapPoint* p = w->mps[mp].get();
  p->mTrackProjX = px; p->mTrackProjY = py; p->mTrackDepth = depth; p->mTrackViewCos = view_cos; p->mnTrackScaleLevel = level; p->mbTrackInView = in_view != 0;
}

// ---- ORBmatcher
int sw_search_by_projection_last(World* w, int cur, int last, float th, float nnratio, int check_ori) {
  return guarded(w, [&] {
    ORBmatcher m(nnratio, check_ori != 0);
    const int n = m.SearchByProjection(*w->frames[cur], *w->frames[last], th, true);
    w->last_grid_from_device = m.dvmLastGridFromDevice();
    return n;
  });
}
int sw_last_grid_from_device(World* w) { return w->last_grid_from_device ? 1 : 0; }
// Frame::ExtractORB (Frame.cc:411) on the World's persistent extractor: (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors, vLapping), then
// what the constructor derives from it (N, mvKeysUn = mvKeys for an undistorted camera, the per-keypoint vectors) and -- keep_device --
// the optional added line `mDvmDevice = mpORBextractorLeft->LastDeviceResult()`
int sw_frame_extract(World* w, int f, const uint8_t* img, int rows, int cols, int stride, int keep_device) {
  return guarded(w, [&] {
    if (!w->extractor) w->extractor.reset(new ORBextractor(1000, 1.2f, 8, 20, 7));
    Frame& F = *w->frames[f];
    cv::Mat image(rows, cols, CV_8UC1, const_cast<uint8_t*>(img), (size_t)stride);
    std::vector<int> lap = {0, 1000};
    cv::_InputArray in(image), mask;
    cv::_OutputArray out(F.mDescriptors);
    (*w->extractor)(in, mask, F.mvKeys, out, lap);
    F.N = (int)F.mvKeys.size();
    F.mvKeysUn = F.mvKeys;
    F.mvuRight.assign(F.N, -1.0f); F.mvDepth.assign(F.N, -1.0f);
    F.mvpMapPoints.assign(F.N, static_cast<MapPoint*>(nullptr));
    F.mvbOutlier.assign(F.N, false);
    F.mDvmDevice = keep_device ? w->extractor->LastDeviceResult() : dvm_device_frame{};
    return F.N;
  });
}
int sw_frame_keypoints(World* w, int f, dvm_keypoint* kps, uint8_t* desc) {
  Frame& F = *w->frames[f];
  if (F.N) std::memcpy(kps, static_cast<const void*>(F.mvKeysUn.data()), sizeof(dvm_keypoint) * (size_t)F.N);
  for (int r = 0; r < F.N; r++) std::memcpy(desc + 32 * (size_t)r, F.mDescriptors.ptr<uint8_t>(r), 32);
  return F.N;
}
int sw_search_by_projection_points(World* w, int f, const int32_t* mps, int n, float th, int far_points, float th_far, float nnratio) {
  return guarded(w, [&] { ORBmatcher m(nnratio, true); return m.SearchByProjection(*w->frames[f], mp_list(w, mps, n), th, far_points != 0, th_far); });
}
int sw_search_by_bow_kf_frame(World* w, int kf, int f, int32_t* out, float nnratio, int check_ori) {
  return guarded(w, [&] {
    ORBmatcher m(nnratio, check_ori != 0);
    std::vector<MapPoint*> v;
    const int n = m.SearchByBoW(w->kfs[kf].get(), *w->frames[f], v);
    for (size_t i = 0; i < v.size(); i++) out[i] = w->mp_index(v[i]);
    return n;
  });
}
int sw_search_by_bow_kf_kf(World* w, int kf1, int kf2, int32_t* out, float nnratio, int check_ori) {
  return guarded(w, [&] {
    ORBmatcher m(nnratio, check_ori != 0);
    std::vector<MapPoint*> v;
    const int n = m.SearchByBoW(w->kfs[kf1].get(), w->kfs[kf2].get(), v);
    for (size_t i = 0; i < v.size(); i++) out[i] = w->mp_index(v[i]);
    return n;
  });
}
int sw_search_for_triangulation(World* w, int kf1, int kf2, int32_t* pairs, int cap, int coarse, int check_ori) {
  return guarded(w, [&] {
    ORBmatcher m(0.6f, check_ori != 0);
    std::vector<std::pair<size_t, size_t>> v;
    const int n = m.SearchForTriangulation(w->kfs[kf1].get(), w->kfs[kf2].get(), v, false, coarse != 0);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) { pairs[2 * i] = (int32_t)v[i].first; pairs[2 * i + 1] = (int32_t)v[i].second; }
    return n;
  });
}
// LocalMapping::CreateNewMapPoints' per-match geometry for one neighbour (host/LocalMapping_shim.h); also returns what the mock
// keyframes hand to it (3x4 poses, camera centres), so that the test can give the oracle the same inputs
int sw_triangulate_matches(World* w, int kf1, int kf2, const int32_t* pairs, int n, int inertial, int far_points, float th_far, float* x3d,
                           int32_t* status, float* T1w, float* T2w, float* Ow1, float* Ow2) {
  return guarded(w, [&] {
    std::vector<std::pair<size_t, size_t>> v(n);
    for (int i = 0; i < n; i++) v[i] = {(size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]};
    std::vector<Eigen::Vector3f> X;
    std::vector<int> st;
    TriangulateMatches(w->kfs[kf1].get(), w->kfs[kf2].get(), v, inertial != 0, far_points != 0, th_far, X, st);
    for (int i = 0; i < n; i++) { for (int k = 0; k < 3; k++) x3d[3 * i + k] = X[i](k); status[i] = st[i]; }
    KeyFrame* kf[2] = {w->kfs[kf1].get(), w->kfs[kf2].get()};
    float* T[2] = {T1w, T2w};
    float* O[2] = {Ow1, Ow2};
    for (int a = 0; a < 2; a++) {
      const Sophus::SE3f Tcw = kf[a]->GetPose();
      const Eigen::Matrix3f R = Tcw.rotationMatrix();
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[a][4 * r + c] = R(r, c);
        T[a][4 * r + 3] = Tcw.translation()(r);
        O[a][r] = kf[a]->GetCameraCenter()(r);
      }
    }
    return 0;
  });
}
// ORBmatcher::Fuse(pKF, vpMapPoints, th): the search on the device, Replace / AddObservation on the mock map
int sw_fuse(World* w, int kf, const int32_t* mps, int n, float th) {
  return guarded(w, [&] { ORBmatcher m(0.6f, true); return m.Fuse(w->kfs[kf].get(), mp_list(w, mps, n), th, false); });
}
int sw_fuse_sim3(World* w, int kf, const float* Scw7, const int32_t* mps, int n, float th, int32_t* replace) {
  return guarded(w, [&] {
    ORBmatcher m(0.6f, true);
    Sophus::Sim3f S = sim3_of(Scw7);
    std::vector<MapPoint*> rep(n, static_cast<MapPoint*>(nullptr));
    const int r = m.Fuse(w->kfs[kf].get(), S, mp_list(w, mps, n), th, rep);
    for (int i = 0; i < n; i++) replace[i] = w->mp_index(rep[i]);
    return r;
  });
}
int sw_search_by_projection_sim3(World* w, int kf, const float* Scw7, const int32_t* mps, int n, int32_t* matched /* kf.N, in / out */, int th, float ratio_hamming) {
  return guarded(w, [&] {
    ORBmatcher m(0.75f, true);
    Sophus::Sim3f S = sim3_of(Scw7);
    KeyFrame* k = w->kfs[kf].get();
    std::vector<MapPoint*> mt = mp_list(w, matched, k->N);
    const int r = m.SearchByProjection(k, S, mp_list(w, mps, n), mt, th, ratio_hamming);
    for (int i = 0; i < k->N; i++) matched[i] = w->mp_index(mt[i]);
    return r;
  });
}
int sw_search_by_sim3(World* w, int kf1, int kf2, int32_t* matches12 /* kf1.N, in / out */, const float* S12_7, float th) {
  return guarded(w, [&] {
    ORBmatcher m(0.75f, true);
    KeyFrame* k1 = w->kfs[kf1].get();
    std::vector<MapPoint*> mt = mp_list(w, matches12, k1->N);
    const int r = m.SearchBySim3(k1, w->kfs[kf2].get(), mt, sim3_of(S12_7), th);
    for (int i = 0; i < k1->N; i++) matches12[i] = w->mp_index(mt[i]);
    return r;
  });
}
int sw_search_for_initialization(World* w, int f1, int f2, float* prev_matched /* 2 x F1.N in / out */, int32_t* matches12, int window, float nnratio, int check_ori) {
  return guarded(w, [&] {
    ORBmatcher m(nnratio, check_ori != 0);
    Frame& A = *w->frames[f1];
    std::vector<cv::Point2f> pm(A.N);
    for (int i = 0; i < A.N; i++) pm[i] = cv::Point2f(prev_matched[2 * i], prev_matched[2 * i + 1]);
    std::vector<int> out;
    const int n = m.SearchForInitialization(A, *w->frames[f2], pm, out, window);
    for (int i = 0; i < A.N; i++) { matches12[i] = out[i]; prev_matched[2 * i] = pm[i].x; prev_matched[2 * i + 1] = pm[i].y; }
    return n;
  });
}

// ---- ORBextractor: construct, run operator() on one image, hand the outputs and one pyramid level back
int sw_extract(World* w, const uint8_t* img, int rows, int cols, int stride, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, int lap0, int lap1,
               dvm_keypoint* kps, uint8_t* desc, int cap, int32_t* n_out, int pyr_level, uint8_t* pyr_out, int32_t* pyr_dims, float* tables /* 4 x nlevels */) {
  return guarded(w, [&] {
    ORBextractor ex(nfeatures, scale_factor, nlevels, ini_th, min_th);
    ex.mbExposePyramid = pyr_out != nullptr;
    cv::Mat image = rows > 0 ? cv::Mat(rows, cols, CV_8UC1, const_cast<uint8_t*>(img), (size_t)stride) : cv::Mat();
    std::vector<cv::KeyPoint> k;
    cv::Mat d;
    std::vector<int> lap = {lap0, lap1};
    cv::_InputArray in(image), mask;
    cv::_OutputArray out(d);
    const int mono = ex(in, mask, k, out, lap);
    *n_out = (int32_t)k.size();
    if ((int)k.size() > cap) throw std::runtime_error("sw_extract: output capacity");
    if (!k.empty()) {
      std::memcpy(kps, static_cast<const void*>(k.data()), sizeof(dvm_keypoint) * k.size());
      if (d.rows != (int)k.size() || d.cols != 32) throw std::runtime_error("sw_extract: descriptor matrix shape");
      for (int r = 0; r < d.rows; r++) std::memcpy(desc + 32 * (size_t)r, d.ptr<uint8_t>(r), 32);
    }
    if (pyr_out && mono >= 0) {
      const cv::Mat& L = ex.mvImagePyramid[pyr_level];
      pyr_dims[0] = L.rows; pyr_dims[1] = L.cols;
      for (int r = 0; r < L.rows; r++) std::memcpy(pyr_out + (size_t)r * L.cols, L.ptr<uint8_t>(r), (size_t)L.cols);
    }
    if (tables) {
      const std::vector<float> a = ex.GetScaleFactors(), b = ex.GetInverseScaleFactors(), c = ex.GetScaleSigmaSquares(), e = ex.GetInverseScaleSigmaSquares();
      for (int l = 0; l < nlevels; l++) { tables[l] = a[l]; tables[nlevels + l] = b[l]; tables[2 * nlevels + l] = c[l]; tables[3 * nlevels + l] = e[l]; }
    }
    return mono;
  });
}

}  // extern "C"
