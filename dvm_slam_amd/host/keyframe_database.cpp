// dvm_slam_amd/host/keyframe_database.cpp -- see keyframe_database.h.
#include "keyframe_database.h"
#include "dvmslam_host.h"

#include <algorithm>
#include <list>

namespace dvm_host {

KeyFrameDatabase::KeyFrameDatabase(int device) {
  if (dvm_bowdb_create(device, &db_) != DVM_OK) db_ = nullptr;
}
KeyFrameDatabase::~KeyFrameDatabase() { if (db_) dvm_bowdb_destroy(db_); }

int KeyFrameDatabase::add(const BowVector& bow, int32_t map_id, uint64_t uuid, int64_t mnId) {
  if (uuid == 0) return DVM_ERR_INVALID;
  Lock l(mMutex_);
  std::vector<int32_t> ids;
  std::vector<double> vals;
  for (const auto& wv : bow) { ids.push_back((int32_t)wv.first); vals.push_back(wv.second); }
  int32_t slot = -1;
  const int rc = dvm_bowdb_add(db_, ids.data(), vals.data(), (int)ids.size(), &slot);
  if (rc != DVM_OK) return rc;
  KF k;
  k.bow = bow; k.map_id = map_id; k.uuid = uuid; k.mnId = mnId; k.seq = (uint64_t)slot;
  kfs_.push_back(k);
  return slot;
}

void KeyFrameDatabase::erase(int slot) {
  Lock l(mMutex_);
  kfs_[slot].erased = true;
  dvm_bowdb_erase(db_, slot);
}

int KeyFrameDatabase::query_device(const BowVector& bow) {
  std::vector<int32_t> ids;
  std::vector<double> vals;
  for (const auto& wv : bow) { ids.push_back((int32_t)wv.first); vals.push_back(wv.second); }
  const int N = dvm_bowdb_size(db_);
  common_.assign(N, 0); first_.assign(N, -1); score_.assign(N, 0.f);
  return dvm_bowdb_query(db_, ids.data(), vals.data(), (int)ids.size(), common_.data(), first_.data(), score_.data());
}

std::vector<int32_t> KeyFrameDatabase::walk_order() const {
  std::vector<int32_t> order;
  for (int s = 0; s < (int)kfs_.size(); s++)
    if (common_[s] > 0) order.push_back(s);
  std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
    return first_[a] != first_[b] ? first_[a] < first_[b] : kfs_[a].seq < kfs_[b].seq;
  });
  return order;
}

int KeyFrameDatabase::CalculateMergeScore(const BowVector& bowVector, uint64_t keyFrameId, int32_t map_id, float& score,
                                          int32_t& bestKeyFrame) {
  if (keyFrameId == 0) return DVM_ERR_INVALID;
  Lock l(mMutex_);
  for (int s = 0; s < (int)kfs_.size(); s++) {   // ResetPlaceRecognitionQuery(map): the keyframes that are in that map NOW
    KF& k = kfs_[s];
    if (!k.erased && map_of(s) == map_id) { k.query = 0; k.words = 0; k.score = 0; }
  }
  const int rc = query_device(bowVector);
  if (rc != DVM_OK) return rc;
  std::vector<int32_t> lKFsSharingWords;
  for (int32_t s : walk_order()) {
    KF& k = kfs_[s];
    if (!(map_of(s) == map_id && !bad_of(s) && k.uuid != keyFrameId)) continue;
    if (k.query != keyFrameId) { k.words = 0; k.score = 0; k.query = keyFrameId; lKFsSharingWords.push_back(s); }
    k.words += common_[s];
  }
  if (lKFsSharingWords.empty()) return DVM_OK;
  int maxCommonWords = 0;
  for (int32_t s : lKFsSharingWords) maxCommonWords = std::max(maxCommonWords, kfs_[s].words);
  const int minCommonWords = maxCommonWords * 0.8f;
  std::vector<std::pair<float, int32_t>> lScoreAndMatch;
  for (int32_t s : lKFsSharingWords)
    if (kfs_[s].words > minCommonWords) { kfs_[s].score = score_[s]; lScoreAndMatch.push_back({score_[s], s}); }
  for (const auto& sm : lScoreAndMatch) {
    float bestScore = sm.first, accScore = bestScore;
    int32_t pBestKF = sm.second;
    for (int32_t s2 : neigh_of(sm.second)) {
      const KF& k2 = kfs_[s2];
      if (k2.query != keyFrameId) continue;
      accScore += k2.score;
      if (k2.score > bestScore) { pBestKF = s2; bestScore = k2.score; }
    }
    if (accScore > score) { score = accScore; bestKeyFrame = pBestKF; }
  }
  return DVM_OK;
}

int KeyFrameDatabase::DetectMergePossibility(const BowVector& bowVector, uint64_t uuid, int32_t map_id, int32_t& bestKeyFrame,
                                             float* score_out, float* baseline_out) {
  Lock l(mMutex_);
  float score = 0;
  bestKeyFrame = -1;
  int rc = CalculateMergeScore(bowVector, uuid, map_id, score, bestKeyFrame);
  if (rc != DVM_OK) return rc;
  if (score_out) *score_out = score;
  if (baseline_out) *baseline_out = 0;
  if (score == 0) return 0;
  float baselineScore = 0;
  int32_t baselineBest = -1;
  const KF b = kfs_[bestKeyFrame];
  rc = CalculateMergeScore(b.bow, b.uuid, map_of(bestKeyFrame), baselineScore, baselineBest);
  if (rc != DVM_OK) return rc;
  if (baseline_out) *baseline_out = baselineScore;
  return score > baselineScore * 0.9 ? 1 : 0;
}

int KeyFrameDatabase::DetectNBestCandidates(int slot, std::vector<int32_t>& vpLoopCand, std::vector<int32_t>& vpMergeCand, int nNumCandidates) {
  vpLoopCand.clear(); vpMergeCand.clear();
  Lock l(mMutex_);
  if (live_) { live_->connected(slot, kfs_[slot].connected); map_of(slot); }
  const KF pKF = kfs_[slot];
  const uint64_t qid = (uint64_t)pKF.mnId;
  const int rc = query_device(pKF.bow);
  if (rc != DVM_OK) return rc;
  std::vector<int32_t> lKFsSharingWords;
  for (int32_t s : walk_order()) {   // s shares common_[s] >= 1 words: it is touched common_[s] times by the walk (:563-575)
    KF& k = kfs_[s];
    if (k.query != qid) {
      if (!pKF.connected.count(s) && k.mnId != pKF.mnId) {
        k.query = qid; lKFsSharingWords.push_back(s);
        k.words = common_[s];          // reset at the first touch, then one increment per shared word
      } else {
        k.words = 1;                   // reset at EVERY touch (the query id is never set), then incremented once
      }
    } else {
      k.words += common_[s];           // same query id as last time: the counter keeps counting
    }
  }
  if (lKFsSharingWords.empty()) return DVM_OK;
  int maxCommonWords = 0;
  for (int32_t s : lKFsSharingWords) maxCommonWords = std::max(maxCommonWords, kfs_[s].words);
  const int minCommonWords = maxCommonWords * 0.8f;
  std::vector<std::pair<float, int32_t>> lScoreAndMatch;
  for (int32_t s : lKFsSharingWords)
    if (kfs_[s].words > minCommonWords) { kfs_[s].score = score_[s]; lScoreAndMatch.push_back({score_[s], s}); }
  if (lScoreAndMatch.empty()) return DVM_OK;
  std::list<std::pair<float, int32_t>> lAccScoreAndMatch;
  for (const auto& sm : lScoreAndMatch) {
    float bestScore = sm.first, accScore = bestScore;
    int32_t pBestKF = sm.second;
    for (int32_t s2 : neigh_of(sm.second)) {
      const KF& k2 = kfs_[s2];
      if (k2.query != qid) continue;
      accScore += k2.score;
      if (k2.score > bestScore) { pBestKF = s2; bestScore = k2.score; }
    }
    lAccScoreAndMatch.push_back({accScore, pBestKF});
  }
  lAccScoreAndMatch.sort([](const std::pair<float, int32_t>& a, const std::pair<float, int32_t>& b) { return a.first > b.first; });
  std::set<int32_t> spAlreadyAddedKF;
  for (const auto& am : lAccScoreAndMatch) {
    if (!((int)vpLoopCand.size() < nNumCandidates || (int)vpMergeCand.size() < nNumCandidates)) break;
    const int32_t s = am.second;
    if (bad_of(s)) continue;   // the reference never advances past a bad keyframe here (:651-652); they do not reach this list
    if (!spAlreadyAddedKF.count(s)) {
      const int32_t ms = map_of(s);
      if (pKF.map_id == ms && (int)vpLoopCand.size() < nNumCandidates) vpLoopCand.push_back(s);
      else if (pKF.map_id != ms && (int)vpMergeCand.size() < nNumCandidates && !bad_maps_.count(ms)) vpMergeCand.push_back(s);
      spAlreadyAddedKF.insert(s);
    }
  }
  return DVM_OK;
}

// KeyFrameDatabase.cc:810-909.  The walk of the inverted file meets a keyframe once per shared word: the first touch of a
// keyframe whose mnRelocQuery is not this frame's id resets its counter and enters it into the list, every touch increments --
// so a keyframe that already carries the id (the same frame asked twice; id 0 against the reset value 0) keeps counting and
// is NOT listed, exactly as in the reference.  No map / bad filter until the very end.
int KeyFrameDatabase::DetectRelocalizationCandidates(const BowVector& bowVector, uint64_t frameId, int32_t map_id,
                                                     std::vector<int32_t>& vpRelocCandidates) {
  vpRelocCandidates.clear();
  Lock l(mMutex_);
  const int rc = query_device(bowVector);
  if (rc != DVM_OK) return rc;
  std::vector<int32_t> lKFsSharingWords;
  for (int32_t s : walk_order()) {
    KF& k = kfs_[s];
    if (k.reloc_query != frameId) { k.reloc_words = 0; k.reloc_query = frameId; lKFsSharingWords.push_back(s); }
    k.reloc_words += common_[s];
  }
  if (lKFsSharingWords.empty()) return DVM_OK;
  int maxCommonWords = 0;
  for (int32_t s : lKFsSharingWords) maxCommonWords = std::max(maxCommonWords, kfs_[s].reloc_words);
  const int minCommonWords = maxCommonWords * 0.8f;
  std::vector<std::pair<float, int32_t>> lScoreAndMatch;
  for (int32_t s : lKFsSharingWords)
    if (kfs_[s].reloc_words > minCommonWords) { kfs_[s].reloc_score = score_[s]; lScoreAndMatch.push_back({score_[s], s}); }
  if (lScoreAndMatch.empty()) return DVM_OK;
  std::vector<std::pair<float, int32_t>> lAccScoreAndMatch;
  float bestAccScore = 0;
  for (const auto& sm : lScoreAndMatch) {
    float bestScore = sm.first, accScore = bestScore;
    int32_t pBestKF = sm.second;
    for (int32_t s2 : neigh_of(sm.second)) {
      const KF& k2 = kfs_[s2];
      if (k2.reloc_query != frameId) continue;
      accScore += k2.reloc_score;
      if (k2.reloc_score > bestScore) { pBestKF = s2; bestScore = k2.reloc_score; }
    }
    lAccScoreAndMatch.push_back({accScore, pBestKF});
    if (accScore > bestAccScore) bestAccScore = accScore;
  }
  const float minScoreToRetain = 0.75f * bestAccScore;
  std::set<int32_t> spAlreadyAddedKF;
  for (const auto& am : lAccScoreAndMatch) {
    if (!(am.first > minScoreToRetain)) continue;
    const int32_t s = am.second;
    if (map_of(s) != map_id) continue;
    if (!spAlreadyAddedKF.count(s)) { vpRelocCandidates.push_back(s); spAlreadyAddedKF.insert(s); }
  }
  return DVM_OK;
}

}  // namespace dvm_host

// ---- C entry points for the Python harness
using dvm_host::BowVector;
using dvm_host::KeyFrameDatabase;
static BowVector to_bow(const int32_t* ids, const double* vals, int n) {
  BowVector b;
  for (int i = 0; i < n; i++) b[(unsigned)ids[i]] = vals[i];
  return b;
}
struct dvmh_kfdb : KeyFrameDatabase {          // the opaque handle of include/dvmslam_host.h
  explicit dvmh_kfdb(int device) : KeyFrameDatabase(device) {}
};
extern "C" {
dvmh_kfdb* dvmh_kfdb_create(int device) {
  dvmh_kfdb* db = new dvmh_kfdb(device);
  if (!db->ok()) { delete db; return nullptr; }
  return db;
}
void dvmh_kfdb_destroy(dvmh_kfdb* db) { delete db; }
int dvmh_kfdb_add(dvmh_kfdb* db, const int32_t* ids, const double* vals, int n, int32_t map_id, uint64_t uuid, int64_t mnId) {
  return db->add(to_bow(ids, vals, n), map_id, uuid, mnId);
}
void dvmh_kfdb_erase(dvmh_kfdb* db, int slot) { db->erase(slot); }
void dvmh_kfdb_set_bad(dvmh_kfdb* db, int slot, int bad) { db->SetBadFlag(slot, bad != 0); }
void dvmh_kfdb_set_map_bad(dvmh_kfdb* db, int32_t map_id, int bad) { db->SetMapBad(map_id, bad != 0); }
void dvmh_kfdb_set_map(dvmh_kfdb* db, int slot, int32_t map_id) { db->SetMap(slot, map_id); }
void dvmh_kfdb_set_neighbours(dvmh_kfdb* db, int slot, const int32_t* neigh, int n) { db->SetBestCovisibilityKeyFrames(slot, neigh, n); }
void dvmh_kfdb_set_connected(dvmh_kfdb* db, int slot, const int32_t* conn, int n) { db->SetConnectedKeyFrames(slot, conn, n); }
void dvmh_kfdb_get_state(dvmh_kfdb* db, int slot, uint64_t* query, int32_t* words, float* score) {
  const KeyFrameDatabase::State s = db->GetState(slot);
  *query = s.query; *words = s.words; *score = s.score;
}
int dvmh_kfdb_merge_score(dvmh_kfdb* db, const int32_t* qids, const double* qvals, int nq, uint64_t keyFrameId, int32_t map_id,
                          float* score, int32_t* bestKeyFrame) {
  return db->CalculateMergeScore(to_bow(qids, qvals, nq), keyFrameId, map_id, *score, *bestKeyFrame);
}
int dvmh_kfdb_detect_merge_possibility(dvmh_kfdb* db, const int32_t* qids, const double* qvals, int nq, uint64_t uuid, int32_t map_id,
                                       int32_t* bestKeyFrame, float* score, float* baseline) {
  return db->DetectMergePossibility(to_bow(qids, qvals, nq), uuid, map_id, *bestKeyFrame, score, baseline);
}
int dvmh_kfdb_detect_reloc(dvmh_kfdb* db, const int32_t* qids, const double* qvals, int nq, uint64_t frame_id, int32_t map_id,
                           int32_t* out, int32_t* n_out) {
  std::vector<int32_t> c;
  const int rc = db->DetectRelocalizationCandidates(to_bow(qids, qvals, nq), frame_id, map_id, c);
  *n_out = (int32_t)c.size();
  std::copy(c.begin(), c.end(), out);
  return rc;
}
void dvmh_kfdb_get_reloc_state(dvmh_kfdb* db, int slot, uint64_t* query, int32_t* words, float* score) {
  const KeyFrameDatabase::State s = db->GetRelocState(slot);
  *query = s.query; *words = s.words; *score = s.score;
}
int dvmh_kfdb_detect_n_best(dvmh_kfdb* db, int slot, int nNum, int32_t* loop, int32_t* n_loop, int32_t* merge, int32_t* n_merge) {
  std::vector<int32_t> l, m;
  const int rc = db->DetectNBestCandidates(slot, l, m, nNum);
  *n_loop = (int32_t)l.size(); *n_merge = (int32_t)m.size();
  std::copy(l.begin(), l.end(), loop); std::copy(m.begin(), m.end(), merge);
  return rc;
}
}
