// oracle/kfdb_oracle.cpp -- CPU restatement of the place-recognition queries of ORB_SLAM3::KeyFrameDatabase as DVM-SLAM
// uses them for map merging.  TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  PARITY UNPINNED (the reference holds no
// test vector); integer bookkeeping + DBoW2's L1 score in double.
//
// Reference files followed (under /root/reference/src/slam_system/orb_slam3/):
//   src/KeyFrameDatabase.cc:43-70     add / erase (inverted file: word -> list<KeyFrame*>, push_back order)
//   src/KeyFrameDatabase.cc:555-669   DetectNBestCandidates            -> orc_kfdb_detect_n_best()
//   src/KeyFrameDatabase.cc:671-678   ResetPlaceRecognitionQuery
//   src/KeyFrameDatabase.cc:688-786   CalculateMergeScore (DVM-SLAM)   -> orc_kfdb_merge_score()
//   src/KeyFrameDatabase.cc:789-808   DetectMergePossibility (DVM-SLAM)-> orc_kfdb_detect_merge_possibility()
//   src/KeyFrameDatabase.cc:810-909   DetectRelocalizationCandidates   -> orc_kfdb_detect_reloc()
//   Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-63  L1Scoring::score   -> orc_bow_score()
// The keyframe state the queries read and write (mnPlaceRecognitionQuery / Words / Score) lives in the database object,
// stale values included, exactly as it lives on the KeyFrame objects in the reference.
#include <algorithm>
#include <cstdint>
#include <list>
#include <map>
#include <set>
#include <vector>

#include "oracle.h"

struct orc_kfdb {
  struct KF {
    std::vector<int32_t> ids;
    std::vector<double> vals;
    int32_t map_id = 0;
    int64_t mnId = 0;
    uint64_t uuid = 0;
    bool bad = false, erased = false;
    std::vector<int32_t> neigh;   // GetBestCovisibilityKeyFrames(10)
    std::set<int32_t> connected;  // GetConnectedKeyFrames()
    uint64_t query = 0;           // mnPlaceRecognitionQuery
    int words = 0;                // mnPlaceRecognitionWords
    float score = 0;              // mPlaceRecognitionScore
    uint64_t reloc_query = 0;     // mnRelocQuery (KeyFrame.cc:60: 0)
    int reloc_words = 0;          // mnRelocWords
    float reloc_score = 0;        // mRelocScore (stale values live on, as on the KeyFrame objects)
  };
  std::vector<KF> kfs;
  std::map<int32_t, std::list<int32_t>> inverted;   // mvInvertedFile
  std::set<int32_t> bad_maps;
};

extern "C" {

orc_kfdb* orc_kfdb_create(void) { return new orc_kfdb; }
void orc_kfdb_destroy(orc_kfdb* db) { delete db; }

int orc_kfdb_add(orc_kfdb* db, const int32_t* ids, const double* vals, int n, int32_t map_id, uint64_t uuid, int64_t mnId) {
  orc_kfdb::KF k;
  k.ids.assign(ids, ids + n); k.vals.assign(vals, vals + n);
  k.map_id = map_id; k.uuid = uuid; k.mnId = mnId;
  const int slot = (int)db->kfs.size();
  db->kfs.push_back(k);
  for (int i = 0; i < n; i++) db->inverted[ids[i]].push_back(slot);   // :46-49
  return slot;
}
void orc_kfdb_erase(orc_kfdb* db, int slot) {   // :52-70
  orc_kfdb::KF& k = db->kfs[slot];
  for (int32_t w : k.ids) db->inverted[w].remove(slot);
  k.erased = true;
}
void orc_kfdb_set_bad(orc_kfdb* db, int slot, int bad) { db->kfs[slot].bad = bad != 0; }
// KeyFrame::UpdateMap: the keyframe now belongs to another map (LoopClosing::MergeLocal, LoopClosing.cc:1558,1767); every query reads pKFi->GetMap() live
void orc_kfdb_set_map(orc_kfdb* db, int slot, int32_t map_id) { db->kfs[slot].map_id = map_id; }
void orc_kfdb_set_map_bad(orc_kfdb* db, int32_t map_id, int bad) { if (bad) db->bad_maps.insert(map_id); else db->bad_maps.erase(map_id); }
void orc_kfdb_set_neighbours(orc_kfdb* db, int slot, const int32_t* neigh, int n) { db->kfs[slot].neigh.assign(neigh, neigh + n); }
void orc_kfdb_set_connected(orc_kfdb* db, int slot, const int32_t* conn, int n) { db->kfs[slot].connected = std::set<int32_t>(conn, conn + n); }
void orc_kfdb_get_state(const orc_kfdb* db, int slot, uint64_t* query, int32_t* words, float* score) {
  *query = db->kfs[slot].query; *words = db->kfs[slot].words; *score = db->kfs[slot].score;
}

void orc_kfdb_merge_score(orc_kfdb* db, const int32_t* qids, const double* qvals, int nq, uint64_t keyFrameId, int32_t map_id,
                          float* score, int32_t* bestKeyFrame) {
  std::list<int32_t> lKFsSharingWords;
  for (orc_kfdb::KF& k : db->kfs)   // ResetPlaceRecognitionQuery(map): every keyframe of that map
    if (k.map_id == map_id && !k.erased) { k.query = 0; k.words = 0; k.score = 0; }
  for (int w = 0; w < nq; w++) {
    auto it = db->inverted.find(qids[w]);
    if (it == db->inverted.end()) continue;
    for (int32_t s : it->second) {
      orc_kfdb::KF& k = db->kfs[s];
      if (k.map_id == map_id && !k.bad && k.uuid != keyFrameId) {
        if (k.query != keyFrameId) { k.words = 0; k.score = 0; k.query = keyFrameId; lKFsSharingWords.push_back(s); }
        k.words++;
      }
    }
  }
  if (lKFsSharingWords.empty()) return;
  int maxCommonWords = 0;
  for (int32_t s : lKFsSharingWords) maxCommonWords = std::max(maxCommonWords, db->kfs[s].words);
  const int minCommonWords = maxCommonWords * 0.8f;
  std::list<std::pair<float, int32_t>> lScoreAndMatch;
  for (int32_t s : lKFsSharingWords) {
    orc_kfdb::KF& k = db->kfs[s];
    if (k.words > minCommonWords) {
      const float si = (float)orc_bow_score(qids, qvals, nq, k.ids.data(), k.vals.data(), (int)k.ids.size());
      k.score = si;
      lScoreAndMatch.push_back({si, s});
    }
  }
  if (lScoreAndMatch.empty()) return;
  for (const auto& sm : lScoreAndMatch) {
    const orc_kfdb::KF& ki = db->kfs[sm.second];
    float bestScore = sm.first, accScore = bestScore;
    int32_t pBestKF = sm.second;
    for (int32_t s2 : ki.neigh) {
      const orc_kfdb::KF& k2 = db->kfs[s2];
      if (k2.query != keyFrameId) continue;
      accScore += k2.score;
      if (k2.score > bestScore) { pBestKF = s2; bestScore = k2.score; }
    }
    if (accScore > *score) { *score = accScore; *bestKeyFrame = pBestKF; }
  }
}

int orc_kfdb_detect_merge_possibility(orc_kfdb* db, const int32_t* qids, const double* qvals, int nq, uint64_t uuid, int32_t map_id,
                                      int32_t* bestKeyFrame, float* score_out, float* baseline_out) {
  float score = 0;
  int32_t best = -1;
  orc_kfdb_merge_score(db, qids, qvals, nq, uuid, map_id, &score, &best);
  *bestKeyFrame = best; *score_out = score; *baseline_out = 0;
  if (score == 0) return 0;
  float baselineScore = 0;
  int32_t baselineBest = -1;
  const orc_kfdb::KF b = db->kfs[best];   // copy: the call below rewrites the state fields only
  orc_kfdb_merge_score(db, b.ids.data(), b.vals.data(), (int)b.ids.size(), b.uuid, b.map_id, &baselineScore, &baselineBest);
  *baseline_out = baselineScore;
  return score > baselineScore * 0.9 ? 1 : 0;
}

void orc_kfdb_detect_n_best(orc_kfdb* db, int slot, int nNumCandidates, int32_t* loop, int32_t* n_loop, int32_t* merge, int32_t* n_merge) {
  *n_loop = 0; *n_merge = 0;
  const orc_kfdb::KF pKF = db->kfs[slot];
  const uint64_t qid = (uint64_t)pKF.mnId;
  std::list<int32_t> lKFsSharingWords;
  for (size_t w = 0; w < pKF.ids.size(); w++) {
    auto it = db->inverted.find(pKF.ids[w]);
    if (it == db->inverted.end()) continue;
    for (int32_t s : it->second) {
      orc_kfdb::KF& k = db->kfs[s];
      if (k.query != qid) {
        k.words = 0;
        if (!pKF.connected.count(s) && k.mnId != pKF.mnId) { k.query = qid; lKFsSharingWords.push_back(s); }
      }
      k.words++;
    }
  }
  if (lKFsSharingWords.empty()) return;
  int maxCommonWords = 0;
  for (int32_t s : lKFsSharingWords) maxCommonWords = std::max(maxCommonWords, db->kfs[s].words);
  const int minCommonWords = maxCommonWords * 0.8f;
  std::list<std::pair<float, int32_t>> lScoreAndMatch;
  for (int32_t s : lKFsSharingWords) {
    orc_kfdb::KF& k = db->kfs[s];
    if (k.words > minCommonWords) {
      const float si = (float)orc_bow_score(pKF.ids.data(), pKF.vals.data(), (int)pKF.ids.size(), k.ids.data(), k.vals.data(), (int)k.ids.size());
      k.score = si;
      lScoreAndMatch.push_back({si, s});
    }
  }
  if (lScoreAndMatch.empty()) return;
  std::list<std::pair<float, int32_t>> lAccScoreAndMatch;
  for (const auto& sm : lScoreAndMatch) {
    const orc_kfdb::KF& ki = db->kfs[sm.second];
    float bestScore = sm.first, accScore = bestScore;
    int32_t pBestKF = sm.second;
    for (int32_t s2 : ki.neigh) {
      const orc_kfdb::KF& k2 = db->kfs[s2];
      if (k2.query != qid) continue;
      accScore += k2.score;
      if (k2.score > bestScore) { pBestKF = s2; bestScore = k2.score; }
    }
    lAccScoreAndMatch.push_back({accScore, pBestKF});
  }
  lAccScoreAndMatch.sort([](const std::pair<float, int32_t>& a, const std::pair<float, int32_t>& b) { return a.first > b.first; });  // compFirst
  std::set<int32_t> spAlreadyAddedKF;
  for (const auto& am : lAccScoreAndMatch) {
    if (!(*n_loop < nNumCandidates || *n_merge < nNumCandidates)) break;
    const int32_t s = am.second;
    const orc_kfdb::KF& k = db->kfs[s];
    if (k.bad) continue;   // the reference spins forever here (no increment, :651-652): bad keyframes never reach this list in practice
    if (!spAlreadyAddedKF.count(s)) {
      if (pKF.map_id == k.map_id && *n_loop < nNumCandidates) loop[(*n_loop)++] = s;
      else if (pKF.map_id != k.map_id && *n_merge < nNumCandidates && !db->bad_maps.count(k.map_id)) merge[(*n_merge)++] = s;
      spAlreadyAddedKF.insert(s);
    }
  }
}

/* KeyFrameDatabase::DetectRelocalizationCandidates(Frame* F, Map* pMap) (:810-909): the frame is (BowVector, mnId).  Note what the
 * reference does and this keeps: the walk is over ALL keyframes of the inverted file (any map, bad or not); the map filter
 * comes last; a neighbour whose mnRelocQuery equals the frame id contributes its mRelocScore even when THIS query did not
 * score it (too few common words: the value of an earlier query, or 0); a frame id equal to a keyframe's mnRelocQuery (0 for
 * the very first frame) never enters that keyframe into the list. */
void orc_kfdb_detect_reloc(orc_kfdb* db, const int32_t* qids, const double* qvals, int nq, uint64_t frame_id, int32_t map_id,
                           int32_t* out, int32_t* n_out) {
  *n_out = 0;
  std::list<int32_t> lKFsSharingWords;
  for (int w = 0; w < nq; w++) {
    auto it = db->inverted.find(qids[w]);
    if (it == db->inverted.end()) continue;
    for (int32_t s : it->second) {
      orc_kfdb::KF& k = db->kfs[s];
      if (k.reloc_query != frame_id) { k.reloc_words = 0; k.reloc_query = frame_id; lKFsSharingWords.push_back(s); }
      k.reloc_words++;
    }
  }
  if (lKFsSharingWords.empty()) return;
  int maxCommonWords = 0;
  for (int32_t s : lKFsSharingWords) maxCommonWords = std::max(maxCommonWords, db->kfs[s].reloc_words);
  const int minCommonWords = maxCommonWords * 0.8f;
  std::list<std::pair<float, int32_t>> lScoreAndMatch;
  for (int32_t s : lKFsSharingWords) {
    orc_kfdb::KF& k = db->kfs[s];
    if (k.reloc_words > minCommonWords) {
      const float si = (float)orc_bow_score(qids, qvals, nq, k.ids.data(), k.vals.data(), (int)k.ids.size());
      k.reloc_score = si;
      lScoreAndMatch.push_back({si, s});
    }
  }
  if (lScoreAndMatch.empty()) return;
  std::list<std::pair<float, int32_t>> lAccScoreAndMatch;
  float bestAccScore = 0;
  for (const auto& sm : lScoreAndMatch) {
    float bestScore = sm.first, accScore = bestScore;
    int32_t pBestKF = sm.second;
    for (int32_t s2 : db->kfs[sm.second].neigh) {
      const orc_kfdb::KF& k2 = db->kfs[s2];
      if (k2.reloc_query != frame_id) continue;
      accScore += k2.reloc_score;
      if (k2.reloc_score > bestScore) { pBestKF = s2; bestScore = k2.reloc_score; }
    }
    lAccScoreAndMatch.push_back({accScore, pBestKF});
    if (accScore > bestAccScore) bestAccScore = accScore;
  }
  const float minScoreToRetain = 0.75f * bestAccScore;
  std::set<int32_t> spAlreadyAddedKF;
  for (const auto& am : lAccScoreAndMatch) {
    if (am.first > minScoreToRetain) {
      const int32_t s = am.second;
      if (db->kfs[s].map_id != map_id) continue;
      if (!spAlreadyAddedKF.count(s)) { out[(*n_out)++] = s; spAlreadyAddedKF.insert(s); }
    }
  }
}
void orc_kfdb_get_reloc_state(const orc_kfdb* db, int slot, uint64_t* query, int32_t* words, float* score) {
  const orc_kfdb::KF& k = db->kfs[slot];
  *query = k.reloc_query; *words = k.reloc_words; *score = k.reloc_score;
}
}  // extern "C"
