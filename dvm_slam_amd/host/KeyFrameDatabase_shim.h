// dvm_slam_amd/host/KeyFrameDatabase_shim.h -- ORB_SLAM3::KeyFrameDatabase (include/KeyFrameDatabase.h:46-107,
// src/KeyFrameDatabase.cc) on the HIP library: the class, with the interface the monocular DVM-SLAM path calls.  It takes the
// place of include/KeyFrameDatabase.h (+ src/KeyFrameDatabase.cc):
//   add / erase / clear / clearMap                         KeyFrameDatabase.cc:43-115        LocalMapping, KeyFrame::SetBadFlag, Atlas
//   CalculateMergeScore, DetectMergePossibility            :688-808 (DVM-SLAM)               orb_slam3_wrapper.cpp:457-618 (a peer's BoW vector)
//   DetectNBestCandidates                                  :555-669                          LoopClosing::NewDetectCommonRegions
//   DetectRelocalizationCandidates                         :810-909                          Tracking::Relocalization
//   ConvertUuidToKeyFrame / GetUuidToKeyFrameMap           :911-925 (DVM-SLAM)
//   SetORBVocabulary                                       :927-931
// Not provided: the deprecated DetectLoopCandidates / DetectCandidates / DetectBestCandidates (nobody calls them) and the Boost
// serialisation hooks (PreSave / PostLoad: map save / load is outside the accelerated path).
//
// How: the keyframes' BoW vectors live on the device (dvm_host::KeyFrameDatabase, host/keyframe_database.h -- one slot per
// keyframe; a query is ONE launch that intersects the query vector with every stored one); the per-keyframe query state
// (mnPlaceRecognitionQuery / Words / Score, mnRelocQuery / Words / Score: nothing outside this class reads them) lives with the
// slots.  What the reference reads LIVE from the keyframes during a query -- isBad(), GetBestCovisibilityKeyFrames(10), the
// query keyframe's GetConnectedKeyFrames(), Map::IsBad() -- is refreshed from the objects at the start of every query.
// boost::uuids::uuid values are mapped to process-unique 64-bit ids (the reference itself reduces them with boost::hash_range,
// ":693 ... only needs to be unique").
#pragma once
#include <algorithm>
#include <map>
#include <mutex>
#include <set>
#include <stdexcept>
#include <utility>
#include <vector>

#include <boost/uuid/uuid.hpp>

#include "Frame.h"
#include "KeyFrame.h"
#include "Map.h"
#include "ORBVocabulary.h"
#include "dvm_device.h"
#include "keyframe_database.h"

namespace ORB_SLAM3 {

class KeyFrameDatabase {
 public:
  KeyFrameDatabase() : db_(new dvm_host::KeyFrameDatabase(dvm_host::device())), live_(this) { db_->SetLiveView(&live_); }
  explicit KeyFrameDatabase(const ORBVocabulary& voc, int device = dvm_host::device())
      : mpVoc(&voc), db_(new dvm_host::KeyFrameDatabase(device)), live_(this) { db_->SetLiveView(&live_); }
  ~KeyFrameDatabase() { delete db_; }
  KeyFrameDatabase(const KeyFrameDatabase&) = delete;

  void add(KeyFrame* pKF) {
    std::unique_lock<std::mutex> lock(mMutex);
    const int slot = db_->add(bow_of(pKF->mBowVec), map_id(pKF->GetMap()), uid(pKF->uuid), (int64_t)pKF->mnId);
    if (slot < 0) throw std::runtime_error(dvm_last_error());
    if ((int)kf_of_.size() <= slot) kf_of_.resize(slot + 1, nullptr);
    kf_of_[slot] = pKF;
    slot_of_[pKF] = slot;
    ever_slot_[pKF] = slot;
    uuidToKeyFrame[pKF->uuid] = pKF;
  }
  void erase(KeyFrame* pKF) {
    std::unique_lock<std::mutex> lock(mMutex);
    const auto it = slot_of_.find(pKF);
    if (it == slot_of_.end()) return;
    db_->erase(it->second);
    slot_of_.erase(it);                    // kf_of_ / ever_slot_ keep the slot: the keyframe object lives on (ORB-SLAM3 never deletes
    uuidToKeyFrame.erase(pKF->uuid);       // keyframes) and other keyframes' covisibility lists may still name it (see Live::best_covisibles)
  }
  void clear() {
    std::unique_lock<std::mutex> lock(mMutex);
    for (const auto& ks : slot_of_) db_->erase(ks.second);
    slot_of_.clear();
    ever_slot_.clear();
    std::fill(kf_of_.begin(), kf_of_.end(), nullptr);
    uuidToKeyFrame.clear();
  }
  void clearMap(Map* pMap) {               // (:95-115: every keyframe of that map leaves the inverted file)
    std::unique_lock<std::mutex> lock(mMutex);
    for (auto it = slot_of_.begin(); it != slot_of_.end();) {
      if (it->first->GetMap() == pMap) {
        db_->erase(it->second);
        kf_of_[it->second] = nullptr;
        ever_slot_.erase(it->first);
        uuidToKeyFrame.erase(it->first->uuid);
        it = slot_of_.erase(it);
      } else {
        ++it;
      }
    }
  }

  void CalculateMergeScore(DBoW2::BowVector bowVector, boost::uuids::uuid keyFrameId, Map* map, float& score, KeyFrame*& bestKeyFrame) {
    std::unique_lock<std::mutex> lock(mMutex);
    refresh(nullptr);
    int32_t best = -1;
    check(db_->CalculateMergeScore(bow_of(bowVector), uid(keyFrameId), map_id(map), score, best));
    if (best >= 0) bestKeyFrame = kf_of_[best];
  }
  std::pair<bool, boost::uuids::uuid> DetectMergePossibility(DBoW2::BowVector bowVector, boost::uuids::uuid uuid, Map* map) {
    std::unique_lock<std::mutex> lock(mMutex);
    refresh(nullptr);
    int32_t best = -1;
    const int r = db_->DetectMergePossibility(bow_of(bowVector), uid(uuid), map_id(map), best);
    check(r < 0 ? r : DVM_OK);
    if (best < 0) return std::make_pair(false, boost::uuids::nil_uuid());     // score == 0 (:796-797)
    return std::make_pair(r == 1, kf_of_[best]->uuid);
  }
  void DetectNBestCandidates(KeyFrame* pKF, std::vector<KeyFrame*>& vpLoopCand, std::vector<KeyFrame*>& vpMergeCand, int nNumCandidates) {
    std::unique_lock<std::mutex> lock(mMutex);
    vpLoopCand.clear(); vpMergeCand.clear();
    const auto it = slot_of_.find(pKF);
    if (it == slot_of_.end()) throw std::invalid_argument("DetectNBestCandidates: the keyframe is not in the database");
    refresh(pKF);
    std::vector<int32_t> l, m;
    check(db_->DetectNBestCandidates(it->second, l, m, nNumCandidates));
    for (int32_t s : l) vpLoopCand.push_back(kf_of_[s]);
    for (int32_t s : m) vpMergeCand.push_back(kf_of_[s]);
  }
  std::vector<KeyFrame*> DetectRelocalizationCandidates(Frame* F, Map* pMap) {
    std::unique_lock<std::mutex> lock(mMutex);
    refresh(nullptr);
    std::vector<int32_t> c;
    check(db_->DetectRelocalizationCandidates(bow_of(F->mBowVec), (uint64_t)F->mnId, map_id(pMap), c));
    std::vector<KeyFrame*> out;
    for (int32_t s : c) out.push_back(kf_of_[s]);
    return out;
  }

  KeyFrame* ConvertUuidToKeyFrame(boost::uuids::uuid uuid) {
    std::unique_lock<std::mutex> lock(mMutex);
    const auto it = uuidToKeyFrame.find(uuid);
    return it == uuidToKeyFrame.end() ? nullptr : it->second;
  }
  std::map<boost::uuids::uuid, KeyFrame*> GetUuidToKeyFrameMap() { std::unique_lock<std::mutex> lock(mMutex); return uuidToKeyFrame; }
  void SetORBVocabulary(ORBVocabulary* pORBVoc) { mpVoc = pORBVoc; }

 protected:
  static void check(int rc) { if (rc != DVM_OK) throw std::runtime_error(dvm_last_error()); }
  static dvm_host::BowVector bow_of(const DBoW2::BowVector& b) { return dvm_host::BowVector(b.begin(), b.end()); }
  int32_t map_id(Map* m) {
    const auto it = map_ids_.find(m);
    if (it != map_ids_.end()) return it->second;
    const int32_t id = (int32_t)map_ids_.size();
    map_ids_[m] = id;
    return id;
  }
  // The reference turns a uuid into its query id with boost::hash_range over the 16 bytes (KeyFrameDatabase.cc:693): a 64-bit hash that
  // cannot meet the small mnId values DetectNBestCandidates writes into the same mnPlaceRecognitionQuery member.  Same here: FNV-1a over
  // the bytes (small consecutive ids DID collide with mnIds: a keyframe last touched by DetectNBestCandidates of keyframe 8 looked
  // "already visited" to the merge query with id 8).  0 is the mirror's reset value: never handed out.
  static uint64_t uid(const boost::uuids::uuid& u) {
    uint64_t h = 1469598103934665603ull;
    for (const auto* p = u.data; p != u.data + 16; ++p) { h ^= (uint64_t)(uint8_t)*p; h *= 1099511628211ull; }
    return h ? h : 1;
  }
  // What the reference reads from the objects DURING a query -- pKFi->GetMap() (LoopClosing::MergeLocal moves keyframes between
  // maps with UpdateMap, LoopClosing.cc:1558,1767), isBad(), GetBestCovisibilityKeyFrames(10), GetConnectedKeyFrames() -- is read
  // from them during the query here too: the mirror asks this view for the slots it reaches (the handful above the common-word
  // bar), not for every stored keyframe.  Map bad flags: a few maps, pushed per query.
  struct Live : dvm_host::LiveKeyFrameView {
    KeyFrameDatabase* o;
    explicit Live(KeyFrameDatabase* o_) : o(o_) {}
    bool isBad(int slot) override { return o->kf_of_[slot]->isBad(); }
    int32_t map_id(int slot) override { return o->map_id(o->kf_of_[slot]->GetMap()); }
    void best_covisibles(int slot, std::vector<int32_t>& out) override {
      out.clear();
      // a neighbour that has LEFT the database still carries the query id / score of the last query that touched it (the reference
      // reads pKF2->mnPlaceRecognitionQuery of the object, in the inverted file or not): its slot outlives erase().  One that was
      // never added has never been touched: no query can match its id.
      for (KeyFrame* n : o->kf_of_[slot]->GetBestCovisibilityKeyFrames(10)) {
        const auto it = o->ever_slot_.find(n);
        if (it != o->ever_slot_.end()) out.push_back(it->second);
      }
    }
    void connected(int slot, std::set<int32_t>& out) override {
      out.clear();
      for (KeyFrame* n : o->kf_of_[slot]->GetConnectedKeyFrames()) {
        const auto it = o->slot_of_.find(n);
        if (it != o->slot_of_.end()) out.insert(it->second);
      }
    }
  };
  void refresh(KeyFrame*) {
    for (const auto& ms : map_ids_) db_->SetMapBad(ms.second, ms.first->IsBad());
  }

  const ORBVocabulary* mpVoc = nullptr;
  std::map<boost::uuids::uuid, KeyFrame*> uuidToKeyFrame;
  std::mutex mMutex;
  dvm_host::KeyFrameDatabase* db_;
  Live live_;
  std::map<KeyFrame*, int> slot_of_;
  std::map<KeyFrame*, int> ever_slot_;      // last slot of every keyframe that has ever been in the database (its query state lives there)
  std::vector<KeyFrame*> kf_of_;
  std::map<Map*, int32_t> map_ids_;
};

}  // namespace ORB_SLAM3
