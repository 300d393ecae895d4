// dvm_slam_amd/host/keyframe_database.h -- host-side mirror of ORB_SLAM3::KeyFrameDatabase's place-recognition queries
// as DVM-SLAM uses them to decide map merges (reference src/KeyFrameDatabase.cc:43-70 add / erase, :555-669
// DetectNBestCandidates, :688-786 CalculateMergeScore, :789-808 DetectMergePossibility) and for relocalisation (:810-909
// DetectRelocalizationCandidates).
//
// Keyframes are slots (the index add() returns) with the few attributes the queries read: BowVector, map, uuid, mnId, isBad,
// GetBestCovisibilityKeyFrames(10), GetConnectedKeyFrames().  The per-keyframe word intersection and the L1 scores of ALL
// stored keyframes come from one device call per query (dvm_bowdb_query); the reference's inverted-file walk is replaced by
// ordering the sharing keyframes by (first shared word, position in that word's inverted list = insertion order), which is
// the order in which the walk meets them.  The per-keyframe query state (mnPlaceRecognitionQuery / Words / Score, stale
// values included), the covisibility accumulation and the candidate selection are replayed on the host.
//
// Threading contract (same as the reference, which serialises add / erase / Detect* / CalculateMergeScore under mMutex because
// LocalMapping, KeyFrame::SetBadFlag, LoopClosing and the wrapper's merge callback reach the database from different threads):
// every public method takes the database's mutex for its whole duration, the device call included, so the device-side CSR
// (re)allocation in add() can never run under a query kernel.  The dvm_bowdb handle must not be used behind the class's back.
// uuid 0 is reserved: 0 is the reset value of mnPlaceRecognitionQuery (KeyFrame.cc:62), a query with id 0 would never enter
// lKFsSharingWords -- add() and the queries return DVM_ERR_INVALID for it.
#pragma once
#include <cstdint>
#include <mutex>
#include <set>
#include <vector>

#include "dvmslam_hip.h"
#include "orb_vocabulary.h"

namespace dvm_host {

// What the reference reads from the KeyFrame / Map OBJECTS while a query runs (pKFi->GetMap(), isBad(), GetBestCovisibilityKeyFrames(10),
// GetConnectedKeyFrames(); KeyFrameDatabase.cc:585-660, 706-790, 850-905): a caller that owns live objects (host/KeyFrameDatabase_shim.h)
// installs this view and the mirror asks it -- only for the slots a query actually reaches, as the reference does -- instead of
// relying on values pushed with the Set* calls (which remain for callers without objects: the C harness, the tests).
struct LiveKeyFrameView {
  virtual ~LiveKeyFrameView() {}
  virtual bool isBad(int slot) = 0;
  virtual int32_t map_id(int slot) = 0;
  virtual void best_covisibles(int slot, std::vector<int32_t>& out) = 0;   // slots of GetBestCovisibilityKeyFrames(10) that are in the database
  virtual void connected(int slot, std::set<int32_t>& out) = 0;            // slots of GetConnectedKeyFrames()
};

class KeyFrameDatabase {
 public:
  explicit KeyFrameDatabase(int device = 0);
  void SetLiveView(LiveKeyFrameView* v) { Lock l(mMutex_); live_ = v; }
  void SetMap(int slot, int32_t map_id) { Lock l(mMutex_); kfs_[slot].map_id = map_id; }   // KeyFrame::UpdateMap (LoopClosing::MergeLocal moves keyframes)
  ~KeyFrameDatabase();
  KeyFrameDatabase(const KeyFrameDatabase&) = delete;
  bool ok() const { return db_ != nullptr; }

  // void add(KeyFrame* pKF): returns the slot
  int add(const BowVector& bow, int32_t map_id, uint64_t uuid, int64_t mnId);
  void erase(int slot);
  void SetBadFlag(int slot, bool bad) { Lock l(mMutex_); kfs_[slot].bad = bad; }
  void SetMapBad(int32_t map_id, bool bad) { Lock l(mMutex_); if (bad) bad_maps_.insert(map_id); else bad_maps_.erase(map_id); }
  void SetBestCovisibilityKeyFrames(int slot, const int32_t* neigh, int n) { Lock l(mMutex_); kfs_[slot].neigh.assign(neigh, neigh + n); }
  void SetConnectedKeyFrames(int slot, const int32_t* conn, int n) { Lock l(mMutex_); kfs_[slot].connected = std::set<int32_t>(conn, conn + n); }

  // void CalculateMergeScore(DBoW2::BowVector bowVector, uuid, Map* map, float& score, KeyFrame*& bestKeyFrame)
  int CalculateMergeScore(const BowVector& bowVector, uint64_t uuid, int32_t map_id, float& score, int32_t& bestKeyFrame);
  // pair<bool, uuid> DetectMergePossibility(DBoW2::BowVector bowVector, uuid, Map* map): returns 1 / 0 (or < 0: dvm_status)
  int DetectMergePossibility(const BowVector& bowVector, uint64_t uuid, int32_t map_id, int32_t& bestKeyFrame, float* score = nullptr,
                             float* baseline = nullptr);
  // void DetectNBestCandidates(KeyFrame* pKF, vector<KeyFrame*>& vpLoopCand, vector<KeyFrame*>& vpMergeCand, int nNumCandidates)
  int DetectNBestCandidates(int slot, std::vector<int32_t>& vpLoopCand, std::vector<int32_t>& vpMergeCand, int nNumCandidates);

  // vector<KeyFrame*> DetectRelocalizationCandidates(Frame* F, Map* pMap): the frame is (mBowVec, mnId)
  int DetectRelocalizationCandidates(const BowVector& bowVector, uint64_t frameId, int32_t map_id, std::vector<int32_t>& vpRelocCandidates);

  struct State { uint64_t query; int32_t words; float score; };
  State GetRelocState(int slot) const { Lock l(mMutex_); return {kfs_[slot].reloc_query, kfs_[slot].reloc_words, kfs_[slot].reloc_score}; }
  State GetState(int slot) const { Lock l(mMutex_); return {kfs_[slot].query, kfs_[slot].words, kfs_[slot].score}; }

 private:
  typedef std::lock_guard<std::recursive_mutex> Lock;   // recursive: DetectMergePossibility runs CalculateMergeScore twice
  mutable std::recursive_mutex mMutex_;
  struct KF {
    BowVector bow;
    int32_t map_id = 0;
    int64_t mnId = 0;
    uint64_t uuid = 0;
    bool bad = false, erased = false;
    uint64_t seq = 0;                 // position key inside every inverted list it is in (push_back order)
    std::vector<int32_t> neigh;
    std::set<int32_t> connected;
    uint64_t query = 0;
    int words = 0;
    float score = 0;
    uint64_t reloc_query = 0;         // mnRelocQuery / mnRelocWords / mRelocScore: a state of their own (KeyFrame.h)
    int reloc_words = 0;
    float reloc_score = 0;
  };
  // attribute reads of a query: the live view when one is installed (the value is also kept, so that the state getters see it)
  int32_t map_of(int s) { if (live_) kfs_[s].map_id = live_->map_id(s); return kfs_[s].map_id; }
  bool bad_of(int s) { if (live_) kfs_[s].bad = live_->isBad(s); return kfs_[s].bad; }
  const std::vector<int32_t>& neigh_of(int s) { if (live_) live_->best_covisibles(s, kfs_[s].neigh); return kfs_[s].neigh; }
  LiveKeyFrameView* live_ = nullptr;
  int query_device(const BowVector& bow);   // fills common_ / first_ / score_
  std::vector<int32_t> walk_order() const;   // slots sharing a word, in inverted-file walk order
  dvm_bowdb* db_ = nullptr;
  std::vector<KF> kfs_;
  std::set<int32_t> bad_maps_;
  std::vector<int32_t> common_, first_;
  std::vector<float> score_;
};

}  // namespace dvm_host
