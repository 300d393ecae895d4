// dvm_slam_amd/host/orb_vocabulary.h -- host-side mirror of ORB_SLAM3::ORBVocabulary
// (= DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>, TF_IDF weighting, L1_NORM scoring) for the accelerated path.
// The per-feature tree descent runs on the GPU (dvm_vocab_transform); BowVector / FeatureVector are the reference's own
// std::map types and are filled in feature order exactly as TemplatedVocabulary::transform does
// (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1025-1086, BowVector.cpp:32-72, FeatureVector.cpp:27-38).
#pragma once
#include <cstdint>
#include <map>
#include <vector>

#include "dvmslam_hip.h"

namespace dvm_host {

typedef std::map<unsigned int, double> BowVector;                     // WordId -> WordValue
typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector;  // NodeId -> feature indices

class ORBVocabulary {
 public:
  // tree as in dvm_vocab_create (CSR children lists); the arrays are copied to the device
  ORBVocabulary(int device, int n_nodes, const int32_t* child_off, const int32_t* children, const uint8_t* desc,
                const double* weight, const int32_t* word_id, int L);
  // bool loadFromTextFile(const std::string& filename) (TemplatedVocabulary.h:1211-1286): the text form ORBvoc.txt ships in --
  // "k L scoring weighting", then one line per node: "parent isLeaf d0 .. d31 weight", node ids in file order from 1 (0 = root),
  // word ids in leaf order.  Returns nullptr when the file does not parse (wrong header, a parent that does not precede its
  // child, no leaf) or the device refuses the tree.  Only L1_NORM scoring / TF_IDF weighting (scoring 0, weighting 0: what
  // ORB-SLAM3's vocabulary uses) are accepted.  A trailing empty line is skipped (the reference's eof loop turns it into a
  // node with an uninitialised parent).
  static ORBVocabulary* loadFromTextFile(int device, const char* filename);
  int k() const { return k_; }
  int L() const { return L_; }
  int nodes() const { return n_nodes_; }
  int size() const { return n_words_; }      // unsigned int size() const: number of words
  ~ORBVocabulary();
  ORBVocabulary(const ORBVocabulary&) = delete;
  bool ok() const { return v_ != nullptr; }
  // void transform(const std::vector<TDescriptor>& features, BowVector& v, FeatureVector& fv, int levelsup) const
  // features: n x 32 bytes.  Returns a dvm_status.
  int transform(const uint8_t* features, int n, BowVector& v, FeatureVector& fv, int levelsup) const;
  // double score(const BowVector& a, const BowVector& b) const -- L1Scoring::score (ScoringObject.cpp:23-63)
  static double score(const BowVector& v1, const BowVector& v2);

 private:
  dvm_vocab* v_ = nullptr;
  int k_ = 0, L_ = 0, n_nodes_ = 0, n_words_ = 0;
};

}  // namespace dvm_host
