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
};

}  // namespace dvm_host
