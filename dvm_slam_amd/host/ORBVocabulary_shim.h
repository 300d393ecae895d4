// dvm_slam_amd/host/ORBVocabulary_shim.h -- ORB_SLAM3::ORBVocabulary on the HIP library.  In the reference the name is a typedef
// (include/ORBVocabulary.h:33: DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>); this header takes that file's place with a
// class of the same name that offers what the monocular path calls on it:
//   loadFromTextFile(filename)                         System.cc (vocabulary loading)           TemplatedVocabulary.h:1211-1286
//   transform(features, BowVector&, FeatureVector&, 4) Frame.cc:787, KeyFrame.cc:220            TemplatedVocabulary.h:1025-1086
//   score(BowVector, BowVector)                        KeyFrameDatabase.cc:164 ...               ScoringObject.cpp:23-63 (L1)
//   size() / empty()                                   KeyFrameDatabase.cc:40,76
// The tree lives on the device (dvm_host::ORBVocabulary, host/orb_vocabulary.h): the per-feature descent is one
// dvm_vocab_transform launch, the two std::maps are filled on the host in feature order as DBoW2 fills them.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <opencv2/core/core.hpp>

#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "dvm_device.h"
#include "orb_vocabulary.h"

namespace ORB_SLAM3 {

class ORBVocabulary {
 public:
  explicit ORBVocabulary(int device = dvm_host::device()) : device_(device) {}
  bool loadFromTextFile(const std::string& filename) {
    voc_.reset(dvm_host::ORBVocabulary::loadFromTextFile(device_, filename.c_str()));
    return voc_ != nullptr;
  }
  bool empty() const { return !voc_ || voc_->size() == 0; }
  unsigned int size() const { return voc_ ? (unsigned int)voc_->size() : 0u; }
  // features: one 1x32 CV_8U row per keypoint (Converter::toDescriptorVector(mDescriptors))
  void transform(const std::vector<cv::Mat>& features, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const {
    v.clear();
    fv.clear();
    if (empty() || features.empty()) return;
    std::vector<uint8_t> f(32 * features.size());
    for (size_t i = 0; i < features.size(); i++) std::copy(features[i].data, features[i].data + 32, f.begin() + 32 * i);
    dvm_host::BowVector bv;
    dvm_host::FeatureVector fvv;
    if (voc_->transform(f.data(), (int)features.size(), bv, fvv, levelsup) != DVM_OK) throw std::runtime_error(dvm_last_error());
    for (const auto& e : bv) v.insert(v.end(), std::make_pair((DBoW2::WordId)e.first, (DBoW2::WordValue)e.second));
    for (const auto& e : fvv) fv.insert(fv.end(), std::make_pair((DBoW2::NodeId)e.first, e.second));
  }
  double score(const DBoW2::BowVector& a, const DBoW2::BowVector& b) const {
    dvm_host::BowVector x(a.begin(), a.end()), y(b.begin(), b.end());
    return dvm_host::ORBVocabulary::score(x, y);
  }

 private:
  int device_;
  std::unique_ptr<dvm_host::ORBVocabulary> voc_;
};

}  // namespace ORB_SLAM3
