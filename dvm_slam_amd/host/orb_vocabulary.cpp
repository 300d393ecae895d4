// dvm_slam_amd/host/orb_vocabulary.cpp -- see orb_vocabulary.h.
#include "orb_vocabulary.h"
#include "dvmslam_host.h"

#include <cmath>
#include <cstring>

namespace dvm_host {

ORBVocabulary::ORBVocabulary(int device, int n_nodes, const int32_t* child_off, const int32_t* children, const uint8_t* desc,
                             const double* weight, const int32_t* word_id, int L) {
  if (dvm_vocab_create(device, n_nodes, child_off, children, desc, weight, word_id, L, &v_) != DVM_OK) v_ = nullptr;
}
ORBVocabulary::~ORBVocabulary() { if (v_) dvm_vocab_destroy(v_); }

int ORBVocabulary::transform(const uint8_t* features, int n, BowVector& v, FeatureVector& fv, int levelsup) const {
  v.clear();
  fv.clear();
  if (!v_) return DVM_ERR_STATE;
  if (n <= 0) return DVM_OK;
  std::vector<int32_t> word(n), node(n);
  std::vector<double> w(n);
  const int rc = dvm_vocab_transform(v_, features, n, levelsup, word.data(), node.data(), w.data(), 0, nullptr);
  if (rc != DVM_OK) return rc;
  // TF_IDF branch of transform(): addWeight / addFeature in feature order, weight > 0 = "not stopped"
  for (int i = 0; i < n; i++) {
    if (w[i] > 0) {
      BowVector::iterator vit = v.lower_bound((unsigned)word[i]);
      if (vit != v.end() && !(v.key_comp()((unsigned)word[i], vit->first))) vit->second += w[i];
      else v.insert(vit, BowVector::value_type((unsigned)word[i], w[i]));
      fv[(unsigned)node[i]].push_back((unsigned)i);
    }
  }
  // L1_NORM scoring must normalise: BowVector::normalize(L1)
  double norm = 0.0;
  for (BowVector::iterator it = v.begin(); it != v.end(); ++it) norm += std::fabs(it->second);
  if (norm > 0.0)
    for (BowVector::iterator it = v.begin(); it != v.end(); ++it) it->second /= norm;
  return DVM_OK;
}

double ORBVocabulary::score(const BowVector& v1, const BowVector& v2) {
  BowVector::const_iterator v1_it = v1.begin(), v2_it = v2.begin();
  const BowVector::const_iterator v1_end = v1.end(), v2_end = v2.end();
  double score = 0;
  while (v1_it != v1_end && v2_it != v2_end) {
    const double vi = v1_it->second, wi = v2_it->second;
    if (v1_it->first == v2_it->first) {
      score += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi);
      ++v1_it; ++v2_it;
    } else if (v1_it->first < v2_it->first) {
      v1_it = v1.lower_bound(v2_it->first);
    } else {
      v2_it = v2.lower_bound(v1_it->first);
    }
  }
  return -score / 2.0;
}

}  // namespace dvm_host

// ---- C entry points for the Python harness: flattened BowVector / FeatureVector
extern "C" int dvmh_vocab_transform(int device, int n_nodes, const int32_t* child_off, const int32_t* children, const uint8_t* desc,
                                    const double* weight, const int32_t* word_id, int L, const uint8_t* features, int n,
                                    int levelsup, int32_t* bow_ids, double* bow_vals, int* n_bow, int32_t* fv_nodes,
                                    int32_t* fv_off, int32_t* fv_feat, int* n_fv) {
  dvm_host::ORBVocabulary voc(device, n_nodes, child_off, children, desc, weight, word_id, L);
  if (!voc.ok()) return DVM_ERR_STATE;
  dvm_host::BowVector v;
  dvm_host::FeatureVector fv;
  const int rc = voc.transform(features, n, v, fv, levelsup);
  if (rc != DVM_OK) return rc;
  int k = 0;
  for (auto& e : v) { bow_ids[k] = (int32_t)e.first; bow_vals[k] = e.second; k++; }
  *n_bow = k;
  int m = 0, t = 0;
  fv_off[0] = 0;
  for (auto& e : fv) {
    fv_nodes[m] = (int32_t)e.first;
    for (unsigned i : e.second) fv_feat[t++] = (int32_t)i;
    fv_off[++m] = t;
  }
  *n_fv = m;
  return DVM_OK;
}

extern "C" double dvmh_bow_score(const int32_t* ids1, const double* vals1, int n1, const int32_t* ids2, const double* vals2, int n2) {
  dvm_host::BowVector a, b;
  for (int i = 0; i < n1; i++) a[(unsigned)ids1[i]] = vals1[i];
  for (int i = 0; i < n2; i++) b[(unsigned)ids2[i]] = vals2[i];
  return dvm_host::ORBVocabulary::score(a, b);
}
