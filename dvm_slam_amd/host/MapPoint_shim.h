// dvm_slam_amd/host/MapPoint_shim.h -- MapPoint::ComputeDistinctiveDescriptors (include/MapPoint.h, src/MapPoint.cc:384-453) on the HIP
// library: the member function body, and the batched form LocalMapping wants.
// The reference walks a point's observations in std::map<KeyFrame*, tuple<int, int>> order, collects the keyframes' descriptor rows
// (left index, then right index), and keeps the one with the least median Hamming distance to the others (median =
// sorted[0.5 * (N - 1)], the first strictly smaller median wins).  Here the rows of ALL the points of a call travel to the device in
// one block and come back as one index per point (dvm_distinctive_descriptors: a wave per point, N x N distances row by row,
// the median by a ballot binary search) -- LocalMapping::ProcessNewKeyFrame / CreateNewMapPoints / SearchInNeighbors call the
// member once per map point, hundreds of times per keyframe: MapPoint::ComputeDistinctiveDescriptorsBatch(points) does them together.
#pragma once
#include <map>
#include <mutex>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "KeyFrame.h"
#include "MapPoint.h"
#include "dvm_device.h"
#include "dvmslam_hip.h"

namespace ORB_SLAM3 {

// static member: it reads mbBad / mObservations and writes mDescriptor under mMutexFeatures, all protected (include/MapPoint.h:210-248).
// The reference's header gains its declaration: `static void ComputeDistinctiveDescriptorsBatch(const std::vector<MapPoint*>&);`
inline void MapPoint::ComputeDistinctiveDescriptorsBatch(const std::vector<MapPoint*>& points) {
  std::vector<uint8_t> rows;
  std::vector<int32_t> off(1, 0);
  std::vector<MapPoint*> who;
  std::vector<std::vector<cv::Mat>> kept;                 // the rows themselves (the winner is cloned into mDescriptor)
  for (MapPoint* p : points) {
    if (!p) continue;
    std::map<KeyFrame*, std::tuple<int, int>> observations;
    {
      // the members themselves, as :391-395 reads them: isBad() and GetObservations() take this same non-recursive mutex
      std::unique_lock<std::mutex> lock1(p->mMutexFeatures);
      if (p->mbBad) continue;
      observations = p->mObservations;
    }
    if (observations.empty()) continue;
    std::vector<cv::Mat> v;
    for (const auto& o : observations) {
      KeyFrame* pKF = o.first;
      if (pKF->isBad()) continue;
      const int leftIndex = std::get<0>(o.second), rightIndex = std::get<1>(o.second);
      if (leftIndex != -1) v.push_back(pKF->mDescriptors.row(leftIndex));
      if (rightIndex != -1) v.push_back(pKF->mDescriptors.row(rightIndex));
    }
    if (v.empty()) continue;
    if (v.size() > 512) throw std::length_error("MapPoint::ComputeDistinctiveDescriptors: more than 512 observations of one map point");
    for (const cv::Mat& d : v) rows.insert(rows.end(), d.data, d.data + 32);
    off.push_back((int32_t)(rows.size() / 32));
    who.push_back(p);
    kept.push_back(std::move(v));
  }
  if (who.empty()) return;
  std::vector<int32_t> best(who.size()), median(who.size());
  dvm_host::use_device();
  if (dvm_distinctive_descriptors(rows.data(), off.data(), (int)who.size(), best.data(), median.data(), 0, nullptr) != DVM_OK)
    throw std::runtime_error(dvm_last_error());
  for (size_t i = 0; i < who.size(); i++) {
    std::unique_lock<std::mutex> lock(who[i]->mMutexFeatures);
    who[i]->mDescriptor = kept[i][best[i]].clone();
  }
}

inline void MapPoint::ComputeDistinctiveDescriptors() { ComputeDistinctiveDescriptorsBatch(std::vector<MapPoint*>(1, this)); }

}  // namespace ORB_SLAM3
