// dvm_slam_amd/host/dvm_device.h -- ONE device setting for every reference-side shim of this directory.
// An agent = a process pinned to one GPU of the node (one agent per GPU, orb_slam3_wrapper.cpp runs one System per process): its
// start-up code calls dvm_host::set_device(k) once, and every shim -- ORBextractor, ORBmatcher, Optimizer, Sim3Solver, ORBVocabulary,
// KeyFrameDatabase, MapPoint, LocalMapping, Frame -- creates its handles on that GPU and selects it for the calling thread before
// an entry point that takes neither a handle nor a device argument (those run on the calling thread's current HIP device, which is
// 0 on a fresh std::thread -- Tracking, LocalMapping and LoopClosing each run on their own).
#pragma once
#include <atomic>
#include <stdexcept>

#include "dvmslam_hip.h"

namespace dvm_host {
inline std::atomic<int>& device_setting() { static std::atomic<int> d{0}; return d; }
inline int device() { return device_setting().load(std::memory_order_relaxed); }
inline void set_device(int d) { device_setting().store(d, std::memory_order_relaxed); }
// for the stateless entry points: make the agent's GPU the calling thread's current device
inline void use_device() {
  if (dvm_set_device(device()) != DVM_OK) throw std::runtime_error(dvm_last_error());
}
}  // namespace dvm_host
