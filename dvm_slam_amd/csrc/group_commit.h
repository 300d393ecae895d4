// dvm_slam_amd/csrc/group_commit.h -- "group commit" for blocking per-agent calls (dvm_orb_pool_*, dvm_match_pool_*, dvm_pose_pool_*).
//
// K agents that share a GPU make the reference's per-frame calls from K tracking threads.  Issued one by one those are K chains of small
// launches that serialise in the runtime; the same work as ONE batched launch costs little more than a single call.  The protocol that
// turns the one into the other without changing what a caller sees:
//   join()    a caller takes the next slot of the lane that COLLECTS calls of its shape key (opening a free lane if none does: one collecting
//             lane per shape); a call that finds its lane full and no lane free waits for the next state change;
//   (the caller writes its inputs into its slot -- outside the lock, all callers at once)
//   arrive()  the caller that took slot 0 LEADS the batch: it waits until the batch is full or nobody has joined for `window_us`, closes
//             it (the other lane starts collecting), waits until every joined caller has written its inputs, and returns true -- the
//             leader then runs the batch and publish()es; every other caller blocks in arrive() until that has happened;
//   result()  status and size of the batch; the caller reads its slot's outputs (no lock);
//   finish()  the last reader frees the lane.
// Several lanes: while batches run and are read, the next one collects.
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>

namespace dvm {

struct GroupCommit {
  using clock = std::chrono::steady_clock;
  static constexpr int kLanes = 4;   // batches in flight: one collects while up to three run / are being read.  (Two lanes and batches of
                                     // 32 kept 32 agents in lockstep at 12 k frames/s; four lanes and batches of 8 let them spread: 23 k+)
  struct Lane {
    enum State { FREE, COLLECT, RUN, DONE } state = FREE;
    int count = 0, copied = 0, readers = 0, rc = 0;
    int64_t key[4] = {0, 0, 0, 0};
    clock::time_point last_join;
    std::string err;
    std::condition_variable cv;   // the batch's own traffic: arrivals and inputs (to the leader), DONE (to the others)
  };
  std::mutex m;
  std::condition_variable cv_join;   // callers waiting for a lane to join
  Lane lane[kLanes];
  int cur = 0;          // the lane new calls join
  int max_batch = 1, window_us = 20;
  int solo_streak = 0;  // consecutive batches of one call
  int inside = 0;       // calls between join() and finish()

  // open(li): called under the lock when a free lane is opened for `key` (allocate / size the lane's buffers); non-zero = error, returned
  template <class Open>
  int join(const int64_t key[4], Open&& open, int& li, int& slot) {
    std::unique_lock<std::mutex> lk(m);
    inside++;
    for (;;) {
      // a collecting lane of THIS shape with room (the current one first); else any free lane is opened for it -- one collecting lane per
      // shape, so that agents with different image sizes / cameras / bounds batch side by side instead of queueing behind each other's
      // batches (round-4 advice: a minority shape could starve while free lanes stood by)
      int pick = -1;
      for (int d = 0; d < kLanes && pick < 0; d++) {
        const int l = (cur + d) % kLanes;
        if (lane[l].state == Lane::COLLECT && lane[l].count < max_batch && std::memcmp(lane[l].key, key, sizeof(lane[l].key)) == 0) pick = l;
      }
      if (pick < 0)
        for (int d = 0; d < kLanes; d++) {
          const int l = (cur + d) % kLanes;
          if (lane[l].state != Lane::FREE) continue;
          const int rc = open(l);
          if (rc != 0) { inside--; return rc; }
          Lane& C = lane[l];
          C.state = Lane::COLLECT; C.count = 0; C.copied = 0; C.readers = 0; C.rc = 0; C.err.clear();
          std::memcpy(C.key, key, sizeof(C.key));
          if (lane[cur].state != Lane::COLLECT) cur = l;
          pick = l;
          break;
        }
      if (pick >= 0) {
        Lane& C = lane[pick];
        li = pick; slot = C.count++; C.last_join = clock::now();
        if (C.count == max_batch) C.cv.notify_all();   // full: the leader need not sit out its window
        return 0;
      }
      cv_join.wait(lk);   // every lane is busy or collects another shape at capacity: the next state change wakes us
    }
  }
  bool arrive(int li, int slot) {
    std::unique_lock<std::mutex> lk(m);
    Lane& L = lane[li];
    L.copied++;
    if (slot != 0) {
      L.cv.notify_all();                             // (the leader may be waiting for this input)
      while (L.state != Lane::DONE) L.cv.wait(lk);
      return false;
    }
    // a caller that has been alone for a while does not wait for company (it would pay the window on every call) -- unless another
    // call is inside the service right now, and every eighth solitary batch waits anyway, so that company is noticed when it comes;
    // the first shared batch brings the window back
    const bool alone = solo_streak >= 8 && inside <= 1 && (solo_streak & 7) != 0;
    const auto window = std::chrono::microseconds(alone ? 0 : window_us);
    while (L.count < max_batch) {
      const auto deadline = L.last_join + window;
      if (clock::now() >= deadline) break;
      L.cv.wait_until(lk, deadline);
    }
    L.state = Lane::RUN;                             // closed: nobody joins any more
    solo_streak = L.count == 1 ? solo_streak + 1 : 0;
    for (int l = 0; l < kLanes; l++)
      if (lane[l].state == Lane::FREE) { cur = l; break; }
    cv_join.notify_all();                            // waiting callers may open the next lane
    while (L.copied < L.count) L.cv.wait(lk);        // every joined caller has written its inputs
    return true;
  }
  int batch_count(int li) {
    std::lock_guard<std::mutex> lk(m);
    return lane[li].count;
  }
  void publish(int li, int rc, const std::string& err) {
    std::lock_guard<std::mutex> lk(m);
    Lane& L = lane[li];
    L.rc = rc; L.err = err; L.readers = L.count; L.state = Lane::DONE;
    L.cv.notify_all();
  }
  int result(int li, std::string* err, int* count) {
    std::lock_guard<std::mutex> lk(m);
    if (err) *err = lane[li].err;
    if (count) *count = lane[li].count;
    return lane[li].rc;
  }
  void finish(int li) {
    std::lock_guard<std::mutex> lk(m);
    Lane& L = lane[li];
    inside--;
    if (--L.readers == 0) {
      L.state = Lane::FREE;
      if (lane[cur].state != Lane::COLLECT && lane[cur].state != Lane::FREE) cur = li;   // nothing is collecting: the freed lane is the next to open
      cv_join.notify_all();
    }
  }
};

}  // namespace dvm
