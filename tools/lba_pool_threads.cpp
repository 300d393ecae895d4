// tools/lba_pool_threads.cpp -- K native threads (one LocalMapping thread per agent), each making `calls` blocking dvm_ba_pool_optimize calls
// on its own window: what a C++ host with more agents than GPUs does.  Built as a small shared object and driven from
// tools/lba_pool_cpp.py (which marshals the windows once: capi.BaWindowBatch).  Returns the wall time of the whole run in ms.
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "dvmslam_hip.h"

extern "C" double lba_pool_threads(dvm_ba_pool* pool, const dvm_ba_window* wins, dvm_ba_stats* stats, int K, int calls, int* launches, int* rc_out) {
  std::atomic<int> ready{0}, rc{0};
  std::atomic<long> windows_in_launches{0}, my_calls{0};
  std::atomic<bool> go{false};
  std::vector<std::thread> th;
  for (int k = 0; k < K; k++)
    th.emplace_back([&, k] {
      ready.fetch_add(1);
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      for (int c = 0; c < calls; c++) {
        int n = 0;
        const int r = dvm_ba_pool_optimize(pool, &wins[k], &stats[k], &n);
        if (r != 0) { rc.store(r); return; }
        windows_in_launches.fetch_add(n); my_calls.fetch_add(1);
      }
    });
  while (ready.load() < K) std::this_thread::yield();
  const auto t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto& t : th) t.join();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  // sum over calls of the size of the launch the call rode in = sum over launches of size^2; with equal sizes: size = that / calls
  if (launches) *launches = my_calls.load() ? (int)(windows_in_launches.load() / my_calls.load()) : 0;
  if (rc_out) *rc_out = rc.load();
  return ms;
}
