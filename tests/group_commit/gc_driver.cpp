// tests/group_commit/gc_driver.cpp -- CPU exercise of csrc/group_commit.h (the batching protocol of dvm_orb_pool / dvm_match_pool /
// dvm_pose_pool) without a GPU: K threads submit jobs of two shapes; a "batch run" squares the inputs of its slots.  Checks that every
// job is run exactly once with its own input, that batches never exceed max_batch and never mix shapes, that batches do form under
// concurrency, that a lone caller drops the window, and that an open() failure is returned without wedging the service.
//   usage: gc_driver <threads> <jobs_per_thread> <max_batch> <window_us>   -> one line of JSON, exit code 0 when every check holds
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "group_commit.h"

int main(int argc, char** argv) {
  const int T = argc > 1 ? std::atoi(argv[1]) : 8, J = argc > 2 ? std::atoi(argv[2]) : 200, B = argc > 3 ? std::atoi(argv[3]) : 8;
  const int W = argc > 4 ? std::atoi(argv[4]) : 50;
  dvm::GroupCommit gc;
  gc.max_batch = B; gc.window_us = W;
  constexpr int NL = dvm::GroupCommit::kLanes;
  std::vector<long> in[NL], out[NL];
  std::vector<int64_t> lane_key(NL, -1);
  for (int l = 0; l < NL; l++) { in[l].assign(B, 0); out[l].assign(B, 0); }
  std::atomic<long> runs{0}, jobs_run{0}, max_count{0}, wrong{0}, mixed{0}, open_fail_seen{0};
  std::atomic<int> fail_next_open{0};
  auto call = [&](int64_t shape, long x, long* y, int* batch) -> int {
    const int64_t key[4] = {shape, 0, 0, 0};
    int li = 0, slot = 0;
    const int rc = gc.join(key, [&](int l) { if (fail_next_open.exchange(0)) return -7; lane_key[l] = shape; return 0; }, li, slot);
    if (rc != 0) return rc;
    if (lane_key[li] != shape) mixed++;
    in[li][slot] = x;
    if (gc.arrive(li, slot)) {
      const int n = gc.batch_count(li);
      if (n > B) wrong++;
      long mc = max_count.load();
      while (n > mc && !max_count.compare_exchange_weak(mc, n)) {}
      std::this_thread::sleep_for(std::chrono::microseconds(30));   // the "kernel"
      for (int s = 0; s < n; s++) out[li][s] = in[li][s] * in[li][s];
      runs++; jobs_run += n;
      gc.publish(li, 0, std::string());
    }
    int count = 0;
    const int r = gc.result(li, nullptr, &count);
    *y = out[li][slot];
    *batch = count;
    gc.finish(li);
    return r;
  };
  // 1. a lone caller: after eight solitary batches the window is dropped (a 20 ms window would make this loop take seconds)
  {
    gc.window_us = 20000;
    long y; int b;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 40; i++) { if (call(1, i, &y, &b) != 0 || y != (long)i * i || b != 1) wrong++; }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms > 40 * 20.0 * 0.6) wrong++;   // 8 + every eighth of the rest pay the window: ~12 x 20 ms, far below 40 x 20 ms
    gc.window_us = W;
  }
  // 2. an open() failure comes back to the caller and the next call works
  {
    long y; int b;
    fail_next_open = 1;
    if (call(1, 3, &y, &b) == -7) open_fail_seen++;
    if (call(1, 4, &y, &b) != 0 || y != 16) wrong++;
  }
  // 3. T threads, two shapes
  const long runs0 = runs, jobs0 = jobs_run;
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++)
    th.emplace_back([&, t] {
      for (int j = 0; j < J; j++) {
        const long x = 1000L * t + j;
        long y = -1; int b = 0;
        const int rc = call((t == T - 1 && j % 3 == 0) ? 2 : 1, x, &y, &b);
        if (rc != 0 || y != x * x || b < 1 || b > B) wrong++;
      }
    });
  for (auto& t : th) t.join();
  const long total = (long)T * J;
  const bool ok = wrong == 0 && mixed == 0 && jobs_run - jobs0 == total && open_fail_seen == 1 && (T == 1 || max_count > 1);
  std::printf("{\"ok\": %s, \"jobs\": %ld, \"batches\": %ld, \"max_batch_seen\": %ld, \"wrong\": %ld, \"mixed\": %ld}\n", ok ? "true" : "false", total,
              runs - runs0, max_count.load(), wrong.load(), mixed.load());
  return ok ? 0 : 1;
}
