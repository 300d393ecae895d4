// tools/check_introsort.cpp -- verifies dvm_slam_amd/csrc/introsort_emul.h against the real
// libstdc++ std::sort (the one the reference's build uses for compareNodes, ORBextractor.cc:549):
// same comparator, same input sequence => the emulation must produce the IDENTICAL permutation,
// including the order of equal keys.  Run by tests/test_host_logic.py.
#include <algorithm>
#include <cstdio>
#include <random>
#include <utility>
#include <vector>

#include "../dvm_slam_amd/csrc/introsort_emul.h"

struct Node { int x0; int id; };
static bool compareNodes(const std::pair<int, Node*>& a, const std::pair<int, Node*>& b) {
  if (a.first < b.first) return true;
  if (a.first > b.first) return false;
  return a.second->x0 < b.second->x0;
}

static bool run_case(const std::vector<std::pair<int, int>>& in) {  // (size, x0)
  const int n = (int)in.size();
  std::vector<Node> nodes(n);
  std::vector<std::pair<int, Node*>> ref(n);
  std::vector<uint32_t> k(n);
  std::vector<uint16_t> v(n);
  for (int i = 0; i < n; i++) {
    nodes[i] = Node{in[i].second, i};
    ref[i] = {in[i].first, &nodes[i]};
    k[i] = ((uint32_t)in[i].first << 12) | (uint32_t)in[i].second;
    v[i] = (uint16_t)i;
  }
  std::sort(ref.begin(), ref.end(), compareNodes);
  std::vector<uint32_t> k2 = k;
  std::vector<uint16_t> v2 = v;
  dvm::KV kv{k.data(), v.data()};
  dvm::kv_std_sort(kv, n);
  for (int i = 0; i < n; i++)
    if (ref[i].second->id != (int)v[i]) return false;
  // split form used on the GPU: serial __introsort_loop, then the stable rank placement
  dvm::KV kv2{k2.data(), v2.data()};
  int stk[144];
  dvm::kv_introsort_loop(kv2, n, stk);
  std::vector<uint16_t> out(n);
  for (int i = 0; i < n; i++) {
    int rank = 0;
    for (int j = 0; j < n; j++) rank += (k2[j] < k2[i]) || (k2[j] == k2[i] && j < i);
    out[rank] = v2[i];
  }
  for (int i = 0; i < n; i++)
    if (ref[i].second->id != (int)out[i]) return false;
  // rank form of the partition (what the device runs with ballots): must leave the SAME array as the serial loop
  std::vector<uint32_t> k3(n);
  std::vector<uint16_t> v3(n);
  for (int i = 0; i < n; i++) { k3[i] = ((uint32_t)in[i].first << 12) | (uint32_t)in[i].second; v3[i] = (uint16_t)i; }
  dvm::KV kv3{k3.data(), v3.data()};
  std::vector<int> I(n + 1), J(n + 1);
  dvm::kv_introsort_loop_ranked(kv3, n, stk, I.data(), J.data());
  for (int i = 0; i < n; i++)
    if (k3[i] != k2[i] || v3[i] != v2[i]) return false;
  return true;
}

int main() {
  std::mt19937 rng(12345);
  long cases = 0;
  for (int rep = 0; rep < 4000; rep++) {
    int n = (rep < 200) ? rep : (int)(rng() % 1500) + 1;
    int nsizes = 1 + rng() % 12, nx = 1 + rng() % 9;  // few distinct values -> many ties
    std::vector<std::pair<int, int>> in(n);
    for (auto& e : in) e = {2 + (int)(rng() % nsizes), (int)(rng() % nx) * 38};
    if (rep % 7 == 0) std::sort(in.begin(), in.end());
    if (rep % 11 == 0) std::reverse(in.begin(), in.end());
    if (!run_case(in)) { std::printf("MISMATCH rep=%d n=%d\n", rep, n); return 1; }
    cases++;
  }
  // adversarial "median-of-3 killer" sequences drive the depth limit to 0 -> heapsort branch
  for (int n : {64, 200, 512, 1000, 2048, 4096}) {
    std::vector<int> a(n);
    int k2 = n / 2;
    for (int i = 0; i < k2; i++) { if (i % 2 == 0) a[i] = i + 1; else a[i] = k2 + i + (k2 % 2 ? 0 : 1); a[k2 + i] = 2 * (i + 1); }
    std::vector<std::pair<int, int>> in(n);
    for (int i = 0; i < n; i++) in[i] = {a[i] % 4000 + 2, (a[i] / 3) % 4000};
    if (!run_case(in)) { std::printf("MISMATCH killer n=%d\n", n); return 1; }
    for (int i = 0; i < n; i++) in[i] = {a[i] + 2, 0};
    if (!run_case(in)) { std::printf("MISMATCH killer2 n=%d\n", n); return 1; }
    cases += 2;
  }
  std::printf("introsort emulation identical to std::sort on %ld cases\n", cases);
  return 0;
}
