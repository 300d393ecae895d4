// tools/check_ba_ordering.cpp -- CPU check of dvm_slam_amd/csrc/ba_ordering.cpp (run by tests/test_host_logic.py):
// the camera order is a permutation, whole tiles stay together, the level schedule respects every dependency of
// the tile Cholesky, and nested dissection shortens the dependency chain of a loop trajectory.
//   g++ -O2 -std=c++17 tools/check_ba_ordering.cpp dvm_slam_amd/csrc/ba_ordering.cpp -o /tmp/check_ba_ordering
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "../dvm_slam_amd/csrc/ba_ordering.h"

using namespace dvm;

static int fail(const char* m) { std::printf("FAIL: %s\n", m); return 1; }

int main(int argc, char** argv) {
  int n = argc > 1 ? std::atoi(argv[1]) : 499;
  const int span = argc > 2 ? std::atoi(argv[2]) : 7;
  bool loop = argc > 3 ? std::atoi(argv[3]) != 0 : true;
  std::vector<std::vector<int>> adj(n);
  if (argc > 4) {   // adjacency from a file: "n" then pairs "a b" (used to inspect real problems)
    FILE* f = std::fopen(argv[4], "r");
    if (!f || std::fscanf(f, "%d", &n) != 1) return fail("cannot read adjacency file");
    adj.assign(n, {});
    int a, b;
    while (std::fscanf(f, "%d %d", &a, &b) == 2) { adj[a].push_back(b); adj[b].push_back(a); }
    std::fclose(f);
    loop = false;
  } else
  for (int a = 0; a < n; a++)
    for (int d = 1; d <= span; d++) {
      int b = a + d;
      if (b >= n) { if (!loop) continue; b -= n; }
      adj[a].push_back(b); adj[b].push_back(a);
    }
  const std::vector<int> pos = ba_order_cameras(adj);
  std::set<int> seen(pos.begin(), pos.end());
  if ((int)seen.size() != n || *seen.begin() != 0 || *seen.rbegin() != n - 1) return fail("not a permutation");
  const int nt = (n + kCamsPerTile - 1) / kCamsPerTile + 1;   // + rhs tile
  std::vector<std::vector<char>> T(nt, std::vector<char>(nt, 0));
  for (int a = 0; a < n; a++)
    for (int b : adj[a]) {
      const int ta = pos[a] / kCamsPerTile, tb = pos[b] / kCamsPerTile;
      T[std::max(ta, tb)][std::min(ta, tb)] = 1;
    }
  const BaTileSchedule S = ba_tile_schedule(T);
  // replay: a column may be factored only after every update into it and into its strips has been applied
  std::vector<int> level_of(nt, -1);
  for (int h = 0; h < S.nlevels; h++)
    for (int c = S.level_off[h]; c < S.level_off[h + 1]; c++) level_of[S.cols[c]] = h;
  for (int k = 0; k < nt; k++) if (level_of[k] < 0) return fail("column missing from the schedule");
  for (int h = 0; h < S.nlevels; h++) {
    for (int s = S.strip_off[h]; s < S.strip_off[h + 1]; s++) {
      const int i = S.strips[2 * s], k = S.strips[2 * s + 1];
      if (level_of[k] != h || i <= k) return fail("strip in the wrong level");
      if (level_of[i] <= h) return fail("a strip row is factored no later than its column");
    }
    for (int t = S.tgt_off[h]; t < S.tgt_off[h + 1]; t++) {
      const int ti = S.targets[4 * t], tj = S.targets[4 * t + 1];
      if (level_of[ti] <= h || level_of[tj] <= h) return fail("update target already factored");
      for (int c = S.targets[4 * t + 2]; c < S.targets[4 * t + 3]; c++) {
        if (level_of[S.contrib[c]] != h) return fail("contribution from another level");
        // k_chol_trsm_update waits on the slices of exactly these two strips: (ti, k) and (tj, k), both solved by THIS level's launch
        const int ti = S.targets[4 * t], tj = S.targets[4 * t + 1], k = S.contrib[c];
        const int si = S.contrib_strip[2 * c], sj = S.contrib_strip[2 * c + 1];
        if (si < S.strip_off[h] || si >= S.strip_off[h + 1] || sj < S.strip_off[h] || sj >= S.strip_off[h + 1]) return fail("contrib_strip outside the level");
        if (S.strips[2 * si] != ti || S.strips[2 * si + 1] != k || S.strips[2 * sj] != tj || S.strips[2 * sj + 1] != k) return fail("contrib_strip names another strip");
      }
    }
    if (S.contrib_strip.size() != 2 * S.contrib.size()) return fail("contrib_strip size");
  }
  // the level whose panel solve is left to the back substitution: only rhs strips below its columns, no update, and every later
  // level is the rhs tile alone
  if (S.root_level < 0 || S.root_level >= S.nlevels) return fail("root_level out of range");
  for (int h = S.root_level + 1; h < S.nlevels; h++)
    if (S.level_off[h + 1] - S.level_off[h] != 1 || S.cols[S.level_off[h]] != nt - 1) return fail("a camera level behind root_level");
  if (S.n_root_raw) {
    const int h = S.root_level;
    if (S.n_root_raw != S.level_off[h + 1] - S.level_off[h] || S.tgt_off[h + 1] != S.tgt_off[h]) return fail("n_root_raw on a level that updates");
    for (int s2 = S.strip_off[h]; s2 < S.strip_off[h + 1]; s2++) if (S.strips[2 * s2] != nt - 1) return fail("n_root_raw with a strip above the rhs row");
  }
  // the top pair (k_chol_pair): the last two columns of `cols`, each alone on its level; a's tiles below the diagonal are (b, a) and the
  // rhs row, b's the rhs row; what a updates is (b, b) and (rhs, b)
  if ((S.pair_a >= 0) != (S.pair_b >= 0)) return fail("half a pair");
  if (S.pair_a >= 0) {
    const int hb = S.root_level, ha = hb - 1;
    if (ha < 0 || S.n_root_raw != 1) return fail("pair without a single raw root column");
    if (S.level_off[ha + 1] - S.level_off[ha] != 1 || S.level_off[hb + 1] - S.level_off[hb] != 1) return fail("pair levels hold other columns");
    if (S.cols[S.level_off[ha]] != S.pair_a || S.cols[S.level_off[hb]] != S.pair_b) return fail("pair columns are not the levels' columns");
    if (S.level_off[hb + 1] != S.level_off[S.nlevels - 1] && S.level_off[hb + 1] != (int)S.cols.size() - 1 && S.level_off[hb + 1] != (int)S.cols.size())
      return fail("pair is not at the end of the column list");
    bool ba = false;
    for (int i : std::vector<int>(S.colstrips.begin() + S.colstrip_off[S.pair_a], S.colstrips.begin() + S.colstrip_off[S.pair_a + 1])) {
      if (i == S.pair_b) ba = true; else if (i != nt - 1) return fail("pair column a has another tile below it");
    }
    if (!ba) return fail("pair without the tile (b, a)");
    for (int i : std::vector<int>(S.colstrips.begin() + S.colstrip_off[S.pair_b], S.colstrips.begin() + S.colstrip_off[S.pair_b + 1]))
      if (i != nt - 1) return fail("pair column b has a tile below it");
  }
  std::printf("ok n=%d tiles=%d levels=%d fill=%.3f strips=%zu targets=%zu pair=%d,%d\n", n, nt, S.nlevels, S.fill, S.strips.size() / 2,
              S.targets.size() / 4, S.pair_a, S.pair_b);
  if (std::getenv("DVM_ORDER_VERBOSE"))
    for (int h = 0; h < S.nlevels; h++)
      std::printf("  level %d: %d columns, %d strips, %d targets\n", h, S.level_off[h + 1] - S.level_off[h], S.strip_off[h + 1] - S.strip_off[h],
                  S.tgt_off[h + 1] - S.tgt_off[h]);
  if (loop && n >= 300 && S.nlevels > nt / 2) return fail("nested dissection did not shorten the dependency chain");
  return 0;
}
