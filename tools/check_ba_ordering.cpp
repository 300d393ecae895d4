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
  // ---- the FLOW form (k_chol_flow): executed symbolically.  Flags as the kernel has them: X of a strip half, L^-1 of a column, PRE of a
  // column, T of a chain-strip half.  Ticketed tasks run when their inputs are there; a DIAG task climbs its chain.  Everything must get
  // done (no circular wait), every strip half and every column must be produced exactly once, a ticketed task may only wait for flags
  // that EARLIER tickets produce (chains wait for later tickets by design: TSLICE / PRE of the columns they climb to), and every tile's
  // contributor list must be the level launches' (same columns, same order, the end-of-level marks where the levels end).
  {
    const int ns = (int)(S.strips.size() / 2), ntask = (int)(S.flow_tasks.size() / 8);
    if ((int)S.flow_col.size() != 8 * nt) return fail("flow_col size");
    std::vector<char> X(2 * ns, 0), L(nt, 0), P(nt, 0), Tf(2 * ns, 0), xdone(nt, 0), done(ntask, 0);
    std::vector<int> producer_ticket_X(2 * ns, -1), producer_ticket_L(nt, -1), producer_ticket_P(nt, -1);   // ticket of the (chain root) task that produces the flag
    // chain state per leaf task: current column, stage
    struct Chain { int task, k; bool active; };
    std::vector<Chain> chains;
    auto contrib_ready = [&](int c, int half, bool slice) {
      const int32_t* e = &S.flow_contrib[4 * (size_t)c];
      return X[2 * e[1] + (slice ? half : 0)] && (slice || X[2 * e[1] + 1]) && X[2 * e[2]] && X[2 * e[2] + 1];
    };
    // reference contributor lists per tile from the level schedule
    auto level_list = [&](int ti, int tj) {
      std::vector<std::pair<int, int>> v;     // (column, last-of-level)
      for (int h = 0; h < S.nlevels; h++)
        for (int t = S.tgt_off[h]; t < S.tgt_off[h + 1]; t++)
          if (S.targets[4 * t] == ti && S.targets[4 * t + 1] == tj)
            for (int c = S.targets[4 * t + 2]; c < S.targets[4 * t + 3]; c++) v.push_back({S.contrib[c], c == S.targets[4 * t + 3] - 1});
      return v;
    };
    int leaves = 0;
    for (int t = 0; t < ntask; t++) {
      const int32_t* T = &S.flow_tasks[8 * (size_t)t];
      const int kind = T[0], ti = T[1], tj = T[2], c0 = T[4], c1 = T[5];
      if (kind == 0) leaves++;
      if (kind == 1 || kind == 4) {
        const auto ref = level_list(ti, tj);
        if ((int)ref.size() != c1 - c0) return fail("flow: a slice's contributor count differs from the level schedule");
        for (int c = c0; c < c1; c++)
          if (S.flow_contrib[4 * c] != ref[c - c0].first || S.flow_contrib[4 * c + 3] != (ref[c - c0].second ? 1 : 0)) return fail("flow: a slice's contributor list differs");
        if (S.strips[2 * T[6]] != ti || S.strips[2 * T[6] + 1] != tj) return fail("flow: a slice names another strip");
      }
      if (kind == 2 || kind == 0) if (ti != tj) return fail("flow: PRE / DIAG off the diagonal");
    }
    if (leaves != S.flow_leaves) return fail("flow_leaves");
    // diagonal tiles: PRE list + the chain's own list + the chain child = the level schedule's list
    for (int k = 0; k + 1 < nt; k++) {
      const auto ref = level_list(k, k);
      std::vector<int> got;
      for (int t = 0; t < ntask; t++) {
        const int32_t* T = &S.flow_tasks[8 * (size_t)t];
        if (T[0] == 2 && T[1] == k) for (int c = T[4]; c < T[5]; c++) got.push_back(S.flow_contrib[4 * c]);
      }
      for (int c = S.flow_col[8 * k + 3]; c < S.flow_col[8 * k + 4]; c++) got.push_back(S.flow_contrib[4 * c]);
      int child = -1;
      for (int m = 0; m + 1 < nt; m++) if (S.flow_col[8 * m] == k && S.flow_col[8 * m + 6]) { if (child >= 0) return fail("flow: two chain children"); child = m; }
      if (child >= 0) got.push_back(child);
      if (got.size() != ref.size()) return fail("flow: a diagonal tile's contributors are not the level schedule's");
      for (size_t i = 0; i < ref.size(); i++) if (got[i] != ref[i].first) return fail("flow: a diagonal tile's contributor order differs");
      if (ref.empty() != (S.flow_col[8 * k + 2] == 0)) return fail("flow: mode 0 <-> no contributor");
    }
    bool progress = true;
    int rounds = 0;
    while (progress) {
      progress = false;
      if (++rounds > 10 * (ntask + nt) + 10) return fail("flow: symbolic execution does not terminate");
      for (int t = 0; t < ntask; t++) {
        if (done[t]) continue;
        const int32_t* T = &S.flow_tasks[8 * (size_t)t];
        const int kind = T[0], ti = T[1], tj = T[2], half = T[3], c0 = T[4], c1 = T[5], self = T[6];
        bool ok = true;
        for (int c = c0; c < c1 && ok; c++) {
          ok = contrib_ready(c, half, kind == 1 || kind == 4);
          // a ticketed task waits only for flags whose producing ticket is EARLIER
          const int32_t* e = &S.flow_contrib[4 * (size_t)c];
          if (ok) for (int f : {2 * e[1] + ((kind == 1 || kind == 4) ? half : 0), 2 * e[2], 2 * e[2] + 1})
            if (producer_ticket_X[f] > t) return fail("flow: a ticketed task waits for a later ticket");
        }
        if (!ok) continue;
        if (kind == 1) {
          if (!L[tj]) continue;
          if (producer_ticket_L[tj] > t) return fail("flow: a slice waits for an L^-1 of a later ticket");
          if (X[2 * self + half]) return fail("flow: a strip half produced twice");
          X[2 * self + half] = 1; producer_ticket_X[2 * self + half] = t;
        } else if (kind == 4) {
          if (Tf[2 * self + half]) return fail("flow: a chain strip half gathered twice");
          Tf[2 * self + half] = 1;
        } else if (kind == 2) {
          if (P[tj]) return fail("flow: PRE twice");
          P[tj] = 1; producer_ticket_P[tj] = t;
        } else {
          if (S.flow_col[8 * ti + 2] != 0) return fail("flow: a ticketed DIAG task with contributors");
          chains.push_back({t, ti, true});
        }
        if (kind != 0) { done[t] = 1; progress = true; }
        else { done[t] = 1; progress = true; }
        (void)ti;
      }
      // the chains: one step each per round
      for (auto& ch : chains) {
        while (ch.active) {
          const int k = ch.k, p = S.flow_col[8 * k], cs = S.flow_col[8 * k + 1];
          if (L[k] == 2) {                                 // published, waiting for the parent's other children
            bool sib = true;
            for (int c = S.flow_col[8 * p + 3]; c < S.flow_col[8 * p + 4]; c++) sib = sib && X[2 * S.flow_contrib[4 * c + 1]] && X[2 * S.flow_contrib[4 * c + 1] + 1];
            if (!sib) break;
            L[k] = 1; ch.k = p; progress = true;
            continue;
          }
          if (L[k]) return fail("flow: a column factorised twice");
          // (the column's own tile is complete here: checked on the way up)
          if (p < 0 && S.flow_col[8 * k + 5]) {            // a root solved in place: needs the gathered rhs strip
            if (!Tf[2 * cs]) break;
            L[k] = 1; xdone[k] = 1; producer_ticket_L[k] = ch.task; ch.active = false; progress = true;
            break;
          }
          if (p < 0) { L[k] = 1; producer_ticket_L[k] = ch.task; ch.active = false; progress = true; break; }
          if (!(Tf[2 * cs] && Tf[2 * cs + 1])) break;      // waits for the two TSLICE halves
          const bool cont = S.flow_col[8 * k + 6] != 0;
          if (cont) {
            const int mode = S.flow_col[8 * p + 2];
            if (mode >= 2 && !P[p]) break;
            bool sib = true;
            for (int c = S.flow_col[8 * p + 3]; c < S.flow_col[8 * p + 4]; c++) sib = sib && X[2 * S.flow_contrib[4 * c + 1]] && X[2 * S.flow_contrib[4 * c + 1] + 1];
            // (the kernel publishes L_k^-1 and X(p,k) BEFORE it waits for the siblings: model that order, or two chains that are each
            //  other's... cannot be: siblings never wait for each other, only the chain child waits)
            if (!L[k]) { L[k] = 1; producer_ticket_L[k] = ch.task; X[2 * cs] = X[2 * cs + 1] = 1; producer_ticket_X[2 * cs] = producer_ticket_X[2 * cs + 1] = ch.task; progress = true; }
            if (!sib) { L[k] = 2; break; }
            L[k] = 1;
            ch.k = p; progress = true;
            continue;
          }
          L[k] = 1; producer_ticket_L[k] = ch.task;
          if (X[2 * cs] || X[2 * cs + 1]) return fail("flow: a chain strip solved twice");
          X[2 * cs] = X[2 * cs + 1] = 1; producer_ticket_X[2 * cs] = producer_ticket_X[2 * cs + 1] = ch.task;
          ch.active = false; progress = true;
        }
      }
    }
    for (int t = 0; t < ntask; t++) if (!done[t]) return fail("flow: a task never becomes runnable (circular wait)");
    for (auto& ch : chains) if (ch.active) return fail("flow: a chain never ends");
    for (int k = 0; k + 1 < nt; k++) if (L[k] != 1) return fail("flow: a column is never factorised");
    for (int st = 0; st < ns; st++) {
      const bool rhs = S.strips[2 * st] == nt - 1;
      const bool folded = rhs && S.flow_col[8 * S.strips[2 * st + 1] + 5];
      if (!folded && (!X[2 * st] || (!rhs && !X[2 * st + 1]))) return fail("flow: a strip half is never published");
    }
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
