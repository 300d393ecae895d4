// dvm_slam_amd/csrc/ba_ordering.cpp -- see ba_ordering.h.
#include "ba_ordering.h"

#include <algorithm>
#include <functional>
#include <queue>

namespace dvm {

namespace {
using Graph = std::vector<std::vector<int>>;

Graph clean(const Graph& g) {
  Graph c(g.size());
  for (size_t a = 0; a < g.size(); a++) {
    c[a] = g[a];
    std::sort(c[a].begin(), c[a].end());
    c[a].erase(std::unique(c[a].begin(), c[a].end()), c[a].end());
    c[a].erase(std::remove(c[a].begin(), c[a].end(), (int)a), c[a].end());
  }
  return c;
}

// BFS level structure of the sub-graph `in` (mask) from `root`; returns levels, fills order (level by level,
// neighbours by increasing degree = Cuthill-McKee)
std::vector<std::vector<int>> bfs_levels(const Graph& g, const std::vector<char>& in, int root, std::vector<char>& seen) {
  std::vector<std::vector<int>> levels;
  std::vector<int> cur{root};
  seen[root] = 1;
  while (!cur.empty()) {
    levels.push_back(cur);
    std::vector<int> next;
    for (int v : cur) {
      std::vector<int> nb;
      for (int u : g[v]) if (in[u] && !seen[u]) { seen[u] = 1; nb.push_back(u); }
      std::sort(nb.begin(), nb.end(), [&](int a, int b) { return g[a].size() != g[b].size() ? g[a].size() < g[b].size() : a < b; });
      next.insert(next.end(), nb.begin(), nb.end());
    }
    cur.swap(next);
  }
  return levels;
}

// pseudo-peripheral node of the component of `start` (George-Liu: repeat BFS from a min-degree node of the last level)
int pseudo_peripheral(const Graph& g, const std::vector<char>& in, int start) {
  int root = start;
  size_t depth = 0;
  for (int it = 0; it < 8; it++) {
    std::vector<char> seen(g.size(), 0);
    auto lv = bfs_levels(g, in, root, seen);
    if (lv.size() <= depth) break;
    depth = lv.size();
    int best = lv.back()[0];
    for (int v : lv.back()) if (g[v].size() < g[best].size()) best = v;
    if (best == root) break;
    root = best;
  }
  return root;
}

// nested dissection of the sub-graph `nodes`: returns the elimination order (separators last)
// (root_sep: how many entries at the END of `out` form the last piece's top separator -- their mutual order is free)
void nested_dissection(const Graph& g, const std::vector<int>& nodes, std::vector<int>& out, int* root_sep = nullptr) {
  if (root_sep) *root_sep = (int)std::min<size_t>(nodes.size(), 1);
  if (nodes.size() <= 2) { out.insert(out.end(), nodes.begin(), nodes.end()); return; }
  std::vector<char> in(g.size(), 0);
  for (int v : nodes) in[v] = 1;
  std::vector<char> done(g.size(), 0);
  for (int s : nodes) {
    if (done[s]) continue;
    const int root = pseudo_peripheral(g, in, s);
    std::vector<char> seen(g.size(), 0);
    auto lv = bfs_levels(g, in, root, seen);
    size_t cnt = 0;
    for (auto& l : lv) { cnt += l.size(); for (int v : l) done[v] = 1; }
    if (lv.size() < 3) {   // no interior level to cut at: a clique-like piece, eliminate as is
      for (auto& l : lv) out.insert(out.end(), l.begin(), l.end());
      if (root_sep) *root_sep = 1;
      continue;
    }
    // separator = the level that best balances the two sides
    size_t acc = 0, mid = 1, best = cnt;
    for (size_t i = 0; i + 1 < lv.size(); i++) {
      if (i >= 1) {
        const size_t a = acc, b = cnt - acc - lv[i].size();
        const size_t unbalance = (a > b ? a - b : b - a) + lv[i].size();
        if (unbalance < best) { best = unbalance; mid = i; }
      }
      acc += lv[i].size();
    }
    std::vector<int> A, B;
    for (size_t i = 0; i < mid; i++) A.insert(A.end(), lv[i].begin(), lv[i].end());
    for (size_t i = mid + 1; i < lv.size(); i++) B.insert(B.end(), lv[i].begin(), lv[i].end());
    nested_dissection(g, A, out);
    nested_dissection(g, B, out);
    out.insert(out.end(), lv[mid].begin(), lv[mid].end());
    if (root_sep) *root_sep = (int)lv[mid].size();
  }
}
}  // namespace

namespace {
// camera sequence `seq` (a banded order) -> tiles of 10 consecutive cameras -> nested dissection of the tile graph
// -> final position of every camera.  When n is not a multiple of 10 one tile is short, and the row mapping
// i -> (i / 10) * 64 + (i % 10) * 6 only allows it as the LAST tile of the elimination order: `short_at` says which group of
// the sequence is the short one (the groups before and behind it are full); the dissection reports the tiles of its
// top separator (`root_tiles`: their mutual order is free, so a short tile among them moves to the end at no cost).  A short
// tile from further down is shifted to the end of the order too -- which costs a level of the dependency chain (the shifted
// tile hangs below the root separator) -- so the caller re-tiles with the short group at one of `root_tiles`.
std::vector<int> order_from_sequence(const Graph& g, const std::vector<int>& seq, int kCamsPerTile, int short_at, std::vector<int>* root_tiles) {
  const int n = (int)seq.size();
  const int nt = (n + kCamsPerTile - 1) / kCamsPerTile;
  const int rem = n % kCamsPerTile;
  if (rem == 0 || short_at < 0 || short_at >= nt) short_at = nt - 1;
  const int s0 = short_at * kCamsPerTile, s1 = s0 + (rem ? rem : kCamsPerTile);   // sequence range of the short group
  std::vector<int> tile_of(n), first(nt + 1, 0);
  for (int i = 0; i < n; i++) {
    const int t = i < s0 ? i / kCamsPerTile : (i < s1 ? short_at : short_at + 1 + (i - s1) / kCamsPerTile);
    tile_of[seq[i]] = t;
    first[t + 1] = i + 1;
  }
  Graph tg(nt);
  for (int a = 0; a < n; a++)
    for (int b : g[a]) if (tile_of[a] != tile_of[b]) tg[tile_of[a]].push_back(tile_of[b]);
  tg = clean(tg);
  std::vector<int> all_tiles(nt), torder;
  for (int t = 0; t < nt; t++) all_tiles[t] = t;
  int root_sep = 1;
  nested_dissection(tg, all_tiles, torder, &root_sep);
  if (root_tiles) root_tiles->assign(torder.end() - std::min<int>(root_sep, nt), torder.end());
  if (rem != 0 && nt > 0 && torder.back() != short_at) {   // (inside the root separator the move is free: its tiles are a chain anyway)
    std::vector<int> t2;
    for (int t : torder) if (t != short_at) t2.push_back(t);
    t2.push_back(short_at);
    torder.swap(t2);
  }
  std::vector<int> pos(n, -1);
  int p = 0;
  for (int t : torder)
    for (int i = first[t]; i < first[t + 1]; i++) pos[seq[i]] = p++;
  return pos;
}

// dependency-chain length (and fill) of the tile Cholesky under camera order `pos`
std::pair<int, double> evaluate(const Graph& g, const std::vector<int>& pos, int kCamsPerTile) {
  const int n = (int)pos.size();
  const int nt = (n + kCamsPerTile - 1) / kCamsPerTile + 1;
  std::vector<std::vector<char>> T(nt, std::vector<char>(nt, 0));
  for (int a = 0; a < n; a++)
    for (int b : g[a]) {
      const int ta = pos[a] / kCamsPerTile, tb = pos[b] / kCamsPerTile;
      T[std::max(ta, tb)][std::min(ta, tb)] = 1;
    }
  const BaTileSchedule S = ba_tile_schedule(T);
  return {S.nlevels, S.fill};
}
}  // namespace

std::vector<int> ba_order_cameras(const std::vector<std::vector<int>>& adj_in, int per_tile) {
  const int n = (int)adj_in.size();
  const Graph g = clean(adj_in);
  // candidate 1: the given order (keyframe ids follow the trajectory: already banded, and a loop closure stays ONE
  // wrap-around block instead of doubling the bandwidth as a breadth-first order of a ring does)
  std::vector<int> natural(n);
  for (int i = 0; i < n; i++) natural[i] = i;
  // candidate 2: Cuthill-McKee per connected component (for inputs whose ids do not follow the co-visibility)
  std::vector<int> cm;
  {
    std::vector<char> all(n, 1), placed(n, 0);
    for (int s = 0; s < n; s++) {
      if (placed[s]) continue;
      const int root = pseudo_peripheral(g, all, s);
      std::vector<char> seen(n, 0);
      for (auto& l : bfs_levels(g, all, root, seen))
        for (int v : l) { cm.push_back(v); placed[v] = 1; }
    }
  }
  // per candidate sequence: the short tile at the end of the sequence first, then where the dissection wants its last tile
  std::vector<int> best;
  std::pair<int, double> best_e{1 << 30, 0.0};
  const int nt = (n + per_tile - 1) / per_tile;
  for (const std::vector<int>* seq : {&natural, &cm}) {
    std::vector<int> todo{nt - 1}, tried;
    for (int attempt = 0; attempt < 6 && !todo.empty(); attempt++) {
      const int short_at = todo.back();
      todo.pop_back();
      tried.push_back(short_at);
      std::vector<int> root;
      const std::vector<int> p = order_from_sequence(g, *seq, per_tile, short_at, &root);
      const auto e = evaluate(g, p, per_tile);
      if (e < best_e) { best_e = e; best = p; }   // shortest dependency chain, then least fill; ties keep the earlier candidate
      if (n % per_tile == 0 || std::find(root.begin(), root.end(), short_at) != root.end()) break;
      for (int t : root)
        if (std::find(tried.begin(), tried.end(), t) == tried.end() && std::find(todo.begin(), todo.end(), t) == todo.end()) todo.push_back(t);
    }
  }
  return best;
}

BaTileSchedule ba_tile_schedule(std::vector<std::vector<char>> T) {
  BaTileSchedule S;
  const int nt = (int)T.size();
  S.ntiles = nt;
  for (int k = 0; k < nt; k++) { T[k][k] = 1; T[nt - 1][k] = 1; }
  // symbolic factorisation in index order, per-column structure
  std::vector<std::vector<int>> col(nt);
  for (int k = 0; k < nt; k++) {
    for (int i = k + 1; i < nt; i++) if (T[i][k]) col[k].push_back(i);
    for (size_t a = 0; a < col[k].size(); a++)
      for (size_t b = 0; b <= a; b++) T[col[k][a]][col[k][b]] = 1;
  }
  size_t nz = 0;
  for (int i = 0; i < nt; i++)
    for (int j = 0; j <= i; j++)
      if (T[i][j]) { nz++; S.nz_tiles.push_back(i); S.nz_tiles.push_back(j); }
  S.fill = (double)nz / ((double)nt * (nt + 1) / 2);
  // elimination tree heights: parent(k) = first row of column k
  std::vector<int> height(nt, 0);
  for (int k = 0; k < nt; k++)
    if (!col[k].empty()) height[col[k][0]] = std::max(height[col[k][0]], height[k] + 1);
  // (heights are final when visited in index order because parent(k) > k)
  int nl = 0;
  for (int k = 0; k < nt; k++) nl = std::max(nl, height[k] + 1);
  S.nlevels = nl;
  S.level_off.assign(1, 0); S.strip_off.assign(1, 0); S.tgt_off.assign(1, 0);
  std::vector<std::vector<int32_t>> strip_id(nt);   // strip_id[i][k] = index of strip (i, k) in S.strips
  for (int h = 0; h < nl; h++) {
    std::vector<std::vector<std::vector<int>>> upd(nt);   // upd[ti][tj] -> contributing columns (dense map is fine: nt is small)
    std::vector<std::pair<int, int>> touched;
    for (int k = 0; k < nt; k++) {
      if (height[k] != h) continue;
      S.cols.push_back(k);
      for (int i : col[k]) {
        if (strip_id[i].empty()) strip_id[i].assign(nt, -1);
        strip_id[i][k] = (int32_t)(S.strips.size() / 2);
        S.strips.push_back(i); S.strips.push_back(k);
      }
      for (size_t a = 0; a < col[k].size(); a++)
        for (size_t b = 0; b <= a; b++) {
          const int ti = col[k][a], tj = col[k][b];
          // the corner tile (rhs, rhs) holds one scalar, the last pivot, which nothing reads (the solve takes y = L^-1 b from
          // the rhs ROW and never factors the root): EVERY column contributes to it, so updating it put an 18-long serial
          // chain of products on the leaf level's critical path (22 us of a level that otherwise needs ~8)
          if (ti == nt - 1 && tj == nt - 1) continue;
          if (upd[ti].empty()) upd[ti].resize(nt);
          if (upd[ti][tj].empty()) touched.push_back({ti, tj});
          upd[ti][tj].push_back(k);
        }
    }
    std::sort(touched.begin(), touched.end());
    for (auto& t : touched) {
      S.targets.push_back(t.first); S.targets.push_back(t.second);
      S.targets.push_back((int32_t)S.contrib.size());
      for (int k : upd[t.first][t.second]) {
        S.contrib.push_back(k);
        S.contrib_strip.push_back(strip_id[t.first][k]); S.contrib_strip.push_back(strip_id[t.second][k]);
      }
      S.targets.push_back((int32_t)S.contrib.size());
    }
    S.level_off.push_back((int32_t)S.cols.size());
    S.strip_off.push_back((int32_t)(S.strips.size() / 2));
    S.tgt_off.push_back((int32_t)(S.targets.size() / 4));
  }
  {
    int hl = S.nlevels - 1;
    if (hl >= 1 && S.level_off[hl + 1] - S.level_off[hl] == 1 && S.strip_off[hl + 1] == S.strip_off[hl] && S.tgt_off[hl + 1] == S.tgt_off[hl]) hl--;
    S.root_level = hl;
    S.n_root_raw = 0;
    if (hl >= 0 && S.tgt_off[hl + 1] == S.tgt_off[hl]) {
      bool only_rhs = true;
      for (int st = S.strip_off[hl]; st < S.strip_off[hl + 1]; st++) only_rhs = only_rhs && S.strips[2 * st] == nt - 1;
      if (only_rhs) S.n_root_raw = S.level_off[hl + 1] - S.level_off[hl];
    }
  }
  // the top pair (ba_ordering.h): root level = one column b with only its rhs strip; the level before = one column a whose strips are
  // (b, a) and (rhs, a) -- its targets are then (b, b) and (rhs, b), nothing else
  if (S.n_root_raw == 1 && S.root_level >= 1) {
    const int hb = S.root_level, ha = hb - 1;
    if (S.level_off[ha + 1] - S.level_off[ha] == 1) {
      const int a = S.cols[S.level_off[ha]], b = S.cols[S.level_off[hb]];
      bool ok = true, has_ba = false;
      for (int st = S.strip_off[ha]; st < S.strip_off[ha + 1]; st++) {
        const int i = S.strips[2 * st];
        if (i == b) has_ba = true; else if (i != nt - 1) ok = false;
      }
      for (int t = S.tgt_off[ha]; t < S.tgt_off[ha + 1]; t++) {
        const int ti = S.targets[4 * t], tj = S.targets[4 * t + 1];
        if (!((ti == b && tj == b) || (ti == nt - 1 && tj == b))) ok = false;
      }
      if (ok && has_ba) { S.pair_a = a; S.pair_b = b; }
    }
  }
  S.colstrip_off.assign(1, 0);
  for (int k = 0; k < nt; k++) {
    for (int i : col[k]) { S.colstrips.push_back(i); S.colstrip_id.push_back(strip_id[i][k]); }
    S.colstrip_off.push_back((int32_t)S.colstrips.size());
  }
  // ---- flow form (ba_ordering.h): per tile the contributors of every level that updates it, flattened in the order the level launches
  // apply them, with an end-of-level mark; then the task list, level by level: PRE and DIAG of the level's columns, then its slices
  {
    std::vector<std::vector<int32_t>> segs((size_t)nt * nt);
    for (int h = 0; h < nl; h++)
      for (int t = S.tgt_off[h]; t < S.tgt_off[h + 1]; t++) {
        auto& v = segs[(size_t)S.targets[4 * t] * nt + S.targets[4 * t + 1]];
        v.push_back(S.targets[4 * t + 2]); v.push_back(S.targets[4 * t + 3]);
      }
    // contributors of the level ranges [sg0, sg1) of a tile -> flow_contrib entries (column, strip of the A operand, strip of the B operand, last of its level)
    auto add_contrib = [&](const std::vector<int32_t>& v, size_t sg0, size_t sg1) {
      for (size_t sg = sg0; sg < sg1; sg++)
        for (int c = v[2 * sg]; c < v[2 * sg + 1]; c++) {
          const int32_t e[4] = {S.contrib[c], S.contrib_strip[2 * c], S.contrib_strip[2 * c + 1], c == v[2 * sg + 1] - 1 ? 1 : 0};
          S.flow_contrib.insert(S.flow_contrib.end(), e, e + 4);
        }
    };
    auto add_task = [&](int kind, int ti, int tj, int half, int32_t c0, int self, int pre) {
      const int32_t t[8] = {kind, ti, tj, half, c0, (int32_t)(S.flow_contrib.size() / 4), self, pre};
      S.flow_tasks.insert(S.flow_tasks.end(), t, t + 8);
    };
    // diagonal tiles: the LAST contributor of tile (p, p) -- the largest column of the highest level that updates it -- is a child k of p
    // in the elimination tree whose first strip is (p, k): p's CHAIN CHILD.  The workgroup that factorises k goes on to p itself (solves
    // the strip (p, k) with the L^-1 it holds in LDS, multiplies, factorises p): no hand-off on the chain.  The levels before the last
    // one are summed by a PRE task off the critical path; the other contributors of the last level (p's other children) by the chain's
    // workgroup itself, in order, before its own product.  flow_col[k] = {parent p (the row of k's first strip) or -1, strip (p, k), mode
    // of k's own tile: 0 no contributor (a ticketed leaf), 1 the last level only (tgt = S), 2 PRE leaves tgt'; the last level's other
    // contributors as a range of flow_contrib; 1 = a root column solved in place; 1 = k is p's chain child; 0}
    S.flow_col.assign((size_t)8 * nt, 0);
    std::vector<char> chain_strip(S.strips.size() / 2, 0);
    std::vector<int32_t> nseg(nt, 0);
    for (int k = 0; k < nt; k++) S.flow_col[8 * k] = S.flow_col[8 * k + 1] = -1;
    // EVERY column's workgroup solves the strip to the column's parent (its first strip) itself, with the L^-1 it holds in LDS -- the
    // chain child's and the other children's alike: the parent waits for all of them, and a child's strip is ready ~4 us earlier than
    // through an L^-1 hand-off to a slice task.  [0] = parent, [1] = that strip (gathered only), [6] = 1: go on to the parent
    for (int k = 0; k + 1 < nt; k++)
      if (!col[k].empty() && col[k][0] != nt - 1) {
        S.flow_col[8 * k] = col[k][0]; S.flow_col[8 * k + 1] = strip_id[col[k][0]][k];
        chain_strip[strip_id[col[k][0]][k]] = 1;
      }
    for (int k = 0; k + 1 < nt; k++) {
      const auto& v = segs[(size_t)k * nt + k];
      nseg[k] = (int)(v.size() / 2);
      if (v.empty()) { S.flow_leaves++; continue; }
      const int clast = v[v.size() - 1] - 1;                    // last contributor: entry clast of `contrib`
      const int child = S.contrib[clast];
      S.flow_col[8 * child + 6] = 1;                           // k's chain child: its workgroup goes on to k
      S.flow_col[8 * k + 2] = nseg[k] >= 2 ? 2 : 1;
      S.flow_col[8 * k + 3] = (int32_t)(S.flow_contrib.size() / 4);
      add_contrib(v, v.size() / 2 - 1, v.size() / 2);
      S.flow_contrib.resize(S.flow_contrib.size() - 4);         // (the chain child's product comes from LDS)
      S.flow_col[8 * k + 4] = (int32_t)(S.flow_contrib.size() / 4);
    }
    // the ROOT columns (the last launched level, when nothing but the rhs row hangs below them: n_root_raw): the level launches leave
    // their panel solve -- one matrix-vector product -- to the back substitution; in the flow form the chain's workgroup does both
    // itself, with the L^-1 it holds in LDS, the moment the column is factored: y = L^-1 r, x = L^-T y (same sums).  flow_col[k][5] = 1,
    // [1] = the rhs strip, which is then gathered only
    if (S.n_root_raw > 0 && S.root_level >= 0)
      for (int c = S.level_off[S.root_level]; c < S.level_off[S.root_level + 1]; c++) {
        const int k = S.cols[c];
        if (k == nt - 1 || col[k].size() != 1 || col[k][0] != nt - 1) continue;
        S.flow_col[8 * k + 5] = 1; S.flow_col[8 * k + 1] = strip_id[nt - 1][k];
        chain_strip[strip_id[nt - 1][k]] = 1;
      }
    for (int h = 0; h < nl; h++) {
      for (int c = S.level_off[h]; c < S.level_off[h + 1]; c++) {
        const int k = S.cols[c];
        if (k == nt - 1 || nseg[k] < 2) continue;
        const auto& v = segs[(size_t)k * nt + k];
        const int32_t c0 = (int32_t)(S.flow_contrib.size() / 4);
        add_contrib(v, 0, v.size() / 2 - 1);
        add_task(2, k, k, 0, c0, -1, 2);
      }
      for (int c = S.level_off[h]; c < S.level_off[h + 1]; c++) {
        const int k = S.cols[c];
        if (k == nt - 1 || nseg[k] != 0) continue;
        add_task(0, k, k, 0, (int32_t)(S.flow_contrib.size() / 4), -1, 0);
      }
      // strips: one task per 32-row half (the rhs row: its one row is in half 0); a chain strip is gathered only (kind 4: T in place)
      for (int st = S.strip_off[h]; st < S.strip_off[h + 1]; st++) {
        const int ti = S.strips[2 * st], tj = S.strips[2 * st + 1];
        const auto& v = segs[(size_t)ti * nt + tj];
        for (int half = 0; half < (ti == nt - 1 ? 1 : 2); half++) {
          const int32_t c0 = (int32_t)(S.flow_contrib.size() / 4);
          add_contrib(v, 0, v.size() / 2);
          add_task(chain_strip[st] ? 4 : 1, ti, tj, half, c0, st, 0);
        }
      }
    }
  }
  return S;
}

}  // namespace dvm
