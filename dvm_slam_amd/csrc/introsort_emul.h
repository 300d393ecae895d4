// dvm_slam_amd/csrc/introsort_emul.h -- step-exact emulation of libstdc++'s std::sort.
//
// Why: ORBextractor::DistributeOctTree (reference src/ORBextractor.cc:549) calls
//   std::sort(vPrevSizeAndPointerToNode.begin(), ..., compareNodes)
// on (size, node) pairs that frequently compare EQUAL (same size, same UL.x).  std::sort is not
// stable, so which of two equal nodes is split first -- and therefore which keypoints survive -- is
// decided by the exact sequence of swaps of GCC's introsort (bits/stl_algo.h: __introsort_loop with
// median-of-3 + unguarded partition, threshold 16, depth limit 2*lg(n) -> heapsort, then
// __final_insertion_sort).  The device octree reproduces that sequence on (key, payload) pairs;
// tools/check_introsort.cpp verifies this header against the real std::sort on the host.
//
// The algorithm is written once over an accessor `A` with
//     uint32_t key(int i);  uint32_t val(int i);  void set(int i, uint32_t k, uint32_t v);
// (KV: plain arrays, used by the host check; KVLds in octree_kernel.hip: LDS-typed pointers for the rare heapsort
// branch).  Three forms of the partition phase: kv_introsort_loop (the serial text of libstdc++), 
// kv_introsort_loop_ranked (the same result stated by ranks -- what can run in parallel), and the wave-parallel
// implementation of the ranked form in octree_kernel.hip (wave_introsort_loop, ballots).
// key ordering: a < b  <=>  key(a) < key(b)  (caller packs (size, UL.x) lexicographically in 32 bits).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DVM_HD __host__ __device__ __forceinline__
#else
#define DVM_HD inline
#endif

namespace dvm {

struct KV {
  uint32_t* k;
  uint16_t* v;
  DVM_HD uint32_t key(int i) const { return k[i]; }
  DVM_HD uint32_t val(int i) const { return v[i]; }
  DVM_HD void set(int i, uint32_t kk, uint32_t vv) { k[i] = kk; v[i] = (uint16_t)vv; }
};

template <class A>
DVM_HD void kv_move(A& a, int dst, int src) { a.set(dst, a.key(src), a.val(src)); }
template <class A>
DVM_HD void kv_swap(A& a, int i, int j) {
  const uint32_t ki = a.key(i), vi = a.val(i), kj = a.key(j), vj = a.val(j);
  a.set(i, kj, vj);
  a.set(j, ki, vi);
}

// std::__adjust_heap + std::__push_heap (max-heap on '<')
template <class A>
DVM_HD void kv_adjust_heap(A& a, int first, int hole, int len, uint32_t vk, uint32_t vv) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (a.key(first + child) < a.key(first + child - 1)) child--;
    kv_move(a, first + hole, first + child);
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    kv_move(a, first + hole, first + child - 1);
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && a.key(first + parent) < vk) {
    kv_move(a, first + hole, first + parent);
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a.set(first + hole, vk, vv);
}

// std::__partial_sort(first, last, last) = __heap_select (make_heap only) + __sort_heap
template <class A>
DVM_HD void kv_heapsort(A& a, int first, int last) {
  const int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      kv_adjust_heap(a, first, parent, len, a.key(first + parent), a.val(first + parent));
      if (parent == 0) break;
      parent--;
    }
  }
  int l = last;
  while (l - first > 1) {
    --l;
    // __pop_heap(first, l, l): value = *l; *l = *first; adjust_heap(first, 0, l-first, value)
    const uint32_t vk = a.key(l), vv = a.val(l);
    kv_move(a, l, first);
    kv_adjust_heap(a, first, 0, l - first, vk, vv);
  }
}

template <class A>
DVM_HD void kv_unguarded_linear_insert(A& a, int last) {
  const uint32_t vk = a.key(last), vv = a.val(last);
  int next = last - 1;
  while (vk < a.key(next)) {
    kv_move(a, last, next);
    last = next;
    --next;
  }
  a.set(last, vk, vv);
}

template <class A>
DVM_HD void kv_insertion_sort(A& a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (a.key(i) < a.key(first)) {
      const uint32_t vk = a.key(i), vv = a.val(i);
      for (int j = i; j > first; --j) kv_move(a, j, j - 1);  // move_backward
      a.set(first, vk, vv);
    } else {
      kv_unguarded_linear_insert(a, i);
    }
  }
}

// Phase 1 of std::sort: __introsort_loop only (median-of-3 quicksort partitions until every
// unsorted run is <= 16 long, heapsort below the depth limit).  This is the only part of std::sort
// whose treatment of EQUAL keys is not "stable".  Phase 2, __final_insertion_sort, is a plain
// insertion sort of the whole array, i.e. the unique STABLE ordering of phase 1's output -- which a
// GPU computes in parallel as rank(i) = #{j : key[j] < key[i]} + #{j < i : key[j] == key[i]}.
// `stk` = caller-provided scratch of 3*48 ints (on the GPU: LDS, so the explicit stack does not
// land in scratch memory).
template <class A, class StackPtr>
DVM_HD void kv_introsort_loop(A& a, int n, StackPtr stk) {
  if (n <= 1) return;
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) lg++;
  // explicit stack instead of the recursion on the right part (disjoint ranges: same result)
  StackPtr sf = stk, sl = stk + 48, sd = stk + 96;
  int sp = 0;
  sf[sp] = 0; sl[sp] = n; sd[sp] = 2 * lg; sp++;
  while (sp > 0) {
    --sp;
    int first = sf[sp], last = sl[sp], depth = sd[sp];
    while (last - first > 16) {
      if (depth == 0) {
        kv_heapsort(a, first, last);
        break;
      }
      --depth;
      // __unguarded_partition_pivot
      const int mid = first + (last - first) / 2;
      {  // __move_median_to_first(first, first+1, mid, last-1)
        const int r = first, x = first + 1, y = mid, z = last - 1;
        const uint32_t kx = a.key(x), ky = a.key(y), kz = a.key(z);
        if (kx < ky) {
          if (ky < kz) kv_swap(a, r, y);
          else if (kx < kz) kv_swap(a, r, z);
          else kv_swap(a, r, x);
        } else if (kx < kz) kv_swap(a, r, x);
        else if (ky < kz) kv_swap(a, r, z);
        else kv_swap(a, r, y);
      }
      int lo = first + 1, hi = last;
      const uint32_t pk = a.key(first);
      while (true) {
        while (a.key(lo) < pk) ++lo;
        --hi;
        while (pk < a.key(hi)) --hi;
        if (!(lo < hi)) break;
        kv_swap(a, lo, hi);
        ++lo;
      }
      const int cut = lo;
      sf[sp] = cut; sl[sp] = last; sd[sp] = depth; sp++;
      last = cut;
    }
  }
}

// The same __introsort_loop with __unguarded_partition in RANK form -- the formulation the device parallelises
// (octree_kernel.hip: one wavefront, ballots instead of the two loops below).  With pivot P = a[first]:
//   I[k] = k-th position in (first, last), ascending, whose key is >= P          (where the `lo` scan stops)
//   J[k] = k-th position in [first, last), descending, whose key is <= P         (where the `hi` scan stops;
//          `first` itself holds P and is the sentinel the unguarded scan relies on)
// The serial loop swaps exactly the pairs (I[k], J[k]) for k < m, m = first k with !(I[k] < J[k]), because a swapped
// element is never examined again (lo steps over it, hi steps under it); it returns cut = min(I[m], J[m-1]):
// lo stops at the next untouched >= P position or at the last position it filled with one, whichever comes first.
// I, J: scratch of n entries each.  tools/check_introsort.cpp checks this against std::sort as well.
template <class A, class StackPtr, class IdxPtr>
DVM_HD void kv_introsort_loop_ranked(A& a, int n, StackPtr stk, IdxPtr I, IdxPtr J) {
  if (n <= 1) return;
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) lg++;
  StackPtr sf = stk, sl = stk + 48, sd = stk + 96;
  int sp = 0;
  sf[sp] = 0; sl[sp] = n; sd[sp] = 2 * lg; sp++;
  while (sp > 0) {
    --sp;
    int first = sf[sp], last = sl[sp], depth = sd[sp];
    while (last - first > 16) {
      if (depth == 0) {
        kv_heapsort(a, first, last);
        break;
      }
      --depth;
      const int mid = first + (last - first) / 2;
      {  // __move_median_to_first(first, first+1, mid, last-1)
        const int r = first, x = first + 1, y = mid, z = last - 1;
        const uint32_t kx = a.key(x), ky = a.key(y), kz = a.key(z);
        if (kx < ky) {
          if (ky < kz) kv_swap(a, r, y);
          else if (kx < kz) kv_swap(a, r, z);
          else kv_swap(a, r, x);
        } else if (kx < kz) kv_swap(a, r, x);
        else if (ky < kz) kv_swap(a, r, z);
        else kv_swap(a, r, y);
      }
      const uint32_t pk = a.key(first);
      int nI = 0, nJ = 0;
      for (int p = first + 1; p < last; p++) if (!(a.key(p) < pk)) I[nI++] = p;
      for (int p = last - 1; p >= first; p--) if (!(pk < a.key(p))) J[nJ++] = p;
      int m = 0;
      while (m < nI && m < nJ && I[m] < J[m]) m++;
      for (int k = 0; k < m; k++) kv_swap(a, I[k], J[k]);
      int cut = 0x7fffffff;
      if (m < nI) cut = I[m];
      if (m > 0 && J[m - 1] < cut) cut = J[m - 1];
      sf[sp] = cut; sl[sp] = last; sd[sp] = depth; sp++;
      last = cut;
    }
  }
}

// std::sort(first, last, comp) of libstdc++ on n elements starting at index 0 (fully serial form).
template <class A>
DVM_HD void kv_std_sort(A& a, int n) {
  if (n <= 1) return;
  int stk[144];
  kv_introsort_loop(a, n, stk);
  // __final_insertion_sort
  if (n > 16) {
    kv_insertion_sort(a, 0, 16);
    for (int i = 16; i != n; ++i) kv_unguarded_linear_insert(a, i);
  } else {
    kv_insertion_sort(a, 0, n);
  }
}

}  // namespace dvm
