// dvm_slam_amd/csrc/introsort_emul.h -- step-exact emulation of libstdc++'s std::sort.
//
// Why: ORBextractor::DistributeOctTree (reference src/ORBextractor.cc:549) calls
//   std::sort(vPrevSizeAndPointerToNode.begin(), ..., compareNodes)
// on (size, node) pairs that frequently compare EQUAL (same size, same UL.x).  std::sort is not
// stable, so which of two equal nodes is split first -- and therefore which keypoints survive -- is
// decided by the exact sequence of swaps of GCC's introsort (bits/stl_algo.h: __introsort_loop with
// median-of-3 + unguarded partition, threshold 16, depth limit 2*lg(n) -> heapsort, then
// __final_insertion_sort).  The device octree reproduces that sequence on (key, payload) pairs held
// in LDS; tools/check_introsort.cpp verifies this header against the real std::sort on the host.
//
// key ordering: a < b  <=>  k[a] < k[b]  (caller packs (size, UL.x) lexicographically into 32 bits).
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
};

DVM_HD void kv_swap(KV a, int i, int j) {
  uint32_t tk = a.k[i]; a.k[i] = a.k[j]; a.k[j] = tk;
  uint16_t tv = a.v[i]; a.v[i] = a.v[j]; a.v[j] = tv;
}

// std::__adjust_heap + std::__push_heap (max-heap on '<')
DVM_HD void kv_adjust_heap(KV a, int first, int hole, int len, uint32_t vk, uint16_t vv) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (a.k[first + child] < a.k[first + child - 1]) child--;
    a.k[first + hole] = a.k[first + child]; a.v[first + hole] = a.v[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a.k[first + hole] = a.k[first + child - 1]; a.v[first + hole] = a.v[first + child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && a.k[first + parent] < vk) {
    a.k[first + hole] = a.k[first + parent]; a.v[first + hole] = a.v[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a.k[first + hole] = vk; a.v[first + hole] = vv;
}

// std::__partial_sort(first, last, last) = __heap_select (make_heap only) + __sort_heap
DVM_HD void kv_heapsort(KV a, int first, int last) {
  const int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      kv_adjust_heap(a, first, parent, len, a.k[first + parent], a.v[first + parent]);
      if (parent == 0) break;
      parent--;
    }
  }
  int l = last;
  while (l - first > 1) {
    --l;
    // __pop_heap(first, l, l): value = *l; *l = *first; adjust_heap(first, 0, l-first, value)
    uint32_t vk = a.k[l]; uint16_t vv = a.v[l];
    a.k[l] = a.k[first]; a.v[l] = a.v[first];
    kv_adjust_heap(a, first, 0, l - first, vk, vv);
  }
}

DVM_HD void kv_unguarded_linear_insert(KV a, int last) {
  uint32_t vk = a.k[last]; uint16_t vv = a.v[last];
  int next = last - 1;
  while (vk < a.k[next]) {
    a.k[last] = a.k[next]; a.v[last] = a.v[next];
    last = next;
    --next;
  }
  a.k[last] = vk; a.v[last] = vv;
}

DVM_HD void kv_insertion_sort(KV a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (a.k[i] < a.k[first]) {
      uint32_t vk = a.k[i]; uint16_t vv = a.v[i];
      for (int j = i; j > first; --j) { a.k[j] = a.k[j - 1]; a.v[j] = a.v[j - 1]; }  // move_backward
      a.k[first] = vk; a.v[first] = vv;
    } else {
      kv_unguarded_linear_insert(a, i);
    }
  }
}

// std::sort(first, last, comp) of libstdc++ on n elements starting at index 0.
DVM_HD void kv_std_sort(KV a, int n) {
  if (n <= 1) {
    return;
  }
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) lg++;
  // explicit stack instead of the recursion on the right part (disjoint ranges: same result)
  int sf[48], sl[48], sd[48];
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
        if (a.k[x] < a.k[y]) {
          if (a.k[y] < a.k[z]) kv_swap(a, r, y);
          else if (a.k[x] < a.k[z]) kv_swap(a, r, z);
          else kv_swap(a, r, x);
        } else if (a.k[x] < a.k[z]) kv_swap(a, r, x);
        else if (a.k[y] < a.k[z]) kv_swap(a, r, z);
        else kv_swap(a, r, y);
      }
      int lo = first + 1, hi = last;
      const uint32_t pk = a.k[first];
      while (true) {
        while (a.k[lo] < pk) ++lo;
        --hi;
        while (pk < a.k[hi]) --hi;
        if (!(lo < hi)) break;
        kv_swap(a, lo, hi);
        ++lo;
      }
      const int cut = lo;
      sf[sp] = cut; sl[sp] = last; sd[sp] = depth; sp++;
      last = cut;
    }
  }
  // __final_insertion_sort
  if (n > 16) {
    kv_insertion_sort(a, 0, 16);
    for (int i = 16; i != n; ++i) kv_unguarded_linear_insert(a, i);
  } else {
    kv_insertion_sort(a, 0, n);
  }
}

}  // namespace dvm
