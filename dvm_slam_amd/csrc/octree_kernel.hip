// dvm_slam_amd/csrc/octree_kernel.hip -- ORBextractor::DistributeOctTree on the device.
//
// Reference: src/ORBextractor.cc:419-610 (+ ExtractorNode::DivideNode :348-400, compareNodes :402-417).
// One workgroup per (pyramid level, frame).  The std::list surgery is restated as array rounds
// (DESIGN.md section 6): a round splits the node sequence P (processing order) and yields
//     list' = reverse(children(p1) ++ ... ++ children(pm)) ++ (list \ P).
//   phase 1: P = all nodes with > 1 key, in list order (the reference's full sweeps)
//   phase 2: P = nodes created by the previous round with > 1 key, ordered by libstdc++'s
//            std::sort(compareNodes) walked from the back, cut as soon as the list holds >= N nodes
// Keys keep a node id; child sizes come from LDS atomics over all candidates (order-free counts).
// The retained key per node is max response, first in vToDistributeKeys order on ties (:594-607),
// i.e. atomicMax of (response << 20 | ~index).  std::sort's treatment of EQUAL (size, UL.x) keys is
// reproduced by the step-exact emulation in introsort_emul.h (one wavefront, partition in rank form).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "introsort_emul.h"
#include "orb_device.h"
#include "orb_kernels.h"

namespace dvm {

constexpr int kOctMaxNodes = 2688;   // 58 B of dynamic LDS per node slot: 152 KB of the 160 KB a gfx950 workgroup may own (+ ~3 KB static)
struct ONodeRec {
  int16_t x0, y0, x1, y1;
  int32_t cnt;
};
// LDS-typed accessor for introsort_emul.h: with plain (generic) pointers the serial sort compiles to
// flat_load/flat_store, several times the latency of ds_read/ds_write -- and that latency IS the
// critical path of this kernel.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) int lds_i32;
struct KVLds {
  lds_u32* k;
  lds_u16* v;
  __device__ __forceinline__ uint32_t key(int i) const { return k[i]; }
  __device__ __forceinline__ uint32_t val(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, uint32_t kk, uint32_t vv) { k[i] = kk; v[i] = (uint16_t)vv; }
};

__device__ __forceinline__ void wave_lds_sync() {   // order this wave's LDS traffic across lanes
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// libstdc++ __introsort_loop (introsort_emul.h, kv_introsort_loop_ranked) executed by ONE wavefront: the control flow
// is wave-uniform, the median swap is a handful of broadcast reads, and __unguarded_partition runs in rank form --
// ballots give every element its rank among the ">= pivot" positions (ascending) and the "<= pivot" positions
// (descending), the k-th pair is swapped by lane k.  Same array as the serial loop after every step
// (tools/check_introsort.cpp), ~10 dependent LDS operations per partition instead of ~5 per ELEMENT.
// key/val: the array; I, J: scratch of n ints each; stk: 144 ints.  All in LDS.
__device__ void wave_introsort_loop(lds_u32* key, lds_u16* val, int n, lds_i32* stk, lds_i32* I, lds_i32* J) {
  const int lane = threadIdx.x & 63;
  if (n <= 1) return;
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) lg++;
  lds_i32 *sf = stk, *sl = stk + 48, *sd = stk + 96;
  int sp = 0;
  sf[sp] = 0; sl[sp] = n; sd[sp] = 2 * lg; sp++;   // every lane writes the same value
  wave_lds_sync();
  while (sp > 0) {
    --sp;
    int first = sf[sp], last = sl[sp], depth = sd[sp];
    while (last - first > 16) {
      if (depth == 0) {
        if (lane == 0) { KVLds kv{key, val}; kv_heapsort(kv, first, last); }
        wave_lds_sync();
        break;
      }
      --depth;
      {  // __move_median_to_first(first, first+1, mid, last-1)
        const int mid = first + (last - first) / 2;
        const int x = first + 1, y = mid, z = last - 1;
        const uint32_t kx = key[x], ky = key[y], kz = key[z];
        int t;
        if (kx < ky) t = (ky < kz) ? y : ((kx < kz) ? z : x);
        else t = (kx < kz) ? x : ((ky < kz) ? z : y);
        if (lane == 0) {
          const uint32_t k0 = key[first], kt = key[t];
          const uint16_t v0 = val[first], vt = val[t];
          key[first] = kt; val[first] = vt; key[t] = k0; val[t] = v0;
        }
        wave_lds_sync();
      }
      const uint32_t pk = key[first];
      int totalLE = 0;
      for (int base = first; base < last; base += 64) {
        const int p = base + lane;
        const bool l = p < last && !(pk < key[p]);
        totalLE += __popcll(__ballot(l));
      }
      int nI = 0, nL = 0;
      for (int base = first; base < last; base += 64) {
        const int p = base + lane;
        const bool valid = p < last;
        const uint32_t k = valid ? key[p] : 0u;
        const bool g = valid && p > first && !(k < pk), l = valid && !(pk < k);
        const unsigned long long bg = __ballot(g), bl = __ballot(l);
        if (g) I[nI + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bg >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bg, 0u))] = p;
        if (l) J[totalLE - 1 - (nL + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bl >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bl, 0u)))] = p;
        nI += __popcll(bg); nL += __popcll(bl);
      }
      wave_lds_sync();
      const int npair = min(nI, totalLE);
      int m = 0;
      for (int base = 0;; base += 64) {
        const int k = base + lane;
        const bool t = k < npair && I[k] < J[k];
        const unsigned long long b = __ballot(t);
        if (b == ~0ull) { m += 64; continue; }
        m += __builtin_ctzll(~b);
        break;
      }
      for (int base = 0; base < m; base += 64) {
        const int k = base + lane;
        if (k < m) {
          const int i = I[k], j = J[k];
          const uint32_t ki = key[i], kj = key[j];
          const uint16_t vi = val[i], vj = val[j];
          key[i] = kj; val[i] = vj; key[j] = ki; val[j] = vi;
        }
      }
      int cut = 0x7fffffff;
      if (m < nI) cut = I[m];
      if (m > 0) cut = min(cut, (int)J[m - 1]);
      wave_lds_sync();
      sf[sp] = cut; sl[sp] = last; sd[sp] = depth; sp++;
      wave_lds_sync();
      last = cut;
    }
  }
}

// exclusive scan of data[0..m) in place, 256 threads (4 waves): per-thread serial chunk, wave scan by
// DPP-free shuffles, 4 wave totals through LDS.  *total (shared) receives the sum.  3 barriers.
__device__ __forceinline__ void block_scan_excl(int* data, int m, int* s_tmp, int* total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (m + 255) / 256;
  const int b = tid * per, e = min(b + per, m);
  int sum = 0;
  for (int i = b; i < e; i++) sum += data[i];
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) s_tmp[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; w++) base += s_tmp[w];
  int run = base + incl - sum;
  for (int i = b; i < e; i++) { int v = data[i]; data[i] = run; run += v; }
  if (tid == 255) *total = base + incl;
  __syncthreads();
}

// (80 SGPRs: a 256-thread workgroup is admitted per CU up to 800 / (ceil(sgpr / 16) * 16 + 16) times -- 6 at the 101 the compiler
// takes on its own, 8 at <= 80, which is also what the 17.5 KB of LDS allow: all 2 048 workgroups of a 256-frame batch resident at once)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) k_octree(const uint32_t* __restrict__ cand_slots, const int32_t* __restrict__ cell_count,
                                                const CellDesc* __restrict__ cells, uint32_t* __restrict__ dense,
                                                int32_t* __restrict__ lvl_count, PipelineDesc PD,
                                                int32_t* __restrict__ nid_scratch, uint32_t* __restrict__ sel,
                                                int32_t* __restrict__ nsel, int32_t* __restrict__ err_flag, int cap,
                                                int level_first) {
  // dynamic LDS, `cap` node slots (host: max level quota + 8, >= 4 * root nodes, multiple of 64):
  // 58 B per slot, so the BASELINE config (cap 256) keeps ~15 KB and 8 workgroups fit a CU
  extern __shared__ __attribute__((aligned(16))) unsigned char oct_smem[];
  ONodeRec* listA = reinterpret_cast<ONodeRec*>(oct_smem);
  ONodeRec* listB = listA + cap;
  int(*cc)[4] = reinterpret_cast<int(*)[4]>(listB + cap);   // child key counts, later child new-index
  int* a_scan = reinterpret_cast<int*>(cc + cap);            // scan workspace 1 (keep index / flags)
  int* b_scan = a_scan + cap;                                // scan workspace 2 (children per processed node)
  uint32_t* skey = reinterpret_cast<uint32_t*>(b_scan + cap);  // sort keys
  int16_t* proc = reinterpret_cast<int16_t*>(skey + cap);    // processing order: list indices
  int16_t* prank = proc + cap;                               // list index -> rank in processing order or -1
  uint16_t* sval = reinterpret_cast<uint16_t*>(prank + cap); // sort payload / expandable set (creation order)
  __shared__ int s_tmp[256];
  __shared__ int s_stack[144];
  __shared__ int s_m, s_np, s_ne, s_total, s_keep, s_phase, s_finish, s_cut;

  const int level = level_first + blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
  const LevelDesc& LV = PD.lv[level];
  // ---- vToDistributeKeys of this level: the per-cell candidate lists (k_fast_cells) concatenated in the reference's
  // cell loop order, into the level's own range of `dense` (this used to be a separate per-frame kernel; doing it here
  // makes a level's octree depend on that level's FAST cells only)
  uint32_t* cand_w = dense + (int64_t)f * PD.cand_frame_slots + LV.cand_off;
  __shared__ int s_carry;
  {
    __shared__ int s_cnt[256];
    if (tid == 0) s_carry = 0;
    __syncthreads();
    const uint32_t* src = cand_slots + (int64_t)f * PD.cand_frame_slots;
    for (int c0 = 0; c0 < LV.cell_count; c0 += 256) {
      const int ci = LV.cell_first + c0 + tid;
      const int cn = (c0 + tid < LV.cell_count) ? cell_count[(int64_t)f * PD.ncells + ci] : 0;
      s_cnt[tid] = cn;
      __syncthreads();
      block_scan_excl(s_cnt, 256, s_tmp, &s_total);
      const int start = s_carry + s_cnt[tid];
      if (cn > 0) {
        const uint32_t* sp = src + cells[ci].cand_base;
        for (int k = 0; k < cn; k++) cand_w[start + k] = sp[k];
      }
      __syncthreads();
      if (tid == 0) s_carry += s_total;
      __syncthreads();
    }
    if (tid == 0) lvl_count[f * kMaxLevels + level] = s_carry;
    __threadfence_block();   // the candidates written above are re-read below by other threads of this workgroup (the
                             // level's range owns its cache lines: cand_off is 128-byte aligned, so no stale L1 line)
    __syncthreads();
  }
  const int n = s_carry;
  const uint32_t* cand = cand_w;
  int32_t* nid = nid_scratch + (int64_t)f * PD.cand_frame_slots + LV.cand_off;
  uint32_t* out = sel + (int64_t)f * PD.sel_frame_slots + LV.sel_off;
  int32_t* out_n = nsel + f * PD.nlevels + level;
  const int N = LV.quota;
  if (n <= 0) {
    if (tid == 0) *out_n = 0;
    return;
  }
  const int W = (LV.w - kEdge + 3) - (kEdge - 3), H = (LV.h - kEdge + 3) - (kEdge - 3);  // maxX-minX, maxY-minY
  const int nIni = (int)roundf((float)W / (float)H);
  if (nIni <= 0 || 4 * nIni > cap || N + 8 > cap) {
    if (tid == 0) { *out_n = 0; __hip_atomic_store(err_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }   // mapped host memory
    return;
  }
  const float hX = (float)W / (float)nIni;
  ONodeRec* L = listA;
  ONodeRec* Lnew = listB;
  // ---- roots (:424-458)
  for (int i = tid; i < nIni; i += 256) {
    ONodeRec r;
    r.x0 = (int16_t)(int)(hX * (float)i);
    r.x1 = (int16_t)(int)(hX * (float)(i + 1));
    r.y0 = 0; r.y1 = (int16_t)H; r.cnt = 0;
    Lnew[i] = r;
    a_scan[i] = 0;
  }
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    int x = (int)(cand[i] & 0xFFFu);
    int r = (int)((float)x / hX);
    nid[i] = r;
    atomicAdd(&Lnew[r].cnt, 1);
  }
  __syncthreads();
  for (int i = tid; i < nIni; i += 256) a_scan[i] = Lnew[i].cnt > 0 ? 1 : 0;
  __syncthreads();
  block_scan_excl(a_scan, nIni, s_tmp, &s_total);
  for (int i = tid; i < nIni; i += 256)
    if (Lnew[i].cnt > 0) L[a_scan[i]] = Lnew[i];
  __syncthreads();
  for (int i = tid; i < n; i += 256) nid[i] = a_scan[nid[i]];
  if (tid == 0) { s_m = s_total; s_phase = 1; s_finish = 0; s_ne = 0; }
  __syncthreads();

  while (true) {
    const int m = s_m;
    const int phase = s_phase;
    // ---- processing order
    if (phase == 1) {
      for (int j = tid; j < m; j += 256) a_scan[j] = L[j].cnt > 1 ? 1 : 0;
      __syncthreads();
      block_scan_excl(a_scan, m, s_tmp, &s_np);
      for (int j = tid; j < m; j += 256)
        if (L[j].cnt > 1) proc[a_scan[j]] = (int16_t)j;
      __syncthreads();
    } else {
      const int ne = s_ne;
      for (int k = tid; k < ne; k += 256) {
        const int j = sval[k];
        skey[k] = ((uint32_t)L[j].cnt << 12) | (uint32_t)(uint16_t)L[j].x0;
      }
      __syncthreads();
      // std::sort = __introsort_loop (the only part that is not a stable sort) on one wavefront in rank form,
      // then __final_insertion_sort == stable ordering of that output, computed in parallel by rank
      if (tid < 64) wave_introsort_loop((lds_u32*)skey, (lds_u16*)sval, ne, (lds_i32*)s_stack, (lds_i32*)a_scan, (lds_i32*)b_scan);
      __syncthreads();
      for (int k = tid; k < ne; k += 256) {
        const uint32_t kk = skey[k];
        int rank = 0;
        for (int j = 0; j < ne; j++) {
          const uint32_t kj = skey[j];
          rank += (kj < kk || (kj == kk && j < k)) ? 1 : 0;
        }
        proc[ne - 1 - rank] = (int16_t)sval[k];  // processed from the back of the sorted vector (:551)
      }
      if (tid == 0) s_np = ne;
      __syncthreads();
    }
    int np = s_np;
    // ---- child sizes of every node in the processing order
    for (int t = tid; t < np; t += 256) { const int j = proc[t]; cc[j][0] = cc[j][1] = cc[j][2] = cc[j][3] = 0; }
    for (int j = tid; j < m; j += 256) prank[j] = -1;
    __syncthreads();
    for (int t = tid; t < np; t += 256) prank[proc[t]] = (int16_t)t;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
      const int j = nid[i];
      if (prank[j] >= 0) {
        const ONodeRec r = L[j];
        const int xm = r.x0 + (r.x1 - r.x0 + 1) / 2, ym = r.y0 + (r.y1 - r.y0 + 1) / 2;  // ceil(float(d)/2)
        const uint32_t c = cand[i];
        const int x = (int)(c & 0xFFFu), y = (int)((c >> 12) & 0xFFFu);
        const int q = (x < xm ? 0 : 1) + (y < ym ? 0 : 2);
        atomicAdd(&cc[j][q], 1);
      }
    }
    __syncthreads();
    // ---- children per processed node (processing order); phase 2: cut where the list reaches N
    for (int t = tid; t < np; t += 256) {
      const int j = proc[t];
      b_scan[t] = (cc[j][0] > 0) + (cc[j][1] > 0) + (cc[j][2] > 0) + (cc[j][3] > 0);
    }
    __syncthreads();
    if (phase == 2) {
      // inclusive gain scan: size after processing t = m + sum_{t'<=t} (nchild - 1)
      for (int t = tid; t < np; t += 256) a_scan[t] = b_scan[t] - 1;
      if (tid == 0) s_cut = np;
      __syncthreads();
      block_scan_excl(a_scan, np, s_tmp, &s_total);
      for (int t = tid; t < np; t += 256)
        if (m + a_scan[t] + (b_scan[t] - 1) >= N) atomicMin(&s_cut, t + 1);
      __syncthreads();
      np = s_cut;
      for (int t = np + tid; t < s_np; t += 256) prank[proc[t]] = -1;  // not reached: stay in the list
      __syncthreads();
    }
    block_scan_excl(b_scan, np, s_tmp, &s_total);
    const int totalC = s_total;
    for (int j = tid; j < m; j += 256) a_scan[j] = prank[j] < 0 ? 1 : 0;
    __syncthreads();
    block_scan_excl(a_scan, m, s_tmp, &s_keep);
    const int newSize = totalC + s_keep;
    if (newSize > cap) {
      if (tid == 0) { *out_n = 0; __hip_atomic_store(err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
      return;
    }
    // ---- build the new list
    for (int j = tid; j < m; j += 256)
      if (prank[j] < 0) { Lnew[totalC + a_scan[j]] = L[j]; a_scan[j] = totalC + a_scan[j]; }
    for (int t = tid; t < np; t += 256) {
      const int j = proc[t];
      const ONodeRec r = L[j];
      const int xm = r.x0 + (r.x1 - r.x0 + 1) / 2, ym = r.y0 + (r.y1 - r.y0 + 1) / 2;
      int posC = b_scan[t];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = cc[j][q];
        if (c > 0) {
          ONodeRec ch;
          ch.x0 = (q & 1) ? (int16_t)xm : r.x0;
          ch.x1 = (q & 1) ? r.x1 : (int16_t)xm;
          ch.y0 = (q & 2) ? (int16_t)ym : r.y0;
          ch.y1 = (q & 2) ? r.y1 : (int16_t)ym;
          ch.cnt = c;
          const int ni = totalC - 1 - posC;
          Lnew[ni] = ch;
          cc[j][q] = ni;
          posC++;
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
      const int j = nid[i];
      if (prank[j] >= 0) {
        const ONodeRec r = L[j];
        const int xm = r.x0 + (r.x1 - r.x0 + 1) / 2, ym = r.y0 + (r.y1 - r.y0 + 1) / 2;
        const uint32_t c = cand[i];
        const int x = (int)(c & 0xFFFu), y = (int)((c >> 12) & 0xFFFu);
        nid[i] = cc[j][(x < xm ? 0 : 1) + (y < ym ? 0 : 2)];
      } else {
        nid[i] = a_scan[j];
      }
    }
    __syncthreads();
    // ---- nodes created this round with > 1 key, in creation order (posC ascending = index descending)
    for (int p = tid; p < totalC; p += 256) b_scan[p] = Lnew[totalC - 1 - p].cnt > 1 ? 1 : 0;
    __syncthreads();
    block_scan_excl(b_scan, totalC, s_tmp, &s_ne);
    for (int p = tid; p < totalC; p += 256)
      if (Lnew[totalC - 1 - p].cnt > 1) sval[b_scan[p]] = (uint16_t)(totalC - 1 - p);
    __syncthreads();
    if (tid == 0) {
      const int nToExpand = s_ne;
      if (newSize >= N || newSize == m) s_finish = 1;
      else if (phase == 1 && newSize + nToExpand * 3 > N) s_phase = 2;
      s_m = newSize;
    }
    ONodeRec* tsw = L; L = Lnew; Lnew = tsw;
    __syncthreads();
    if (s_finish) break;
  }

  // ---- keep the best key of every node (:594-607)
  const int m = s_m;
  for (int j = tid; j < m; j += 256) a_scan[j] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    const uint32_t key = ((cand[i] >> 24) << 20) | (0xFFFFFu - (uint32_t)i);
    atomicMax(reinterpret_cast<unsigned int*>(&a_scan[nid[i]]), key);
  }
  __syncthreads();
  const int mo = min(m, LV.sel_cap);
  for (int j = tid; j < mo; j += 256) out[j] = cand[0xFFFFFu - ((uint32_t)a_scan[j] & 0xFFFFFu)];
  if (tid == 0) *out_n = mo;
}

// root nodes of a level: nIni = round((maxX - minX) / (maxY - minY)) (ORBextractor.cc:423)
int octree_root_nodes(const LevelDesc& L) {
  const int W = L.w - 2 * (kEdge - 3), H = L.h - 2 * (kEdge - 3);
  return H > 0 ? (int)roundf((float)W / (float)H) : 0;
}
// node slots k_octree needs for this pyramid (a level ends with up to max(quota + 2, 4 * nIni) nodes: the first
// sweep splits every root unconditionally, later sweeps are guarded by the "size + 3 * nToExpand > N" test)
int octree_required_nodes(const PipelineDesc& PD) {
  int need = 64;
  for (int l = 0; l < PD.nlevels; l++) need = std::max(need, std::max(PD.lv[l].quota + 8, 4 * octree_root_nodes(PD.lv[l]) + 4));
  return (need + 63) / 64 * 64;
}
bool octree_fits_device(const PipelineDesc& PD) { return octree_required_nodes(PD) <= kOctMaxNodes; }

static size_t octree_lds_bytes(int cap) { return (size_t)cap * (2 * sizeof(ONodeRec) + 16 + 4 + 4 + 4 + 2 + 2 + 2); }

// Called once per configuration (OrbPipeline::configure), on the handle's device: beyond the default 48 KB of dynamic
// LDS the limit has to be raised per device, and a configuration the hardware cannot hold must be refused BEFORE anything
// is launched (a failed k_octree launch would leave the selection arrays of the following kernels undefined).
bool octree_prepare_device(const PipelineDesc& PD) {
  if (!octree_fits_device(PD)) return false;
  const size_t bytes = octree_lds_bytes(octree_required_nodes(PD));
  if (bytes <= 48 * 1024) return true;
  return raise_dynamic_lds(reinterpret_cast<const void*>(k_octree), (int)bytes);
}

void launch_octree(hipStream_t s, const uint32_t* d_cand, const int32_t* d_cell_count, const CellDesc* d_cells, uint32_t* d_dense,
                   int32_t* d_lvl_count, const PipelineDesc& PD, int32_t* d_nid, uint32_t* d_sel, int32_t* d_nsel, int32_t* d_err,
                   int batch, int level_first, int level_num) {
  if (level_num <= 0) return;
  const int cap = std::min(octree_required_nodes(PD), kOctMaxNodes);
  hipLaunchKernelGGL(k_octree, dim3(level_num, batch), dim3(256), octree_lds_bytes(cap), s, d_cand, d_cell_count, d_cells, d_dense,
                     d_lvl_count, PD, d_nid, d_sel, d_nsel, d_err, cap, level_first);
}

}  // namespace dvm
