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
#include "blur_tile.h"

namespace dvm {

constexpr int kOctMaxNodes = 2688;   // 58 B of dynamic LDS per node slot: 152 KB of the 160 KB a gfx950 workgroup may own (+ ~3 KB static)
struct ONodeRec {
  int16_t x0, y0, x1, y1;
  int32_t cnt;
};
struct OSplitRec {   // a node that has been split, in the slot of its ONodeRec: split point and the list indices of its children (-1: none)
  int16_t xm, ym;
  int16_t child[4];
};
static_assert(sizeof(OSplitRec) == sizeof(ONodeRec), "a split record replaces the node's record in place");
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

// exclusive prefix of v over the 64 lanes; total = the wave's sum (uniform)
__device__ __forceinline__ int wave_excl_scan_i32(int v, int& total) {
  const int lane = threadIdx.x & 63;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  total = __shfl(x, 63, 64);
  return x - v;
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

#ifdef DVM_OCT_PROF
__device__ unsigned long long g_oct_prof[32];
#define OCT_T(k) do { if (tid == 0 && level == 0 && f == 0) { const unsigned long long now_ = wall_clock64(); g_oct_prof[k] += now_ - oct_last_; oct_last_ = now_; } } while (0)
#else
#define OCT_T(k) do {} while (0)
#endif
// (1 024 threads for the latency variant: measured slower, 41 us against 37 -- a pass over the keys is bound by the CU's LDS
// throughput under random access, ~11 LDS operations per key, not by the latency of one key's chain)
extern __shared__ __attribute__((aligned(16))) unsigned char oct_smem[];
template <bool LAT>
__device__ __forceinline__ void octree_level(const uint32_t* __restrict__ cand_slots, const int32_t* __restrict__ cell_count,
                                             const CellDesc* __restrict__ cells, uint32_t* __restrict__ dense,
                                             int32_t* __restrict__ lvl_count, const PipelineDesc& PD,
                                             int32_t* __restrict__ nid_scratch, uint32_t* __restrict__ sel,
                                             int32_t* __restrict__ nsel, int32_t* __restrict__ err_flag, int cap,
                                             int level, int f, int cap_n) {
  // dynamic LDS, `cap` node slots (host: max level quota + 8, >= 4 * root nodes, multiple of 64): 58 B per slot in the throughput
  // form (the BASELINE config, cap 256, keeps ~15 KB and 8 workgroups fit a CU), 56 B per slot + the keys in the latency form
  ONodeRec* listA = reinterpret_cast<ONodeRec*>(oct_smem);
  ONodeRec* listB = listA + cap;
  int(*cc)[4] = reinterpret_cast<int(*)[4]>(listB + cap);   // child key counts of the processed nodes
  int* a_scan = reinterpret_cast<int*>(cc + cap);            // kept node -> index in the new list / scan workspace
  // throughput form (octree_rounds.inc)
  int* b_scan = nullptr;              // scan workspace 2 (children per processed node)
  int16_t* prank = nullptr;           // list index -> rank in processing order or -1
  // latency form (octree_rounds_wave.inc)
  int16_t *prankA = nullptr, *prankB = nullptr;   // the same for two rounds in flight
  lds_u32* cand_l = nullptr;          // cap_n keys + node ids behind the node slots
  lds_i32* nid_l = nullptr;
  uint32_t* skey;                     // sort keys (wave form also: first child slot per processed node, best key per node)
  int16_t* proc;                      // processing order: list indices
  uint16_t* sval;                     // sort payload / expandable set (creation order)
  if constexpr (LAT) {
    skey = reinterpret_cast<uint32_t*>(a_scan + cap);
    proc = reinterpret_cast<int16_t*>(skey + cap);
    prankA = proc + cap;
    prankB = prankA + cap;
    sval = reinterpret_cast<uint16_t*>(prankB + cap);
    cand_l = (lds_u32*)(oct_smem + ((cap * 56 + 15) & ~15));   // (56 B per node slot above: octree_lat_lds_bytes)
    nid_l = reinterpret_cast<lds_i32*>(cand_l + cap_n);
  } else {
    b_scan = a_scan + cap;
    skey = reinterpret_cast<uint32_t*>(b_scan + cap);
    proc = reinterpret_cast<int16_t*>(skey + cap);
    prank = proc + cap;
    sval = reinterpret_cast<uint16_t*>(prank + cap);
  }
  constexpr int NT = 256;
  __shared__ int s_tmp[256];
  __shared__ int s_stack[144];
  __shared__ int s_m, s_np, s_ne, s_total, s_keep, s_phase, s_finish, s_cut;

  const int tid = threadIdx.x;
#ifdef DVM_OCT_PROF
  unsigned long long oct_last_ = wall_clock64();
#endif
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
        for (int k = 0; k < cn; k++) {
          const uint32_t v = sp[k];
          cand_w[start + k] = v;
          if constexpr (LAT) { if (start + k < cap_n) cand_l[start + k] = v; }
        }
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
  OCT_T(0);
  uint32_t* out = sel + (int64_t)f * PD.sel_frame_slots + LV.sel_off;
  int32_t* out_n = nsel + f * PD.nlevels + level;
  const int N = LV.quota;
  if (n <= 0) {
    if (tid == 0) *out_n = 0;
    return;
  }
  if constexpr (LAT) {
    if (n <= cap_n) {   // the level's keys and their node ids stay in LDS
      const lds_u32* cand = cand_l;
      lds_i32* nid = nid_l;
#include "octree_rounds_wave.inc"
    } else {
      const uint32_t* cand = cand_w;
      int32_t* nid = nid_scratch + (int64_t)f * PD.cand_frame_slots + LV.cand_off;
#include "octree_rounds_wave.inc"
    }
  } else {
    const uint32_t* cand = cand_w;
    int32_t* nid = nid_scratch + (int64_t)f * PD.cand_frame_slots + LV.cand_off;
#include "octree_rounds.inc"
  }
}

// (80 SGPRs: a 256-thread workgroup is admitted per CU up to 800 / (ceil(sgpr / 16) * 16 + 16) times -- 6 at the 101 the compiler
// takes on its own, 8 at <= 80, which is also what the 17.5 KB of LDS allow: all 2 048 workgroups of a 256-frame batch resident at once)
template <bool LAT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) k_octree(const uint32_t* __restrict__ cand_slots, const int32_t* __restrict__ cell_count,
                                                const CellDesc* __restrict__ cells, uint32_t* __restrict__ dense,
                                                int32_t* __restrict__ lvl_count, PipelineDesc PD,
                                                int32_t* __restrict__ nid_scratch, uint32_t* __restrict__ sel,
                                                int32_t* __restrict__ nsel, int32_t* __restrict__ err_flag, int cap,
                                                int level_first, int cap_n) {
  octree_level<LAT>(cand_slots, cell_count, cells, dense, lvl_count, PD, nid_scratch, sel, nsel, err_flag, cap, level_first + (int)blockIdx.x,
                    (int)blockIdx.y, cap_n);
}
// One-frame path: the octree of every level AND the 7x7 blur of the whole pyramid in one launch.  The blur depends on the pyramid only;
// as a launch of its own it was 9 us on the chain (or a side stream whose fork / join events cost more than that).  Here its tiles are
// the workgroups behind the octree's (one per level, dispatched first): they fill the chip the octree's handful of
// latency-bound workgroups leave idle and are done long before those are.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) k_octree_blur(const uint32_t* __restrict__ cand_slots, const int32_t* __restrict__ cell_count,
                                                const CellDesc* __restrict__ cells, uint32_t* __restrict__ dense,
                                                int32_t* __restrict__ lvl_count, PipelineDesc PD,
                                                int32_t* __restrict__ nid_scratch, uint32_t* __restrict__ sel,
                                                int32_t* __restrict__ nsel, int32_t* __restrict__ err_flag, int cap, int cap_n,
                                                const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, const TileDesc* __restrict__ tiles,
                                                uint4 gauss) {
  const int b = (int)blockIdx.x, f = (int)blockIdx.y;   // x: the frame's levels, then its blur tiles
  if (b < PD.nlevels) {
    octree_level<true>(cand_slots, cell_count, cells, dense, lvl_count, PD, nid_scratch, sel, nsel, err_flag, cap, b, f, cap_n);
  } else {
    __shared__ __attribute__((aligned(16))) uint8_t raw[(kBlurTH + 6) * kRawPitch];
    __shared__ __attribute__((aligned(16))) uint32_t hpt[kBlurTW * kHtPitch];
    blur_tile(pyr, PD.pyr_frame_bytes, blur, PD.blur_frame_bytes, tiles[b - PD.nlevels], PD, f, gauss.x, gauss.y, gauss.z, gauss.w, raw, hpt);
  }
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

static size_t octree_lds_bytes(int cap) { return (size_t)cap * (2 * sizeof(ONodeRec) + 16 + 4 + 4 + 4 + 2 + 2 + 2); }   // throughput form: 58 B / slot
static_assert(2 * sizeof(ONodeRec) + 16 + 4 + 4 + 2 + 2 + 2 + 2 == 56, "k_octree<true> places its key arrays after 56 B per node slot");
// latency variant: keys + node ids of a level in LDS as well, as many as fit next to the node slots (8 B per key)
constexpr int kOctLatKeys = 8192;
constexpr int kOctBlurKeys = 4096;   // k_octree_blur: its blur workgroups reserve the launch's LDS too -- 32 KB of keys keeps two of them on a CU
static int octree_lat_keys(int cap) {
  const size_t base = ((size_t)cap * 56 + 15) & ~(size_t)15;
  const size_t room = base < 156 * 1024 ? 156 * 1024 - base : 0;
  return (int)std::min<size_t>(kOctLatKeys, room / 8);
}
static size_t octree_lat_lds_bytes(int cap) { return (((size_t)cap * 56 + 15) & ~(size_t)15) + (size_t)octree_lat_keys(cap) * 8; }

// Called once per configuration (OrbPipeline::configure), on the handle's device: beyond the default 48 KB of dynamic
// LDS the limit has to be raised per device, and a configuration the hardware cannot hold must be refused BEFORE anything
// is launched (a failed k_octree launch would leave the selection arrays of the following kernels undefined).
bool octree_prepare_device(const PipelineDesc& PD) {
  if (!octree_fits_device(PD)) return false;
  const size_t bytes = octree_lds_bytes(octree_required_nodes(PD));
  if (!raise_dynamic_lds(reinterpret_cast<const void*>(k_octree<true>), (int)octree_lat_lds_bytes(octree_required_nodes(PD)))) return false;
  if (octree_blur_fits(PD) && !raise_dynamic_lds(reinterpret_cast<const void*>(k_octree_blur), 64 * 1024)) return false;
  if (bytes <= 48 * 1024) return true;
  return raise_dynamic_lds(reinterpret_cast<const void*>(k_octree<false>), (int)bytes);
}

void launch_octree(hipStream_t s, const uint32_t* d_cand, const int32_t* d_cell_count, const CellDesc* d_cells, uint32_t* d_dense,
                   int32_t* d_lvl_count, const PipelineDesc& PD, int32_t* d_nid, uint32_t* d_sel, int32_t* d_nsel, int32_t* d_err,
                   int batch, int level_first, int level_num, bool latency) {
  if (level_num <= 0) return;
  const int cap = std::min(octree_required_nodes(PD), kOctMaxNodes);
  if (latency) {
    hipLaunchKernelGGL(k_octree<true>, dim3(level_num, batch), dim3(256), octree_lat_lds_bytes(cap), s, d_cand, d_cell_count, d_cells, d_dense,
                       d_lvl_count, PD, d_nid, d_sel, d_nsel, d_err, cap, level_first, octree_lat_keys(cap));
    return;
  }
  hipLaunchKernelGGL(k_octree<false>, dim3(level_num, batch), dim3(256), octree_lds_bytes(cap), s, d_cand, d_cell_count, d_cells, d_dense,
                     d_lvl_count, PD, d_nid, d_sel, d_nsel, d_err, cap, level_first, 0);
}

#ifdef DVM_OCT_PROF
extern "C" int dvm_debug_oct_prof(unsigned long long* out, int reset) {
  unsigned long long h[32];
  int rc = (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_oct_prof), sizeof(h));
  for (int i = 0; i < 32; i++) out[i] = h[i];
  if (reset) { for (auto& v : h) v = 0; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_oct_prof), h, sizeof(h)); }
  return rc;
}
#endif
// k_octree_blur reserves, per workgroup, the octree's dynamic LDS plus the blur's and the octree's static arrays: taken only while two
// workgroups still fit a CU (small node capacities -- the usual configurations); otherwise the two launches stay separate
bool octree_blur_fits(const PipelineDesc& PD) {
  const int cap = std::min(octree_required_nodes(PD), kOctMaxNodes);
  const int keys = std::min(octree_lat_keys(cap), kOctBlurKeys);
  const size_t lds = (((size_t)cap * 56 + 15) & ~(size_t)15) + (size_t)keys * 8 + 20 * 1024;
  return keys >= 1024 && lds <= 76 * 1024;
}
// the one-frame path's launch: octrees of all levels + the blur of all levels (k_octree_blur); gauss7: the 8.8 kernel of the blur
void launch_octree_blur(hipStream_t s, const uint32_t* d_cand, const int32_t* d_cell_count, const CellDesc* d_cells, uint32_t* d_dense,
                        int32_t* d_lvl_count, const PipelineDesc& PD, int32_t* d_nid, uint32_t* d_sel, int32_t* d_nsel, int32_t* d_err, int batch,
                        const uint8_t* d_pyr, uint8_t* d_blur, const TileDesc* d_tiles, const int* gauss7) {
  const int cap = std::min(octree_required_nodes(PD), kOctMaxNodes);
  const int keys = std::min(octree_lat_keys(cap), kOctBlurKeys);
  const size_t lds = (((size_t)cap * 56 + 15) & ~(size_t)15) + (size_t)keys * 8;
  hipLaunchKernelGGL(k_octree_blur, dim3(PD.nlevels + PD.ntiles, batch), dim3(256), lds, s, d_cand, d_cell_count, d_cells, d_dense, d_lvl_count, PD,
                     d_nid, d_sel, d_nsel, d_err, cap, keys, d_pyr, d_blur, d_tiles,
                     make_uint4((uint32_t)gauss7[0], (uint32_t)gauss7[1], (uint32_t)gauss7[2], (uint32_t)gauss7[3]));
}

}  // namespace dvm
