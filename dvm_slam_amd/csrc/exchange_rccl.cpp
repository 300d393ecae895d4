// dvm_slam_amd/csrc/exchange_rccl.cpp -- include/dvmslam_rccl.h: the inter-agent exchange over RCCL, for a C++ agent node (the
// reference's is C++: src/slam_system/src/orb_slam3_wrapper.cpp).  Built into libdvmslam_rccl.so (links librccl; libdvmslam_hip.so
// itself stays free of it).  dvm_slam_amd/exchange.py is the same protocol over torch.distributed for the Python harness.
#include "../../include/dvmslam_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

namespace {
thread_local std::string g_err;
int fail(const std::string& what) { g_err = what; return -1; }
int hipc(hipError_t e, const char* what) { return e == hipSuccess ? 0 : fail(std::string(what) + ": " + hipGetErrorString(e)); }
int nc(ncclResult_t r, const char* what) { return r == ncclSuccess ? 0 : fail(std::string(what) + ": " + ncclGetErrorString(r)); }
#define EX_TRY(x) do { const int rc_ = (x); if (rc_ != 0) return rc_; } while (0)
}  // namespace

struct dvm_exchange {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int rank = 0, world = 1;
  void* scratch = nullptr;          // device staging: sizes, host-memory reductions, padded ragged blocks
  size_t scratch_bytes = 0;
  int reserve(size_t bytes) {
    if (bytes <= scratch_bytes) return 0;
    if (scratch) { EX_TRY(hipc(hipStreamSynchronize(stream), "sync")); hipFree(scratch); scratch = nullptr; scratch_bytes = 0; }
    bytes = std::max<size_t>(bytes, 1 << 16);
    EX_TRY(hipc(hipMalloc(&scratch, bytes), "hipMalloc(exchange scratch)"));
    scratch_bytes = bytes;
    return 0;
  }
};

extern "C" {

const char* dvm_exchange_last_error(void) { return g_err.c_str(); }

int dvm_exchange_unique_id(void* id128) {
  if (!id128) return fail("dvm_exchange_unique_id: null");
  static_assert(sizeof(ncclUniqueId) == 128, "the id is a 128-byte blob");
  ncclUniqueId id;
  EX_TRY(nc(ncclGetUniqueId(&id), "ncclGetUniqueId"));
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}
int dvm_exchange_comm_init(const void* id128, int rank, int world, int device, void** out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return fail("dvm_exchange_comm_init: bad arguments");
  EX_TRY(hipc(hipSetDevice(device), "hipSetDevice"));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  EX_TRY(nc(ncclCommInitRank(&c, world, id, rank), "ncclCommInitRank"));
  *out = c;
  return 0;
}
void dvm_exchange_comm_destroy(void* c) { if (c) ncclCommDestroy(static_cast<ncclComm_t>(c)); }

int dvm_exchange_create(void* comm, void* stream, dvm_exchange** out) {
  if (!comm || !out) return fail("dvm_exchange_create: null communicator");
  dvm_exchange* ex = new dvm_exchange();
  ex->comm = static_cast<ncclComm_t>(comm);
  ex->stream = static_cast<hipStream_t>(stream);
  if (nc(ncclCommUserRank(ex->comm, &ex->rank), "ncclCommUserRank") || nc(ncclCommCount(ex->comm, &ex->world), "ncclCommCount")) { delete ex; return -1; }
  *out = ex;
  return 0;
}
void dvm_exchange_destroy(dvm_exchange* ex) {
  if (!ex) return;
  if (ex->scratch) { hipStreamSynchronize(ex->stream); hipFree(ex->scratch); }
  delete ex;
}
int dvm_exchange_rank(const dvm_exchange* ex) { return ex ? ex->rank : -1; }
int dvm_exchange_world(const dvm_exchange* ex) { return ex ? ex->world : -1; }

int dvm_exchange_allgather_blocks(dvm_exchange* ex, const void* d_block, int64_t bytes, void* d_recv) {
  if (!ex || bytes < 0 || (bytes && (!d_block || !d_recv))) return fail("dvm_exchange_allgather_blocks: bad arguments");
  if (bytes == 0) return 0;
  return nc(ncclAllGather(d_block, d_recv, (size_t)bytes, ncclUint8, ex->comm, ex->stream), "ncclAllGather(blocks)");
}

int dvm_exchange_allgather_varlen(dvm_exchange* ex, const void* d_block, int64_t bytes, void* d_recv, int64_t cap, int64_t* sizes_out) {
  if (!ex || bytes < 0 || cap < 0 || !sizes_out || (bytes && !d_block) || (cap && !d_recv)) return fail("dvm_exchange_allgather_varlen: bad arguments");
  // (size, cap) first: two int64 per rank through the scratch buffer.  The slot capacity must be the same on every rank (the second
  // all-gather's count); gathering it with the sizes lets every rank see a mismatch and fail together instead of hanging in RCCL
  const int W = ex->world;
  EX_TRY(ex->reserve(sizeof(int64_t) * (size_t)(2 * W + 2) + (size_t)cap));
  int64_t* d_sizes = static_cast<int64_t*>(ex->scratch);
  const int64_t mine[2] = {bytes, cap};
  std::vector<int64_t> all((size_t)2 * W);
  EX_TRY(hipc(hipMemcpyAsync(d_sizes + 2 * W, mine, 2 * sizeof(int64_t), hipMemcpyHostToDevice, ex->stream), "copy size"));
  EX_TRY(nc(ncclAllGather(d_sizes + 2 * W, d_sizes, 2, ncclInt64, ex->comm, ex->stream), "ncclAllGather(sizes)"));
  EX_TRY(hipc(hipMemcpyAsync(all.data(), d_sizes, sizeof(int64_t) * (size_t)(2 * W), hipMemcpyDeviceToHost, ex->stream), "copy sizes"));
  EX_TRY(hipc(hipStreamSynchronize(ex->stream), "sync"));
  int64_t mx = 0;
  bool caps_agree = true;
  for (int r = 0; r < W; r++) { sizes_out[r] = all[2 * r]; mx = std::max(mx, sizes_out[r]); caps_agree = caps_agree && all[2 * r + 1] == cap; }
  if (!caps_agree) { g_err = "dvm_exchange_allgather_varlen: the ranks passed different slot capacities"; return -3; }
  if (mx > cap) { g_err = "dvm_exchange_allgather_varlen: a block of " + std::to_string(mx) + " bytes exceeds the slot capacity"; return -3; }
  if (cap == 0) return 0;
  // every rank contributes a full slot (its block, zero padded): staged behind the sizes in the scratch buffer
  uint8_t* d_pad = reinterpret_cast<uint8_t*>(d_sizes + 2 * W + 2);
  EX_TRY(hipc(hipMemsetAsync(d_pad, 0, (size_t)cap, ex->stream), "pad"));
  if (bytes) EX_TRY(hipc(hipMemcpyAsync(d_pad, d_block, (size_t)bytes, hipMemcpyDeviceToDevice, ex->stream), "stage block"));
  return nc(ncclAllGather(d_pad, d_recv, (size_t)cap, ncclUint8, ex->comm, ex->stream), "ncclAllGather(ragged blocks)");
}

int dvm_exchange_send_block(dvm_exchange* ex, const void* d_block, int64_t bytes, int peer) {
  if (!ex || bytes < 0 || peer < 0 || peer >= ex->world || (bytes && !d_block)) return fail("dvm_exchange_send_block: bad arguments");
  return nc(ncclSend(d_block, (size_t)bytes, ncclUint8, peer, ex->comm, ex->stream), "ncclSend");
}
int dvm_exchange_recv_block(dvm_exchange* ex, void* d_block, int64_t bytes, int peer) {
  if (!ex || bytes < 0 || peer < 0 || peer >= ex->world || (bytes && !d_block)) return fail("dvm_exchange_recv_block: bad arguments");
  return nc(ncclRecv(d_block, (size_t)bytes, ncclUint8, peer, ex->comm, ex->stream), "ncclRecv");
}
int dvm_exchange_broadcast(dvm_exchange* ex, void* d_buf, int64_t bytes, int root) {
  if (!ex || bytes < 0 || root < 0 || root >= ex->world || (bytes && !d_buf)) return fail("dvm_exchange_broadcast: bad arguments");
  if (bytes == 0) return 0;
  return nc(ncclBroadcast(d_buf, d_buf, (size_t)bytes, ncclUint8, root, ex->comm, ex->stream), "ncclBroadcast");
}
int dvm_exchange_broadcast_sim3(dvm_exchange* ex, double* d_sim3, int root) { return dvm_exchange_broadcast(ex, d_sim3, 8 * (int64_t)sizeof(double), root); }

int dvm_exchange_allreduce(void* ctx, void* buf, int64_t n, int on_host, int op, void* stream) {
  dvm_exchange* ex = static_cast<dvm_exchange*>(ctx);
  if (!ex || n < 0 || (n && !buf) || (op != 0 && op != 1)) return fail("dvm_exchange_allreduce: bad arguments");
  if (n == 0) return 0;
  const ncclRedOp_t rop = op == 0 ? ncclSum : ncclMax;
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ex->stream;
  if (!on_host) return nc(ncclAllReduce(buf, buf, (size_t)n, ncclDouble, rop, ex->comm, s), "ncclAllReduce");
  EX_TRY(ex->reserve(sizeof(double) * (size_t)n));
  EX_TRY(hipc(hipMemcpyAsync(ex->scratch, buf, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s), "stage in"));
  EX_TRY(nc(ncclAllReduce(ex->scratch, ex->scratch, (size_t)n, ncclDouble, rop, ex->comm, s), "ncclAllReduce(host)"));
  EX_TRY(hipc(hipMemcpyAsync(buf, ex->scratch, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s), "stage out"));
  return hipc(hipStreamSynchronize(s), "sync");
}
int dvm_exchange_max_over_ranks(dvm_exchange* ex, double* value) {
  if (!ex || !value) return fail("dvm_exchange_max_over_ranks: bad arguments");
  return dvm_exchange_allreduce(ex, value, 1, 1, 1, ex->stream);
}

}  // extern "C"
