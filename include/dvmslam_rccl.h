/* include/dvmslam_rccl.h -- the inter-agent exchange of a DVM-SLAM agent node over RCCL (xGMI), native C ABI (libdvmslam_rccl.so).
 *
 * One agent per rank / GPU; the per-frame hot path has NO collective.  What travels between agents is what the reference's agent node
 * already ships over ROS 2 topics and services (src/slam_system/src/orb_slam3_wrapper.cpp):
 *   C1/C2  new keyframes -- BoW vectors, keypoints, descriptors, poses, map points -- as serialized blocks
 *          (:212-384 publish / receive new keyframes in batches of >= 5, :359-370 the serialized payload, :524-528 the receiving side)
 *          -> dvm_exchange_allgather_blocks (fixed stride) / dvm_exchange_allgather_varlen (DVMW blocks of include/dvmslam_wire.h, ragged),
 *             dvm_exchange_send_block / dvm_exchange_recv_block for one peer;
 *   C4     the Sim3 change of reference frame a successful merge broadcasts, and merge-state flags (:920-949)
 *          -> dvm_exchange_broadcast_sim3 (8 doubles: s, qx, qy, qz, qw, tx, ty, tz) / dvm_exchange_broadcast (bytes);
 *   config 5 (global BA sharded by landmark, dvm_ba_set_problem_sharded): the all-reduce of the partial reduced camera systems
 *          -> dvm_exchange_allreduce, a ready dvm_allreduce_fn for dvm_ba_set_allreduce (ctx = the dvm_exchange*).
 * The communicator (ncclComm_t) is the host's: dvm_exchange_create borrows it.  For a host that has none yet,
 * dvm_exchange_unique_id / dvm_exchange_comm_init wrap ncclGetUniqueId / ncclCommInitRank (the id travels out of band: one 128-byte blob).
 * All payload pointers are DEVICE memory; every call enqueues on the exchange's stream and returns after the stream has been synchronised
 * only where a host result is produced (sizes, max).  Status: 0 ok, < 0 error (dvm_exchange_last_error).  Not thread-safe per handle. */
#ifndef DVMSLAM_RCCL_H
#define DVMSLAM_RCCL_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct dvm_exchange dvm_exchange;

const char* dvm_exchange_last_error(void);
/* communicator helpers (optional: a host with its own ncclComm_t skips them) */
int dvm_exchange_unique_id(void* id128);
int dvm_exchange_comm_init(const void* id128, int rank, int world, int device, void** nccl_comm_out);
void dvm_exchange_comm_destroy(void* nccl_comm);

int dvm_exchange_create(void* nccl_comm, void* hip_stream, dvm_exchange** out);
void dvm_exchange_destroy(dvm_exchange* ex);
int dvm_exchange_rank(const dvm_exchange* ex);
int dvm_exchange_world(const dvm_exchange* ex);

/* every agent's block of `bytes` bytes into d_recv[world * bytes], rank order */
int dvm_exchange_allgather_blocks(dvm_exchange* ex, const void* d_block, int64_t bytes, void* d_recv);
/* ragged blocks: d_recv has world slots of cap bytes; sizes_out[world] (host) = every agent's size.  Fails with -3 if a block exceeds cap
 * (sizes_out is filled all the same, so that the caller can size the buffer and repeat).  cap must be the same on every rank (it is the
 * count of the collective): the capacities are gathered with the sizes and every rank returns -3 together when they differ. */
int dvm_exchange_allgather_varlen(dvm_exchange* ex, const void* d_block, int64_t bytes, void* d_recv, int64_t cap, int64_t* sizes_out);
int dvm_exchange_send_block(dvm_exchange* ex, const void* d_block, int64_t bytes, int peer);
int dvm_exchange_recv_block(dvm_exchange* ex, void* d_block, int64_t bytes, int peer);
int dvm_exchange_broadcast(dvm_exchange* ex, void* d_buf, int64_t bytes, int root);
int dvm_exchange_broadcast_sim3(dvm_exchange* ex, double* d_sim3, int root);
/* max over ranks of a host double (bench timing) */
int dvm_exchange_max_over_ranks(dvm_exchange* ex, double* value);
/* a dvm_allreduce_fn (include/dvmslam_hip.h): in-place all-reduce of n doubles, device (on `stream`) or host memory (staged through a device
 * scratch buffer of the exchange); op 0 sum, 1 max */
int dvm_exchange_allreduce(void* ctx, void* buf, int64_t n, int on_host, int op, void* stream);

#ifdef __cplusplus
}
#endif
#endif
