"""One rank of libdvmslam_rccl.so (include/dvmslam_rccl.h) on the GPU box, for tests/test_gpu_rccl_native.py: a communicator made by the
library's own helpers (ncclGetUniqueId / ncclCommInitRank, world 1: the box has one GPU), the exchange collectives on device buffers,
and the landmark-sharded global BA with the NATIVE all-reduce as its callback -- dvm_ba_set_allreduce(h, dvm_exchange_allreduce, ex, ...):
no Python, no torch.distributed on the solver's critical path."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvm_slam_amd import capi, synth  # noqa: E402


def main():
    out = sys.argv[1]
    torch.cuda.set_device(0)
    R = C.CDLL(os.path.join(ROOT, "dvm_slam_amd", "lib", "libdvmslam_rccl.so"))
    R.dvm_exchange_last_error.restype = C.c_char_p

    def ok(rc):
        assert rc == 0, R.dvm_exchange_last_error().decode()
    uid = (C.c_uint8 * 128)()
    ok(R.dvm_exchange_unique_id(uid))
    comm = C.c_void_p()
    ok(R.dvm_exchange_comm_init(uid, 0, 1, 0, C.byref(comm)))
    stream = torch.cuda.Stream()
    ex = C.c_void_p()
    ok(R.dvm_exchange_create(comm, C.c_void_p(stream.cuda_stream), C.byref(ex)))
    rec = {"rank": R.dvm_exchange_rank(ex), "world": R.dvm_exchange_world(ex)}
    with torch.cuda.stream(stream):
        blk = torch.arange(1000, dtype=torch.int64, device="cuda").to(torch.uint8)
        recv = torch.zeros(1000, dtype=torch.uint8, device="cuda")
        ok(R.dvm_exchange_allgather_blocks(ex, C.c_void_p(blk.data_ptr()), C.c_int64(1000), C.c_void_p(recv.data_ptr())))
        stream.synchronize()
        rec["blocks_ok"] = bool(torch.equal(recv, blk))
        slot = torch.full((4096,), 7, dtype=torch.uint8, device="cuda")
        sizes = (C.c_int64 * 1)()
        ok(R.dvm_exchange_allgather_varlen(ex, C.c_void_p(blk.data_ptr()), C.c_int64(1000), C.c_void_p(slot.data_ptr()), C.c_int64(4096), sizes))
        stream.synchronize()
        rec["varlen_ok"] = sizes[0] == 1000 and bool(torch.equal(slot[:1000], blk)) and int(slot[1000:].sum().item()) == 0
        rec["varlen_too_small"] = R.dvm_exchange_allgather_varlen(ex, C.c_void_p(blk.data_ptr()), C.c_int64(1000), C.c_void_p(slot.data_ptr()), C.c_int64(512), sizes)
        s = torch.tensor([1.25, 0, 0, 0, 1, 3, 4, 5], dtype=torch.float64, device="cuda")
        s2 = s.clone()
        ok(R.dvm_exchange_broadcast_sim3(ex, C.c_void_p(s2.data_ptr()), 0))
        stream.synchronize()
        rec["sim3_ok"] = bool(torch.equal(s, s2))
        v = C.c_double(1.5)
        ok(R.dvm_exchange_max_over_ranks(ex, C.byref(v)))
        rec["max_over_ranks"] = v.value
    # the sharded global BA with the native callback
    pr = synth.ba_problem(n_kf=60, n_pts=2500, seed=60 * 31 + 2500)
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    delta = float(np.sqrt(5.991))
    ba = capi.BundleAdjuster()
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    s1 = ba.optimize(8)
    p1, x1 = ba.result()
    ba.close()
    sb = capi.BundleAdjuster()
    sb.set_problem_sharded(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta, 0, 1)
    n = sb.allreduce_doubles()
    buf = torch.zeros(n, dtype=torch.float64, device="cuda")
    f = sb.L.dvm_ba_set_allreduce
    f.restype = C.c_int32; f.argtypes = None
    fn = C.cast(R.dvm_exchange_allreduce, C.c_void_p)          # the C function itself: libdvmslam_hip.so calls libdvmslam_rccl.so
    capi.check(f(sb.h, fn, ex, C.c_void_p(buf.data_ptr()), C.c_int64(n)))
    s2_ = sb.optimize(8)
    p2, x2 = sb.result()
    sb.close()
    rec["ba_trials_equal"] = s1["trials"] == s2_["trials"]
    rec["ba_bits_equal"] = bool(np.array_equal(p1, p2) and np.array_equal(x1, x2))
    rec["ba_allreduce_doubles"] = int(n)
    R.dvm_exchange_destroy(ex)
    R.dvm_exchange_comm_destroy(comm)
    # the same through dvm_slam_amd.sharded_ba.ShardedBundleAdjuster(native=True): what bench.py's sharded leg runs under DVM_SHARDED_NATIVE=1
    from dvm_slam_amd import sharded_ba
    sn = sharded_ba.ShardedBundleAdjuster(0, native=True)
    sn.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    s3 = sn.optimize(8)
    p3, x3 = sn.result()
    sn.close()
    rec["class_native_bits_equal"] = bool(s1["trials"] == s3["trials"] and np.array_equal(p1, p3) and np.array_equal(x1, x3))
    rec["class_python_calls"] = int(sn.calls)          # the Python callback must not have run
    json.dump(rec, open(out, "w"))


if __name__ == "__main__":
    main()
