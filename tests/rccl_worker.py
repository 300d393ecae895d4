"""One RCCL rank (backend "nccl", world_size 1) for tests/test_gpu_rccl.py: the test box has one GPU, and RCCL refuses two ranks
on the same device, so this is the largest RCCL job it can hold.  Everything the N > 1 job does over RCCL runs here with its
single member: communicator creation on the device, the exchange collectives on device tensors, the sharded bundle
adjustment's all-reduce on the solver's own HIP stream."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvm_slam_amd import capi, exchange, sharded_ba, synth  # noqa: E402


def main():
    out = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    exchange.SHORTCUT_SINGLE_RANK = False
    rec = {"backend": dist.get_backend()}
    # bench.py's timing reduction and the ragged DVMW gather, on device tensors
    rec["max_over_ranks"] = exchange.max_over_ranks(1.5, device="cuda")
    blk = torch.arange(1000, dtype=torch.int64, device="cuda").to(torch.uint8)
    got = exchange.all_gather_varlen(blk)
    rec["varlen_ok"] = len(got) == 1 and got[0].is_cuda and bool(torch.equal(got[0], blk))
    fixed = exchange.all_gather_blocks(blk)
    rec["blocks_ok"] = len(fixed) == 1 and bool(torch.equal(fixed[0], blk))
    s = torch.tensor([1.25, 0, 0, 0, 1, 3, 4, 5], dtype=torch.float64, device="cuda")
    rec["sim3_ok"] = bool(torch.equal(exchange.broadcast_sim3(s.clone(), 0), s))
    # sharded global BA over RCCL: the tile all-reduce is enqueued on the solver's stream (torch.cuda.ExternalStream)
    pr = synth.ba_problem(n_kf=60, n_pts=2500, seed=60 * 31 + 2500)
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    delta = float(np.sqrt(5.991))
    ba = capi.BundleAdjuster()
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    s1 = ba.optimize(8)
    p1, x1 = ba.result()
    ba.close()
    sb = sharded_ba.ShardedBundleAdjuster(0)
    assert sb.on_gpu
    sb.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    s2 = sb.optimize(8)
    p2, x2 = sb.result()
    rec["ba_trials_equal"] = s1["trials"] == s2["trials"]
    rec["ba_bits_equal"] = bool(np.array_equal(p1, p2) and np.array_equal(x1, x2))
    rec["ba_calls"] = sb.calls
    rec["ba_bytes"] = sb.bytes_reduced
    rec["ba_expected_calls"] = 2 * sum(s1["trials"]) + 1 + 2 + 1
    sb.close()
    dist.barrier()
    dist.destroy_process_group()
    json.dump(rec, open(out, "w"))


if __name__ == "__main__":
    main()
