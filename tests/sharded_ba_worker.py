"""One rank of tests/test_gpu_sharded_ba.py (launched by torch.distributed.run, gloo, all ranks on GPU 0)."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvm_slam_amd import capi, sharded_ba, synth  # noqa: E402


def main():
    n_kf, n_pts, delta, iters, out = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    pr = synth.ba_problem(n_kf=n_kf, n_pts=n_pts, seed=n_kf * 31 + n_pts)
    if len(sys.argv) > 6 and sys.argv[6] == "unobserved":       # three landmarks nobody observes (one per residue class mod 3): "not a vertex", they
        pr["points"] = np.concatenate([pr["points"], [[1.5, -2.5, 3.5], [4.0, 5.0, 6.0], [-7.0, 8.0, 9.0]]])   # must come back exactly as they went in
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    sb = sharded_ba.ShardedBundleAdjuster(0)
    sb.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    st = sb.optimize(iters)
    p, x = sb.result()
    np.savez(f"{out}.rank{rank}.npz", poses=p, points=x, trials=np.array(st["trials"]), chi2=np.array(st["chi2"]), lam=np.array(st["lam"]),
             chi2_initial=st["chi2_initial"], calls=sb.calls, bytes=sb.bytes_reduced)
    sb.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
