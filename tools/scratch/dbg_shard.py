import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch.distributed as dist
from dvm_slam_amd import capi, sharded_ba, synth
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")
rank = dist.get_rank()
pr = synth.ba_problem(n_kf=40, n_pts=1500, seed=40*31+1500)
e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
sb = sharded_ba.ShardedBundleAdjuster(0)
orig = sb._allreduce
def dbg(buf, n, on_host, op, stream):
    import ctypes as C
    if on_host:
        a = np.ctypeslib.as_array((C.c_double * n).from_address(buf)).copy()
    r = orig(buf, n, on_host, op, stream)
    if on_host:
        b = np.ctypeslib.as_array((C.c_double * n).from_address(buf)).copy()
        print(rank, "host", op, a, "->", b, flush=True)
    else:
        print(rank, "dev", n, float(sb.buf[:n].abs().sum()), flush=True)
    return r
sb._allreduce = dbg
sb.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
st = sb.optimize(2)
print(rank, st["trials"], st["chi2_initial"], st["chi2"], flush=True)
if rank == 0:
    ba = capi.BundleAdjuster(); ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
    s1 = ba.optimize(2); print("single", s1["trials"], s1["chi2_initial"], s1["chi2"], s1["lam"], flush=True)
dist.barrier()
