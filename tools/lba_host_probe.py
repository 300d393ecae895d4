"""K = 32 fast windows from pre-marshalled host arrays: median call time (the figure bench_legs.lba_batch quotes), for A/B of the host-side
switches (DVM_STAGE_THREADS, DVM_STAGE_PIECES, DVM_BA_BUILD_THREADS: read once per process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dvm_slam_amd import capi, synth
delta = float(np.sqrt(np.float32(5.991)))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
wins = []
for a in range(K):
    pr = synth.ba_problem(n_kf=30, n_pts=3000, k_obs=5, seed=0x1BA + a, radius=12.0)
    pr["fixed"][:10] = 1
    wins.append(dict(poses=pr["poses"], fixed=pr["fixed"], points=pr["points"], edges=capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"]),
                     intrinsics=pr["intrinsics"], huber_delta=delta, iterations=10))
batch = capi.BaWindowBatch(wins)
ts = []
for i in range(25):
    t0 = time.perf_counter(); batch.run(0, None, True, collect=False); ts.append(time.perf_counter() - t0)
ts = np.array(ts[5:]) * 1e3
its = sum(r["stats"]["iterations"] for r in batch.results())
print("K=%d median %.2f ms  min %.2f  -> %.1f k it/s  kernel %.2f ms  [%s]" % (K, np.median(ts), ts.min(), its / np.median(ts), batch.results()[0]["stats"]["kernel_us"] / 1e3,
      " ".join("%s=%s" % (k, os.environ[k]) for k in ("DVM_STAGE_THREADS", "DVM_STAGE_PIECES", "DVM_BA_BUILD_THREADS") if k in os.environ)))
