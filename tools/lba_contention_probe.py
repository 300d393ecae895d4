"""tools/lba_contention_probe.py -- dvm_ba_optimize_windows_fast for 32 windows (256 workgroups that must be co-resident) WHILE another thread keeps
the chip busy with 256-frame extraction batches: the cluster barriers may time out; the call must still return the right results (the G = 1 repeat)
and must never hang."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dvm_slam_amd import capi, synth
delta = float(np.sqrt(np.float32(5.991)))
wins = []
for a in range(32):
    pr = synth.ba_problem(n_kf=30, n_pts=3000, k_obs=5, seed=0x1BA + a, radius=12.0)
    pr["fixed"][:10] = 1
    wins.append(dict(poses=pr["poses"], fixed=pr["fixed"], points=pr["points"], edges=capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"]),
                     intrinsics=pr["intrinsics"], huber_delta=delta, iterations=10))
batch = capi.BaWindowBatch(wins)
ref = [dict(poses=r["poses"].copy(), points=r["points"].copy()) for r in batch.run(fast=True)]
frames = synth.frame_stream(64)
ext = capi.OrbExtractor(max_batch=256)
big = np.concatenate([frames] * 4)
stop = False
def hog():
    while not stop:
        ext.extract_batch_host(big); ext.sync()
th = threading.Thread(target=hog); th.start()
time.sleep(0.3)
ts = []
ok = True
for i in range(12):
    t0 = time.perf_counter(); res = batch.run(fast=True); ts.append(time.perf_counter() - t0)
    ok = ok and all(np.array_equal(r["poses"], q["poses"]) and np.array_equal(r["points"], q["points"]) for r, q in zip(res, ref))
stop = True; th.join()
print("under contention: identical", ok, "ms per call", [round(t * 1e3, 1) for t in ts])
