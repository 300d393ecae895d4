"""tools/lba_pool_cpp.py [K ...] -- dvm_ba_pool_optimize from K NATIVE threads (tools/lba_pool_threads.cpp), 30-keyframe windows of the LBA legs:
LM iterations/s over all agents, ms per round (= every agent one call).  The Python-thread form of bench_legs.lba_batch spends as long
re-entering the interpreter between calls as the launch takes; this is the figure of a C++ host.  JSON on stdout."""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dvm_slam_amd import capi, synth
so = os.path.join(ROOT, "tools", "bin", "liblba_pool_threads.so")
src = os.path.join(ROOT, "tools", "lba_pool_threads.cpp")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + os.path.join(ROOT, "include"), src, "-o", so,
                           "-L" + os.path.join(ROOT, "dvm_slam_amd", "lib"), "-ldvmslam_hip", "-Wl,-rpath," + os.path.join(ROOT, "dvm_slam_amd", "lib")])
H = C.CDLL(so)
H.lba_pool_threads.restype = C.c_double
delta = float(np.sqrt(np.float32(5.991)))
Ks = [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64]
wins = []
for a in range(max(Ks)):
    pr = synth.ba_problem(n_kf=30, n_pts=3000, k_obs=5, seed=0x1BA + a, radius=12.0)
    pr["fixed"][:10] = 1
    wins.append(dict(poses=pr["poses"], fixed=pr["fixed"], points=pr["points"], edges=capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"]),
                     intrinsics=pr["intrinsics"], huber_delta=delta, iterations=10))
solo = capi.ba_optimize_windows(wins[:2], fast=True)
out = {"call": "dvm_ba_pool_optimize from K native threads, one window (30 KF / 20 free / 3 000 landmarks) per thread and call", "by_K": {}}
for K in Ks:
    batch = capi.BaWindowBatch(wins[:K])
    pool = capi.BaPool(0, max_batch=min(K, 32))
    calls, per, rc = 24, C.c_int32(0), C.c_int32(0)
    H.lba_pool_threads(pool.p, batch.wins, batch.stats, C.c_int32(K), C.c_int32(6), C.byref(per), C.byref(rc))     # warm-up
    ms = H.lba_pool_threads(pool.p, batch.wins, batch.stats, C.c_int32(K), C.c_int32(calls), C.byref(per), C.byref(rc))
    assert rc.value == 0, rc.value
    res = batch.results()
    its = sum(r["stats"]["iterations"] for r in res) * calls
    same = all(np.array_equal(res[k]["poses"], solo[k]["poses"]) and np.array_equal(res[k]["edge_chi2"], solo[k]["edge_chi2"]) for k in range(min(K, 2)))
    out["by_K"][str(K)] = {"value": its / ms * 1e3, "ms_per_round": ms / calls, "mean_windows_per_launch": per.value, "same_bits_as_a_solo_call": bool(same)}
    pool.close()
print(json.dumps(out))
