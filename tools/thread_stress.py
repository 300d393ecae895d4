"""tools/thread_stress.py [seconds] -- the five caller threads of tests/test_gpu_threads.py::test_concurrent_callers (two default extractors, one
8 000-feature extractor, the tile solver, PoseOptimization) looping for a while: calls made and results that differ from the reference bits, by kind."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dvm_slam_amd import capi, synth
from oracle import pyoracle as po
frames = synth.frame_stream(6)
orc = po.OrbOracle()
ref_ext = [orc.extract(f) for f in frames]
pr = synth.ba_problem(n_kf=14, n_pts=400, seed=5)
e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
delta = float(np.sqrt(5.991))
def run_ba():
    ba = capi.BundleAdjuster(); ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    st = ba.optimize(6); p, x = ba.result(); ba.close(); return st["trials"], p, x
ba_ref = run_ba()
rng = np.random.default_rng(0)
Xw = rng.uniform(-3, 3, (300, 3)) + [0, 0, 8]
K = np.array([500.0, 500.0, 320.0, 240.0])
obs = np.stack([K[0] * Xw[:, 0] / Xw[:, 2] + K[2], K[1] * Xw[:, 1] / Xw[:, 2] + K[3]], 1) + rng.normal(0, 0.5, (300, 2))
pose0 = np.array([0.05, -0.03, 0.1, 0.0, 0.0, 0.0, 1.0])
po_ref = capi.pose_optimize(pose0[None], Xw[None], obs[None], np.ones((1, 300)), [300], K)
big_img = synth.small_image(77, 600, 800)
big_ref = po.OrbOracle(8000, 1.2, 8, 12, 5).extract(big_img, cap=4 * 8000 + 256)
T = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
t_end = time.time() + T
bad = {"extract": 0, "big": 0, "ba": 0, "pose": 0, "exc": []}
cnt = {"extract": 0, "big": 0, "ba": 0, "pose": 0}
def guard(fn):
    def w():
        try: fn()
        except Exception as ex: bad["exc"].append(repr(ex)[:300])
    return w
def t_extract(k):
    def f():
        ext = capi.OrbExtractor(max_batch=1); it = 0
        while time.time() < t_end:
            i = (it + k) % len(frames); it += 1
            n, kp, d, m = ext.extract(frames[i]); n_o, k_o, d_o, m_o = ref_ext[i]
            cnt["extract"] += 1
            if not ((n, m) == (n_o, m_o) and np.array_equal(d, d_o) and np.array_equal(kp["x"], k_o["x"])): bad["extract"] += 1
        ext.close()
    return f
def t_big():
    while time.time() < t_end:
        ext = capi.OrbExtractor(8000, 1.2, 8, 12, 5, max_batch=1)
        for _ in range(3):
            n, kp, d, m = ext.extract(big_img); cnt["big"] += 1
            if not ((n, m) == (big_ref[0], big_ref[3]) and np.array_equal(d, big_ref[2])): bad["big"] += 1
        ext.close()
def t_ba():
    while time.time() < t_end:
        tr, p, x = run_ba(); cnt["ba"] += 1
        if not (tr == ba_ref[0] and np.array_equal(p, ba_ref[1]) and np.array_equal(x, ba_ref[2])): bad["ba"] += 1
def t_pose():
    while time.time() < t_end:
        p, o, n = capi.pose_optimize(pose0[None], Xw[None], obs[None], np.ones((1, 300)), [300], K); cnt["pose"] += 1
        if not (np.array_equal(p, po_ref[0]) and np.array_equal(o, po_ref[1]) and n[0] == po_ref[2][0]): bad["pose"] += 1
th = [threading.Thread(target=guard(f)) for f in (t_extract(0), t_extract(1), t_big, t_ba, t_pose)]
for t in th: t.start()
for t in th: t.join()
print("counts", cnt, "bad", bad)
