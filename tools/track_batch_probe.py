"""tools/track_batch_probe.py [K] -- dvmh_track_with_motion_model_batch in a loop on the bench stream (for rocprofv3 --kernel-trace --stats)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dvm_slam_amd import capi, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
frames = synth.frame_stream(9)
ext1 = capi.OrbExtractor(max_batch=1)
ext = capi.OrbExtractor(max_batch=K)
tab = ext.tables(); scale, inv_s2 = tab["scale"], tab["inv_sigma2"]
trk = capi.TrackerBatch(ext, K)
Kc = np.array([500.0, 500.0, 320.0, 240.0], np.float32); B = np.array([0, 640, 0, 480], np.float32)
rng = np.random.default_rng(9)
T = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
lasts, imgs = [], []
for a in range(K):
    t = 1 + a % 8
    n0, k0, d0, _ = ext1.extract(frames[t - 1])
    z = rng.uniform(3, 9, n0).astype(np.float32)
    mps = np.zeros(n0, capi.MAP_POINT_DTYPE)
    mps["pos"][:, 0] = (k0["x"] - Kc[2]) / Kc[0] * z; mps["pos"][:, 1] = (k0["y"] - Kc[3]) / Kc[1] * z; mps["pos"][:, 2] = z
    mps["desc"] = d0; mps["n_obs"] = 1
    lasts.append((k0.copy(), np.arange(n0, dtype=np.int32), None, mps)); imgs.append(frames[t])
ins, keep = trk.prepare([T] * K, lasts)
imgs = np.stack(imgs)
ts = []
for i in range(40):
    t0 = time.perf_counter(); r = trk.track(imgs, ins, Kc, B, scale, inv_s2); ts.append(time.perf_counter() - t0)
ts = np.sort(ts[5:]) * 1e3
print(f"K={K}: {ts[len(ts)//2]:.3f} ms per tick = {K / ts[len(ts)//2] * 1e3:.0f} frames/s; matches {r[0]['nmatches_search']}")
