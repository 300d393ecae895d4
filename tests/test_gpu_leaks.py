"""Handles come and go in a SLAM process (a BA handle per LocalBundleAdjustment call in the shims, extractor / grid handles per agent, the pools'
idle lists): device memory in use must come back to where it started.  hipMemGetInfo before and after many create / use / destroy cycles."""
import numpy as np
import pytest
import torch

from dvm_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def _used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return total - free


def _rss():
    import psutil
    return psutil.Process().memory_info().rss


def test_ba_handles_release_their_memory():
    delta = float(np.sqrt(5.991))
    probs = [synth.ba_problem(n_kf=6, n_pts=40, k_obs=4, seed=3, radius=20.0), synth.ba_problem(n_kf=30, n_pts=900, k_obs=8, seed=3, radius=50.0),
             synth.ba_problem(n_kf=120, n_pts=4000, seed=13)]
    probs.append(synth.ba_problem(n_kf=150, n_pts=5000, seed=9, laps=2, long_range_frac=0.01))
    edges = [capi.make_edges(p["edge_pose"], p["edge_point"], p["obs"], p["inv_sigma2"]) for p in probs]

    def cycle():
        for p, e in zip(probs, edges):
            ba = capi.BundleAdjuster()
            ba.set_problem(p["poses"], p["fixed"], p["points"], e, p["intrinsics"], delta)
            ba.optimize(2)
            ba.result()
            ba.set_problem(p["poses"], p["fixed"], p["points"], e, p["intrinsics"], 0.0)     # a second problem on the same handle
            ba.optimize(1)
            ba.close()
    cycle()                       # first use: the runtime's own pools, code objects, pinned arenas
    cycle()
    base, rss0 = _used(), _rss()
    for _ in range(25):
        cycle()
    grown = _used() - base
    assert grown <= 8 << 20, f"{grown / 2**20:.1f} MiB of device memory not returned after 100 BA handles"
    assert _rss() - rss0 <= 96 << 20, f"host memory grew by {(_rss() - rss0) / 2**20:.1f} MiB over 100 BA handles"


def test_extractor_and_grid_handles_release_their_memory():
    frames = synth.frame_stream(4)
    H, W = frames.shape[1:]

    def cycle():
        ext = capi.OrbExtractor(max_batch=4)
        grid = capi.FrameGrid(capacity=2048, slots=4)
        out = ext.extract(frames[0])
        kps, desc = out[1], out[2]
        grid.build(kps, desc, bounds=(0.0, float(W), 0.0, float(H)))
        ext.close()
        grid.close()
    cycle(); cycle()
    base, rss0 = _used(), _rss()
    for _ in range(40):
        cycle()
    grown = _used() - base
    assert grown <= 8 << 20, f"{grown / 2**20:.1f} MiB of device memory not returned after 40 extractor / grid handles"
    assert _rss() - rss0 <= 96 << 20, f"host memory grew by {(_rss() - rss0) / 2**20:.1f} MiB over 40 extractor / grid handles"
