#!/usr/bin/env python3
"""tools/pose_bench.py -- Optimizer::PoseOptimization throughput (dvm_pose_optimize): a batch of B frames with S matches each,
host arrays in, poses / outlier flags out (the call Tracking makes after SearchByProjection).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from dvm_slam_amd import capi
    from test_gpu_ba import _pose_case
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    cases = [_pose_case(100 + i, n_pts=S, out_frac=0.1) for i in range(8)]
    poses = np.stack([cases[i % 8][0] for i in range(B)])
    Xw = np.stack([cases[i % 8][1] for i in range(B)]); obs = np.stack([cases[i % 8][2] for i in range(B)])
    w = np.stack([cases[i % 8][3] for i in range(B)]); n = np.full(B, S, np.int32)
    capi.pose_optimize(poses, Xw, obs, w, n, cases[0][4])
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        capi.pose_optimize(poses, Xw, obs, w, n, cases[0][4])
    dt = (time.perf_counter() - t0) / reps
    one = min(_time_one(capi, cases[0]) for _ in range(5))
    print(json.dumps({"metric": "PoseOptimization (4 x optimize(10), host arrays in / out)", "batch": B, "matches": S,
                      "frames_per_s": B / dt, "ms_per_batch": dt * 1e3, "ms_single_frame": one * 1e3}))


def _time_one(capi, c):
    t0 = time.perf_counter()
    capi.pose_optimize(c[0][None], c[1][None], c[2][None], c[3][None], np.array([len(c[1])], np.int32), c[4])
    return time.perf_counter() - t0


if __name__ == "__main__":
    main()
