#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (run in the build container).

The reference holds no golden vector for this path and cannot be built or imported here (C++ needing
OpenCV/Eigen/Boost/ROS 2), so these fixtures are produced by the oracle -- which is itself pinned
against independent numpy restatements (tests/test_oracle_*.py).  They freeze today's behaviour:
both the oracle and the HIP path must keep reproducing them bit for bit (BA: within 1e-9).
Fixtures are DATA only (inputs + expected outputs).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dvm_slam_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def orb_case(name, img, params):
    o = po.OrbOracle(*params)
    n, k, d, mono = o.extract(img)
    lv = [o.level_dims(l) for l in range(params[2])]
    cand_counts = [len(o.candidates(l)[0]) for l in range(params[2])]
    np.savez_compressed(os.path.join(HERE, name), image=img, params=np.array(params, np.float64), n=n, mono=mono,
                        kp_x=k["x"], kp_y=k["y"], kp_size=k["size"], kp_angle=k["angle"], kp_response=k["response"],
                        kp_octave=k["octave"], desc=d, level_dims=np.array(lv, np.int32), cand_counts=np.array(cand_counts, np.int32))
    print(name, n, mono, cand_counts)


def main():
    orb_case("orb_160x120.npz", synth.small_image(1, 120, 160), (300, 1.2, 4, 20, 7))
    orb_case("orb_320x240.npz", synth.frame_stream(1, start=40)[0][100:340, 200:520].copy(), (500, 1.2, 8, 20, 7))
    # matching: two frames' keypoints -> windowed best/second best
    fr = synth.frame_stream(2, start=7)
    o = po.OrbOracle(500, 1.2, 8, 20, 7)
    _, k0, d0, _ = o.extract(fr[0][:240, :320].copy())
    _, k1, d1, _ = o.extract(fr[1][:240, :320].copy())
    sc = o.tables()["scale"]
    g = po.Grid(k1, 0.0, 320.0, 0.0, 240.0)
    qr = (np.float32(15) * sc[k0["octave"]]).astype(np.float32)
    m = g.match_window(d1, d0, k0["x"], k0["y"], qr, k0["octave"] - 1, k0["octave"] + 1)
    np.savez_compressed(os.path.join(HERE, "match_320x240.npz"), k0=k0, d0=d0, k1=k1, d1=d1, qr=qr, **m)
    print("match", len(k0), len(k1), int((m["best_dist"] <= 100).sum()))
    # bundle adjustment: small window
    pr = synth.ba_problem(n_kf=10, n_pts=200, seed=99)
    e = po.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    poses, pts, st, chi = po.ba_optimize(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], np.sqrt(5.991), 10)
    np.savez_compressed(os.path.join(HERE, "ba_10kf_200pt.npz"), poses0=pr["poses"], fixed=pr["fixed"], points0=pr["points"],
                        edges=e, intrinsics=pr["intrinsics"], delta=np.sqrt(5.991), iters=10, poses=poses, points=pts,
                        trials=np.array(st["trials"]), chi2=np.array(st["chi2"]), edge_chi2=chi)
    print("ba", st["trials"], st["chi2"][-1])


if __name__ == "__main__":
    main()
