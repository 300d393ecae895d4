"""BA leg of bench.py: iterations/s of the 500-keyframe / 20 000-landmark global bundle adjustment
(BASELINE.json metric, second half) + its CPU baseline + a smoke check."""
from __future__ import annotations

import time

import numpy as np

# per single-trial iteration, dense reduced system n = 6*499 = 2994
FLOPS_DENSE_CHOLESKY = 2994 ** 3 / 3.0
FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X vendor FP64 matrix peak (CDNA4), also the FP64 vector peak


def _prewarm(seconds: float):
    """Keep the GPU busy so that it leaves its idle clock before the timed LM iterations (standalone runs only;
    inside bench.py the extract leg has just done that)."""
    import torch
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            a @ a
        torch.cuda.synchronize()


def run(device: int, iters: int = 10, cpu_seconds: float = 6.0, prewarm_s: float = 0.0):
    from dvm_slam_amd import capi, synth
    if prewarm_s > 0:
        _prewarm(prewarm_s)
    pr = synth.ba_problem()
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    delta = float(np.sqrt(5.991))
    ba = capi.BundleAdjuster(device)
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    ba.optimize(2)  # warm-up (kernel load)
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    t0 = time.perf_counter()
    st = ba.optimize(iters)
    dt = time.perf_counter() - t0
    poses_g, points_g = ba.result()
    ba.close()
    out = {
        "metric": "BA iterations/sec, 500 KF / 20k landmarks / 160k observations (outer LM iterations)",
        "value": st["iterations"] / dt, "unit": "iterations/s", "iterations": st["iterations"],
        "trials": st["total_trials"], "ms_per_iteration": dt / max(st["iterations"], 1) * 1e3,
        "ms_graph_build_excluded": st["ms_structure"], "chi2_initial": st["chi2_initial"], "chi2_final": st["chi2_final"],
        "dtype": "f64", "huber_delta": delta,
        "gpu_state": "hot (timed right after GPU work; an idle MI355X stays at its 584 MHz idle clock under this host-"
                     "synchronised LM loop: ~315 it/s cold vs ~1250 it/s hot)",
        "roofline": {"bound": "mfma", "kernel": "k_chol_diag/k_chol_trsm/k_chol_update: reduced-camera Cholesky on v_mfma_f64_16x16x4",
                     "achieved": FLOPS_DENSE_CHOLESKY * st["total_trials"] / dt / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": FLOPS_DENSE_CHOLESKY * st["total_trials"] / dt / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                     "note": "dense-equivalent rate: n^3/3 FLOP of a dense n=2994 factorisation per trial over the WHOLE "
                             "iteration time (edge pass, Schur, solve, update); the solver itself skips structurally zero "
                             "64x64 tiles (symbolic tile fill), so executed FLOPs are lower -- the solve is a dependency chain of "
                             "elimination-tree levels (9 at this size after nested dissection, 8 launched; 47 tile columns before), "
                             "each level = 3 launches, not MFMA-throughput bound (DESIGN.md section 3)"},
    }
    if cpu_seconds > 0:
        out["cpu_baseline"], (poses_c, points_c, st_c) = cpu_baseline(pr, delta, cpu_seconds, iters)
        # the same problem has just been solved on both sides: the bench line carries the parity of THIS run
        par = {"trials_equal": bool(st["trials"] == st_c["trials"]),
               "chi2_final_rel": abs(st["chi2_final"] - st_c["chi2"][-1]) / st_c["chi2"][-1],
               "max_abs_pose": float(np.abs(poses_g - poses_c).max()), "max_abs_landmark": float(np.abs(points_g - points_c).max())}
        out["parity_vs_cpu"] = par
        if not (par["trials_equal"] and par["chi2_final_rel"] <= 1e-9 and par["max_abs_pose"] < 1e-6 and par["max_abs_landmark"] < 1e-6):
            raise RuntimeError(f"BA leg: GPU result differs from the CPU oracle on the benchmarked problem: {par}")
    return out


def cpu_baseline(pr, delta, budget_s, iters):
    """The oracle on the same problem, `iters` LM iterations per run, repeated for about budget_s seconds.  Returns the
    baseline record and the last run's (poses, points, stats) for the parity check."""
    import os
    import subprocess
    from oracle import pyoracle as po
    root = os.path.dirname(os.path.abspath(__file__))
    libpath = "/tmp/liboracle_native.so"
    try:
        subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s", f"OUT={libpath}", "ARCHFLAGS=-march=native"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        libpath = None
    e = po.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    done, runs = 0, 0
    t0 = time.perf_counter()
    while True:
        p, x, st, _ = po.ba_optimize(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta, iters, libpath=libpath)
        done += st["iterations"]; runs += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or runs >= 50:
            break
    rec = {"value": done / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
           "sample": f"{done} LM iterations ({runs} runs of {iters}) of the same 500 KF / 20k landmark problem, oracle (envelope sparse "
                     f"Cholesky), 1 thread, {dt:.1f} s"}
    return rec, (p, x, st)


def smoke():
    """Tiny BA on cuda:0 checked against the CPU oracle (poses within 1e-6, same LM trial sequence)."""
    from dvm_slam_amd import capi, synth
    from oracle import pyoracle as po  # checker only
    pr = synth.ba_problem(n_kf=12, n_pts=300, seed=3)
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    delta = float(np.sqrt(5.991))
    ba = capi.BundleAdjuster()
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    st = ba.optimize(10)
    pg, ptg = ba.result()
    ba.close()
    po_, pto, so, _ = po.ba_optimize(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta, 10)
    assert st["trials"] == so["trials"], "LM trial sequence differs from oracle"
    assert np.abs(pg - po_).max() < 1e-6 and np.abs(ptg - pto).max() < 1e-6, "BA result differs from oracle"
