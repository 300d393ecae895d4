"""BA leg of bench.py: iterations/s of the 500-keyframe / 20 000-landmark global bundle adjustment
(BASELINE.json metric, second half) + its CPU baseline + a smoke check."""
from __future__ import annotations

import time

import numpy as np

# per single-trial iteration, dense reduced system n = 6*499 = 2994 (the dense-equivalent figure, kept as a note)
FLOPS_DENSE_CHOLESKY = 2994 ** 3 / 3.0
FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X vendor FP64 matrix peak (CDNA4), also the FP64 vector peak
HBM_PEAK_GBS = 8000.0
T3 = 64 ** 3


def executed_flops_per_trial(info):
    """FLOPs the tile solver actually executes per LM trial, from the symbolic factorisation (dvm_ba_schedule_info): per tile
    column a 64x64 Cholesky + triangular inverse (2 * 64^3 / 3), per strip a 64^3 GEMM (trsm by the inverse), per
    (target, contributor) product a 64^3 GEMM (three quadrants of four on diagonal targets), back substitution 2 * 64^2 per
    strip and column."""
    gemm = 2.0 * T3
    return (info["columns"] * gemm / 3.0 + info["strips"] * gemm + (info["products"] - 0.25 * info["products_on_diagonal_targets"]) * gemm
            + (info["strips"] + info["columns"]) * 2.0 * 64 * 64)


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


def run(device: int, iters: int = 10, cpu_seconds: float = 6.0, prewarm_s: float = 0.0, repeats: int = 470):
    """The timed region is `repeats` optimisations of `iters` LM iterations each of the same problem (set_problem, i.e. the
    reference's graph construction, outside it): 470 x 9 iterations = ~1.07 s of dvm_ba_optimize, long enough for clock sampling to
    see it (each run is preceded by ~8 ms of set_problem on the host, during which the GPU idles)."""
    from dvm_slam_amd import capi, synth
    if prewarm_s > 0:
        _prewarm(prewarm_s)
    pr = synth.ba_problem()
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    delta = float(np.sqrt(5.991))
    ba = capi.BundleAdjuster(device)
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    ba.optimize(2)  # warm-up (kernel load)
    dt, dt_py, its, trials = 0.0, 0.0, 0, 0
    for _ in range(max(1, repeats)):
        ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
        t0 = time.perf_counter()
        st = ba.optimize(iters)
        dt_py += time.perf_counter() - t0
        dt += st["ms_optimize"] * 1e-3          # wall time of dvm_ba_optimize measured inside the C ABI (the ctypes call and the
        its += st["iterations"]; trials += st["total_trials"]   # Python dict of the statistics add ~1 % on top: value_python_wall)
    poses_g, points_g = ba.result()
    info = ba.schedule_info()
    # also the GBA form, bRobust = false (LoopClosing.cc:2282) -- same problem, no Huber kernel
    dt0, its0, tr0 = 0.0, 0, 0
    for _ in range(max(1, repeats // 16)):
        ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], 0.0)
        st0 = ba.optimize(iters)
        dt0 += st0["ms_optimize"] * 1e-3
        its0 += st0["iterations"]; tr0 += st0["total_trials"]
    huber_off = {"value": its0 / dt0, "unit": "iterations/s", "iterations": its0, "trials": tr0, "chi2_final": st0["chi2_final"]}
    # second pass with HIP events around the phases of every trial (not part of `value`)
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    ba.profile(1)
    ba.optimize(iters)
    prof = ba.profile(0)
    ba.close()
    fl = executed_flops_per_trial(info)
    t_solve = prof["ms_cholesky_solve"] / max(prof["trials"], 1) * 1e-3
    # HBM-side traffic per launch from the committed PMC passes (tools/run_profiles_r03.sh: --pmc FETCH_SIZE / WRITE_SIZE, read side
    # calibrated x2 on known byte counts): the solve's kernels per trial, and the streaming kernels of a trial
    traffic, pmc_k, pmc_src = None, {}, None
    try:
        import json
        import os
        import sys
        root = os.path.dirname(os.path.abspath(__file__))
        for cand in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):
            try:
                fold = json.load(open(os.path.join(root, "profiles", cand)))
                break
            except Exception:   # noqa: BLE001
                fold = None
        pmc_k = fold["kernels"]
        sys.path.insert(0, os.path.join(root, "tools"))
        from kernels_sha import kernels_sha
        sha_now = kernels_sha()
        pmc_src = {"file": f"profiles/{cand}", "kernels_sha_of_fold": fold.get("kernels_sha"), "kernels_sha_running": sha_now,
                   "stale": None if not fold.get("kernels_sha") else bool(fold["kernels_sha"] != sha_now)}
        # the solve = whichever k_chol_* kernels the schedule of THIS build launches: their HBM bytes per LM trial (per-launch bytes x the
        # dispatches of the profiled run / its trials); folds older than round 4 carry no per-trial figure
        solve = {k: v["hbm_bytes_per_ba_trial"] for k, v in pmc_k.items() if k.startswith("dvm::k_chol_") and "hbm_bytes_per_ba_trial" in v}
        traffic = sum(solve.values()) if solve else None
    except Exception:   # noqa: BLE001
        traffic = None
    nblk_pairs = None
    E, L, nfree = info["edges"], len(pr["points"]), info["free_cameras"]
    # algorithmic bytes of the memory-bound phase of a trial / measured event time: landmark back
    # substitution + state update, then the trial state's evaluation WITH Jacobians and the Hpp / Hll accumulation (the
    # speculative linearisation an accepted trial hands to the next iteration)
    hbm = {
        "landmarks_update_linearise": {"bytes": E * (18 * 8 + 64) + L * 9 * 8 * 2 + E * (256 + 192),
                                       "ms": prof["ms_update_chi2"] / max(prof["trials"], 1)},
    }
    for v in hbm.values():
        v["GBps"] = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else None
        v["frac_of_8TBps"] = v["GBps"] / HBM_PEAK_GBS if v["GBps"] else None
    if pmc_k:
        try:
            hbm["landmarks_update_linearise"]["traffic"] = sum(pmc_k["dvm::" + k]["hbm_bytes_per_launch"] for k in ("k_point_backsub", "k_edge_eval", "k_accum"))
            # k_schur as a roofline object of its own: algorithmic bytes = every Hpl block W (144 B) read once + the landmark's Dinv (72 B per
            # landmark) + the non-zero 6x6 blocks written once ("re-reads Hpl 144 B x E = 23 MB")
            sch_bytes = E * 144 + L * 72 + info["nz_blocks"] * 288
            sch_ms = prof["ms_schur"] / max(prof["trials"], 1)
            hbm["schur"] = {"bound": "hbm", "kernel": "k_schur", "bytes": sch_bytes, "ms": sch_ms, "achieved": sch_bytes / (sch_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": sch_bytes / (sch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_k["dvm::k_schur"]["hbm_bytes_per_launch"],
                            "traffic_over_algorithmic": pmc_k["dvm::k_schur"]["hbm_bytes_per_launch"] / sch_bytes,
                            "traffic_GBps": pmc_k["dvm::k_schur"]["hbm_bytes_per_launch"] / (sch_ms * 1e-3) / 1e9,
                            "note": "W rows gathered per (edge, edge) pair: the counter traffic above the algorithmic bytes is re-reads of W"}
            hbm["traffic_source"] = pmc_src
        except Exception:   # noqa: BLE001
            pass
    out = {
        "metric": "BA iterations/sec, 500 KF / 20k landmarks / 160k observations (outer LM iterations)",
        "value": its / dt, "unit": "iterations/s", "iterations": its, "trials": trials, "runs": max(1, repeats),
        "ms_per_iteration": dt / max(its, 1) * 1e3, "timed_seconds": dt, "value_python_wall": its / dt_py,
        "speculation": {"trials_enqueued_on_the_device_decision": st["spec_trials"], "kept_by_the_host_check": st["spec_kept"]},
        "ms_graph_build_excluded": st["ms_structure"], "chi2_initial": st["chi2_initial"], "chi2_final": st["chi2_final"],
        "dtype": "f64", "huber_delta": delta, "huber_off": huber_off,
        "gpu_state": "hot (timed right after GPU work; the LM loop polls mapped host memory instead of synchronising the stream)",
        "roofline": {"bound": "mfma", "kernel": "k_chol_trsm_update<true> (a level: diagonal tiles + strips + update) / k_chol_update / k_chol_pair (+ k_chol_backsolve): tile Cholesky of the reduced "
                                               "camera system on v_mfma_f64_16x16x4",
                     "achieved": fl / t_solve / 1e12 if t_solve > 0 else None, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": fl / t_solve / 1e12 / FP64_MATRIX_PEAK_TFLOPS if t_solve > 0 else None,
                     "executed_flop_per_trial": fl, "solve_ms_per_trial": t_solve * 1e3, "schedule": info,
                     "traffic": traffic, "traffic_source": pmc_src,
                     "note": "EXECUTED FLOPs of the symbolic tile factorisation (dvm_ba_schedule_info) over the HIP-event time of the "
                             "factorisation + back substitution launches of a trial.  The solve is a dependency chain of elimination-tree "
                             "levels (diagonal tiles -> strips -> update per level), not matrix-pipe bound; dense-equivalent (n^3/3 per trial over the "
                             f"whole iteration): {FLOPS_DENSE_CHOLESKY * trials / dt / 1e12:.2f} TFLOP/s"},
        "phase_ms": {"first_linearisation_per_run": prof["ms_linearise"],
                     "schur_per_trial": prof["ms_schur"] / max(prof["trials"], 1), "cholesky_solve_per_trial": t_solve * 1e3,
                     "landmarks_update_linearise_per_trial": prof["ms_update_chi2"] / max(prof["trials"], 1),
                     "note": "a trial linearises its own state (edge pass with Jacobians + accumulation into alternate buffers); "
                             "an accepted trial's buffers are swapped in, so only the first iteration of a run linearises separately"},
        "hbm": hbm,
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


def run_loop_closed(device: int, iters: int = 10, repeats: int = 20, cpu_iters: int = 2):
    """Second workload of the BA metric: 500 keyframes / 20 000 landmarks again, but a map AFTER loop closures -- the camera has gone round the loop
    twice (keyframe i and i + 250 see the same landmarks: bands far off the diagonal of the reduced camera matrix) and 0.2 % of the landmarks carry
    three long-range observations from anywhere on the loop: 38 % of the 50 x 50 tile pairs are coupled before factorisation (the headline
    problem: a ring, 7.6 % of the tiles non-zero, 7 elimination levels).  Parity: the first `cpu_iters` iterations against the CPU oracle (its
    envelope Cholesky is close to dense here: seconds per trial)."""
    from dvm_slam_amd import capi, synth
    pr = synth.ba_problem(laps=2, long_range_frac=0.002, long_range_obs=3)
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    delta = float(np.sqrt(5.991))
    ba = capi.BundleAdjuster(device)
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    ba.optimize(2)
    dt, its, trials = 0.0, 0, 0
    for _ in range(repeats):
        ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
        st = ba.optimize(iters)
        dt += st["ms_optimize"] * 1e-3; its += st["iterations"]; trials += st["total_trials"]
    info = ba.schedule_info()
    solver = ba.solve_info()
    fl = executed_flops_per_trial(info)
    # HIP events around the phases (an extra, untimed pass): the reduced solve's share, for its roofline
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    ba.profile(1)
    ba.optimize(iters)
    prof = ba.profile(0)
    t_solve = prof["ms_cholesky_solve"] / max(prof["trials"], 1) * 1e-3
    roof = {"bound": "mfma", "kernel": "k_chol_flow (one persistent launch: tile tasks + chains + back substitution)" if solver["form"] == "flow" else "k_chol_* level launches",
            "achieved": fl / t_solve / 1e12 if t_solve > 0 else None, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": fl / t_solve / 1e12 / FP64_MATRIX_PEAK_TFLOPS if t_solve > 0 else None, "solve_ms_per_trial": t_solve * 1e3,
            "levels": info["levels"], "us_per_level": t_solve * 1e6 / max(info["levels"], 1),
            "note": "EXECUTED FLOPs of the symbolic tile factorisation over the HIP-event time of the solve; a dependency chain of `levels` tile "
                    "factorisations (7.7 us each, one wave's column chain) bounds it, not the matrix pipe"}
    out = {"problem": "500 KF / 20 000 landmarks, two laps of the loop + 0.2 % landmarks with 3 long-range observations (synth.ba_problem(laps=2, long_range_frac=0.002))",
           "observations": len(e), "value": its / dt, "unit": "iterations/s", "iterations": its, "trials": trials, "ms_per_iteration": dt / max(its, 1) * 1e3,
           "ms_graph_build_excluded": st["ms_structure"], "schedule": info, "tile_fill_of_factor": info["nz_tiles"] / (info["tiles_per_side"] * (info["tiles_per_side"] + 1) / 2),
           "executed_flop_per_trial": fl, "chi2_initial": st["chi2_initial"], "chi2_final": st["chi2_final"], "solver": solver["form"], "solver_info": solver,
           "roofline": roof, "phase_ms": {"schur_per_trial": prof["ms_schur"] / max(prof["trials"], 1), "cholesky_solve_per_trial": t_solve * 1e3,
                                          "landmarks_update_linearise_per_trial": prof["ms_update_chi2"] / max(prof["trials"], 1)}}
    if cpu_iters > 0:
        from oracle import pyoracle as po   # checker + cpu_baseline of this problem
        ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
        sg = ba.optimize(cpu_iters)
        pg, xg = ba.result()
        t0 = time.perf_counter()
        pc, xc, sc, _ = po.ba_optimize(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta, cpu_iters)
        tc = time.perf_counter() - t0
        par = {"iterations_compared": cpu_iters, "trials_equal": bool(sg["trials"] == sc["trials"]), "chi2_rel": abs(sg["chi2_final"] - sc["chi2_final"]) / sc["chi2_final"],
               "max_abs_pose": float(np.abs(pg - pc).max()), "max_abs_landmark": float(np.abs(xg - xc).max())}
        out["parity_vs_cpu"] = par
        out["cpu_baseline"] = {"value": sc["iterations"] / tc, "unit": "iterations/s", "cores": 1, "kind": "port", "sample": f"{sc['iterations']} iterations ({sc['total_trials']} trials), {tc:.1f} s"}
        if not (par["trials_equal"] and par["chi2_rel"] <= 1e-9 and par["max_abs_pose"] < 1e-6 and par["max_abs_landmark"] < 1e-6):
            raise RuntimeError(f"loop-closed BA leg: GPU result differs from the CPU oracle: {par}")
    ba.close()
    return out


def run_sharded(device: int, iters: int = 10, repeats: int = 5):
    """Config 5: the same problem, landmark-sharded over all ranks of the job (every rank calls this).  Returns the record on
    every rank; wall time is the max over ranks."""
    import torch
    import torch.distributed as dist
    from dvm_slam_amd import capi, sharded_ba, synth
    torch.cuda.set_device(device)   # the current device is per THREAD: bench.py calls this from a watchdog thread
    pr = synth.ba_problem()
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    delta = float(np.sqrt(5.991))
    sb = sharded_ba.ShardedBundleAdjuster(device)
    sb.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    sb.optimize(2)
    dt, its, trials = 0.0, 0, 0
    for _ in range(repeats):
        sb.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
        sb.calls = sb.bytes_reduced = 0
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = sb.optimize(iters)
        torch.cuda.synchronize()
        dt += time.perf_counter() - t0
        its += st["iterations"]; trials += st["total_trials"]
    t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{device}" if sb.on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    rec = {"metric": "BA iterations/sec, 500 KF / 20k landmarks, landmark-sharded over the ranks (config 5)", "value": its / dt,
           "unit": "iterations/s", "ranks": sb.world, "iterations": its, "trials": trials, "ms_per_iteration": dt / max(its, 1) * 1e3,
           "allreduce_calls_per_run": sb.calls, "allreduce_bytes_per_run": sb.bytes_reduced, "chi2_final": st["chi2_final"],
           "backend": "nccl (RCCL)" if sb.on_gpu else "gloo (through the host: test configuration)"}
    sb.close()
    return rec


def run_replica(device: int, iters: int = 10, repeats: int = 5):
    """SURVEY 8(e) "replicas": every rank solves its own copy of the 500-keyframe problem at the same time; returns this rank's
    iterations/s (bench.py adds them up)."""
    import torch
    from dvm_slam_amd import capi, synth
    torch.cuda.set_device(device)
    pr = synth.ba_problem()
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    delta = float(np.sqrt(5.991))
    ba = capi.BundleAdjuster(device)
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    ba.optimize(2)
    dt, its = 0.0, 0
    for _ in range(repeats):
        ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
        t0 = time.perf_counter()
        st = ba.optimize(iters)
        dt += time.perf_counter() - t0
        its += st["iterations"]
    ba.close()
    return its / dt


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
