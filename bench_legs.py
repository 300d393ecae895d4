"""Further legs of bench.py's JSON line (rank 0, N = 1): what BASELINE.json's configs 2 and 3 cost per CALL, as the reference's
threads would see it through the drop-in boundary -- host pointers in, host pointers out, one frame / one window at a time --
each with the CPU oracle timed beside it in the same run (`cpu_baseline`, kind "port", 1 thread).

  latency   one-frame dvm_orb_extract, SearchByProjection(Cur, Last) through libdvmslam_host, dvm_pose_optimize with batch 1, and the three
            as ONE device chain (dvmh_track_with_motion_model): median / p95 over 500 calls            (reference: Tracking.cc:1423-1426, :2610, :2632; Frame.cc:408-417 timers)
  lba       LocalBundleAdjustment-sized window (30 keyframes of which 10 fixed, 3 000 landmarks, ~15 k observations): LM
            iterations/s and the end-to-end call (set_problem + optimize(10) + get_result + edge_chi2)   (Optimizer.cc:1030-1387)
  ba_cold   the 500-keyframe global BA started on a GPU that has been idle for 2 s (the reference runs it from an otherwise
            idle LoopClosing / GBA thread)
  merge     BASELINE config 3, stage by stage: vocabulary transform -> DetectMergePossibility over a 500-keyframe database ->
            SearchByBoW -> 200 Sim3 hypotheses -> OptimizeSim3 -> SearchBySim3   (LoopClosing.cc:644-953)
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0
FP64_MATRIX_PEAK_TFLOPS = 78.6


def _stats(ts):
    a = np.sort(np.asarray(ts)) * 1e3
    return {"median_ms": float(a[len(a) // 2]), "p95_ms": float(a[int(0.95 * (len(a) - 1))]), "mean_ms": float(a.mean()), "calls": len(a)}


def _time_calls(fn, n, warm=5):
    for i in range(warm):
        fn(i)
    ts = []
    for i in range(n):
        t0 = time.perf_counter()
        fn(i)
        ts.append(time.perf_counter() - t0)
    return ts


def _pose_case(seed, n_pts=300, out_frac=0.1):
    from dvm_slam_amd import synth
    rng = np.random.default_rng(seed)
    pr = synth.ba_problem(n_kf=4, n_pts=n_pts, k_obs=4, seed=seed, noise_px=0.5, outlier_frac=0.0, radius=20.0)
    kf = 1 + seed % 3
    sel = pr["edge_pose"] == kf
    Xw = pr["points_gt"][pr["edge_point"][sel]]
    obs = pr["obs"][sel].copy()
    bad = rng.random(len(obs)) < out_frac
    obs[bad] += rng.choice([-1.0, 1.0], size=(int(bad.sum()), 2)) * 35.0
    return pr["poses"][kf], Xw, obs, pr["inv_sigma2"][sel], pr["intrinsics"]


def latency(capi, frames, device, calls=500, cpu_calls=24):
    """Per-call latency of the three calls Tracking makes per frame, through the drop-in boundary (host arrays in and out)."""
    ext = capi.OrbExtractor(max_batch=1, device=device)
    scale = ext.tables()["scale"]
    nfr = len(frames)
    t_ext = _time_calls(lambda i: ext.extract(frames[i % nfr]), calls)
    # SearchByProjection(Cur, Last): consecutive frames of the stream, last frame's keypoints carry map points at 3..9 m
    rng = np.random.default_rng(9)
    K = np.array([500.0, 500.0, 320.0, 240.0], np.float32)
    pairs = []
    prev = ext.extract(frames[0])
    for t in range(1, 9):
        cur = ext.extract(frames[t])
        _, k0, d0, _ = prev
        _, k1, d1, _ = cur
        z = rng.uniform(3, 9, len(k0)).astype(np.float32)
        mps = np.zeros(len(k0), capi.MAP_POINT_DTYPE)
        mps["pos"][:, 0] = (k0["x"] - K[2]) / K[0] * z; mps["pos"][:, 1] = (k0["y"] - K[3]) / K[1] * z; mps["pos"][:, 2] = z
        mps["desc"] = d0; mps["n_obs"] = 1
        pairs.append(dict(kps_c=k1, desc_c=d1, mp_c=np.full(len(k1), -1, np.int32), Tcw=np.array([0, 0, 0, 1, 0, 0, 0], np.float32), K=K,
                          bounds=np.array([0, 640, 0, 480], np.float32), scale_factors=scale, kps_l=k0, mp_l=np.arange(len(k0), dtype=np.int32),
                          outlier_l=None, mps=mps))
        prev = cur
    nm = []
    t_sbp = _time_calls(lambda i: nm.append(capi.search_by_projection_frames(th=15.0, device=device, **pairs[i % len(pairs)])[0]), calls)
    # the same search when the current frame's keypoints + descriptors are still in HBM where the extractor left them (dvm_orb_last_result ->
    # dvmh_frame_view::dev; the shims' Frame::mDvmDevice): extraction of frame t is outside the timed call, as it is in Tracking
    used = []
    def sbp_dev(i):
        t = 1 + i % 8
        ext.extract(frames[t])
        ref = ext.last_result()
        t0 = time.perf_counter()
        r = capi.search_by_projection_frames(th=15.0, device=device, dev_c=ref, **pairs[t - 1])
        used.append(r[3])
        return time.perf_counter() - t0
    for i in range(5):
        sbp_dev(i)
    t_sbp_dev = [sbp_dev(i) for i in range(calls)]
    cases = [_pose_case(100 + i) for i in range(8)]

    def pose(i):
        c = cases[i % 8]
        capi.pose_optimize(c[0][None], c[1][None], c[2][None], c[3][None], np.array([len(c[1])], np.int32), c[4], device)
    t_pose = _time_calls(pose, calls)
    # the whole tracked frame as ONE device chain behind one synchronisation (dvm_track_begin / dvm_track_finish through
    # dvmh_track_with_motion_model): pixels in -> keypoints, descriptors, mvpMapPoints after the outlier drop and the optimised pose out
    inv_s2 = ext.tables()["inv_sigma2"]
    trk = capi.Tracker(ext, device=device)
    tinfo = []

    def track(i):
        t = 1 + i % 8
        p = pairs[t - 1]
        r = trk.track(frames[t], p["Tcw"], K, p["bounds"], scale, inv_s2, p["kps_l"], p["mp_l"], None, p["mps"], th=15.0)
        tinfo.append((r["nmatches_search"], r["n_inliers"], r["tracked"], r["replayed_on_host"], r["wide_window"], r["n_requeried"]))
    t_track = _time_calls(track, calls)
    trk.close()
    ext.close()
    out = {"unit": "ms per call, host arrays in -> host arrays out (Python / ctypes harness around the C ABI)",
           "orb_extract_one_frame": _stats(t_ext), "search_by_projection_cur_last": dict(_stats(t_sbp), matches_per_call=float(np.mean(nm))),
           "search_by_projection_cur_last_device_resident": dict(_stats(t_sbp_dev), grid_built_from_hbm_fraction=float(np.mean(used)),
                                                                 note="Frame.cc:411 -> ORBmatcher.cc:1553 without re-uploading the frame: dvm_orb_last_result + dvmh_frame_view::dev"),
           "pose_optimization_one_frame": dict(_stats(t_pose), matches=int(np.mean([len(c[1]) for c in cases]))),
           "track_with_motion_model_one_frame": dict(_stats(t_track), matches_per_call=float(np.mean([a[0] for a in tinfo])), inliers_per_call=float(np.mean([a[1] for a in tinfo])),
                                                     tracked_fraction=float(np.mean([a[2] for a in tinfo])), replayed_on_host_fraction=float(np.mean([a[3] for a in tinfo])),
                                                     wide_window_fraction=float(np.mean([a[4] for a in tinfo])), windows_searched_again_per_call=float(np.mean([a[5] for a in tinfo])),
                                                     note="Frame::Frame -> ExtractORB + TrackWithMotionModel (SearchByProjection(Cur, Last) -> PoseOptimization -> outlier drop) as one "
                                                          "enqueue: dvmh_track_with_motion_model, host image in -> host results out"),
           "reference": "Tracking.cc:1423-1426 (Frame ctor -> ORBextractor::operator()), :2610 (SearchByProjection), :2632 (PoseOptimization)"}
    if cpu_calls > 0:
        from oracle import pyoracle as po   # cpu_baseline leg: the checker timed as the baseline
        orc = po.OrbOracle()
        c_ext = _time_calls(lambda i: orc.extract(frames[i % nfr]), cpu_calls, warm=1)
        c_sbp = _time_calls(lambda i: po.search_by_projection_frames(th=15.0, **pairs[i % len(pairs)]), cpu_calls, warm=1)
        c_pose = _time_calls(lambda i: po.pose_optimize(*cases[i % 8]), cpu_calls, warm=1)
        out["cpu_baseline"] = {"kind": "port", "cores": 1, "sample": f"{cpu_calls} calls each of the same inputs, CPU oracle",
                               "orb_extract_one_frame": _stats(c_ext), "search_by_projection_cur_last": _stats(c_sbp),
                               "pose_optimization_one_frame": _stats(c_pose)}
    return out


def batch_sweep(batches=(1, 8, 32), steps=3, frames_per_step=4096):
    """The headline metric at the launch-group sizes an ONLINE system sees -- 1 frame (one agent), 8 / 32 (that many agents' simultaneous frames on
    one GPU) -- next to the 256 of the main line: bench.py itself with --batch B (same stream, same two-lane pipeline, frames resident in HBM,
    extract + grid + match per launch group), one subprocess per size."""
    import json
    import subprocess
    out = {"unit": "frames/s (device-resident frames, two pipeline lanes)", "by_batch": {},
           "note": "bench.py --batch B --no-ba --no-legs --no-pcie --no-exclusive --cpu-seconds 0; launch-group latency = ms_per_launch_group"}
    for B in batches:
        chunks = max(2, min(frames_per_step // B, 1024))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", str(B), "--chunks-per-step", str(chunks), "--steps", str(steps), "--warmup", "1",
               "--stream-frames", str(max(256, 4 * B)), "--no-ba", "--no-legs", "--no-pcie", "--no-exclusive", "--cpu-seconds", "0"]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
            out["by_batch"][str(B)] = {"value": d["value"], "ms_per_launch_group": d["ms_per_step"] / chunks, "launch_groups": steps * chunks,
                                       "k_fast_cells_ms_per_launch": d["roofline"]["avg_launch_ms"] if d.get("roofline") else None}
        except Exception as ex:   # noqa: BLE001
            out["by_batch"][str(B)] = {"error": repr(ex)}
    return out


def low_texture(capi, device, cpu=True, steps=3):
    """Second workload of the extract + match metric (the headline stream is corner-rich: ~8 % of its pixels are FAST corners at iniThFAST, every
    level fills its keypoint quota): a weakly textured scene (synth.low_texture) -- ~1 % corners, many cells re-run at minThFAST
    (ORBextractor.cc:664-670), levels short of their quota.  Same pipeline, same batch; parity with the oracle asserted on frames of THIS stream."""
    import json
    import subprocess
    from dvm_slam_amd import synth
    out = {"unit": "frames/s", "stream": "synth.low_texture through the same camera path; bench.py --texture low (256-frame launch groups, two lanes, frames resident in HBM)"}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--texture", "low", "--steps", str(steps), "--warmup", "1", "--stream-frames", "512", "--no-ba", "--no-legs",
           "--no-pcie", "--cpu-seconds", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    out.update(value=d["value"], ms_per_launch_group=d["ms_per_step"] / d["config"]["launch_groups_per_step"],
               k_fast_cells={"avg_launch_ms": d["roofline"]["avg_launch_ms"], "achieved_GBps": d["roofline"]["achieved"], "frac_of_hbm_peak": d["roofline"]["frac"],
                             "exclusive": d["roofline"].get("exclusive")})
    frames = synth.frame_stream(3, texture="low")
    ext = capi.OrbExtractor(max_batch=1, device=device)
    got = [ext.extract(f) for f in frames]
    ext.close()
    out["keypoints_per_frame"] = [int(g[0]) for g in got]
    if cpu:
        from oracle import pyoracle as po   # checker + cpu_baseline of this stream
        orc = po.OrbOracle()
        t0 = time.perf_counter()
        ref = [orc.extract(f) for f in frames]
        dt = time.perf_counter() - t0
        ok = all(g[0] == o[0] and all(np.array_equal(g[1][k], o[1][k]) for k in ("x", "y", "size", "angle", "response", "octave")) and np.array_equal(g[2], o[2])
                 for g, o in zip(got, ref))
        out["parity_vs_cpu"] = {"frames": len(frames), "bit_identical": bool(ok)}
        out["cpu_baseline"] = {"value": len(frames) / dt, "unit": "frames/s (extract only)", "cores": 1, "kind": "port", "sample": f"{len(frames)} frames of this stream"}
        if not ok:
            raise RuntimeError("low_texture leg: extraction differs from the CPU oracle")
    return out


def online_agents(capi, frames, device, Ks=(1, 4, 8, 16, 32, 64), frames_per_agent=150):
    """K agents tracking on ONE GPU through the drop-in boundary, a host thread and an extractor handle each (the online shape of BASELINE
    config 4 with more agents than GPUs): per frame ORBextractor::operator() -> SearchByProjection(Cur, Last) -> PoseOptimization, host
    arrays in and out, every call synchronous as Tracking makes them.  Whole-job frames/s over the K threads; results of every thread are
    checked against the single-thread run of the same frames (same calls, same bits)."""
    import threading
    scale = None
    K4 = np.array([500.0, 500.0, 320.0, 240.0], np.float32)
    rng = np.random.default_rng(9)
    ext0 = capi.OrbExtractor(max_batch=1, device=device)
    scale = ext0.tables()["scale"]
    cyc = 8
    exts = [ext0.extract(frames[t]) for t in range(cyc + 1)]
    pairs = []
    for t in range(1, cyc + 1):
        _, k0, d0, _ = exts[t - 1]
        _, k1, d1, _ = exts[t]
        z = rng.uniform(3, 9, len(k0)).astype(np.float32)
        mps = np.zeros(len(k0), capi.MAP_POINT_DTYPE)
        mps["pos"][:, 0] = (k0["x"] - K4[2]) / K4[0] * z; mps["pos"][:, 1] = (k0["y"] - K4[3]) / K4[1] * z; mps["pos"][:, 2] = z
        mps["desc"] = d0; mps["n_obs"] = 1
        pairs.append(dict(kps_c=k1, desc_c=d1, mp_c=np.full(len(k1), -1, np.int32), Tcw=np.array([0, 0, 0, 1, 0, 0, 0], np.float32), K=K4,
                          bounds=np.array([0, 640, 0, 480], np.float32), scale_factors=scale, kps_l=k0, mp_l=np.arange(len(k0), dtype=np.int32),
                          outlier_l=None, mps=mps))
    cases = [_pose_case(100 + i) for i in range(cyc)]
    ext0.close()

    def agent(nframes, out, gate=None):
        ext = capi.OrbExtractor(max_batch=1, device=device)
        ext.extract(frames[0])      # buffers sized, kernels loaded, and this thread's grid handle / staging context created: outside the timed region
        capi.search_by_projection_frames(th=15.0, device=device, **pairs[0])
        c0 = cases[0]
        capi.pose_optimize(c0[0][None], c0[1][None], c0[2][None], c0[3][None], np.array([len(c0[1])], np.int32), c0[4], device)
        if gate is not None:
            gate.wait()             # all agents ready
            gate.wait()             # clock started
        acc = []
        for i in range(nframes):
            t = 1 + i % cyc
            n, k, d, _ = ext.extract(frames[t])
            nm = capi.search_by_projection_frames(th=15.0, device=device, **pairs[t - 1])[0]
            c = cases[t - 1]
            po = capi.pose_optimize(c[0][None], c[1][None], c[2][None], c[3][None], np.array([len(c[1])], np.int32), c[4], device)
            if i < cyc:
                acc.append((n, int(d[:n].astype(np.int64).sum()), int(nm), po[0].tobytes(), int(po[2][0])))
        ext.close()
        out.append(acc)

    ref = []
    agent(2 * cyc, ref)        # warm-up + the single-thread reference
    out = {"unit": "frames/s over all agents (extract + SearchByProjection + PoseOptimization per frame, host arrays in -> out, one thread per agent)",
           "note": "Python threads around the C ABI: the calls release the GIL, the argument marshalling between them does not -- a C++ host would sit above these numbers",
           "by_agents": {}, "reference": "Tracking.cc:1423-1426, :2610, :2632 per frame; one System per agent (orb_slam3_wrapper.cpp)"}
    same = True
    for K in Ks:
        res, ths = [], []
        gate = threading.Barrier(K + 1)
        for _ in range(K):
            th = threading.Thread(target=agent, args=(frames_per_agent, res, gate))
            th.start(); ths.append(th)
        gate.wait()
        t0 = time.perf_counter()
        gate.wait()
        for th in ths:
            th.join()
        dt = time.perf_counter() - t0
        same = same and len(res) == K and all(r == ref[0] for r in res)
        out["by_agents"][str(K)] = {"value": K * frames_per_agent / dt, "ms_per_frame_per_agent": dt / frames_per_agent * 1e3}
    out["identical_to_single_thread"] = bool(same)
    if not same:
        raise RuntimeError("online_agents leg: a thread's results differ from the single-thread run")
    # the same from a C++ host (tools/online_agents.cpp: std::threads on the C ABI, no interpreter lock between the calls)
    try:
        out["cpp_host"] = _online_agents_cpp(frames, cyc, scale, pairs, cases, device, Ks, frames_per_agent)
    except Exception as e:   # the tool is an extra: a missing compiler must not take the leg down
        out["cpp_host"] = {"skipped": f"{type(e).__name__}: {e}"}
    return out


def _online_agents_cpp(frames, cyc, scale, pairs, cases, device, Ks, frames_per_agent):
    import json
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(root, "tools", "bin", "online_agents")
    src = os.path.join(root, "tools", "online_agents.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", src, "-I" + os.path.join(root, "include"), "-L" + os.path.join(root, "dvm_slam_amd", "lib"),
                               "-ldvmslam_host", "-ldvmslam_hip", "-lpthread", "-Wl,-rpath," + os.path.join(root, "dvm_slam_amd", "lib"), "-o", exe])
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        path = f.name
        fr = np.ascontiguousarray(frames[:cyc + 1], np.uint8)
        f.write(np.array([cyc, fr.shape[1], fr.shape[2]], np.int32).tobytes())
        f.write(fr.tobytes())
        f.write(np.ascontiguousarray(scale[:8], np.float32).tobytes())
        for p in pairs:
            f.write(np.array([len(p["kps_c"]), len(p["kps_l"])], np.int32).tobytes())
            f.write(np.ascontiguousarray(p["kps_c"]).tobytes()); f.write(np.ascontiguousarray(p["desc_c"], np.uint8).tobytes())
            f.write(np.ascontiguousarray(p["kps_l"]).tobytes()); f.write(np.ascontiguousarray(p["mps"]).tobytes())
        for c in cases:
            f.write(np.array([len(c[1])], np.int32).tobytes())
            f.write(np.ascontiguousarray(c[0], np.float64).tobytes()); f.write(np.ascontiguousarray(c[1], np.float64).tobytes())
            f.write(np.ascontiguousarray(c[2], np.float64).tobytes()); f.write(np.ascontiguousarray(c[3], np.float64).tobytes())
            f.write(np.ascontiguousarray(c[4][:4], np.float64).tobytes())
    try:
        r = subprocess.run([exe, "--pool=8", path, str(device), str(frames_per_agent)] + [str(k) for k in Ks], capture_output=True, text=True, timeout=300)
    finally:
        os.unlink(path)
    if r.returncode != 0:
        raise RuntimeError(f"online_agents (C++ host) failed: {r.stderr.strip()[-300:]}")
    res = json.loads(r.stdout.strip().splitlines()[-1])
    if not res.get("identical_results_across_agents", False):
        raise RuntimeError("online_agents (C++ host): agents disagree")
    return res


def lba(device, iters=10, repeats=40, cpu_seconds=4.0):
    """LocalBundleAdjustment as LocalMapping calls it per keyframe: a covisibility window with anchors."""
    import ba_bench
    from dvm_slam_amd import capi, synth
    pr = synth.ba_problem(n_kf=30, n_pts=3000, k_obs=5, seed=0x1BA, radius=12.0)
    pr["fixed"][:10] = 1                                         # lFixedCameras: keyframes that see the window's points from outside it
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    delta = float(np.sqrt(np.float32(5.991)))
    ba = capi.BundleAdjuster(device)
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    ba.optimize(2)
    t_call, t_opt, its, trials = [], 0.0, 0, 0
    for _ in range(repeats):
        t0 = time.perf_counter()
        ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
        st = ba.optimize(iters)
        pg, xg = ba.result()
        chi, front = ba.edge_chi2()
        t_call.append(time.perf_counter() - t0)
        t_opt += st["ms_optimize"] * 1e-3; its += st["iterations"]; trials += st["total_trials"]
    info = ba.schedule_info()
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    ba.profile(1)
    ba.optimize(iters)
    prof = ba.profile(0)
    ba.close()
    E, L = len(e), len(pr["points"])
    fl = ba_bench.executed_flops_per_trial(info)
    t_solve = prof["ms_cholesky_solve"] / max(prof["trials"], 1) * 1e-3
    t_edge = prof["ms_update_chi2"] / max(prof["trials"], 1) * 1e-3
    edge_bytes = E * (18 * 8 + 64) + L * 9 * 8 * 2 + E * (256 + 192)
    out = {"problem": {"keyframes": 30, "fixed_keyframes": 10, "landmarks": L, "observations": E, "huber_delta": delta, "iterations_per_call": iters},
           "value": its / t_opt, "unit": "LM iterations/s (inside dvm_ba_optimize)", "ms_per_iteration": t_opt / its * 1e3, "iterations": its, "trials": trials,
           "end_to_end_call": dict(_stats(t_call), includes="dvm_ba_set_problem + dvm_ba_optimize(10) + dvm_ba_get_result + dvm_ba_edge_chi2, host arrays"),
           "ms_graph_build": st["ms_structure"], "schedule": info,
           "roofline": {"solve": {"bound": "mfma", "achieved": fl / t_solve / 1e12 if t_solve > 0 else None, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": fl / t_solve / 1e12 / FP64_MATRIX_PEAK_TFLOPS if t_solve > 0 else None, "ms_per_trial": t_solve * 1e3,
                                  "executed_flop_per_trial": fl},
                        "landmarks_update_linearise": {"bound": "hbm", "achieved": edge_bytes / t_edge / 1e9 if t_edge > 0 else None, "peak": HBM_PEAK_GBS,
                                                       "unit": "GB/s", "frac": edge_bytes / t_edge / 1e9 / HBM_PEAK_GBS if t_edge > 0 else None,
                                                       "ms_per_trial": t_edge * 1e3, "bytes_per_trial": edge_bytes},
                        "note": "at this size every kernel is a handful of workgroups: the call is launch- and dependency-latency bound, the fractions say so"},
           "phase_ms_per_trial": {"schur": prof["ms_schur"] / max(prof["trials"], 1), "cholesky_solve": t_solve * 1e3, "landmarks_update_linearise": t_edge * 1e3},
           "reference": "Optimizer.cc:1030-1387 (LocalMapping.cc:172)"}
    if cpu_seconds > 0:
        from oracle import pyoracle as po   # cpu_baseline leg
        eo = po.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
        done, runs, t0 = 0, 0, time.perf_counter()
        tc = []
        while True:
            t1 = time.perf_counter()
            p, x, so, chio = po.ba_optimize(pr["poses"], pr["fixed"], pr["points"], eo, pr["intrinsics"], delta, iters)
            po.ba_edge_chi2(p, x, eo, pr["intrinsics"])
            tc.append(time.perf_counter() - t1)
            done += so["iterations"]; runs += 1
            if time.perf_counter() - t0 >= cpu_seconds or runs >= 200:
                break
        out["cpu_baseline"] = {"value": done / sum(tc), "unit": "LM iterations/s", "cores": 1, "kind": "port", "end_to_end_call": _stats(tc),
                               "sample": f"{runs} runs of optimize({iters}) + edge chi2 on the same window, CPU oracle, {sum(tc):.1f} s"}
        out["parity_vs_cpu"] = {"trials_equal": bool(st["trials"] == so["trials"]), "max_abs_pose": float(np.abs(pg - p).max()),
                                "max_abs_landmark": float(np.abs(xg - x).max())}
        if not (out["parity_vs_cpu"]["trials_equal"] and out["parity_vs_cpu"]["max_abs_pose"] < 1e-6 and out["parity_vs_cpu"]["max_abs_landmark"] < 1e-6):
            raise RuntimeError(f"lba leg: GPU result differs from the CPU oracle: {out['parity_vs_cpu']}")
    return out


def lba_batch(device, Ks=(1, 8, 32), iters=10, repeats=6, cpu_windows=4):
    """BASELINE config 4 with more agents than GPUs: the LocalBundleAdjustment windows of K agents solved side by side by ONE launch of the
    sequential-order kernel (dvm_ba_optimize_windows; a workgroup per window, g2o's summation order -> bit-identical to the oracle), next to
    the same K windows solved one after the other through the tile solver's handle (the `lba` leg's call) and solved concurrently by
    dvm_ba_optimize_batch (pooled handles, host threads of the call).  K different windows: every
    agent has its own map (different seeds), 30 keyframes / 20 free / 3 000 landmarks / ~15 000 observations each."""
    from dvm_slam_amd import capi, synth
    delta = float(np.sqrt(np.float32(5.991)))
    kmax = max(Ks)
    wins = []
    for a in range(kmax):
        pr = synth.ba_problem(n_kf=30, n_pts=3000, k_obs=5, seed=0x1BA + a, radius=12.0)
        pr["fixed"][:10] = 1
        wins.append(dict(poses=pr["poses"], fixed=pr["fixed"], points=pr["points"], edges=capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"]),
                         intrinsics=pr["intrinsics"], huber_delta=delta, iterations=iters))
    capi.ba_optimize_windows(wins[:1], device)         # kernel load
    out = {"problem": {"keyframes": 30, "fixed_keyframes": 10, "landmarks": 3000, "observations_window_0": len(wins[0]["edges"]), "iterations_per_call": iters,
                       "huber_delta": delta}, "unit": "LM iterations/s, summed over the K windows of a call (host arrays in, results out)", "by_K": {},
           "reference": "Optimizer.cc:1030-1387 per window; one window per agent (orb_slam3_wrapper.cpp runs one System per agent)"}
    res = None
    for K in Ks:
        ts, its = [], 0
        for _ in range(repeats):
            t0 = time.perf_counter()
            res = capi.ba_optimize_windows(wins[:K], device)
            ts.append(time.perf_counter() - t0)
            its = sum(r["stats"]["iterations"] for r in res)
        best = float(np.median(ts))
        out["by_K"][str(K)] = {"value": its / best, "ms_per_call": best * 1e3, "ms_per_window": best * 1e3 / K, "iterations": its,
                               "host_setup_ms_per_window": res[0]["stats"]["ms_structure"]}
    # the same K windows by ONE launch of the FAST form (dvm_ba_optimize_windows_fast: LM control on the device, tree sums in a fixed order;
    # the general solver's tolerance contract instead of bit identity).  K = 128: four times the 32 windows (the launch fills 128 CUs)
    capi.ba_optimize_windows(wins[:1], device, fast=True)
    fast = {"call": "dvm_ba_optimize_windows_fast (one launch, a cluster of up to 8 workgroups per window: k_ba_window_cluster)", "by_K": {}}
    resf = None
    for K in tuple(Ks) + (4 * kmax,):
        batch = capi.BaWindowBatch([wins[a % kmax] for a in range(K)])     # marshalled once (what a C++ agent node holds); run() = the library call
        ts, its = [], 0
        for _ in range(repeats):
            t0 = time.perf_counter()
            batch.run(device, fast=True, collect=False)
            ts.append(time.perf_counter() - t0)
            resf = batch.results()
            its = sum(r["stats"]["iterations"] for r in resf)
        best = float(np.median(ts))
        fast["by_K"][str(K)] = {"value": its / best, "ms_per_call": best * 1e3, "ms_per_window": best * 1e3 / K, "iterations": its,
                                "host_setup_ms_per_window": resf[0]["stats"]["ms_structure"], "ms_upload_launch_download": resf[0]["stats"]["ms_optimize"]}
    dpo = max(float(np.abs(resf[a]["poses"] - res[a]["poses"]).max()) for a in range(min(len(res), kmax)))
    dxo = max(float(np.abs(resf[a]["points"] - res[a]["points"]).max()) for a in range(min(len(res), kmax)))
    same_trials = all(list(resf[a]["stats"]["trials"]) == list(res[a]["stats"]["trials"]) for a in range(min(len(res), kmax)))
    fast["vs_sequential_order_kernel"] = {"max_abs_pose": dpo, "max_abs_landmark": dxo, "same_trial_sequence": bool(same_trials), "tolerance": 1e-6,
                                          "note": "the sequential-order kernel's results are the CPU oracle's bit for bit (parity_vs_cpu below)"}
    if not (dpo < 1e-6 and dxo < 1e-6 and same_trials):
        raise RuntimeError(f"lba_batch leg: the fast form differs from the sequential-order kernel beyond the contract: {dpo} {dxo} {same_trials}")
    out["fast_windows"] = fast
    # ... and the same windows as BLOCKING one-window calls of K threads (one LocalMapping thread per agent) through dvm_ba_pool_*
    try:
        import threading
        pool = capi.BaPool(device, max_batch=32)
        by = {}
        for K in (8, kmax):
            batches = [capi.BaWindowBatch([wins[a]]) for a in range(K)]
            sizes, calls = [], 12
            def agent(a):
                for _ in range(calls):
                    sizes.append(pool.optimize(batches[a])[1])
            for a in range(K):
                pool.optimize(batches[a])
            th = [threading.Thread(target=agent, args=(a,)) for a in range(K)]
            t0 = time.perf_counter()
            for x in th: x.start()
            for x in th: x.join()
            dt = time.perf_counter() - t0
            its = sum(b.results()[0]["stats"]["iterations"] for b in batches) * calls
            by[str(K)] = {"value": its / dt, "ms_per_call_per_agent": dt / calls * 1e3, "mean_windows_per_launch": float(np.mean(sizes))}
        pool.close()
        out["fast_windows_pool"] = {"call": "dvm_ba_pool_optimize from K Python threads (the calls release the interpreter lock)", "by_K": by}
    except Exception as ex:   # noqa: BLE001
        out["fast_windows_pool"] = {"error": repr(ex)}
    # the same K windows through the tile solver, one handle, one after the other (what K agents queueing on one GPU get today)
    ba = capi.BundleAdjuster(device)
    t_seq, its_seq = [], 0
    for a in range(min(kmax, 8)):
        w = wins[a]
        t0 = time.perf_counter()
        ba.set_problem(w["poses"], w["fixed"], w["points"], w["edges"], w["intrinsics"], delta)
        st = ba.optimize(iters)
        ba.result(); ba.edge_chi2()
        t_seq.append(time.perf_counter() - t0); its_seq += st["iterations"]
    ba.close()
    out["tile_solver_sequential"] = {"value": its_seq / sum(t_seq), "ms_per_window": float(np.median(t_seq)) * 1e3, "windows": len(t_seq),
                                     "note": "dvm_ba_set_problem + dvm_ba_optimize + get_result + edge_chi2 per window, one after the other"}
    # ... and solved concurrently: dvm_ba_optimize_batch, host threads of the call each driving a pooled handle (what K agents' LocalMapping
    # threads sharing the GPU get).  Results must be the sequential ones bit for bit: the same solver runs each window.
    seq_ref = []
    ba = capi.BundleAdjuster(device)
    for a in range(min(kmax, 4)):
        w = wins[a]
        ba.set_problem(w["poses"], w["fixed"], w["points"], w["edges"], w["intrinsics"], delta)
        ba.optimize(iters)
        seq_ref.append(ba.result())
    ba.close()
    capi.ba_optimize_batch(wins[:kmax], device)   # the pool's handles exist from here on
    conc = {"call": "dvm_ba_optimize_batch (threads = library default)", "by_K": {}}
    resb = None
    for K in Ks:
        ts, its = [], 0
        for _ in range(repeats):
            t0 = time.perf_counter()
            resb = capi.ba_optimize_batch(wins[:K], device)
            ts.append(time.perf_counter() - t0)
            its = sum(r["stats"]["iterations"] for r in resb)
        best = float(np.median(ts))
        conc["by_K"][str(K)] = {"value": its / best, "ms_per_call": best * 1e3, "ms_per_window": best * 1e3 / K, "iterations": its}
    same = all(np.array_equal(resb[a]["poses"].view(np.int64), seq_ref[a][0].view(np.int64)) and
               np.array_equal(resb[a]["points"].view(np.int64), seq_ref[a][1].view(np.int64)) for a in range(min(len(seq_ref), len(resb))))
    conc["identical_to_sequential_handle"] = bool(same)
    if not same:
        raise RuntimeError("lba_batch leg: dvm_ba_optimize_batch differs from the same windows solved one after the other")
    out["concurrent_tile_solver"] = conc
    if cpu_windows > 0:
        from oracle import pyoracle as po   # cpu_baseline leg + parity of THIS run
        tc, itc = [], 0
        ident = True
        cpu_ref = []
        for a in range(min(cpu_windows, kmax)):
            w = wins[a]
            t0 = time.perf_counter()
            p, x, so, chio = po.ba_optimize(w["poses"], w["fixed"], w["points"], w["edges"], w["intrinsics"], delta, iters)
            tc.append(time.perf_counter() - t0); itc += so["iterations"]
            cpu_ref.append((p, x))
            g = res[a]
            ident = ident and bool(np.array_equal(g["poses"].view(np.int64), p.view(np.int64)) and np.array_equal(g["points"].view(np.int64), x.view(np.int64)) and
                                   np.array_equal(g["edge_chi2"].view(np.int64), chio.view(np.int64)) and list(g["stats"]["trials"]) == list(so["trials"]))
        out["cpu_baseline"] = {"value": itc / sum(tc), "unit": "LM iterations/s", "cores": 1, "kind": "port",
                               "sample": f"{len(tc)} of the windows, optimize({iters}) each, CPU oracle, {sum(tc):.1f} s"}
        out["parity_vs_cpu"] = {"windows_checked": len(tc), "bit_identical": ident}
        if not ident:
            raise RuntimeError("lba_batch leg: a window's result is not bit-identical to the CPU oracle")
        if cpu_ref:   # the concurrent call against the same oracle runs, at the general solver's tolerance
            dp = max(float(np.abs(resb[a]["poses"] - cpu_ref[a][0]).max()) for a in range(min(len(cpu_ref), len(resb))))
            dx = max(float(np.abs(resb[a]["points"] - cpu_ref[a][1]).max()) for a in range(min(len(cpu_ref), len(resb))))
            out["concurrent_tile_solver"]["parity_vs_cpu"] = {"max_abs_pose": dp, "max_abs_landmark": dx, "tolerance": 1e-6}
            if not (dp < 1e-6 and dx < 1e-6):
                raise RuntimeError(f"lba_batch leg: dvm_ba_optimize_batch differs from the CPU oracle: {dp} {dx}")
    return out


def lba_fast(device, K=32, iters=10, repeats=12):
    """K LocalBundleAdjustment windows (30 keyframes / 20 free / 3 000 landmarks / ~15 000 observations each, different maps) by ONE launch of
    dvm_ba_optimize_windows_fast (LM control on the device, a cluster of workgroups per window): the short form of lba_batch's `fast_windows` for the
    default run's contract line.  Reference: Optimizer.cc:1030-1387 per window, LocalMapping.cc:172."""
    from dvm_slam_amd import capi, synth
    delta = float(np.sqrt(np.float32(5.991)))
    wins = []
    for a in range(K):
        pr = synth.ba_problem(n_kf=30, n_pts=3000, k_obs=5, seed=0x1BA + a, radius=12.0)
        pr["fixed"][:10] = 1
        wins.append(dict(poses=pr["poses"], fixed=pr["fixed"], points=pr["points"], edges=capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"]),
                         intrinsics=pr["intrinsics"], huber_delta=delta, iterations=iters))
    batch = capi.BaWindowBatch(wins)          # marshalled once; run() = the library call, host arrays in -> results out
    for _ in range(3):            # (the pooled builder threads, the page-locked table block and the upload streams exist from the first call on)
        batch.run(device, fast=True)
    ts, its, res = [], 0, None
    for _ in range(repeats):
        t0 = time.perf_counter()
        batch.run(device, fast=True, collect=False)          # host arrays in -> host arrays out: the library call
        ts.append(time.perf_counter() - t0)
        res = batch.results()
        its = sum(r["stats"]["iterations"] for r in res)
    best = float(np.median(ts))
    # roofline of k_ba_window_cluster: ALGORITHMIC bytes of its phases (what each phase has to read and write once, doubles and index words;
    # DESIGN.md section 3) over the launch's HIP-event duration, against the HBM peak
    alg = 0.0
    for w, r in zip(wins, res):
        e = w["edges"]
        free = ~np.asarray(w["fixed"], bool)
        ef = free[e["pose"]]
        E, F, nact = len(e), int(ef.sum()), len(np.unique(e["point"]))
        deg = np.bincount(e["point"][ef])
        pairs = float((deg * (deg + 1) // 2).sum())
        per_iteration = 152.0 * E + 264.0 * F + 80.0 * E + 120.0 * F + 96.0 * nact                 # edge pass with Jacobians; Hll / bl, Hpp / bp
        per_trial = 432.0 * F + 48.0 * F + 288.0 * pairs + 168.0 * F + 24.0 * F + 196.0 * nact + 88.0 * E   # T rows; rhs + blocks; W^T x; landmarks; chi2 pass + sums
        alg += per_iteration * r["stats"]["iterations"] + per_trial * sum(r["stats"]["trials"])
    k_ms = res[0]["stats"]["kernel_us"] / 1e3
    traffic = None
    try:   # counter traffic of the same launch shape (K = 32: 256 workgroups) from the committed separate --pmc passes, calibrated as the main fold
        import csv
        import json
        cal = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")))["calibration"]
        tot = 0.0
        for cn, fac in (("FETCH_SIZE", cal["read"]["8"]), ("WRITE_SIZE", cal["write"]["8"])):
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"r06_pmc_lba_cluster_{cn}.csv"))) if int(r["Grid_Size"]) == 512 * 8 * 32]
            tot += float(np.mean(v)) * 1024.0 * fac
        traffic = tot if K == 32 else None
    except Exception:   # noqa: BLE001
        traffic = None
    roof = {"bound": "hbm", "kernel": "k_ba_window_cluster", "achieved": alg / (k_ms / 1e3) / 1e9 if k_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg / (k_ms / 1e3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else None, "algorithmic_bytes_per_launch": alg, "kernel_ms": k_ms, "traffic": traffic,
            "note": "algorithmic bytes of the data-parallel phases / the launch's HIP-event duration; counter traffic: profiles/r06_pmc_lba_cluster_*.csv"}
    return {"value": its / best, "unit": "LM iterations/s summed over the K windows of a call (host arrays in, results out)", "K": K, "ms_per_call": best * 1e3,
            "iterations": its, "call": "dvm_ba_optimize_windows_fast", "ms_upload_launch_download": res[0]["stats"]["ms_optimize"], "roofline": roof}


def ba_cold(device, iters=10, runs=5, idle_s=2.0):
    """The 500-keyframe global BA on a GPU that has been idle: the clocks have dropped, the first kernels pay the ramp."""
    from dvm_slam_amd import capi, synth
    pr = synth.ba_problem()
    e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    delta = float(np.sqrt(5.991))
    ba = capi.BundleAdjuster(device)
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    ba.optimize(2)
    vals, calls = [], []
    for _ in range(runs):
        ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
        ba.result()                                              # (drains the stream)
        time.sleep(idle_s)
        t0 = time.perf_counter()
        st = ba.optimize(iters)
        calls.append(time.perf_counter() - t0)
        vals.append(st["iterations"] / (st["ms_optimize"] * 1e-3))
    ba.close()
    return {"value": float(np.median(vals)), "unit": "iterations/s", "runs": runs, "idle_seconds_before_each_run": idle_s, "iterations_per_run": iters,
            "all_runs": [float(v) for v in vals], "optimize_call": _stats(calls),
            "note": "same problem and call as `ba`, but every optimize(10) starts after the GPU has sat idle (nothing else queued, power state dropped)"}


class _Timed:
    """Proxy around an ops object (merge.GpuOps or the oracle operators): accumulates wall time per stage."""

    def __init__(self, ops):
        self.ops, self.t, self.n = ops, {}, {}

    def __getattr__(self, name):
        fn = getattr(self.ops, name)
        if not callable(fn):
            return fn

        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - t0
            self.n[name] = self.n.get(name, 0) + 1
            return r
        return w


class _TimedDb:
    def __init__(self, db, acc):
        self.db, self.acc = db, acc

    def __getattr__(self, name):
        fn = getattr(self.db, name)
        if name != "detect_merge_possibility":
            return fn

        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            self.acc.t["detect_merge_possibility"] = self.acc.t.get("detect_merge_possibility", 0.0) + time.perf_counter() - t0
            self.acc.n["detect_merge_possibility"] = self.acc.n.get("detect_merge_possibility", 0) + 1
            return r
        return w


def merge(device, reps=20, cpu_reps=3, db_keyframes=500):
    """Config 3 as one chain per new keyframe, ms per stage (host arrays in / out per stage, as LoopClosing would call them)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dvm_slam_amd import capi, synth
    from dvm_slam_amd import merge as mg
    from merge_scene import make_two_agent_scene
    voc = synth.vocabulary(k=10, L=4, seed=5)
    sc = make_two_agent_scene(capi, 1, n_distract=db_keyframes - 1, s_w=0.7)     # (capi supplies the record dtypes the scene builder needs)
    triples = np.random.default_rng(1).integers(0, 1 << 30, (200, 3)).astype(np.int64)
    stages = ("transform", "detect_merge_possibility", "search_by_bow", "sim3_hypotheses", "optimize_sim3", "search_by_sim3")

    def chain(ops, reps_):
        T = _Timed(ops)
        t0 = time.perf_counter()
        peers = [dict(p) for p in sc["peers"]]                       # fill_database attaches each keyframe's BoW / feature vector
        db = mg.fill_database(ops, peers, 2)
        t_fill = time.perf_counter() - t0
        tdb = _TimedDb(db, T)
        res = None
        tot = []
        for _ in range(reps_):
            t1 = time.perf_counter()
            res = mg.merge_with_peer(T, sc["a"], sc["pa"], peers, sc["peer_pts"], tdb, 2, triples)
            tot.append(time.perf_counter() - t1)
        per = {s: T.t.get(s, 0.0) / max(T.n.get(s, 1), 1) * 1e3 for s in stages}
        return res, per, tot, t_fill

    gops = mg.GpuOps(voc, device)
    chain(gops, 2)                                                # warm-up (kernel load, handle creation)
    res, per, tot, t_fill = chain(gops, reps)
    ok = res["candidate"] == sc["true_idx"] and res.get("n_sim3_inliers", 0) > 80 and abs(res["S12"][7] / sc["gt"]["s"] - 1) < 0.02
    out = {"unit": "ms per stage call (host arrays in / out)", "database_keyframes": db_keyframes, "stage_ms": per, "chain": _stats(tot),
           "database_fill_ms_per_keyframe": t_fill / db_keyframes * 1e3,
           "result": {"candidate_found": bool(res["candidate"] == sc["true_idx"]), "bow_matches": int(res.get("n_bow_matches", 0)),
                      "sim3_inliers": int(res.get("n_sim3_inliers", 0)), "scale_error": float(abs(res["S12"][7] / sc["gt"]["s"] - 1)) if res.get("S12") is not None else None,
                      "matches_after_search_by_sim3": int((res["matches"] >= 0).sum()) if res.get("matches") is not None else None},
           "reference": "orb_slam3_wrapper.cpp:457-618 -> KeyFrameDatabase.cc:789-808; LoopClosing.cc:644-953 (SearchByBoW, Sim3Solver, OptimizeSim3, SearchBySim3)"}
    if not ok:
        raise RuntimeError(f"merge leg: the chain did not recover the planted similarity: {out['result']}")
    if cpu_reps > 0:
        from oracle import pyoracle as po   # cpu_baseline leg

        class OracleOps:
            def transform(self, desc, levelsup): return po.vocab_transform(voc, desc, levelsup)
            def new_database(self): return po.KeyFrameDatabase()
            def search_by_bow(self, a, b, nnratio): return po.search_by_bow_kf_kf(a["kps"], a["desc"], a["mp"], a["bad"], a["fv"], b["kps"], b["desc"], b["mp"], b["bad"], b["fv"], nnratio, True)
            def sim3_hypotheses(self, P1c, P2c, e1, e2, K1, K2, tr): return po.sim3_hypotheses(P1c, P2c, e1, e2, K1, K2, tr, False)
            def optimize_sim3(self, S12, P1c, P2c, o1, o2, w1, w2, K1, K2, th2): return po.optimize_sim3(S12, False, P1c, P2c, o1, o2, w1, w2, K1, K2, th2)
            def search_by_sim3(self, a, pa, b, pb, m12, idx2, S12, th): return po.search_by_sim3(a, pa, b, pb, S12, th, m12, idx2)
        res_c, per_c, tot_c, fill_c = chain(OracleOps(), cpu_reps)
        out["cpu_baseline"] = {"kind": "port", "cores": 1, "stage_ms": per_c, "chain": _stats(tot_c), "database_fill_ms_per_keyframe": fill_c / db_keyframes * 1e3,
                               "sample": f"{cpu_reps} chains on the same scene and database, CPU oracle"}
        out["parity_vs_cpu"] = {"candidate_equal": bool(res_c["candidate"] == res["candidate"]), "bow_matches_equal": bool(np.array_equal(res_c["bow_matches"], res["bow_matches"])),
                                "sim3_max_abs_diff": float(np.abs(res_c["S12"] - res["S12"]).max())}
    return out
