#!/usr/bin/env python3
"""bench.py -- throughput of the DVM-SLAM hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)

A "step" = `--chunks-per-step` x `--batch` (default 80 x 256 = 20 480) synthetic 640x480 frames of the agent's stream through
the whole front end on one GPU: ORB extract (8-level pyramid, per-cell FAST, octree, orientation, blur, rBRIEF, 1000 features)
+ frame-to-frame windowed Hamming matching against the previous frame (M3 rule, th=15), launched 256 frames at a time.
1024 distinct frames (315 MB, more than the 256 MiB MALL) are resident in HBM before the timed region and cycled; the
driver's 20 steps make a timed region of ~2 s.  One agent (= one frame stream) per GPU, no data-path
collective: `value` = frames processed by all ranks / max-over-ranks wall time ("weak" scaling).

Rank 0 prints, as the LAST stdout line, ONE compact JSON object (a few KB: compact()).  Besides the contract keys it carries
  roofline      -- dominant kernel (FAST cells): algorithmic bytes per launch / HIP-event duration
  cpu_baseline  -- the CPU oracle (port of the reference path) timed single-threaded on this host
  ba            -- bundle-adjustment iterations/s (second half of BASELINE.json's metric): ring map + loop-closed map, summary only
The complete record (per-stage event spans, VALU issue model, schedules ...) is written to gpurun_out/bench_full.json (--full-json).
With --legs (N = 1) the per-call costs of configs 2, 3 and 4 through the drop-in boundary (bench_legs.py: batch_sweep, low_texture,
latency, online_agents, lba, lba_batch, merge, ba_cold) run too; each prints its own `{"leg": ...}` stdout line BEFORE the contract line.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# compulsory stage traffic per 640x480 frame (pyramid materialised)
BYTES_PER_FRAME_TOTAL = 4_421_474
BYTES_PER_FRAME_FAST = 950_532  # the FAST stage reads every pyramid pixel once
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE of the launcher, else 1)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per launch group (per GPU)")
    ap.add_argument("--chunks-per-step", type=int, default=80, help="launch groups per step: a step is chunks x batch frames")
    ap.add_argument("--stream-frames", type=int, default=1024, help="distinct synthetic frames resident in HBM (cycled)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample budget (0 = skip)")
    ap.add_argument("--lanes", type=int, default=2, help="double-buffered pipeline instances the batches alternate over")
    ap.add_argument("--no-exclusive", action="store_true", help="skip the one-lane k_fast_cells pass (clean rocprofv3 averages)")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-resident-input leg (N=1 only)")
    ap.add_argument("--ba-iters", type=int, default=10)
    ap.add_argument("--legs", action="store_true", help="also run the per-call legs (batch sweep, latency, online agents, lba, merge, ba_cold; N=1 only): "
                                                         "each prints its own stdout line BEFORE the contract line")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the short config-2 / config-3 per-call legs of the default run")
    ap.add_argument("--no-legs", action="store_true", help="(default since round 5; accepted for old command lines)")
    ap.add_argument("--full-json", default=None, help="where the complete (uncompacted) record goes; default gpurun_out/bench_full.json")
    ap.add_argument("--texture", choices=("rich", "low"), default="rich", help="synthetic stream: the BASELINE corner-rich one, or the weakly textured second workload")
    return ap.parse_args()


def cpu_all_cores(libpath, seconds=3.0, max_procs=64):
    """One agent per core: the same single-thread oracle in `procs` processes at once (oracle/cpu_agent.py),
    aggregate frames/s."""
    procs = max(1, min(os.cpu_count() or 1, max_procs))
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_agent.py"), str(seconds), libpath or "-"]
    t0 = time.perf_counter()
    ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(procs)]
    res = []
    for p in ps:
        try:
            out, _ = p.communicate(timeout=120)
            n, t = out.split()[-2:]
            res.append((int(n), float(t)))
        except Exception:   # noqa: BLE001
            p.kill()
    wall = time.perf_counter() - t0
    if not res:
        raise RuntimeError("no CPU agent finished")
    return {"value": sum(n / t for n, t in res), "unit": "frames/s", "cores": len(res),
            "sample": f"{len(res)} processes x ~{seconds:.0f} s, one agent each ({sum(n for n, _ in res)} frames, {wall:.1f} s wall incl. start-up)"}


def cpu_baseline(frames: np.ndarray, budget_s: float):
    """Oracle (CPU port of the reference path), single thread: extract + windowed match per frame."""
    from oracle import pyoracle as po
    libpath = None
    try:  # -march=native copy for this host (the committed build is portable)
        out = "/tmp/liboracle_native.so"
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", f"OUT={out}", "ARCHFLAGS=-march=native"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        libpath = out
    except Exception:
        libpath = None
    orc = po.OrbOracle(libpath=libpath)
    scale = orc.tables()["scale"]
    prev = None
    done = 0
    t_total = 0.0
    i = 0
    while True:
        f = frames[i % len(frames)]
        t0 = time.perf_counter()
        n, k, d, _ = orc.extract(f)
        if prev is not None:
            g = po.Grid(k)
            kq, dq = prev
            g.match_window(d, dq, kq["x"], kq["y"], (np.float32(15) * scale[kq["octave"]]).astype(np.float32),
                           kq["octave"] - 1, kq["octave"] + 1)
        t_total += time.perf_counter() - t0
        prev = (k, d)
        done += 1
        i += 1
        if t_total >= budget_s or done >= 4096:
            break
    out = {"value": done / t_total, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{done} frames of the same synthetic 640x480 stream, extract+match, oracle built "
                     f"{'-O3 -march=native' if libpath else '-O3 portable'}, {os.cpu_count()} host cores present"}
    try:   # the node's CPU as the reference would use it for several agents: one agent per core, up to 64 at once
        out["one_agent_per_core"] = cpu_all_cores(libpath)
    except Exception as ex:   # noqa: BLE001 -- the single-thread figure above stands on its own
        out["one_agent_per_core"] = {"error": repr(ex)}
    return out


def pcie_inclusive_leg(capi, frames, B, steps, device):
    """Frames start in page-locked HOST memory: two extractor handles used alternately, so that one batch's PCIe
    transfer (dvm_orb_extract_staged) overlaps the other's kernels.  Reported next to `value`, never as `value`."""
    import torch
    H, W = frames.shape[1:]
    bounds = (0.0, float(W), 0.0, float(H))
    lanes = []
    for k in range(2):
        ext = capi.OrbExtractor(max_batch=B, device=device)
        grid = capi.FrameGrid(capacity=2048, slots=B, device=device)
        ext.staging(B, H, W)[:] = frames[(k * B) % len(frames):(k * B) % len(frames) + B]   # the camera driver's job, untimed
        ext.extract_staged(B, H, W)
        ext.sync()
        kp, dp, np_, cap = ext.result_device(0)
        lanes.append(dict(ext=ext, grid=grid, st=ext.stream(), res=(kp, dp, np_, cap), scale=ext.scale_factors_device(),
                          carry=(torch.zeros((cap, 7), dtype=torch.int32, device="cuda"), torch.zeros((cap, 32), dtype=torch.uint8, device="cuda"),
                                 torch.zeros(1, dtype=torch.int32, device="cuda")),
                          matches=torch.zeros((B, cap, 4), dtype=torch.int32, device="cuda"), nq=torch.zeros(B, dtype=torch.int32, device="cuda")))

    def step(i):
        ln = lanes[i & 1]
        ext, grid, (kp, dp, np_, cap) = ln["ext"], ln["grid"], ln["res"]
        ext.staging(B, H, W)            # waits until this handle's previous batch is through; the frames are already there
        ext.extract_staged(B, H, W)
        grid.build_batch_device(0, B, kp, cap, dp, cap * 32, np_, bounds, stream=ln["st"])
        grid.match_frames_batch(0, B, kp, cap, dp, cap * 32, np_, tuple(t.data_ptr() for t in ln["carry"]), cap, 15.0, ln["scale"], 8,
                                ln["matches"].data_ptr(), cap, ln["nq"].data_ptr(), stream=ln["st"])
        ext.copy_result(B - 1, *(t.data_ptr() for t in ln["carry"]))

    for i in range(4):
        step(i)
    for ln in lanes:
        ln["ext"].sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    for ln in lanes:
        ln["ext"].sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for ln in lanes:
        ln["ext"].close(); ln["grid"].close()
    return {"value": steps * B / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "h2d_gbps": steps * B * H * W / dt / 1e9,
            "note": "input in pinned host memory (dvm_orb_staging), 2 handles ping-pong: H2D of one batch under the kernels of the other"}


LEG_KEYS = ("batch_sweep", "low_texture", "latency", "online_agents", "lba", "lba_batch", "lba_fast", "merge", "ba_cold")


def pick(d, *keys):
    """Sub-dict of the keys that exist (None-safe)."""
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact(out):
    """The contract line: the keys the driver parses + roofline + cpu_baseline + a BA summary, a few KB.  Everything else
    of the record (per-kernel event spans, the VALU issue model, the legs) is in the full record (emit())."""
    c = pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data")
    cfg = dict(out["config"])
    cfg["ranks"] = pick(cfg.get("ranks", {}), "count", "backend", "cuda_device_of_rank", "launched_by")
    c["config"] = cfg
    r = out.get("roofline")
    if r:
        cr = pick(r, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "avg_launch_ms", "frames_per_launch",
                  "pipeline_achieved", "pipeline_frac", "all_stage_bytes_per_frame", "pipeline_traffic_over_algorithmic")
        cr["traffic_source"] = pick(r.get("traffic_source") or {}, "file", "stale")
        if r.get("exclusive"):
            cr["exclusive"] = pick(r["exclusive"], "avg_launch_ms", "achieved", "frac")
        try:
            cr["valu_busy"] = r["valu_issue"]["counter_valu_busy"]["k_fast_cells"]
        except Exception:   # noqa: BLE001
            cr["valu_busy"] = None
        c["roofline"] = cr
    else:
        c["roofline"] = None
    cb = out.get("cpu_baseline")
    if cb:
        ccb = pick(cb, "value", "unit", "cores", "kind", "sample")
        if isinstance(cb.get("one_agent_per_core"), dict):
            ccb["one_agent_per_core"] = pick(cb["one_agent_per_core"], "value", "cores", "error")
        c["cpu_baseline"] = ccb
    ba = out.get("ba")
    if isinstance(ba, dict):
        cba = pick(ba, "metric", "value", "unit", "ms_per_iteration", "dtype", "iterations", "trials")
        if isinstance(ba.get("roofline"), dict):
            cba["roofline"] = pick(ba["roofline"], "bound", "achieved", "peak", "unit", "frac", "solve_ms_per_trial", "executed_flop_per_trial", "traffic")
        try:
            cba["hbm_schur"] = pick(ba["hbm"]["schur"], "achieved", "frac", "traffic", "traffic_over_algorithmic")
        except Exception:   # noqa: BLE001
            pass
        if isinstance(ba.get("cpu_baseline"), dict):
            cba["cpu_baseline"] = pick(ba["cpu_baseline"], "value", "unit", "cores", "kind")
        if isinstance(ba.get("parity_vs_cpu"), dict):
            cba["parity_vs_cpu"] = ba["parity_vs_cpu"]
        lc = ba.get("loop_closed")
        if isinstance(lc, dict):
            clc = pick(lc, "value", "unit", "ms_per_iteration", "solver", "tile_fill_of_factor", "executed_flop_per_trial", "error")
            if isinstance(lc.get("roofline"), dict):
                clc["roofline"] = pick(lc["roofline"], "bound", "kernel", "achieved", "peak", "unit", "frac")
            if isinstance(lc.get("cpu_baseline"), dict):
                clc["cpu_baseline"] = pick(lc["cpu_baseline"], "value", "cores")
            if isinstance(lc.get("parity_vs_cpu"), dict):
                clc["parity_vs_cpu"] = lc["parity_vs_cpu"]
            cba["loop_closed"] = clc
        c["ba"] = cba
    if isinstance(out.get("ba_sharded"), dict):
        sh = out["ba_sharded"]
        c["ba_sharded"] = pick(sh, "value", "unit", "ranks", "ms_per_iteration", "error")
        if isinstance(sh.get("replicas"), dict):
            c["ba_sharded"]["replicas"] = pick(sh["replicas"], "value", "ranks")
    if isinstance(out.get("pcie_inclusive"), dict):
        c["pcie_inclusive"] = pick(out["pcie_inclusive"], "value", "unit", "h2d_gbps")
    c["sanity_matches_le_TH_HIGH_last_step"] = out.get("sanity_matches_le_TH_HIGH_last_step")
    cl = {}
    try:
        lat = out["latency"]
        cl["unit"] = "ms per call (median), host arrays in -> host arrays out"
        for k_out, k_in in (("extract", "orb_extract_one_frame"), ("search_by_projection", "search_by_projection_cur_last"),
                            ("pose_optimization", "pose_optimization_one_frame"), ("track_frame_one_chain", "track_with_motion_model_one_frame")):
            cl[k_out] = round(lat[k_in]["median_ms"], 4)
    except Exception:   # noqa: BLE001
        pass
    try:
        cl["lba_window_call"] = round(out["lba"]["end_to_end_call"]["median_ms"], 4)
        cl["lba_window_iterations_per_s"] = round(out["lba"]["value"], 1)
    except Exception:   # noqa: BLE001
        pass
    try:
        cl["lba_32_windows_one_launch_iterations_per_s"] = round(out["lba_fast"]["value"], 1)
        cl["lba_32_windows_roofline"] = pick(out["lba_fast"]["roofline"], "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms")
    except Exception:   # noqa: BLE001
        pass
    try:
        cl["merge_chain"] = round(out["merge"]["chain"]["median_ms"], 4)
    except Exception:   # noqa: BLE001
        pass
    if cl:
        c["config_legs"] = cl
    return c


def emit(out, a):
    """stdout: one line per leg (when --legs ran them), then -- LAST -- the compact contract line.  The complete record goes to a
    file (gpurun_out/bench_full.json travels back from the GPU box) so that nothing the old one-line form carried is lost."""
    path = a.full_json or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f)
    except OSError as ex:
        print(f"bench.py: full record not written ({ex})", file=sys.stderr)
    for k in LEG_KEYS:
        if k in out and a.legs:       # (the default run's short config legs live in the full record and in `config_legs` of the line)
            print(json.dumps({"leg": k, "record": out[k]}), flush=True)
    line = json.dumps(compact(out))
    print(line, flush=True)


def relaunch_distributed(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): become the launcher -- the same
    command under `torch.distributed.run --nproc-per-node N`, one rank per GPU over RCCL, rendezvous on 127.0.0.1."""
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["DVM_BENCH_RELAUNCHED"] = "1"
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if a.gpus is None:
        a.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_distributed(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks (one rank per GPU: they must agree)")
    # test hooks (tests/test_gpu_bench.py): several ranks sharing GPU 0 over gloo exercise the N > 1 control flow on a 1-GPU box
    backend = os.environ.get("DVM_BENCH_BACKEND", "nccl")
    if os.environ.get("DVM_BENCH_SHARE_GPU") == "1":
        local = 0
    import torch
    import torch.distributed as dist

    from dvm_slam_amd import capi, exchange, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    torch.cuda.set_device(local)
    # DVM_BENCH_FORCE_DIST=1 (tests/test_gpu_rccl.py): initialise the process group even for one rank, so that the RCCL branch --
    # communicator on this GPU, barriers, the max-over-ranks all-reduce, the sharded-BA collectives on the solver's stream --
    # executes on a 1-GPU box exactly as it does for N > 1
    use_dist = world > 1 or os.environ.get("DVM_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29650")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        exchange.SHORTCUT_SINGLE_RANK = not (world == 1)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    def flush_c_stdio():
        # RCCL announces itself ("Librccl path : ...") through C stdio, which is block-buffered on a pipe and would otherwise
        # land AFTER the JSON line when the process exits: push it out now, on every rank
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:   # noqa: BLE001
            pass

    if use_dist:
        dist.barrier()          # creates the communicator (and makes RCCL print) before anything is timed
        flush_c_stdio()
    B = a.batch
    nstream = max(a.stream_frames, B)
    nstream = (nstream + B - 1) // B * B
    # every rank is its own agent: a different segment of the camera path
    frames = synth.frame_stream(nstream, start=exchange.agent_stream_segment(rank, nstream), texture=a.texture)
    d_frames = torch.from_numpy(frames).cuda()
    H, W = frames.shape[1:]

    # Double-buffered pipeline instances ("lanes"): consecutive 256-frame batches of the agent's stream go to alternating
    # extractor handles, so that the latency-bound tail of one batch (octree, descriptors, matching) overlaps the
    # throughput-bound head of the next (pyramid, FAST).  Frame order is preserved: the frame-to-frame match of a batch's
    # first frame takes its queries from the previous batch's last frame (the carry), across lanes, ordered by an event.
    nl = max(1, a.lanes)
    bounds = (0.0, float(W), 0.0, float(H))
    lanes = []
    for k in range(nl):
        ext = capi.OrbExtractor(max_batch=B, device=local)
        grid = capi.FrameGrid(capacity=2048, slots=B, device=local)
        ext.extract_batch_device(d_frames.data_ptr(), B, H, W)       # one un-timed call sizes the device buffers
        ext.sync()
        k_ptr, d_ptr, n_ptr, cap = ext.result_device(0)
        lanes.append(dict(ext=ext, grid=grid, st=ext.stream(), ts=torch.cuda.ExternalStream(ext.stream()), res=(k_ptr, d_ptr, n_ptr),
                          carry=(torch.zeros((cap, 7), dtype=torch.int32, device="cuda"), torch.zeros((cap, 32), dtype=torch.uint8, device="cuda"),
                                 torch.zeros(1, dtype=torch.int32, device="cuda")),
                          carry_ready=torch.cuda.Event(), match_done=torch.cuda.Event(), matches=torch.zeros((B, cap, 4), dtype=torch.int32, device="cuda"),
                          nq=torch.zeros(B, dtype=torch.int32, device="cuda"), scale=ext.scale_factors_device()))
    nbatches = nstream // B

    chunks = max(1, a.chunks_per_step)

    def step(s_idx):
        for j in range(chunks):
            chunk(s_idx * chunks + j)

    def chunk(i):
        ln, prev = lanes[i % nl], lanes[(i - 1) % nl]
        ext, grid, (k_ptr, d_ptr, n_ptr) = ln["ext"], ln["grid"], ln["res"]
        off = (i % nbatches) * B
        ext.extract_batch_device(d_frames.data_ptr() + off * H * W, B, H, W)
        grid.build_batch_device(0, B, k_ptr, cap, d_ptr, cap * 32, n_ptr, bounds, stream=ln["st"])
        if nl > 1 and i > 0:
            ln["ts"].wait_event(prev["carry_ready"])                 # the previous batch's last frame has been copied out
        grid.match_frames_batch(0, B, k_ptr, cap, d_ptr, cap * 32, n_ptr, tuple(t.data_ptr() for t in prev["carry"]), cap, 15.0,
                                ln["scale"], 8, ln["matches"].data_ptr(), cap, ln["nq"].data_ptr(), stream=ln["st"])
        ln["match_done"].record(ln["ts"])
        if nl > 1 and i > 1:
            ln["ts"].wait_event(prev["match_done"])                  # the reader of this lane's previous carry has finished
        ext.copy_result(B - 1, *(t.data_ptr() for t in ln["carry"]))
        ln["carry_ready"].record(ln["ts"])

    def barrier():
        for ln in lanes:
            ln["ext"].sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    for ln in lanes:
        ln["ext"].sync()
        ln["ext"].profiling(2)          # HIP events around the dominant kernel only: the roofline's live launch duration.  Events around
        ln["ext"].profile_reset()       # all six stages are twelve marker packets per launch group = 1-2 % of the timed region
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    for ln in lanes:
        ln["ext"].profiling(False)
    dt = exchange.max_over_ranks(dt, device="cuda" if backend == "nccl" else "cpu")
    # which GPU every rank ran on, gathered over the same process group the timing used (the line reports what was launched, not what was asked for)
    rank_devices = [local]
    if use_dist:
        mine = torch.tensor([rank, local, torch.cuda.current_device()], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        rank_devices = [int(g[2].item()) for g in sorted(got, key=lambda g: int(g[0].item()))]
    prof = {}
    parts = [ln["ext"].profile_get("fast") for ln in lanes]
    prof["fast"] = (sum(p[0] for p in parts), sum(p[1] for p in parts))
    # the other stages' event spans from two further, untimed steps with events around everything (not part of `value`)
    for ln in lanes:
        ln["ext"].profiling(True)
        ln["ext"].profile_reset()
    for i in range(min(2, a.steps)):
        step(a.warmup + a.steps + i)
    barrier()
    for ln in lanes:
        ln["ext"].profiling(False)
    for k in ("pyramid", "octree", "assemble", "blur", "orient_desc"):
        parts = [ln["ext"].profile_get(k) for ln in lanes]
        prof[k] = (sum(p[0] for p in parts), sum(p[1] for p in parts))
    last = lanes[((a.warmup + a.steps) * chunks - 1) % nl]
    valid = torch.arange(last["matches"].shape[1], device="cuda")[None, :] < last["nq"][:, None]     # rows beyond a pair's query count are stale
    nmatched = int(((last["matches"][:, :, 1] <= 100) & valid).sum().item())  # TH_HIGH gate, sanity only
    ext, grid = lanes[0]["ext"], lanes[0]["grid"]
    # k_fast_cells alone on the chip (one lane, nothing overlapping it): the kernel-quality figure next to the one the
    # pipelined timed region yields; same live HIP-event measurement, 8 further steps, not part of `value`
    excl = None
    if nl > 1 and not a.no_exclusive:
        ext.profiling(True); ext.profile_reset()
        for i in range(8):
            ext.extract_batch_device(d_frames.data_ptr() + (i % nbatches) * B * H * W, B, H, W)
        ext.sync()
        ext.profiling(False)
        excl = ext.profile_get("fast")

    # BASELINE config 5: the 500-keyframe global BA landmark-sharded over all ranks (dvm_slam_amd/sharded_ba.py), reported beside
    # the one-GPU number of the `ba` leg.  Every rank takes part; a watchdog bounds the damage if a collective wedges.
    sharded_rec = None
    if use_dist and not a.no_ba:
        import threading
        box = {}

        def _sharded():
            try:
                import ba_bench
                box["rec"] = ba_bench.run_sharded(local, a.ba_iters)
                # "replicas": all ranks solve their own copy at once; the sum is what N independent global BAs deliver
                dist.barrier()
                mine = ba_bench.run_replica(local, a.ba_iters)
                t = torch.tensor([mine], dtype=torch.float64, device=f"cuda:{local}" if backend == "nccl" else "cpu")
                dist.all_reduce(t)
                box["rec"]["replicas"] = {"value": float(t.item()), "unit": "iterations/s (sum over ranks)", "ranks": world,
                                          "this_rank": mine}
            except Exception as ex:   # noqa: BLE001
                box["rec"] = {"error": repr(ex)}

        th = threading.Thread(target=_sharded, daemon=True)
        th.start()
        th.join(timeout=180.0)
        sharded_rec = box.get("rec", {"error": "timed out after 180 s"})
        hard_exit = th.is_alive()
    else:
        hard_exit = False
    if rank == 0:
        total_frames = world * a.steps * chunks * B
        fast_ms, fast_n = prof["fast"]
        roof = None
        if fast_n:
            per_launch_s = fast_ms / fast_n / 1e3
            frames_per_launch = a.steps * chunks * B / fast_n          # = B (one k_fast_cells launch per 256-frame chunk)
            ach = BYTES_PER_FRAME_FAST * frames_per_launch / per_launch_s / 1e9
            traffic = None
            pmc, pmc_file = None, None
            for cand in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):   # the newest committed rocprofv3 PMC fold
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", cand)))
                    pmc_file = cand
                    break
                except Exception:
                    pmc = None
            try:   # FAST is two launches per launch group since round 6 (k_fast_cells1: one wave per cell; k_fast_cells: the taller cells): one entry
                k1, k0 = pmc["kernels"].get("dvm::k_fast_cells1"), pmc["kernels"].get("dvm::k_fast_cells")
                if k1 and k0:
                    m = dict(k0)
                    for f in ("hbm_bytes_per_launch", "hbm_bytes_per_launch_raw", "valu_wave_instr_per_launch", "profiled_duration_us"):
                        if f in k0 and f in k1:
                            m[f] = k0[f] + k1[f]
                    for f, w in (("valu_busy_frac_counter", "profiled_duration_us"), ("mean_issue_cycles_per_valu_instr", "valu_wave_instr_per_launch")):
                        if f in k0 and f in k1 and w in k0 and w in k1:
                            m[f] = (k0[f] * k0[w] + k1[f] * k1[w]) / (k0[w] + k1[w])
                    pmc["kernels"]["dvm::k_fast_cells"] = m
            except Exception:   # noqa: BLE001
                pass
            try:  # HBM bytes per launch from the committed PMC passes (same launch-group size only)
                if pmc["batch"] == frames_per_launch:
                    traffic = pmc["kernels"]["dvm::k_fast_cells"]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
            # which build the quoted counters were taken on (tools/kernels_sha.py, stamped into the fold when it is collected) against the build
            # that is running: `stale` says the fold describes OTHER kernels than the ones just timed
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from kernels_sha import kernels_sha
                sha_now = kernels_sha()
            except Exception:   # noqa: BLE001
                sha_now = None
            pmc_src = {"file": f"profiles/{pmc_file}" if pmc_file else None, "kernels_sha_of_fold": (pmc or {}).get("kernels_sha"), "kernels_sha_running": sha_now,
                       "stale": None if not pmc or not (pmc or {}).get("kernels_sha") or not sha_now else bool(pmc["kernels_sha"] != sha_now)}
            valu = None
            try:  # VALU issue time of one 256-frame launch group from the committed SQ_INSTS_VALU pass
                if pmc["batch"] == frames_per_launch:
                    per_chunk = {"dvm::k_pyr_level0": 1, "dvm::k_pyr_resize": 7, "dvm::k_fast_cells": 1, "dvm::k_blur7": 1, "dvm::k_octree": 1,
                                 "dvm::k_assemble": 1, "dvm::k_orient_desc": 1, "dvm::k_frame_build": 1, "dvm::k_match_window": 1}
                    wi = sum(pmc["kernels"][k]["valu_wave_instr_per_launch"] * n for k, n in per_chunk.items())
                    chunk_ms = dt / (a.steps * chunks) * 1e3
                    # ONE number: every kernel's dynamic wave-instruction count (SQ_INSTS_VALU) x the mean issue cost of ITS static
                    # instruction mix by measured cost class (tools/valu_mix.py: 2.25 / 4.15 / 4.6 / 8.2 / 16.3 cycles)
                    cyc = sum(pmc["kernels"][k]["valu_wave_instr_per_launch"] * pmc["kernels"][k].get("mean_issue_cycles_per_valu_instr", 4.15) * n
                              for k, n in per_chunk.items())
                    issue_ms = cyc / (1024 * 2.4e9) * 1e3
                    ms2, ms4 = wi * 2.25 / (1024 * 2.4e9) * 1e3, wi * 4.15 / (1024 * 2.4e9) * 1e3
                    # the COUNTER: SQ_ACTIVE_INST_VALU (quad-cycles of VALU execution, chip-wide) x 4 / (1024 SIMDs x the launch's cycles), per kernel of the
                    # profiled pass, weighted by the kernels' launches in a launch group
                    busy = None
                    try:
                        num = sum(pmc["kernels"][k]["valu_busy_frac_counter"] * pmc["kernels"][k]["profiled_duration_us"] * n for k, n in per_chunk.items())
                        den = sum(pmc["kernels"][k]["profiled_duration_us"] * n for k, n in per_chunk.items())
                        busy = {"frac_of_kernel_time": num / den, "k_fast_cells": pmc["kernels"]["dvm::k_fast_cells"]["valu_busy_frac_counter"],
                                "per_kernel": {k.replace("dvm::", ""): pmc["kernels"][k]["valu_busy_frac_counter"] for k in per_chunk},
                                "note": "counter-based: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x launch cycles) of the profiled pass (kernel-trace duration x the clock GRBM_GUI_ACTIVE gives)"}
                    except Exception:   # noqa: BLE001
                        busy = None
                    valu = {"kind": "model (static instruction mix x dynamic SQ_INSTS_VALU)", "counter_valu_busy": busy,
                            "frac": issue_ms / chunk_ms, "issue_ms_per_launch_group": issue_ms, "launch_group_ms": chunk_ms,
                            "wave_instr_per_launch_group": wi, "mean_issue_cycles_per_wave_instr": cyc / wi,
                            "per_kernel_issue_ms": {k.replace("dvm::", ""): pmc["kernels"][k]["valu_wave_instr_per_launch"] * pmc["kernels"][k].get("mean_issue_cycles_per_valu_instr", 4.15) * n
                                                    / (1024 * 2.4e9) * 1e3 for k, n in per_chunk.items()},
                            "bounds_if_all_full_rate_or_all_half_rate": [ms2 / chunk_ms, ms4 / chunk_ms], "source": f"profiles/{pmc_file} + its valu_mix",
                            "note": "VALU issue time of one 256-frame launch group / its wall time.  Issue time = sum over kernels of SQ_INSTS_VALU (dynamic, "
                                    "rocprofv3 --pmc) x the mean cycles per wave-instruction of the kernel's static instruction mix, each mnemonic priced by the "
                                    "issue cost measured on this GPU at 8 waves per SIMD (profiles/r02_valu_issue*.jsonl: 2.25 full-rate integer / f32 add-mul, 4.15 "
                                    "half-rate VOP3 / packed / dot / mad / conversions, 8.2 min3 / max3_u16), over 1024 SIMDs x 2.4 GHz.  Static mix, dynamic "
                                    "count: loop bodies are not weighted."}
            except Exception:
                valu = None
            # the whole launch group's counter traffic against its algorithmic bytes (SURVEY 8d's numerator: pyramid build + FAST read + blur
            # read / write; the per-keypoint patch gathers of orientation / descriptors and level 0's bordered copy are NOT in it, they are
            # what the ratio shows: the blurred and the raw pyramid are read once more by k_orient_desc, whose patches cover every level)
            pipe_ratio = None
            try:
                if pmc["batch"] == frames_per_launch:
                    per_chunk_all = {"dvm::k_pyr_level0": 1, "dvm::k_pyr_resize": 7, "dvm::k_fast_cells": 1, "dvm::k_blur7": 1, "dvm::k_octree": 1,
                                     "dvm::k_assemble": 1, "dvm::k_orient_desc": 1, "dvm::k_frame_build": 1, "dvm::k_match_window": 1}
                    tot = sum(pmc["kernels"][k]["hbm_bytes_per_launch"] * n for k, n in per_chunk_all.items())
                    pipe_ratio = tot / (BYTES_PER_FRAME_TOTAL * frames_per_launch)
            except Exception:   # noqa: BLE001
                pipe_ratio = None
            roof = {"bound": "hbm", "kernel": "k_fast_cells1 + k_fast_cells (FAST of one launch group: two launches)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "pipeline_traffic_over_algorithmic": pipe_ratio,
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": pmc_src,
                    "bytes_per_launch": BYTES_PER_FRAME_FAST * frames_per_launch, "avg_launch_ms": fast_ms / fast_n,
                    "frames_per_launch": frames_per_launch,
                    "all_stage_bytes_per_frame": BYTES_PER_FRAME_TOTAL,
                    "pipeline_achieved": BYTES_PER_FRAME_TOTAL * total_frames / world / dt / 1e9,   # whole step, GB/s per GPU
                    "pipeline_frac": BYTES_PER_FRAME_TOTAL * total_frames / world / dt / 1e9 / HBM_PEAK_GBS,
                    "gpu_kernel_event_ms_per_launch": {k: (v[0] / v[1] if v[1] else None) for k, v in prof.items()},
                    "gpu_kernel_event_note": "k_fast_cells: live in the timed region; the other stages: two untimed steps behind it with events around "
                                             "every stage.  HIP-event span of each stage's launches on its own stream; the two lanes and the low-priority "
                                             "blur stream overlap, so these are NOT additive work figures (blur alone spans most of a launch group)",
                    "valu_issue": valu,
                    "note": f"{nl} pipeline lanes: k_fast_cells launches of one batch overlap the tail kernels of the previous batch, so the "
                            "per-launch duration above includes that contention" if nl > 1 else None}
            if excl and excl[1]:
                e_ms = excl[0] / excl[1]
                roof["exclusive"] = {"avg_launch_ms": e_ms, "achieved": BYTES_PER_FRAME_FAST * B / (e_ms / 1e3) / 1e9,
                                     "frac": BYTES_PER_FRAME_FAST * B / (e_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                                     "note": "same kernel, same batch, one lane only (nothing else on the chip), 8 launches"}
        out = {
            "metric": "frames/sec ORB extract+match 640x480x8lvl", "value": total_frames / dt, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "single agent tracking front end: 640x480 8-level ORB extract (1000 features) + "
                                   "frame-to-frame windowed Hamming match, one agent per GPU",
                       "frames_per_step_per_gpu": chunks * B, "frames_per_launch_group": B, "launch_groups_per_step": chunks,
                       "resident_stream_frames": nstream, "nfeatures": 1000, "nlevels": 8, "scale_factor": 1.2,
                       "ini_th_fast": 20, "min_th_fast": 7, "match": "SearchByProjection(Cur,Last) window th=15",
                       "parallelism": f"agents{world}", "pipeline_lanes": nl, "stream_texture": a.texture,
                       "ranks": {"count": world, "backend": ("rccl" if backend == "nccl" else backend) if use_dist else "none (single process)",
                                 "cuda_device_of_rank": rank_devices, "launched_by": "bench.py itself (--gpus N without a launcher)"
                                 if os.environ.get("DVM_BENCH_RELAUNCHED") == "1" else ("torch.distributed.run" if "WORLD_SIZE" in os.environ else "python")}},
            "roofline": roof, "sanity_matches_le_TH_HIGH_last_step": nmatched,
        }
        if world == 1 and not a.no_pcie:
            for ln in lanes:             # a handle owns two streams; more than four per process share hardware queues
                ln["ext"].close(); ln["grid"].close()
            out["pcie_inclusive"] = pcie_inclusive_leg(capi, frames, B, max(16, a.steps * chunks // 8), local)
        # BA leg first, while the GPU is still in its operating power state from the extract leg: the LM loop alone
        # (host-synchronised, mostly single-workgroup kernels) does not lift an idle MI355X off its 584 MHz idle clock
        # (measured: 315 it/s cold vs 1050 it/s hot); the CPU baseline below leaves the GPU idle for ~12 s.
        if not a.no_ba:
            try:
                import ba_bench
                out["ba"] = ba_bench.run(local, a.ba_iters, cpu_seconds=6.0 if a.cpu_seconds > 0 else 0.0)
                if world == 1:
                    try:       # second workload of the BA metric: the same size with loop-closure bands and long-range observations
                        out["ba"]["loop_closed"] = ba_bench.run_loop_closed(local, a.ba_iters, cpu_iters=2 if a.cpu_seconds > 0 else 0)
                    except Exception as ex:   # noqa: BLE001
                        out["ba"]["loop_closed"] = {"error": repr(ex)}
            except ImportError:
                out["ba"] = None
        if sharded_rec is not None:
            out["ba_sharded"] = sharded_rec
        if world == 1 and a.legs and not a.no_legs:
            # what BASELINE configs 2 and 3 cost per call through the drop-in boundary (bench_legs.py), CPU oracle beside each
            import bench_legs
            cpu = a.cpu_seconds > 0
            for name, fn in (("batch_sweep", lambda: bench_legs.batch_sweep()),
                             ("low_texture", lambda: bench_legs.low_texture(capi, local, cpu=cpu)),
                             ("latency", lambda: bench_legs.latency(capi, frames[:64], local, cpu_calls=24 if cpu else 0)),
                             ("online_agents", lambda: bench_legs.online_agents(capi, frames[:16], local)),
                             ("lba", lambda: bench_legs.lba(local, cpu_seconds=4.0 if cpu else 0.0)),
                             ("lba_batch", lambda: bench_legs.lba_batch(local, cpu_windows=4 if cpu else 0)),
                             ("merge", lambda: bench_legs.merge(local, cpu_reps=3 if cpu else 0)),
                             ("ba_cold", lambda: bench_legs.ba_cold(local))):
                try:
                    out[name] = fn()
                except Exception as ex:   # noqa: BLE001 -- a leg that fails must not take the contract line with it
                    out[name] = {"error": repr(ex)}
        if world == 1 and not a.no_config_legs:
            # configs 2 / 3 in every default run (short forms of the --legs records, no CPU legs): per-call latencies of the tracked frame,
            # the local-BA window call and the merge chain through the drop-in boundary -> `config_legs` of the contract line
            import bench_legs
            for name, fn in (("latency", lambda: bench_legs.latency(capi, frames[:64], local, calls=200, cpu_calls=0)),
                             ("lba", lambda: bench_legs.lba(local, repeats=12, cpu_seconds=0.0)),
                             ("merge", lambda: bench_legs.merge(local, reps=8, cpu_reps=0)),
                             ("lba_fast", lambda: bench_legs.lba_fast(local))):
                if name in out and "error" not in out[name]:
                    continue
                try:
                    out[name] = fn()
                except Exception as ex:   # noqa: BLE001
                    out[name] = {"error": repr(ex)}
        if a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(frames, a.cpu_seconds)
        flush_c_stdio()
        emit(out, a)
    if hard_exit:          # a wedged collective: the line is out, do not wait for the process group
        sys.stdout.flush()
        os._exit(0)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
