"""bench.py on the GPU box: the JSON contract (one line, required keys, roofline / cpu_baseline objects) and the equality
of the pipelined (two lanes) and the plain (one lane) schedule on the same frames."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _contract_line(stdout):
    """The driver's view: the LAST stdout line is the contract object, compact enough to survive an 8 KB tail."""
    lines = stdout.rstrip("\n").splitlines()
    assert lines and lines[-1].startswith("{"), stdout[-2000:]
    assert len(lines[-1]) < 6000, len(lines[-1])
    d = json.loads(lines[-1])
    assert "leg" not in d
    return d, [json.loads(l) for l in lines[:-1] if l.startswith("{")]


def _run(*extra, tmp):
    full = os.path.join(str(tmp), "full.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--batch", "64", "--stream-frames", "128",
                          "--chunks-per-step", "3", "--full-json", full, *extra], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d, before = _contract_line(out.stdout)
    assert before == [], out.stdout[-2000:]          # no --legs: the contract line is the only JSON on stdout
    return d, json.load(open(full))


def test_bench_contract_and_lane_equivalence(tmp_path):
    d2, full = _run("--cpu-seconds", "2", "--ba-iters", "3", tmp=tmp_path)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d2, k
    assert d2["n_gpus"] == 1 and d2["steps"] == 4 and d2["warmup"] == 2 and d2["vs_baseline"] is None and d2["dtype"] == "u8"
    assert "workload" in d2["config"] and d2["config"]["pipeline_lanes"] == 2
    assert d2["config"]["frames_per_step_per_gpu"] == 3 * 64 and abs(d2["value"] - 4 * 3 * 64 / (d2["ms_per_step"] * 4 / 1e3)) < 1e-6 * d2["value"]
    r = d2["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert r["exclusive"]["avg_launch_ms"] > 0
    c = d2["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert c["one_agent_per_core"]["value"] > 0
    assert d2["pcie_inclusive"]["value"] > 0
    # the BA half of the metric rides on the same line as a summary: ring map and loop-closed map, roofline of the reduced solve
    ba = d2["ba"]
    assert ba["value"] > 0 and ba["dtype"] == "f64" and 0 < ba["roofline"]["frac"] < 1 and ba["cpu_baseline"]["value"] > 0
    assert ba["parity_vs_cpu"]["trials_equal"] and ba["parity_vs_cpu"]["max_abs_pose"] < 1e-6
    assert ba["loop_closed"]["value"] > 0, ba["loop_closed"]
    # configs 2 / 3 ride along as per-call figures (the short legs of the default run)
    cl = d2["config_legs"]
    for k in ("extract", "search_by_projection", "pose_optimization", "track_frame_one_chain", "lba_window_call", "lba_32_windows_one_launch_iterations_per_s", "merge_chain"):
        assert cl[k] > 0, (k, cl)
    # the complete record keeps what the line drops
    assert full["value"] == d2["value"] and "gpu_kernel_event_ms_per_launch" in full["roofline"] and "schedule" in full["ba"]["roofline"]
    d1, _ = _run("--cpu-seconds", "0", "--lanes", "1", "--no-pcie", "--no-ba", "--no-config-legs", tmp=tmp_path)
    assert d1["config"]["pipeline_lanes"] == 1
    assert d1["sanity_matches_le_TH_HIGH_last_step"] == d2["sanity_matches_le_TH_HIGH_last_step"] > 1000


def test_bench_two_ranks_control_flow(tmp_path):
    """The N > 1 path of bench.py (rendezvous, barriers, max-over-ranks timing, rank-0-only reporting) with two ranks that
    share the one GPU of the test box over gloo (DVM_BENCH_SHARE_GPU / DVM_BENCH_BACKEND are test hooks; the driver launches
    the same file with one rank per GPU over RCCL)."""
    env = dict(os.environ, DVM_BENCH_SHARE_GPU="1", DVM_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29577",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32", "--stream-frames", "64",
           "--chunks-per-step", "2", "--ba-iters", "3", "--cpu-seconds", "0", "--full-json", str(tmp_path / "full.json")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d, before = _contract_line(out.stdout)
    assert before == [], out.stdout[-2000:]              # rank 0 only, one line
    assert d["ba_sharded"]["ranks"] == 2 and d["ba_sharded"]["value"] > 0 and d["ba_sharded"]["replicas"]["ranks"] == 2
    d = json.load(open(tmp_path / "full.json"))
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "agents2"
    assert abs(d["value"] - 2 * 3 * 2 * 32 / (d["ms_per_step"] * 3 / 1e3)) < 1e-6 * d["value"]   # whole-job frames over the max-over-ranks time
    assert "pcie_inclusive" not in d                     # N = 1 only
    # config 5 beside the one-GPU BA: the landmark-sharded solve over both ranks (gloo here, RCCL under the driver)
    sh, one = d["ba_sharded"], d["ba"]
    assert "error" not in sh, sh
    assert sh["ranks"] == 2 and sh["value"] > 0 and sh["allreduce_bytes_per_run"] > 1e6
    assert abs(sh["chi2_final"] - one["chi2_final"]) <= 1e-9 * one["chi2_final"]
    assert one["roofline"]["frac"] > 0 and one["roofline"]["schedule"]["nz_tiles"] > 100
    # "replicas" (SURVEY 8e): both ranks solve their own copy at once; the GBA form (no Huber kernel) beside the robust one
    rep = sh["replicas"]
    assert rep["ranks"] == 2 and rep["value"] >= rep["this_rank"] > 0
    assert one["huber_off"]["value"] > 0 and one["huber_off"]["iterations"] > 0


def test_bench_plain_command_honours_gpus():
    """The driver's command shape without a launcher: `python bench.py --gpus 2` must itself start two ranks (here both on the
    one GPU of the box, over gloo) and report n_gpus = 2 -- not one rank on GPU 0 under the label it was asked for."""
    env = dict(os.environ, DVM_BENCH_SHARE_GPU="1", DVM_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "32", "--stream-frames", "64",
           "--chunks-per-step", "2", "--no-ba", "--cpu-seconds", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d, before = _contract_line(out.stdout)
    assert before == [], out.stdout[-2000:]
    assert d["n_gpus"] == 2 and d["config"]["ranks"]["count"] == 2 and len(d["config"]["ranks"]["cuda_device_of_rank"]) == 2
    assert d["config"]["ranks"]["launched_by"].startswith("bench.py itself")
    # a launcher whose rank count disagrees with --gpus is refused, not mislabelled
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "must agree" in (bad.stderr + bad.stdout)
