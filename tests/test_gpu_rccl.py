"""The RCCL (torch.distributed backend "nccl") branches of the multi-GPU path, executed on the one GPU the test box has
("the nccl branch has never executed anywhere").  RCCL does not admit two ranks on one device, so the job has a
single member -- but it is a real communicator on the device, and the code that runs is the code N ranks run: exchange.py's
collectives on device tensors, the sharded BA's tile all-reduce enqueued on the solver's HIP stream, and bench.py's
distributed control flow (barriers, max-over-ranks, the sharded-BA leg)."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    return env


def test_exchange_and_sharded_ba_over_rccl():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rec.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), out], capture_output=True, text=True,
                           timeout=600, env=_env(29655))
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
        rec = json.load(open(out))
    assert rec["backend"] == "nccl"
    assert rec["max_over_ranks"] == 1.5 and rec["varlen_ok"] and rec["blocks_ok"] and rec["sim3_ok"]
    # one rank's "sum over ranks" is the identity: the sharded solver must reproduce the unsharded one bit for bit, having
    # gone through RCCL once per LM trial (+ the start state's chi2, lambda init, the final landmark exchange)
    assert rec["ba_trials_equal"] and rec["ba_bits_equal"]
    assert rec["ba_calls"] == rec["ba_expected_calls"] and rec["ba_bytes"] > 0


def test_bench_distributed_control_flow_over_rccl(tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--chunks-per-step", "2",
           "--stream-frames", "256", "--cpu-seconds", "0", "--no-pcie", "--no-exclusive", "--ba-iters", "3", "--full-json", str(tmp_path / "full.json")]
    env = _env(29656)
    env["DVM_BENCH_FORCE_DIST"] = "1"
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = r.stdout.strip().splitlines()
    assert lines[-1].startswith("{"), "the JSON line must be the last thing on stdout (RCCL's banner flushed before it)"
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["ba_sharded"]["value"] > 0 and len(lines[-1]) < 6000
    line = json.load(open(tmp_path / "full.json"))          # the complete record (the contract line carries a summary)
    sh = line["ba_sharded"]
    assert "error" not in sh, sh
    assert sh["backend"].startswith("nccl") and sh["ranks"] == 1 and sh["value"] > 0
    # same problem, same solver: the single-rank sharded run ends on the unsharded run's chi2
    assert abs(sh["chi2_final"] - line["ba"]["chi2_final"]) <= 1e-9 * line["ba"]["chi2_final"]
