"""bench.py on the GPU box: the JSON contract (one line, required keys, roofline / cpu_baseline objects) and the equality
of the pipelined (two lanes) and the plain (one lane) schedule on the same frames."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--batch", "64", "--stream-frames", "64",
                          "--no-ba", *extra], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_contract_and_lane_equivalence():
    d2 = _run("--cpu-seconds", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d2, k
    assert d2["n_gpus"] == 1 and d2["steps"] == 4 and d2["warmup"] == 2 and d2["vs_baseline"] is None and d2["dtype"] == "u8"
    assert "workload" in d2["config"] and d2["config"]["pipeline_lanes"] == 2
    r = d2["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert r["exclusive"]["avg_launch_ms"] > 0
    c = d2["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert d2["pcie_inclusive"]["value"] > 0
    d1 = _run("--cpu-seconds", "0", "--lanes", "1", "--no-pcie")
    assert d1["config"]["pipeline_lanes"] == 1
    assert d1["sanity_matches_le_TH_HIGH_last_step"] == d2["sanity_matches_le_TH_HIGH_last_step"] > 1000
