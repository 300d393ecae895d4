"""libdvmslam_rccl.so (include/dvmslam_rccl.h): the inter-agent exchange of a C++ agent node over RCCL -- the reference's node is C++
(src/slam_system/src/orb_slam3_wrapper.cpp:359-370 serialized keyframe payloads, :524-528 their reception, :920-949 the Sim3 / flag
broadcasts) -- and the stock all-reduce callback of the landmark-sharded global BA.  One real rank (the box has one GPU; RCCL refuses
two ranks on one device), in a process of its own: communicator from the library's helpers, collectives on device buffers, the sharded
solver calling the native callback once per LM trial and ending on the unsharded solver's bits."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_exchange_and_sharded_ba_allreduce():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rec.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_native_worker.py"), out], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
        rec = json.load(open(out))
    assert rec["rank"] == 0 and rec["world"] == 1
    assert rec["blocks_ok"] and rec["varlen_ok"] and rec["sim3_ok"] and rec["max_over_ranks"] == 1.5
    assert rec["varlen_too_small"] == -3
    assert rec["ba_trials_equal"] and rec["ba_bits_equal"] and rec["ba_allreduce_doubles"] > 1000
    assert rec["class_native_bits_equal"] and rec["class_python_calls"] == 0
