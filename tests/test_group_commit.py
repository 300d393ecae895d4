"""CPU test of the batching protocol behind dvm_orb_pool / dvm_match_pool / dvm_pose_pool (dvm_slam_amd/csrc/group_commit.h): a small
C++ driver (tests/group_commit/gc_driver.cpp, plain g++, no HIP) runs it with many threads and jobs of two shapes."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("gc") / "gc_driver")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "group_commit", "gc_driver.cpp"),
                           "-I" + os.path.join(ROOT, "dvm_slam_amd", "csrc"), "-o", exe])
    return exe


@pytest.mark.parametrize("threads,jobs,max_batch,window_us", [(8, 300, 8, 50), (16, 150, 4, 20), (3, 400, 32, 100), (1, 200, 8, 50)])
def test_group_commit_protocol(driver, threads, jobs, max_batch, window_us):
    r = subprocess.run([driver, str(threads), str(jobs), str(max_batch), str(window_us)], capture_output=True, text=True, timeout=120)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and out["ok"], out
    assert out["jobs"] == threads * jobs and out["wrong"] == 0 and out["mixed"] == 0
    assert out["max_batch_seen"] <= max_batch
    if threads > 1:
        assert out["batches"] < out["jobs"], "calls never shared a batch"
