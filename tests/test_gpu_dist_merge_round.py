"""BASELINE.json config 4 on the HIP path: two agent processes (world_size 2, gloo for the exchange, both on the test box's one GPU)
run the decentralised merge round of dvm_slam_amd/agents.py with the device operators (merge.GpuOps: vocabulary transform, keyframe
database query, SearchByBoW, Sim3 hypotheses, OptimizeSim3, SearchBySim3 -- all through libdvmslam_hip / libdvmslam_host) and, in the same
processes, with the CPU-oracle operators: recognised keyframes, shipped blocks, BoW matches, inlier counts, the solved similarity and
the announcement every rank hears must agree.  (The N > 1 RCCL form of the same collectives is the driver's multi-GPU run.)"""
import pytest

from test_dist_merge_round import run_round

pytestmark = pytest.mark.gpu


def test_two_agent_merge_round_hip_operators():
    run_round(gpu=True)
