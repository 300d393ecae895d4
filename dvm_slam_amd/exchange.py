"""Inter-agent exchange over torch.distributed (RCCL on GPUs: backend "nccl"; gloo in CPU tests).

One agent per rank / GPU.  The hot path (extract, match, local BA) has NO collective; the only
traffic is what DVM-SLAM's decentralised map merge already ships between agents:
  C2  new keyframes {uuid, pose, keypoints, 32-B descriptors}  reference src/slam_system/src/orb_slam3_wrapper.cpp:212-384
      (ROS topic robotN/new_key_frames carrying a Boost archive) -> fixed-stride SoA blocks, all_gather
  C4  Sim3 coordinate-frame changes / flags  (:920-949)          -> broadcast of 8 doubles
and the max-over-ranks reduction bench.py needs for timing.  Blocks are plain byte tensors so they
can be gathered straight out of HBM.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

# One rank has nobody to talk to: the collectives below return at once.  Tests on a 1-GPU box clear this to push a single
# rank through RCCL all the same (tests/test_gpu_rccl.py).
SHORTCUT_SINGLE_RANK = True


def _alone() -> bool:
    return not dist.is_initialized() or (dist.get_world_size() == 1 and SHORTCUT_SINGLE_RANK)


KP_BYTES = 28   # dvm_keypoint / cv::KeyPoint
HEADER_BYTES = 64  # uuid[16] | n int32 | agent int32 | pose 7 x f32 | pad


def block_bytes(cap: int) -> int:
    return HEADER_BYTES + cap * (KP_BYTES + 32)


def pack_keyframe(uuid: bytes, agent: int, pose7: np.ndarray, kps: np.ndarray, desc: np.ndarray, cap: int) -> torch.Tensor:
    """Keyframe -> fixed-stride byte block (uint8 tensor of block_bytes(cap))."""
    n = len(kps)
    assert n <= cap and len(uuid) == 16 and kps.dtype.itemsize == KP_BYTES
    buf = np.zeros(block_bytes(cap), np.uint8)
    buf[:16] = np.frombuffer(uuid, np.uint8)
    buf[16:24] = np.array([n, agent], np.int32).view(np.uint8)
    buf[24:52] = np.asarray(pose7, np.float32).view(np.uint8)
    buf[HEADER_BYTES:HEADER_BYTES + n * KP_BYTES] = kps.view(np.uint8).reshape(-1)[: n * KP_BYTES]
    off = HEADER_BYTES + cap * KP_BYTES
    buf[off:off + n * 32] = np.ascontiguousarray(desc, np.uint8).reshape(-1)
    return torch.from_numpy(buf)


def unpack_keyframe(block: torch.Tensor, cap: int, kp_dtype):
    b = block.cpu().numpy()
    uuid = bytes(b[:16])
    n, agent = b[16:24].view(np.int32)
    pose = b[24:52].view(np.float32).copy()
    kps = b[HEADER_BYTES:HEADER_BYTES + int(n) * KP_BYTES].view(kp_dtype).copy()
    off = HEADER_BYTES + cap * KP_BYTES
    desc = b[off:off + int(n) * 32].reshape(int(n), 32).copy()
    return uuid, int(agent), pose, kps, desc


def all_gather_blocks(block: torch.Tensor) -> list[torch.Tensor]:
    """C2: every agent receives every agent's block (same size on all ranks)."""
    if _alone():
        return [block]
    out = [torch.empty_like(block) for _ in range(dist.get_world_size())]
    dist.all_gather(out, block)
    return out


def broadcast_sim3(sim3: torch.Tensor, src: int) -> torch.Tensor:
    """C4: (s, qx,qy,qz,qw, tx,ty,tz) float64 from the merging agent to all."""
    if not _alone():
        dist.broadcast(sim3, src)
    return sim3


def max_over_ranks(seconds: float, device=None) -> float:
    if _alone():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def agent_stream_segment(rank: int, frames_per_agent: int) -> int:
    """First frame index of the camera path segment an agent (rank) processes: agents are independent."""
    return rank * frames_per_agent


def all_gather_varlen(block: torch.Tensor) -> list[torch.Tensor]:
    """C2 with ragged payloads (DVMW blocks, dvm_slam_amd/wire.py): sizes first (one int64 all_gather), then one all_gather of
    blocks padded to the largest; returns every agent's block trimmed to its own size.  Tensors stay on their device."""
    if _alone():
        return [block]
    world = dist.get_world_size()
    n = torch.tensor([block.numel()], dtype=torch.int64, device=block.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    mx = int(max(int(s.item()) for s in sizes))
    padded = torch.zeros(mx, dtype=torch.uint8, device=block.device)
    padded[:block.numel()] = block
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)
    return [o[:int(s.item())] for o, s in zip(out, sizes)]
