"""N>1 path on CPU: world_size-2 gloo run of the inter-agent exchange and of bench.py's sharding /
max-over-ranks timing logic (the GPU kernels themselves need no collective)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dvm_slam_amd import exchange
    from oracle import pyoracle as po
    cap = 64
    rng = np.random.default_rng(100 + rank)
    n = 40 + rank * 7                                   # ragged keyframe sizes
    kps = np.zeros(n, po.KP_DTYPE)
    kps["x"] = rng.uniform(0, 640, n).astype(np.float32)
    kps["y"] = rng.uniform(0, 480, n).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pose = np.arange(7, dtype=np.float32) + rank
    blk = exchange.pack_keyframe(bytes([rank] * 16), rank, pose, kps, desc, cap)
    got = exchange.all_gather_blocks(blk)
    ok = len(got) == world
    for r, b in enumerate(got):
        uuid, agent, p, k, d = exchange.unpack_keyframe(b, cap, po.KP_DTYPE)
        rr = np.random.default_rng(100 + r)
        nn = 40 + r * 7
        ok &= uuid == bytes([r] * 16) and agent == r and len(k) == nn and np.array_equal(p, np.arange(7, dtype=np.float32) + r)
        ok &= np.array_equal(k["x"], rr.uniform(0, 640, nn).astype(np.float32))
    # cross-agent place-recognition primitive: my descriptors vs every peer's (oracle stands in for the HIP kernel on CPU)
    peer = exchange.unpack_keyframe(got[1 - rank], cap, po.KP_DTYPE)
    D = po.hamming_matrix(desc, peer[4])
    ok &= D.shape == (n, 40 + (1 - rank) * 7)
    # DVMW map deltas of different sizes: every agent receives and decodes every agent's block
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from dvm_slam_amd import capi, wire
    from wire_scene import assert_equal_delta, make_delta
    kfs, mps = make_delta(wire, capi, 50 + rank, n_kf=2 + rank, n_mp=10 + 20 * rank, agent=rank)
    mine = torch.from_numpy(wire.build(kfs, mps, sender_agent=rank))
    blocks = exchange.all_gather_varlen(mine)
    ok &= len(blocks) == world and len({b.numel() for b in blocks}) == world
    for r, b in enumerate(blocks):
        ek, em = make_delta(wire, capi, 50 + r, n_kf=2 + r, n_mp=10 + 20 * r, agent=r)
        parsed = wire.parse(b.numpy())
        ok &= int(parsed[0]["sender_agent"]) == r
        assert_equal_delta(wire, ek, em, parsed)
    sim3 = torch.arange(8, dtype=torch.float64) * (1.0 if rank == 1 else 0.0)
    exchange.broadcast_sim3(sim3, src=1)
    ok &= bool((sim3 == torch.arange(8, dtype=torch.float64)).all())
    t = exchange.max_over_ranks(1.0 + rank)
    ok &= t == float(world)
    ok &= exchange.agent_stream_segment(rank, 128) == 128 * rank
    q.put((rank, bool(ok), D.sum()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_agents_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    # Hamming matrix of (A vs B) and (B vs A) are transposes: same total
    assert res[0][2] == res[1][2]


def test_single_process_degenerates():
    from dvm_slam_amd import exchange
    b = torch.zeros(exchange.block_bytes(4), dtype=torch.uint8)
    assert exchange.all_gather_blocks(b)[0] is b
    assert exchange.max_over_ranks(2.5) == 2.5
