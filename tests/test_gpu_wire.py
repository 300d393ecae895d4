"""GPU side of the DVMW sender / receiver: keypoints + descriptors gathered on the device from the
extractor's result arrays must give byte for byte the block the host assembles from the downloaded results, and the
receiver builds its frame grid straight from the block's sections in HBM."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_device_gather_equals_host_block(capi, oracle, frames):
    from dvm_slam_amd import wire
    B = len(frames)
    e = capi.OrbExtractor(max_batch=B)
    e.extract_batch_host(np.stack(frames))
    e.sync()
    host = [e.download(f) for f in range(B)]       # (n, kps, desc, mono)
    rng = np.random.default_rng(0)
    kfs_host, kfs_head = [], []
    for f, (n, kps, desc, _) in enumerate(host):
        base = dict(uuid=rng.integers(0, 256, 16, dtype=np.uint8), mn_id=f, frame_id=f, fx=149.0, fy=149.0, cx=320.0, cy=240.0, min_x=0.0,
                    max_x=640.0, min_y=0.0, max_y=480.0, scale_factor=1.2, n_levels=8, creator_agent=0)
        kfs_host.append(dict(base, kps=kps[:n], desc=desc[:n]))
        kfs_head.append(dict(base, n_kp=n))
    want = wire.build(kfs_host, [], sender_agent=0)
    head = wire.build(kfs_head, [], sender_agent=0, head_only=True)
    d_block = torch.zeros(want.size, dtype=torch.uint8, device="cuda")
    d_block[:head.size] = torch.from_numpy(head).cuda()
    k0, d0, _, cap = e.result_device(0)
    k1, d1, _, _ = e.result_device(1)
    torch.cuda.synchronize()
    wire.gather_keypoints_device(d_block.data_ptr(), 0, B, k0, (k1 - k0) // 28, d0, d1 - d0)
    torch.cuda.synchronize()
    got = d_block.cpu().numpy()
    assert np.array_equal(got, want)
    wire.validate(got)
    # a sub-range of keyframes only
    d_block2 = torch.zeros_like(d_block); d_block2[:head.size] = torch.from_numpy(head).cuda()
    wire.gather_keypoints_device(d_block2.data_ptr(), 1, 2, k1, (k1 - k0) // 28, d1, d1 - d0)
    torch.cuda.synchronize()
    S_all, S_part = wire.sections(got), wire.sections(d_block2.cpu().numpy())
    rec = S_all[1]
    a, b = int(rec[1]["kp_off"]), int(rec[2]["kp_off"]) + int(rec[2]["n_kp"])
    assert np.array_equal(S_part[3][a:b], S_all[3][a:b]) and np.array_equal(S_part[4][a:b], S_all[4][a:b])
    assert not S_part[3][:a]["x"].any() and not S_part[4][b:].any()

    # receiver: grid + window search straight from the block in HBM == from the sender's host arrays
    off, _, _ = wire.layout(got[:64].view(wire.HEADER))
    n1 = int(rec[1]["n_kp"])
    g_dev = capi.FrameGrid(2048); g_host = capi.FrameGrid(2048)
    L = capi.lib()
    import ctypes as C
    capi.check(L.dvm_frame_build(g_dev.h, 0, C.c_void_p(d_block.data_ptr() + off[3] + 28 * a), C.c_void_p(d_block.data_ptr() + off[4] + 32 * a), n1, None,
                                 0.0, 640.0, 0.0, 480.0, 1, None))
    torch.cuda.synchronize()
    g_host.build(host[1][1][:n1], host[1][2][:n1])
    q = host[0]
    nq = q[0]
    args = (q[2][:nq], q[1]["x"][:nq], q[1]["y"][:nq], np.full(nq, 15.0, np.float32), q[1]["octave"][:nq] - 1, q[1]["octave"][:nq] + 1)
    m_dev = g_dev.match_window(*args); m_host = g_host.match_window(*args)
    assert np.array_equal(m_dev, m_host) and (m_host["best_idx"] >= 0).sum() > 100
    e.close(); g_dev.close(); g_host.close()
