"""tools/h2d_bw.py -- raw pinned-host -> HBM copy rate of the box (the floor of the PCIe-inclusive bench leg)."""
import time

import torch

x = torch.empty(78_643_200, dtype=torch.uint8).pin_memory()
d = torch.empty_like(x, device="cuda")
for _ in range(3):
    d.copy_(x, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    d.copy_(x, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"pinned H2D {x.numel() / 1e6:.1f} MB: {dt * 1e3:.3f} ms = {x.numel() / dt / 1e9:.1f} GB/s")
