import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from dvm_slam_amd import capi, synth
frames = synth.frame_stream(256)
d = torch.from_numpy(frames).cuda()
ext = capi.OrbExtractor(max_batch=256)
ext.extract_batch_device(d.data_ptr(), 256, 480, 640); ext.sync()
out = (C.c_ulonglong * 16)()
L = capi.lib()
L.dvm_debug_fast_stamps(out, 1)
for _ in range(4):
    ext.extract_batch_device(d.data_ptr(), 256, 480, 640)
ext.sync()
L.dvm_debug_fast_stamps(out, 0)
v = np.array(out[:8], dtype=np.float64)
n = v[0]
names = ["(cells)", "tile load + zero + sync", "A pre-test", "B strength", "sync", "C nms", "compaction + out"]
tot = v[1:7].sum()
for i in range(1, 7): print(f"{names[i]:26s} {v[i]/n:9.0f} cycles/cell-frame  {100*v[i]/tot:5.1f} %")
print("cells", n, "sum cycles/cell", tot / n)
