# shader-clock stamps of k_octree per level (needs a -DDVM_OCT_DEBUG build of octree_kernel.hip)
import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from dvm_slam_amd import capi, synth
frames = synth.frame_stream(256)
d = torch.from_numpy(frames).cuda()
ext = capi.OrbExtractor(max_batch=256)
ext.extract_batch_device(d.data_ptr(), 256, 480, 640); ext.sync()
out = (C.c_ulonglong * 128)()
L = capi.lib()
L.dvm_debug_oct_stamps(out, 1)
for _ in range(4):
    ext.extract_batch_device(d.data_ptr(), 256, 480, 640)
ext.sync()
L.dvm_debug_oct_stamps(out, 0)
v = np.array(out[:128], dtype=np.float64).reshape(8, 16)
names = {1: "distribute", 2: "roots", 3: "order ph1", 4: "sort ph2", 11: "rank ph2", 5: "child count", 6: "scans+build", 7: "reassign+ne", 8: "final"}
for lv in range(8):
    n = v[lv, 0]
    if n == 0: continue
    tot = sum(v[lv, i] for i in names)
    print(f"level {lv}: wgs {n:.0f} rounds/wg {(v[lv,9]+v[lv,10])/n:.2f} (phase2 {v[lv,10]/n:.2f}) total {tot/n:.0f} cycles/wg")
    for i, nm in names.items(): print(f"   {nm:14s} {v[lv,i]/n:9.0f}  {100*v[lv,i]/tot:5.1f} %")
sp = (C.c_ulonglong * (8 * 256 * 3))()
L.dvm_debug_oct_span(sp)
a = np.array(sp[:], dtype=np.int64).reshape(8, 256, 3)
t0 = a[:, :, 0].min()
st = (a[:, :, 0] - t0) / 100.0; en = (a[:, :, 1] - t0) / 100.0   # us (100 MHz)
print("last launch: span us", en.max(), "start pctl", np.percentile(st, [0, 25, 50, 75, 90, 100]), "dur pctl", np.percentile(en - st, [0, 50, 100]))
for lv in range(8): print(" level", lv, "start med", np.median(st[lv]), "max", st[lv].max(), "dur med", np.median(en[lv] - st[lv]), "end max", en[lv].max())
hw = a[:, :, 2]
print("distinct HW_ID", len(np.unique(hw)))
