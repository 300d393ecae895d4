"""Cycle stamps of k_chol_diag (workgroup 0 of the last launch): build with tools/build_chol_stamps.sh (adds -DDVM_CHOL_DEBUG), run on the GPU.
Prints the phases of the kernel -- tile load, the four panels, the trailing updates between them, the L^-1 tail -- in cycles."""
import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ba_bench
from dvm_slam_amd import capi
ba_bench.run(0, 5, cpu_seconds=0, prewarm_s=1.0)
out = (C.c_longlong * 32)()
print(capi.lib().dvm_debug_chol_stamps(out))
v = np.array(out[:18], dtype=np.int64)
print("stamps (cycles from start):", (v - v[0]).tolist())
names = ["tile load", "tile -> LDS", "panel 0", "barrier", "trailing 0", "panel 1", "barrier", "trailing 1", "panel 2", "barrier", "trailing 2", "panel 3", "barrier", "-", "L^-1 row 3", "barrier", "store L^-1"]
d = np.diff(v)
w = np.array(out[18:21], dtype=np.int64) - v[0]
print('panel 3 end: wave0', int(v[12]-v[0]), 'wave1', int(w[0]), 'wave2', int(w[1]), 'wave3', int(w[2]))
for n, x in zip(names, d): print(f"{n:24s} {x:8d} cyc  {x/2400:7.2f} us")
