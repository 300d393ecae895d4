import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
import ba_bench
from dvm_slam_amd import capi
ba_bench.run(0, 5, cpu_seconds=0, prewarm_s=1.0)
out = (C.c_longlong * 32)()
print(capi.lib().dvm_debug_chol_stamps(out))
v = np.array(out[:18], dtype=np.int64)
print("stamps (cycles from start):", (v - v[0]).tolist())
names = ["load","-","p0","s","tr0","p1","s","tr1","p2","s","tr2","p3","s","inv3+store","Linv","sync","storeLinv"]
d = np.diff(v)
for n, x in zip(names, d): print(f"{n:24s} {x:8d} cyc  {x/2400:7.2f} us")
