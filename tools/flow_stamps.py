"""Per-task wall-clock stamps of k_chol_flow (the last launch): build with tools/build_flow_stamps.sh (-DDVM_FLOW_DEBUG), run on the GPU.
Prints, per elimination-tree level, when its diagonal tiles and strips started / finished, and the phase split of the tasks on the
critical path.  python tools/flow_stamps.py [ring|loop]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DVM_BA_FLOW"] = "1"
os.environ.setdefault("DVM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dvm_slam_amd", "lib", "libdvmslam_hip_flowdbg.so"))
from dvm_slam_amd import capi, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "ring"
pr = synth.ba_problem() if which == "ring" else synth.ba_problem(laps=2, long_range_frac=0.002)
e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
ba = capi.BundleAdjuster()
ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
for _ in range(5):
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
    st = ba.optimize(3)
info = ba.schedule_info()
N = 4096
out = (C.c_longlong * (N * 8))()
print("rc", capi.lib().dvm_debug_flow_stamps(out, N * 8), "levels", info["levels"], "tiles", info["nz_tiles"])
v = np.array(out[:], dtype=np.int64).reshape(N, 8)
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"flow_stamps_{which}.npy"), v)
meta = v[:, 6]
kind = (meta >> 12) & 15; ti = (meta >> 16) & 0xFFFF; tj = (meta >> 32) & 0xFFFF; ncon = (meta >> 48) & 0xFFFF; wg = meta & 0xFFF
chain = v[3500:3500 + 200].copy()
v[3500:] = 0
used = v[:, 0] > 0
n = int(used.sum())
t0 = v[used, 0].min()
us = lambda x: (x - t0) / 100.0
print(f"{n} tasks, span {us(v[used, 4].max()):.1f} us")
# diagonal tiles in completion order: the chain
d = [i for i in range(n) if kind[i] == 0]
d.sort(key=lambda i: v[i, 4])
print("diagonal tiles by completion: col ncontrib | start  gather_done  in_lds  factored  published | waited(us)  wg xcc")
for i in d:
    print(f"  col {ti[i]:3d} nc {ncon[i]:3d} | {us(v[i,0]):7.1f} {us(v[i,1]):7.1f} {us(v[i,2]):7.1f} {us(v[i,3]):7.1f} {us(v[i,4]):7.1f} | wait {v[i,5]/100.0:6.1f}  wg {wg[i]} xcc {v[i,7]}")
print("chain (per column, by factor start): col | factor_start factored  L^-1+T in LDS  X in LDS+stores issued  iteration done | T prefetched")
cc = [i for i in range(200) if chain[i, 0] > 0]
cc.sort(key=lambda i: chain[i, 0])
for i in cc:
    print(f"  col {i:3d} | {us(chain[i,0]):7.1f} {us(chain[i,1]):7.1f} {us(chain[i,2]) if chain[i,2] else 0:7.1f} {us(chain[i,3]) if chain[i,3] else 0:7.1f} {us(chain[i,4]) if chain[i,4] else 0:7.1f} | {chain[i,6]}")
pp = [i for i in range(n) if kind[i] == 2]
pp.sort(key=lambda i: v[i, 4])
print("PRE tasks by completion: col nc | start gather_done staged stored published | waited")
for i in pp:
    print(f"  col {ti[i]:3d} nc {ncon[i]:3d} | {us(v[i,0]):7.1f} {us(v[i,1]):7.1f} {us(v[i,2]):7.1f} {us(v[i,3]):7.1f} {us(v[i,4]):7.1f} | wait {v[i,5]/100.0:6.1f}")
s = [i for i in range(n) if kind[i] in (1, 4)]
print("slices (first / last 40 by completion): (i,k) nc | start gather_done linv_there computed published | waited")
s.sort(key=lambda i: v[i, 4])
for i in s:
    print(f"  k{kind[i]} ({ti[i]:3d},{tj[i]:3d}) nc {ncon[i]:3d} | {us(v[i,0]):7.1f} {us(v[i,1]):7.1f} {us(v[i,2]):7.1f} {us(v[i,3]):7.1f} {us(v[i,4]):7.1f} | wait {v[i,5]/100.0:6.1f}")
b = [i for i in range(n) if kind[i] == 3]
b.sort(key=lambda i: v[i, 4])
print("back substitution: col | start flags_there done")
for i in b[:8] + b[-8:]:
    print(f"  col {ti[i]:3d} | {us(v[i,0]):7.1f} {us(v[i,1]):7.1f} {us(v[i,4]):7.1f}")
# phase statistics
dd = np.array(d); ss = np.array(s)
if len(dd):
    print("diag: factor us mean", np.mean((v[dd, 3] - v[dd, 2]) / 100.0), "publish", np.mean((v[dd, 4] - v[dd, 3]) / 100.0), "gather per contributor",
          np.sum((v[dd, 1] - v[dd, 0] - v[dd, 5]) / 100.0) / max(1, ncon[dd].sum()))
if len(ss):
    print("strip: trsm us mean", np.mean((v[ss, 3] - v[ss, 2]) / 100.0), "publish", np.mean((v[ss, 4] - v[ss, 3]) / 100.0), "linv wait after gather", np.mean((v[ss, 2] - v[ss, 1]) / 100.0),
          "gather per contributor (incl. waits)", np.sum((v[ss, 1] - v[ss, 0]) / 100.0) / max(1, ncon[ss].sum()), "contributors", int(ncon[ss].sum()))
if len(pp):
    ppa = np.array(pp)
    print("pre: gather per contributor (incl. waits)", np.sum((v[ppa, 1] - v[ppa, 0]) / 100.0) / max(1, ncon[ppa].sum()), "contributors", int(ncon[ppa].sum()))
