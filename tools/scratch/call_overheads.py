import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dvm_slam_amd import capi
rng = np.random.default_rng(0)
def t(fn, n=30):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
A = rng.integers(0, 256, (1000, 32), dtype=np.uint8); B = rng.integers(0, 256, (1000, 32), dtype=np.uint8)
print("hamming_matrix 1000x1000      ms", t(lambda: capi.hamming_matrix(A, B)))
import torch
x = torch.zeros(1, device="cuda")
def malloc_free():
    import ctypes
    p = ctypes.c_void_p()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20)); hip.hipFree(p)
print("hipMalloc + hipFree 1 MB       ms", t(malloc_free, 50))
from test_gpu_ba import _pose_case
c = _pose_case(1, n_pts=300, out_frac=0.1)
print("pose_optimize 1 frame x 300    ms", t(lambda: capi.pose_optimize(c[0][None], c[1][None], c[2][None], c[3][None], np.array([300], np.int32), c[4])))
