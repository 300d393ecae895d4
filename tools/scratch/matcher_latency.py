import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dvm_slam_amd import capi
from oracle import pyoracle as po
from matcher_scene import make_scene
sc = make_scene(po, 0)
def t(fn, n=30):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
print("SearchByProjection(Cur, Last) host mirror, N =", len(sc["kps_c"]), "ms", t(lambda: capi.search_by_projection_frames(th=15.0, **sc)))
n_o, _ = po.search_by_projection_frames(th=15.0, **sc)
print("  oracle (CPU) ms", t(lambda: po.search_by_projection_frames(th=15.0, **sc), 10))
