"""oracle/cpu_agent.py seconds [libpath] -- ONE agent on ONE core for bench.py's one-agent-per-core CPU baseline:
the oracle's extract + windowed frame-to-frame match over the synthetic 640x480 stream for ~`seconds`; prints
"<frames> <seconds>".  TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/oracle.h)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvm_slam_amd import synth        # noqa: E402  (synthetic frames only: no GPU code is touched)
from oracle import pyoracle as po      # noqa: E402


def main():
    seconds = float(sys.argv[1])
    libpath = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
    frames = synth.frame_stream(8)
    orc = po.OrbOracle(libpath=libpath)
    scale = orc.tables()["scale"]
    prev, done, t_total, i = None, 0, 0.0, 0
    while t_total < seconds and done < 4096:
        f = frames[i % len(frames)]
        t0 = time.perf_counter()
        n, k, d, _ = orc.extract(f)
        if prev is not None:
            kq, dq = prev
            po.Grid(k).match_window(d, dq, kq["x"], kq["y"], (np.float32(15) * scale[kq["octave"]]).astype(np.float32), kq["octave"] - 1, kq["octave"] + 1)
        t_total += time.perf_counter() - t0
        prev = (k, d); done += 1; i += 1
    print(done, t_total)


if __name__ == "__main__":
    main()
