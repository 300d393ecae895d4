#!/usr/bin/env python3
"""tools/kernels_sha.py -- one hash over the kernel sources of the product (dvm_slam_amd/csrc/*.{hip,cpp,h,inc} and the two Makefile flag
lines that shape the code).  The rocprofv3 folds under profiles/ are stamped with it when they are collected, bench.py computes it again
when it runs and says so in its JSON line when a fold it quotes was taken on OTHER kernels (`stale`)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dvm_slam_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.cpp")) + glob.glob(os.path.join(d, "*.h")) +
                       glob.glob(os.path.join(d, "*.inc")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernels_sha())
