#!/usr/bin/env python3
"""tools/f64_probe.py -- runs tools/bin/f64_probe on random doubles and compares the device's sqrt / division / reciprocal / sin / cos
bit for bit with numpy (glibc, x86-64) on the host.  Prints the number of differing results and the worst ulp distance."""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = 1 << 22
rng = np.random.default_rng(7)
a = np.concatenate([rng.standard_normal(n // 2) * 10.0 ** rng.integers(-12, 12, n // 2), rng.uniform(-0.8, 0.8, n // 4), rng.uniform(-1e-3, 1e-3, n // 4)])
b = rng.standard_normal(n) * 10.0 ** rng.integers(-12, 12, n)
np.concatenate([a, b]).tofile("/tmp/f64_in.bin")
subprocess.check_call([os.path.join(ROOT, "tools", "bin", "f64_probe"), "/tmp/f64_in.bin", "/tmp/f64_out.bin"])
o = np.fromfile("/tmp/f64_out.bin").reshape(5, n)
ref = [np.sqrt(np.abs(a)), a / b, 1.0 / b, np.sin(a), np.cos(a)]
out = {}
for name, g, r in zip(("sqrt", "div", "rcp", "sin", "cos"), o, ref):
    d = np.abs(g.view(np.int64) - r.view(np.int64))
    out[name] = {"differ": int((d != 0).sum()), "max_ulp": int(d.max()), "n": n}
    small = np.abs(a) < 0.8
    if name in ("sin", "cos"):
        out[name]["differ_abs_lt_0.8"] = int((d[small] != 0).sum()); out[name]["n_abs_lt_0.8"] = int(small.sum())
print(json.dumps(out))
