"""A/B of the two forms of the reduced solve on the bench's two BA workloads: level launches (DVM_BA_FLOW=0) vs the flow kernel (=1).
Run on the GPU box: python tools/ba_flow_ab.py [repeats]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ba_bench  # noqa: E402

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 100
out = {}
for mode in ("0", "1"):
    os.environ["DVM_BA_FLOW"] = mode; os.environ["DVM_BA_BORDER"] = os.environ.get("AB_BORDER", "1")
    r = ba_bench.run(0, 10, cpu_seconds=0.0, repeats=rep)
    lc = ba_bench.run_loop_closed(0, 10, repeats=max(4, rep // 10), cpu_iters=0)
    out[mode] = {"ring_it_s": r["value"], "ring_ms": r["ms_per_iteration"], "ring_phase_ms": r.get("phase_ms"), "ring_chi2": r["chi2_final"],
                 "loop_it_s": lc["value"], "loop_ms": lc["ms_per_iteration"], "loop_chi2": lc["chi2_final"]}
    print(mode, json.dumps(out[mode]), flush=True)
