import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ba_bench
ba_bench._prewarm(2.0)
r = ba_bench.run_loop_closed(0, 10, repeats=40, cpu_iters=0)
print(json.dumps({k: r[k] for k in ("value", "ms_per_iteration", "solver", "solver_info", "roofline", "phase_ms", "schedule")}))
