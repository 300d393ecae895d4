"""tools/lba_fast_leg.py -- bench_legs.lba_batch alone (the sequential-order kernel, the fast form, the tile solver batch), JSON on stdout"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs
r = bench_legs.lba_batch(0, cpu_windows=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
print(json.dumps({k: r[k] for k in r if k in ("by_K", "fast_windows", "fast_windows_pool", "concurrent_tile_solver", "parity_vs_cpu", "tile_solver_sequential")}))
