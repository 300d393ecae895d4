import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ba_bench
print(json.dumps(ba_bench.run(0, 10, cpu_seconds=0, prewarm_s=2.0)))
