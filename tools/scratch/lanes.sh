for q in 4 8; do for l in 2 3 4; do
  echo -n "HWQ=$q lanes=$l: "
  GPU_MAX_HW_QUEUES=$q python bench.py --lanes $l --no-ba --cpu-seconds 0 --no-pcie --no-exclusive --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step']/80,4))"
done; done
