cd $GRAFT_REPO_ROOT/dvm_slam_amd/csrc
for k in 2 3 5 6; do
  sed -i "s/^constexpr int kDescPerWave = [0-9]*;/constexpr int kDescPerWave = $k;/" orb_kernels.hip
  make -s > /dev/null 2>&1
  echo -n "kDescPerWave=$k: "; python $GRAFT_REPO_ROOT/bench.py --no-ba --cpu-seconds 0 --no-pcie --no-exclusive 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['gpu_kernel_event_ms_per_launch']['orient_desc'])"
done
