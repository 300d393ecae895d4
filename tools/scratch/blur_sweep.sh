cd $GRAFT_REPO_ROOT/dvm_slam_amd/csrc
git -C $GRAFT_REPO_ROOT checkout dvm_slam_amd/csrc/orb_kernels.hip 2>/dev/null
for k in 56 48 64; do
  sed -i "s/^constexpr int kBlurTW = 64, kBlurTH = [0-9]*;/constexpr int kBlurTW = 64, kBlurTH = $k;/" orb_device.h
  make -s > /dev/null 2>&1
  echo -n "kBlurTH=$k: "; python $GRAFT_REPO_ROOT/bench.py --no-ba --cpu-seconds 0 --no-pcie --no-exclusive 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['gpu_kernel_event_ms_per_launch']['blur'])"
done
python -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_orb.py -x -q 2>&1 | tail -1
