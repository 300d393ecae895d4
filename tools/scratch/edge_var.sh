cd $GRAFT_REPO_ROOT/dvm_slam_amd/csrc
python $GRAFT_REPO_ROOT/tools/ba_only.py 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', b['value'], b['phase_ms']['landmarks_update_linearise_per_trial'])"
for v in EDGE_NOSTORE; do
  touch ba_kernels.hip; make -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-unused-result -D$v" > /dev/null 2>&1
  echo -n "$v: "; python $GRAFT_REPO_ROOT/tools/ba_only.py 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['value'], b['phase_ms']['landmarks_update_linearise_per_trial'])"
done
