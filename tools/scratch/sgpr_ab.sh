# extractor throughput A/B: library variants under dvm_slam_amd/lib/<name>.so
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "$@"; do cp dvm_slam_amd/lib/$v.so dvm_slam_amd/lib/libdvmslam_hip.so; echo "== $v $(python bench.py --no-ba --no-pcie --cpu-seconds 0 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["value"]), j["roofline"].get("kernel_ms"), j["roofline"].get("exclusive"))')"; done; done
