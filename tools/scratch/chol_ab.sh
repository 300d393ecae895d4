# A/B of DVM_CHOL_DEBUG builds of the library on the same box (dvm_slam_amd/lib/<name>_dbg.so for every name given)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "$@"; do cp dvm_slam_amd/lib/${v}_dbg.so dvm_slam_amd/lib/libdvmslam_hip.so; echo "== $v"; python tools/scratch/chol_stamps.py 2>&1 | grep "stamps (cycles"; done; done
