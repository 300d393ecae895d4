#!/bin/bash
# a SEPARATE library with wall-clock stamps in k_chol_flow (-DDVM_FLOW_DEBUG): dvm_slam_amd/lib/libdvmslam_hip_flowdbg.so; use it with
# DVM_HIP_LIB=<that path> python tools/flow_stamps.py.  The release library is not touched.
set -e
cd "$(dirname "$0")/../dvm_slam_amd/csrc"
mkdir -p build_dbg
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DDVM_FLOW_DEBUG -c ba_kernels.hip -o build_dbg/ba_kernels.o
objs=$(ls build/*.o | grep -v ba_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libdvmslam_hip_flowdbg.so $objs build_dbg/ba_kernels.o
