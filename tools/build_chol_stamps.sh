#!/bin/bash
# rebuild ba_kernels.o with cycle stamps in k_chol_diag (DVM_CHOL_DEBUG) and relink; `touch ba_kernels.hip; make` undoes it
set -e
cd "$(dirname "$0")/../dvm_slam_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DDVM_CHOL_DEBUG -c ba_kernels.hip -o build/ba_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libdvmslam_hip.so build/*.o
