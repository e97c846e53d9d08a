#!/bin/bash
# ablation build: libssm_hip_<name>.so = libssm_hip.so with kernels_pnp.hip compiled with extra flags.  Usage: scripts/build_pnp_variant.sh <name> <flags...>
# On the GPU box: cp semantic_slam_mapping_amd/libssm_hip_<name>.so semantic_slam_mapping_amd/libssm_hip.so (the box's copy is scratch)
set -e
N=$1; shift
cd "$(dirname "$0")/../semantic_slam_mapping_amd/csrc"
make -j6 >/dev/null
mkdir -p build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -w -c kernels_pnp.hip -o build_var/kernels_pnp_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libssm_hip_$N.so $(ls build/*.o | grep -v "kernels_pnp") build_var/kernels_pnp_$N.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built ../libssm_hip_$N.so
