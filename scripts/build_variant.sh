#!/bin/bash
# build container: a variant of libssm_hip.so with extra -D flags for ONE source, into build_probe/<name>/ (not tracked; travels to the GPU box, where an
# experiment copies it over semantic_slam_mapping_amd/libssm_hip.so of the box's scratch copy).  Usage: scripts/build_variant.sh <name> <source.hip> <flags...>
set -e
N=$1; SRC=$2; shift 2
C=/root/repo/semantic_slam_mapping_amd/csrc
mkdir -p /root/repo/build_probe/$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c $C/$SRC -o /tmp/variant_$N.o 2>&1 | grep -E "error|Spill: [1-9]" || true
OBJS=$(ls $C/build/*.o | grep -v "${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build_probe/$N/libssm_hip.so $OBJS /tmp/variant_$N.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
ls -la /root/repo/build_probe/$N/libssm_hip.so
