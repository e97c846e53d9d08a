#!/bin/bash
# ablation build: libssm_hip_<name>.so = libssm_hip.so with the given kernel files compiled with extra flags.
# Usage: scripts/build_variant.sh <name> "<file stems>" <flags...>     e.g. scripts/build_variant.sh prio3 "kernels_match kernels_orb" -DSSM_MFMA_PRIO=3
# On the GPU box: cp semantic_slam_mapping_amd/libssm_hip_<name>.so semantic_slam_mapping_amd/libssm_hip.so (the box's copy is scratch)
set -e
N=$1; FILES=$2; shift 2
cd "$(dirname "$0")/../semantic_slam_mapping_amd/csrc"
make -j6 >/dev/null
mkdir -p build_var
EXCL=""
for f in $FILES; do
  extra=""; [ $f = kernels_match ] && extra="-fno-honor-nans"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $extra "$@" -w -c $f.hip -o build_var/${f}_$N.o &
  EXCL="$EXCL\|/$f.o"
done
wait
OBJS=$(ls build/*.o | grep -v "NONE$EXCL")
VAR=$(for f in $FILES; do echo build_var/${f}_$N.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libssm_hip_$N.so $OBJS $VAR -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built ../libssm_hip_$N.so
