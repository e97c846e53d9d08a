#!/bin/bash
# ablation build: libssm_hip_prof.so = libssm_hip.so with -DSSM_PNP_PROF (shader clocks per section of the pose chain, printed by ssm_tracker_run to stderr).
# On the GPU box: cp semantic_slam_mapping_amd/libssm_hip_prof.so semantic_slam_mapping_amd/libssm_hip.so (the box's copy is scratch), then bench.py --solve-poses --pnp-device 1
set -e
cd "$(dirname "$0")/../semantic_slam_mapping_amd/csrc"
make -j6 >/dev/null
mkdir -p build_prof
for f in kernels_pnp ssm_track ssm_abi; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSSM_PNP_PROF -w -c $f.hip -o build_prof/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libssm_hip_prof.so $(ls build/*.o | grep -v "kernels_pnp\|ssm_track\|ssm_abi") build_prof/kernels_pnp.o build_prof/ssm_track.o build_prof/ssm_abi.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built ../libssm_hip_prof.so
