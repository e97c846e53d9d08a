#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_tracker.py tests/test_gpu_pnp.py tests/test_host_cpp.py -x -q -m gpu > $O/t_c.log 2>&1; tail -5 $O/t_c.log
timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-other-configs --solve-poses --pose-frames 400 --pnp-device 1 > $O/line_poses_dev.json 2> $O/line_poses_dev.err; python3 -c "
import json; d=json.loads(open('$O/line_poses_dev.json').read().splitlines()[-1]); print(d['value'], d['solve_poses'])"
timeout 300 python3 bench.py --leaf 0.02 --no-other-configs --steps 5 --warmup 2 > $O/line_leaf002.json 2> $O/line_leaf002.err; tail -c 700 $O/line_leaf002.json; tail -2 $O/line_leaf002.err
# section clocks of the chain (ablation build)
cp semantic_slam_mapping_amd/libssm_hip.so /tmp/keep.so; cp semantic_slam_mapping_amd/libssm_hip_prof.so semantic_slam_mapping_amd/libssm_hip.so
timeout 300 python3 bench.py --steps 1 --warmup 1 --no-cpu --no-other-configs --solve-poses --pose-frames 400 --pnp-device 1 > $O/pnp_prof.json 2> $O/pnp_prof.err; grep "pnp chain" $O/pnp_prof.err | tail -3
cp /tmp/keep.so semantic_slam_mapping_amd/libssm_hip.so
