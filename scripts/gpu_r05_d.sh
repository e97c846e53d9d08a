#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_sgbm.py tests/test_gpu_stereo_seq.py tests/test_gpu_tracker.py -x -q -m gpu > $O/t_d.log 2>&1; tail -4 $O/t_d.log
timeout 300 python3 bench.py --stereo --steps 3 --warmup 1 --no-cpu > $O/line_stereo.json 2> $O/line_stereo.err; python3 -c "
import json; d=json.loads(open('$O/line_stereo.json').read().splitlines()[-1]); print('stereo', d['value'], d['roofline']['stages_ms_per_frame'])"
bash scripts/stereo_profile.sh 64 256 2>&1 | tail -30
timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-other-configs --solve-poses --pose-frames 400 --pnp-device 1 > $O/line_poses_dev.json 2> $O/line_poses_dev.err; python3 -c "
import json; d=json.loads(open('$O/line_poses_dev.json').read().splitlines()[-1]); print(d['value'], d['solve_poses']['ms'], d['solve_poses']['work'])"
cp semantic_slam_mapping_amd/libssm_hip.so /tmp/keep.so; cp semantic_slam_mapping_amd/libssm_hip_prof.so semantic_slam_mapping_amd/libssm_hip.so
timeout 300 python3 bench.py --steps 1 --warmup 1 --no-cpu --no-other-configs --solve-poses --pose-frames 400 --pnp-device 1 > $O/pnp_prof.json 2> $O/pnp_prof.err; grep "pnp chain" $O/pnp_prof.err | tail -2
cp /tmp/keep.so semantic_slam_mapping_amd/libssm_hip.so
