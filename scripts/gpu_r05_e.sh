#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_sgbm.py tests/test_gpu_stereo_seq.py -x -q -m gpu > $O/t_e.log 2>&1; tail -3 $O/t_e.log
bash scripts/stereo_profile.sh 64 256 2>&1 | grep -E "sweep8|cost_kernel|rows8|^[0-9]"
timeout 300 python3 bench.py --stereo --steps 3 --warmup 1 --no-cpu > $O/line_stereo.json 2> $O/line_stereo.err; python3 -c "
import json; d=json.loads(open('$O/line_stereo.json').read().splitlines()[-1]); print('stereo', d['value'], d['roofline']['stages_ms_per_frame'])"
run() { python3 bench.py --steps 10 --warmup 2 --no-cpu --no-other-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
export SSM_BENCH_H2D=0
run base
SSM_MAP_CUS=64 run map64
SSM_MAP_CUS=64 SSM_MAP_CUS_SPREAD=1 run map64spread
SSM_MAP_CUS=96 SSM_MAP_CUS_SPREAD=1 run map96spread
SSM_MAP_CUS=96 SSM_MAP_CUS_SPREAD=1 SSM_CHAIN_CUS=1 run map96spread_chaincompl
SSM_MAP_CUS=64 SSM_MAP_CUS_SPREAD=1 SSM_CHAIN_CUS=1 run map64spread_chaincompl
SSM_MAP_CUS=128 SSM_MAP_CUS_SPREAD=1 run map128spread
run base2
