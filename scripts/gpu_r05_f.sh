#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_sgbm.py tests/test_gpu_stereo_seq.py -x -q -m gpu > $O/t_f.log 2>&1; tail -3 $O/t_f.log
timeout 600 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k sgbm > $O/t_f2.log 2>&1; tail -2 $O/t_f2.log
bash scripts/stereo_profile.sh 64 256 2>&1 | grep -E "sweep8|cost_kernel|rows8|^[0-9]"
timeout 300 python3 bench.py --stereo --steps 3 --warmup 1 --no-cpu > $O/line_stereo.json 2> $O/line_stereo.err; python3 -c "
import json; d=json.loads(open('$O/line_stereo.json').read().splitlines()[-1]); print('stereo', d['value'], d['roofline']['stages_ms_per_frame'])"
