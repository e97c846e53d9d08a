#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
run() { python3 bench.py --stereo --steps 3 --warmup 1 --no-cpu "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$*', d['value'], d['roofline']['stages_ms_per_frame'])"; }
run --stereo-batch 64 --frames 256
run --stereo-batch 128 --frames 512
run --stereo-batch 96 --frames 384
run --stereo-batch 128 --frames 256
SSM_SGBM_STREAMS=3 run --stereo-batch 64 --frames 256
SSM_SGBM_STREAMS=1 run --stereo-batch 64 --frames 256
run --stereo-batch 64 --frames 256
