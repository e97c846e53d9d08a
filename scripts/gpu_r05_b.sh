#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_sgbm.py tests/test_gpu_stereo_seq.py tests/test_gpu_tracker.py tests/test_host_cpp.py tests/test_gpu_pnp.py tests/test_vo.py tests/test_gpu_quad.py -x -q -m gpu > $O/t_b.log 2>&1; tail -8 $O/t_b.log
timeout 300 python3 bench.py --leaf 0.02 --no-other-configs --steps 5 --warmup 2 > $O/line_leaf002.json 2> $O/line_leaf002.err; tail -c 600 $O/line_leaf002.json; tail -2 $O/line_leaf002.err
