#!/bin/bash
# round 5, first GPU call: the new other_configs test, the timed default line, the SegNet traffic passes
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "bench" > $O/t_bench.log 2>&1; tail -5 $O/t_bench.log
( time timeout 600 python3 bench.py > $O/line_default.json 2> $O/line_default.err ) 2> $O/line_default.time; cat $O/line_default.time; tail -c 1500 $O/line_default.json; tail -3 $O/line_default.err
bash scripts/collect_profiles.sh pmc_segnet
