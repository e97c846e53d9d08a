#!/bin/bash
# GPU box: what the driver runs at round end -- the -m gpu tests, smoke(), the default bench line.  Usage: gpurun --timeout 2400 -- bash scripts/gpu_gate.sh
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 1800 python3 -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; grep -E "passed|failed" $O/t_all.log | tail -2
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 300 python3 bench.py > $O/gate_line.json 2> $O/gate_line.err ) 2>&1 | grep real
python3 -c "
import json; d=json.loads(open('$O/gate_line.json').read().splitlines()[-1]); oc=d['other_configs']
print(d['value'], d['roofline']['frac'], {k:(v.get('value'), v['roofline']['frac']) for k,v in oc.items() if isinstance(v,dict)})"
