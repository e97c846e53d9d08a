#!/bin/bash
# GPU box: the 8-context sharding emulation, a long random fuzz hunt (SSM_FUZZ_SCALE=30: ~13 k cases, 2 min) and the SGBM hand-off soak.  Usage: gpurun --timeout 3600 -- bash scripts/gpu_hunt.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_sharding.py -x -q -m gpu 2>&1 | grep -E "passed|failed" 
SSM_FUZZ_SCALE=30 timeout 1500 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/t_hunt.log 2>&1; grep -E "passed|failed|FUZZ" gpurun_out/t_hunt.log | tail -5; wc -l gpurun_out/fuzz_cases.log
timeout 900 python3 scripts/sgbm_soak.py 10 256 2>&1 | tail -3
