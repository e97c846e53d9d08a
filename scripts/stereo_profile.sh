#!/bin/bash
# GPU box: per-kernel profile of the batched stereo path (bench.py --stereo, serialised stages).  Usage: gpurun -- 'bash scripts/stereo_profile.sh [batch] [frames]'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
B=${1:-32}; F=${2:-128}
rm -rf gpurun_out/p_st
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_st -o runc -- python3 bench.py --stereo --stereo-batch $B --frames $F --steps 2 --warmup 1 --no-cpu --serial-only > gpurun_out/p_st.log 2>&1
python3 scripts/prof_summary.py gpurun_out/p_st gpurun_out/p_st.md gpurun_out/p_st.log
grep -o '"value": [0-9.]*' gpurun_out/p_st.log | head -1
sed -n 12,45p gpurun_out/p_st.md
