# GPU box: the headline with each variant library given (names), three runs each.  Usage: bash scripts/headline_variants.sh base prio3 ...
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 10 --warmup 2 --no-cpu --no-other-configs"
cp semantic_slam_mapping_amd/libssm_hip.so /tmp/libssm_hip_base.so
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/libssm_hip_base.so semantic_slam_mapping_amd/libssm_hip.so; else cp semantic_slam_mapping_amd/libssm_hip_$v.so semantic_slam_mapping_amd/libssm_hip.so; fi
  for r in 1 2 3; do SSM_BENCH_H2D=0 timeout 300 $B 2>/dev/null | python3 -c "import json,sys; l=json.loads(sys.stdin.readlines()[-1]); print('$v', l['value'], l['ms_per_step'])"; done
done
cp /tmp/libssm_hip_base.so semantic_slam_mapping_amd/libssm_hip.so
