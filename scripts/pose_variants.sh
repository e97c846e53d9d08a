# GPU box: the closed pose loop with each variant library given (names), three runs each.  Usage: bash scripts/pose_variants.sh base fs3 fs4
cd $GRAFT_REPO_ROOT
B="python3 bench.py --solve-poses --pose-frames 400 --pnp-device 1 --steps 1 --warmup 1 --no-cpu --no-other-configs"
cp semantic_slam_mapping_amd/libssm_hip.so /tmp/libssm_hip_base.so
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/libssm_hip_base.so semantic_slam_mapping_amd/libssm_hip.so; else cp semantic_slam_mapping_amd/libssm_hip_$v.so semantic_slam_mapping_amd/libssm_hip.so; fi
  for r in 1 2 3; do timeout 300 $B 2>/dev/null | python3 -c "import json,sys; l=json.loads(sys.stdin.readlines()[-1]); sp=l['solve_poses']; print('$v', sp.get('frames_per_s'), sp['ms']['pose_chain'])"; done
done
cp /tmp/libssm_hip_base.so semantic_slam_mapping_amd/libssm_hip.so
