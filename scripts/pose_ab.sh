cd $GRAFT_REPO_ROOT
B="python3 bench.py --solve-poses --pose-frames 400 --pnp-device 1 --steps 1 --warmup 1 --no-cpu --no-other-configs"
for r in 1 2 3; do timeout 300 $B 2>/dev/null | python3 -c "import json,sys; l=json.loads(sys.stdin.readlines()[-1]); sp=l['solve_poses']; print('fps', sp.get('frames_per_s'), {k:v for k,v in sp.items() if 'ms' in k})"; done
timeout 900 python3 -m pytest tests/test_gpu_tracker.py tests/test_gpu_pnp.py -x -q -m gpu 2>&1 | tail -3
