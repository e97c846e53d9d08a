#!/bin/bash
# GPU box helper for development runs: gpurun -- 'bash scripts/gpu_try.sh <what>'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
case "$1" in
  sgbm)
    timeout 900 python3 -m pytest tests/test_sgbm.py tests/test_gpu_stereo_seq.py -x -q -m gpu 2>&1 | tail -15
    for F in 1 2; do
      echo "== form $F"; SSM_SGBM_FORM=$F timeout 300 python3 bench.py --stereo --steps 3 --warmup 1 --no-cpu 2>$O/st_form$F.err | tee $O/st_form$F.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline'].get('stages_ms_per_frame'), d['roofline'].get('frac'))"
    done
    bash scripts/stereo_profile.sh 64 256 2>&1 | tail -40
    ;;
  sgbm2)
    timeout 900 python3 -m pytest tests/test_sgbm.py tests/test_gpu_stereo_seq.py -x -q -m gpu 2>&1 | tail -5
    run() { echo "== $*"; env "$@" timeout 300 python3 bench.py --stereo --steps 3 --warmup 1 --no-cpu 2>>$O/st_try.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline'].get('stages_ms_per_frame'), d['roofline'].get('frac'))"; }
    run SSM_SGBM_FORM=2
    run SSM_SGBM_SWEEP_LANES=16
    run SSM_SGBM_STRIP=64
    run SSM_SGBM_STREAMS=3
    bash scripts/stereo_profile.sh 64 256 2>&1 | tail -12
    ;;
  segpmc)
    rm -rf $O/p_seg_pmc
    timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p_seg_pmc -o runc -- python3 bench.py --segnet --frames 256 --batch 128 --steps 3 --warmup 1 --no-cpu --serial-only > $O/p_seg_pmc.log 2>&1
    python3 scripts/segnet_layers.py $O/p_seg_pmc 64 $O/segnet_layers.md; tail -3 $O/segnet_layers.md
    ;;
  mapper)
    timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "viewer_map or generate_point or voxel" 2>&1 | tail -5
    timeout 900 python3 -m pytest tests/test_host_cpp.py -x -q -m gpu 2>&1 | tail -5
    timeout 600 python3 scripts/mapper_update_cost.py 150 2>&1 | tail -8 | tee $O/mapper_update_cost.md
    ;;
  fuzz)
    rm -f $O/fuzz_cases.log
    (time timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu) 2>&1 | tail -25
    wc -l $O/fuzz_cases.log
    ;;
  pnp)
    for G in 4 1 2 8; do
      echo "== blocks $G"; SSM_PNP_BLOCKS=$G timeout 600 python3 -m pytest tests/test_gpu_pnp.py tests/test_gpu_tracker.py -x -q -m gpu 2>&1 | tail -3
      SSM_PNP_BLOCKS=$G timeout 300 python3 bench.py --solve-poses --pose-frames 400 --pnp-device 1 --steps 3 --warmup 1 --no-cpu 2>$O/pnp_g$G.err | tee $O/pnp_g$G.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); si=[v for v in (d.get('solve_poses'), d.get('config',{}).get('solve_poses')) if v]; print(d['value'], si[0]['frames_per_s'] if si else None, si[0]['ms'] if si else d.keys())"
    done
    SSM_FUZZ_SCALE=1 timeout 600 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "pnp or stereo_vo" 2>&1 | tail -3
    ;;
  alltests)
    timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15
    ;;
esac
