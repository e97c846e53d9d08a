#!/bin/bash
# Runs on the GPU box (gpurun): the bench lines and the rocprofv3 runs the summaries under profiles/ are made from.
# Usage: gpurun --timeout 1500 -- 'bash scripts/collect_profiles.sh stats|pmc|pmc_stereo|pmc_segnet'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out
# the sources a counter pass ran: written next to its output, quoted by scripts/pmc_traffic.py (bench.py reports a traffic profile only while these match)
sha_of_sources() { python3 - "$1" <<'PY'
import hashlib, json, os, sys
c = "semantic_slam_mapping_amd/csrc"
json.dump({f: hashlib.sha256(open(os.path.join(c, f), "rb").read()).hexdigest() for f in sorted(os.listdir(c)) if f.endswith((".hip", ".inc"))}, open(os.path.join(sys.argv[1], "sources_sha256.json"), "w"), indent=1)
PY
}
if [ "$1" = "stats" ]; then
  ( time timeout 300 python3 bench.py > $O/line_default.json 2> $O/line_default.err ) 2> $O/line_default.time      # the command the driver times: configs[1] + other_configs
  timeout 300 python3 bench.py --leaf 0.02 --no-other-configs --steps 5 --warmup 2 > $O/line_leaf002.json 2> $O/line_leaf002.err      # SURVEY s.8(d)'s second leaf (the map grows from 2^20 slots)
  timeout 300 python3 bench.py --segnet --frames 256 --batch 128 --steps 3 --warmup 1 > $O/line_segnet.json 2> $O/line_segnet.err
  timeout 300 python3 bench.py --stereo --steps 3 --warmup 1 > $O/line_stereo.json 2> $O/line_stereo.err
  timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --solve-poses --pose-frames 400 --pnp-device 0 > $O/line_poses_host.json 2> $O/line_poses_host.err
  timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --solve-poses --pose-frames 400 --pnp-device 1 > $O/line_poses_dev.json 2> $O/line_poses_dev.err
  timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --solve-poses --pose-frames 400 --pnp-device 1 --pose-threads 16 > $O/line_poses_dev16.json 2> $O/line_poses_dev16.err
  timeout 120 python3 scripts/tracker_concurrency.py 1 2 4 8 16 > $O/tracker_concurrency.txt 2>&1
  timeout 300 python3 scripts/per_call_latency.py $O/per_call.md > /dev/null 2>&1
  timeout 600 python3 scripts/mapper_update_cost.py 150 > $O/mapper_update_cost.md 2>&1
  rm -rf $O/p_stats $O/p_seg $O/p_st
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_stats -o runc -- python3 bench.py --steps 3 --warmup 1 --no-cpu --serial-only --no-other-configs > $O/p_stats.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_seg -o runc -- python3 bench.py --segnet --frames 256 --batch 128 --steps 3 --warmup 1 --no-cpu --serial-only > $O/p_seg.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_st -o runc -- python3 bench.py --stereo --steps 2 --warmup 1 --no-cpu --serial-only > $O/p_st.log 2>&1
  tail -c 600 $O/line_default.json; echo; tail -c 300 $O/line_segnet.json; echo; tail -c 300 $O/line_stereo.json
elif [ "$1" = "pmc_segnet" ]; then
  rm -rf $O/p_seg_pmc
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p_seg_pmc -o runc -- python3 bench.py --segnet --frames 256 --batch 128 --steps 3 --warmup 1 --no-cpu --serial-only > $O/p_seg_pmc.log 2>&1
  python3 scripts/segnet_layers.py $O/p_seg_pmc 64 $O/segnet_layers.md; tail -2 $O/segnet_layers.md
  # HBM traffic of the SegNet stage (roofline.traffic of configs[2]): one serialised step of 128 frames = two SegNet launch groups of 64
  rm -rf $O/p_fetch_seg $O/p_write_seg
  A="--segnet --frames 128 --batch 128 --steps 1 --warmup 0 --no-cpu --serial-only"
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p_fetch_seg -o runc -- python3 bench.py $A > $O/p_fetch_seg.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p_write_seg -o runc -- python3 bench.py $A > $O/p_write_seg.log 2>&1
  sha_of_sources $O/p_fetch_seg
  python3 scripts/pmc_traffic.py $O/p_fetch_seg $O/p_write_seg profiles/${ROUND:-r06}_segnet_traffic.json 64 128 > /dev/null      # (the box's copy: a `stats` pass in the same call quotes it)
  tail -1 $O/p_fetch_seg.log | cut -c1-200
elif [ "$1" = "pmc_stereo" ]; then
  rm -rf $O/p_sq_st $O/p_fetch_st $O/p_write_st
  A="--stereo --stereo-batch 32 --frames 64 --steps 1 --warmup 0 --no-cpu --serial-only"
  timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/p_sq_st -o runc -- python3 bench.py $A > $O/p_sq_st.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p_fetch_st -o runc -- python3 bench.py $A > $O/p_fetch_st.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p_write_st -o runc -- python3 bench.py $A > $O/p_write_st.log 2>&1
  sha_of_sources $O/p_fetch_st
  python3 scripts/pmc_traffic.py $O/p_fetch_st $O/p_write_st profiles/${ROUND:-r06}_stereo_traffic.json 32 64 > /dev/null
  tail -2 $O/p_sq_st.log | cut -c1-300
else
  export SSM_BENCH_H2D=0
  rm -rf $O/p_sq $O/p_fetch $O/p_write
  timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/p_sq -o runc -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-other-configs > $O/p_sq.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p_fetch -o runc -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-other-configs > $O/p_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p_write -o runc -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-other-configs > $O/p_write.log 2>&1
  sha_of_sources $O/p_fetch
  python3 scripts/pmc_traffic.py $O/p_fetch $O/p_write profiles/${ROUND:-r06}_traffic.json 250 > /dev/null
  tail -3 $O/p_sq.log | cut -c1-300
fi
