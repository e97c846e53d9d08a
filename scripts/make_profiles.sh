#!/bin/bash
# Turns the raw outputs of `scripts/collect_profiles.sh stats` + `pmc` + `pmc_stereo` (merged back into gpurun_out/) into the committed summaries under profiles/.
# Usage (repo root, in the build container): bash scripts/make_profiles.sh r03
set -e
R=${1:-r05}; O=gpurun_out
python3 scripts/prof_summary.py $O/p_stats profiles/${R}_kernel_stats.md $O/p_stats.log --warm 0.25 --frames-per-launch 250 > /dev/null
python3 scripts/prof_summary.py $O/p_seg profiles/${R}_segnet_kernel_stats.md $O/p_seg.log --warm 0.25 --frames-per-launch 64 > /dev/null
python3 scripts/prof_summary.py $O/p_st profiles/${R}_stereo_kernel_stats.md $O/p_st.log --warm 0.34 --frames-per-launch 128 > /dev/null
python3 scripts/sq_summary.py $O/p_sq profiles/${R}_sq_counters.md "${R}: rocprofv3 --pmc SQ_* per kernel (bench.py --steps 1 --warmup 0, SSM_BENCH_H2D=0: two passes of 1000 frames, batch 250)" 2000
python3 scripts/pmc_traffic.py $O/p_fetch $O/p_write profiles/${R}_traffic.json 250 > /dev/null
if [ -d $O/p_sq_st ]; then
  python3 scripts/sq_summary.py $O/p_sq_st profiles/${R}_stereo_sq_counters.md "${R}: rocprofv3 --pmc SQ_* per kernel of the batched stereo path (bench.py --stereo --frames 64 --steps 1 --warmup 0 --serial-only: 64 frame pairs, 32 per launch)" 64
  python3 scripts/pmc_traffic.py $O/p_fetch_st $O/p_write_st profiles/${R}_stereo_traffic.json 32 64 > /dev/null
fi
if [ -d $O/p_fetch_seg ]; then
  python3 scripts/pmc_traffic.py $O/p_fetch_seg $O/p_write_seg profiles/${R}_segnet_traffic.json 64 128 > /dev/null
fi
tail -1 $O/line_default.json > profiles/${R}_bench_line.json
[ -s $O/line_leaf002.json ] && tail -1 $O/line_leaf002.json > profiles/${R}_bench_line_leaf002.json
tail -1 $O/line_segnet.json > profiles/${R}_bench_line_segnet.json
tail -1 $O/line_stereo.json > profiles/${R}_bench_line_stereo.json
[ -s $O/line_poses_host.json ] && tail -1 $O/line_poses_host.json > profiles/${R}_bench_line_poses_host.json
[ -s $O/line_poses_dev.json ] && tail -1 $O/line_poses_dev.json > profiles/${R}_bench_line_poses_device.json
[ -s $O/line_poses_dev16.json ] && tail -1 $O/line_poses_dev16.json > profiles/${R}_bench_line_poses_device_16seq.json
[ -s $O/tracker_concurrency.txt ] && cp $O/tracker_concurrency.txt profiles/${R}_tracker_concurrency.txt
[ -s $O/per_call.md ] && cp $O/per_call.md profiles/${R}_per_call_latency.md
[ -s $O/mapper_update_cost.md ] && cp $O/mapper_update_cost.md profiles/${R}_mapper_update_cost.md
echo "profiles/${R}_* regenerated"
