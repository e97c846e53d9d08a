#!/usr/bin/env python3
"""`Mapping cost time` of Mapper::viewer (reference src/mapper.cpp:157-160 prints it per update) with the viewer's map on the host (mapper_device_map=0: the
reference's schedule through ssm_backproject + host transform + ssm_voxel_filter of the whole map) against the device-resident form (ssm_backproject_dev +
ssm_viewer_map_update), on the synthetic stream with every frame a key-frame.  Runs on the GPU box: python3 scripts/mapper_update_cost.py [frames]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "semantic_slam_mapping_amd", "host")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
base = open(os.path.join(HOST, "parameters_test.txt")).read().replace("end_index=8", f"end_index={N}")
print(f"| viewer map | updates | updates at >= 50 key-frames | mean ms per update there | max ms | mean ms, all updates | map points at the end |\n|---|---:|---:|---:|---:|---:|---:|")
res = {}
for dev in (0, 1):
    with tempfile.TemporaryDirectory() as d:
        prm = os.path.join(d, "p.txt")
        open(prm, "w").write(base.replace("map_output=/tmp/ssm_test_map.pcd", f"map_output={d}/map.pcd") + f"\nmapper_device_map={dev}\nmapper_drain_ms=3000\nframe_period_ms=8\n")
        r = subprocess.run([os.path.join(HOST, "exp_mapping"), prm], capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print(r.stdout[-1500:], r.stderr[-1500:]); sys.exit(1)
        cost = [float(x) for x in re.findall(r"Mapping cost time: ([0-9.e+-]+)ms", r.stdout)]
        pts = [int(x) for x in re.findall(r"points in global map: (\d+)", r.stdout)]
        # the viewer prints after each update; the number of key-frames at an update is not printed: updates are ordered, take the last third as ">= 50" when N >= 100
        late = cost[len(cost) // 2:]
        res[dev] = (sum(late) / max(len(late), 1))
        print(f"| {'device (round 4)' if dev else 'host (round 3)'} | {len(cost)} | {len(late)} | {sum(late) / max(len(late), 1):.2f} | {max(cost):.2f} | {sum(cost) / max(len(cost), 1):.2f} | {pts[-1] if pts else 0} |")
print(f"\nratio host / device over the later half of the updates: {res[0] / max(res[1], 1e-9):.1f}x")
