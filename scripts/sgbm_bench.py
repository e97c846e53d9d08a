#!/usr/bin/env python3
"""Times ssm_stereo_depth (host images in, depth out) at the configs[3] geometry (1241 x 376, 80 disparities) against the
CPU oracle (oracle/sgbm.c, one thread).  Usage: python3 scripts/sgbm_bench.py [reps]"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import semantic_slam_mapping_amd as ssm
from oracle.binding import Oracle
from test_sgbm import stereo_pair, KITTI

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
left, right, _ = stereo_pair(376, 1241, 11, planes=((12, None), (30, (0.4, 0.9, 0.2, 0.5)), (60, (0.5, 0.95, 0.6, 0.8))), noise=5)
ctx = ssm.Context(0, width=640, height=480)
o = Oracle()
ctx.stereo_depth(left, right, **KITTI)
t0 = time.perf_counter()
for _ in range(reps):
    depth, disp = ctx.stereo_depth(left, right, **KITTI)
tg = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
ref = o.sgbm(left, right, o.sgbm_params()); dref = o.disparity_to_depth(ref, **KITTI)
tc = time.perf_counter() - t0
vol = (1241 - 80) * 376 * 80
print(f"GPU ssm_stereo_depth: {tg * 1e3:.2f} ms/frame ({1 / tg:.0f} frames/s), CPU oracle: {tc * 1e3:.1f} ms/frame, ratio {tc / tg:.0f}x; "
      f"identical: {np.array_equal(disp, ref) and np.array_equal(depth, dref)}; cost volume {vol / 1e6:.1f} M entries")
