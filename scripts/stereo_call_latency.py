#!/usr/bin/env python3
"""latency of the per-pair stereo entry points (what Tracker::estimateVO + FrameReader's KITTI branch call once per frame) at 1241 x 376.  Usage (GPU box): python scripts/stereo_call_latency.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import semantic_slam_mapping_amd as ssm
from bench import stereo_sequence, KITTI


def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    ts.sort(); return ts[len(ts) // 2] * 1e3


L, R = stereo_sequence(3, 1241, 376, 100)
c = ssm.Context(0, width=1241, height=376, max_batch=1, orb_features=2000)
rgb = np.stack([L[1]] * 3, -1).copy()
print("| call | median ms |\n|---|---:|")
print("| ssm_orb_extract 1241x376 BGR (2000 features) | %.3f |" % timeit(lambda: c.detect_features(rgb, None)))
qm = c.quad_track(L[1], R[1], L[0], R[0])
print("| ssm_quad_track (GFTT + 4 LK passes + filter; %d quad matches) | %.3f |" % (len(qm), timeit(lambda: c.quad_track(L[1], R[1], L[0], R[0]))))
from semantic_slam_mapping_amd.api import GlibcRand
rnd = GlibcRand(0)
samples = np.array([[int(rnd.draws(1)[0]) % len(qm), int(rnd.draws(1)[0]) % (len(qm) - 1), int(rnd.draws(1)[0]) % (len(qm) - 2)] for _ in range(200)], np.int32)
try:
    print("| ssm_vo_estimate (200 hypotheses) | %.3f |" % timeit(lambda: c.vo_estimate(qm, KITTI["f"], KITTI["cu"], KITTI["cv"], KITTI["baseline"], samples)))
except Exception as e:
    print("| ssm_vo_estimate | %r |" % e)
print("| ssm_stereo_depth (SGBM 80 disparities + depth conversion, one pair) | %.3f |" % timeit(lambda: c.stereo_depth(L[1], R[1], **KITTI), n=10))
c.close()
