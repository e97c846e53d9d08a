#!/usr/bin/env python3
"""Ablation helper: with libssm_hip.so built with -DSSM_OT_PROF, prints the shader clocks octree_kernel's thread 0 spends per section and level
(summed over the blocks of a 250-frame batch of the bench stream).  Usage (GPU box): python scripts/octree_prof.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semantic_slam_mapping_amd as ssm            # noqa: E402

c = ssm.Context(0, orb_features=1000, max_batch=50, camera=(318.6, 255.3, 517.3, 516.5, 1000.0))
n = 50
H, W = 480, 640
db = c.dev_alloc(n * H * W * 3); dd = c.dev_alloc(n * H * W * 2); ds = c.dev_alloc(n * H * W * 3); dp = c.dev_alloc(n * 128)
c.synth_frames_dev(0x5EED0000, 0, n, db, dd, ds, dp); c.sync()          # the bench's synthetic stream, made on the device
lib = ctypes.CDLL(os.path.join(ROOT, "semantic_slam_mapping_amd", "libssm_hip.so"))
c.seq_process(db, dd, None, None, n, stages=ssm.api.STAGE_ORB); c.sync()
lib.ssm_debug_octree_prof()                        # (discard the first call)
c.seq_process(db, dd, None, None, n, stages=ssm.api.STAGE_ORB); c.sync()
print(f"per level, {n} blocks each:", file=sys.stderr)
lib.ssm_debug_octree_prof()
c.close()
