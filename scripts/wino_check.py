#!/usr/bin/env python3
"""one conv layer through ssm_segnet_debug_op on integer data against numpy (direct 3x3 correlation): the Winograd kernel's bring-up check.  Usage: wino_check.py [layer h w]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import semantic_slam_mapping_amd as ssm
layer, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 16, 32)
c = ssm.Context(0, width=640, height=480, max_batch=1)
cin, cout, _, _ = c.segnet_layers()[layer]
rng = np.random.default_rng(5)
wt = rng.integers(-1, 2, (cout, cin, 3, 3)).astype(np.float32)
sc = np.ones(cout, np.float32); sh = np.zeros(cout, np.float32)
c.segnet_set_layer(layer, wt, sc, sh)
x = rng.integers(-4, 5, (h, w, cin)).astype(np.float32)
xp = np.zeros((h, w, (cin + 15) // 16 * 16), np.float16); xp[:, :, :cin] = x
print("launching", layer, h, w, cin, cout, flush=True)
got = c.segnet_debug_conv(layer, xp).astype(np.float32)
pad = np.zeros((h + 2, w + 2, cin), np.float32); pad[1:-1, 1:-1] = x
y = np.zeros((h, w, cout), np.float32)
for dy in range(3):
    for dx in range(3):
        y += pad[dy:dy + h, dx:dx + w] @ wt[:, :, dy, dx].T
y = np.maximum(y, 0)
bad = np.argwhere(got != y)
print("mismatches:", len(bad), "of", y.size)
if len(bad):
    print("first:", bad[:8].tolist(), got[tuple(bad[0])], y[tuple(bad[0])])
    print("rows with errors:", sorted(set(bad[:, 0].tolist()))[:20], "cols:", sorted(set(bad[:, 1].tolist()))[:40], "chans:", sorted(set(bad[:, 2].tolist()))[:20])
c.close()
