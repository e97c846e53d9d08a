#!/usr/bin/env python3
"""one ssm_orb_extract call (640x480 BGR + depth from host memory) as a kernel timeline.  Usage (GPU box):
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/orbtl -- python3 scripts/orb_call_timeline.py run     (20 calls)
   python3 scripts/orb_call_timeline.py report gpurun_out/orbtl                                                      (the last call's kernels: start offset, duration)"""
import glob
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import semantic_slam_mapping_amd as ssm
    c = ssm.Context(0, orb_features=1000, max_batch=1, voxel_capacity_log2=14, camera=(318.6, 255.3, 517.3, 516.5, 1000.0))
    H, W = 480, 640
    tb = c.dev_alloc(H * W * 3); td = c.dev_alloc(H * W * 2); ts = c.dev_alloc(H * W * 3); tp = c.dev_alloc(128)
    c.synth_frames_dev(0x5EED0000, 0, 1, tb, td, ts, tp); c.sync()
    bgr = c.d2h(tb, (H, W, 3), np.uint8); dep = c.d2h(td, (H, W), np.uint16)
    ts_ = []
    for i in range(20):
        t = time.perf_counter(); c.detect_features(bgr, dep); ts_.append(time.perf_counter() - t)
        time.sleep(0.002)
    print("call ms:", " ".join("%.3f" % (x * 1e3) for x in ts_))
    c.close()
else:
    import csv
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # calls are separated by the 2 ms sleeps: split where the gap exceeds 1 ms
    calls, cur = [], []
    for r in rows:
        if cur and int(r["Start_Timestamp"]) - int(cur[-1]["End_Timestamp"]) > 1_000_000:
            calls.append(cur); cur = []
        cur.append(r)
    calls.append(cur)
    last = calls[-1]; t0 = int(last[0]["Start_Timestamp"])
    print("| kernel | start us | duration us |\n|---|---:|---:|")
    for r in last:
        print("| %s | %.1f | %.1f |" % (r["Kernel_Name"][:60], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print("\nfirst kernel start -> last kernel end: %.1f us; sum of kernel durations: %.1f us; %d kernels" % ((int(last[-1]["End_Timestamp"]) - t0) / 1e3, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e3, len(last)))
