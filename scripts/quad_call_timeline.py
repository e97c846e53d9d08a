#!/usr/bin/env python3
"""one ssm_quad_track call (1241 x 376) as a kernel timeline: `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/quad_call_timeline.py run`, then `... report DIR`"""
import glob, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import semantic_slam_mapping_amd as ssm
    from bench import stereo_sequence
    L, R = stereo_sequence(3, 1241, 376, 100)
    c = ssm.Context(0, width=1241, height=376, max_batch=1, orb_features=1000)
    ts = []
    for i in range(12):
        t = time.perf_counter(); c.quad_track(L[1], R[1], L[0], R[0]); ts.append(time.perf_counter() - t); time.sleep(0.003)
    print("call ms:", " ".join("%.3f" % (x * 1e3) for x in ts)); c.close()
else:
    import csv
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    calls, cur = [], []
    for r in rows:
        if cur and int(r["Start_Timestamp"]) - int(cur[-1]["End_Timestamp"]) > 1_500_000:
            calls.append(cur); cur = []
        cur.append(r)
    calls.append(cur)
    last = calls[-1]; t0 = int(last[0]["Start_Timestamp"])
    print("| kernel | start us | duration us |\n|---|---:|---:|")
    for r in last:
        print("| %s | %.1f | %.1f |" % (r["Kernel_Name"][:50], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print("\nfirst kernel start -> last kernel end: %.1f us; sum of kernel durations: %.1f us; %d kernels" % ((int(last[-1]["End_Timestamp"]) - t0) / 1e3, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e3, len(last)))
