#!/usr/bin/env python3
"""Per-layer SegNet kernel durations from a rocprofv3 --kernel-trace run (median over the forward passes).
Usage: layer_times.py <dir>"""
import csv, glob, statistics, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows
       if any(k in r["Kernel_Name"] for k in ("conv3x3", "unpool", "segnet_prep", "label_color"))]
fw, cur = [], []
for s in seq:
    if "prep" in s[0] and cur:
        fw.append(cur); cur = []
    cur.append(s)
fw.append(cur)
L = max(len(x) for x in fw); fw = [x for x in fw if len(x) == L]
tot = 0
for i in range(L):
    t = statistics.median(x[i][1] for x in fw); tot += t
    n = fw[-1][i][0]
    print(i, n[n.find("conv3x3"):][:34] if "conv3x3" in n else n[:24], round(t, 1))
print("total", round(tot, 1), "passes", len(fw))
