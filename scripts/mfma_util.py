#!/usr/bin/env python3
"""Per-dispatch clock and MFMA-pipe utilisation of the SegNet conv kernels from
`rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv`.
clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x clock x duration).
Usage: mfma_util.py <dir>"""
import collections, csv, glob, statistics, sys
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: (int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(tr))}
val = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    val[r["Dispatch_Id"]][r["Counter_Name"]] = val[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = sorted((dur[k][0], k) for k in val if k in dur and "conv3x3" in dur[k][2])
seq = []
for _, k in rows:
    t = dur[k][1] * 1e-9; c = val[k]
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / t
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * clk * t) if clk else 0
    n = dur[k][2]; seq.append((n[n.find("conv3x3"):][:30], t * 1e6, clk / 1e9, busy))
L = 26
fw = [seq[i:i + L] for i in range(0, len(seq) - L + 1, L)]
for i in range(L):
    print(i, fw[-1][i][0], "us %.1f  clock %.2f GHz  mfma busy %.3f" % (statistics.median(x[i][1] for x in fw), statistics.median(x[i][2] for x in fw), statistics.median(x[i][3] for x in fw)))
