#!/usr/bin/env python3
"""Per-kernel clock and MFMA-pipe utilisation from
`rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv`.
clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x clock x duration).
Usage: kernel_clock.py <dir> [name substring]"""
import collections, csv, glob, statistics, sys
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(tr))}
val = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    val[r["Dispatch_Id"]][r["Counter_Name"]] = val[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
by = collections.defaultdict(list)
for k, c in val.items():
    if k not in dur or pat not in dur[k][1]: continue
    t = dur[k][0] * 1e-9
    if t <= 0: continue
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / t
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * clk * t) if clk else 0
    by[dur[k][1].split("(")[0][:40]].append((t * 1e6, clk / 1e9, busy))
for n, v in sorted(by.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    print("%-42s n %3d  us %8.1f  clock %.2f GHz  mfma busy %.3f" % (n, len(v), statistics.median(x[0] for x in v), statistics.median(x[1] for x in v), statistics.median(x[2] for x in v)))
