#!/usr/bin/env python3
"""per-kernel-name medians of a rocprofv3 --pmc run of `bench.py --segnet`: duration, MFMA busy, LDS active / bank-conflict cycles (per CU-cycle)"""
import collections, csv, glob, statistics, sys
d = sys.argv[1]
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
val = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    val[r["Dispatch_Id"]][r["Counter_Name"]] = val[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
by = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    n = r["Kernel_Name"]
    if "conv3x3" not in n: continue
    key = ("wino" if "wino" in n else "dma2" if "dma2" in n else "first") + " grid %s" % r["Grid_Size_X"]
    t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    c = val.get(r["Dispatch_Id"], {})
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / t if t else 0
    by[key].append((t, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * clk * t) if clk else 0, c.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * clk * t) if clk else 0,
                    c.get("SQ_LDS_BANK_CONFLICT", 0) / (256 * clk * t) if clk else 0, clk))
for k, v in sorted(by.items()):
    print("%-22s n=%4d  t=%8.1f us  mfma_busy=%.3f  lds_active=%.3f  lds_conflict=%.3f  clk=%.2f GHz  total=%.1f ms" % (k, len(v), statistics.median(x[0] for x in v) * 1e6,
          statistics.median(x[1] for x in v), statistics.median(x[2] for x in v), statistics.median(x[3] for x in v), statistics.median(x[4] for x in v) / 1e9, sum(x[0] for x in v) * 1e3))
