#!/usr/bin/env python3
"""Per-kernel busy fraction of the texture-address units (the vector-memory front end of a CU) from a
`rocprofv3 --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum --kernel-trace` run.
Usage: ta_summary.py gpurun_out/<dir> profiles/<name>.md "<title>"
TA busy = TA_BUSY_avr (busy cycles, averaged over the CUs' TA instances) / (GRBM_GUI_ACTIVE / 8 XCDs)."""
import collections
import csv
import glob
import re
import sys


def main():
    src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
    f = glob.glob(src + "/**/*counter_collection.csv", recursive=True)[0]
    tr = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); ns = collections.defaultdict(float); n = collections.defaultdict(int); seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if "rocprim" in k or k.startswith("__amd") or k.startswith("synth"):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); ns[k] += dur.get(r["Dispatch_Id"], 0); n[k] += 1
    with open(dst, "w") as o:
        o.write(f"# {title}\n\n`TA busy` = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8): the fraction of the kernel's cycles in which a CU's texture-address unit (the front end of every "
                "vector-memory instruction: address generation, splitting into cache-line accesses) had work, averaged over the CUs; `max` = the busiest unit; "
                "`TCP pending` = TCP_PENDING_STALL_CYCLES per CU-cycle (the L1 waiting for L2).  A kernel near 1.0 is bound by its vector-memory INSTRUCTIONS "
                "(count and how many lines each touches), whatever its HBM bytes and its VALU rate say.\n\n")
        o.write("| kernel | launches | ms | TA busy | TA busy (max unit) | TCP pending |\n|---|---:|---:|---:|---:|---:|\n")
        for k, c in sorted(agg.items(), key=lambda kv: -ns[kv[0]]):
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            if cyc <= 0:
                continue
            o.write(f"| {k} | {n[k]} | {ns[k] / 1e6:.3f} | {c['TA_BUSY_avr'] / cyc:.2f} | {c['TA_BUSY_max'] / cyc:.2f} | {c['TCP_PENDING_STALL_CYCLES_sum'] / 256.0 / cyc:.2f} |\n")


if __name__ == "__main__":
    main()
