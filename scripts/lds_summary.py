#!/usr/bin/env python3
"""Per-kernel LDS load from a `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES --kernel-trace` run.
Usage: lds_summary.py gpurun_out/<dir> "<title>" >> profiles/<name>.md
LDS active = SQ_LDS_IDX_ACTIVE per CU-cycle (the share of the kernel's cycles in which a CU's LDS was indexing), conflicts = SQ_LDS_BANK_CONFLICT per CU-cycle (part of it),
CU busy = SQ_BUSY_CU_CYCLES per CU-cycle (a CU holds at least one wave)."""
import collections
import csv
import glob
import re
import sys


def main():
    src, title = sys.argv[1], sys.argv[2]
    f = glob.glob(src + "/**/*counter_collection.csv", recursive=True)[0]
    tr = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); ns = collections.defaultdict(float); seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if "rocprim" in k or k.startswith("__amd") or k.startswith("synth"):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); ns[k] += dur.get(r["Dispatch_Id"], 0)
    print(f"## {title}\n")
    print("| kernel | ms | LDS active | of which bank conflicts | CU busy |\n|---|---:|---:|---:|---:|")
    for k, c in sorted(agg.items(), key=lambda kv: -ns[kv[0]]):
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        if cyc <= 0 or ns[k] < 20000:
            continue
        print(f"| {k} | {ns[k] / 1e6:.3f} | {c['SQ_LDS_IDX_ACTIVE'] / 256.0 / cyc:.2f} | {c['SQ_LDS_BANK_CONFLICT'] / 256.0 / cyc:.2f} | {c['SQ_BUSY_CU_CYCLES'] / 256.0 / cyc:.2f} |")
    print()


if __name__ == "__main__":
    main()
