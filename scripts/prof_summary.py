#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats run (gpurun_out/<dir>) into profiles/<name>.md: one row per kernel
with calls / total / average / share, kernel names shortened.  Usage: prof_summary.py gpurun_out/prof_x profiles/r01_x.md [log]"""
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(.*", "", n)
    if "rocprim" in n:
        m = re.search(r"wrapped_(\w+?)_config", n)
        return "rocprim::" + (m.group(1) if m else "kernel")
    return n.replace("void ", "")


def main():
    src, dst = sys.argv[1], sys.argv[2]
    f = glob.glob(src + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0])
        a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    with open(dst, "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats summary\n\nsource: `{f}`\n\n")
        if len(sys.argv) > 3:
            for line in open(sys.argv[3]):
                if line.startswith("{"):
                    o.write("bench line of the profiled run:\n\n```json\n" + line.strip() + "\n```\n\n")
        o.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write(f"| {k} | {c} | {t / 1e6:.3f} | {t / c / 1e3:.2f} | {100.0 * t / tot:.2f} |\n")


if __name__ == "__main__":
    main()
