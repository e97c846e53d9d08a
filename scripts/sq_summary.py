#!/usr/bin/env python3
"""Per-kernel SQ counter table from a `rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD
SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace` run.  Usage: sq_summary.py gpurun_out/<dir> profiles/<name>.md "<title>"
VALU IPC/SIMD = SQ_INSTS_VALU / (duration x clock x 1024 SIMDs) with clock = SQ_BUSY_CYCLES / 32 shader engines / duration."""
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
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); ns = collections.defaultdict(float); seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if "rocprim" in k or k.startswith("__amd"):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); ns[k] += dur.get(r["Dispatch_Id"], 0)
    import json
    frames = int(sys.argv[4]) if len(sys.argv) > 4 else 2000            # frames processed in the profiled run
    js = {"note": "rocprofv3 --pmc SQ_INSTS_VALU per kernel: wave-instructions; x64 = lane-ops", "frames_in_run": frames, "kernels": {}}
    for k, c in agg.items():
        js["kernels"][k] = {"valu_wave_insts_per_frame": c["SQ_INSTS_VALU"] / frames, "waves_per_frame": c["SQ_WAVES"] / frames}
    json.dump(js, open(dst.rsplit(".", 1)[0] + ".json", "w"), indent=1)
    with open(dst, "w") as o:
        o.write(f"# {title}\n\nper-wave averages; clock = SQ_BUSY_CYCLES / 32 / duration; `VALU IPC/SIMD` = SQ_INSTS_VALU / (duration x clock x 1024 SIMDs); "
                "the issue ceiling for these instruction mixes is 0.25 (one wave64 instruction per 4 clk per SIMD: profiles/r02_valu_rate.md)\n\n")
        o.write("| kernel | ms | waves | VALU/wave | LDS/wave | SALU/wave | VMEM_RD/wave | cycles/wave | clock GHz | VALU IPC/SIMD |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        for k, c in sorted(agg.items(), key=lambda kv: -ns[kv[0]]):
            w = max(c["SQ_WAVES"], 1.0); t = ns[k] * 1e-9
            clk = c["SQ_BUSY_CYCLES"] / 32.0 / t if t > 0 else 0.0
            ipc = c["SQ_INSTS_VALU"] / (t * clk * 1024.0) if t > 0 and clk > 0 else 0.0
            o.write(f"| {k} | {ns[k] / 1e6:.3f} | {int(w)} | {c['SQ_INSTS_VALU'] / w:.0f} | {c['SQ_INSTS_LDS'] / w:.0f} | {c['SQ_INSTS_SALU'] / w:.0f} | "
                    f"{c['SQ_INSTS_VMEM_RD'] / w:.1f} | {c['SQ_WAVE_CYCLES'] / w:.0f} | {clk / 1e9:.2f} | {ipc:.3f} |\n")


if __name__ == "__main__":
    main()
