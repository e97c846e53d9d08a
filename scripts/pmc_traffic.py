#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per MI355X_MICROARCH.md s.HBM:
separate passes, FETCH_SIZE in KiB and x2 on gfx950 for wide coalesced reads (reported both raw and corrected;
narrow-access kernels are uncalibrated and say so).  Usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [frames_per_launch]"""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot = collections.defaultdict(float); calls = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        tot[k] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); calls[k] += 1
    return tot, calls


# Calibration (scripts/ubench/fetch_calib.hip, 1 GiB streams, profiles/r02_fetch_calib.md): FETCH_SIZE reports 0.500 of the bytes for EVERY read pattern
# these kernels use (contiguous / 32-B / 48-B strided 16-byte loads, dwords, 12-B strided dwords; 0.516 for unaligned 16-byte loads), WRITE_SIZE is exact
# for dword and 16-byte stores: so fetch x 2 + write, for every kernel.


def main():
    fd, wd, out = sys.argv[1:4]
    fpl = int(sys.argv[4]) if len(sys.argv) > 4 else 125
    fetch, fc = load(fd, "FETCH_SIZE")
    write, wc = load(wd, "WRITE_SIZE")
    # frames processed in the profiled run: every sub-batch launches fast_kernel once (bench.py runs the steps twice: the timed pass and the serialised pass)
    frames_total = int(sys.argv[5]) if len(sys.argv) > 5 else fpl * max(fc.get("fast_kernel", 0), wc.get("fast_kernel", 0), 1)
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if "rocprim" in k or "rocclr" in k:
            continue
        n = max(fc.get(k, 0), wc.get(k, 0), 1)
        f_kib, w_kib = fetch.get(k, 0.0) / n, write.get(k, 0.0) / n
        res[k] = {"launches": n, "fetch_bytes_per_launch_raw": f_kib * 1024, "fetch_bytes_per_launch_x2": 2 * f_kib * 1024,
                  "write_bytes_per_launch": w_kib * 1024, "frames_per_launch": fpl,
                  "bytes_per_frame_raw": (f_kib + w_kib) * 1024 / fpl, "bytes_per_frame_fetch_x2": (2 * f_kib + w_kib) * 1024 / fpl,
                  "total_bytes_per_frame_fetch_x2": (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024 / frames_total,
                  "total_bytes_per_frame_raw": (fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024 / frames_total}
        res[k]["total_bytes_per_frame"] = res[k]["total_bytes_per_frame_fetch_x2"]
    # the kernel sources these counters belong to: bench.py quotes a traffic file only while the source of the kernel it describes is byte for byte the same
    import hashlib, os
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "semantic_slam_mapping_amd", "csrc")
    at_collection = os.path.join(fd, "sources_sha256.json")      # written on the GPU box by scripts/collect_profiles.sh next to the counters: the sources that actually ran
    if os.path.exists(at_collection):
        sha = json.load(open(at_collection))
    else:
        sha = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest() for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".inc"))}
    json.dump({"sources_sha256": sha, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KiB units); gfx950 reports half of the bytes read "
                       "(MI355X_MICROARCH.md s.HBM; calibrated for this repo's access patterns by scripts/ubench/fetch_calib.hip): total = fetch x 2 + write",
               "frames_in_profiled_run": frames_total,
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["total_bytes_per_frame"]):
        print(f"{k[:36]:36s} launches {v['launches']:4d}  HBM bytes per frame (all launches): raw {v['total_bytes_per_frame_raw'] / 1e6:8.3f} MB   fetch x2 {v['total_bytes_per_frame_fetch_x2'] / 1e6:8.3f} MB")


if __name__ == "__main__":
    main()
