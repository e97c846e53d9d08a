#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats run (gpurun_out/<dir>) into profiles/<name>.md.

Per kernel, from the per-dispatch trace (`*kernel_trace.csv`): calls, total time, the plain average, and the numbers a roofline should be
recomputed from -- MEDIAN and MINIMUM launch duration and the STEADY average (of every kernel's launches, in start order, the first `--warm`
fraction is dropped: the first pass of a bench run pays cold caches / first-touch page faults / code upload, up to 2.5x on the long kernels; the
profiled bench commands run 1 warm-up + 3 timed passes (`--warm 0.25`, the default) or 1 + 2 (`--warm 0.34`)) -- plus the launch geometry (grid in
workgroups x workgroup size, LDS, registers) of the most frequent launch shape.  Kernels that run side by side on several queues (the SGBM stage,
the two SegNet blocks ...) are not additive: the per-STAGE table gives, for every group of kernels in `STAGES`, the union of their busy intervals.
Template arguments are demangled and kept (they distinguish the instantiations of one kernel).

Usage: prof_summary.py gpurun_out/prof_x profiles/r04_x.md [bench log] [--frames-per-launch N] [--warm 0.34]"""
import csv
import glob
import re
import statistics
import sys

STAGES = [  # (stage name, regex on the short kernel name); first match wins
    ("sgbm", r"^sgbm_"), ("quad", r"^(gftt_|mineig|pyrdown|scharr|lk_|filter_tracks|window_match)"), ("vo", r"^vo_"),
    ("segnet", r"^(conv3x3|segnet_|unpool2x2|label_color)"), ("fast", r"^fast_"), ("pyramid", r"^resize"), ("gray", r"^gray"),
    ("octree", r"^octree"), ("blur", r"^blur"), ("describe", r"^(kp_prepare|orient|angle|brief)"), ("match", r"^match_"),
    ("map_fuse", r"^(map_stream|class_bits|vdilate)"), ("map_export", r"^(vox_|rocprim|k_voxel|voxel_)"), ("pnp", r"^pnp_"), ("synth", r"^synth"),
]


def demangle(n):
    """rocprofv3's trace holds mangled names for template kernels, and binutils' c++filt stops at _Float16 parameters (DF16_): the kernel name and its
    literal template arguments (bool / int) are all that is needed"""
    m = re.match(r"_Z(\d+)", n)
    if not m:
        return n
    ln = int(m.group(1)); p = m.end()
    name, rest = n[p:p + ln], n[p + ln:]
    if not rest.startswith("I"):
        return name
    args, q = [], 1
    while q < len(rest) and rest[q] != "E":
        a = re.match(r"L([a-z])(n?)(\d+)E", rest[q:])
        if not a:
            return name + "<...>"
        v = ("-" if a.group(2) else "") + a.group(3)
        args.append({"0": "false", "1": "true"}.get(v, v) if a.group(1) == "b" else v)
        q += a.end()
    return name + "<" + ", ".join(args) + ">"


_SHORT = {}


def short(n):
    if n in _SHORT:
        return _SHORT[n]
    r = _short(n)
    _SHORT[n] = r
    return r


def _short(n):
    n = demangle(n.strip())
    if "rocprim" in n:
        m = re.search(r"wrapped_(\w+?)_config", n)
        return "rocprim::" + (m.group(1) if m else "kernel")
    n = re.sub(r"^void ", "", n)
    # drop the argument list (the last top-level parenthesis group), keep template arguments
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            n = n[:i]
            break
    return n.replace("(bool)1", "true").replace("(bool)0", "false")


def stage_of(k):
    for name, pat in STAGES:
        if re.search(pat, k):
            return name
    return "other"


def union_ns(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = sys.argv[1:]
    src, dst = args[0], args[1]
    warm = 0.25
    fpl = None
    for i, a in enumerate(opts):
        if a == "--warm":
            warm = float(opts[i + 1])
        if a == "--frames-per-launch":
            fpl = int(opts[i + 1])
    args = [a for a in args if not re.fullmatch(r"[0-9.]+", a)]
    f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    per = {}
    for r in rows:
        k = short(r["Kernel_Name"])
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        wg = [int(r[f"Workgroup_Size_{a}"]) for a in "XYZ"]
        gr = [int(r[f"Grid_Size_{a}"]) for a in "XYZ"]
        shape = (tuple(g // max(w, 1) for g, w in zip(gr, wg)), wg[0] * wg[1] * wg[2], int(r["LDS_Block_Size"]), int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]), int(r["SGPR_Count"]))
        per.setdefault(k, []).append((s, e, shape))
    tot = sum(e - s for v in per.values() for s, e, _ in v)
    import math
    steady = {}
    for k, v in per.items():
        v.sort()
        ndrop = math.ceil(len(v) * warm - 1e-9) if len(v) >= 3 else 0
        steady[k] = v[ndrop:] or v
    big = max(per, key=lambda k: sum(e - s for s, e, _ in per[k]) if len(per[k]) >= 3 else 0)
    cut = steady[big][0][0]                                   # the stage table's steady part: from the first kept launch of the largest kernel
    with open(dst, "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats summary\n\nsource: `{f}` ({len(rows)} dispatches over {(t1 - t0) / 1e6:.1f} ms)\n\n")
        if len(args) > 2:
            for line in open(args[2]):
                if line.startswith("{"):
                    o.write("bench line of the profiled run:\n\n```json\n" + line.strip() + "\n```\n\n")
        o.write(f"`steady us` = average without the first {warm:.0%} of each kernel's launches (the cold pass: first-touch, cold caches, code "
                "upload; kernels with fewer than 3 launches: all); `median` / `min` over all launches.  Recompute rooflines from `steady` or `median`, not from `avg`.  `grid x wg` = workgroups x threads of the "
                "most frequent launch shape" + (f"; frames per launch of the batched kernels: {fpl}" if fpl else "") + ".\n\n")
        o.write("| kernel | calls | total ms | avg us | steady us | median us | min us | % | grid x wg | LDS B | VGPR | SGPR |\n|---|---:|---:|---:|---:|---:|---:|---:|---|---:|---:|---:|\n")
        for k, v in sorted(per.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
            d = [e - s for s, e, _ in v]
            st = [e - s for s, e, _ in steady[k]]
            shapes = {}
            for _, _, sh in v:
                shapes[sh] = shapes.get(sh, 0) + 1
            sh = max(shapes.items(), key=lambda kv: kv[1])[0]
            g = "x".join(str(x) for x in sh[0] if x > 1) or "1"
            o.write(f"| {k} | {len(d)} | {sum(d) / 1e6:.3f} | {sum(d) / len(d) / 1e3:.2f} | {sum(st) / len(st) / 1e3:.2f} | {statistics.median(d) / 1e3:.2f} | {min(d) / 1e3:.2f} | "
                    f"{100.0 * sum(d) / tot:.2f} | {g} x {sh[1]} | {sh[2]} | {sh[3]} | {sh[4]} |\n")
        # per stage: sum of kernel times vs union of busy intervals (kernels of a stage may run side by side on several queues)
        stages = {}
        for k, v in per.items():
            stages.setdefault(stage_of(k), []).extend((s, e) for s, e, _ in v)
        o.write("\n## per stage: sum of kernel durations vs union of their busy intervals (side-by-side kernels are not additive)\n\n")
        o.write("| stage | launches | sum ms | union ms | union ms, steady part | overlap factor |\n|---|---:|---:|---:|---:|---:|\n")
        for sname, iv in sorted(stages.items(), key=lambda kv: -union_ns(kv[1])):
            sm, un = sum(e - s for s, e in iv), union_ns(iv)
            uns = union_ns([(s, e) for s, e in iv if s >= cut])
            o.write(f"| {sname} | {len(iv)} | {sm / 1e6:.3f} | {un / 1e6:.3f} | {uns / 1e6:.3f} | {sm / max(un, 1):.2f} |\n")


if __name__ == "__main__":
    main()
