#!/usr/bin/env python3
"""Per-LAYER table of the SegNet stage (configs[2]): layer name, kernel instantiation (demangled template arguments), duration per launch, TFLOP/s,
held clock and MFMA-pipe utilisation -- /root/reference/src/segnet.cpp:99 is one opaque `net_->ForwardPrefilled()`, so this table is the only place where
the stage's time can be read per layer.

Input: a `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv` run of
`bench.py --segnet ...` (counters optional: a plain --kernel-trace run gives the time columns only).  The conv launches of a forward pass are walked in
launch order and named after the network definition (VGG-16 encoder / mirrored decoder); medians over the forward passes, the first pass
dropped.  clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x clock x duration).

Round 6: with `--fetch <dir> --write <dir>` (the FETCH_SIZE / WRITE_SIZE passes of scripts/collect_profiles.sh pmc_segnet, which carry a kernel trace too) the table gets
each layer's BYTE columns: algorithmic bytes per frame (input activation once -- the pooled tensor + codes for an un-pool-on-load layer --, output once -- pooled + codes
for a pool layer, labels for the ArgMax layer --, weights once per launch), the time those bytes take at 8 TB/s (the layer's byte roof), the bytes measured
(FETCH x 2 + WRITE, MI355X_MICROARCH.md's gfx950 correction) and how the measured reads split: weights once per launch, the input once, and the rest = halo overlap of the
32 x 16 tiles (x 1.195) and the re-reads of the input by the layer's Cout / 64 cout tiles that missed L2 / MALL.

Usage: segnet_layers.py <rocprof dir> <frames per launch> [out.md] [--fetch <dir> --write <dir>]"""
import collections
import csv
import glob
import re
import statistics
import sys

# (name, Cin, Cout, H, W) in launch order
ENC = [("conv1_1", 3, 64, 360, 480), ("conv1_2 + pool", 64, 64, 360, 480), ("conv2_1", 64, 128, 180, 240), ("conv2_2 + pool", 128, 128, 180, 240),
       ("conv3_1", 128, 256, 90, 120), ("conv3_2", 256, 256, 90, 120), ("conv3_3 + pool", 256, 256, 90, 120),
       ("conv4_1", 256, 512, 45, 60), ("conv4_2", 512, 512, 45, 60), ("conv4_3 + pool", 512, 512, 45, 60),
       ("conv5_1", 512, 512, 23, 30), ("conv5_2", 512, 512, 23, 30), ("conv5_3 + pool", 512, 512, 23, 30)]
DEC = [("conv5_3_D", 512, 512, 23, 30), ("conv5_2_D", 512, 512, 23, 30), ("conv5_1_D", 512, 512, 23, 30),
       ("conv4_3_D", 512, 512, 45, 60), ("conv4_2_D", 512, 512, 45, 60), ("conv4_1_D", 512, 256, 45, 60),
       ("conv3_3_D (un-pool on load)", 256, 256, 90, 120), ("conv3_2_D", 256, 256, 90, 120), ("conv3_1_D", 256, 128, 90, 120),
       ("conv2_2_D (un-pool on load)", 128, 128, 180, 240), ("conv2_1_D", 128, 64, 180, 240),
       ("conv1_2_D (un-pool on load)", 64, 64, 360, 480), ("conv1_1_D 64->12 + ArgMax", 64, 12, 360, 480)]
LAYERS = ENC + DEC


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


def short(n):
    n = re.sub(r"^void ", "", demangle(n.strip()))
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


def per_dispatch(d, counter):
    """{dispatch id: value} and the conv / helper dispatches of the run in launch order, split into forward passes"""
    tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    val = collections.defaultdict(float)
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] == counter:
            val[r["Dispatch_Id"]] += float(r["Counter_Value"])
    rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
    seg = [r for r in rows if any(k in r["Kernel_Name"] for k in ("conv3x3", "unpool2x2", "segnet_prep", "label_color"))]
    passes, cur = [], []
    for r in seg:
        if "segnet_prep" in r["Kernel_Name"] and cur:
            passes.append(cur); cur = []
        cur.append(r)
    if cur:
        passes.append(cur)
    full = max(len(p) for p in passes)
    passes = [p for p in passes if len(p) == full]
    return [[val.get(r["Dispatch_Id"], 0.0) for r in p] for p in passes]


def main():
    argv = list(sys.argv)
    fetch_dir = write_dir = None
    if "--fetch" in argv:
        i = argv.index("--fetch"); fetch_dir = argv[i + 1]; del argv[i:i + 2]
    if "--write" in argv:
        i = argv.index("--write"); write_dir = argv[i + 1]; del argv[i:i + 2]
    d, fpl = argv[1], int(argv[2])
    out = open(argv[3], "w") if len(argv) > 3 else sys.stdout
    traffic = None
    if fetch_dir and write_dir:
        fp, wp = per_dispatch(fetch_dir, "FETCH_SIZE"), per_dispatch(write_dir, "WRITE_SIZE")
        n = min(len(fp[0]), len(wp[0]))
        traffic = [(statistics.median(p[i] for p in fp) * 1024 * 2, statistics.median(p[i] for p in wp) * 1024) for i in range(n)]      # bytes per launch: reads (x 2), writes
    tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    val = collections.defaultdict(dict)
    if cc:
        for r in csv.DictReader(open(cc[0])):
            val[r["Dispatch_Id"]][r["Counter_Name"]] = val[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
    seg = [r for r in rows if any(k in r["Kernel_Name"] for k in ("conv3x3", "unpool2x2", "segnet_prep", "label_color"))]
    passes, cur = [], []
    for r in seg:
        if "segnet_prep" in r["Kernel_Name"] and cur:
            passes.append(cur); cur = []
        cur.append(r)
    if cur:
        passes.append(cur)
    full = max(len(p) for p in passes)
    passes = [p for p in passes if len(p) == full]
    if len(passes) > 1:
        passes = passes[1:]                                   # the first forward pass is cold
    tab = []
    for i in range(full):
        rs = [p[i] for p in passes]
        t = statistics.median((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rs) * 1e-9
        clk = busy = None
        cs = [val.get(r["Dispatch_Id"]) for r in rs if val.get(r["Dispatch_Id"])]
        if cs:
            clks, busys = [], []
            for r in rs:
                c = val.get(r["Dispatch_Id"])
                if not c:
                    continue
                tt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
                ck = c.get("GRBM_GUI_ACTIVE", 0) / 8 / tt
                clks.append(ck)
                if ck:
                    busys.append(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * ck * tt))
            clk = statistics.median(clks) if clks else None
            busy = statistics.median(busys) if busys else None
        r0 = rs[0]
        wg = int(r0["Workgroup_Size_X"]) * int(r0["Workgroup_Size_Y"]) * int(r0["Workgroup_Size_Z"])
        grid = int(r0["Grid_Size_X"]) * int(r0["Grid_Size_Y"]) * int(r0["Grid_Size_Z"]) // max(wg, 1)
        tab.append((short(r0["Kernel_Name"]), t, clk, busy, grid, wg))
    li = 0
    total_t = sum(x[1] for x in tab)
    total_f = 0.0
    out.write(f"# SegNet stage per layer (configs[2]): time, MFMA utilisation" + (", bytes against the HBM roof" if traffic else "") + f"\n\n"
              f"`rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace` of `bench.py --segnet --frames 256 --batch 128` ({fpl} frames per SegNet launch); "
              f"clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x clock x duration)" +
              ("; bytes: FETCH_SIZE x 2 + WRITE_SIZE of separate passes, per frame; byte roof = algorithmic bytes / 8 TB/s" if traffic else "") + ".  scripts/segnet_layers.py\n\n")
    hdr = f"| # | layer | kernel | workgroups x threads | us per launch ({fpl} frames) | us per frame | GFLOP per frame | TFLOP/s | clock GHz | MFMA busy | share of the stage |"
    sep = "|---:|---|---|---|---:|---:|---:|---:|---:|---:|---:|"
    if traffic:
        hdr += " algorithmic MB per frame (in + out + weights / launch) | byte roof us per frame | measured MB per frame (read x 2 + write) | of the reads: weights / input once / halo + re-reads MB |"
        sep += "---:|---:|---:|---|"
    out.write(hdr + "\n" + sep + "\n")
    tot_alg = tot_meas = tot_w = tot_in = tot_rest = 0.0
    for i, (k, t, clk, busy, grid, wg) in enumerate(tab):
        if "conv3x3" in k and li < len(LAYERS):
            name, ci, co, h, w = LAYERS[li]; li += 1
            gf = 2.0 * 9 * ci * co * h * w / 1e9
            total_f += gf
            lname, gfs, tf = f"{name} {ci}->{co} @{h}x{w}", f"{gf:.2f}", f"{gf * fpl / t / 1e3:.0f}"
        else:
            lname, gfs, tf = ("pre-processing (resize + planar fp16)" if "prep" in k else "un-pool (512 channels)" if "unpool" in k else "label colouring (id resize + palette)"), "", ""
        line = (f"| {i} | {lname} | `{k}` | {grid} x {wg} | {t * 1e6:.1f} | {t * 1e6 / fpl:.2f} | {gfs} | {tf} | " +
                (f"{clk / 1e9:.2f}" if clk else "") + " | " + (f"{busy:.3f}" if busy is not None else "") + f" | {100 * t / total_t:.1f} % |")
        if traffic and i < len(traffic):
            rd, wr = traffic[i][0] / fpl, traffic[i][1] / fpl
            if "conv3x3" in k:
                cs = lambda c: (c + 31) // 32 * 32                      # activations live in 32-channel chunks
                up, pool, amax = "un-pool on load" in name, "+ pool" in name, "ArgMax" in name
                ph, pw = (h + 1) // 2, (w + 1) // 2
                cin_s = 8 if ci == 3 else cs(ci)
                b_in = (ph * pw * cin_s * 3) if up else h * w * cin_s * 2       # pooled fp16 + 1 code byte per element
                b_out = h * w if amax else (ph * pw * cs(co) * 3 if pool else h * w * cs(co) * 2)
                b_wt = 9 * cin_s * ((co + 63) // 64 * 64) * 2
                alg = b_in + b_out + b_wt / fpl
                rest = max(rd - b_in - b_wt / fpl, 0.0)
                line += f" {alg / 1e6:.2f} | {alg / 8e12 * 1e6:.2f} | {(rd + wr) / 1e6:.2f} | {b_wt / fpl / 1e6:.2f} / {b_in / 1e6:.2f} / {rest / 1e6:.2f} |"
                tot_alg += alg; tot_w += b_wt / fpl; tot_in += b_in; tot_rest += rest
            else:
                line += f" | | {(rd + wr) / 1e6:.2f} | |"
            tot_meas += rd + wr
        out.write(line + "\n")
    out.write(f"\nstage total: {total_t * 1e6:.1f} us per launch = {total_t * 1e6 / fpl:.2f} us per frame; {total_f:.2f} GFLOP per frame -> {total_f * fpl / total_t / 1e3:.0f} TFLOP/s "
              f"({total_f * fpl / total_t / 1e3 / 2500:.3f} of the 2.5 PFLOP/s dense fp16 peak); medians over {len(passes)} forward passes\n")
    if traffic:
        out.write(f"\nbytes per frame: algorithmic {tot_alg / 1e6:.0f} MB (= {tot_alg / 8e12 * 1e6:.1f} us at 8 TB/s), measured {tot_meas / 1e6:.0f} MB ({tot_meas / tot_alg:.2f} x) = "
                  f"{tot_meas / (total_t / fpl) / 1e12:.2f} TB/s over the stage's time.  Of the conv layers' reads: weights {tot_w / 1e6:.1f} MB (once per launch of {fpl} frames), "
                  f"the inputs once {tot_in / 1e6:.0f} MB, halo overlap + re-reads by further cout tiles that missed L2 / MALL {tot_rest / 1e6:.0f} MB.\n")


if __name__ == "__main__":
    main()
