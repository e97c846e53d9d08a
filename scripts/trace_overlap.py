#!/usr/bin/env python3
"""Busy-time analysis of a rocprofv3 --kernel-trace run of the default (overlapped) bench: over the last `frac` of the traced interval, the union of all kernel
intervals (GPU has at least one kernel in flight), the sum of durations (overlap factor) and the idle gaps.  Usage: trace_overlap.py <dir> [frac]"""
import csv, glob, sys
d = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r["Queue_Id"]) for r in csv.DictReader(open(f))]
t0 = min(r[0] for r in rows); t1 = max(r[1] for r in rows)
cut = t1 - frac * (t1 - t0)
sel = sorted(r for r in rows if r[0] >= cut)
tot = sum(e - s for s, e, _, _ in sel)
un = 0; cs, ce = sel[0][0], sel[0][1]; gaps = []
for s, e, _, _ in sel[1:]:
    if s <= ce: ce = max(ce, e)
    else: gaps.append(s - ce); un += ce - cs; cs, ce = s, e
un += ce - cs
wall = sel[-1][1] - sel[0][0]
print(f"window {wall/1e6:.2f} ms: union busy {un/1e6:.2f} ms ({100*un/wall:.1f} %), sum of kernel durations {tot/1e6:.2f} ms (x{tot/un:.2f} overlap), {len(gaps)} idle gaps totalling {sum(gaps)/1e6:.3f} ms (largest {max(gaps or [0])/1e3:.1f} us)")
qs = {}
for s, e, n, q in sel: qs.setdefault(q, []).append(e - s)
for q, v in sorted(qs.items()): print(f"  queue {q}: {len(v)} kernels, {sum(v)/1e6:.2f} ms")
