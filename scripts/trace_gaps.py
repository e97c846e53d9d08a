#!/usr/bin/env python3
"""Idle time of the GPU inside a rocprofv3 --kernel-trace run: the union of the kernels' busy intervals against the wall span, per step or for the whole trace.
Usage: trace_gaps.py <rocprof output dir> [marker kernel | none]   (marker: the kernel a step starts with, default vox_clear_kernel = ssm_map_clear)"""
import csv, sys, glob
fn = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else 'vox_clear_kernel'
rows = list(csv.DictReader(open(fn)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40]) for r in rows)
ev = [e for e in ev if 'synth' not in e[2]]
def report(seg, nxt, label):
    s0 = seg[0][0]; e1 = max(x[1] for x in seg)
    cur_s, cur_e = seg[0][0], seg[0][1]; busy = 0; gaps = []
    for s, e, n in seg[1:]:
        if s > cur_e: busy += cur_e - cur_s; gaps.append((s - cur_e, n, (s - s0) / 1e3)); cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps.sort(reverse=True)
    print('%s: %d kernels, span %.2f ms, busy %.2f ms (%.1f %%), to next start %.2f ms; gaps > 20 us: %d = %.2f ms; top gaps (us, before kernel, at us):' %
          (label, len(seg), (e1 - s0) / 1e6, busy / 1e6, 100.0 * busy / max(e1 - s0, 1), (nxt - s0) / 1e6, sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6),
          [(round(g / 1e3, 1), n[:24], round(t)) for g, n, t in gaps[:6]])
if marker == 'none':
    report(ev, ev[-1][1], 'trace')
else:
    starts = [i for i, e in enumerate(ev) if e[2].startswith(marker) and (i == 0 or not ev[i - 1][2].startswith(marker))]
    print('steps (groups of %s):' % marker, len(starts))
    for a, b in zip(starts, starts[1:] + [len(ev)]):
        if b - a < 40: continue
        report(ev[a:b], ev[b][0] if b < len(ev) else max(x[1] for x in ev[a:b]), 'step')
