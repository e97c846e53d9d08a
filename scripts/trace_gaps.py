import csv, sys, glob
fn = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40]) for r in rows]
ev.sort()
ev = [e for e in ev if 'synth' not in e[2]]
# steps start at the fast_kernel following a vox_clear_kernel group; segment by gaps: a step = from first kernel after a big clear to the last kernel before the next
starts = [i for i, e in enumerate(ev) if e[2].startswith('vox_clear_kernel') and (i == 0 or not ev[i - 1][2].startswith('vox_clear_kernel'))]
print('map_clear groups:', len(starts))
segs = []
for a, b in zip(starts, starts[1:] + [len(ev)]):
    seg = ev[a:b]
    if len(seg) < 40: continue
    s0 = seg[0][0]; e1 = max(x[1] for x in seg)
    cur_s, cur_e = seg[0][0], seg[0][1]; busy = 0; gaps = []
    for s, e, n in seg[1:]:
        if s > cur_e: busy += cur_e - cur_s; gaps.append((s - cur_e, n, (s - s0) / 1e3)); cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps.sort(reverse=True)
    nxt = ev[b][0] if b < len(ev) else e1
    print('step: %d kernels, span %.2f ms, busy %.2f ms, to next step start %.2f ms; top gaps (us, before kernel, at us):' % (len(seg), (e1 - s0) / 1e6, busy / 1e6, (nxt - s0) / 1e6),
          [(round(g / 1e3, 1), n[:24], round(t)) for g, n, t in gaps[:6]])
