#!/usr/bin/env python3
"""How far do pose chains of independent sequences overlap?  T trackers (own_stream = 1) from T host threads, one 20-frame sequence each.
Usage (GPU box): python scripts/tracker_concurrency.py [T ...]"""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")      # HIP multiplexes streams onto 4 hardware queues by default; two chains on one queue run one after the other.  GPU_MAX_HW_QUEUES=4 shows that
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semantic_slam_mapping_amd as ssm                      # noqa: E402
from semantic_slam_mapping_amd._lib import SeqOutDev          # noqa: E402

CAM = (318.6, 255.3, 517.3, 516.5, 1000.0)
CH = 20


def main():
    Ts = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    NS = max(Ts)
    c = ssm.Context(0, orb_features=1000, max_batch=40, camera=CAM)
    # frame 0 of the bench's synthetic stream, made on the device
    H, W = 480, 640
    tb = c.dev_alloc(H * W * 3); td = c.dev_alloc(H * W * 2); ts = c.dev_alloc(H * W * 3); tp = c.dev_alloc(128)
    c.synth_frames_dev(0x5EED0000, 0, 1, tb, td, ts, tp); c.sync()
    base = c.d2h(tb, (H, W, 3), np.uint8)
    for p_ in (tb, td, ts, tp):
        c.dev_free(p_)
    bgr = np.stack([np.roll(base, (k % CH, 2 * (k % CH)), (0, 1)) for k in range(NS * CH)])
    dep = np.full((NS * CH, 480, 640), 2000, np.uint16)
    db = c.dev_alloc(bgr.nbytes); dd = c.dev_alloc(dep.nbytes)
    c.h2d(db, bgr); c.h2d(dd, dep)
    o = c.seq_process(db, dd, None, None, NS * CH, stages=ssm.api.STAGE_ORB | ssm.api.STAGE_MATCH)
    c.sync()
    view = lambda a0: SeqOutDev(o.kps + a0 * o.cap * 28, o.desc + a0 * o.cap * 32, o.pos3d + a0 * o.cap * 12, o.nkp + a0 * 4, o.matches + a0 * o.R * o.cap * 16,
                                o.nmatch + a0 * o.R * 4, o.npoints + a0 * 4, o.cap, o.R)
    for T in Ts:
        trk = [ssm.Tracker(c, use_device=True, own_stream=True) for _ in range(T)]
        for k in range(T):
            trk[k].run(view(k * CH), CH); trk[k].reset()                # warm: scratch allocated
        spans = [None] * T
        def walk(k):
            t0 = time.perf_counter(); trk[k].run(view(k * CH), CH); spans[k] = (t0, time.perf_counter())
        th = [threading.Thread(target=walk, args=(k,)) for k in range(T)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        wall = time.perf_counter() - t0
        runs = [(b - a) * 1e3 for a, b in spans]
        print(f"T={T}: wall {wall * 1e3:.1f} ms, per-run {min(runs):.1f} .. {max(runs):.1f} ms, {T * CH / wall:.0f} frames/s")
        for t in trk: t.close()
    c.close()


if __name__ == "__main__":
    main()
