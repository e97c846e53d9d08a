#!/usr/bin/env python3
"""Soak test of the SGBM sweep's strip hand-offs (kernels_sgbm.hip: tagged granules between the strips of a frame) under the load of the whole batched stereo path:
`rounds` x ssm_stereo_seq_process over `frames` KITTI-size pairs (quad matcher, VO and the second SGBM stream running beside it), every disparity image hashed and
compared with the hashes of the FIRST round and of the four-volume form of round 3 (SSM_SGBM_FORM=1, run in a child process: the form is read once per process).
A stale or torn hand-off would change a disparity somewhere.  Usage (GPU box): python3 scripts/sgbm_soak.py [rounds] [frames]"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(rounds, frames):
    import numpy as np
    import semantic_slam_mapping_amd as ssm
    sys.path.insert(0, ROOT)
    import bench
    W, H = 1241, 376
    ctx = ssm.Context(0, width=640, height=480, max_batch=64)
    L, R = bench.stereo_sequence(frames, W, H, 100)
    dl = ctx.dev_alloc(L.nbytes); dr = ctx.dev_alloc(R.nbytes); ds = ctx.dev_alloc(frames * 200 * 3 * 4)
    ctx.h2d(dl, L); ctx.h2d(dr, R); ctx.h2d(ds, ssm.api.GlibcRand(0).draws(frames * 200 * 3))
    vo = (bench.KITTI["f"], bench.KITTI["cu"], bench.KITTI["cv"], bench.KITTI["baseline"], 2.0, True)
    out_hashes = []
    for r in range(rounds):
        out = ctx.stereo_seq_process(dl, dr, frames, W, H, vo=vo, ransac_iters=200, rand_stream_dev=ds, **bench.KITTI)
        ctx.sync()
        res = ctx.stereo_seq_fetch(out, frames, W, H)
        out_hashes.append([hashlib.sha1(res["disp"][f].tobytes()).hexdigest()[:16] for f in range(frames)])
    ctx.close()
    return out_hashes


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        print(json.dumps(run(int(sys.argv[2]), int(sys.argv[3]))))
        sys.exit(0)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    ref = json.loads(subprocess.run([sys.executable, __file__, "--child", "1", str(frames)], env=dict(os.environ, SSM_SGBM_FORM="1"), capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])[0]
    bad = 0
    for lanes in ("8", "16"):
        hs = json.loads(subprocess.run([sys.executable, __file__, "--child", str(rounds), str(frames)], env=dict(os.environ, SSM_SGBM_FORM="2", SSM_SGBM_SWEEP_LANES=lanes), capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
        for r, h in enumerate(hs):
            diff = [f for f in range(frames) if h[f] != ref[f]]
            if diff:
                bad += 1; print(f"lanes {lanes} round {r}: {len(diff)} frames differ from the four-volume form: {diff[:8]}")
        print(f"sweep with {lanes} lanes per pixel: {rounds} rounds x {frames} pairs, {sum(1 for h in hs if h == ref)} rounds identical to the four-volume form")
    sys.exit(1 if bad else 0)
