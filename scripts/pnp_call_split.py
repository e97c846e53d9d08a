#!/usr/bin/env python3
"""ssm_pnp_solve per call, split into the kernel (hipEvents around the launch: ssm_set_profiling) and the rest (three uploads, launch, downloads, the wait), for the
cluster of eight blocks (round 6) and one block (SSM_PNP_BLOCKS=1).  Usage (GPU box): python scripts/pnp_call_split.py [n_correspondences] -> one markdown row per form."""
import os
import subprocess
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(n):
    import semantic_slam_mapping_amd as ssm
    from test_pnp import _case, _pose, CAM
    c = ssm.Context(0, width=640, height=480, max_batch=1)
    img, obj, _ = _case(10, n, 10, 7, 0.5)
    kcam = (CAM[2], CAM[3], CAM[0], CAM[1]); T0 = _pose(0.01, 0.0, -0.01, (0.01, 0.0, 0.02))
    for _ in range(5):
        c.pnp_solve(img, obj, kcam, T0)
    ts = []
    for _ in range(50):
        t = time.perf_counter(); c.pnp_solve(img, obj, kcam, T0); ts.append(time.perf_counter() - t)
    c.set_profiling(1); ks = []
    for _ in range(20):
        c.pnp_solve(img, obj, kcam, T0); ks.append(c.stage_times()["pnp"][0])
    c.set_profiling(0); c.close()
    ts.sort(); ks.sort()
    print("| %s | %d | %.3f | %.3f | %.3f |" % (os.environ.get("SSM_PNP_BLOCKS", "8"), n, ts[len(ts) // 2] * 1e3, ks[len(ks) // 2], ts[len(ts) // 2] * 1e3 - ks[len(ks) // 2]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "child":
        one(int(sys.argv[1]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
        print("| blocks | correspondences | call ms (median of 50) | kernel ms (hipEvents) | copies + launch + wait ms |\n|---|---:|---:|---:|---:|", flush=True)
        for b in ("8", "1"):
            subprocess.run([sys.executable, os.path.abspath(__file__), str(n), "child"], env=dict(os.environ, SSM_PNP_BLOCKS=b), check=True)
