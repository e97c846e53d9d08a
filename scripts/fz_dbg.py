import numpy as np, sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import semantic_slam_mapping_amd as ssm
from oracle.binding import Oracle, build
build(); oracle = Oracle()
from test_gpu_fuzz import _texture
ctx = ssm.Context(0, width=640, height=480, max_batch=1)
def run(seed, h, w, nd, sad, min_d=0, uniq=0, noise=0, smooth=1):
    rng = np.random.default_rng(seed)
    tex = _texture(rng, h, w + 2 * nd + 40, smooth)
    right = tex[:, nd + 20:nd + 20 + w].copy(); left = tex[:, 20 + 6:20 + 6 + w].copy()
    po = oracle.sgbm_params(num_disp=nd, sad=sad, min_disp=min_d, uniqueness=uniq, speckle_window=0)
    try:
        raw_g = ctx.sgbm(left, right, po, raw=True)
    except Exception as e:
        print(f"w {w} nd {nd} sad {sad}: GPU error {e}"); return
    raw_o = oracle.sgbm(left, right, po, raw=True)
    bad = np.argwhere(raw_g != raw_o)
    print(f"w {w} nd {nd} sad {sad} (w1 {w-nd}): differ {len(bad)} of {raw_o.size}", [(int(y), int(x), int(raw_g[y,x]), int(raw_o[y,x])) for y,x in bad[:6]])
for w in (97, 98, 99, 100, 104, 112):
    run(0, 14, w, 96, 3)
run(0, 14, 97, 96, 9); run(0, 20, 81, 80, 11); run(0, 20, 17, 16, 3); run(0, 20, 18, 16, 5); run(0, 30, 66, 64, 5)
