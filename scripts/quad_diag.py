import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import semantic_slam_mapping_amd as ssm
from oracle.binding import Oracle
o = Oracle()
ctx = ssm.Context(0, orb_features=1000, max_batch=1, voxel_capacity_log2=10)
g0 = o.bgr2gray(o.synth_frame(0x5EED0000, 0)[0]); lc = np.tile(g0, (1, 2))[:376, :1241].copy()
for mc, q, md in ((200, 0.01, 5.0), (5000, 0.1, 12.0)):
    g = ctx.gftt(lc, mc, q, md); r = o.gftt(lc, mc, q, md)
    n = min(len(g), len(r)); bad = np.where((g[:n] != r[:n]).any(1))[0]
    print(mc, q, md, len(g), len(r), 'first mismatch', bad[:1])
    if len(bad):
        i = bad[0]; e = o.min_eigen_map(lc)
        print(' gpu', g[i-1:i+3].tolist(), '\n ref', r[i-1:i+3].tolist())
        for p in list(g[i:i+2]) + list(r[i:i+2]):
            print('  eig at', p, e[int(p[1]), int(p[0])])
