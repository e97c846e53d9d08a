"""SegNet driving_webdemo topology (26 conv3x3 layers, 5 pool / 5 unpool) and seeded He-normal weights for tests and
bench.py -- the reference's .caffemodel is not in its tree (README.md:25-32).  numpy only."""
import numpy as np

LAYERS = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256), (256, 512), (512, 512), (512, 512),
          (512, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 256),
          (256, 256), (256, 256), (256, 128), (128, 128), (128, 64), (64, 64), (64, 12)]
# op list: ints = conv layer index, ("pool", i) / ("unpool", i, H, W)
OPS = [0, 1, ("pool", 0), 2, 3, ("pool", 1), 4, 5, 6, ("pool", 2), 7, 8, 9, ("pool", 3), 10, 11, 12, ("pool", 4),
       ("unpool", 4, 23, 30), 13, 14, 15, ("unpool", 3, 45, 60), 16, 17, 18, ("unpool", 2, 90, 120), 19, 20, 21,
       ("unpool", 1, 180, 240), 22, 23, ("unpool", 0, 360, 480), 24, 25]


def make_weights(seed=1234):
    """He-normal conv weights (Caffe blob order [Cout][Cin][3][3]); folded BN: scale ~ 1, shift ~ 0 (seeded)"""
    rng = np.random.default_rng(seed)
    out = []
    for cin, cout in LAYERS:
        w = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
        if cin == 3:
            w /= 128.0                                  # input is 0..255 with mean 0 (src/segnet.cpp:84)
        scale = (1.0 + 0.05 * rng.standard_normal(cout)).astype(np.float32)
        shift = (0.05 * rng.standard_normal(cout)).astype(np.float32)
        out.append((w, scale, shift))
    return out


def flops():
    hw = {0: (360, 480), 1: (360, 480), 2: (180, 240), 3: (180, 240), 4: (90, 120), 5: (90, 120), 6: (90, 120), 7: (45, 60), 8: (45, 60), 9: (45, 60),
          10: (23, 30), 11: (23, 30), 12: (23, 30), 13: (23, 30), 14: (23, 30), 15: (23, 30), 16: (45, 60), 17: (45, 60), 18: (45, 60),
          19: (90, 120), 20: (90, 120), 21: (90, 120), 22: (180, 240), 23: (180, 240), 24: (360, 480), 25: (360, 480)}
    return sum(2.0 * 9 * cin * cout * hw[i][0] * hw[i][1] for i, (cin, cout) in enumerate(LAYERS))


