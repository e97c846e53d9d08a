#!/usr/bin/env python3
"""Mints the committed golden fixtures under tests/golden/ from the independent numpy/Python restatement in pyref.py
(NOT from the C oracle and NOT from the HIP path).  The reference ships no test vectors, so these are
the pins of the CPU oracle; the palette fixture is cross-checked against /root/reference/000000.png when present.
Run:  python tests/golden/make_golden.py        (about a minute; pure-Python ORB on a small image)"""
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pyref  # noqa: E402

CAM = (318.6, 255.3, 517.3, 516.5, 1000.0)


def pattern_from_inc():
    import re
    t = open(os.path.join(HERE, "..", "..", "oracle", "orb_pattern.inc")).read()
    t = t[t.index("*/") + 2:]
    return [int(x) for x in re.findall(r"-?\d+", t)]


def g_matcher():
    out = {}
    rng = np.random.default_rng(1234)
    for name, nq, nt in (("n2", 2, 2), ("n3", 3, 3), ("n64", 64, 64), ("n1000", 1000, 1000)):
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8); t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        if nq >= 64:      # engineered structure: duplicates, exact copies (d0 = 0), near copies, all-equal rows
            t[5] = t[4]; t[40] = t[4]; q[0] = t[4]; q[1] = t[40]
            q[2] = t[7]; q[2, 0] ^= 1
            q[3] = 0; t[10] = 0; t[11] = 0; t[12] = 255
        idx, dist = pyref.knn2(q, t)
        m = pyref.match(q, t, 0.8)
        out[name + "_q"] = q; out[name + "_t"] = t; out[name + "_idx"] = idx; out[name + "_dist"] = dist
        out[name + "_match"] = np.array([(a, b, c, d) for a, b, c, d in m], np.float64).reshape(-1, 4)
    lut = np.array([[pyref.ratio_keep(d0, d1, 0.8) for d1 in range(257)] for d0 in range(257)], np.uint8)
    out["ratio_lut_0p8"] = np.packbits(lut)
    np.savez_compressed(os.path.join(HERE, "matcher.npz"), **out)


def synth_small(rng, h, w):
    img = rng.integers(90, 140, (h, w, 3)).astype(np.uint8)
    img = (img // 8 * 8).astype(np.uint8)
    for _ in range(60):
        x0, y0 = rng.integers(0, w - 8), rng.integers(0, h - 8)
        ww, hh = rng.integers(4, 25), rng.integers(4, 25)
        img[y0:y0 + hh, x0:x0 + ww] = rng.integers(0, 256, 3)
    return img


def g_mapper():
    rng = np.random.default_rng(77)
    h, w = 24, 32
    depth = rng.integers(300, 3000, (h, w)).astype(np.uint16)
    depth[rng.random((h, w)) < 0.1] = 0
    depth[3, 3] = 50000
    cls = rng.integers(0, 12, (h // 4, w // 4))
    sem = np.array(pyref.PALETTE_BGR, np.uint8)[np.kron(cls, np.ones((4, 4), int))]
    sem[0, 0] = (1, 2, 3)
    rgb = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    a = 0.4
    T = [[np.cos(a), 0, np.sin(a), 0.5], [0, 1, 0, -0.25], [-np.sin(a), 0, np.cos(a), 2.0], [0, 0, 0, 1]]
    mask = pyref.moving_mask(sem)
    cam = (15.5, 11.5, 30.0, 31.0, 1000.0)
    pts = pyref.backproject(depth, rgb, sem, mask, cam, T, 2.5)
    pts0 = pyref.backproject(depth, rgb, sem, mask, cam, None, 40.0)
    vox = pyref.voxel_filter(pts0, 0.1)
    vox2 = pyref.voxel_filter(pts, 0.05)
    arr = lambda p: np.array([(float(x), float(y), float(z), b, g, r, l) for x, y, z, b, g, r, l in p], np.float64).reshape(-1, 7)
    np.savez_compressed(os.path.join(HERE, "mapper.npz"), depth=depth, sem=sem, rgb=rgb, T=np.array(T, np.float64), cam=np.array(cam),
                        mask=mask, pts_T_2p5=arr(pts), pts_cam_40=arr(pts0), vox_cam_0p1=arr(vox), vox_T_0p05=arr(vox2))


def g_orb():
    rng = np.random.default_rng(2024)
    bgr = synth_small(rng, 150, 180)
    gray = pyref.bgr2gray(bgr)
    pat = pattern_from_inc()
    kps, desc = pyref.orb_extract(gray, 120, 1.2, 3, 20, 7, pat)
    small = pyref.resize_linear(gray, 150, 125)
    blur = pyref.gaussian7(gray)
    S = np.array([[pyref.fast_S(gray, x, y) for x in range(3, 60)] for y in range(3, 40)], np.int32)
    ang_in = rng.integers(-40000, 40000, (200, 2)).astype(np.float32)
    ang_in[:4] = [[0, 0], [0, 5], [-3, 0], [7, 7]]
    ang = np.array([pyref.fast_atan2(y, x) for y, x in ang_in], np.float32)
    sc_in = rng.uniform(0, 6.3, 300).astype(np.float32)
    sc = np.array([pyref.contract_sincos(a) for a in sc_in], np.float32)
    np.savez_compressed(os.path.join(HERE, "orb.npz"), bgr=bgr, gray=gray, resized_150x125=small, blur=blur, fast_S_3_3=S,
                        kps=np.array([[float(v) for v in k] for k in kps], np.float64).reshape(-1, 7), desc=desc,
                        atan2_in=ang_in, atan2_out=ang, sincos_in=sc_in, sincos_out=sc,
                        taps=np.array(pyref.gaussian_taps()), umax=np.array(pyref.umax_table()),
                        feat_1000_8=np.array(pyref.level_params(1000, 1.2, 8)[2]), feat_2000_8=np.array(pyref.level_params(2000, 1.2, 8)[2]))


def g_palette():
    info = {"palette_bgr": pyref.PALETTE_BGR, "source": "src/mapper.cpp:42-54 comments; SegNet driving_webdemo id order"}
    ref_png = "/root/reference/000000.png"
    if os.path.exists(ref_png):
        from PIL import Image
        im = np.array(Image.open(ref_png).convert("RGB"))
        cols = sorted({(int(c[2]), int(c[1]), int(c[0])) for c in np.unique(im.reshape(-1, 3), axis=0)})
        info["colours_in_reference_000000_png_bgr"] = cols
        info["all_png_colours_in_palette"] = all(tuple(c) in [tuple(p) for p in pyref.PALETTE_BGR] for c in cols)
    ref2 = "/root/reference/0002.png"                    # the README's SegNet output sample, at the net's 480 x 360
    if os.path.exists(ref2):
        from PIL import Image
        im = np.array(Image.open(ref2).convert("RGB"))
        cols, cnt = np.unique(im.reshape(-1, 3), axis=0, return_counts=True)
        pal = [tuple(p) for p in pyref.PALETTE_BGR]
        info["reference_0002_png"] = {"size_wh": [int(im.shape[1]), int(im.shape[0])],
                                      "label_histogram": {str(pal.index((int(c[2]), int(c[1]), int(c[0])))): int(n) for c, n in zip(cols, cnt)
                                                          if (int(c[2]), int(c[1]), int(c[0])) in pal},
                                      "all_colours_in_palette": all((int(c[2]), int(c[1]), int(c[0])) in pal for c in cols)}
    json.dump(info, open(os.path.join(HERE, "palette.json"), "w"), indent=1)


def g_quad():
    rng = np.random.default_rng(31)
    a = synth_small(rng, 72, 96)
    lc = pyref.bgr2gray(a)
    rc = np.roll(lc, -5, axis=1).copy(); rc[:, -5:] = 100
    pts = pyref.gftt(lc, 40, 0.04, 8.0)
    nxt, st, err = pyref.lk_track(lc, rc, pts[:12])
    edge = np.array([[1.0, 1.0], [94.5, 70.5], [-20.0, 5.0], [50.0, 90.0]], np.float32)
    e_out, e_st, e_err = pyref.lk_track(lc, rc, edge)
    np.savez_compressed(os.path.join(HERE, "quad.npz"), lc=lc, rc=rc, eig=pyref.min_eigen_map(lc), gftt_40_0p04_8=pts, gftt_15_0p1_12=pyref.gftt(lc, 15, 0.1, 12.0),
                        pyrdown=pyref.pyrdown(lc), scharr=pyref.scharr(lc), lk_next=nxt, lk_status=st, lk_err=err,
                        edge_pts=edge, edge_next=e_out, edge_status=e_st, edge_err=e_err)


def g_segnet():
    """G6: SegNet building blocks from PyTorch-CPU fp32 on small seeded integer-valued tensors, on which
    fp16 storage / fp32 accumulation is exact: conv3x3 pad 1 + folded BN + ReLU for five layer shapes (incl. the 3-channel
    first and the 12-channel last layer, odd sizes), max_pool2d(2, 2, ceil_mode, return_indices), max_unpool2d(output_size),
    channel arg-max with first-maximum ties."""
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from semantic_slam_mapping_amd.segnet_model import LAYERS
    out = {}
    for layer, h, w in ((0, 21, 34), (1, 23, 30), (3, 12, 15), (12, 17, 33), (25, 31, 17)):
        cin, cout = LAYERS[layer][:2]
        rng = np.random.default_rng(9000 + layer)
        wt = rng.integers(-1, 2, (cout, cin, 3, 3)).astype(np.float32)
        sc = (2.0 ** rng.integers(-7, -4, cout)).astype(np.float32); sh = rng.integers(-3, 4, cout).astype(np.float32)
        x = rng.integers(-4, 5, (h, w, cin)).astype(np.float32)
        y = F.conv2d(torch.from_numpy(x.transpose(2, 0, 1))[None], torch.from_numpy(wt), padding=1)[0].numpy() * sc[:, None, None] + sh[:, None, None]
        if layer != len(LAYERS) - 1:
            y = np.maximum(y, 0)
        y16 = y.transpose(1, 2, 0).astype(np.float16)
        assert np.array_equal(y16.astype(np.float32), y.transpose(1, 2, 0)), "fixture must be exact in fp16"
        k = f"conv{layer}_"
        out[k + "w"] = wt.astype(np.int8); out[k + "scale_log2"] = np.log2(sc).astype(np.int8); out[k + "shift"] = sh.astype(np.int8)
        out[k + "x"] = x.astype(np.int8); out[k + "y"] = y16
    for name, h, w, c in (("a", 23, 30, 64), ("b", 5, 7, 96)):
        rng = np.random.default_rng(9100 + h)
        x = rng.integers(-3, 4, (h, w, c)).astype(np.float32)
        xt = torch.from_numpy(x.transpose(2, 0, 1))[None]
        p, idx = F.max_pool2d(xt, 2, 2, ceil_mode=True, return_indices=True)
        u = F.max_unpool2d(p, idx, 2, 2, output_size=(h, w))
        out[f"pool_{name}_x"] = x.astype(np.int8); out[f"pool_{name}_y"] = p[0].numpy().transpose(1, 2, 0).astype(np.int8)
        out[f"pool_{name}_idx"] = idx[0].numpy().transpose(1, 2, 0).astype(np.int32)      # flat index h*W + w of the (first) maximum
        out[f"pool_{name}_unpool"] = u[0].numpy().transpose(1, 2, 0).astype(np.int8)
    np.savez_compressed(os.path.join(HERE, "segnet.npz"), **out)


def g_sgbm():
    """cv::StereoSGBM vectors from pyref.sgbm_raw / sgbm (volume-form restatement): the reference's parameters (80 disparities, SAD 11, stereo.cpp:11-30) on a
    96 x 320 pair, and a second set with other parameters incl. a negative minDisparity"""
    sys.path.insert(0, os.path.join(HERE, ".."))
    from test_sgbm import stereo_pair
    out = {}
    cases = {"ref": (96, 320, 0, dict(ndisp=80, SAD=11), ((20, None), (45, (0.3, 0.75, 0.3, 0.7))), 0),
             "alt": (50, 200, 5, dict(ndisp=48, SAD=7, uniquenessRatio=5, disp12MaxDiff=2), ((12, None), (30, (0.2, 0.9, 0.4, 0.8))), 6),
             "neg": (30, 100, 7, dict(ndisp=32, SAD=3, minD=-8), ((4, None), (10, (0.2, 0.8, 0.2, 0.8))), 4)}
    for name, (h, w, seed, kw, planes, noise) in cases.items():
        left, right, _ = stereo_pair(h, w, seed, planes=planes, noise=noise)
        raw = pyref.sgbm_raw(left, right, **kw)
        out[name + "_left"] = left; out[name + "_right"] = right; out[name + "_raw"] = raw
        out[name + "_disp"] = pyref.filter_speckles(pyref.median3_s16(raw), (kw.get("minD", 0) - 1) * 16, 100, 16 * 32)
        out[name + "_params"] = np.array([kw.get("minD", 0), kw["ndisp"], kw["SAD"], kw.get("uniquenessRatio", 10), kw.get("disp12MaxDiff", 1)], np.int32)
    np.savez_compressed(os.path.join(HERE, "sgbm.npz"), **out)


def g_pnp():
    """PnPSolver::solvePnP vectors from pyref.pnp_solve: exact and noisy correspondences, gross outliers, depth-less rows (the reference's index mix-up),
    more than 1024 edges (the lane order of the numeric contract wraps), too few points, a poor initial transform"""
    def pose(rx, ry, rz, t):
        cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
        R = np.array([[cy * cz, -cy * sz, sy], [sx * sy * cz + cx * sz, -sx * sy * sz + cx * cz, -sx * cy], [-cx * sy * cz + sx * sz, cx * sy * sz + sx * cz, cx * cy]])
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
        return T
    out = {}
    cases = {"exact": (21, 120, 0, 0, 0.0, (0, 0, 0, (0, 0, 0))), "outliers": (22, 260, 9, 0, 0.2, (0.01, 0.0, -0.01, (0.01, 0.0, 0.02))),
             "nodepth": (23, 90, 6, 4, 0.3, (0, 0, 0, (0, 0, 0))), "lanes": (24, 1300, 11, 7, 0.4, (0.005, -0.004, 0.003, (0.02, -0.01, 0.0))),
             "few": (25, 7, 0, 3, 0.0, (0, 0, 0, (0, 0, 0))), "farinit": (26, 150, 8, 0, 0.1, (0.08, -0.05, 0.06, (0.3, -0.2, 0.25)))}
    for name, (seed, n, oe, ze, noise, init) in cases.items():
        rng = np.random.default_rng(seed)
        X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(1.0, 6.0, n)], 1).astype(np.float32)
        Tgt = pose(0.02, -0.03, 0.015, (0.05, -0.02, 0.08))
        Pc = (Tgt[:3, :3] @ X.astype(np.float64).T).T + Tgt[:3, 3]
        uv = np.stack([Pc[:, 0] / Pc[:, 2] * CAM[2] + CAM[0], Pc[:, 1] / Pc[:, 2] * CAM[3] + CAM[1]], 1)
        if noise:
            uv += rng.normal(0, noise, uv.shape)
        if oe:
            uv[::oe] += rng.uniform(25, 60, (len(uv[::oe]), 2))
        if ze:
            X[3::ze] = 0
        img = uv.astype(np.float32); T0 = pose(*init)
        ok, T, inl = pyref.pnp_solve(img, X, CAM, T0)
        out[name + "_img"] = img; out[name + "_obj"] = X; out[name + "_T0"] = T0; out[name + "_T"] = T; out[name + "_inl"] = inl; out[name + "_ok"] = np.array([int(ok)], np.int32)
    np.savez_compressed(os.path.join(HERE, "pnp.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:                           # e.g. `make_golden.py sgbm`: only the named fixtures
        for n in sys.argv[1:]:
            globals()["g_" + n]()
    else:
        g_palette(); g_matcher(); g_mapper(); g_orb(); g_quad(); g_segnet(); g_sgbm(); g_pnp()
    print("golden fixtures written to", HERE)
