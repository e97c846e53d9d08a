"""Stereo quad-matcher rows (a5/a6): HIP path vs the CPU oracle (oracle/quad.c), bit-exact -- the window sums are exact
integers and the corner order is fully specified, so GFTT corner lists, LK tracks and the filtered quad matches must
be identical.  Synthetic rectified stereo pairs: right = left shifted by a disparity, previous = current shifted by a flow."""
import numpy as np
import pytest
from conftest import SEED, rand_desc

pytestmark = pytest.mark.gpu


def stereo_pair(oracle, w=1241, h=376, disp=12, flow=(3, 1), fid=0):
    g = oracle.bgr2gray(oracle.synth_frame(SEED, fid)[0])
    big = np.tile(g, (1, 2))[:h, :w].copy()
    sh = lambda im, dx, dy=0: np.roll(np.roll(im, dx, axis=1), dy, axis=0).copy()
    lc = big; rc = sh(big, -disp); lp = sh(big, flow[0], flow[1]); rp = sh(lp, -disp)
    return lc, rc, lp, rp


def test_gftt_matches_oracle(ctx, oracle):
    lc, _, _, _ = stereo_pair(oracle)
    for mc, q, md in ((1000, 0.04, 8.0), (200, 0.01, 5.0), (5000, 0.1, 12.0), (3000, 0.02, 3.0)):
        g = ctx.gftt(lc, mc, q, md)
        o = oracle.gftt(lc, mc, q, md)
        assert len(g) == len(o) and np.array_equal(g, o)
    flat = np.full((64, 80), 50, np.uint8)
    assert len(ctx.gftt(flat)) == 0 == len(oracle.gftt(flat))


def test_gftt_min_distance_and_order(ctx, oracle):
    lc, _, _, _ = stereo_pair(oracle, 320, 240)
    p = ctx.gftt(lc, 1000, 0.04, 8.0)
    e = oracle.min_eigen_map(lc)
    vals = e[p[:, 1].astype(int), p[:, 0].astype(int)]
    assert (np.diff(vals) <= 0).all()                                   # strongest first
    d = np.sqrt(((p[:, None, :] - p[None, :, :]) ** 2).sum(2)) + np.eye(len(p)) * 1e9
    assert d.min() >= 8.0                                               # minDistance


def test_lk_track_matches_oracle(ctx, oracle):
    lc, rc, lp, rp = stereo_pair(oracle)
    pts = oracle.gftt(lc, 600)
    for a, b in ((lc, rc), (lc, lp)):
        g, gs, ge = ctx.lk_track(a, b, pts)
        o, os_, oe = oracle.lk_track(a, b, pts)
        assert np.array_equal(gs, os_) and g.tobytes() == o.tobytes() and ge.tobytes() == oe.tobytes()
    # points at / beyond the border and in flat regions (status 0 paths)
    edge = np.array([[0.5, 0.5], [1240.0, 375.0], [-30.0, 10.0], [600.0, 400.0], [3.0, 200.0]], np.float32)
    g, gs, ge = ctx.lk_track(lc, rc, edge)
    o, os_, oe = oracle.lk_track(lc, rc, edge)
    assert np.array_equal(gs, os_) and g.tobytes() == o.tobytes()
    flat = np.full((376, 1241), 77, np.uint8)
    g, gs, _ = ctx.lk_track(flat, flat, pts[:50])
    o, os_, _ = oracle.lk_track(flat, flat, pts[:50])
    assert (gs == 0).all() and np.array_equal(gs, os_) and g.tobytes() == o.tobytes()


@pytest.mark.parametrize("w,h,disp,flow", [(1241, 376, 12, (3, 1)), (640, 480, 7, (-2, 0)), (1241, 376, 2, (1, 1))])
def test_quad_track_matches_oracle(ctx, oracle, w, h, disp, flow):
    ims = stereo_pair(oracle, w, h, disp, flow, fid=3)
    g = ctx.quad_track(*ims)
    o = oracle.quad_track(*ims)
    assert len(g) == len(o) and g.tobytes() == o.tobytes()
    if disp > 3:
        assert len(g) > 500 and abs(np.median(g["u1c"] - g["u2c"]) - disp) < 0.1
        assert abs(np.median(g["u1p"] - g["u1c"]) - flow[0]) < 0.1 and abs(np.median(g["v1p"] - g["v1c"]) - flow[1]) < 0.1
    else:
        assert len(g) == 0                                            # disparity <= 3 px is rejected (quadmatcher.cpp:438,478)


def test_window_match_and_chain(ctx, oracle):
    rng = np.random.default_rng(8)
    n = 400
    k_lc = rng.uniform(0, 600, (n, 2)).astype(np.float32)
    k_rc = k_lc + np.array([-9.0, 0.5], np.float32); k_rp = k_rc + np.array([2.0, -3.0], np.float32); k_lp = k_rp + np.array([9.0, 0.3], np.float32)
    d = rand_desc(rng, n)
    noisy = lambda: d ^ (rng.random((n, 32)) < 0.02).astype(np.uint8)
    d_rc, d_rp, d_lp = noisy(), noisy(), noisy()
    perm = rng.permutation(n)
    k_rc, d_rc = k_rc[perm], d_rc[perm]
    ms = []
    for (ka, da, kb, db, sw, sh) in ((k_lc, d, k_rc, d_rc, 20, 2), (k_rc, d_rc, k_rp, d_rp, 20, 20), (k_rp, d_rp, k_lp, d_lp, 20, 2)):
        g = ctx.window_match(ka, da, kb, db, sw, sh, 80.0)
        o = oracle.window_match(ka, da, kb, db, sw, sh, 80.0)
        assert g.tobytes() == o.tobytes()
        ms.append(g)
    assert (ms[0]["trainIdx"] >= 0).mean() > 0.9 and (ms[0]["imgIdx"] == -1).all()
    # empty window -> trainIdx -1 with the 999999999.9f sentinel distance
    far = ctx.window_match(k_lc[:3] + 5000, d[:3], k_rc, d_rc, 20, 2, 80.0)
    assert (far["trainIdx"] == -1).all() and (far["distance"] == np.float32(999999999.9)).all()
    chain = oracle.quad_chain(k_lc, k_rc, k_rp, k_lp, *ms)
    assert len(chain) > 250 and (chain["i2c"] > 0).all()              # index 0 counts as "unmatched" in the reference (quirk 9)


def test_filter_tracks_gates(oracle):
    base = np.array([[100.0, 50.0]], np.float32)
    mk = lambda dx=0, dy=0: base + np.array([dx, dy], np.float32)
    ok = oracle.filter_tracks(mk(), mk(-10), mk(2, 1), mk(-8, 1), mk(2, 1))
    assert len(ok) == 1 and ok[0]["u1c"] == 100 and ok[0]["u2c"] == 90 and ok[0]["i1c"] == 0
    assert len(oracle.filter_tracks(mk(), mk(-3), mk(2, 1), mk(-1, 1), mk(2, 1))) == 0           # disparity must be > 3
    assert len(oracle.filter_tracks(mk(), mk(-10, 21), mk(2, 1), mk(-8, 1), mk(2, 1))) == 0      # stereo row difference < 20
    assert len(oracle.filter_tracks(mk(), mk(-10), mk(2, 1), mk(-8, 1), mk(3.6, 1))) == 0        # loop closure error < 1 px after cvRound
    assert len(oracle.filter_tracks(mk(-101), mk(-111), mk(-99, 1), mk(-109, 1), mk(-99, 1))) == 0  # x must be > 0 (inside 1280x960)
