"""The batched, device-resident stereo path (ssm_stereo_seq_process; BASELINE configs[3]): n frame pairs of a rectified sequence per call --
quad matcher of frame f against f - 1 (/root/reference/src/track.cpp:45-59, src/quadmatcher.cpp:548-588), SGBM depth of frame f
(src/stereo.cpp:11-30, src/rgbdframe.cpp:81-116), stereo VO on the quad matches with the host class's rand() stream
(src/vo_stereo.cpp:47-152, src/vo.cpp:74-93).  Every output must equal, byte for byte, the CPU oracle run frame by frame the way the
reference walks the sequence, and the per-pair entry points (which run the same kernels with one frame)."""
import os
import sys
import numpy as np
import pytest
from conftest import SEED
from test_sgbm import KITTI

pytestmark = pytest.mark.gpu

F, CU, CV, BASE = KITTI["f"], KITTI["cu"], KITTI["cv"], KITTI["baseline"]
VO = (F, CU, CV, BASE, 2.0, True)


def stereo_sequence(oracle, n, w, h, disp=12, flow=(3, 1), fid=0, blank=()):
    """left_k = a textured image moved by k * flow against frame 0, right_k = left_k shifted by the disparity; frames in `blank` are flat
    (no corners: no quad matches, the VO draws nothing from the rand() stream)"""
    g = oracle.bgr2gray(oracle.synth_frame(SEED, fid)[0])
    reps = (-(-h // g.shape[0]), -(-w // g.shape[1]))
    big = np.tile(g, reps)[:h, :w].copy()
    sh = lambda im, dx, dy=0: np.roll(np.roll(im, dx, axis=1), dy, axis=0).copy()
    L = [sh(big, -k * flow[0], -k * flow[1]) for k in range(n)]
    for k in blank:
        L[k] = np.full((h, w), 90, np.uint8)
    R = [sh(l, -disp) for l in L]
    return np.stack(L), np.stack(R)


def reference_walk(oracle, L, R, iters, sgbm_params, max_corners=1000, rand_state=None, prev=None, depth=True):
    """the reference's frame-by-frame walk with the CPU oracle: per frame (quad matches, VO result, disparity, depth); the rand() state runs through"""
    st = oracle.rand_state(0) if rand_state is None else rand_state
    out = []
    for f in range(len(L)):
        lp, rp = (L[f - 1], R[f - 1]) if f > 0 else (prev if prev is not None else (None, None))
        qm = oracle.quad_track(L[f], R[f], lp, rp, max_corners) if lp is not None else None
        vo = None
        if qm is not None and len(qm) >= 6:
            smp = oracle.vo_samples(st, len(qm), iters)
            vo = oracle.vo_estimate(qm, oracle.vo_params(*VO), smp)
        dd = None
        if depth:
            disp = oracle.sgbm(L[f], R[f], sgbm_params)
            dd = (disp, oracle.disparity_to_depth(disp, **KITTI))
        out.append((qm, vo, dd))
    return out, st


def check_against_walk(res, walk, f0=0, depth=True):
    for i, (qm, vo, dd) in enumerate(walk):
        f = f0 + i
        if qm is None:
            assert res["nquad"][f] == -1 and tuple(res["vo_result"][f]) == (0, 0) and not res["tr"][f].any()
        else:
            assert res["nquad"][f] == len(qm), (f, res["nquad"][f], len(qm))
            assert res["quad"][f, :len(qm)].tobytes() == qm.tobytes(), f
            if vo is None:
                assert tuple(res["vo_result"][f]) == (0, 0) and not res["tr"][f].any()
            else:
                ok, tr, inl = vo
                assert tuple(res["vo_result"][f]) == (len(inl), int(ok)), f
                assert np.array_equal(res["inliers"][f, :len(inl)], inl) and res["tr"][f].tobytes() == tr.tobytes(), f
        if depth:
            assert np.array_equal(res["disp"][f], dd[0]), f
            assert np.array_equal(res["depth"][f], dd[1]), f


def run_seq(c, L, R, rng, iters=200, cont=False, stages=0, max_corners=1000, sgbm=None):
    n, h, w = L.shape
    dl = c.dev_alloc(L.nbytes); dr = c.dev_alloc(R.nbytes); ds = c.dev_alloc(max(n * iters * 3 * 4, 4))
    try:
        c.h2d(dl, L); c.h2d(dr, R); c.h2d(ds, rng.draws(n * iters * 3))
        out = c.stereo_seq_process(dl, dr, n, w, h, continue_sequence=cont, stages=stages, max_corners=max_corners, sgbm=sgbm, vo=VO, ransac_iters=iters,
                                   rand_stream_dev=ds, **KITTI)
        c.sync()
        res = c.stereo_seq_fetch(out, n, w, h, stages or 7)
        if (stages or 7) & 4:
            rng.rewind(n * iters * 3 - res["rand_draws_used"])          # the host class's stream advances only by what was drawn
    finally:
        for p in (dl, dr, ds):
            c.dev_free(p)
    return res


@pytest.mark.parametrize("batch", [2, 16])
def test_sequence_equals_the_frame_by_frame_walk(oracle, batch, monkeypatch):
    """5 frames of 480 x 200 (a blank frame in the middle: its quad matcher finds nothing, the frame after it matches against a flat image, and neither
    draws from the rand() stream), sub-batches of 2 (carry between sub-batches) and one launch; a second call continues the sequence"""
    import semantic_slam_mapping_amd as ssm
    from semantic_slam_mapping_amd.api import GlibcRand
    c = ssm.Context(0, width=640, height=480, max_batch=2, stereo_batch=batch)
    try:
        assert c.stereo_batch() == batch
        n, w, h, iters = 5, 480, 200, 60
        sp = c.sgbm_params(numberOfDisparities=48, SADWindowSize=7)
        L, R = stereo_sequence(oracle, n + 3, w, h, disp=9, flow=(2, 1), blank=(2,))
        op = oracle.sgbm_params(48, 7)
        walk, st = reference_walk(oracle, L[:n], R[:n], iters, op)
        assert len(walk[1][0]) > 200 and walk[1][1][0] and len(walk[2][0]) == 0 and len(walk[4][0]) > 200
        rng = GlibcRand(0)
        res = run_seq(c, L[:n], R[:n], rng, iters, sgbm=sp)
        check_against_walk(res, walk)
        assert res["rand_draws_used"] == sum(1 for q, _, _ in walk if q is not None and len(q) >= 6) * iters * 3 >= 2 * iters * 3     # frames 1 and 4 (frame 2 is blank)
        # continuation: three more frames as a second call see frame n - 1 as their previous frame, and the rand() stream goes on
        walk2, _ = reference_walk(oracle, L[n:], R[n:], iters, op, rand_state=st, prev=(L[n - 1], R[n - 1]))
        res2 = run_seq(c, L[n:], R[n:], rng, iters, cont=True, sgbm=sp)
        check_against_walk(res2, walk2)
        assert res2["nquad"][0] > 200
        # the GFTT corners of the current-left image and the per-pair entry points (same kernels, one frame; they end the sequence)
        for f in (1, 4):
            gc = oracle.gftt(L[f], 1000)
            assert res["ncorners"][f] == len(gc) and np.array_equal(res["corners"][f, :len(gc)], gc)
            assert c.quad_track(L[f], R[f], L[f - 1], R[f - 1]).tobytes() == walk[f][0].tobytes()
        dep, dsp = c.stereo_depth(L[3], R[3], params=sp, **KITTI)
        assert np.array_equal(dsp, walk[3][2][0]) and np.array_equal(dep, walk[3][2][1])
        res4 = run_seq(c, L[n:n + 1], R[n:n + 1], GlibcRand(0), iters, cont=True, stages=1)
        assert res4["nquad"][0] == -1                                    # a per-pair call in between ended the sequence
        # without continue_sequence the first frame has no previous frame
        res3 = run_seq(c, L[n:], R[n:], GlibcRand(0), iters, cont=False, stages=1 | 4)
        assert res3["nquad"][0] == -1 and res3["nquad"][1] == res2["nquad"][1]
    finally:
        c.close()


def test_kitti_size_sequence(oracle):
    """1241 x 376, 80 disparities, SAD 11, 1000 corners, 200 hypotheses: the configs[3] shape, three frames"""
    import semantic_slam_mapping_amd as ssm
    from semantic_slam_mapping_amd.api import GlibcRand
    c = ssm.Context(0, width=640, height=480, max_batch=4)
    try:
        n, w, h = 3, 1241, 376
        L, R = stereo_sequence(oracle, n, w, h, disp=12, flow=(3, 1), fid=3)
        walk, _ = reference_walk(oracle, L, R, 200, oracle.sgbm_params())
        res = run_seq(c, L, R, GlibcRand(0), 200)
        check_against_walk(res, walk)
        assert res["nquad"][1] > 500 and res["vo_result"][1][1] == 1 and res["vo_result"][2][1] == 1
        # the motion: the scene moved by (3, 1) px at disparity 12 -> the translation is recovered from the matches
        assert abs(res["tr"][1][3]) > 1e-3
    finally:
        c.close()


def test_stage_selection_and_errors(oracle):
    import semantic_slam_mapping_amd as ssm
    from semantic_slam_mapping_amd.api import GlibcRand, SsmError
    c = ssm.Context(0, width=640, height=480, max_batch=3)
    try:
        n, w, h = 4, 320, 160
        L, R = stereo_sequence(oracle, n, w, h, disp=8, flow=(2, 0), fid=5)
        sp = c.sgbm_params(numberOfDisparities=32, SADWindowSize=5)
        full = run_seq(c, L, R, GlibcRand(0), 40, sgbm=sp, max_corners=100)
        only_d = run_seq(c, L, R, GlibcRand(0), 40, stages=2, sgbm=sp, max_corners=100)
        only_q = run_seq(c, L, R, GlibcRand(0), 40, stages=1, max_corners=100)
        assert np.array_equal(only_d["disp"], full["disp"]) and np.array_equal(only_d["depth"], full["depth"])
        assert np.array_equal(only_q["nquad"], full["nquad"]) and only_q["quad"].tobytes() == full["quad"].tobytes()
        assert (full["ncorners"] <= 100).all() and full["ncorners"][1] == 100           # max_corners caps the list (strongest first)
        gc = oracle.gftt(L[1], 100)
        assert np.array_equal(full["corners"][1, :100], gc)
        dl = c.dev_alloc(L.nbytes)
        try:
            with pytest.raises(SsmError):                                 # VO without the quad stage
                c.stereo_seq_process(dl, dl, n, w, h, stages=4, vo=VO, rand_stream_dev=dl, **KITTI)
            with pytest.raises(SsmError):                                 # VO without a rand() stream
                c.stereo_seq_process(dl, dl, n, w, h, stages=5, vo=VO, rand_stream_dev=None, **KITTI)
            with pytest.raises(SsmError):                                 # 112 disparities: not a supported lane split
                c.stereo_seq_process(dl, dl, n, w, h, stages=2, sgbm=c.sgbm_params(numberOfDisparities=112), **KITTI)
            out = c.stereo_seq_process(dl, dl, 0, w, h, stages=1)       # empty sequence
            c.sync()
        finally:
            c.dev_free(dl)
    finally:
        c.close()


@pytest.mark.gpu
def test_sweep_strip_handoffs_soak():
    """scripts/sgbm_soak.py: the whole batched stereo path (quad matcher, VO and the second SGBM stream running beside the sweep) over 128 KITTI-size pairs, four rounds per
    sweep kernel: every disparity image must hash to what the four-volume form of round 3 (no strips, no hand-offs) gives -- a stale or torn granule between two strips of a
    frame (kernels_sgbm.hip: sc1 tagged granules, 9 seams x 376 rows x 2 directions per frame) would change a disparity somewhere"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "sgbm_soak.py"), "4", "128"], capture_output=True, text=True, timeout=900)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0 and r.stdout.count("4 rounds identical to the four-volume form") == 2


def test_stereo_and_segnet_contexts_fit_side_by_side_at_bench_sizes():
    """the SGBM workspace is sized by the configured formulation (three cost-volume-sized buffers for the default form 2, not six): at the bench's
    128 pairs per launch of 1241 x 376 x 80 the two workspaces take <= 60 GB (they were 115 GB), so a stereo context and a SegNet-loaded RGB-D context of bench size
    live on one device side by side"""
    import semantic_slam_mapping_amd as ssm
    from semantic_slam_mapping_amd import segnet_model
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import stereo_sequence, KITTI
    from semantic_slam_mapping_amd.api import GlibcRand
    Wd, Hd, F, ITERS = 1241, 376, 256, 200
    st = ssm.Context(0, width=640, height=480, max_batch=128)
    free0 = st.mem_info()[0]
    sg = None
    try:
        L, R = stereo_sequence(F, Wd, Hd, 100)
        dl = st.dev_alloc(L.nbytes); dr = st.dev_alloc(R.nbytes); ds = st.dev_alloc(F * ITERS * 3 * 4)
        st.h2d(dl, L); st.h2d(dr, R); st.h2d(ds, GlibcRand(0).draws(F * ITERS * 3))
        vo = (KITTI["f"], KITTI["cu"], KITTI["cv"], KITTI["baseline"], 2.0, True)
        out = st.stereo_seq_process(dl, dr, F, Wd, Hd, vo=vo, ransac_iters=ITERS, rand_stream_dev=ds, **KITTI); st.sync()
        used_stereo = free0 - st.mem_info()[0]
        print("stereo context at 128 pairs per launch: %.1f GB" % (used_stereo / 1e9))
        # two SGBM workspaces of 128 pairs: 2 x 128 x (3 x 69.8 MB of volumes + 18 MB) = 58.4 GB; + the quad matcher's pyramids, the sequence outputs and inputs of 256 frames
        assert used_stereo < 70e9, used_stereo                      # (round 5: 115 GB for the workspaces alone)
        sg = ssm.Context(0, orb_features=1000, max_batch=128, voxel_capacity_log2=22, camera=(318.6, 255.3, 517.3, 516.5, 1000.0))
        for l, (wt, sc, sh) in enumerate(segnet_model.make_weights(1234)):
            sg.segnet_set_layer(l, wt, sc, sh)
        n = 128
        bufs = [sg.dev_alloc(n * 640 * 480 * 3), sg.dev_alloc(n * 640 * 480 * 2), sg.dev_alloc(n * 640 * 480 * 3), sg.dev_alloc(n * 128)]
        sg.synth_frames_dev(0x5EED0000, 0, n, *bufs)
        sg.seq_process(bufs[0], bufs[1], None, bufs[3], n, stages=ssm.api.STAGE_MAP | ssm.api.STAGE_SEGNET); sg.sync()
        assert sg.map_size() > 1000
        res = st.stereo_seq_fetch(out, F, Wd, Hd, 1 | 4)
        assert float(res["vo_result"][1:, 1].mean()) > 0.9
        for p in bufs:
            sg.dev_free(p)
        for p in (dl, dr, ds):
            st.dev_free(p)
    finally:
        if sg is not None:
            sg.close()
        st.close()
