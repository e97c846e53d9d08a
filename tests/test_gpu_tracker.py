"""The bulk pose chain (ssm_tracker_run) against the reference's frame-by-frame walk done with the CPU oracle: Tracker::updateFrame in RGB-D mode
(/root/reference/src/track.cpp:8-36,140-212) = ORB (oracle/orb.c), OrbFeature::match for every member of the refFrames deque (oracle/match.c), the
correspondence gather of track.cpp:150-163, PnPSolver::solvePnP (oracle/pnp.c), the 15-correspondence / 15-inlier tests, cntLost -> LOST -> lostRecover.
The walk below shares nothing with the product but the input frames; every pose must be the same bits."""
import numpy as np
import pytest
from conftest import CAM, SEED

pytestmark = pytest.mark.gpu


def iso_mul(A, B):
    """4 x 4 product in the operation order of Eigen::Isometry3d::operator* as include/ssm/compat.h writes it (s = 0; s += a_ik b_kj, k = 0..3)"""
    C = np.zeros((4, 4))
    for i in range(4):
        for j in range(4):
            s = 0.0
            for k in range(4):
                s += float(A[i, k]) * float(B[k, j])
            C[i, j] = s
    return C


def iso_inv(M):
    R = np.eye(4)
    for i in range(3):
        for j in range(3):
            R[i, j] = M[j, i]
    for i in range(3):
        R[i, 3] = -(float(R[i, 0]) * float(M[0, 3]) + float(R[i, 1]) * float(M[1, 3]) + float(R[i, 2]) * float(M[2, 3]))
    return R


def iso_apply(M, p):
    out = []
    v = (float(p[0]), float(p[1]), float(p[2]), 1.0)
    for i in range(3):
        s = 0.0
        for k in range(4):
            s += float(M[i, k]) * v[k]
        out.append(s)
    return out


def reference_walk(oracle, frames, ref_frames, max_lost, ratio, nfeat):
    """returns (poses, info rows (state, tracked, n_matches, n_inliers))"""
    feats = []
    for bgr, dep in frames:
        kps, desc = oracle.orb_extract(oracle.bgr2gray(bgr), nfeatures=nfeat)
        pos = np.array([oracle.project2dTo3d(dep, CAM, int(k["x"]), int(k["y"])) for k in kps], np.float32).reshape(-1, 3)
        feats.append((kps, desc, pos))
    state, cnt_lost = 0, 0
    speed, last = np.eye(4), np.eye(4)
    refs = []                                     # (frame index, pose)
    poses, infos = [], []
    for f, (kps, desc, pos) in enumerate(feats):
        tracked, nm, ninl = 0, -1, 0
        if state == 0:
            T = np.eye(4); refs.append((f, T)); speed = np.eye(4); state = 1; tracked = 1
        elif state == 2:
            T = refs[-1][1].copy(); refs = [(f, T)]; state = 1; cnt_lost = 0; tracked = 1
        else:
            T = iso_mul(speed, refs[-1][1])
            img, obj = [], []
            for rf, rpose in refs:
                rk, rd, rp = feats[rf]
                m = oracle.match(rd, desc, ratio) if len(rd) >= 1 and len(desc) >= 2 else []
                inv = iso_inv(rpose)
                for q in m:
                    p = rp[q["queryIdx"]]
                    if p[0] == 0 and p[1] == 0 and p[2] == 0:
                        continue
                    obj.append(np.array(iso_apply(inv, p)).astype(np.float32)); img.append((kps[q["trainIdx"]]["x"], kps[q["trainIdx"]]["y"]))
            nm = len(img)
            ok = nm >= 15
            if ok:
                T0 = iso_mul(speed, last)
                _, Ts, inl = oracle.pnp_solve(np.array(img, np.float32).reshape(-1, 2), np.array(obj, np.float32).reshape(-1, 3), CAM, T0, min_inliers=10)
                ninl = len(inl); ok = ninl >= 15
            if not ok:
                cnt_lost += 1
                if cnt_lost > max_lost:
                    state = 2
            else:
                T = Ts; cnt_lost = 0; speed = iso_mul(T, iso_inv(last)); last = T.copy()
                refs.append((f, T)); refs = refs[-ref_frames:]; tracked = 1
        poses.append(T); infos.append((state, tracked, nm, ninl))
    return poses, infos


@pytest.mark.parametrize("use_device", [False, True, 1, 2, 4, "timeout"])
def test_bulk_tracker_equals_the_oracle_walk(oracle, use_device, monkeypatch):
    """14 frames of 640 x 480 (500 features): a rigid plane scene that tracks, one flat frame (fails: the deque then reaches behind the match-table window,
    so the following frames need on-demand matches), later two flat frames in a row with tracker_max_lost_frame = 1 (LOST, then lostRecover)"""
    import semantic_slam_mapping_amd as ssm
    # use_device True: the default form of the device chain (a cluster of eight blocks per chain, kernels_pnp.hip); 1 / 2 / 4: that many blocks -- the same bits in every form
    # "timeout": the cluster's exchange reports a time-out at once (test hook) -> the tracker must fall back to one block per chain and still produce the walk
    blocks, timeout = 0, use_device == "timeout"
    if timeout:
        monkeypatch.setenv("SSM_PNP_TEST_TIMEOUT", "1"); use_device = True
    elif use_device is not True and use_device:
        blocks, use_device = int(use_device), True                                   # ssm_tracker_params.blocks (no environment variable, no subprocess)
    n, W, H, nfeat = 14, 640, 480, 500
    bgr0 = oracle.synth_frame(SEED, 21)[0]
    frames = []
    for k in range(n):
        bgr = np.roll(np.roll(bgr0, 2 * k, axis=1), k, axis=0).copy()
        if k in (4, 9, 10):
            bgr[:] = 100
        frames.append((bgr, np.full((H, W), 2000, np.uint16)))
    c = ssm.Context(0, orb_features=nfeat, max_batch=3, voxel_capacity_log2=14, camera=CAM)
    trk = ssm.Tracker(c, max_lost_frame=1, use_device=use_device, blocks=blocks)
    db = c.dev_alloc(n * W * H * 3); dd = c.dev_alloc(n * W * H * 2)
    try:
        c.h2d(db, np.stack([f[0] for f in frames])); c.h2d(dd, np.stack([f[1] for f in frames]))
        # two calls (8 + 6 frames): the tracker's state and the matcher's history rows both carry over
        out = c.seq_process(db, dd, None, None, 8, stages=3)
        pa, ia = trk.run(out, 8)
        out = c.seq_process(db + 8 * W * H * 3, dd + 8 * W * H * 2, None, None, n - 8, continue_sequence=True, stages=3)
        pb, ib = trk.run(out, n - 8)
        poses = np.concatenate([pa, pb]); info = np.concatenate([ia, ib])
        wposes, winfo = reference_walk(oracle, frames, c.R, 1, c.cfg.knn_match_ratio, nfeat)
        for f in range(n):
            assert tuple(int(v) for v in info[f]) == winfo[f], (f, info[f], winfo[f])
            assert poses[f].tobytes() == np.asarray(wposes[f], np.float64).tobytes(), f
        assert [int(i["tracked"]) for i in info] == [1, 1, 1, 1, 0, 1, 1, 1, 1, 0, 0, 1, 1, 1]
        assert int(info[10]["state"]) == 2 and int(info[11]["state"]) == 1 and info[5]["n_inliers"] > 100
        assert ("one block per chain" in trk.last_error()) == timeout                 # the downgrade is visible to the caller, and only then
        dev_frames, host_frames = trk.stats()
        assert dev_frames + host_frames == n and (dev_frames >= 5 if use_device else dev_frames == 0)     # the regular stretches ran as one-block chains on the GPU
        # the solved poses feed the map stage
        dp = c.dev_alloc(n * 128)
        try:
            c.h2d(dp, np.ascontiguousarray(poses.transpose(0, 2, 1)).reshape(n, 16))
            c.map_clear()
            c.seq_process(db, dd, db, dp, n, stages=4); c.sync()
            assert c.map_size() > 100
        finally:
            c.dev_free(dp)
    finally:
        c.dev_free(db); c.dev_free(dd); trk.close(); c.close()


def test_trackers_on_their_own_streams_from_two_threads(oracle):
    """own_stream = 1: two trackers driven from two host threads walk two independent sequences (views of one seq_process call) side by side, each chain on
    a stream of its own; the poses are those of one tracker walking them one after the other"""
    import threading
    import semantic_slam_mapping_amd as ssm
    from semantic_slam_mapping_amd._lib import SeqOutDev
    c = ssm.Context(0, width=640, height=480, max_batch=8, camera=CAM)
    try:
        n, half = 16, 8
        frames = [oracle.synth_frame(SEED, f) for f in range(2)]
        base = frames[0][0]
        bgr = np.stack([np.roll(base, (k % half, 2 * (k % half)), (0, 1)) for k in range(n)])      # two identical 8-frame panning sequences
        dep = np.full((n, 480, 640), 2000, np.uint16)
        db = c.dev_alloc(bgr.nbytes); dd = c.dev_alloc(dep.nbytes)
        try:
            c.h2d(db, bgr); c.h2d(dd, dep)
            o = c.seq_process(db, dd, None, None, n, stages=ssm.api.STAGE_ORB | ssm.api.STAGE_MATCH)
            c.sync()
            def view(a0):
                return SeqOutDev(o.kps + a0 * o.cap * 28, o.desc + a0 * o.cap * 32, o.pos3d + a0 * o.cap * 12, o.nkp + a0 * 4, o.matches + a0 * o.R * o.cap * 16,
                                 o.nmatch + a0 * o.R * 4, o.npoints + a0 * 4, o.cap, o.R)
            one = ssm.Tracker(c, use_device=True)
            ref = []
            for a0 in (0, half):
                one.reset(); ref.append(one.run(view(a0), half))
            one.close()
            trk = [ssm.Tracker(c, use_device=True, own_stream=True) for _ in range(2)]
            got = [None, None]
            def walk(k):
                got[k] = trk[k].run(view(k * half), half)
            th = [threading.Thread(target=walk, args=(k,)) for k in range(2)]
            for t in th: t.start()
            for t in th: t.join()
            for k in range(2):
                assert got[k] is not None and got[k][0].tobytes() == ref[k][0].tobytes() and got[k][1].tobytes() == ref[k][1].tobytes()
                assert trk[k].stats()[0] >= half - 2                                # the chains ran on the device
                trk[k].close()
            assert ref[0][1]["tracked"].sum() >= half - 1
        finally:
            c.dev_free(db); c.dev_free(dd)
    finally:
        c.close()


@pytest.mark.parametrize("blocks,nfeat", [(8, 1000), (2, 1000), (1, 1000), (8, 150), (8, 40)])
def test_device_chain_equals_the_host_chain_on_the_panning_stream(oracle, blocks, nfeat):
    """bench.py's closed pose loop in small: frame 0 seen by a panning camera, 1000 features and five reference frames (up to five edges per contract lane), two
    20-frame sequences with different noise in the depth.  The host chain (include/ssm/pnp_core.h on one core: the arithmetic oracle/pnp.c pins and
    test_bulk_tracker_equals_the_oracle_walk walks) and the device chain must give the same bits -- with eight blocks that is the round-5 form: the fused pass's sums
    over two waves per group, an iteration's first round building the next iteration's system beside its three candidates, rejected streaks in one round.
    150 / 40 features: contract groups (and whole blocks of the cluster) without an edge, solves that fail the 15-correspondence tests."""
    import semantic_slam_mapping_amd as ssm
    from semantic_slam_mapping_amd._lib import SeqOutDev
    n, CH, W, H = 40, 20, 640, 480
    c = ssm.Context(0, width=W, height=H, max_batch=8, camera=CAM, orb_features=nfeat)
    try:
        base = oracle.synth_frame(SEED, 3)[0]
        bgr = np.stack([np.roll(base, (k % CH, 2 * (k % CH)), (0, 1)) for k in range(n)])
        rng = np.random.default_rng(SEED + 5)
        dep = np.full((n, H, W), 2000, np.uint16)
        dep[CH:] += rng.integers(0, 40, (n - CH, H, W)).astype(np.uint16)                          # the second sequence: noisy depth (more rejected trials, outliers)
        db = c.dev_alloc(bgr.nbytes); dd = c.dev_alloc(dep.nbytes)
        try:
            c.h2d(db, bgr); c.h2d(dd, dep)
            o = c.seq_process(db, dd, None, None, n, stages=ssm.api.STAGE_ORB | ssm.api.STAGE_MATCH)
            c.sync()
            def view(a0):
                return SeqOutDev(o.kps + a0 * o.cap * 28, o.desc + a0 * o.cap * 32, o.pos3d + a0 * o.cap * 12, o.nkp + a0 * 4, o.matches + a0 * o.R * o.cap * 16,
                                 o.nmatch + a0 * o.R * 4, o.npoints + a0 * 4, o.cap, o.R)
            host = ssm.Tracker(c, use_device=False); dev = ssm.Tracker(c, use_device=True, blocks=blocks)
            try:
                for a0 in (0, CH):
                    host.reset(); dev.reset()
                    ph, ih = host.run(view(a0), CH)
                    pd, idv = dev.run(view(a0), CH)
                    assert ih.tobytes() == idv.tobytes(), a0
                    for f in range(CH):
                        assert ph[f].tobytes() == pd[f].tobytes(), (a0, f)
                    if nfeat == 1000:
                        assert int(ih["tracked"].sum()) >= CH - 2 and int(ih["n_inliers"][5:].min()) > 100, a0
                assert (dev.stats()[0] >= n - 4 or nfeat < 1000) and dev.last_error() == ""          # the chains ran on the device, no downgrade
            finally:
                host.close(); dev.close()
        finally:
            c.dev_free(db); c.dev_free(dd)
    finally:
        c.close()
