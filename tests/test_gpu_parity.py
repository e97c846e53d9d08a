"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.  Bit-exact: every
kernel on this path is integer/byte work or float work under an explicit rounding contract."""
import numpy as np
import pytest
from conftest import CAM, SEED, rand_desc

pytestmark = pytest.mark.gpu


def same_struct(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


# ---------------------------------------------------------------- K6 matcher (src/orb.cpp:16-29)
@pytest.mark.parametrize("nq,nt", [(1, 2), (3, 2), (2, 3), (64, 64), (257, 1000), (1000, 1000), (1024, 1025), (1500, 2100)])
def test_knn2_random(ctx, oracle, nq, nt):
    rng = np.random.default_rng(nq * 7919 + nt)
    q, t = rand_desc(rng, nq), rand_desc(rng, nt)
    gi, gd = ctx.knn2(q, t)
    oi, od = oracle.knn2(q, t)
    assert np.array_equal(gd, od) and np.array_equal(gi, oi)


def test_knn2_ties_duplicates_and_zero_distance(ctx, oracle):
    rng = np.random.default_rng(5)
    base = rand_desc(rng, 40)
    t = np.concatenate([base, base, base[:7]])          # every train row duplicated: ties must go to the lower index
    q = np.concatenate([base[:20], rand_desc(rng, 20)]) # first 20 have d0 == d1 == 0
    gi, gd = ctx.knn2(q, t)
    oi, od = oracle.knn2(q, t)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    assert (gd[:20] == 0).all() and (gi[:20, 0] == np.arange(20)).all() and (gi[:20, 1] == np.arange(20) + 40).all()
    t2 = np.zeros((5, 32), np.uint8)                   # all-equal train set
    gi, gd = ctx.knn2(q, t2)
    assert (gi[:, 0] == 0).all() and (gi[:, 1] == 1).all()


@pytest.mark.parametrize("ratio", [0.8, 0.5, 1.0, 0.95])
def test_match_ratio(ctx, oracle, ratio):
    rng = np.random.default_rng(11)
    t = rand_desc(rng, 900)
    q = t[rng.permutation(900)[:700]].copy()
    flip = rng.integers(0, 256, size=(700, 4))
    for i in range(700):                               # corrupt up to 4 random bits so that some pass and some fail the ratio
        for b in flip[i][: i % 5]:
            q[i, b // 8] ^= 1 << (b % 8)
    q = np.concatenate([q, rand_desc(rng, 300)])
    g = ctx.match(q, t, ratio)
    o = oracle.match(q, t, ratio)
    assert same_struct(g, o)
    assert (np.diff(g["queryIdx"]) > 0).all() and (g["imgIdx"] == 0).all()


def test_match_edge_cases(ctx):
    import semantic_slam_mapping_amd as ssm
    rng = np.random.default_rng(0)
    assert len(ctx.match(np.zeros((0, 32), np.uint8), rand_desc(rng, 5))) == 0     # empty query set
    with pytest.raises(ssm.SsmError) as e:                                         # orb.cpp:25 would index [1] out of range
        ctx.match(rand_desc(rng, 5), rand_desc(rng, 1))
    assert e.value.code == -5
    with pytest.raises(ssm.SsmError):
        ctx.knn2(rand_desc(rng, 5), np.zeros((0, 32), np.uint8))


# ---------------------------------------------------------------- K1..K5 ORB (include/orb.h:32-53)
def check_orb(ctx, oracle, bgr, depth):
    gk, gd, gp = ctx.detect_features(bgr, depth)
    ok, od = oracle.orb_extract(oracle.bgr2gray(bgr), nfeatures=ctx.cfg.orb_features)
    assert len(gk) == len(ok)
    for f in ("x", "y", "size", "response", "octave", "class_id", "angle"):
        assert np.array_equal(gk[f], ok[f]), f
    assert np.array_equal(gd, od)
    for i in range(len(ok)):
        ref = oracle.project2dTo3d(depth, CAM, int(ok["x"][i]), int(ok["y"][i]))
        assert ref.tobytes() == gp[i].tobytes()
    return gk


def test_orb_synthetic_frames(ctx, oracle, frames):
    for f in (0, 1, 5):
        bgr, dep, _, _, _ = frames[f]
        k = check_orb(ctx, oracle, bgr, dep)
        assert 900 <= len(k) <= ctx.cap
        assert set(np.unique(k["octave"])) == set(range(8))


def test_orb_gray_input_and_strided(ctx, oracle, frames):
    bgr, dep, _, _, _ = frames[2]
    gray = oracle.bgr2gray(bgr)
    k1, d1, _ = ctx.detect_features(gray, dep)
    big = np.zeros((480, 700, 3), np.uint8); big[:, :640] = bgr
    k2, d2, _ = ctx.detect_features(big[:, :640], dep)           # non-contiguous rows (stride 2100)
    ok, od = oracle.orb_extract(gray)
    assert same_struct(k1, ok) and np.array_equal(d1, od) and same_struct(k2, ok) and np.array_equal(d2, od)


def test_orb_low_texture_uses_min_threshold_and_flat_image(ctx, oracle):
    rng = np.random.default_rng(3)
    img = np.full((480, 640), 100, np.uint8)
    img[::40, :] = 112; img[:, ::40] = 112                       # faint grid: corners only reach the min threshold
    img = (img + rng.integers(0, 2, img.shape)).astype(np.uint8)
    k, d, _ = ctx.detect_features(img)
    ok, od = oracle.orb_extract(img)
    assert same_struct(k, ok) and np.array_equal(d, od)
    flat = np.full((480, 640), 77, np.uint8)
    k, d, _ = ctx.detect_features(flat)
    assert len(k) == 0


def test_orb_noise_image_many_candidates(ctx, oracle):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (480, 640), dtype=np.uint8)       # dense corners: stresses the quad-tree split phases
    k, d, _ = ctx.detect_features(img)
    ok, od = oracle.orb_extract(img)
    assert same_struct(k, ok) and np.array_equal(d, od)


# ---------------------------------------------------------------- K10/K11 mapper front half (src/mapper.cpp)
def test_moving_mask(ctx, oracle, frames):
    for f in (0, 3):
        sem = frames[f][2]
        assert np.array_equal(ctx.moving_mask(sem), oracle.moving_mask(sem))
    sem = np.zeros((480, 640, 3), np.uint8)
    sem[0, 0] = (0, 64, 64); sem[479, 639] = (192, 128, 0); sem[100, 637] = (0, 64, 64)   # corners / borders
    m = ctx.moving_mask(sem)
    assert np.array_equal(m, oracle.moving_mask(sem)) and m.sum() == 255 * (9 + 9 + 5 * 5)


@pytest.mark.parametrize("with_pose", [False, True])
def test_generate_point_cloud(ctx, oracle, frames, with_pose):
    bgr, dep, sem, _, T = frames[4]
    if with_pose:
        a = 0.3
        T = np.array([[np.cos(a), -np.sin(a), 0, 0.25], [np.sin(a), np.cos(a), 0, -1.5], [0, 0, 1, 3.0], [0, 0, 0, 1]])
    else:
        T = None
    g = ctx.generate_point_cloud(dep, bgr, sem, T)
    o = oracle.backproject(dep, bgr, sem, oracle.moving_mask(sem), CAM, T, 40.0)
    assert same_struct(g, o)
    assert 100000 < len(g) < 640 * 480


def test_generate_point_cloud_gates(ctx, oracle, frames):
    bgr, dep, sem, _, _ = frames[1]
    dep = dep.copy(); dep[:50] = 0; dep[50:60] = 65535            # holes and beyond max_distance*scale
    g = ctx.generate_point_cloud(dep, bgr, sem, None, max_distance=1.5)
    o = oracle.backproject(dep, bgr, sem, oracle.moving_mask(sem), CAM, None, 1.5)
    assert same_struct(g, o)
    g0 = ctx.generate_point_cloud(np.zeros_like(dep), bgr, sem, None)
    assert len(g0) == 0


# ---------------------------------------------------------------- K12 voxel grid
@pytest.mark.parametrize("leaf", [0.1, 0.02])
def test_voxel_filter(ctx, oracle, frames, leaf):
    bgr, dep, sem, _, T = frames[2]
    pts = ctx.generate_point_cloud(dep, bgr, sem, T)
    g = ctx.voxel_filter(pts, leaf)
    o = oracle.voxel_filter(pts, np.float32(leaf))
    assert same_struct(g, o)
    assert (np.diff(oracle.voxel_table(pts, np.float32(leaf))["key"]) > 0).all()


def test_voxel_negative_coords_and_range_guard(ctx, oracle):
    import semantic_slam_mapping_amd as ssm
    rng = np.random.default_rng(2)
    pts = np.zeros(5000, ssm.POINT_DTYPE)
    pts["x"] = rng.uniform(-3, 3, 5000); pts["y"] = rng.uniform(-2, 2, 5000); pts["z"] = rng.uniform(-1, 1, 5000)
    pts["r"] = rng.integers(0, 256, 5000); pts["g"] = rng.integers(0, 256, 5000); pts["b"] = rng.integers(0, 256, 5000)
    pts["label"] = rng.integers(0, 14, 5000); pts["w"] = 1.0
    assert same_struct(ctx.voxel_filter(pts, 0.25), oracle.voxel_filter(pts, np.float32(0.25)))
    far = pts.copy(); far["x"][0] = 3000.0; far["y"][1] = -3000.0; far["z"][2] = 3000.0
    with pytest.raises(ssm.SsmError) as e:                        # pcl::VoxelGrid: dx*dy*dz > INT_MAX
        ctx.voxel_filter(far, 0.001)
    assert e.value.code == -6
    assert len(ctx.voxel_filter(pts[:0], 0.1)) == 0


def test_voxel_filter_grows_its_table_and_skips_unkeyable_points(oracle):
    """pcl::VoxelGrid has no capacity: a context whose table is far too small must still filter (the temporary table is re-allocated); points that are
    not finite or whose voxel index leaves (-2^20, 2^20) are skipped like PCL skips non-finite points -- same as the oracle -- and the MAP entry
    points report them once with SSM_E_VOXEL_RANGE"""
    import semantic_slam_mapping_amd as ssm
    c = ssm.Context(0, orb_features=500, max_batch=1, voxel_capacity_log2=8, camera=CAM)
    try:
        rng = np.random.default_rng(4)
        pts = np.zeros(20000, ssm.POINT_DTYPE)
        pts["x"] = rng.uniform(-4, 4, len(pts)); pts["y"] = rng.uniform(-3, 3, len(pts)); pts["z"] = rng.uniform(0, 5, len(pts))
        pts["r"] = rng.integers(0, 256, len(pts)); pts["label"] = rng.integers(0, 12, len(pts)); pts["w"] = 1.0
        ref = oracle.voxel_filter(pts, np.float32(0.1))
        assert len(ref) > 4096                                       # 16x the 256-slot table
        assert same_struct(c.voxel_filter(pts, 0.1), ref)
        bad = pts[:2000].copy()
        bad["x"][3] = np.nan; bad["y"][5] = np.inf; bad["z"][7] = -np.inf; bad["x"][11] = 2.0e5          # 2e5 / 0.1 = 2e6 > 2^20
        good = np.delete(bad, [3, 5, 7, 11])
        assert same_struct(oracle.voxel_filter(bad, np.float32(0.1)), oracle.voxel_filter(good, np.float32(0.1)))
        big = ssm.Context(0, orb_features=500, max_batch=1, voxel_capacity_log2=16, camera=CAM)
        try:
            big.map_clear()
            with pytest.raises(ssm.SsmError) as e:
                big.map_insert(bad)
            assert e.value.code == -6
            assert same_struct(big.map_export(), oracle.voxel_filter(good, np.float32(big.cfg.mapper_resolution)))      # flag reported once, map intact
            big.map_insert(good[:10]); big.sync()
        finally:
            big.close()
    finally:
        c.close()


def test_map_at_its_size_limit_refuses_before_it_loses_and_stays_refused_after_a_loss():
    """The context map grows by itself (below); voxel_max_capacity_log2 is the one limit left.  An insert of a known size that cannot fit is refused BEFORE anything
    is added (the map stays what it was); the sequence path cannot know in advance, so contributions that found neither a slot nor a larger table are lost, and then
    (round-2 advisor finding) the incomplete map is reported once by ssm_sync and refused by every export until ssm_map_clear"""
    import semantic_slam_mapping_amd as ssm
    c = ssm.Context(0, orb_features=500, max_batch=2, voxel_capacity_log2=8, voxel_max_capacity_log2=8, camera=CAM)
    try:
        rng = np.random.default_rng(7)
        pts = np.zeros(20000, ssm.POINT_DTYPE)
        pts["x"] = rng.uniform(-4, 4, len(pts)); pts["y"] = rng.uniform(-3, 3, len(pts)); pts["z"] = rng.uniform(0, 5, len(pts)); pts["w"] = 1.0
        c.map_clear()
        c.map_insert(pts[:20]); c.sync()
        before = c.map_export_table().tobytes()
        with pytest.raises(ssm.SsmError) as e:
            c.map_insert(pts)
        assert e.value.code == -4 and "voxel_max_capacity_log2" in str(e.value)
        c.sync()
        assert c.map_export_table().tobytes() == before              # refused, nothing added, nothing lost
        # the sequence path: two frames into 256 slots
        W, H, n = 640, 480, 2
        bufs = [c.dev_alloc(n * W * H * 3), c.dev_alloc(n * W * H * 2), c.dev_alloc(n * W * H * 3), c.dev_alloc(n * 128)]
        c.synth_frames_dev(SEED, 0, n, *bufs)
        with pytest.raises(ssm.SsmError) as e:
            c.seq_process(*bufs, n, stages=ssm.api.STAGE_MAP); c.sync()
        assert e.value.code == -4
        for call in (c.map_size, c.map_export, c.map_export_table):  # the incomplete map is not handed out
            with pytest.raises(ssm.SsmError) as e:
                call()
            assert e.value.code == -4
        c.map_clear()
        c.map_insert(pts[:50]); c.sync()
        assert 0 < c.map_size() <= 50
        for p in bufs:
            c.dev_free(p)
    finally:
        c.close()


def test_context_map_grows_from_a_tiny_table(oracle):
    """the reference's globalMap has no capacity (src/mapper.cpp:121-158): a context that starts with 2^8 slots takes a cloud with thousands of voxels, merges a
    table, and runs the 7-frame sequence (one frame per map launch while the table is small, the overflow list in between) -- the map equals the oracle's"""
    import semantic_slam_mapping_amd as ssm
    c = ssm.Context(0, orb_features=500, max_batch=4, voxel_capacity_log2=8, camera=CAM)
    try:
        rng = np.random.default_rng(11)
        pts = np.zeros(30000, ssm.POINT_DTYPE)
        pts["x"] = rng.uniform(-4, 4, len(pts)); pts["y"] = rng.uniform(-3, 3, len(pts)); pts["z"] = rng.uniform(0, 5, len(pts))
        pts["r"] = rng.integers(0, 256, len(pts)); pts["g"] = rng.integers(0, 256, len(pts)); pts["label"] = rng.integers(0, 12, len(pts)); pts["w"] = 1.0
        leaf = np.float32(c.cfg.mapper_resolution)
        ref = oracle.voxel_filter(pts, leaf)
        assert len(ref) > 20 * 256
        c.map_clear(); c.map_insert(pts[:17000]); c.map_insert(pts[17000:])
        assert same_struct(c.map_export(), ref)
        tab = c.map_export_table()
        c.map_clear(); c.map_insert(pts[:100]); c.map_merge_table(tab)          # a table far larger than the (cleared, but grown) ... and than a fresh context's
        d = ssm.Context(0, orb_features=500, max_batch=4, voxel_capacity_log2=8, camera=CAM)
        try:
            d.map_insert(pts[:100]); d.map_merge_table(tab)
            assert d.map_export_table().tobytes() == c.map_export_table().tobytes()
            assert same_struct(d.map_export(), oracle.voxel_filter(np.concatenate([pts[:100], pts]), leaf))
        finally:
            d.close()
        # the sequence path from 2^8 slots
        W, H, n = 640, 480, 7
        c.map_clear()
        bufs = [c.dev_alloc(n * W * H * 3), c.dev_alloc(n * W * H * 2), c.dev_alloc(n * W * H * 3), c.dev_alloc(n * 128)]
        e = ssm.Context(0, orb_features=500, max_batch=4, voxel_capacity_log2=8, camera=CAM)
        try:
            e.synth_frames_dev(SEED, 0, n, *bufs)       # (device memory is the device's: any context can fill it)
            e.seq_process(*bufs, n, stages=ssm.api.STAGE_MAP); e.sync()
            clouds = []
            for i in range(n):
                bgr, dep, sem, _, T = oracle.synth_frame(SEED, i)
                clouds.append(oracle.backproject(dep, bgr, sem, oracle.moving_mask(sem), CAM, T, 40.0))
            assert same_struct(e.map_export(), oracle.voxel_filter(np.concatenate(clouds), leaf))
        finally:
            e.close()
        for p in bufs:
            c.dev_free(p)
    finally:
        c.close()


def test_configs1_at_leaf_002_from_a_tiny_table_equals_the_large_table_run():
    """SURVEY.md s.8(d)'s second leaf: the 1000-frame configs[1] stream at mapper_resolution 0.02 from voxel_capacity_log2 = 10 (growth by re-hashing between the map
    launches of ssm_seq_process) gives byte for byte the table of a context that starts with 2^24 slots"""
    import semantic_slam_mapping_amd as ssm
    N, W, H = 1000, 640, 480
    tabs = []
    big = ssm.Context(0, orb_features=1000, max_batch=250, voxel_capacity_log2=24, mapper_resolution=0.02, camera=CAM)
    bufs = [big.dev_alloc(N * W * H * 3), big.dev_alloc(N * W * H * 2), big.dev_alloc(N * W * H * 3), big.dev_alloc(N * 128)]
    try:
        big.synth_frames_dev(SEED, 0, N, *bufs)
        big.seq_process(*bufs, N, stages=ssm.api.STAGE_MAP); big.sync()
        tabs.append(big.map_export_table())
        small = ssm.Context(0, orb_features=1000, max_batch=250, voxel_capacity_log2=10, mapper_resolution=0.02, camera=CAM)
        try:
            small.seq_process(*bufs, N, stages=ssm.api.STAGE_MAP); small.sync()
            tabs.append(small.map_export_table())
            # and in two calls with an export in between (the growth state carries over)
            small.map_clear()
            small.seq_process(*bufs, 400, stages=ssm.api.STAGE_MAP); assert small.map_size() > 0
            off = (W * H * 3, W * H * 2, W * H * 3, 128)
            small.seq_process(*[b + 400 * o for b, o in zip(bufs, off)], 600, stages=ssm.api.STAGE_MAP); small.sync()
            tabs.append(small.map_export_table())
        finally:
            small.close()
        assert len(tabs[0]) > 100000
        assert tabs[1].tobytes() == tabs[0].tobytes() and tabs[2].tobytes() == tabs[0].tobytes()
    finally:
        for p in bufs:
            big.dev_free(p)
        big.close()


def test_map_growth_follows_the_streams_voxel_rate_at_a_leaf_below_the_pixel_footprint():
    """leaf 4 mm: nearly every kept pixel is a voxel of its own (~10^5 new voxels per frame), so a 20-frame launch adds more than a 2^20-slot table and its overflow list hold
    together.  The context takes its first frame alone, learns the stream's rate and grows the table for what a launch is expected to add BEFORE the launch: nothing is lost,
    and the table equals the one of a context that starts with 2^25 slots"""
    import semantic_slam_mapping_amd as ssm
    N, W, H = 40, 640, 480
    big = ssm.Context(0, orb_features=500, max_batch=20, voxel_capacity_log2=25, mapper_resolution=0.004, camera=CAM)
    bufs = [big.dev_alloc(N * W * H * 3), big.dev_alloc(N * W * H * 2), big.dev_alloc(N * W * H * 3), big.dev_alloc(N * 128)]
    try:
        big.synth_frames_dev(SEED, 0, N, *bufs)
        big.seq_process(*bufs, N, stages=ssm.api.STAGE_MAP); big.sync()
        ref = big.map_export_table()
        assert len(ref) > (1 << 20) + (1 << 18)                     # more than table + overflow list of the default context
        c = ssm.Context(0, orb_features=500, max_batch=20, voxel_capacity_log2=20, mapper_resolution=0.004, camera=CAM)
        try:
            c.seq_process(*bufs, N, stages=ssm.api.STAGE_MAP); c.sync()
            assert c.map_export_table().tobytes() == ref.tobytes()
            c.map_clear()                                            # (the rate is the stream's: a second pass needs no learning frame and no growth)
            c.seq_process(*bufs, N, stages=ssm.api.STAGE_MAP); c.sync()
            assert c.map_export_table().tobytes() == ref.tobytes()
        finally:
            c.close()
    finally:
        for p in bufs:
            big.dev_free(p)
        big.close()


def test_page_locked_inputs_are_used_in_place_and_give_the_same_features(oracle, frames):
    """ssm_host_alloc (round 6): the synchronous ssm_orb_extract reads a page-locked image with the DMA engine where it is (no staging pass) and the depth image -- needed only
    at the keypoints, by the last kernel -- through the buffer's device mapping.  Same keypoints, descriptors and 3-D positions as with pageable inputs and as the oracle;
    also a strided page-locked image (staged: rows are repacked) and a buffer that is overwritten right after the call returns"""
    import semantic_slam_mapping_amd as ssm
    c = ssm.Context(0, orb_features=1000, max_batch=1, voxel_capacity_log2=14, camera=CAM)
    try:
        bgr, dep = frames[0][0], frames[0][1]
        H, W = dep.shape
        pb = c.host_alloc((H, W, 3), np.uint8); pd = c.host_alloc((H, W), np.uint16)
        pb[:] = bgr; pd[:] = dep
        k0, d0, p0 = c.detect_features(bgr, dep)
        k1, d1, p1 = c.detect_features(pb, pd)
        assert k0.tobytes() == k1.tobytes() and np.array_equal(d0, d1) and np.array_equal(p0, p1)
        check_orb(c, oracle, pb, pd)
        pb[:] = 0; pd[:] = 0                                              # the call has returned: the buffers are the caller's again
        k2, d2, p2 = c.detect_features(bgr, pd)                           # pageable image + page-locked (now zero) depth: features the same, positions the (0, 0, 0) sentinel
        assert k2.tobytes() == k0.tobytes() and np.array_equal(d2, d0) and not p2.any()
        wide = c.host_alloc((H, W + 16, 3), np.uint8); wide[:, :W] = bgr     # a strided view of page-locked memory (the python mirror packs it first: the pageable path)
        k3, d3, _ = c.detect_features(wide[:, :W], dep)
        assert k3.tobytes() == k0.tobytes() and np.array_equal(d3, d0)
        f, t = c.mem_info(); assert 0 < f <= t
        for a in (pb, pd, wide):
            c.host_free(a)
    finally:
        c.close()


def _wall_then_stream(ctx, n_wall, n_all, W=640, H=480):
    """device buffers of the configs[1] stream whose first n_wall frames see a wall at 0.4 m (every depth pixel 400 at scale 1000) and whose other frames are the stream's"""
    bufs = [ctx.dev_alloc(n_all * W * H * 3), ctx.dev_alloc(n_all * W * H * 2), ctx.dev_alloc(n_all * W * H * 3), ctx.dev_alloc(n_all * 128)]
    ctx.synth_frames_dev(SEED, 0, n_all, *bufs)
    ctx.h2d(bufs[1], np.full((n_wall, H, W), 400, np.uint16))
    return bufs


def test_map_stays_lossless_when_the_voxel_rate_jumps(monkeypatch):
    """VERDICT r05 item 5 / ADVICE r05: the table is sized from the stream's own voxel rate, and that rate can jump.  Leaf 4 mm; the first 20 frames see a wall at 0.4 m
    (a pixel's footprint is 0.8 mm there: ~10^4 voxels per frame), the next 20 the normal stream (1 - 5 m: nearly every kept pixel a voxel of its own, ~2 x 10^5 per
    frame).  From 2^10 slots (small launches, settled in front of each) and from 2^20 slots (whole sub-batches per launch, the table grown for the rate learnt on the
    wall): blocks of the map kernel that start while the overflow list is beyond its high-water mark add nothing, log themselves and are run again after the table
    has grown -- no SSM_E_CAPACITY, and the table equals the one of a context that starts with 2^25 slots, byte for byte.  The third context's list is cut down to the
    minimum (test hook), so that the skip / redo path certainly runs; also with the map stage on the chains' own streams (SSM_MAP_STREAM=0)."""
    import semantic_slam_mapping_amd as ssm
    N, NW = 40, 20
    big = ssm.Context(0, orb_features=500, max_batch=20, voxel_capacity_log2=25, mapper_resolution=0.004, camera=CAM)
    bufs = _wall_then_stream(big, NW, N)
    try:
        big.seq_process(*bufs, N, stages=ssm.api.STAGE_MAP); big.sync()
        ref = big.map_export_table()
        per_frame_wall, per_frame_all = None, len(ref) / N
        for log2, hook in ((10, None), (20, None), (20, "1"), (20, "map_stream0")):
            if hook == "1":
                monkeypatch.setenv("SSM_MAP_TEST_SMALL_LIST", "1")
            if hook == "map_stream0":
                monkeypatch.setenv("SSM_MAP_STREAM", "0")
            c = ssm.Context(0, orb_features=500, max_batch=20, voxel_capacity_log2=log2, mapper_resolution=0.004, camera=CAM)
            try:
                stages = ssm.api.STAGE_MAP if hook != "map_stream0" else 0           # (all stages: the two-chain mode whose map stage alternates between streams)
                c.seq_process(*bufs, NW, stages=stages); c.sync()
                if per_frame_wall is None:
                    per_frame_wall = c.map_size() / NW
                    assert per_frame_all > 5 * per_frame_wall                       # the rate does jump
                off = (640 * 480 * 3, 640 * 480 * 2, 640 * 480 * 3, 128)
                c.seq_process(*[b + NW * o for b, o in zip(bufs, off)], N - NW, continue_sequence=True, stages=stages); c.sync()
                assert c.map_export_table().tobytes() == ref.tobytes(), (log2, hook)
                st = c.map_stats()
                assert st[1] >= 1                                                    # it grew
                if hook == "1":
                    assert st[2] > 0, st                                             # and blocks were run again
            finally:
                c.close()
                monkeypatch.delenv("SSM_MAP_TEST_SMALL_LIST", raising=False); monkeypatch.delenv("SSM_MAP_STREAM", raising=False)
    finally:
        for p in bufs:
            big.dev_free(p)
        big.close()


def test_map_more_than_half_full_at_its_size_limit_stays_readable():
    """voxel_max_capacity_log2 bounds the table; a COMPLETE map that is more than half full at that bound has lost nothing and must stay readable
    (size, export, sync) -- only something that cannot be placed is refused"""
    import semantic_slam_mapping_amd as ssm
    c = ssm.Context(0, orb_features=500, max_batch=1, voxel_capacity_log2=10, voxel_max_capacity_log2=10, camera=CAM)
    try:
        rng = np.random.default_rng(3)
        pts = np.zeros(600, ssm.POINT_DTYPE)
        pts["x"] = rng.uniform(-40, 40, len(pts)); pts["y"] = rng.uniform(-30, 30, len(pts)); pts["z"] = rng.uniform(0, 50, len(pts)); pts["w"] = 1.0
        c.map_clear()
        for a in range(0, 600, 100):          # 100 points at a time: each insert announces 100 new voxels, which 1024 slots take until the table is more than half full
            try:
                c.map_insert(pts[a:a + 100])
            except ssm.SsmError as e:
                assert e.code == -4
                break
        n = c.map_size()
        assert 256 < n <= 600
        c.sync()
        assert len(c.map_export()) == n and len(c.map_export_table()) == n
        # fill it beyond a half through the table merge of a map that fits (reserve = its size), then read again
        d = ssm.Context(0, orb_features=500, max_batch=1, voxel_capacity_log2=12, camera=CAM)
        try:
            d.map_insert(pts); tab = d.map_export_table()
        finally:
            d.close()
        assert len(tab) > 512
        with pytest.raises(ssm.SsmError) as e:
            c.map_merge_table(tab)                                                   # cannot be placed: refused before anything is added
        assert e.value.code == -4
        assert c.map_size() == n and len(c.map_export_table()) == n                  # ... and the map stays what it was, readable
    finally:
        c.close()


# ---------------------------------------------------------------- bench.py contract (and its all-gather path on one GPU)
def test_bench_line_and_allgather_path():
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSM_FORCE_MERGE="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--frames", "40", "--batch", "20", "--steps", "2", "--warmup", "1", "--cpu-frames", "3"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["unit"] == "frames/s" and line["n_gpus"] == 1 and line["steps"] == 2 and line["vs_baseline"] is None and line["data"] == "synthetic"
    assert line["value"] > 1000 and set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["value"] > 1
    assert line["per_frame"]["voxels_in_map"] > 500
    assert line["merge_verified"] is True and line["allgather_ms_per_step"] > 0       # SSM_FORCE_MERGE: ssm_voxel_allgather with a 1-rank communicator inside the step


def test_bench_default_line_carries_the_other_configs():
    """the command the driver times (`python bench.py`, N = 1, no mode flag) also reports configs[2] (SegNet), configs[3] (stereo) and the closed pose loop as
    `other_configs`, each with value / ms_per_step / steps / roofline (frac recomputable from achieved / peak) / cpu_baseline.  Run here at a tenth of the size."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--frames", "60", "--batch", "30", "--steps", "1", "--warmup", "1", "--cpu-frames", "3", "--other-scale", "0.1"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["workload"].startswith("configs[1]") and line["unit"] == "frames/s" and "solve_poses" not in line
    oc = line["other_configs"]
    for name, unit, bound in (("configs[2]", "frames/s", "mfma"), ("configs[3]", "frame pairs/s", "hbm"), ("pose_loop", "frames/s", "valu_f64")):
        leg = oc[name]
        assert "error" not in leg, leg
        for k in ("metric", "value", "unit", "steps", "ms_per_step", "roofline", "cpu_baseline", "config"):
            assert k in leg, (name, k)
        assert leg["unit"] == unit and leg["value"] > 0 and leg["steps"] >= 1 and leg["ms_per_step"] > 0
        rf = leg["roofline"]
        assert rf["bound"] == bound and set(("achieved", "peak", "unit", "frac", "traffic")) <= set(rf)
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 2e-3 * max(rf["frac"], 1e-3) + 1e-4
        cb = leg["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == unit
    assert oc["configs[3]"]["cpu_baseline"]["quad_matches_equal_gpu"] is True
    assert oc["pose_loop"]["detail"]["tracked_frames"] == oc["pose_loop"]["detail"]["frames"]
    # round 6: the PRODUCT in the same line -- host/exp_mapping (the C++ drop-in of experiment/exp_mapping.cpp) run as child processes before bench.py's first GPU call:
    # the per-frame Tracker / Mapper loop (stream poses, solved poses), the bulk trackers (RGB-D with and without the chain, stereo), their maps and poses compared
    em = oc["exp_mapping"]
    assert "error" not in em and all("error" not in v for v in em["runs"].values()), em
    for k in ("per_frame_fps", "per_frame_solved_fps", "batched_fps", "batched_solved_fps", "batched_stereo_pairs_per_s", "per_frame_stereo_pairs_per_s"):
        assert em[k] is not None and em[k] > 0, (k, em.get(k))
    assert em["map_fnv_equal"] is True and em["pose_fnv_equal"] is True
    assert em["cpu_baseline"]["value"] > 0 and em["cpu_baseline"]["kind"] == "port"
    assert set(("detect_ms", "match_ms", "pnp_ms")) <= set(em["per_call_ms"]) and all(int(v["lost"]) == 0 for v in em["runs"].values())


def test_full_size_properties_of_configs1(oracle):
    """BASELINE.json configs[1] at its full size (1000 frames of 640x480 resident in HBM, 1000 keypoints, 5 reference
    frames): size-independent properties instead of the oracle, which needs ~0.1 s per frame.
      * batching is invisible: 8 launches of 125 frames and 13 ragged launches of 77(+76) give the same per-frame
        keypoints / descriptors / match tables / point counts (a CRC per frame, then a CRC over the CRCs) and the same map;
      * the map does not depend on frame order (exact integer voxel sums): batches fed last-to-first give the same voxels;
      * sharding is exact: two half-sequence tables merged (the all-gather step of the multi-GPU path) == the whole;
      * spot checks against the oracle inside the long run: ORB of frames 499 and 999, the match table 998 -> 999."""
    import zlib
    import semantic_slam_mapping_amd as ssm
    N, W, H = 1000, 640, 480
    c = ssm.Context(0, orb_features=1000, max_batch=125, voxel_capacity_log2=20, camera=CAM)
    R = c.R
    KEEP = (499, 998, 999); kept = {}
    bufs = [c.dev_alloc(N * W * H * 3), c.dev_alloc(N * W * H * 2), c.dev_alloc(N * W * H * 3), c.dev_alloc(N * 128)]
    off = (W * H * 3, W * H * 2, W * H * 3, 128)
    try:
        for s in range(0, N, 125):
            c.synth_frames_dev(SEED, s, 125, *[b + s * o for b, o in zip(bufs, off)])

        def run(batch, order=None, stages=0, fetch=True):
            crcs = {}
            starts = list(range(0, N, batch)) if order is None else order
            for bi, s in enumerate(starts):
                n = min(batch, N - s)
                out = c.seq_process(*[b + s * o for b, o in zip(bufs, off)], n, continue_sequence=(order is None and bi > 0), stages=stages)
                c.sync()
                if not fetch:
                    continue
                res = c.seq_fetch(out, n)
                for i in range(n):
                    k = int(res["nkp"][i]); h = zlib.crc32(res["kps"][i, :k].tobytes()); h = zlib.crc32(res["desc"][i, :k].tobytes(), h)
                    h = zlib.crc32(res["pos3d"][i, :k].tobytes(), h); h = zlib.crc32(res["nmatch"][i].tobytes(), h)
                    for r in range(R):
                        m = int(res["nmatch"][i, r])
                        if m > 0:
                            h = zlib.crc32(res["matches"][i, r, :m].tobytes(), h)
                    crcs[s + i] = (h, k, int(res["npoints"][i]), tuple(int(v) for v in res["nmatch"][i]))
                    if s + i in KEEP:
                        kept[s + i] = (res["kps"][i, :k].copy(), res["desc"][i, :k].copy(), res["matches"][i, R - 1, :max(int(res["nmatch"][i, R - 1]), 0)].copy())
            return crcs

        c.map_clear(); a = run(125); map_a = c.map_export()
        od = {}
        for f in KEEP:
            ok, od[f] = oracle.orb_extract(oracle.bgr2gray(oracle.synth_frame(SEED, f)[0]), nfeatures=c.cfg.orb_features)
            assert same_struct(kept[f][0], ok) and np.array_equal(kept[f][1], od[f])
        assert same_struct(kept[999][2], oracle.match(od[998], od[999], c.cfg.knn_match_ratio))
        c.map_clear(); b = run(77); map_b = c.map_export()
        assert sorted(a) == list(range(N)) and a == b
        assert zlib.crc32(np.array([a[i][0] for i in range(N)], np.uint32).tobytes()) == zlib.crc32(np.array([b[i][0] for i in range(N)], np.uint32).tobytes())
        assert same_struct(map_a, map_b) and len(map_a) > 10000
        assert all(500 <= a[i][1] <= c.cfg.orb_features + 64 for i in range(N)) and all(a[i][2] > 50000 for i in range(N))
        assert all(a[i][3][:R - min(i, R)] == (-1,) * (R - min(i, R)) and min(a[i][3][R - min(i, R):], default=1) > 0 for i in range(N))
        # order independence of the map (map stage only, batches last to first)
        c.map_clear(); run(125, order=list(range(N - 125, -1, -125)), stages=4, fetch=False)
        assert same_struct(c.map_export(), map_a)
        # two shards merged == whole (ssm_map_export_table / ssm_map_merge_table: the multi-GPU all-gather step)
        c.map_clear(); run(125, order=[0, 125, 250, 375], stages=4, fetch=False); t0 = c.map_export_table()
        c.map_clear(); run(125, order=[500, 625, 750, 875], stages=4, fetch=False); c.map_merge_table(t0)
        assert same_struct(c.map_export(), map_a)
        c.map_clear()
    finally:
        for p in bufs:
            c.dev_free(p)
        c.close()


@pytest.mark.parametrize("scale,levels", [(1.5, 5), (2.0, 3), (1.1, 8)])
def test_orb_other_scale_factors(oracle, frames, scale, levels):
    """pyramid scale factors other than 1.2: at 1.5 and 2.0 the source window of four output pixels no longer fits the
    streaming resize kernel (the general LDS-staged one runs); 1.1 stays on the streaming kernel with other offsets"""
    import semantic_slam_mapping_amd as ssm
    c = ssm.Context(0, orb_features=600, orb_levels=levels, orb_scale=scale, max_batch=1, voxel_capacity_log2=12, camera=CAM)
    try:
        bgr = frames[2][0]
        gk, gd, _ = c.detect_features(bgr)
        ok, od = oracle.orb_extract(oracle.bgr2gray(bgr), nfeatures=600, scale=scale, nlevels=levels)
        assert len(gk) == len(ok) and len(ok) > 300
        for f in ("x", "y", "size", "response", "octave", "class_id", "angle"):
            assert np.array_equal(gk[f], ok[f]), f
        assert np.array_equal(gd, od)
    finally:
        c.close()


@pytest.mark.parametrize("map_stream", ["1", "0"])
def test_two_chain_mode_equals_single_stream(monkeypatch, map_stream):
    """ssm_seq_process runs alternate sub-batches as two chains on two streams (own workspace each, one event per chain for the
    matcher's reference descriptors) and the map stage on a third stream (SSM_MAP_STREAM=0: on the chain's stream, chain 1 first); profiling mode 2 keeps every kernel on one stream with one workspace.  Same outputs, bit for bit, incl. ragged
    last sub-batch and a continued sequence."""
    import semantic_slam_mapping_amd as ssm
    monkeypatch.setenv("SSM_MAP_STREAM", map_stream)
    W, H, n = 640, 480, 23
    c = ssm.Context(0, orb_features=500, max_batch=4, voxel_capacity_log2=18, camera=CAM)
    bufs = [c.dev_alloc(n * W * H * 3), c.dev_alloc(n * W * H * 2), c.dev_alloc(n * W * H * 3), c.dev_alloc(n * 128)]
    off = (W * H * 3, W * H * 2, W * H * 3, 128)
    try:
        c.synth_frames_dev(SEED, 200, n, *bufs)
        def run(mode):
            c.set_profiling(mode); c.map_clear()
            out = c.seq_process(*bufs, 15); c.sync(); a = c.seq_fetch(out, 15)
            out = c.seq_process(*[b + 15 * o for b, o in zip(bufs, off)], n - 15, continue_sequence=True); c.sync(); b = c.seq_fetch(out, n - 15)
            c.set_profiling(0)
            return a, b, c.map_export()
        a1, b1, m1 = run(0)
        a2, b2, m2 = run(2)
        for x, y in ((a1, a2), (b1, b2)):
            for k in ("nkp", "nmatch", "npoints"):
                assert np.array_equal(x[k], y[k]), k
            for f in range(len(x["nkp"])):
                kk = int(x["nkp"][f])
                assert same_struct(x["kps"][f, :kk], y["kps"][f, :kk]) and np.array_equal(x["desc"][f, :kk], y["desc"][f, :kk])
                for r in range(x["nmatch"].shape[1]):
                    m = int(x["nmatch"][f, r])
                    if m > 0:
                        assert same_struct(x["matches"][f, r, :m], y["matches"][f, r, :m])
        assert same_struct(m1, m2) and len(m1) > 1000
        assert (b1["nmatch"][0] > 0).all()                       # the continued call saw the previous call's frames as references
    finally:
        for p in bufs:
            c.dev_free(p)
        c.close()


def test_map_reads_on_the_map_stream_stay_ordered_with_the_calls_around_them(oracle):
    """ssm_map_size / ssm_map_export_table_dev wait for the MAP's stream only (round 5): after an ssm_seq_process whose map stage ran on a side stream they return
    while the call's ORB -> match chain may still be running.  Whatever is called next must still see, and be seen by, the map in call order: an export in the
    middle of a run of sequences, a host insert right behind it, a second sequence on top, a clear, the outputs of the chain fetched last."""
    import semantic_slam_mapping_amd as ssm
    from semantic_slam_mapping_amd.api import VOXEL_DTYPE, POINT_DTYPE
    W, H, n = 640, 480, 12
    c = ssm.Context(0, orb_features=500, max_batch=4, voxel_capacity_log2=18, camera=CAM)
    bufs = [c.dev_alloc(n * W * H * 3), c.dev_alloc(n * W * H * 2), c.dev_alloc(n * W * H * 3), c.dev_alloc(n * 128)]
    cap = 1 << 17
    dtab = c.dev_alloc(cap * VOXEL_DTYPE.itemsize)
    rng = np.random.default_rng(SEED + 11)
    extra = np.zeros(3000, POINT_DTYPE)
    extra["x"], extra["y"], extra["z"] = rng.uniform(-2, 2, 3000), rng.uniform(-2, 2, 3000), rng.uniform(0.5, 4, 3000)
    extra["r"], extra["g"], extra["b"] = rng.integers(0, 256, (3, 3000))
    try:
        c.synth_frames_dev(SEED, 300, n, *bufs)
        def table_dev():
            k = c.map_export_table_dev(dtab, cap)
            return c.d2h(dtab, k, VOXEL_DTYPE).copy()
        # the reference: every step followed by a full ssm_sync, tables through the host export (which works on the context stream)
        c.set_profiling(2); c.map_clear()
        out = c.seq_process(*bufs, n); c.sync(); ref_seq = c.seq_fetch(out, n); t1 = c.map_export_table()
        c.map_insert(extra); c.sync(); t2 = c.map_export_table()
        c.seq_process(*bufs, n); c.sync(); t3 = c.map_export_table()
        c.set_profiling(0)
        # the same calls back to back, the map read on its own stream
        c.map_clear()
        out = c.seq_process(*bufs, n)
        g1 = table_dev()                                         # (the chain of the last sub-batch may still be running here)
        c.map_insert(extra)
        assert c.map_size() == len(t2)
        g2 = table_dev()
        c.seq_process(*bufs, n)
        assert c.map_size() == len(t3)
        g3 = table_dev()
        c.map_clear()
        assert c.map_size() == 0
        out = c.seq_process(*bufs, n)
        g4 = table_dev()
        got_seq = c.seq_fetch(out, n)                            # the chain's outputs, ordered on the context stream
        for a, b in ((t1, g1), (t2, g2), (t3, g3), (t1, g4)):
            assert len(a) == len(b) > 1000 and a.tobytes() == b.tobytes()
        for k in ("nkp", "nmatch", "npoints"):
            assert np.array_equal(ref_seq[k], got_seq[k]), k
        for f in range(n):
            kk = int(ref_seq["nkp"][f])
            assert same_struct(ref_seq["kps"][f, :kk], got_seq["kps"][f, :kk]) and np.array_equal(ref_seq["desc"][f, :kk], got_seq["desc"][f, :kk])
    finally:
        for p in bufs + [dtab]:
            c.dev_free(p)
        c.close()


@pytest.mark.parametrize("max_batch,n", [(2, 9), (1, 7), (3, 11)])
def test_three_chains_with_sub_batches_shorter_than_the_reference_window(max_batch, n):
    """max_batch < tracker_ref_frames: the matcher of a sub-batch reads descriptor rows of SEVERAL preceding sub-batches, which run on the other two
    chains' streams -- it must wait for every chain's newest ORB event, not only its predecessor's (round-2 advisor finding).  Ten repetitions of the
    three-chain run against the serialised run (profiling mode 2: one stream, one workspace)."""
    import semantic_slam_mapping_amd as ssm
    W, H = 640, 480
    c = ssm.Context(0, orb_features=500, max_batch=max_batch, voxel_capacity_log2=18, camera=CAM)
    bufs = [c.dev_alloc(n * W * H * 3), c.dev_alloc(n * W * H * 2), c.dev_alloc(n * W * H * 3), c.dev_alloc(n * 128)]
    try:
        c.synth_frames_dev(SEED, 300, n, *bufs)
        def run(mode):
            c.set_profiling(mode); c.map_clear()
            out = c.seq_process(*bufs, n); c.sync(); a = c.seq_fetch(out, n)
            c.set_profiling(0)
            return a, c.map_export()
        ref, mref = run(2)
        assert (ref["nmatch"][c.R:] > 0).all()
        for rep in range(10):
            a, m = run(0)
            for k in ("nkp", "nmatch", "npoints"):
                assert np.array_equal(a[k], ref[k]), (rep, k)
            for f in range(n):
                kk = int(ref["nkp"][f])
                assert np.array_equal(a["desc"][f, :kk], ref["desc"][f, :kk]), (rep, f)
                for r in range(c.R):
                    mm = int(ref["nmatch"][f, r])
                    if mm > 0:
                        assert same_struct(a["matches"][f, r, :mm], ref["matches"][f, r, :mm]), (rep, f, r)
            assert same_struct(m, mref)
    finally:
        for p in bufs:
            c.dev_free(p)
        c.close()


@pytest.mark.parametrize("nfeat", [2000, 4500])
def test_orb_more_features_use_the_larger_quadtree_variants(oracle, frames, nfeat):
    """the quad-tree kernel has 256-, 512- and 1024-node instantiations chosen by the feature count (1000 features: 256);
    2000 and 4500 features exercise the other two"""
    import semantic_slam_mapping_amd as ssm
    c = ssm.Context(0, orb_features=nfeat, max_batch=1, voxel_capacity_log2=12, camera=CAM)
    try:
        bgr = frames[3][0]
        gk, gd, _ = c.detect_features(bgr)
        ok, od = oracle.orb_extract(oracle.bgr2gray(bgr), nfeatures=nfeat)
        assert len(gk) == len(ok) and len(ok) > nfeat * 0.8
        for f in ("x", "y", "size", "response", "octave", "class_id", "angle"):
            assert np.array_equal(gk[f], ok[f]), f
        assert np.array_equal(gd, od)
    finally:
        c.close()


def test_fast_stage_overflow_path_in_subprocess():
    """a FAST tile stages its local maxima in LDS (680 of them) and reserves their place in the global list once; more than
    that go to the list one by one.  No test image has such a tile (noise: 452 at most), so the path is forced with a
    16-entry staging area (SSM_FAST_STAGE_CAP, read once per process) and the ORB parity tests are run again"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSM_FAST_STAGE_CAP="16")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu",
                        "-k", "orb_synthetic_frames or orb_noise_image or orb_gray_input or seq_process_matches_oracle"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "deselected" in r.stdout


def test_sequence_path_small_leaf_and_many_keypoints(oracle):
    """two fallbacks of the sequence path: (1) a 1 cm leaf gives far more voxels per 12 K pixels than the block-local LDS table of
    map_stream_kernel holds (256), so most updates take the direct global path; (2) 1500 features per frame means train sets
    above the matcher's 1024-descriptor LDS chunk (two chunks per pair, in both halves of the split pair).  Both against the oracle."""
    import semantic_slam_mapping_amd as ssm
    W, H, n = 640, 480, 4
    c = ssm.Context(0, orb_features=1500, max_batch=2, voxel_capacity_log2=22, camera=CAM, mapper_resolution=0.01)
    R = c.R
    bufs = [c.dev_alloc(n * W * H * 3), c.dev_alloc(n * W * H * 2), c.dev_alloc(n * W * H * 3), c.dev_alloc(n * 128)]
    try:
        c.synth_frames_dev(SEED, 300, n, *bufs)
        fr = [oracle.synth_frame(SEED, 300 + i) for i in range(n)]
        c.map_clear()
        out = c.seq_process(*bufs, n); c.sync()
        res = c.seq_fetch(out, n)
        descs, clouds = [], []
        for i in range(n):
            ok, od = oracle.orb_extract(oracle.bgr2gray(fr[i][0]), nfeatures=1500)
            k = int(res["nkp"][i])
            assert k == len(ok) and k > 1100 and same_struct(res["kps"][i, :k], ok) and np.array_equal(res["desc"][i, :k], od)
            descs.append(od)
            for r in range(R):
                ref = i - R + r
                if ref < 0:
                    continue
                om = oracle.match(descs[ref], od, c.cfg.knn_match_ratio)
                assert res["nmatch"][i, r] == len(om) and same_struct(res["matches"][i, r, :len(om)], om)
            clouds.append(oracle.backproject(fr[i][1], fr[i][0], fr[i][2], oracle.moving_mask(fr[i][2]), CAM, fr[i][4], 40.0))
        ref_map = oracle.voxel_filter(np.concatenate(clouds), np.float32(0.01))
        got = c.map_export()
        assert len(ref_map) > 50000 and same_struct(got, ref_map)
    finally:
        for p in bufs:
            c.dev_free(p)
        c.close()


def test_sequence_with_sparse_and_empty_frames(oracle):
    """frames with very few keypoints inside a sequence: a frame with a small textured patch (tens of keypoints: the second block of
    a split match pair then has no queries and only appends the first block's count), a flat frame (no keypoints at all: every
    pair that has it as the train set reports -1 like a missing reference, pairs that have it as the query set report 0 matches)"""
    import semantic_slam_mapping_amd as ssm
    W, H, n = 640, 480, 5
    c = ssm.Context(0, orb_features=1000, max_batch=2, voxel_capacity_log2=18, camera=CAM)
    R = c.R
    fr = [list(oracle.synth_frame(SEED, 400 + i)) for i in range(n)]
    patch = fr[1][0][200:260, 300:380].copy()
    fr[1][0] = np.full_like(fr[1][0], 90); fr[1][0][200:260, 300:380] = patch          # sparse frame
    fr[3][0] = np.full_like(fr[3][0], 120)                                              # flat frame
    bufs = [c.dev_alloc(n * W * H * 3), c.dev_alloc(n * W * H * 2), c.dev_alloc(n * W * H * 3), c.dev_alloc(n * 128)]
    try:
        c.h2d(bufs[0], np.stack([f[0] for f in fr])); c.h2d(bufs[1], np.stack([f[1] for f in fr])); c.h2d(bufs[2], np.stack([f[2] for f in fr]))
        c.h2d(bufs[3], np.stack([f[4].T.reshape(16) for f in fr]))
        out = c.seq_process(*bufs, n, stages=3); c.sync()
        res = c.seq_fetch(out, n)
        descs = []
        for i in range(n):
            ok, od = oracle.orb_extract(oracle.bgr2gray(fr[i][0]), nfeatures=1000)
            k = int(res["nkp"][i])
            assert k == len(ok) and same_struct(res["kps"][i, :k], ok) and np.array_equal(res["desc"][i, :k], od)
            descs.append(od)
        assert 0 < len(descs[1]) < 200 and len(descs[3]) == 0
        for i in range(n):
            for r in range(R):
                ref = i - R + r
                if ref < 0 or len(descs[i]) < 2:
                    assert res["nmatch"][i, r] == -1
                    continue
                om = oracle.match(descs[ref], descs[i], c.cfg.knn_match_ratio) if len(descs[ref]) else np.zeros(0, res["matches"].dtype)
                assert res["nmatch"][i, r] == len(om) and same_struct(res["matches"][i, r, :len(om)], om)
    finally:
        for p in bufs:
            c.dev_free(p)
        c.close()


def test_valu_variants_of_matcher_and_blur(oracle, frames, monkeypatch):
    """The matcher and the blur have two implementations each: the default (matrix cores) and the earlier one
    (SSM_MATCH_VARIANT=0 / SSM_BLUR_VARIANT=0, read when a context is created).  The rest of this file runs the defaults; this runs
    the earlier kernels through the same checks: host matcher API, ORB of one frame (the descriptors depend on the blur), and a short sequence (match
    tables, point counts and the fused map) against the oracle."""
    import semantic_slam_mapping_amd as ssm
    monkeypatch.setenv("SSM_MATCH_VARIANT", "0")
    monkeypatch.setenv("SSM_BLUR_VARIANT", "0")
    c = ssm.Context(0, orb_features=1000, max_batch=4, voxel_capacity_log2=18, camera=CAM)
    try:
        rng = np.random.default_rng(77)
        base = rand_desc(rng, 300)
        t = np.concatenate([base, base[:50], rand_desc(rng, 683)])      # ties inside, 1033 rows: one partial tile in the matrix-core layout
        q = np.concatenate([base[:100], rand_desc(rng, 400)])
        gi, gd = c.knn2(q, t)
        oi, od = oracle.knn2(q, t)
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)
        assert same_struct(c.match(q, t, 0.8), oracle.match(q, t, 0.8))
        check_orb(c, oracle, frames[0][0], frames[0][1])
        n, W, H, R = 6, 640, 480, c.R
        bufs = [c.dev_alloc(n * W * H * 3), c.dev_alloc(n * W * H * 2), c.dev_alloc(n * W * H * 3), c.dev_alloc(n * 128)]
        try:
            c.synth_frames_dev(SEED, 20, n, *bufs)
            c.map_clear()
            out = c.seq_process(*bufs, n)
            c.sync()
            res = c.seq_fetch(out, n)
            descs, clouds = [], []
            for i in range(n):
                fr = oracle.synth_frame(SEED, 20 + i)
                cl = oracle.backproject(fr[1], fr[0], fr[2], oracle.moving_mask(fr[2]), CAM, fr[4], 40.0)
                assert res["npoints"][i] == len(cl)
                clouds.append(cl)
                ok, od = oracle.orb_extract(oracle.bgr2gray(fr[0]), nfeatures=1000)
                k = res["nkp"][i]
                assert k == len(ok) and np.array_equal(res["desc"][i, :k], od)
                descs.append(od)
                for r in range(R):
                    ref = i - R + r
                    if ref < 0:
                        assert res["nmatch"][i, r] == -1
                        continue
                    om = oracle.match(descs[ref], od, c.cfg.knn_match_ratio)
                    assert res["nmatch"][i, r] == len(om) and same_struct(res["matches"][i, r, :len(om)], om)
            assert same_struct(c.map_export(), oracle.voxel_filter(np.concatenate(clouds), np.float32(c.cfg.mapper_resolution)))
        finally:
            for b in bufs:
                c.dev_free(b)
    finally:
        c.close()


@pytest.mark.parametrize("nq,nt", [(5, 33), (31, 32), (300, 31), (256, 257), (513, 4097)])
def test_knn2_tile_boundaries(ctx, oracle, nq, nt):
    """train counts around the 32-row tiles and query counts around the 64 / 256-query blocks of the matrix-core matcher, with many exact ties"""
    rng = np.random.default_rng(nq * 31 + nt)
    few = rand_desc(rng, 7)
    t = few[rng.integers(0, 7, size=nt)]                     # only 7 distinct train rows: every query has large tie groups
    q = np.concatenate([few, rand_desc(rng, nq)])[:nq]
    gi, gd = ctx.knn2(q, t)
    oi, od = oracle.knn2(q, t)
    assert np.array_equal(gd, od) and np.array_equal(gi, oi)


# ---------------------------------------------------------------- device-resident Mapper (round 4)
def test_viewer_map_on_device_equals_the_host_schedule(ctx, oracle, frames):
    """Mapper::viewer (/root/reference/src/mapper.cpp:96-171) with the key-frame clouds and the map in HBM: ssm_backproject_dev keeps a frame's gated camera-frame
    cloud on the device (frame->pointcloud, mapper.cpp:17-20), ssm_viewer_map_update transforms the chosen clouds by their CURRENT poses, adds the previous
    centroids and runs the VoxelGrid pass.  A viewer schedule (rebuild from every 2nd key-frame at update 0 and 15, the last <= 5 key-frames otherwise, poses that
    change between updates like a pose-graph correction) must give, update by update, the bytes of the host form: ssm_backproject (T = NULL) -> transform in
    double on the host (pcl::transformPointCloud's arithmetic) -> concatenate -> ssm_voxel_filter; and the oracle's filter of the same points."""
    rng = np.random.default_rng(7)
    host_clouds = [ctx.generate_point_cloud(f[1], f[0], f[2]) for f in frames]
    dev_clouds = [ctx.backproject_dev(f[1], f[0], f[2]) for f in frames]
    try:
        assert [ctx.cloud_size(c) for c in dev_clouds] == [len(h) for h in host_clouds]
        assert same_struct(ctx.cloud_fetch(dev_clouds[0]), host_clouds[0])

        def transform(cl, T):
            out = cl.copy(); x, y, z = (cl[k].astype(np.float64) for k in "xyz")
            for i, k in enumerate("xyz"):
                out[k] = (T[i, 0] * x + T[i, 1] * y + T[i, 2] * z + T[i, 3]).astype(np.float32)      # left to right, every product and sum rounded to double
            return out

        def pose(k, jitter):
            T = frames[k][4].copy(); a = 0.02 * jitter
            R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
            T[:3, :3] = R @ T[:3, :3]; T[:3, 3] += 0.013 * jitter
            return T
        assert same_struct(ctx.cloud_fetch(dev_clouds[3], pose(3, 2)), transform(host_clouds[3], pose(3, 2)))
        host_map = np.zeros(0, host_clouds[0].dtype)
        for upd in range(18):
            nkf = min(len(frames), 2 + upd // 2)                                  # key-frames known at this update
            rebuild = upd % 15 == 0
            sel = list(range(0, nkf, 2)) if rebuild else list(range(nkf - 1, max(nkf - 6, -1), -1))
            poses = [pose(k, upd % 4) for k in sel]                               # the pose graph moves key-frames between updates
            parts = ([] if rebuild else [host_map]) + [transform(host_clouds[k], T) for k, T in zip(sel, poses)]
            allp = np.concatenate(parts)
            host_map = ctx.voxel_filter(allp, leaf=0.1)
            n = ctx.viewer_map_update([dev_clouds[k] for k in sel], poses, rebuild=rebuild, leaf=0.1)
            dev_map = ctx.viewer_map_fetch(n)
            assert n == len(host_map) and same_struct(dev_map, host_map), f"update {upd}: {n} vs {len(host_map)} voxels"
            if upd in (0, 7):
                assert same_struct(dev_map, oracle.voxel_filter(allp, np.float32(0.1)))
        # an update that adds nothing (the reference's size_t wrap with fewer than 6 key-frames, mapper.cpp:134) re-filters the previous centroids
        n = ctx.viewer_map_update([], [], rebuild=False, leaf=0.1)
        assert same_struct(ctx.viewer_map_fetch(n), ctx.voxel_filter(host_map, leaf=0.1))
        # rebuild from nothing: empty map
        assert ctx.viewer_map_update([], [], rebuild=True, leaf=0.1) == 0 and len(ctx.viewer_map_fetch(0)) == 0
    finally:
        for c in dev_clouds:
            ctx.cloud_free(c)


# ---------------------------------------------------------------- asynchronous per-frame calls (round 4)
def test_async_per_frame_calls_equal_the_synchronous_ones(ctx, oracle, frames):
    """ssm_orb_extract_async / ssm_match_async + ssm_wait (the calls of Tracker::trackRefFrame, /root/reference/src/track.cpp:140-163, enqueued back to back):
    the same bytes as the synchronous forms and as the oracle; many calls pending at once, in call order; more pending matcher calls than the staging ring
    holds (the call that finds it full completes the earlier ones first); a failing call (one train descriptor) leaves the pending ones intact"""
    from semantic_slam_mapping_amd.api import SsmError
    sync = [ctx.detect_features(f[0], f[1]) for f in frames[:4]]
    hold = [ctx.detect_features_async(f[0], f[1]) for f in frames[:4]]            # four extractions in flight on one workspace: stream order keeps them apart
    ctx.wait()
    for (k, d, p), h in zip(sync, hold):
        ak, ad, ap = h()
        assert same_struct(ak, k) and np.array_equal(ad, d) and np.array_equal(ap, p)
    ok, od = oracle.orb_extract(oracle.bgr2gray(frames[0][0]), nfeatures=1000)
    assert same_struct(sync[0][0], ok) and np.array_equal(sync[0][1], od)
    # a tracker frame: the current frame against its reference frames, one wait
    cur = sync[3][1]
    hm = [ctx.match_async(sync[i][1], cur) for i in range(3)]
    ctx.wait()
    for i in range(3):
        assert same_struct(hm[i](), oracle.match(sync[i][1], cur, 0.8)) and same_struct(hm[i](), ctx.match(sync[i][1], cur))
    # 40 matcher calls pending: far more than the 8 MB rings hold; mixed sizes and ratios
    rng = np.random.default_rng(3)
    sets = [rng.integers(0, 256, (int(n), 32), dtype=np.uint8) for n in rng.integers(2, 1100, 12)]
    jobs = [(sets[int(a)], sets[int(b)], float(r)) for a, b, r in zip(rng.integers(0, 12, 40), rng.integers(0, 12, 40), rng.choice([0.6, 0.8, 1.0], 40))]
    hs = [ctx.match_async(q, t, r) for q, t, r in jobs]
    ctx.wait()
    for h, (q, t, r) in zip(hs, jobs):
        assert same_struct(h(), oracle.match(q, t, r))
    # an invalid call between pending ones is refused at once and does not disturb them
    h0 = ctx.match_async(sets[0], sets[1])
    with pytest.raises(SsmError):
        ctx.match_async(sets[0], sets[1][:1])
    h1 = ctx.match_async(sets[2], sets[3])
    ctx.sync()                                                                   # ssm_sync completes pending calls too
    assert same_struct(h0(), oracle.match(sets[0], sets[1], 0.8)) and same_struct(h1(), oracle.match(sets[2], sets[3], 0.8))


def test_match_refs_equals_pairwise_match(ctx, oracle, frames):
    """ssm_match_refs: Tracker::trackRefFrame's loop over the reference frames (/root/reference/src/track.cpp:150-152) as one launch -- list i must be
    ssm_match(refs[i], cur) = the oracle's; reference sets of different sizes, an empty one, more rows than the current frame, one reference only"""
    from semantic_slam_mapping_amd.api import SsmError
    rng = np.random.default_rng(11)
    descs = [ctx.detect_features(f[0], f[1])[1] for f in frames[:6]]
    cur = descs[5]
    for refs in (descs[:5], [descs[0][:300], descs[1][:0], descs[2], rng.integers(0, 256, (1500, 32), dtype=np.uint8), descs[3][:1]], [descs[4]]):
        got = ctx.match_refs(refs, cur)
        assert len(got) == len(refs)
        for r, g in zip(refs, got):
            ref = oracle.match(r, cur, 0.8) if len(r) else np.zeros(0, g.dtype)
            assert same_struct(g, ref)
            if len(r):
                assert same_struct(g, ctx.match(r, cur))
    assert ctx.match_refs([], cur) == []
    with pytest.raises(SsmError):
        ctx.match_refs(descs[:2], cur[:1])                                      # knnMatch(k = 2) on one train descriptor
    for ratio in (0.6, 1.0):
        got = ctx.match_refs(descs[:3], cur, ratio)
        assert all(same_struct(g, oracle.match(r, cur, ratio)) for r, g in zip(descs[:3], got))

