"""Python mirror of the reference's operator interface over the C ABI.

Names follow the reference: ``detect_features`` = OrbFeature::detectFeatures (include/orb.h:32-53), ``match`` =
OrbFeature::match (src/orb.cpp:16-29), ``moving_mask`` = Mapper::semantic_motion_fuse (src/mapper.cpp:189-216),
``generate_point_cloud`` = Mapper::generatePointCloud (src/mapper.cpp:12-94), ``voxel_filter`` = pcl::VoxelGrid as
used in Mapper::viewer (src/mapper.cpp:154-155).
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import Camera, Config, FramesDev, SeqOutDev, SgbmParams, VoParams, StereoFramesDev, StereoOutDev, TrackerParams

KEYPOINT_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"),
                           ("octave", "i4"), ("class_id", "i4")])
DMATCH_DTYPE = np.dtype([("queryIdx", "i4"), ("trainIdx", "i4"), ("imgIdx", "i4"), ("distance", "f4")])
POINT_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("w", "f4"), ("b", "u1"), ("g", "u1"), ("r", "u1"),
                        ("a", "u1"), ("label", "u4"), ("pad", "u4", (2,))])
VOXEL_DTYPE = np.dtype([("key", "i8"), ("sx", "i8"), ("sy", "i8"), ("sz", "i8"), ("sr", "u8"), ("sg", "u8"),
                        ("sb", "u8"), ("n", "u8"), ("hist", "u4", (12,))])
PMATCH_DTYPE = np.dtype([("u1p", "f4"), ("v1p", "f4"), ("i1p", "i4"), ("u2p", "f4"), ("v2p", "f4"), ("i2p", "i4"), ("u1c", "f4"), ("v1c", "f4"),
                         ("i1c", "i4"), ("u2c", "f4"), ("v2c", "f4"), ("i2c", "i4"), ("dis_c", "i2"), ("dis_p", "i2")])
VO_PARAMS_DTYPE = np.dtype([("f", "f8"), ("cu", "f8"), ("cv", "f8"), ("base", "f8"), ("inlier_threshold", "f8"), ("reweighting", "i4"), ("pad", "i4")])
assert PMATCH_DTYPE.itemsize == 52
assert KEYPOINT_DTYPE.itemsize == 28 and DMATCH_DTYPE.itemsize == 16 and POINT_DTYPE.itemsize == 32 and VOXEL_DTYPE.itemsize == 112

STAGE_ORB, STAGE_MATCH, STAGE_MAP, STAGE_SEGNET = 1, 2, 4, 8
STEREO_QUAD, STEREO_DEPTH, STEREO_VO = 1, 2, 4
COMM_ID_BYTES = 128
SEG_NET_W, SEG_NET_H, SEG_CLASSES = 480, 360, 12


class GlibcRand:
    """The rand() stream VisualOdometry::getRandomSample draws from (srand(0) in the constructor, src/vo.cpp:17): glibc's TYPE_3 additive
    feedback generator, private to the object like in include/ssm/vo_stereo.hpp.  `draws(k)` returns the next k raw outputs (what the batched
    stereo path takes as rand_stream); `rewind(k)` gives unused draws back (frames with fewer than 6 quad matches draw nothing)."""

    def __init__(self, seed=0):
        seed = seed or 1
        r = [seed & 0xFFFFFFFF]
        for i in range(1, 31):
            hi, lo = divmod(r[i - 1] if r[i - 1] < 2 ** 31 else r[i - 1] - 2 ** 32, 127773)
            w = 16807 * lo - 2836 * hi
            if w < 0:
                w += 2147483647
            r.append(w & 0xFFFFFFFF)
        for i in range(31, 34):
            r.append(r[i - 31])
        self._r = r                      # the whole history is kept: rewinding is an index move
        self._pos = 34
        for _ in range(310):
            self._step()
        self._out = []                   # outputs from position _base on
        self._taken = 0

    def _step(self):
        r = self._r
        r.append((r[-3] + r[-31]) & 0xFFFFFFFF)
        return r[-1] >> 1

    def draws(self, k):
        while len(self._out) < self._taken + k:
            self._out.append(self._step())
        a = np.array(self._out[self._taken:self._taken + k], np.uint32)
        self._taken += k
        return a

    def rewind(self, k):
        assert 0 <= k <= self._taken
        self._taken -= k


class SsmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ssm error {code}: {msg}")
        self.code = code


def default_config(**kw):
    lib = _lib.load()
    cfg = Config()
    lib.ssm_config_default(C.byref(cfg))
    cam = kw.pop("camera", None)
    if cam is not None:
        cfg.camera = Camera(*cam)
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise KeyError(k)
        setattr(cfg, k, v)
    return cfg


def _ptr(a):
    return a.ctypes.data if a is not None else None


TRACK_INFO_DTYPE = np.dtype([("state", "i4"), ("tracked", "i4"), ("n_matches", "i4"), ("n_inliers", "i4")])


class Tracker:
    """ssm_tracker: Tracker::updateFrame (RGB-D mode, src/track.cpp:8-36,140-212) for all frames of a seq_process call -- the bulk consumer of the
    match tables.  run(out, n) -> (poses n x 4 x 4 = T_f_w per frame, info structured array)."""

    def __init__(self, ctx, max_lost_frame=10, pnp_min_inliers=10, use_device=False, first_pose=None, own_stream=False, blocks=0):
        self.ctx = ctx; self.lib = ctx.lib
        p = TrackerParams()
        self.lib.ssm_tracker_params_default(C.byref(p))
        p.max_lost_frame = max_lost_frame; p.ref_frames = ctx.R; p.pnp_min_inliers = pnp_min_inliers; p.use_device = int(use_device); p.own_stream = int(own_stream); p.blocks = int(blocks)
        if first_pose is not None:
            fp = np.ascontiguousarray(np.asarray(first_pose, np.float64).reshape(4, 4).T).reshape(16)       # column-major
            for i in range(16):
                p.first_pose[i] = float(fp[i])
        h = C.c_void_p()
        rc = self.lib.ssm_tracker_create(ctx.h, C.byref(p), C.byref(h))
        if rc != 0:
            raise SsmError(rc, "ssm_tracker_create")
        self.h = h

    def reset(self):
        self.lib.ssm_tracker_reset(self.h)

    def run(self, seq_out, n):
        poses = np.zeros((max(n, 1), 16), np.float64); info = np.zeros(max(n, 1), TRACK_INFO_DTYPE)
        rc = self.lib.ssm_tracker_run(self.h, C.byref(seq_out), n, _ptr(poses), _ptr(info))
        if rc != 0:
            raise SsmError(rc, (self.lib.ssm_tracker_last_error(self.h) or b"").decode())
        return poses[:n].reshape(n, 4, 4).transpose(0, 2, 1).copy(), info[:n]

    def last_error(self):
        """the last call's error text, or a note (a downgrade of the device chain to one block) after a call that succeeded"""
        return (self.lib.ssm_tracker_last_error(self.h) or b"").decode()

    def stats(self):
        """(frames solved by the device chain, frames solved by the host path)"""
        a, b = C.c_int64(0), C.c_int64(0)
        self.lib.ssm_tracker_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def work(self):
        """device chain so far: (Levenberg iterations, chi2 passes, edges evaluated by the iterations' fused passes, edges evaluated by the chi2 passes)"""
        w = (C.c_int64 * 4)()
        self.lib.ssm_tracker_work(self.h, C.byref(w))
        return tuple(int(x) for x in w)

    def close(self):
        if self.h:
            self.lib.ssm_tracker_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One ssm_ctx: device workspace + stream on one GPU."""

    def __init__(self, device=0, cfg=None, **kw):
        self.lib = _lib.load()
        self.cfg = cfg if cfg is not None else default_config(**kw)
        self._pattern_keep = None
        self._inflight = []                 # output buffers of asynchronous calls that have not been waited for
        h = C.c_void_p()
        rc = self.lib.ssm_create(device, C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise SsmError(rc, (self.lib.ssm_last_error(None) or b"").decode())
        self.h = h
        self.cap = self.lib.ssm_orb_capacity(self.h)
        self.W, self.H = self.cfg.width, self.cfg.height
        self.R = max(1, self.cfg.tracker_ref_frames)

    def close(self):
        if getattr(self, "h", None):
            self.lib.ssm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise SsmError(rc, (self.lib.ssm_last_error(self.h) or b"").decode())

    # ---- OrbFeature
    def detect_features(self, img, depth=None):
        img = np.ascontiguousarray(img, np.uint8)
        ch = 1 if img.ndim == 2 else img.shape[2]
        h, w = img.shape[:2]
        if depth is not None:
            depth = np.ascontiguousarray(depth, np.uint16)
        kps = np.zeros(self.cap, KEYPOINT_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        pos = np.zeros((self.cap, 3), np.float32)
        n = C.c_int(0)
        self._chk(self.lib.ssm_orb_extract(self.h, _ptr(img), w, h, img.strides[0], ch, _ptr(depth), _ptr(kps), _ptr(desc),
                                           _ptr(pos), self.cap, C.byref(n)))
        return kps[:n.value], desc[:n.value], pos[:n.value]

    # ---- asynchronous per-frame calls: results are in the returned holders after wait()
    def detect_features_async(self, img, depth=None):
        """ssm_orb_extract_async: returns a holder h; after ctx.wait(), h() gives (keypoints, descriptors, positions)"""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape[:2]; ch = 1 if img.ndim == 2 else img.shape[2]
        depth = None if depth is None else np.ascontiguousarray(depth, np.uint16)
        kps = np.zeros(self.cap, KEYPOINT_DTYPE); desc = np.zeros((self.cap, 32), np.uint8); pos = np.zeros((self.cap, 3), np.float32)
        n = C.c_int(-1)
        self._inflight.append((kps, desc, pos, n))      # the C finisher writes into these at the next wait (explicit, or implied by a full ring / ssm_sync): they must outlive the holder
        self._chk(self.lib.ssm_orb_extract_async(self.h, _ptr(img), w, h, img.strides[0], ch, _ptr(depth), _ptr(kps), _ptr(desc), _ptr(pos), self.cap, C.byref(n)))
        return lambda: (kps[:n.value], desc[:n.value], pos[:n.value])

    def match_async(self, q, t, ratio=None):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        out = np.zeros(max(len(q), 1), DMATCH_DTYPE); n = C.c_int(-1)
        r = self.cfg.knn_match_ratio if ratio is None else ratio
        self._inflight.append((out, n))
        self._chk(self.lib.ssm_match_async(self.h, _ptr(q), len(q), _ptr(t), len(t), r, _ptr(out), len(out), C.byref(n)))
        return lambda: out[:n.value]

    def match_refs(self, refs, cur, ratio=None):
        """ssm_match_refs: [match(r, cur) for r in refs] in one launch"""
        refs = [np.ascontiguousarray(r, np.uint8).reshape(-1, 32) for r in refs]; cur = np.ascontiguousarray(cur, np.uint8).reshape(-1, 32)
        k = len(refs)
        outs = [np.zeros(max(len(r), 1), DMATCH_DTYPE) for r in refs]
        pr = (C.c_void_p * max(k, 1))(*[r.ctypes.data for r in refs]); po = (C.c_void_p * max(k, 1))(*[o.ctypes.data for o in outs])
        nr = (C.c_int * max(k, 1))(*[len(r) for r in refs]); caps = (C.c_int * max(k, 1))(*[len(o) for o in outs]); n = (C.c_int * max(k, 1))()
        self._chk(self.lib.ssm_match_refs(self.h, pr, nr, k, _ptr(cur), len(cur), self.cfg.knn_match_ratio if ratio is None else ratio, po, caps, n))
        return [o[:n[i]] for i, o in enumerate(outs)]

    def wait(self):
        try:
            self._chk(self.lib.ssm_wait(self.h))
        finally:
            self._inflight.clear()

    def knn2(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx = np.zeros((len(q), 2), np.int32)
        dist = np.zeros((len(q), 2), np.int32)
        self._chk(self.lib.ssm_hamming_knn2(self.h, _ptr(q), len(q), _ptr(t), len(t), _ptr(idx), _ptr(dist)))
        return idx, dist

    def match(self, q, t, ratio=None):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        out = np.zeros(max(len(q), 1), DMATCH_DTYPE)
        n = C.c_int(0)
        r = self.cfg.knn_match_ratio if ratio is None else ratio
        self._chk(self.lib.ssm_match(self.h, _ptr(q), len(q), _ptr(t), len(t), r, _ptr(out), len(out), C.byref(n)))
        return out[:n.value]

    # ---- QuadFeatureMatch (stereo)
    def quad_track(self, lc, rc, lp, rp, max_corners=1000):
        ims = [np.ascontiguousarray(a, np.uint8) for a in (lc, rc, lp, rp)]
        h, w = ims[0].shape
        out = np.zeros(max_corners, PMATCH_DTYPE); n = C.c_int(0)
        self._chk(self.lib.ssm_quad_track(self.h, _ptr(ims[0]), _ptr(ims[1]), _ptr(ims[2]), _ptr(ims[3]), w, h, w, max_corners, _ptr(out), len(out), C.byref(n)))
        return out[:n.value]

    # ---- device-resident batched stereo path (configs[3])
    def stereo_batch(self):
        return self.lib.ssm_stereo_batch(self.h)

    def stereo_seq_process(self, left_dev, right_dev, n, w, h, continue_sequence=False, stages=0, max_corners=1000, sgbm=None,
                           baseline=0.0, cu=0.0, cv=0.0, f=1.0, roix=0.0, roiy=0.0, roiz=0.0, scale=1.0,
                           vo=None, ransac_iters=200, rand_stream_dev=None):
        """ssm_stereo_seq_process on n device frame pairs: quad matcher (frame f against f - 1), SGBM depth, stereo VO.  vo = (f, cu, cv, base,
        inlier_threshold, reweighting); rand_stream_dev = device pointer to n * ransac_iters * 3 raw rand() draws (GlibcRand.draws)."""
        fr = StereoFramesDev()
        fr.left, fr.right, fr.n, fr.w, fr.h = left_dev, right_dev, n, w, h
        fr.continue_sequence, fr.stages, fr.max_corners = int(continue_sequence), stages, max_corners
        sp = self.sgbm_params() if sgbm is None else np.ascontiguousarray(sgbm, np.int32)
        fr.sgbm = SgbmParams(*[int(v) for v in sp])
        fr.baseline, fr.cu, fr.cv, fr.f, fr.roix, fr.roiy, fr.roiz, fr.scale = baseline, cu, cv, f, roix, roiy, roiz, scale
        if vo is not None:
            fr.vo = VoParams(vo[0], vo[1], vo[2], vo[3], vo[4], int(vo[5]), 0)
        fr.ransac_iters = ransac_iters
        fr.rand_stream = rand_stream_dev
        out = StereoOutDev()
        self._chk(self.lib.ssm_stereo_seq_process(self.h, C.byref(fr), C.byref(out)))
        return out

    def stereo_seq_fetch(self, out, n, w, h, stages=7):
        """Copy the outputs of stereo_seq_process back to the host (test helper)."""
        mc = out.max_corners
        res = {}
        if stages & STEREO_QUAD:
            res["nquad"] = self.d2h(out.nquad, n, np.int32); res["quad"] = self.d2h(out.quad, (n, mc), PMATCH_DTYPE)
            res["ncorners"] = self.d2h(out.ncorners, n, np.int32); res["corners"] = self.d2h(out.corners, (n, mc, 2), np.float32)
        if stages & STEREO_DEPTH:
            res["disp"] = self.d2h(out.disp, (n, h, w), np.int16); res["depth"] = self.d2h(out.depth, (n, h, w), np.uint16)
        if stages & STEREO_VO:
            res["tr"] = self.d2h(out.tr, (n, 6), np.float64); res["inliers"] = self.d2h(out.inliers, (n, mc), np.int32)
            res["vo_result"] = self.d2h(out.vo_result, (n, 2), np.int32); res["rand_draws_used"] = int(self.d2h(out.rand_draws_used, 1, np.int32)[0])
        return res

    @staticmethod
    def sgbm_params(**kw):
        """cv::StereoSGBM fields as src/stereo.cpp:11-30 sets them; override by keyword"""
        d = dict(minDisparity=0, numberOfDisparities=80, SADWindowSize=11, P1=None, P2=None, disp12MaxDiff=1, preFilterCap=63, uniquenessRatio=10,
                 speckleWindowSize=100, speckleRange=32)
        d.update(kw)
        if d["P1"] is None: d["P1"] = 4 * d["SADWindowSize"] ** 2
        if d["P2"] is None: d["P2"] = 32 * d["SADWindowSize"] ** 2
        return np.array([d[k] for k in ("minDisparity", "numberOfDisparities", "SADWindowSize", "P1", "P2", "disp12MaxDiff", "preFilterCap", "uniquenessRatio",
                                        "speckleWindowSize", "speckleRange")], np.int32)

    def sgbm(self, left, right, params=None, raw=False):
        """cv::StereoSGBM::operator(): int16 disparity x16 ((minDisparity-1)*16 = invalid); raw=True stops before medianBlur / filterSpeckles"""
        left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8); h, w = left.shape
        params = self.sgbm_params() if params is None else np.ascontiguousarray(params, np.int32)
        disp = np.zeros((h, w), np.int16)
        self._chk(self.lib.ssm_sgbm(self.h, _ptr(left), _ptr(right), w, h, left.strides[0], _ptr(params), int(raw), _ptr(disp)))
        return disp

    def stereo_depth(self, left, right, baseline, cu, cv, f, roix, roiy, roiz, scale, params=None):
        """FrameReader's KITTI depth step (src/rgbdframe.cpp:81-116): (depth u16, disparity int16)"""
        left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8); h, w = left.shape
        params = self.sgbm_params() if params is None else np.ascontiguousarray(params, np.int32)
        depth = np.zeros((h, w), np.uint16); disp = np.zeros((h, w), np.int16)
        self._chk(self.lib.ssm_stereo_depth(self.h, _ptr(left), _ptr(right), w, h, left.strides[0], _ptr(params), baseline, cu, cv, f, roix, roiy, roiz, scale,
                                            _ptr(depth), _ptr(disp)))
        return depth, disp

    def vo_estimate(self, matches, f, cu, cv, base, samples, inlier_threshold=2.0, reweighting=True):
        """VisualOdometryStereo::estimateMotion on quad matches: (success, tr[6], inlier indices)"""
        m = np.ascontiguousarray(matches, PMATCH_DTYPE); samples = np.ascontiguousarray(samples, np.int32).reshape(-1, 3)
        prm = np.array([(f, cu, cv, base, inlier_threshold, int(reweighting), 0)], VO_PARAMS_DTYPE)
        tr = np.zeros(6, np.float64); inl = np.zeros(max(len(m), 1), np.int32); n = C.c_int(0); ok = C.c_int(0)
        self._chk(self.lib.ssm_vo_estimate(self.h, _ptr(m), len(m), _ptr(prm), _ptr(samples), len(samples), _ptr(tr), _ptr(inl), len(inl), C.byref(n), C.byref(ok)))
        return bool(ok.value), tr, inl[:n.value].copy()

    def pnp_solve(self, img, obj, cam, T0, min_inliers=10):
        """PnPSolver::solvePnP (src/pnp.cpp:5-118) on the device: img n x 2 pixels in frame 2, obj n x 3 points in frame 1's camera frame, cam = (fx, fy, cx, cy),
        T0 = 4 x 4 initial transform.  Returns (success, T 4 x 4, inlier flags uint8[n] as pnp.cpp keeps them, number of set flags)."""
        img = np.ascontiguousarray(img, np.float32).reshape(-1, 2); obj = np.ascontiguousarray(obj, np.float32).reshape(-1, 3)
        assert len(img) == len(obj)
        camv = np.ascontiguousarray(cam, np.float64).reshape(4)
        T = np.ascontiguousarray(np.asarray(T0, np.float64).reshape(4, 4).T).copy()          # column-major
        inl = np.zeros(max(len(img), 1), np.uint8); n = C.c_int(0); ok = C.c_int(0)
        self._chk(self.lib.ssm_pnp_solve(self.h, _ptr(img), _ptr(obj), len(img), _ptr(camv), int(min_inliers), _ptr(T), _ptr(inl), C.byref(n), C.byref(ok)))
        return bool(ok.value), T.T.copy(), inl[:len(img)].copy(), n.value

    def gftt(self, img, max_corners=1000, quality=0.04, min_distance=8.0):
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
        pts = np.zeros((max_corners, 2), np.float32); n = C.c_int(0)
        self._chk(self.lib.ssm_gftt(self.h, _ptr(img), w, h, img.strides[0], max_corners, quality, min_distance, _ptr(pts), max_corners, C.byref(n)))
        return pts[:n.value]

    def lk_track(self, prev, nxt, pts, max_count=200, epsilon=0.01, min_eig=1e-6):
        prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8); h, w = prev.shape
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        out = np.zeros_like(pts); st = np.zeros(len(pts), np.uint8); err = np.zeros(len(pts), np.float32)
        self._chk(self.lib.ssm_lk_track(self.h, _ptr(prev), _ptr(nxt), w, h, w, _ptr(pts), len(pts), _ptr(out), _ptr(st), _ptr(err), max_count, epsilon, min_eig))
        return out, st, err

    def window_match(self, kp1, d1, kp2, d2, sw, sh, thr):
        kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2); kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
        d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
        out = np.zeros(max(len(kp1), 1), DMATCH_DTYPE)
        self._chk(self.lib.ssm_window_match(self.h, _ptr(kp1), _ptr(d1), len(kp1), _ptr(kp2), _ptr(d2), len(kp2), sw, sh, thr, _ptr(out)))
        return out[:len(kp1)]

    # ---- Classifier (SegNet)
    def segnet_layers(self):
        out = []
        for l in range(self.lib.ssm_segnet_num_layers()):
            a, b, h, w = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            self.lib.ssm_segnet_layer_shape(l, C.byref(a), C.byref(b), C.byref(h), C.byref(w))
            out.append((a.value, b.value, h.value, w.value))
        return out

    def segnet_set_layer(self, layer, weight, scale, shift):
        w = np.ascontiguousarray(weight, np.float32); sc = np.ascontiguousarray(scale, np.float32); sh = np.ascontiguousarray(shift, np.float32)
        self._chk(self.lib.ssm_segnet_set_layer(self.h, layer, _ptr(w), _ptr(sc), _ptr(sh)))

    def classify(self, bgr, want_sem=True):
        """Classifier::Classify: returns (labels 360x480 u8, colour-label image at frame size or None)"""
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w = bgr.shape[:2]
        labels = np.zeros((SEG_NET_H, SEG_NET_W), np.uint8)
        sem = np.zeros((h, w, 3), np.uint8) if want_sem else None
        self._chk(self.lib.ssm_segnet_forward(self.h, _ptr(bgr), w, h, bgr.strides[0], _ptr(labels), _ptr(sem)))
        return labels, sem

    def segnet_logits(self):
        out = np.zeros((SEG_NET_H, SEG_NET_W, SEG_CLASSES), np.float32)
        self._chk(self.lib.ssm_segnet_logits(self.h, _ptr(out)))
        return out

    def segnet_debug_conv(self, layer, x_hwc_f16):
        cin, cout, _, _ = self.segnet_layers()[layer]
        x = np.ascontiguousarray(x_hwc_f16, np.float16); h, w, c = x.shape
        assert c == (cin + 15) // 16 * 16
        out = np.zeros((h, w, (cout + 15) // 16 * 16), np.float16)
        self._chk(self.lib.ssm_segnet_debug_op(self.h, 0, layer, _ptr(x), h, w, _ptr(out), None))
        return out[:, :, :cout]

    def segnet_debug_conv_pool(self, layer, x_hwc_f16):
        """conv + BN + ReLU + 2x2 max-pool of `layer` through the fused kernel the network uses: (pooled, codes)."""
        cin, cout, _, _ = self.segnet_layers()[layer]
        x = np.ascontiguousarray(x_hwc_f16, np.float16); h, w, c = x.shape
        ci16, co16 = (cin + 15) & ~15, (cout + 15) & ~15
        xin = np.zeros((h, w, ci16), np.float16); xin[:, :, :c] = x
        out = np.zeros(((h + 1) // 2, (w + 1) // 2, co16), np.float16); code = np.zeros(out.shape, np.uint8)
        self._chk(self.lib.ssm_segnet_debug_op(self.h, 3, layer, _ptr(xin), h, w, _ptr(out), _ptr(code)))
        return out[:, :, :cout], code[:, :, :cout]

    def segnet_debug_unpool_conv(self, layer, pooled_hwc_f16, code, h, w):
        """un-pool (to h x w) + conv + BN + ReLU of `layer` through the fused kernel the network uses."""
        cin, cout, _, _ = self.segnet_layers()[layer]
        x = np.ascontiguousarray(pooled_hwc_f16, np.float16); ph, pw, c = x.shape
        ci16, co16 = (cin + 15) & ~15, (cout + 15) & ~15
        xin = np.zeros((ph, pw, ci16), np.float16); xin[:, :, :c] = x
        cin_code = np.zeros((ph, pw, ci16), np.uint8); cin_code[:, :, :c] = code
        out = np.zeros((h, w, co16), np.float16)
        self._chk(self.lib.ssm_segnet_debug_op(self.h, 4, layer, _ptr(xin), h, w, _ptr(out), _ptr(cin_code)))
        return out[:, :, :cout]

    def segnet_debug_pool(self, x_hwc_f16):
        x = np.ascontiguousarray(x_hwc_f16, np.float16); h, w, c = x.shape
        out = np.zeros(((h + 1) // 2, (w + 1) // 2, c), np.float16); code = np.zeros(out.shape, np.uint8)
        self._chk(self.lib.ssm_segnet_debug_op(self.h, 1, c, _ptr(x), h, w, _ptr(out), _ptr(code)))
        return out, code

    def segnet_debug_unpool(self, x_hwc_f16, code, h, w):
        x = np.ascontiguousarray(x_hwc_f16, np.float16); code = np.ascontiguousarray(code, np.uint8); c = x.shape[2]
        out = np.zeros((h, w, c), np.float16)
        self._chk(self.lib.ssm_segnet_debug_op(self.h, 2, c, _ptr(x), h, w, _ptr(out), _ptr(code)))
        return out

    def segnet_forward_dev(self, bgr_dev, n, labels_dev=None, sem_dev=None, flags=0):
        self._chk(self.lib.ssm_segnet_forward_dev(self.h, bgr_dev, n, labels_dev, sem_dev, flags))

    # ---- Mapper
    def moving_mask(self, sem):
        sem = np.ascontiguousarray(sem, np.uint8)
        h, w = sem.shape[:2]
        mask = np.zeros((h, w), np.uint8)
        self._chk(self.lib.ssm_moving_mask(self.h, _ptr(sem), w, h, sem.strides[0], _ptr(mask)))
        return mask

    def generate_point_cloud(self, depth, rgb, sem, T=None, camera=None, max_distance=None):
        depth = np.ascontiguousarray(depth, np.uint16)
        rgb = np.ascontiguousarray(rgb, np.uint8)
        sem = np.ascontiguousarray(sem, np.uint8)
        h, w = depth.shape
        cam = Camera(*camera) if camera is not None else self.cfg.camera
        md = self.cfg.mapper_max_distance if max_distance is None else max_distance
        Tc = None if T is None else np.ascontiguousarray(np.asarray(T, np.float64).reshape(4, 4).T)  # column-major
        out = np.zeros(w * h, POINT_DTYPE)
        n = C.c_int(0)
        self._chk(self.lib.ssm_backproject(self.h, _ptr(depth), _ptr(rgb), _ptr(sem), w, h, C.byref(cam), _ptr(Tc), md,
                                           _ptr(out), len(out), C.byref(n)))
        return out[:n.value]

    # ---- device-resident Mapper: key-frame clouds and the viewer's map stay in HBM (ssm_backproject_dev / ssm_viewer_map_*)
    def backproject_dev(self, depth, rgb, sem, camera=None, max_distance=None):
        """camera-frame cloud of one frame, left on the device: returns an opaque handle (cloud_free it)"""
        depth = np.ascontiguousarray(depth, np.uint16); rgb = np.ascontiguousarray(rgb, np.uint8); sem = np.ascontiguousarray(sem, np.uint8)
        h, w = depth.shape
        cam = Camera(*camera) if camera is not None else self.cfg.camera
        md = self.cfg.mapper_max_distance if max_distance is None else max_distance
        cl = C.c_void_p()
        self._chk(self.lib.ssm_backproject_dev(self.h, _ptr(depth), _ptr(rgb), _ptr(sem), w, h, C.byref(cam), md, C.byref(cl)))
        return cl.value

    def cloud_size(self, cloud):
        return self.lib.ssm_cloud_size(cloud)

    def cloud_fetch(self, cloud, T=None):
        n = self.cloud_size(cloud)
        out = np.zeros(max(n, 1), POINT_DTYPE); m = C.c_int(0)
        Tc = None if T is None else np.ascontiguousarray(np.asarray(T, np.float64).reshape(4, 4).T)
        self._chk(self.lib.ssm_cloud_fetch(self.h, cloud, _ptr(Tc), _ptr(out), len(out), C.byref(m)))
        return out[:m.value]

    def cloud_free(self, cloud):
        self.lib.ssm_cloud_free(self.h, cloud)

    def viewer_map_update(self, clouds, poses, rebuild=False, leaf=None):
        """Mapper::viewer's update on the device: map <- VoxelGrid((rebuild ? nothing : previous map) + sum poses[i] * clouds[i]); returns the voxel count"""
        leaf = self.cfg.mapper_resolution if leaf is None else leaf
        arr = (C.c_void_p * max(len(clouds), 1))(*clouds)
        P = np.ascontiguousarray(np.stack([np.asarray(T, np.float64).reshape(4, 4).T.reshape(16) for T in poses]) if len(poses) else np.zeros((1, 16)))
        n = C.c_int(0)
        self._chk(self.lib.ssm_viewer_map_update(self.h, int(rebuild), arr, _ptr(P), len(clouds), leaf, C.byref(n)))
        return n.value

    def viewer_map_fetch(self, n):
        out = np.zeros(max(n, 1), POINT_DTYPE); m = C.c_int(0)
        self._chk(self.lib.ssm_viewer_map_fetch(self.h, _ptr(out), len(out), C.byref(m)))
        return out[:m.value]

    def voxel_filter(self, pts, leaf=None, cap=None):
        pts = np.ascontiguousarray(pts, POINT_DTYPE)
        leaf = self.cfg.mapper_resolution if leaf is None else leaf
        out = np.zeros(cap if cap is not None else max(len(pts), 1), POINT_DTYPE)
        n = C.c_int(0)
        self._chk(self.lib.ssm_voxel_filter(self.h, _ptr(pts), len(pts), leaf, _ptr(out), len(out), C.byref(n)))
        return out[:n.value]

    def map_clear(self):
        self._chk(self.lib.ssm_map_clear(self.h))

    def map_insert(self, pts):
        pts = np.ascontiguousarray(pts, POINT_DTYPE)
        self._chk(self.lib.ssm_map_insert(self.h, _ptr(pts), len(pts)))

    def map_stats(self):
        """(log2 slots, times grown, blocks of the map kernel run again, overflow-list records): ssm_map_stats"""
        st = (C.c_int64 * 4)()
        self._chk(self.lib.ssm_map_stats(self.h, st))
        return tuple(int(v) for v in st)

    def map_size(self):
        n = C.c_int(0)
        self._chk(self.lib.ssm_map_size(self.h, C.byref(n)))
        return n.value

    def map_export(self):
        out = np.zeros(max(self.map_size(), 1), POINT_DTYPE)
        n = C.c_int(0)
        self._chk(self.lib.ssm_map_export(self.h, _ptr(out), len(out), C.byref(n)))
        return out[:n.value]

    def map_export_table(self):
        out = np.zeros(max(self.map_size(), 1), VOXEL_DTYPE)
        n = C.c_int(0)
        self._chk(self.lib.ssm_map_export_table(self.h, _ptr(out), len(out), C.byref(n)))
        return out[:n.value]

    def map_merge_table(self, tab):
        tab = np.ascontiguousarray(tab, VOXEL_DTYPE)
        self._chk(self.lib.ssm_map_merge_table(self.h, _ptr(tab), len(tab)))

    def map_export_table_dev(self, dptr, cap):
        n = C.c_int(0)
        self._chk(self.lib.ssm_map_export_table_dev(self.h, dptr, cap, C.byref(n)))
        return n.value

    def map_merge_table_dev(self, dptr, n):
        self._chk(self.lib.ssm_map_merge_table_dev(self.h, dptr, n))

    # ---- multi-GPU: the communicator lives in the context (RCCL inside libssm_hip.so, no torch on the data path)
    def comm_unique_id(self):
        """rank 0: ncclGetUniqueId as SSM_COMM_ID_BYTES bytes, to be shipped to every rank"""
        buf = (C.c_ubyte * COMM_ID_BYTES)()
        self._chk(self.lib.ssm_comm_get_unique_id(buf))
        return bytes(buf)

    def comm_init_rank(self, nranks, rank, unique_id):
        assert len(unique_id) == COMM_ID_BYTES
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._chk(self.lib.ssm_comm_init_rank(self.h, nranks, rank, buf))

    def comm_finalize(self):
        self._chk(self.lib.ssm_comm_finalize(self.h))

    def voxel_allgather(self, rccl_comm=None):
        """merge the context maps of all ranks: one RCCL all-gather of the voxel tables + re-insertion of the remote ones"""
        self._chk(self.lib.ssm_voxel_allgather(self.h, rccl_comm))

    # ---- device-resident sequence path
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.ssm_dev_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, p):
        self._chk(self.lib.ssm_dev_free(self.h, p))

    def host_alloc(self, shape, dtype):
        """a numpy array in page-locked host memory (ssm_host_alloc): such inputs are read by the device where they are.  Free it with host_free(array)."""
        shape = tuple(int(v) for v in np.atleast_1d(shape)); dt = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        self._chk(self.lib.ssm_host_alloc(max(nbytes, 1), C.byref(p)))
        arr = np.frombuffer((C.c_char * nbytes).from_address(p.value), dtype=dt).reshape(shape)
        self._pinned = getattr(self, "_pinned", {}); self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        p = self._pinned.pop(arr.ctypes.data)
        self._chk(self.lib.ssm_host_free(C.c_void_p(p)))

    def mem_info(self):
        """(free, total) bytes of the context's device"""
        f, t = C.c_size_t(0), C.c_size_t(0)
        self._chk(self.lib.ssm_dev_mem_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.lib.ssm_memcpy_h2d(self.h, dptr, _ptr(arr), arr.nbytes))

    def d2h(self, dptr, shape, dtype):
        out = np.zeros(shape, dtype)
        if out.nbytes:
            self._chk(self.lib.ssm_memcpy_d2h(self.h, _ptr(out), dptr, out.nbytes))
        return out

    def synth_frames_dev(self, seed, first, n, bgr, depth, sem, pose, labels=None):
        self._chk(self.lib.ssm_synth_frames_dev(self.h, seed, first, n, self.W, self.H, bgr, depth, sem, labels, pose))

    def seq_process(self, bgr, depth, sem, pose, n, continue_sequence=False, stages=0):
        fr = FramesDev(bgr, depth, sem, pose, n, int(continue_sequence), stages)
        out = SeqOutDev()
        self._chk(self.lib.ssm_seq_process(self.h, C.byref(fr), C.byref(out)))
        return out

    def sync(self):
        try:
            self._chk(self.lib.ssm_sync(self.h))
        finally:
            self._inflight.clear()

    def last_error(self):
        """ssm_last_error: the last failing call's text, or a "note: ..." a successful call left (an SGBM sweep repeated in form 1)"""
        return (self.lib.ssm_last_error(self.h) or b"").decode()

    def set_profiling(self, on):
        self._chk(self.lib.ssm_set_profiling(self.h, int(on)))

    def stage_times(self):
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        ln = (C.c_int * 32)()
        n = C.c_int(0)
        self._chk(self.lib.ssm_get_stage_times(self.h, names, ms, ln, 32, C.byref(n)))
        return {names[i].decode(): (ms[i], ln[i]) for i in range(min(n.value, 32))}

    def seq_fetch(self, out, n):
        """Copy the outputs of seq_process back to the host (test helper)."""
        cap, R = out.cap, out.R
        res = {
            "nkp": self.d2h(out.nkp, n, np.int32),
            "kps": self.d2h(out.kps, (n, cap), KEYPOINT_DTYPE),
            "desc": self.d2h(out.desc, (n, cap, 32), np.uint8),
            "pos3d": self.d2h(out.pos3d, (n, cap, 3), np.float32),
            "nmatch": self.d2h(out.nmatch, (n, R), np.int32),
            "matches": self.d2h(out.matches, (n, R, cap), DMATCH_DTYPE),
            "npoints": self.d2h(out.npoints, n, np.int32),
        }
        return res
