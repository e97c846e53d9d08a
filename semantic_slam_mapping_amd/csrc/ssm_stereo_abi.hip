// ssm_stereo_abi.hip -- the stereo path behind the C ABI: QuadFeatureMatch (GFTT + LK), cv::StereoSGBM + the depth conversion, VisualOdometryStereo, and
// PnPSolver::solvePnP for one correspondence list.  Kernels: kernels_quad.hip, kernels_sgbm.hip, kernels_vo.hip, kernels_pnp.hip.
#include "ssm_ctx.h"

// ---------------------------------------------------------------- stereo path: QuadFeatureMatch, StereoSGBM depth, VisualOdometryStereo
void stereo_free(StereoState* q)
{
    void* p[] = { q->pyr, q->der, q->gw.eig, q->gw.cand_at, q->gw.cand_bits, q->gw.keys, q->gw.kept, q->gw.deps, q->gw.depn, q->gw.state, q->gw.maxord, q->gw.count, q->gw.nkept, q->overflow, q->sg_fail, q->ncorner, q->has_prev, q->pts, q->status, q->err,
                  q->tr_all, q->vcount, q->rand_off, q->consumed, q->sg_wsN[0], q->dminN[0], q->sg_wsN[1], q->dminN[1], q->sg_wsN[2], q->dminN[2], q->quad, q->nquad, q->corners, q->ncorners, q->disp, q->depth, q->tr,
                  q->inliers, q->vo_result, q->in_stage };
    for (void* x : p) if (x) hipFree(x);
}
// exact: the row stride of the sequence outputs is max_corners, so the sequence path wants exactly that many; the per-call entry points take any workspace that is large enough
static int stereo_init(ssm_ctx* c, int w, int h, int maxc, bool exact = false)
{
    if (c->stereo && c->stereo->w == w && c->stereo->h == h && (exact ? c->stereo->maxc == maxc : c->stereo->maxc >= maxc)) return SSM_OK;
    if (w < 4 || h < 2 || w > 4096 || h > 4096) FAIL(c, SSM_E_INVAL, "stereo path: image size must be at most 4096 x 4096");
    if (maxc < 1 || maxc > 32767) FAIL(c, SSM_E_INVAL, "max_corners must be 1..32767");
    if (c->stereo) { hipDeviceSynchronize(); stereo_free(c->stereo); delete c->stereo; c->stereo = nullptr; }
    StereoState* q = new StereoState(); c->stereo = q;
    q->w = w; q->h = h; q->maxc = maxc; q->B = c->stereo_B;
    const int B = q->B;
    QuadBatch& b = q->qb;
    int off = 0;
    for (int l = 0; l < 4; l++) { b.w[l] = l ? (b.w[l-1] + 1) / 2 : w; b.h[l] = l ? (b.h[l-1] + 1) / 2 : h; b.off[l] = off; off += b.w[l] * b.h[l]; off = (off + 15) & ~15; }
    b.slot_elems = (size_t)off; b.B1 = B + 1;
    DALLOC(c, q->pyr, (size_t)2 * b.B1 * b.slot_elems); DALLOC(c, q->der, (size_t)2 * b.B1 * b.slot_elems * 2);
    b.pyr = q->pyr; b.der = q->der;
    const size_t np = (size_t)w * h;
    q->keycap = w * h / 4 + 1024;                            // 3x3 local maxima: at most one per 2x2 pixels
    GfttWork& g = q->gw; g.cap = q->keycap;
    DALLOC(c, g.eig, (size_t)B * np); DALLOC(c, g.cand_at, (size_t)B * np); DALLOC(c, g.keys, (size_t)B * q->keycap); DALLOC(c, g.kept, (size_t)B * q->keycap);
    DALLOC(c, g.deps, (size_t)B * q->keycap * k_quad_gftt_deps_per_candidate()); DALLOC(c, g.depn, (size_t)B * q->keycap); DALLOC(c, g.state, (size_t)B * q->keycap);
    HIPCHK(c, hipMemset(g.cand_at, 0, (size_t)B * np * 4));      // gftt_finish_kernel keeps the map zeroed between calls
    DALLOC(c, g.cand_bits, (size_t)B * k_quad_gftt_bits_words(w, h));
    DALLOC(c, g.maxord, B); DALLOC(c, g.count, B); DALLOC(c, g.nkept, B); DALLOC(c, q->overflow, 1); DALLOC(c, q->sg_fail, SG_FAIL_WORDS); DALLOC(c, q->ncorner, B); DALLOC(c, q->has_prev, B);
    g.overflow = q->overflow;
    HIPCHK(c, hipMemset(q->overflow, 0, 4));
    HIPCHK(c, hipMemset(q->sg_fail, 0, 4 * SG_FAIL_WORDS));
    DALLOC(c, q->pts, (size_t)5 * B * maxc * 2); DALLOC(c, q->status, maxc); DALLOC(c, q->err, maxc);
    DALLOC(c, q->rand_off, B); DALLOC(c, q->consumed, 1);
    return SSM_OK;
}
static int stereo_ensure_seq(ssm_ctx* c, int n)
{
    StereoState* q = c->stereo;
    if (n <= q->seq_cap) return SSM_OK;
    HIPCHK(c, hipDeviceSynchronize());
    void* olds[] = { q->quad, q->nquad, q->corners, q->ncorners, q->disp, q->depth, q->tr, q->inliers, q->vo_result };
    for (void* p : olds) if (p) hipFree(p);
    q->quad = nullptr; q->nquad = nullptr; q->corners = nullptr; q->ncorners = nullptr; q->disp = nullptr; q->depth = nullptr; q->tr = nullptr; q->inliers = nullptr; q->vo_result = nullptr;
    q->seq_cap = 0;
    const size_t np = (size_t)q->w * q->h;
    DALLOC(c, q->quad, (size_t)n * q->maxc); DALLOC(c, q->nquad, n); DALLOC(c, q->corners, (size_t)n * q->maxc * 2); DALLOC(c, q->ncorners, n);
    DALLOC(c, q->disp, (size_t)n * np); DALLOC(c, q->depth, (size_t)n * np);
    DALLOC(c, q->tr, (size_t)n * 6); DALLOC(c, q->inliers, (size_t)n * q->maxc); DALLOC(c, q->vo_result, (size_t)n * 2);
    q->seq_cap = n;
    return SSM_OK;
}
static int stereo_ensure_vo(ssm_ctx* c, int iters)
{
    StereoState* q = c->stereo;
    if (iters <= q->vo_iters) return SSM_OK;
    HIPCHK(c, hipDeviceSynchronize());
    if (q->tr_all) hipFree(q->tr_all); if (q->vcount) hipFree(q->vcount);
    q->tr_all = nullptr; q->vcount = nullptr; q->vo_iters = 0;
    DALLOC(c, q->tr_all, (size_t)q->B * iters * 6); DALLOC(c, q->vcount, (size_t)q->B * iters);
    q->vo_iters = iters;
    return SSM_OK;
}
static int stereo_ensure_sgbm(ssm_ctx* c, const ssm_sgbm_params& p, int nb, int which = 0)
{
    StereoState* q = c->stereo;
    const size_t need = k_sgbm_workspace_bytes(q->w, q->h, p, nb, c->sgbm_form_cfg);
    void*& ws = q->sg_wsN[which]; size_t& have = q->sg_ws_bytesN[which];
    if (!q->dminN[which]) DALLOC(c, q->dminN[which], 128);
    if (need <= have) return SSM_OK;
    HIPCHK(c, hipDeviceSynchronize());
    if (ws) hipFree(ws);
    ws = nullptr; have = 0;
    uint8_t* p8; int r = dalloc(c, &p8, need); if (r) return r;
    ws = p8; have = need;
    return SSM_OK;
}
static int sgbm_check_params(ssm_ctx* c, const ssm_sgbm_params* params, int w, int h)
{
    if (!params) FAIL(c, SSM_E_INVAL, "null SGBM parameters");
    const int D = params->numberOfDisparities, SW = params->SADWindowSize > 0 ? params->SADWindowSize : 5;
    if (D <= 0 || D % 16 || D > 128 || D / 16 == 7) FAIL(c, SSM_E_INVAL, "numberOfDisparities must be 16, 32, 48, 64, 80, 96 or 128");
    if (!(SW & 1) || h <= SW || w <= SW) FAIL(c, SSM_E_INVAL, "SADWindowSize must be odd and smaller than the image");
    if ((long long)w * h >= (1ll << 30)) FAIL(c, SSM_E_INVAL, "image too large");
    { int tx; size_t lds; if (!sgbm_cost_geometry(D, SW, &tx, &lds)) FAIL(c, SSM_E_INVAL, "SADWindowSize too large for this numberOfDisparities (the cost kernel keeps SADWindowSize rows of 4 columns x D sums in LDS)"); }
    return SSM_OK;
}
// the sequence path on device images; the caller holds the context lock
// pair_call: the caller is ssm_quad_track (frame 0 = the previous pair of ONE matcher call: only its pyramids and derivatives are needed -- its corners and tracks
// are nobody's output, and skipping them takes a third off the call: 1.06 -> 0.8 ms at 1241 x 376)
static int stereo_seq_run(ssm_ctx* c, const ssm_stereo_frames_dev* in, ssm_stereo_out_dev* out, bool pair_call = false)
{
    const int n = in->n, w = in->w, h = in->h;
    const int stages = in->stages ? in->stages : (SSM_STEREO_QUAD | SSM_STEREO_DEPTH | SSM_STEREO_VO);
    if (n < 0 || !in->left || !in->right) FAIL(c, SSM_E_INVAL, "bad arguments");
    if ((stages & SSM_STEREO_VO) && !(stages & SSM_STEREO_QUAD)) FAIL(c, SSM_E_INVAL, "SSM_STEREO_VO needs SSM_STEREO_QUAD");
    if ((stages & SSM_STEREO_VO) && (in->ransac_iters < 0 || (in->ransac_iters > 0 && !in->rand_stream))) FAIL(c, SSM_E_INVAL, "the VO stage needs rand_stream (n * ransac_iters * 3 draws)");
    const int maxc = in->max_corners > 0 ? in->max_corners : 1000;
    if ((stages & SSM_STEREO_QUAD) && (w < 32 || h < 32)) FAIL(c, SSM_E_INVAL, "quad matcher: image size must be 32..4096");
    int r = stereo_init(c, w, h, maxc, true); if (r) return r;
    StereoState* q = c->stereo;
    if (stages & SSM_STEREO_DEPTH) { r = sgbm_check_params(c, &in->sgbm, w, h); if (r) return r; }
    r = stereo_ensure_seq(c, n > 0 ? n : 1); if (r) return r;
    if (stages & SSM_STEREO_VO) { r = stereo_ensure_vo(c, in->ransac_iters > 0 ? in->ransac_iters : 1); if (r) return r; }
    const int B = q->B;
    if (stages & SSM_STEREO_DEPTH) { r = stereo_ensure_sgbm(c, in->sgbm, n < B ? (n > 0 ? n : 1) : B); if (r) return r; }
    const size_t np = (size_t)w * h;
    const QuadBatch& qb = q->qb;
    hipStream_t sq = c->stream, sd = c->stream;
    // the quad matcher + VO chain (many small latency-bound kernels) and SGBM (volume kernels) of a sub-batch share nothing but the input images:
    // SGBM runs on the second context stream beside the chain; sub-batches follow each other on both streams without a join in between
    const bool two = (stages & SSM_STEREO_DEPTH) && (stages & SSM_STEREO_QUAD) && !c->serialize;
    if (two) { r = ensure_side_streams(c); if (r) return r; sd = c->stream2; HIPCHK(c, hipEventRecord(c->ev_fork, c->stream)); HIPCHK(c, hipStreamWaitEvent(sd, c->ev_fork, 0)); }
    // ... and with more than one sub-batch SGBM alternates between TWO streams with a workspace each: the cost kernel and the small kernels of one
    // sub-batch (LDS / latency-bound) run beside the scan-direction and winner-takes-all kernels of the other (HBM-bound)
    const int nsub = (n + B - 1) / B;
    const int nsg = two ? (c->stereo_sgbm_streams < nsub ? c->stereo_sgbm_streams : nsub) : 1;
    hipStream_t sgs[3] = {sd, two ? c->stream3 : sd, two ? c->stream4 : sd};
    for (int k = 1; k < nsg; k++) { r = stereo_ensure_sgbm(c, in->sgbm, B, k); if (r) return r; HIPCHK(c, hipStreamWaitEvent(sgs[k], c->ev_fork, 0)); }
    if (c->profiling) { c->recs.clear(); c->pool_used = 0; }
    const bool prev0 = in->continue_sequence && q->have_prev;
    if (stages & SSM_STEREO_VO) HIPCHK(c, hipMemsetAsync(q->consumed, 0, 4, sq));
    for (int f0 = 0; f0 < n; f0 += B) {
        const int nb = n - f0 < B ? n - f0 : B;
        if (stages & SSM_STEREO_QUAD) {
            prof_begin(c, "quad_track");
            // level 0 of the nb frames into slots 1 .. nb of both sides, then the pyramids and derivatives
            HIPCHK(c, hipMemcpy2DAsync(q->pyr + (size_t)(0 * qb.B1 + 1) * qb.slot_elems, qb.slot_elems, in->left + (size_t)f0 * np, np, np, nb, hipMemcpyDeviceToDevice, sq));
            HIPCHK(c, hipMemcpy2DAsync(q->pyr + (size_t)(1 * qb.B1 + 1) * qb.slot_elems, qb.slot_elems, in->right + (size_t)f0 * np, np, np, nb, hipMemcpyDeviceToDevice, sq));
            HIPCHK(c, k_quad_pyramids(qb, nb, sq));
            HIPCHK(c, hipMemsetAsync(q->has_prev, 1, 4 * (size_t)nb, sq));                      // non-zero = true
            if (f0 == 0 && !prev0) HIPCHK(c, hipMemsetAsync(q->has_prev, 0, 4, sq));
            if (!(pair_call && f0 == 0 && nb == 1 && !prev0)) {
            HIPCHK(c, k_quad_gftt(qb, nb, maxc, 0.04, 8.0, q->gw, q->pts, maxc, q->ncorner, sq));      // quadmatcher.cpp:301-308
            HIPCHK(c, k_quad_track(qb, nb, q->pts, maxc, q->ncorner, q->has_prev, q->quad + (size_t)f0 * maxc, q->nquad + f0, sq));
            HIPCHK(c, hipMemcpyAsync(q->corners + (size_t)f0 * maxc * 2, q->pts, (size_t)nb * maxc * 8, hipMemcpyDeviceToDevice, sq));
            HIPCHK(c, hipMemcpyAsync(q->ncorners + f0, q->ncorner, (size_t)nb * 4, hipMemcpyDeviceToDevice, sq));
            }
            // carry: the last frame of the sub-batch becomes slot 0 (images and derivatives, both sides)
            for (int side = 0; side < 2; side++) {
                HIPCHK(c, hipMemcpyAsync(q->pyr + (size_t)(side * qb.B1) * qb.slot_elems, q->pyr + (size_t)(side * qb.B1 + nb) * qb.slot_elems, qb.slot_elems, hipMemcpyDeviceToDevice, sq));
                HIPCHK(c, hipMemcpyAsync(q->der + (size_t)(side * qb.B1) * qb.slot_elems * 2, q->der + (size_t)(side * qb.B1 + nb) * qb.slot_elems * 2, qb.slot_elems * 4, hipMemcpyDeviceToDevice, sq));
            }
            prof_end(c);
        }
        if (stages & SSM_STEREO_VO) {
            prof_begin(c, "vo");
            HIPCHK(c, k_vo_estimate_batch(q->quad + (size_t)f0 * maxc, maxc, q->nquad + f0, nb, in->vo, in->rand_stream, in->ransac_iters, q->consumed, q->rand_off,
                                          q->tr_all, q->vcount, q->tr + (size_t)f0 * 6, q->inliers + (size_t)f0 * maxc, q->vo_result + (size_t)f0 * 2, sq));
            prof_end(c);
        }
        if (stages & SSM_STEREO_DEPTH) {
            const int alt = (f0 / B) % nsg;
            hipStream_t sg = sgs[alt];
            struct StreamSet { ssm_ctx* c; hipStream_t keep; StreamSet(ssm_ctx* c_, hipStream_t s_) : c(c_), keep(c_->stream) { c->stream = s_; } ~StreamSet() { c->stream = keep; } } on(c, sg);   // stage events on SGBM's stream
            prof_begin(c, "sgbm");
            HIPCHK(c, k_sgbm(in->left + (size_t)f0 * np, in->right + (size_t)f0 * np, w, h, nb, in->sgbm, q->sg_wsN[alt], q->sg_ws_bytesN[alt], q->disp + (size_t)f0 * np, 0, sg, q->sg_fail + (f0 / B) % SG_FAIL_WORDS,
                             c->sgbm_form_cfg, nsg));
            HIPCHK(c, k_sgbm_depth(q->disp + (size_t)f0 * np, w, h, nb, in->baseline, in->cu, in->cv, in->f, in->roix, in->roiy, in->roiz, in->scale, q->dminN[alt], q->depth + (size_t)f0 * np, sg));
            prof_end(c);
        }
    }
    if (two) { HIPCHK(c, hipEventRecord(c->ev_join, sd)); HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0)); }
    if (nsg > 1) { HIPCHK(c, hipEventRecord(c->ev_join3, c->stream3)); HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join3, 0)); }
    if (nsg > 2) { HIPCHK(c, hipEventRecord(c->ev_join4, c->stream4)); HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join4, 0)); }
    if (n > 0) q->have_prev = (stages & SSM_STEREO_QUAD) != 0;
    q->sg_pending.valid = (stages & SSM_STEREO_DEPTH) && n > 0;
    if (q->sg_pending.valid) { q->sg_pending.in = *in; q->sg_pending.B = B; }
    if (out) {
        out->quad = q->quad; out->nquad = q->nquad; out->corners = q->corners; out->ncorners = q->ncorners; out->disp = q->disp; out->depth = q->depth;
        out->tr = q->tr; out->inliers = q->inliers; out->vo_result = q->vo_result; out->rand_draws_used = q->consumed; out->max_corners = maxc;
    }
    return SSM_OK;
}
extern "C" int ssm_stereo_batch(const ssm_ctx* c) { return c ? c->stereo_B : 0; }
extern "C" int ssm_stereo_seq_process(ssm_ctx* c, const ssm_stereo_frames_dev* in, ssm_stereo_out_dev* out)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!in) FAIL(c, SSM_E_INVAL, "null argument");
    return stereo_seq_run(c, in, out);
}
// host images -> packed device staging: slot k of the staging area holds image k ([h][w] bytes each); through pinned memory (a pageable copy of a
// 1241x376 image costs ~1 ms)
static int stereo_stage_images(ssm_ctx* c, const uint8_t* const* imgs, int nimg, int w, int h, int stride, uint8_t** dev_out)
{
    StereoState* q = c->stereo;
    const size_t np = (size_t)w * h;
    int r = ensure_pinned(c, np * 6 > (size_t)nimg * np ? np * 6 : (size_t)nimg * np); if (r) return r;
    if ((size_t)nimg * np > q->in_stage_bytes) {
        HIPCHK(c, hipDeviceSynchronize());
        if (q->in_stage) hipFree(q->in_stage);
        q->in_stage = nullptr; q->in_stage_bytes = 0;
        DALLOC(c, q->in_stage, (size_t)4 * np); q->in_stage_bytes = (size_t)4 * np;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));               // the previous call's copies out of the staging buffer are done
    for (int k = 0; k < nimg; k++)
        for (int y = 0; y < h; y++) memcpy(c->h_pinned + (size_t)k * np + (size_t)y * w, imgs[k] + (size_t)y * stride, w);
    HIPCHK(c, hipMemcpyAsync(q->in_stage, c->h_pinned, (size_t)nimg * np, hipMemcpyHostToDevice, c->stream));
    *dev_out = q->in_stage;
    return SSM_OK;
}
extern "C" int ssm_quad_track(ssm_ctx* c, const uint8_t* lc, const uint8_t* rc, const uint8_t* lp, const uint8_t* rp, int w, int h, int stride,
                              int max_corners, ssm_pmatch* out, int cap, int* n_out)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!lc || !rc || !lp || !rp || !n_out || stride < w || max_corners < 1) FAIL(c, SSM_E_INVAL, "bad arguments");
    int r = stereo_init(c, w, h, max_corners, true); if (r) return r;
    // a two-frame sequence: frame 0 = the previous pair, frame 1 = the current pair (left images first, then the right ones)
    const uint8_t* imgs[4] = { lp, lc, rp, rc };
    uint8_t* dev = nullptr;
    r = stereo_stage_images(c, imgs, 4, w, h, stride, &dev); if (r) return r;
    ssm_stereo_frames_dev in; memset(&in, 0, sizeof(in));
    in.left = dev; in.right = dev + (size_t)2 * w * h; in.n = 2; in.w = w; in.h = h; in.stages = SSM_STEREO_QUAD; in.max_corners = max_corners;
    ssm_stereo_out_dev o;
    r = stereo_seq_run(c, &in, &o, true); if (r) return r;
    c->stereo->have_prev = false;                                // a per-pair call is not part of a sequence
    int m = 0;
    HIPCHK(c, hipMemcpyAsync(&m, o.nquad + 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    r = check_device_flags(c, false); if (r) return r;
    *n_out = m;
    if (m > cap) FAIL(c, SSM_E_CAPACITY, "pmatch buffer too small (need " + std::to_string(m) + ")");
    if (m > 0) HIPCHK(c, hipMemcpy(out, o.quad + o.max_corners, (size_t)m * sizeof(ssm_pmatch), hipMemcpyDeviceToHost));
    return SSM_OK;
}
extern "C" int ssm_gftt(ssm_ctx* c, const uint8_t* img, int w, int h, int stride, int max_corners, double quality, double min_distance,
                        float* pts, int cap, int* n_out)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!img || !pts || !n_out || stride < w || max_corners < 1 || !(min_distance >= 1.0)) FAIL(c, SSM_E_INVAL, "bad arguments (max_corners >= 1, min_distance >= 1)");
    if (min_distance > 64.0) FAIL(c, SSM_E_INVAL, "min_distance must be <= 64");
    if (w < 32 || h < 32) FAIL(c, SSM_E_INVAL, "quad matcher: image size must be 32..4096");
    if (max_corners > 32767) FAIL(c, SSM_E_INVAL, "max_corners must be <= 32767");
    int r = stereo_init(c, w, h, max_corners); if (r) return r;
    StereoState* q = c->stereo; const QuadBatch& qb = q->qb;
    HIPCHK(c, hipMemcpy2DAsync(q->pyr + (size_t)1 * qb.slot_elems, w, img, stride, w, h, hipMemcpyHostToDevice, c->stream));        // side 0, slot 1, level 0
    HIPCHK(c, k_quad_gftt(qb, 1, max_corners, quality, min_distance, q->gw, q->pts, q->maxc, q->ncorner, c->stream));
    int n = 0;
    HIPCHK(c, hipMemcpyAsync(&n, q->ncorner, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    r = check_device_flags(c, false); if (r) return r;
    *n_out = n;
    if (n > cap) FAIL(c, SSM_E_CAPACITY, "point buffer too small");
    if (n) HIPCHK(c, hipMemcpy(pts, q->pts, (size_t)n * 8, hipMemcpyDeviceToHost));
    return SSM_OK;
}
extern "C" int ssm_lk_track(ssm_ctx* c, const uint8_t* prev, const uint8_t* next, int w, int h, int stride, const float* prev_pts, int n,
                            float* next_pts, uint8_t* status, float* err, int max_count, double epsilon, double min_eig_threshold)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!prev || !next || n < 0 || (n && (!prev_pts || !next_pts)) || stride < w || max_count < 1) FAIL(c, SSM_E_INVAL, "bad arguments");
    if (n == 0) return SSM_OK;
    if (w < 32 || h < 32) FAIL(c, SSM_E_INVAL, "quad matcher: image size must be 32..4096");
    int r = stereo_init(c, w, h, n > 1000 ? n : 1000); if (r) return r;
    StereoState* q = c->stereo; const QuadBatch& qb = q->qb;
    // previous image = (side 0, slot 1), next image = (side 1, slot 1)
    HIPCHK(c, hipMemcpy2DAsync(q->pyr + (size_t)1 * qb.slot_elems, w, prev, stride, w, h, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(q->pyr + (size_t)(qb.B1 + 1) * qb.slot_elems, w, next, stride, w, h, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, k_quad_pyramids(qb, 1, c->stream));
    float* d_in = q->pts; float* d_out = q->pts + (size_t)2 * q->maxc;
    HIPCHK(c, hipMemcpyAsync(d_in, prev_pts, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, k_quad_lk(qb, d_in, n, d_out, q->status, q->err, max_count, (float)(epsilon * epsilon), (float)min_eig_threshold, c->stream));
    HIPCHK(c, hipMemcpyAsync(next_pts, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    if (status) HIPCHK(c, hipMemcpyAsync(status, q->status, n, hipMemcpyDeviceToHost, c->stream));
    if (err) HIPCHK(c, hipMemcpyAsync(err, q->err, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stereo->have_prev = false;
    return SSM_OK;
}
extern "C" int ssm_window_match(ssm_ctx* c, const float* kp1, const uint8_t* d1, int n1, const float* kp2, const uint8_t* d2, int n2,
                                int search_width, int search_height, float distance_threshold, ssm_dmatch* out)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (n1 < 0 || n2 < 0 || (n1 && (!kp1 || !d1 || !out)) || (n2 && (!kp2 || !d2))) FAIL(c, SSM_E_INVAL, "bad arguments");
    if (n1 == 0) return SSM_OK;
    const size_t a1 = ((size_t)n1 * 8 + 255) & ~(size_t)255, a2 = ((size_t)n2 * 8 + 255) & ~(size_t)255, b1 = ((size_t)n1 * 32 + 255) & ~(size_t)255, b2 = ((size_t)n2 * 32 + 255) & ~(size_t)255;
    int r = ensure_scratch(c, a1 + a2 + b1 + b2 + (size_t)n1 * 16 + 256); if (r) return r;
    uint8_t* p = (uint8_t*)c->d_scratch;
    float* dk1 = (float*)p; p += a1; float* dk2 = (float*)p; p += a2; uint8_t* dd1 = p; p += b1; uint8_t* dd2 = p; p += b2; ssm_dmatch* dm = (ssm_dmatch*)p;
    HIPCHK(c, hipMemcpyAsync(dk1, kp1, (size_t)n1 * 8, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipMemcpyAsync(dd1, d1, (size_t)n1 * 32, hipMemcpyHostToDevice, c->stream));
    if (n2) { HIPCHK(c, hipMemcpyAsync(dk2, kp2, (size_t)n2 * 8, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipMemcpyAsync(dd2, d2, (size_t)n2 * 32, hipMemcpyHostToDevice, c->stream)); }
    HIPCHK(c, k_quad_window_match(dk1, dd1, n1, dk2, dd2, n2, search_width, search_height, distance_threshold, dm, c->stream));
    HIPCHK(c, hipMemcpyAsync(out, dm, (size_t)n1 * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SSM_OK;
}

// ---------------------------------------------------------------- depth from stereo (cv::StereoSGBM + FrameReader's conversion)
extern "C" void ssm_sgbm_params_default(ssm_sgbm_params* p)
{
    if (!p) return;
    p->minDisparity = 0; p->numberOfDisparities = 80; p->SADWindowSize = 11; p->P1 = 4 * 11 * 11; p->P2 = 32 * 11 * 11;       // src/stereo.cpp:16-27
    p->disp12MaxDiff = 1; p->preFilterCap = 63; p->uniquenessRatio = 10; p->speckleWindowSize = 100; p->speckleRange = 32;
}
// one host pair through the batched kernels (nb = 1): images staged on the device, disparity (and depth) left in the sequence output buffers
static int sgbm_run(ssm_ctx* c, const uint8_t* left, const uint8_t* right, int w, int h, int stride, const ssm_sgbm_params* params, int stage,
                    int16_t** d_disp_out, uint16_t** d_depth_out, int form)
{
    if (!left || !right || !params || w < 3 || h < 1 || stride < w) FAIL(c, SSM_E_INVAL, "bad arguments");
    int r = sgbm_check_params(c, params, w, h); if (r) return r;
    r = stereo_init(c, w, h, c->stereo && c->stereo->w == w && c->stereo->h == h ? c->stereo->maxc : 1000); if (r) return r;
    r = stereo_ensure_seq(c, 1); if (r) return r;
    r = stereo_ensure_sgbm(c, *params, 1); if (r) return r;
    StereoState* q = c->stereo;
    const uint8_t* imgs[2] = { left, right };
    uint8_t* dev = nullptr;
    r = stereo_stage_images(c, imgs, 2, w, h, stride, &dev); if (r) return r;
    if (c->profiling) { c->recs.clear(); c->pool_used = 0; }      // ssm_get_stage_times then reports this call ("sgbm": all kernels of k_sgbm)
    prof_begin(c, "sgbm");
    HIPCHK(c, k_sgbm(dev, dev + (size_t)w * h, w, h, 1, *params, q->sg_wsN[0], q->sg_ws_bytesN[0], q->disp, stage, c->stream, q->sg_fail, form, 1));
    prof_end(c);
    q->sg_pending.valid = false;                                  // (the staged pair is this call's: the host-pointer entry points repeat a timed-out sweep themselves)
    *d_disp_out = q->disp; *d_depth_out = q->depth;
    return SSM_OK;
}
// cv::StereoSGBM cannot fail (src/stereo.cpp:11-30); form 2's sweep can: its strips wait for each other, and when a hand-off exceeds its spin bound every block
// leaves mid-image with the sub-batch's fail word set.  Called with the streams drained: every sub-batch of the last sequence call whose word is set is computed again
// with form 1 (independent paths, no cross-block waits; the workspace holds its volumes anyway) from the caller's input images, so that the call's disparities and depths
// are the oracle's after all.  Reported through ssm_last_error (a note, the call succeeds) and counted in sgbm_fallbacks.
int sgbm_recover(ssm_ctx* c)
{
    StereoState* q = c->stereo;
    int32_t sf[SG_FAIL_WORDS];
    HIPCHK(c, hipMemcpy(sf, q->sg_fail, sizeof(sf), hipMemcpyDeviceToHost));
    bool any = false; for (int k = 0; k < SG_FAIL_WORDS; k++) any = any || sf[k] != 0;
    if (!any) return SSM_OK;
    HIPCHK(c, hipMemset(q->sg_fail, 0, sizeof(sf)));
    if (!q->sg_pending.valid) FAIL(c, SSM_E_HIP, "SGBM sweep: a strip hand-off timed out and the call that launched it is no longer known (the disparities are incomplete)");
    const ssm_stereo_frames_dev& in = q->sg_pending.in; const int B = q->sg_pending.B, w = q->w, h = q->h; const size_t np = (size_t)w * h;
    int redone = 0;
    for (int f0 = 0, bi = 0; f0 < in.n; f0 += B, bi++) {
        if (!sf[bi % SG_FAIL_WORDS]) continue;
        const int nb = in.n - f0 < B ? in.n - f0 : B;
        HIPCHK(c, k_sgbm(in.left + (size_t)f0 * np, in.right + (size_t)f0 * np, w, h, nb, in.sgbm, q->sg_wsN[0], q->sg_ws_bytesN[0], q->disp + (size_t)f0 * np, 0, c->stream, q->sg_fail + bi % SG_FAIL_WORDS, 1, 1));
        HIPCHK(c, k_sgbm_depth(q->disp + (size_t)f0 * np, w, h, nb, in.baseline, in.cu, in.cv, in.f, in.roix, in.roiy, in.roiz, in.scale, q->dminN[0], q->depth + (size_t)f0 * np, c->stream));
        redone++;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->sgbm_fallbacks += redone;
    c->err = "note: the SGBM sweep of " + std::to_string(redone) + " sub-batch(es) timed out in a strip hand-off; they were repeated with form 1 (results complete)";
    return SSM_OK;
}
// the sweep kernel's time-out word, copied to the front of the pinned area with the results of a host-pointer call
static bool sgbm_failed(ssm_ctx* c)
{
    int32_t sf; memcpy(&sf, c->h_pinned, 4);
    if (sf) hipMemset(c->stereo->sg_fail, 0, 4);
    return sf != 0;
}
extern "C" int ssm_sgbm(ssm_ctx* c, const uint8_t* left, const uint8_t* right, int w, int h, int stride, const ssm_sgbm_params* params, int stage, int16_t* disp)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!disp) FAIL(c, SSM_E_INVAL, "null argument");
    c->err.clear();
    int16_t* dd; uint16_t* ddepth;
    const size_t np = (size_t)w * h;
    bool repeated = false;
    for (int attempt = 0; ; attempt++) {                      // a sweep whose hand-off timed out is repeated once, in form 1 (no cross-block waits)
        int r = sgbm_run(c, left, right, w, h, stride, params, stage, &dd, &ddepth, attempt ? 1 : c->sgbm_form_cfg); if (r) return r;
        HIPCHK(c, hipMemcpyAsync(c->h_pinned + 2 * np, dd, np * 2, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->stereo->sg_fail, 4, hipMemcpyDeviceToHost, c->stream));      // (the staged input images at the front of the pinned area are consumed)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!sgbm_failed(c)) break;
        if (attempt) FAIL(c, SSM_E_HIP, "SGBM: the time-out word is set after a form-1 run");
        c->sgbm_fallbacks++; repeated = true;
    }
    memcpy(disp, c->h_pinned + 2 * np, np * 2);
    if (repeated) c->err = "note: the SGBM sweep of this pair timed out in a strip hand-off and was repeated with form 1 (results complete; " + std::to_string(c->sgbm_fallbacks) + " such repeats on this context so far)";
    return SSM_OK;
}
extern "C" int ssm_stereo_depth(ssm_ctx* c, const uint8_t* left, const uint8_t* right, int w, int h, int stride, const ssm_sgbm_params* params,
                                double baseline, double cu, double cv, double f, double roix, double roiy, double roiz, double scale,
                                uint16_t* depth, int16_t* disp)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!depth) FAIL(c, SSM_E_INVAL, "null argument");
    c->err.clear();
    int16_t* dd; uint16_t* ddepth;
    const size_t np = (size_t)w * h;
    bool repeated = false;
    for (int attempt = 0; ; attempt++) {
        int r = sgbm_run(c, left, right, w, h, stride, params, 0, &dd, &ddepth, attempt ? 1 : c->sgbm_form_cfg); if (r) return r;
        HIPCHK(c, k_sgbm_depth(dd, w, h, 1, baseline, cu, cv, f, roix, roiy, roiz, scale, c->stereo->dminN[0], ddepth, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_pinned + 4 * np, ddepth, np * 2, hipMemcpyDeviceToHost, c->stream));
        if (disp) HIPCHK(c, hipMemcpyAsync(c->h_pinned + 2 * np, dd, np * 2, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->stereo->sg_fail, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!sgbm_failed(c)) break;
        if (attempt) FAIL(c, SSM_E_HIP, "SGBM: the time-out word is set after a form-1 run");
        c->sgbm_fallbacks++; repeated = true;
    }
    memcpy(depth, c->h_pinned + 4 * np, np * 2);
    if (disp) memcpy(disp, c->h_pinned + 2 * np, np * 2);
    if (repeated) c->err = "note: the SGBM sweep of this pair timed out in a strip hand-off and was repeated with form 1 (results complete; " + std::to_string(c->sgbm_fallbacks) + " such repeats on this context so far)";
    return SSM_OK;
}

// ---------------------------------------------------------------- VisualOdometryStereo::estimateMotion
extern "C" int ssm_vo_estimate(ssm_ctx* c, const ssm_pmatch* matches, int n, const ssm_vo_params* params, const int32_t* samples, int iters,
                               double tr[6], int32_t* inliers, int cap, int* n_inliers, int* success)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (n < 0 || iters < 0 || !params || !tr || !n_inliers || !success || (n && !matches) || (iters && !samples)) FAIL(c, SSM_E_INVAL, "bad arguments");
    for (int k = 0; k < 6; k++) tr[k] = 0.0;
    *n_inliers = 0; *success = 0;
    if (n < 6) return SSM_OK;                                 // estimateMotion returns an empty vector (vo_stereo.cpp:61-63)
    for (int k = 0; k < 3 * iters; k++) if (samples[k] < 0 || samples[k] >= n) FAIL(c, SSM_E_INVAL, "sample index out of range");
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_m = 0, o_s = o_m + al((size_t)n * sizeof(ssm_pmatch)), o_tr = o_s + al((size_t)iters * 12 + 16), o_cnt = o_tr + al((size_t)iters * 48 + 48),
                 o_out = o_cnt + al((size_t)iters * 4 + 16), o_inl = o_out + 256, o_res = o_inl + al((size_t)n * 4), total = o_res + 256;
    int r = ensure_scratch(c, total); if (r) return r;
    uint8_t* p = (uint8_t*)c->d_scratch; hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(p + o_m, matches, (size_t)n * sizeof(ssm_pmatch), hipMemcpyHostToDevice, s));
    if (iters) HIPCHK(c, hipMemcpyAsync(p + o_s, samples, (size_t)iters * 12, hipMemcpyHostToDevice, s));
    if (c->profiling) { c->recs.clear(); c->pool_used = 0; }
    prof_begin(c, "vo");
    HIPCHK(c, k_vo_estimate((const ssm_pmatch*)(p + o_m), n, *params, (const int32_t*)(p + o_s), iters, (double*)(p + o_tr), (int32_t*)(p + o_cnt),
                            (double*)(p + o_out), (int32_t*)(p + o_inl), (int32_t*)(p + o_res), s));
    prof_end(c);
    int32_t res[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(tr, p + o_out, 48, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(res, p + o_res, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    *n_inliers = res[0]; *success = res[1];
    if (inliers && res[0] > 0) {
        if (res[0] > cap) FAIL(c, SSM_E_CAPACITY, "inlier buffer too small (need " + std::to_string(res[0]) + ")");
        HIPCHK(c, hipMemcpy(inliers, p + o_inl, (size_t)res[0] * 4, hipMemcpyDeviceToHost));
    }
    return SSM_OK;
}

// PnPSolver::solvePnP (reference src/pnp.cpp:5-118) for one correspondence list: the solve that the pose chain of kernels_pnp.hip runs per frame.
// Round 6: a cluster of eight blocks per call, like the chain's (every block evaluates an eighth of the lanes' edges in a pass and the blocks trade the partial sums:
// same bits as one block, the per-frame Tracker's largest call 0.53 -> ~0.35 ms); one block after a cluster timed out once (the blocks need eight CUs at the same
// time) or with SSM_PNP_BLOCKS=1.
static int pnp_solve_impl(ssm_ctx* c, const float* img, const float* obj, int n, const double cam[4], int min_inliers, double T[16], uint8_t* inliers, int* n_inliers, int* success, int G)
{
    // ONE upload ([img | obj | header: T in, T out, time-out word, inlier count] staged in pinned memory) and ONE download ([T out .. count | the inlier flags]): the
    // call was three uploads, a fill of the exchange ring and four downloads (0.09 ms of its 0.42).  The cluster's ring lives in the context and is zeroed when its
    // pass numbers wrap, not per call.
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t ne = (size_t)(n > 0 ? n : 1), slice = al(ne);
    const size_t o_img = 0, o_obj = o_img + ne * 8, o_hdr = (o_obj + ne * 12 + 15) & ~(size_t)15, HDR = 288 /* T in 128 | T out 128 | xfail 4, pad 4 | count 4, pad 4 | pad 16 */,
                 o_inl = o_hdr + HDR, o_dec = o_inl + G * slice, o_le = al(o_dec + G * slice), o_err = o_le + al(G * slice * k_pnp_edge_bytes()), total = o_err + al(G * slice * 16);
    const size_t up = o_hdr + HDR, down = HDR - 128 + (size_t)n;               // download: [T out | xfail | count | pad][inl slice 0: n bytes]
    int r = ensure_scratch(c, total); if (r) return r;
    r = ensure_pinned(c, al(up) + al(down)); if (r) return r;
    if (G > 1) {
        if (!c->d_pnp_xchg) { HIPCHK(c, hipMalloc((void**)&c->d_pnp_xchg, k_pnp_xchg_bytes())); c->pnp_epoch = 0; }
        if (c->pnp_epoch == 0) HIPCHK(c, hipMemsetAsync(c->d_pnp_xchg, 0, k_pnp_xchg_bytes(), c->stream));
    }
    uint8_t* p = (uint8_t*)c->d_scratch; hipStream_t s = c->stream;
    uint8_t* hu = c->h_pinned; uint8_t* hd = c->h_pinned + al(up);
    if (n) { memcpy(hu + o_img, img, (size_t)n * 8); memcpy(hu + o_obj, obj, (size_t)n * 12); }
    memset(hu + o_hdr, 0, HDR); memcpy(hu + o_hdr, T, 128);
    { const char* tv = getenv("SSM_PNP_TEST_TIMEOUT"); if (G > 1 && tv && atoi(tv) != 0) { const unsigned one = 1; memcpy(hu + o_hdr + 256, &one, 4); } }      // tests: the time-out word set from the start -> the one-block retry
    HIPCHK(c, hipMemcpyAsync(p, hu, up, hipMemcpyHostToDevice, s));
    PnpSolveArgs a; a.img = (const float*)(p + o_img); a.obj = (const float*)(p + o_obj); a.n = n;
    a.cam.fx = cam[0]; a.cam.fy = cam[1]; a.cam.cx = cam[2]; a.cam.cy = cam[3];
    a.T = (double*)(p + o_hdr); a.inl = p + o_inl; a.dec = p + o_dec; a.ledges = (LEdge*)(p + o_le); a.err = (double2*)(p + o_err); a.n_inliers = (int32_t*)(p + o_hdr + 264); a.edges_in_lds = 0;
    a.blocks = G; a.slice = slice; a.xchg = c->d_pnp_xchg; a.xfail = reinterpret_cast<unsigned*>(p + o_hdr + 256);
    a.seq_base = (unsigned)c->pnp_epoch << 20;                  // a solve makes a few thousand passes at most; the ring is zeroed again when the epoch wraps
    if (G > 1) c->pnp_epoch = (c->pnp_epoch + 1) & 4095;
    if (c->profiling) { c->recs.clear(); c->pool_used = 0; }
    prof_begin(c, "pnp");
    HIPCHK(c, k_pnp_solve(a, s));
    prof_end(c);
    HIPCHK(c, hipMemcpyAsync(hd, p + o_hdr + 128, down, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    unsigned failed = 0; int32_t m = 0;
    memcpy(&failed, hd + 128, 4); memcpy(&m, hd + 136, 4);
    if (G > 1 && failed) { c->pnp_epoch = 0; return 1; }        // an exchange of the cluster timed out: nothing of this attempt is used (and the ring starts clean next time)
    memcpy(T, hd, 128);
    if (inliers && n) memcpy(inliers, hd + HDR - 128, (size_t)n);
    *n_inliers = m;
    if (success) *success = n > min_inliers;                   // pnp.cpp:115 tests the flag vector's LENGTH (quirk 14)
    return SSM_OK;
}
extern "C" int ssm_pnp_solve(ssm_ctx* c, const float* img, const float* obj, int n, const double cam[4], int min_inliers, double T[16],
                             uint8_t* inliers, int* n_inliers, int* success)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (n < 0 || !cam || !T || !n_inliers || (n && (!img || !obj))) FAIL(c, SSM_E_INVAL, "bad arguments");
    if (n > 65535) FAIL(c, SSM_E_CAPACITY, "at most 65535 correspondences");
    static const int env_blocks = [] { const char* e = getenv("SSM_PNP_BLOCKS"); return e ? atoi(e) : 8; }();
    if (env_blocks == 8 && !c->pnp_solve_one_block) {
        const int r = pnp_solve_impl(c, img, obj, n, cam, min_inliers, T, inliers, n_inliers, success, 8);
        if (r <= 0) return r;
        c->pnp_solve_one_block = true;                          // (T is untouched: the retry starts from the caller's initial value)
    }
    return pnp_solve_impl(c, img, obj, n, cam, min_inliers, T, inliers, n_inliers, success, 1);
}
