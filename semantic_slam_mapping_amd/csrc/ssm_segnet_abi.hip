// ssm_segnet_abi.hip -- Classifier (reference src/segnet.cpp) behind the C ABI: weight packing, the layer schedule of the driving_webdemo network, ssm_segnet_*.
// Kernels: kernels_segnet.hip.
#include "ssm_ctx.h"

// ---------------------------------------------------------------- SegNet (Classifier)
static inline uint16_t f32_to_f16(float f)
{
    _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u;
}
extern "C" int ssm_segnet_num_layers(void) { return SEG_LAYERS; }
extern "C" int ssm_segnet_layer_shape(int l, int* cin, int* cout, int* h, int* w)
{
    if (l < 0 || l >= SEG_LAYERS) return SSM_E_INVAL;
    if (cin) *cin = k_seg_layers[l].cin; if (cout) *cout = k_seg_layers[l].cout; if (h) *h = k_seg_layers[l].h; if (w) *w = k_seg_layers[l].w;
    return SSM_OK;
}
int seg_init(ssm_ctx* c)
{
    if (c->seg) return SSM_OK;
    SegNetState* g = new SegNetState();
    c->seg = g;
    for (int l = 0; l < SEG_LAYERS; l++) {
        g->cinp[l] = k_seg_layers[l].cin <= 8 ? 8 : (k_seg_layers[l].cin + 63) & ~63;   // <= 8 channels: the first-layer kernel ([H][W][8] input)
        g->coutp[l] = (k_seg_layers[l].cout + 63) & ~63;
        g->coutstore[l] = (k_seg_layers[l].cout + 31) & ~31;            // activations live in 32-channel chunks: [C/32][H][W][32]
    }
    {   // frames per SegNet launch: 64 by default (more tiles per launch: better balance in the small layers; the 64-channel layers bound it to 96 by their 2^31-byte buffers)
        int sb = 64;
        g->batch = c->B < sb ? c->B : sb;
    }
    const size_t act = (size_t)g->batch * SEG_NW * SEG_NH * 64 * 2;
    uint8_t* p;
    int r = dalloc(c, &p, act); if (r) return r; g->actA = p;
    r = dalloc(c, &p, act); if (r) return r; g->actB = p;
    const int ph[5] = {180, 90, 45, 23, 12}, pw[5] = {240, 120, 60, 30, 15}, pc[5] = {64, 128, 256, 512, 512};
    for (int i = 0; i < 5; i++) DALLOC(c, g->code[i], (size_t)g->batch * ph[i] * pw[i] * pc[i]);
    DALLOC(c, g->labels, (size_t)g->batch * SEG_NW * SEG_NH);
    DALLOC(c, g->d_sem_gen, (size_t)c->B * c->g.W * c->g.H * 3);
    auto up = [&](int ssize, int dsize, int32_t** o, int16_t** a) -> int {
        std::vector<int32_t> ofs; std::vector<int16_t> co; resize_tables(ssize, dsize, ofs, co);
        int rr = dalloc(c, o, ofs.size()); if (rr) return rr; rr = dalloc(c, a, co.size()); if (rr) return rr;
        if (hipMemcpy(*o, ofs.data(), ofs.size() * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(*a, co.data(), co.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { c->err = "segnet table upload"; return SSM_E_HIP; }
        return SSM_OK;
    };
    if ((r = up(c->g.W, SEG_NW, &g->pre_xofs, &g->pre_xa)) || (r = up(c->g.H, SEG_NH, &g->pre_yofs, &g->pre_ya)) ||
        (r = up(SEG_NW, c->g.W, &g->post_xofs, &g->post_xa)) || (r = up(SEG_NH, c->g.H, &g->post_yofs, &g->post_ya))) return r;
    return SSM_OK;
}
// the layer's Winograd weights when that kernel is to run it: SSM_CONV_WINOGRAD=1 (A/B runs and tests).  The default is the direct kernel on every layer: the
// Winograd F(2, 3) kernel is bit-exact on integer data and 1.5 x lighter on the matrix cores, but it stages 2.8 x the bytes per MFMA (32-cout tiles: the sixteen
// accumulators a 64-cout tile needs do not fit two waves per SIMD; 12 taps instead of 9) and ends up bound by the L2 -> LDS path: 203 vs 182 us per frame for the
// stage (profiles/r06_segnet_winograd.md)
static const void* seg_wino(const SegNetState* g, int l)
{
    static const bool on = [] { const char* e = getenv("SSM_CONV_WINOGRAD"); return e && atoi(e) != 0; }();
    return on ? g->ww[l] : nullptr;
}
extern "C" int ssm_segnet_set_layer(ssm_ctx* c, int l, const float* weight, const float* scale, const float* shift)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (l < 0 || l >= SEG_LAYERS || !weight || !scale || !shift) FAIL(c, SSM_E_INVAL, "bad arguments");
    int r = seg_init(c); if (r) return r;
    SegNetState* g = c->seg;
    const int cin = k_seg_layers[l].cin, cout = k_seg_layers[l].cout, cinp = g->cinp[l], coutp = g->coutp[l];
    std::vector<uint16_t> w;
    if (cinp == 8) {
        // first-layer kernel: [Cout tile of 64][K step 5][half 2][cout in tile 64][8 channels], tap = 2 step + half (tap 9: zeros)
        w.assign((size_t)coutp * 10 * 8, 0);
        for (int o = 0; o < cout; o++) for (int i = 0; i < cin; i++) for (int t = 0; t < 9; t++)
            w[((((size_t)(o / 64) * 5 + t / 2) * 2 + t % 2) * 64 + o % 64) * 8 + i] = f32_to_f16(weight[((size_t)o * cin + i) * 9 + t]);
    } else {
        // LDS-DMA kernel: [Cout tile of 64][Cin chunk of 32][tap][c8 (4)][cout in tile (64)][8 channels]
        w.assign((size_t)coutp * 9 * cinp, 0);
        const int nck = cinp / 32;
        for (int o = 0; o < cout; o++) for (int i = 0; i < cin; i++) for (int t = 0; t < 9; t++) {
            const size_t idx = ((((((size_t)(o / 64) * nck + i / 32) * 9 + t) * 4 + (i % 32) / 8) * 64 + o % 64) * 8) + i % 8;
            w[idx] = f32_to_f16(weight[((size_t)o * cin + i) * 9 + t]);
        }
    }
    // the same weights for the Winograd F(2, 3) kernel (plain conv + BN + ReLU layers with whole 32-channel chunks on both sides): per dy the three taps of a row become
    // U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2, in fp32, stored fp16 as [cout tile 64][cin chunk 32][tap = 4 dy + k][c8 4][cout 64][8]
    std::vector<uint16_t> wwv;
    if (cinp % 64 == 0 && cout % 32 == 0 && cinp != 8) {
        wwv.assign((size_t)coutp * 12 * cinp, 0);
        const int nck = cinp / 32;
        for (int o = 0; o < cout; o++) for (int i = 0; i < cin; i++) for (int dy = 0; dy < 3; dy++) {
            const float* gw = weight + ((size_t)o * cin + i) * 9 + dy * 3;
            const float u[4] = { gw[0], (gw[0] + gw[1] + gw[2]) * 0.5f, (gw[0] - gw[1] + gw[2]) * 0.5f, gw[2] };
            for (int k = 0; k < 4; k++) {
                const size_t idx = ((((((size_t)(o / 64) * nck + i / 32) * 12 + (dy * 4 + k)) * 4 + (i % 32) / 8) * 64 + o % 64) * 8) + i % 8;
                wwv[idx] = f32_to_f16(u[k]);
            }
        }
    }
    if (!wwv.empty()) {
        if (!g->ww[l]) { uint16_t* p; r = dalloc(c, &p, wwv.size()); if (r) return r; g->ww[l] = p; }
        HIPCHK(c, hipMemcpy(g->ww[l], wwv.data(), wwv.size() * 2, hipMemcpyHostToDevice));
    }
    if (!g->w[l]) { uint16_t* p; r = dalloc(c, &p, w.size()); if (r) return r; g->w[l] = p; DALLOC(c, g->scale[l], coutp); DALLOC(c, g->shift[l], coutp); }
    std::vector<float> sc(coutp, 0.f), sh(coutp, 0.f);
    for (int o = 0; o < cout; o++) { sc[o] = scale[o]; sh[o] = shift[o]; }
    HIPCHK(c, hipMemcpy(g->w[l], w.data(), w.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(g->scale[l], sc.data(), coutp * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(g->shift[l], sh.data(), coutp * 4, hipMemcpyHostToDevice));
    g->set[l] = true;
    return SSM_OK;
}
// forward for nb <= seg->batch device frames already pre-processed into actA; leaves logits in the returned buffer
// logits_out != nullptr: the class logits are materialised (returned buffer) and the caller runs the ArgMax kernel;
// logits_out == nullptr: the last layer writes the labels (g->labels) straight from its epilogue.
static int seg_forward_core(ssm_ctx* c, int nb, void** logits_out)
{
    SegNetState* g = c->seg; hipStream_t s = c->stream;
    void* cur = g->actA; void* nxt = g->actB;
    HIPCHK(c, k_segnet_begin(s));
    auto conv = [&](int l) -> int {
        const SegLayerDef& d = k_seg_layers[l];
        HIPCHK(c, k_segnet_conv(cur, g->w[l], g->scale[l], g->shift[l], nxt, nb, d.h, d.w, g->cinp[l], d.cout, l != SEG_LAYERS - 1, s, seg_wino(g, l)));
        std::swap(cur, nxt); return SSM_OK;
    };
    auto unpool = [&](int i, int PH, int PW, int C, int H, int W) -> int { HIPCHK(c, k_segnet_unpool(cur, g->code[i], nb, PH, PW, C, nxt, H, W, s)); std::swap(cur, nxt); return SSM_OK; };
    // un-pool + the convolution that consumes it as one kernel (the 4x sparse tensor is never written); the other conv kernels
    // (SSM_CONV_VARIANT) run the two steps
    const bool fused_up = k_segnet_conv_unpool_available() != 0;
    auto unpool_conv = [&](int i, int PH, int PW, int C, int H, int W, int l) -> int {
        // measured per layer (32 frames): the fused form wins where the un-pooled tensor is large (64 ch @360x480: 357 vs 586 us,
        // 128 ch @180x240: 330 vs 433, 256 ch @90x120: 338 vs 354) and loses on the small 512-channel images, where the masking
        // pass on the stage's critical path costs more than the separate un-pool (362 vs 327, 121 vs 103 us)
        if (!fused_up || C > 256) { int r_ = unpool(i, PH, PW, C, H, W); return r_ ? r_ : conv(l); }
        const SegLayerDef& d = k_seg_layers[l];
        HIPCHK(c, k_segnet_conv_unpool(cur, g->code[i], g->w[l], g->scale[l], g->shift[l], nxt, nb, d.h, d.w, g->cinp[l], d.cout, s));
        std::swap(cur, nxt); return SSM_OK;
    };
    // conv + pool pairs run as one kernel (the full-resolution activation of the pooled layer is never written)
    auto conv_pool = [&](int l, int i) -> int {
        const SegLayerDef& d = k_seg_layers[l];
        HIPCHK(c, k_segnet_conv_pool(cur, g->w[l], g->scale[l], g->shift[l], nxt, g->code[i], nb, d.h, d.w, g->cinp[l], d.cout, s));
        std::swap(cur, nxt); return SSM_OK;
    };
    int r;
    if ((r = conv(0)) || (r = conv_pool(1, 0))) return r;
    if ((r = conv(2)) || (r = conv_pool(3, 1))) return r;
    if ((r = conv(4)) || (r = conv(5)) || (r = conv_pool(6, 2))) return r;
    if ((r = conv(7)) || (r = conv(8)) || (r = conv_pool(9, 3))) return r;
    if ((r = conv(10)) || (r = conv(11)) || (r = conv_pool(12, 4))) return r;
    if ((r = unpool_conv(4, 12, 15, 512, 23, 30, 13)) || (r = conv(14)) || (r = conv(15))) return r;
    if ((r = unpool_conv(3, 23, 30, 512, 45, 60, 16)) || (r = conv(17)) || (r = conv(18))) return r;
    if ((r = unpool_conv(2, 45, 60, 256, 90, 120, 19)) || (r = conv(20)) || (r = conv(21))) return r;
    if ((r = unpool_conv(1, 90, 120, 128, 180, 240, 22)) || (r = conv(23))) return r;
    if ((r = unpool_conv(0, 180, 240, 64, 360, 480, 24))) return r;
    if (logits_out) { if ((r = conv(25))) return r; *logits_out = cur; }
    else {
        const SegLayerDef& d = k_seg_layers[25];
        HIPCHK(c, k_segnet_conv_argmax(cur, g->w[25], g->scale[25], g->shift[25], g->labels, nb, d.h, d.w, g->cinp[25], d.cout, s));
    }
    return SSM_OK;
}
int seg_forward_dev(ssm_ctx* c, const uint8_t* bgr, int n, uint8_t* labels_net, uint8_t* sem_bgr, int flags)
{
    int r = seg_init(c); if (r) return r;
    SegNetState* g = c->seg;
    for (int l = 0; l < SEG_LAYERS; l++) if (!g->set[l]) FAIL(c, SSM_E_INVAL, "SegNet layer " + std::to_string(l) + " has no weights (ssm_segnet_set_layer)");
    const int W = c->g.W, H = c->g.H; hipStream_t s = c->stream;
    for (int f0 = 0; f0 < n; f0 += g->batch) {
        const int nb = n - f0 < g->batch ? n - f0 : g->batch;
        HIPCHK(c, k_segnet_prep(bgr + (size_t)f0 * W * H * 3, nb, W, H, SEG_NW, SEG_NH, g->pre_xofs, g->pre_xa, g->pre_yofs, g->pre_ya, g->actA, s));
        if (flags & 4) {                           // keep the class logits (ssm_segnet_forward / ssm_segnet_logits): separate ArgMax kernel
            void* logits = nullptr;
            r = seg_forward_core(c, nb, &logits); if (r) return r;
            g->last_logits = logits;               // frame f0 of the last sub-batch starts the buffer
            HIPCHK(c, k_segnet_argmax(logits, nb, SEG_NW * SEG_NH, g->coutstore[SEG_LAYERS - 1], SEG_NCLS, g->labels, s));
        } else {
            r = seg_forward_core(c, nb, nullptr); if (r) return r;
            g->last_logits = nullptr;
        }
        if (labels_net) HIPCHK(c, hipMemcpyAsync(labels_net + (size_t)f0 * SEG_NW * SEG_NH, g->labels, (size_t)nb * SEG_NW * SEG_NH, hipMemcpyDeviceToDevice, s));
        if (sem_bgr) HIPCHK(c, k_segnet_color(g->labels, nb, SEG_NW, SEG_NH, W, H, g->post_xofs, g->post_xa, g->post_yofs, g->post_ya,
                                              !(flags & 2), flags & 1, sem_bgr + (size_t)f0 * W * H * 3, nullptr, s));
    }
    return SSM_OK;
}
extern "C" int ssm_segnet_forward_dev(ssm_ctx* c, const uint8_t* bgr, int n, uint8_t* labels_net, uint8_t* sem_bgr, int flags)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!bgr || n < 0) FAIL(c, SSM_E_INVAL, "bad arguments");
    return seg_forward_dev(c, bgr, n, labels_net, sem_bgr, flags);
}
extern "C" int ssm_segnet_forward(ssm_ctx* c, const uint8_t* bgr, int w, int h, int stride, uint8_t* labels_net, uint8_t* sem_bgr)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!bgr) FAIL(c, SSM_E_INVAL, "null argument");
    if (w != c->g.W || h != c->g.H) FAIL(c, SSM_E_INVAL, "frame size differs from the context configuration");
    if (stride < w * 3) FAIL(c, SSM_E_INVAL, "stride smaller than a row");
    HIPCHK(c, hipMemcpy2DAsync(c->d_in_img, (size_t)w * 3, bgr, stride, (size_t)w * 3, h, hipMemcpyHostToDevice, c->stream));
    int r = ensure_scratch(c, (size_t)SEG_NW * SEG_NH); if (r) return r;
    r = seg_forward_dev(c, c->d_in_img, 1, labels_net ? (uint8_t*)c->d_scratch : nullptr, sem_bgr ? c->d_in_sem : nullptr, 4); if (r) return r;
    if (labels_net) HIPCHK(c, hipMemcpyAsync(labels_net, c->d_scratch, (size_t)SEG_NW * SEG_NH, hipMemcpyDeviceToHost, c->stream));
    if (sem_bgr) HIPCHK(c, hipMemcpyAsync(sem_bgr, c->d_in_sem, (size_t)w * h * 3, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SSM_OK;
}
extern "C" int ssm_segnet_debug_op(ssm_ctx* c, int op, int arg, const uint16_t* in, int H, int W, uint16_t* out, uint8_t* code)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!in || !out || H < 1 || W < 1 || (size_t)H * W > (size_t)SEG_NW * SEG_NH) FAIL(c, SSM_E_INVAL, "bad arguments");
    int r = seg_init(c); if (r) return r;
    SegNetState* g = c->seg; hipStream_t s = c->stream;
    const int PH = (H + 1) / 2, PW = (W + 1) / 2;
    if (op == 0) {
        if (arg < 0 || arg >= SEG_LAYERS || !g->set[arg]) FAIL(c, SSM_E_INVAL, "layer not set");
        // host tensors are NHWC with channels padded to 16; the device layout is [C/32][H][W][32]
        const int ci16 = (k_seg_layers[arg].cin + 15) & ~15, co16 = (k_seg_layers[arg].cout + 15) & ~15;
        std::vector<uint16_t> hin((size_t)H * W * g->cinp[arg], 0), hout((size_t)H * W * g->coutstore[arg]);
        if (g->cinp[arg] == 8) { for (size_t p = 0; p < (size_t)H * W; p++) for (int ch = 0; ch < k_seg_layers[arg].cin; ch++) hin[p * 8 + ch] = in[p * ci16 + ch]; }
        else for (size_t p = 0; p < (size_t)H * W; p++) for (int ch = 0; ch < ci16; ch++) hin[((size_t)(ch / 32) * H * W + p) * 32 + ch % 32] = in[p * ci16 + ch];
        HIPCHK(c, hipMemcpyAsync(g->actA, hin.data(), hin.size() * 2, hipMemcpyHostToDevice, s));
        HIPCHK(c, k_segnet_conv(g->actA, g->w[arg], g->scale[arg], g->shift[arg], g->actB, 1, H, W, g->cinp[arg], k_seg_layers[arg].cout, arg != SEG_LAYERS - 1, s, seg_wino(g, arg)));
        HIPCHK(c, hipMemcpyAsync(hout.data(), g->actB, hout.size() * 2, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        for (size_t p = 0; p < (size_t)H * W; p++) for (int ch = 0; ch < co16; ch++) out[p * co16 + ch] = hout[((size_t)(ch / 32) * H * W + p) * 32 + ch % 32];
    } else if (op == 1 || op == 2) {
        const int C = arg;
        if (C < 32 || C > 512 || (C & 31) || !code) FAIL(c, SSM_E_INVAL, "channel count must be a multiple of 32 (the activation chunk)");
        r = ensure_scratch(c, (size_t)PH * PW * C); if (r) return r;
        uint8_t* dcode = (uint8_t*)c->d_scratch;
        // host NHWC <-> device [C/32][h][w][32]; the arg-max codes use the same element order as the pooled tensor
        auto to_dev = [&](const uint16_t* src, int hh, int ww, std::vector<uint16_t>& d) { d.assign((size_t)hh * ww * C, 0); for (size_t p = 0; p < (size_t)hh * ww; p++) for (int ch = 0; ch < C; ch++) d[((size_t)(ch / 32) * hh * ww + p) * 32 + ch % 32] = src[p * C + ch]; };
        auto to_host = [&](const std::vector<uint16_t>& d, int hh, int ww, uint16_t* dst) { for (size_t p = 0; p < (size_t)hh * ww; p++) for (int ch = 0; ch < C; ch++) dst[p * C + ch] = d[((size_t)(ch / 32) * hh * ww + p) * 32 + ch % 32]; };
        std::vector<uint16_t> hin, hout; std::vector<uint8_t> hcode((size_t)PH * PW * C);
        if (op == 1) {
            to_dev(in, H, W, hin); hout.resize((size_t)PH * PW * C);
            HIPCHK(c, hipMemcpyAsync(g->actA, hin.data(), hin.size() * 2, hipMemcpyHostToDevice, s));
            HIPCHK(c, k_segnet_pool(g->actA, 1, H, W, C, g->actB, dcode, s));
            HIPCHK(c, hipMemcpyAsync(hout.data(), g->actB, hout.size() * 2, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(hcode.data(), dcode, hcode.size(), hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            to_host(hout, PH, PW, out);
            for (size_t p = 0; p < (size_t)PH * PW; p++) for (int ch = 0; ch < C; ch++) code[p * C + ch] = hcode[((size_t)(ch / 32) * PH * PW + p) * 32 + ch % 32];
        } else {
            to_dev(in, PH, PW, hin); hout.resize((size_t)H * W * C);
            for (size_t p = 0; p < (size_t)PH * PW; p++) for (int ch = 0; ch < C; ch++) hcode[((size_t)(ch / 32) * PH * PW + p) * 32 + ch % 32] = code[p * C + ch];
            HIPCHK(c, hipMemcpyAsync(g->actA, hin.data(), hin.size() * 2, hipMemcpyHostToDevice, s));
            HIPCHK(c, hipMemcpyAsync(dcode, hcode.data(), hcode.size(), hipMemcpyHostToDevice, s));
            HIPCHK(c, k_segnet_unpool(g->actA, dcode, 1, PH, PW, C, g->actB, H, W, s));
            HIPCHK(c, hipMemcpyAsync(hout.data(), g->actB, hout.size() * 2, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            to_host(hout, H, W, out);
        }
    } else if (op == 3) {                 // conv + BN + ReLU + max-pool of layer `arg` as the network runs it (one kernel)
        if (arg < 0 || arg >= SEG_LAYERS || !g->set[arg] || !code || g->cinp[arg] == 8) FAIL(c, SSM_E_INVAL, "layer not set, or not one the fused conv+pool kernel takes");
        const int ci16 = (k_seg_layers[arg].cin + 15) & ~15, co16 = (k_seg_layers[arg].cout + 15) & ~15, cs = g->coutstore[arg];
        std::vector<uint16_t> hin((size_t)H * W * g->cinp[arg], 0), hout((size_t)PH * PW * cs);
        std::vector<uint8_t> hcode((size_t)PH * PW * cs);
        for (size_t p = 0; p < (size_t)H * W; p++) for (int ch = 0; ch < ci16; ch++) hin[((size_t)(ch / 32) * H * W + p) * 32 + ch % 32] = in[p * ci16 + ch];
        r = ensure_scratch(c, hcode.size()); if (r) return r;
        uint8_t* dcode = (uint8_t*)c->d_scratch;
        HIPCHK(c, hipMemcpyAsync(g->actA, hin.data(), hin.size() * 2, hipMemcpyHostToDevice, s));
        HIPCHK(c, k_segnet_conv_pool(g->actA, g->w[arg], g->scale[arg], g->shift[arg], g->actB, dcode, 1, H, W, g->cinp[arg], k_seg_layers[arg].cout, s));
        HIPCHK(c, hipMemcpyAsync(hout.data(), g->actB, hout.size() * 2, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(hcode.data(), dcode, hcode.size(), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        for (size_t p = 0; p < (size_t)PH * PW; p++) for (int ch = 0; ch < co16; ch++) {
            out[p * co16 + ch] = hout[((size_t)(ch / 32) * PH * PW + p) * 32 + ch % 32];
            code[p * co16 + ch] = hcode[((size_t)(ch / 32) * PH * PW + p) * 32 + ch % 32];
        }
    } else if (op == 4) {                 // un-pool (in = pooled PH x PW image of the layer's input channels, code = its arg-max codes) + conv + BN + ReLU of layer `arg`, one kernel
        if (arg < 0 || arg >= SEG_LAYERS || !g->set[arg] || !code || g->cinp[arg] == 8) FAIL(c, SSM_E_INVAL, "layer not set, or not one the fused un-pool + conv kernel takes");
        if (!k_segnet_conv_unpool_available()) FAIL(c, SSM_E_INVAL, "the selected conv kernel (SSM_CONV_VARIANT) has no un-pool-on-load form");
        const int ci16 = (k_seg_layers[arg].cin + 15) & ~15, co16 = (k_seg_layers[arg].cout + 15) & ~15, cs = g->coutstore[arg], cip = g->cinp[arg];
        std::vector<uint16_t> hin((size_t)PH * PW * cip, 0), hout((size_t)H * W * cs);
        std::vector<uint8_t> hcode((size_t)PH * PW * cip, 0);
        for (size_t p = 0; p < (size_t)PH * PW; p++) for (int ch = 0; ch < ci16; ch++) {
            hin[((size_t)(ch / 32) * PH * PW + p) * 32 + ch % 32] = in[p * ci16 + ch];
            hcode[((size_t)(ch / 32) * PH * PW + p) * 32 + ch % 32] = code[p * ci16 + ch];
        }
        r = ensure_scratch(c, hcode.size()); if (r) return r;
        uint8_t* dcode = (uint8_t*)c->d_scratch;
        HIPCHK(c, k_segnet_begin(s));
        HIPCHK(c, hipMemcpyAsync(g->actA, hin.data(), hin.size() * 2, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(dcode, hcode.data(), hcode.size(), hipMemcpyHostToDevice, s));
        HIPCHK(c, k_segnet_conv_unpool(g->actA, dcode, g->w[arg], g->scale[arg], g->shift[arg], g->actB, 1, H, W, cip, k_seg_layers[arg].cout, s));
        HIPCHK(c, hipMemcpyAsync(hout.data(), g->actB, hout.size() * 2, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        for (size_t p = 0; p < (size_t)H * W; p++) for (int ch = 0; ch < co16; ch++) out[p * co16 + ch] = hout[((size_t)(ch / 32) * H * W + p) * 32 + ch % 32];
    } else FAIL(c, SSM_E_INVAL, "unknown op");
    HIPCHK(c, hipStreamSynchronize(s));
    return SSM_OK;
}
extern "C" int ssm_segnet_logits(ssm_ctx* c, float* out)
{
    if (!c || !out) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!c->seg) FAIL(c, SSM_E_INVAL, "no forward has run");
    if (!c->seg->last_logits) FAIL(c, SSM_E_INVAL, "no forward has run");
    const int cs = c->seg->coutstore[SEG_LAYERS - 1];
    std::vector<uint16_t> h((size_t)SEG_NW * SEG_NH * cs);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(h.data(), c->seg->last_logits, h.size() * 2, hipMemcpyDeviceToHost));
    for (size_t p = 0; p < (size_t)SEG_NW * SEG_NH; p++)
        for (int k = 0; k < SEG_NCLS; k++) { _Float16 v; memcpy(&v, &h[p * cs + k], 2); out[p * SEG_NCLS + k] = (float)v; }
    return SSM_OK;
}


