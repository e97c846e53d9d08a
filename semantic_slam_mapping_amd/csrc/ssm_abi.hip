// ssm_abi.hip -- host side of libssm_hip.so: context, geometry, device workspace and the extern "C" entry points
// declared in include/ssm_hip.h.  No computation of the path happens on the host: this file only sizes buffers,
// moves caller data and enqueues the kernels of kernels_*.hip on the context stream.  There is NO CPU fallback: if
// HIP is unusable ssm_create fails with SSM_E_NODEVICE / SSM_E_HIP.
#include "ssm_ctx.h"

static const int8_t k_default_pattern[1024] = {
#include "orb_pattern.inc"
};
thread_local std::string g_create_err;

// ---------------------------------------------------------------- geometry (mirrors ORBextractor ctor / ComputePyramid)
static int build_geometry(const ssm_config& c, OrbGeom& g, std::string& err)
{
    memset(&g, 0, sizeof(g));
    if (c.orb_levels < 1 || c.orb_levels > SSM_MAX_LEVELS) { err = "orb_levels must be 1..12"; return SSM_E_INVAL; }
    if (c.orb_features < 1) { err = "orb_features must be >= 1"; return SSM_E_INVAL; }
    if (c.orb_iniThFAST < 1 || c.orb_minThFAST < 1 || c.orb_minThFAST > 254 || c.orb_iniThFAST > 254) { err = "FAST thresholds must be 1..254"; return SSM_E_INVAL; }
    if (c.width < 64 || c.height < 64 || c.width > 4000 || c.height > 4000) { err = "frame size must be 64..4000"; return SSM_E_INVAL; }
    if (!(c.orb_scale > 1.0f)) { err = "orb_scale must be > 1"; return SSM_E_INVAL; }
    g.nlevels = c.orb_levels; g.W = c.width; g.H = c.height; g.ini_th = c.orb_iniThFAST; g.min_th = c.orb_minThFAST;
    const double scaleFactor = (double)c.orb_scale;
    float sf[SSM_MAX_LEVELS], inv[SSM_MAX_LEVELS];
    sf[0] = 1.0f;
    for (int i = 1; i < g.nlevels; i++) sf[i] = (float)(sf[i-1] * scaleFactor);
    for (int i = 0; i < g.nlevels; i++) inv[i] = 1.0f / sf[i];
    int feat[SSM_MAX_LEVELS];
    {
        const float factor = (float)(1.0f / scaleFactor);
        float nd = c.orb_features * (1 - factor) / (1 - (float)pow((double)factor, (double)g.nlevels));
        int sum = 0;
        for (int l = 0; l < g.nlevels - 1; l++) { feat[l] = cv_round_f(nd); sum += feat[l]; nd *= factor; }
        feat[g.nlevels-1] = c.orb_features - sum > 0 ? c.orb_features - sum : 0;
    }
    {
        const int vmax = (int)floor(SSM_HALF_PATCH * sqrt(2.0) / 2 + 1), vmin = (int)ceil(SSM_HALF_PATCH * sqrt(2.0) / 2);
        const double hp2 = SSM_HALF_PATCH * SSM_HALF_PATCH;
        int um[SSM_HALF_PATCH + 2] = {0};
        for (int v = 0; v <= vmax; ++v) um[v] = (int)lrint(sqrt(hp2 - v * v));
        for (int v = SSM_HALF_PATCH, v0 = 0; v >= vmin; --v) { while (um[v0] == um[v0 + 1]) ++v0; um[v] = v0; ++v0; }
        for (int v = 0; v <= SSM_HALF_PATCH; v++) g.umax[v] = um[v];
    }
    int off = 0, cells = 0, cands = 0, sels = 0, tiles = 0, ftiles = 0, btiles = 0, bunits = 0, boff = 0;
    for (int l = 0; l < g.nlevels; l++) {
        LevelGeom& L = g.L[l];
        L.w = cv_round_f((float)c.width * inv[l]); L.h = cv_round_f((float)c.height * inv[l]);
        if (L.w < 2 * SSM_EDGE + 8 + 30 || L.h < 2 * SSM_EDGE + 8 + 30) { err = "pyramid level too small for the ORB border; lower orb_levels"; return SSM_E_INVAL; }
        L.stride = (L.w + 15) & ~15; L.img_off = off; off += L.stride * L.h; L.boff = boff; boff += L.stride * ((L.h + 7) & ~7);      /* rows 16-B aligned: wide loads/stores everywhere */
        L.minBX = SSM_EDGE - 3; L.minBY = SSM_EDGE - 3; L.maxBX = L.w - SSM_EDGE + 3; L.maxBY = L.h - SSM_EDGE + 3;
        const float width = (float)(L.maxBX - L.minBX), height = (float)(L.maxBY - L.minBY);
        L.nCols = (int)(width / 30.f); L.nRows = (int)(height / 30.f);
        L.wCell = (int)ceilf(width / L.nCols); L.hCell = (int)ceilf(height / L.nRows);
        if (L.wCell < 17 || L.hCell < 5) { err = "FAST cell too small"; return SSM_E_INVAL; }   /* <= 8x8 cells per 128x32 tile */
        L.mulW = (uint32_t)(((1ull << 32) + L.wCell - 1) / L.wCell); L.mulH = (uint32_t)(((1ull << 32) + L.hCell - 1) / L.hCell);
        L.cell_off = cells; cells += L.nCols * L.nRows;
        L.tiles_x = (L.w + 127) / 128; L.mulTX = (uint32_t)(((1ull << 32) + L.tiles_x - 1) / L.tiles_x); L.tile_off = tiles; tiles += L.tiles_x * ((L.h + 31) / 32);
        /* FAST reports nothing within SSM_EDGE of the border: its tile grid starts there (640x480, 8 levels: 233 tiles instead of 278) */
        L.ftiles_x = (L.w - 2 * SSM_EDGE + 127) / 128; L.fmulTX = (uint32_t)(((1ull << 32) + L.ftiles_x - 1) / L.ftiles_x); L.ftile_off = ftiles; ftiles += L.ftiles_x * ((L.h - 2 * SSM_EDGE + 31) / 32);
        L.bt_x = (L.stride + 127) / 128; L.bt_off = btiles; btiles += L.bt_x; L.bt_units_off = bunits; bunits += (L.stride + 31) / 32;
        if (L.nCols * L.nRows >= (1 << 17)) { err = "too many FAST cells"; return SSM_E_INVAL; }
        L.nfeat = feat[l];
        if (L.nfeat + 3 > SSM_MAX_NODES - 8) { err = "too many features per level for the LDS quad-tree (max 1013 per level)"; return SSM_E_INVAL; }
        L.cand_off = cands; L.cand_cap = ((L.w + 1) / 2 + L.nCols + 1) * ((L.h + 1) / 2 + L.nRows + 1); cands += L.cand_cap;
        if (L.cand_cap > 65535 * 16) { err = "level too large"; return SSM_E_INVAL; }
        L.sel_off = sels; L.sel_cap = L.nfeat + 3; sels += L.sel_cap;
        int nIni = (int)roundf((float)(L.maxBX - L.minBX) / (float)(L.maxBY - L.minBY)); if (nIni < 1) nIni = 1;
        L.nIni = nIni; L.hX = (float)(L.maxBX - L.minBX) / nIni;
        if (4 * nIni + 8 > SSM_MAX_NODES) { err = "aspect ratio too extreme"; return SSM_E_INVAL; }
        L.sf = sf[l];
    }
    g.bt_total = btiles; g.bt_units_total = bunits; g.blur_bytes = boff;
    g.pyr_bytes = off; g.tiles_total = tiles; g.ftiles_total = ftiles; g.cells_total = cells; g.cand_total = cands; g.sel_total = sels;
    g.cap = c.orb_features + 3 * g.nlevels;
    return SSM_OK;
}
void resize_tables(int ssize, int dsize, std::vector<int32_t>& ofs, std::vector<int16_t>& coef)
{
    ofs.resize(dsize); coef.resize(2 * dsize);
    const double inv_scale = (double)dsize / ssize, scale = 1.0 / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
        ofs[d] = s;
        coef[2*d] = (int16_t)cv_round_f((1.f - f) * 2048.f); coef[2*d+1] = (int16_t)cv_round_f(f * 2048.f);
    }
}

// ---------------------------------------------------------------- helpers
int ensure_scratch(ssm_ctx* c, size_t bytes)
{
    if (bytes <= c->scratch_bytes) return SSM_OK;
    if (c->d_scratch) { hipStreamSynchronize(c->stream); hipFree(c->d_scratch); c->d_scratch = nullptr; c->scratch_bytes = 0; }
    uint8_t* p; int r = dalloc(c, &p, bytes); if (r) return r;
    c->d_scratch = p; c->scratch_bytes = bytes; return SSM_OK;
}
int ensure_pinned(ssm_ctx* c, size_t bytes)
{
    if (bytes <= c->pinned_bytes) return SSM_OK;
    if (c->h_pinned) { hipStreamSynchronize(c->stream); hipHostFree(c->h_pinned); c->h_pinned = nullptr; c->pinned_bytes = 0; }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { c->err = "hipHostMalloc failed"; return SSM_E_HIP; }
    c->h_pinned = (uint8_t*)p; c->pinned_bytes = bytes; return SSM_OK;
}
int ensure_scratch2(ssm_ctx* c, size_t bytes)
{
    if (bytes <= c->scratch2_bytes) return SSM_OK;
    if (c->d_scratch2) { hipStreamSynchronize(c->stream); hipFree(c->d_scratch2); c->d_scratch2 = nullptr; c->scratch2_bytes = 0; }
    uint8_t* p; int r = dalloc(c, &p, bytes); if (r) return r;
    c->d_scratch2 = p; c->scratch2_bytes = bytes; return SSM_OK;
}
void prof_begin(ssm_ctx* c, const char* name)
{
    if (!c->profiling) return;
    auto get = [&]() { if (c->pool_used == c->pool.size()) { hipEvent_t e; hipEventCreate(&e); c->pool.push_back(e); } return c->pool[c->pool_used++]; };
    StageRec r; r.name = name; r.a = get(); r.b = get();
    hipEventRecord(r.a, c->stream);
    c->recs.push_back(r);
}
void prof_end(ssm_ctx* c) { if (c->profiling) hipEventRecord(c->recs.back().b, c->stream); }

// ORB scratch overflow (d_status) is checked after every ORB entry point; the voxel-table-full flag (counters[1]) belongs to the MAP entry points
// (ssm_sync after ssm_seq_process, ssm_map_*): it is reported once, so that one overflowing call does not fail every later call on the
// context (the map then lacks the dropped points: ssm_map_clear / a larger voxel_capacity_log2 is the remedy the message names)
int check_device_flags(ssm_ctx* c, bool with_map)
{
    int32_t st = 0, cnt[2] = {0, 0};
    HIPCHK(c, hipMemcpy(&st, c->d_status, 4, hipMemcpyDeviceToHost));
    if (st) { hipMemset(c->d_status, 0, 4); FAIL(c, SSM_E_CAPACITY, "ORB scratch capacity exceeded (status " + std::to_string(st) + ")"); }
    if (c->stereo) {
        int32_t ov = 0;
        HIPCHK(c, hipMemcpy(&ov, c->stereo->overflow, 4, hipMemcpyDeviceToHost));
        if (ov) { hipMemset(c->stereo->overflow, 0, 4); FAIL(c, SSM_E_CAPACITY, "goodFeaturesToTrack: more corner candidates than the buffer holds (w*h/4 + 1024)"); }
        if (c->stereo->sg_fail) { const int r = sgbm_recover(c); if (r) return r; }
    }
    if (with_map) {
        { const int r = map_settle(c, c->stream, 0); if (r) return r; }
        HIPCHK(c, hipMemcpy(cnt, c->map.counters, 8, hipMemcpyDeviceToHost));
        if (cnt[1]) {
            // bit 1 (skipped points: a defined contract, DESIGN.md "voxel key range") is reported once and cleared.  Bit 0 (table full: points were DROPPED,
            // the map is incomplete) stays set on the device until ssm_map_clear, so that ssm_map_size / ssm_map_export* / ssm_voxel_allgather keep
            // refusing the incomplete map (and every rank of an all-gather sees it); this function reports it once per fill.
            if (cnt[1] & 2) { const int32_t keep = cnt[1] & 1; hipMemcpy(c->map.counters + 1, &keep, 4, hipMemcpyHostToDevice); }
            if ((cnt[1] & 1) && !c->map_full_reported) {
                c->map_full_reported = true;
                FAIL(c, SSM_E_CAPACITY, "voxel map: contributions were dropped (table and overflow list full between two growth checks): ssm_map_clear and start from a larger voxel_capacity_log2");
            }
            if (cnt[1] & 2) FAIL(c, SSM_E_VOXEL_RANGE, "points with a non-finite coordinate or a voxel index outside (-2^20, 2^20) were skipped (leaf too small for the extent, or a bad pose)");
        }
    }
    return SSM_OK;
}

// ---------------------------------------------------------------- lifecycle
extern "C" void ssm_config_default(ssm_config* c)
{
    memset(c, 0, sizeof(*c));
    c->width = 640; c->height = 480;
    c->orb_features = 2000; c->orb_scale = 1.2f; c->orb_levels = 8; c->orb_iniThFAST = 20; c->orb_minThFAST = 7;   // parameters.txt:66-71
    c->knn_match_ratio = 0.8; c->tracker_ref_frames = 5;                                                            // :72,:81
    c->mapper_resolution = 0.1; c->mapper_max_distance = 40;                                                        // :97-98
    c->camera.cx = 318.6; c->camera.cy = 255.3; c->camera.fx = 517.3; c->camera.fy = 516.5; c->camera.scale = 1000.0;
    c->max_batch = 16; c->voxel_capacity_log2 = 20; c->brief_pattern = nullptr;
    c->voxel_max_capacity_log2 = 28; c->sgbm_form = 0; c->sgbm_streams = 0; c->stereo_batch = 0;
}
extern "C" const char* ssm_version(void) { return "ssm_hip 0.1 (gfx950)"; }
extern "C" const char* ssm_last_error(const ssm_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

static int ctx_init(ssm_ctx* c)
{
    const ssm_config& cfg = c->cfg; const OrbGeom& g = c->g; const int B = c->B, W = g.W, H = g.H;
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    DALLOC(c, c->d_pattern, 1024);
    HIPCHK(c, hipMemcpy(c->d_pattern, cfg.brief_pattern ? cfg.brief_pattern : k_default_pattern, 1024, hipMemcpyHostToDevice));
    {   float pf[1024]; const int8_t* src = cfg.brief_pattern ? cfg.brief_pattern : k_default_pattern;
        for (int i = 0; i < 1024; i++) pf[i] = (float)src[i];
        { const char* e = getenv("SSM_BLUR_VARIANT"); c->blur_mfma = !(e && atoi(e) == 0); }
        { std::vector<uint8_t> bt(blur_mfma_table_bytes(c->g)); blur_mfma_tables(c->g, bt.data());
          uint8_t* dbt; DALLOC(c, dbt, bt.size()); c->d_blur_tab = dbt;
          HIPCHK(c, hipMemcpy(c->d_blur_tab, bt.data(), bt.size(), hipMemcpyHostToDevice)); }
        DALLOC(c, c->d_pattern_f, 1024);
        HIPCHK(c, hipMemcpy(c->d_pattern_f, pf, sizeof(pf), hipMemcpyHostToDevice)); }
    for (int l = 1; l < g.nlevels; l++) {
        std::vector<int32_t> xo, yo; std::vector<int16_t> xa, ya;
        resize_tables(g.L[l-1].w, g.L[l].w, xo, xa); resize_tables(g.L[l-1].h, g.L[l].h, yo, ya);
        while (yo.size() & 3) { yo.push_back(yo.back()); ya.push_back(ya[ya.size() - 2]); ya.push_back(ya[ya.size() - 2]); }   // resize4_kernel reads the y tables four rows at a time
        DALLOC(c, c->d_xofs[l], xo.size()); DALLOC(c, c->d_xa[l], xa.size()); DALLOC(c, c->d_yofs[l], yo.size()); DALLOC(c, c->d_ya[l], ya.size());
        HIPCHK(c, hipMemcpy(c->d_xofs[l], xo.data(), xo.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_xa[l], xa.data(), xa.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_yofs[l], yo.data(), yo.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_ya[l], ya.data(), ya.size() * 2, hipMemcpyHostToDevice));
        // per 4-pixel group: the (a0, a1) pairs, the byte offset of the first pixel's left neighbour and each pixel's offset from it; the streaming
        // kernel takes the group's source bytes with one 8-byte load per row, so every offset + 1 must lie inside those 8 bytes
        const int groups = g.L[l].stride / 4, dw = g.L[l].w;
        std::vector<uint32_t> xg((size_t)groups * 8, 0u); bool fits = true;
        for (int q = 0; q < groups; q++) {
            uint32_t* e = &xg[(size_t)q * 8];
            const int x0 = 4 * q;
            if (x0 >= dw) continue;                                               // padding group: coefficients 0 -> zeros, window at 0
            const int base = xo[x0];
            e[4] = (uint32_t)base;
            for (int k = 0; k < 4 && x0 + k < dw; k++) {
                const int off = xo[x0 + k] - base;
                if (off < 0 || off > 6) fits = false;
                e[k] = (uint32_t)(uint16_t)xa[2 * (x0 + k)] | ((uint32_t)(uint16_t)xa[2 * (x0 + k) + 1] << 16);
                e[5] |= (uint32_t)(off & 15) << (4 * k);
            }
        }
        if (fits) {
            uint32_t* d = nullptr; DALLOC(c, d, xg.size());
            HIPCHK(c, hipMemcpy(d, xg.data(), xg.size() * 4, hipMemcpyHostToDevice));
            c->d_xgrp[l] = d;
        }
    }
    // d_pyr + 16: resize4_kernel's 8-byte windows may end past the last row
    DALLOC(c, c->d_pyr, (size_t)B * g.pyr_bytes + 16); DALLOC(c, c->d_blur, (size_t)B * g.blur_bytes);
    // the per-level candidate counters and the cell maxima (+ k_fast's retry list) share ONE allocation, counters first: k_fast zeroes both with one fill
    DALLOC(c, c->d_ncand, k_fast_ncand_pad(B, g) + k_fast_cellmax_ints(B, g)); c->d_cellmax = c->d_ncand + k_fast_ncand_pad(B, g);
    DALLOC(c, c->d_cand, (size_t)B * g.cand_total); DALLOC(c, c->d_nodeof, (size_t)B * g.cand_total);
    DALLOC(c, c->d_sel, (size_t)B * g.sel_total); DALLOC(c, c->d_nsel, (size_t)B * g.nlevels);
    DALLOC(c, c->d_kpaux, (size_t)B * g.sel_total * 2);          // KpAux + KpRec per slot
    DALLOC(c, c->d_status, 1); HIPCHK(c, hipMemset(c->d_status, 0, 4));
    const int chunks = backproject_chunks(W, H);
    DALLOC(c, c->d_mask, (size_t)B * W * H); DALLOC(c, c->d_chunk_cnt, (size_t)B * chunks); DALLOC(c, c->d_chunk_off, (size_t)B * chunks);
    DALLOC(c, c->d_total, 2); DALLOC(c, c->d_points, (size_t)B * W * H);
    DALLOC(c, c->d_in_img, (size_t)W * H * 3); DALLOC(c, c->d_in_sem, (size_t)W * H * 3); DALLOC(c, c->d_in_depth, (size_t)W * H); DALLOC(c, c->d_in_pose, 16);
    DALLOC(c, c->map.ovf, VOX_OVF_RECORDS); c->map.ovf_cap = VOX_OVF_RECORDS;
    { void* hp = nullptr; HIPCHK(c, hipHostMalloc(&hp, 64, hipHostMallocDefault)); c->h_map_snap = (int32_t*)hp; memset(hp, 0, 64); }
    for (int k = 0; k < 2; k++) HIPCHK(c, hipEventCreateWithFlags(&c->map_snap_ev[k], hipEventDisableTiming));
    int r = table_alloc(c, c->map, cfg.voxel_capacity_log2); if (r) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SSM_OK;
}
extern "C" int ssm_create(int device, const ssm_config* cfg, ssm_ctx** out)
{
    if (!cfg || !out) { g_create_err = "null argument"; return SSM_E_INVAL; }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { g_create_err = std::string("no HIP device: ") + hipGetErrorString(e); return SSM_E_NODEVICE; }
    if (device < 0 || device >= ndev) { g_create_err = "device index out of range"; return SSM_E_INVAL; }
    if ((e = hipSetDevice(device)) != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return SSM_E_HIP; }
    ssm_ctx* c = new ssm_ctx();
    c->device = device; c->cfg = *cfg;
    // the stereo path's knobs come from the configuration (two contexts of a process may differ); the environment variables remain as overrides for ablation runs
    { int b = cfg->stereo_batch > 0 ? cfg->stereo_batch : (cfg->max_batch > 0 ? cfg->max_batch : 1); c->stereo_B = b < 1 ? 1 : b > 128 ? 128 : b; }
    { const int v = cfg->sgbm_streams; if (v > 0) c->stereo_sgbm_streams = v > 3 ? 3 : v; }
    c->sgbm_form_cfg = cfg->sgbm_form;
    { const char* e = getenv("SSM_MAP_STREAM"); c->map_stream = e ? atoi(e) : 1; }
    { const char* e = getenv("SSM_MATCH_VARIANT"); c->match_mfma = !(e && atoi(e) == 0); }      // 0: the VALU matcher in the sequence path (A/B runs)
    c->B = cfg->max_batch > 0 ? cfg->max_batch : 1; c->R = cfg->tracker_ref_frames > 0 ? cfg->tracker_ref_frames : 1;
    int r = build_geometry(*cfg, c->g, c->err);
    if (!r && (cfg->voxel_capacity_log2 < 8 || cfg->voxel_capacity_log2 > 28)) { c->err = "voxel_capacity_log2 must be 8..28"; r = SSM_E_INVAL; }
    if (!r && cfg->voxel_max_capacity_log2 != 0 && (cfg->voxel_max_capacity_log2 < cfg->voxel_capacity_log2 || cfg->voxel_max_capacity_log2 > 28)) { c->err = "voxel_max_capacity_log2 must be voxel_capacity_log2..28 (0: 28)"; r = SSM_E_INVAL; }
    if (!r) c->vox_max_log2 = cfg->voxel_max_capacity_log2 ? cfg->voxel_max_capacity_log2 : 28;
    if (!r && !(cfg->mapper_resolution > 0)) { c->err = "mapper_resolution must be > 0"; r = SSM_E_INVAL; }
    if (!r && (cfg->sgbm_form < 0 || cfg->sgbm_form > 3 || cfg->sgbm_streams < 0 || cfg->sgbm_streams > 3 || cfg->stereo_batch < 0)) { c->err = "sgbm_form must be 0..3, sgbm_streams 0..3, stereo_batch >= 0"; r = SSM_E_INVAL; }
    if (!r && c->B > 16384) { c->err = "max_batch must be <= 16384"; r = SSM_E_INVAL; }
    if (!r) r = ctx_init(c);
    if (r) { g_create_err = c->err; ssm_destroy(c); return r; }
    c->cfg.brief_pattern = nullptr;
    *out = c;
    return SSM_OK;
}
extern "C" void ssm_destroy(ssm_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    void* ptrs[] = { c->d_pattern, c->d_pyr, c->d_blur, /* d_cellmax: inside d_ncand's allocation */ c->d_cand, c->d_nodeof, c->d_ncand, c->d_sel, c->d_nsel, c->d_status, c->d_mask,
                     c->d_chunk_cnt, c->d_chunk_off, c->d_total, c->d_points, c->d_in_img, c->d_in_sem, c->d_in_depth, c->d_in_pose,
                     c->d_scratch, c->d_scratch2, c->d_kps, c->d_desc_all, c->d_nkp_all, c->d_pos3d, c->d_matches, c->d_nmatch, c->d_match_pend, c->d_npoints,
                     c->d_hist_tmp, c->map.tab, c->tmp.tab, c->d_kpaux, c->d_pattern_f, c->d_exp_q, c->d_exp_t, c->d_knn, c->d_blur_tab, c->d_vmap, c->d_vcat };
    for (void* p : ptrs) if (p) hipFree(p);
    if (c->map.ovf) hipFree(c->map.ovf);
    if (c->d_pnp_xchg) hipFree(c->d_pnp_xchg);
    if (c->h_map_snap) hipHostFree(c->h_map_snap);
    for (int k = 0; k < 2; k++) if (c->map_snap_ev[k]) hipEventDestroy(c->map_snap_ev[k]);
    { void* ap[] = { c->alt.pyr, c->alt.blur, c->alt.cand, c->alt.nodeof, c->alt.ncand, c->alt.sel, c->alt.nsel, c->alt.mask, c->alt.kpaux };
      for (void* p : ap) if (p) hipFree(p); }
    for (int i = 0; i < 3; i++) if (c->ev_orb[i]) hipEventDestroy(c->ev_orb[i]);
    { void* ap2[] = { c->alt2.pyr, c->alt2.blur, c->alt2.cand, c->alt2.nodeof, c->alt2.ncand, c->alt2.sel, c->alt2.nsel, c->alt2.mask, c->alt2.kpaux };
      for (void* p : ap2) if (p) hipFree(p); }
    if (c->stream4) hipStreamDestroy(c->stream4);
    if (c->ev_join4) hipEventDestroy(c->ev_join4);
    for (int l = 0; l < SSM_MAX_LEVELS; l++) { if (c->d_xofs[l]) hipFree(c->d_xofs[l]); if (c->d_xa[l]) hipFree(c->d_xa[l]); if (c->d_yofs[l]) hipFree(c->d_yofs[l]); if (c->d_ya[l]) hipFree(c->d_ya[l]); if (c->d_xgrp[l]) hipFree(c->d_xgrp[l]); }
    if (c->seg) {
        SegNetState* g = c->seg;
        void* sp[] = { g->actA, g->actB, g->labels, g->d_sem_gen, g->pre_xofs, g->pre_yofs, g->post_xofs, g->post_yofs, g->pre_xa, g->pre_ya, g->post_xa, g->post_ya,
                       g->code[0], g->code[1], g->code[2], g->code[3], g->code[4] };
        for (void* p : sp) if (p) hipFree(p);
        for (int l = 0; l < SEG_LAYERS; l++) { if (g->ww[l]) hipFree(g->ww[l]); if (g->w[l]) hipFree(g->w[l]); if (g->scale[l]) hipFree(g->scale[l]); if (g->shift[l]) hipFree(g->shift[l]); }
        delete g;
    }
    if (c->stereo) { stereo_free(c->stereo); delete c->stereo; }
    if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
    if (c->d_comm_counts) hipFree(c->d_comm_counts);
    for (hipEvent_t e : c->pool) hipEventDestroy(e);
    for (hipStream_t st : {c->stream, c->stream2, c->stream3, c->stream4}) if (st) { k_sgbm_release_stream(st); k_segnet_release_stream(st); }
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    for (auto& sl : c->cloud_slabs) if (sl.d) hipFree(sl.d);
    if (c->h_ring) hipHostFree(c->h_ring);
    if (c->d_ring) hipFree(c->d_ring);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->stream3) hipStreamDestroy(c->stream3);
    if (c->ev_join3) hipEventDestroy(c->ev_join3);
    if (c->map.skip) hipFree(c->map.skip); if (c->d_redo) hipFree(c->d_redo);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    delete c;
}
void ssm_internal_get_config(const ssm_ctx* c, ssm_config* out) { *out = c->cfg; }
int ssm_internal_get_device(const ssm_ctx* c) { return c->device; }
extern "C" int ssm_orb_capacity(const ssm_ctx* c) { return c ? c->g.cap : 0; }
extern "C" void* ssm_stream(ssm_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int ssm_sync(ssm_ctx* c)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    hipSetDevice(c->device);
    c->err.clear();                                               // (after a successful ssm_sync ssm_last_error is empty, or the note of a repeated SGBM sweep)
    { int r = wait_pending(c); if (r) return r; }                 // asynchronous per-frame calls still in flight are completed (their results delivered) too
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->map_tail = nullptr;                                        // (every stream of the context is idle: the context stream joined them)
    return check_device_flags(c, true);
}
extern "C" int ssm_set_profiling(ssm_ctx* c, int on) { if (!c) return SSM_E_INVAL; std::lock_guard<std::mutex> lk(c->mu); c->profiling = on != 0; c->serialize = on == 2; return SSM_OK; }
extern "C" int ssm_get_stage_times(ssm_ctx* c, const char** names, float* ms, int* launches, int cap, int* n_out)
{
    if (!c || !n_out) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stage_names.clear(); c->stage_ms.clear(); c->stage_launches.clear();
    for (const StageRec& r : c->recs) {
        float t = 0.f; hipEventElapsedTime(&t, r.a, r.b);
        size_t i = 0;
        for (; i < c->stage_names.size(); i++) if (c->stage_names[i] == r.name) break;
        if (i == c->stage_names.size()) { c->stage_names.push_back(r.name); c->stage_ms.push_back(0.f); c->stage_launches.push_back(0); }
        c->stage_ms[i] += t; c->stage_launches[i] += 1;
    }
    const int n = (int)c->stage_names.size();
    *n_out = n;
    for (int i = 0; i < n && i < cap; i++) { if (names) names[i] = c->stage_names[i].c_str(); if (ms) ms[i] = c->stage_ms[i]; if (launches) launches[i] = c->stage_launches[i]; }
    return SSM_OK;
}

// second workspace for the two-chain mode of ssm_seq_process (same sizes as ctx_init's)
// The side streams of ssm_seq_process (stream2: second chain / SegNet + map side; stream3: map stage; stream4: third chain) are created at first use, not
// with the context: HIP spreads streams over a few hardware queues in creation order, and a context that only serves per-frame calls (the stereo bench
// runs eight of them) should take ONE slot of that rotation -- with four streams per context every context's main stream landed on the same queue
// (configs[3]: 233 instead of 346-386 frame pairs/s).
int ensure_side_streams(ssm_ctx* c)
{
    if (c->side_ready) return SSM_OK;
    // each handle is created only if it is still missing: a call that failed half-way leaves side_ready false and the next call resumes
    // (Round 5 measured a static split of the machine -- the map stage's stream confined to N compute units, the chains' streams to the rest, hipExtStreamCreateWithCUMask --
    // against the hardware's block-by-block arbitration: -9 %, profiles/r05_cu_split.md.  The switches are gone.)
    auto mk_stream = [&](hipStream_t* st, int = 0) -> hipError_t { return *st ? hipSuccess : hipStreamCreateWithFlags(st, hipStreamNonBlocking); };
    auto mk_event = [&](hipEvent_t* ev) -> hipError_t { return *ev ? hipSuccess : hipEventCreateWithFlags(ev, hipEventDisableTiming); };
    HIPCHK(c, mk_stream(&c->stream2, 2));
    HIPCHK(c, mk_stream(&c->stream3, 1)); HIPCHK(c, mk_event(&c->ev_join3));
    HIPCHK(c, mk_event(&c->ev_fork)); HIPCHK(c, mk_event(&c->ev_join));
    for (int i = 0; i < 3; i++) HIPCHK(c, mk_event(&c->ev_orb[i]));
    HIPCHK(c, mk_stream(&c->stream4, 2)); HIPCHK(c, mk_event(&c->ev_join4));
    c->side_ready = true;
    return SSM_OK;
}
static int ensure_alt_ws(ssm_ctx* c, ssm_ctx::AltWork& a)
{
    if (a.ready) return SSM_OK;
    const OrbGeom& g = c->g; const int B = c->B;
    DALLOC(c, a.pyr, (size_t)B * g.pyr_bytes + 16); DALLOC(c, a.blur, (size_t)B * g.blur_bytes);
    DALLOC(c, a.ncand, k_fast_ncand_pad(B, g) + k_fast_cellmax_ints(B, g)); a.cellmax = a.ncand + k_fast_ncand_pad(B, g);
    DALLOC(c, a.cand, (size_t)B * g.cand_total); DALLOC(c, a.nodeof, (size_t)B * g.cand_total);
    DALLOC(c, a.sel, (size_t)B * g.sel_total); DALLOC(c, a.nsel, (size_t)B * g.nlevels);
    DALLOC(c, a.mask, (size_t)B * g.W * g.H); DALLOC(c, a.kpaux, (size_t)B * g.sel_total * 2);
    a.ready = true;
    return SSM_OK;
}
static int ensure_alt(ssm_ctx* c)
{
    int r = ensure_alt_ws(c, c->alt); if (r) return r;
    if (c->nchains >= 3) { r = ensure_alt_ws(c, c->alt2); if (r) return r; }
    return SSM_OK;
}
struct ChainSwap {                    // chains 1, 2 of ssm_seq_process: the helpers use c->stream and the c->d_* workspace; point both at that chain's set
    ssm_ctx* c; int chain;
    void swap_all() { ssm_ctx::AltWork& a = chain == 1 ? c->alt : c->alt2;
                      std::swap(c->stream, chain == 1 ? c->stream2 : c->stream4); std::swap(c->d_pyr, a.pyr); std::swap(c->d_blur, a.blur); std::swap(c->d_cellmax, a.cellmax);
                      std::swap(c->d_cand, a.cand); std::swap(c->d_nodeof, a.nodeof); std::swap(c->d_ncand, a.ncand); std::swap(c->d_sel, a.sel);
                      std::swap(c->d_nsel, a.nsel); std::swap(c->d_mask, a.mask); std::swap(c->d_kpaux, a.kpaux); }
    ChainSwap(ssm_ctx* c_, int chain_) : c(c_), chain(chain_) { if (chain) swap_all(); }
    ~ChainSwap() { if (chain) swap_all(); }
};
// ---------------------------------------------------------------- the ORB front end for nb frames already on the device
static int run_orb(ssm_ctx* c, const uint8_t* d_img, int channels, const uint16_t* d_depth, int nb,
                   ssm_keypoint* kps, uint8_t* desc, float* pos3d, int32_t* nkp)
{
    const OrbGeom& g = c->g; hipStream_t s = c->stream;
    prof_begin(c, "gray");      HIPCHK(c, k_gray(d_img, channels, nb, g, c->d_pyr, s)); prof_end(c);
    prof_begin(c, "pyramid");   HIPCHK(c, k_pyramid(nb, g, c->d_pyr, c->d_xofs, c->d_xa, c->d_yofs, c->d_ya, c->d_xgrp, s)); prof_end(c);
    prof_begin(c, "fast");      HIPCHK(c, k_fast(nb, g, c->d_pyr, c->d_cand, c->d_ncand, c->d_cellmax, s)); prof_end(c);
    prof_begin(c, "octree");    HIPCHK(c, k_octree(nb, g, c->d_cand, c->d_ncand, c->d_cellmax, c->d_nodeof, c->d_sel, c->d_nsel, c->d_status, s)); prof_end(c);
    prof_begin(c, "blur");      HIPCHK(c, c->blur_mfma ? k_blur_mfma(nb, g, c->d_pyr, c->d_blur, c->d_blur_tab, s) : k_blur(nb, g, c->d_pyr, c->d_blur, s)); prof_end(c);
    prof_begin(c, "describe");  HIPCHK(c, k_describe(nb, g, c->d_pyr, c->d_blur, c->d_sel, c->d_nsel, c->d_pattern_f, d_depth, c->cfg.camera, c->d_kpaux, kps, desc, pos3d, nkp, s)); prof_end(c);
    return SSM_OK;
}

// ---- the per-frame calls of the reference's unchanged loop (Tracker::trackRefFrame: detectFeatures, then match against every reference frame,
// /root/reference/src/track.cpp:140-163).  Each call: inputs into the pinned ring (one memcpy per image), ONE host-to-device copy per image, the kernels, ONE
// device-to-host copy of a result block that carries the count with the payload (capacity-sized: no round trip to learn the count first), and a finisher that
// ssm_wait runs after the stream has drained.  The synchronous forms are the asynchronous ones + ssm_wait.
int wait_pending(ssm_ctx* c)
{
    if (c->pending.empty()) {
        // (an enqueue that failed behind its ring_take has advanced the offsets without registering a finisher: drain what may still read the ring, then rewind)
        if (c->h_ring_off || c->d_ring_off) { HIPCHK(c, hipStreamSynchronize(c->stream)); c->h_ring_off = 0; c->d_ring_off = 0; }
        return SSM_OK;
    }
    int rc = SSM_OK;
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { c->err = std::string("hipStreamSynchronize: ") + hipGetErrorString(e); rc = SSM_E_HIP; }
    std::vector<std::function<int(ssm_ctx*)>> fins; fins.swap(c->pending);
    for (auto& f : fins) { if (rc == SSM_OK) { const int r = f(c); if (r != SSM_OK) rc = r; } }      // after a failure the later calls' outputs stay untouched
    c->h_ring_off = 0; c->d_ring_off = 0;
    return rc;
}
static int ring_take(ssm_ctx* c, size_t hbytes, size_t dbytes, uint8_t** hp, uint8_t** dp)
{
    hbytes = (hbytes + 255) & ~(size_t)255; dbytes = (dbytes + 255) & ~(size_t)255;
    if (c->h_ring_off + hbytes > c->ring_bytes || c->d_ring_off + dbytes > c->ring_bytes) {
        int r = wait_pending(c); if (r) return r;                                // out of room: finish what is in flight (its results are delivered now)
        const size_t need = hbytes > dbytes ? hbytes : dbytes;
        if (need > c->ring_bytes) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->h_ring) hipHostFree(c->h_ring); if (c->d_ring) hipFree(c->d_ring);
            c->h_ring = nullptr; c->d_ring = nullptr; c->ring_bytes = 0;
            const size_t nb = need * 4 > ((size_t)8 << 20) ? need * 4 : ((size_t)8 << 20);
            void* hp_ = nullptr;
            if (hipHostMalloc(&hp_, nb, hipHostMallocDefault) != hipSuccess) FAIL(c, SSM_E_HIP, "hipHostMalloc of the staging ring failed");
            c->h_ring = (uint8_t*)hp_;
            if (hipMalloc((void**)&c->d_ring, nb) != hipSuccess) { hipHostFree(c->h_ring); c->h_ring = nullptr; FAIL(c, SSM_E_HIP, "hipMalloc of the result ring failed"); }
            c->ring_bytes = nb;
        }
        if (c->h_ring_off + hbytes > c->ring_bytes || c->d_ring_off + dbytes > c->ring_bytes) FAIL(c, SSM_E_HIP, "staging ring: the request does not fit after the wait");
    }
    *hp = c->h_ring + c->h_ring_off; *dp = c->d_ring + c->d_ring_off;
    c->h_ring_off += hbytes; c->d_ring_off += dbytes;
    return SSM_OK;
}
// in_place: the caller's buffers outlive the device work (the synchronous form) -- page-locked inputs are then used where they are
static int orb_extract_enqueue(ssm_ctx* c, const uint8_t* img, int w, int h, int stride, int channels, const uint16_t* depth,
                               ssm_keypoint* kps, uint8_t* desc, float* pos3d, int cap, int* n_out, bool in_place = false)
{
    if (!img || !kps || !desc || !n_out) FAIL(c, SSM_E_INVAL, "null argument");
    if (w != c->g.W || h != c->g.H) FAIL(c, SSM_E_INVAL, "frame size differs from the context configuration");
    if (channels != 1 && channels != 3) FAIL(c, SSM_E_INVAL, "channels must be 1 or 3");
    if (stride < w * channels) FAIL(c, SSM_E_INVAL, "stride smaller than a row");
    const int ocap = c->g.cap;
    const size_t row = (size_t)w * channels, ib = row * h, db = depth ? (size_t)w * h * 2 : 0;
    // result block: [n, status, pad][keypoints][descriptors][positions]
    const size_t blk = 64 + (size_t)ocap * (sizeof(ssm_keypoint) + 32 + 12);
    uint8_t *hp, *dp;
    int r = ring_take(c, ib + db + 64 + blk, blk, &hp, &dp); if (r) return r;
    uint8_t* h_in = hp; uint8_t* h_out = hp + ((ib + db + 63) & ~(size_t)63);
    // round 6: the call is a latency chain (20 launches of 4 - 40 us for one frame + the staging copies).  The depth image is read by the LAST kernel only, and only at
    // the <= cap keypoints: it is staged into the pinned ring while gray .. quad-tree run (after their launches, before the describe launches) and the kernel reads it
    // there, through the ring's device mapping -- no 0.6 MB upload, no staging time in front of the first kernel.  (Measured and dropped: the blur on a side stream
    // beside FAST + the quad-tree -- two cross-stream events cost more than the 17 us they hide: 177 -> 223 us from first to last kernel.)
    const bool img_direct = in_place && (size_t)stride == row && host_is_pinned(img), depth_direct = in_place && depth && host_is_pinned(depth);
    if (img_direct) HIPCHK(c, hipMemcpyAsync(c->d_in_img, img, ib, hipMemcpyHostToDevice, c->stream));       // page-locked input (ssm_host_alloc): no staging pass
    else {
        if ((size_t)stride == row) memcpy(h_in, img, ib);
        else for (int y = 0; y < h; y++) memcpy(h_in + (size_t)y * row, img + (size_t)y * stride, row);
        HIPCHK(c, hipMemcpyAsync(c->d_in_img, h_in, ib, hipMemcpyHostToDevice, c->stream));
    }
    int32_t* dn = reinterpret_cast<int32_t*>(dp);
    ssm_keypoint* dk = reinterpret_cast<ssm_keypoint*>(dp + 64);
    uint8_t* dd = reinterpret_cast<uint8_t*>(dk + ocap);
    float* dps = reinterpret_cast<float*>(dd + (size_t)ocap * 32);
    {
        const OrbGeom& g = c->g; hipStream_t s = c->stream;
        prof_begin(c, "gray");      HIPCHK(c, k_gray(c->d_in_img, channels, 1, g, c->d_pyr, s)); prof_end(c);
        prof_begin(c, "pyramid");   HIPCHK(c, k_pyramid(1, g, c->d_pyr, c->d_xofs, c->d_xa, c->d_yofs, c->d_ya, c->d_xgrp, s)); prof_end(c);
        prof_begin(c, "fast");      HIPCHK(c, k_fast(1, g, c->d_pyr, c->d_cand, c->d_ncand, c->d_cellmax, s)); prof_end(c);
        prof_begin(c, "octree");    HIPCHK(c, k_octree(1, g, c->d_cand, c->d_ncand, c->d_cellmax, c->d_nodeof, c->d_sel, c->d_nsel, c->d_status, s)); prof_end(c);
        prof_begin(c, "blur");      HIPCHK(c, c->blur_mfma ? k_blur_mfma(1, g, c->d_pyr, c->d_blur, c->d_blur_tab, s) : k_blur(1, g, c->d_pyr, c->d_blur, s)); prof_end(c);
        const uint16_t* d_depth = nullptr;
        if (depth) {
            if (!depth_direct) memcpy(h_in + ib, depth, db);       // (the device is busy with the launches above meanwhile)
            void* mapped = nullptr;
            HIPCHK(c, hipHostGetDevicePointer(&mapped, depth_direct ? const_cast<uint16_t*>(depth) : reinterpret_cast<uint16_t*>(h_in + ib), 0));
            d_depth = reinterpret_cast<const uint16_t*>(mapped);
        }
        prof_begin(c, "describe");  HIPCHK(c, k_describe(1, g, c->d_pyr, c->d_blur, c->d_sel, c->d_nsel, c->d_pattern_f, d_depth, c->cfg.camera, c->d_kpaux, dk, dd, dps, dn, s)); prof_end(c);
    }
    HIPCHK(c, hipMemcpyAsync(dn + 1, c->d_status, 4, hipMemcpyDeviceToDevice, c->stream));          // the ORB scratch-overflow word travels in the block's header
    HIPCHK(c, hipMemcpyAsync(h_out, dp, blk, hipMemcpyDeviceToHost, c->stream));
    c->pending.push_back([=](ssm_ctx* cc) -> int {
        int32_t hdr[2]; memcpy(hdr, h_out, 8);
        if (hdr[1]) { hipMemset(cc->d_status, 0, 4); FAIL(cc, SSM_E_CAPACITY, "ORB scratch capacity exceeded (status " + std::to_string(hdr[1]) + ")"); }
        const int n = hdr[0];
        *n_out = n;
        if (n > cap) FAIL(cc, SSM_E_CAPACITY, "keypoint buffer too small (need " + std::to_string(n) + ")");
        memcpy(kps, h_out + 64, sizeof(ssm_keypoint) * (size_t)n);
        memcpy(desc, h_out + 64 + (size_t)ocap * sizeof(ssm_keypoint), (size_t)n * 32);
        if (pos3d) memcpy(pos3d, h_out + 64 + (size_t)ocap * (sizeof(ssm_keypoint) + 32), (size_t)n * 12);
        return SSM_OK;
    });
    return SSM_OK;
}
extern "C" int ssm_orb_extract_async(ssm_ctx* c, const uint8_t* img, int w, int h, int stride, int channels, const uint16_t* depth,
                                     ssm_keypoint* kps, uint8_t* desc, float* pos3d, int cap, int* n_out)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    return orb_extract_enqueue(c, img, w, h, stride, channels, depth, kps, desc, pos3d, cap, n_out);
}
extern "C" int ssm_orb_extract(ssm_ctx* c, const uint8_t* img, int w, int h, int stride, int channels, const uint16_t* depth,
                               ssm_keypoint* kps, uint8_t* desc, float* pos3d, int cap, int* n_out)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    int r = orb_extract_enqueue(c, img, w, h, stride, channels, depth, kps, desc, pos3d, cap, n_out, true); if (r) return r;
    return wait_pending(c);
}
extern "C" int ssm_wait(ssm_ctx* c)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    return wait_pending(c);
}

// ---------------------------------------------------------------- matcher, host pointers
static int match_host(ssm_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio, bool want_knn,
                      int32_t* idx, int32_t* dist, ssm_dmatch* out, int cap, int* n_out)
{
    if (nq < 0 || nt < 0 || (nq && !q) || (nt && !t)) FAIL(c, SSM_E_INVAL, "bad descriptor arguments");
    if (nt < 2) FAIL(c, SSM_E_TOO_FEW_TRAIN, "knnMatch(k=2) needs at least 2 train descriptors");
    if (nt > 65535) FAIL(c, SSM_E_INVAL, "at most 65535 train descriptors per call");
    if (nq == 0) { if (n_out) *n_out = 0; return SSM_OK; }
    if (c->match_mfma && !want_knn) {
        // the matrix-core matcher on a two-row "sequence" (row 0 = query set, row 1 = train set) through the rings: one upload, one result block
        const int capm = nq > nt ? nq : nt, capT = (capm + 31) & ~31;
        const size_t rowb = (size_t)capm * 32, expb = (size_t)2 * capT * SSM_MATCH_DESC_BYTES;
        const size_t inb = 2 * rowb + 16, outb = 64 + (size_t)nq * sizeof(ssm_dmatch);
        const size_t devb = ((inb + 255) & ~(size_t)255) + 2 * expb + (((size_t)capT * 8 + 255) & ~(size_t)255) + outb;
        uint8_t *hp, *dp;
        int r = ring_take(c, inb + 64 + outb, devb, &hp, &dp); if (r) return r;
        uint8_t* h_out = hp + ((inb + 63) & ~(size_t)63);
        memcpy(hp, q, (size_t)nq * 32); memcpy(hp + rowb, t, (size_t)nt * 32);
        const int32_t hn[2] = {nq, nt}; memcpy(hp + 2 * rowb, hn, 8);
        uint8_t* dd = dp; int32_t* dnk = reinterpret_cast<int32_t*>(dd + 2 * rowb);
        uint8_t* eq = dp + ((inb + 255) & ~(size_t)255); uint8_t* et = eq + expb;
        uint2* knn = reinterpret_cast<uint2*>(et + expb);
        uint8_t* dout = reinterpret_cast<uint8_t*>(knn) + (((size_t)capT * 8 + 255) & ~(size_t)255);
        int32_t* dn = reinterpret_cast<int32_t*>(dout); ssm_dmatch* dm = reinterpret_cast<ssm_dmatch*>(dout + 64);
        HIPCHK(c, hipMemcpyAsync(dd, hp, inb, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, k_match_expand(dd, dnk, 0, 2, capm, capT, eq, et, c->stream));
        HIPCHK(c, k_match_seq_mfma(eq, et, dnk, 0, 1, 1, 1, ratio, capm, capT, knn, dm, dn, c->stream));
        HIPCHK(c, hipMemcpyAsync(h_out, dout, outb, hipMemcpyDeviceToHost, c->stream));
        c->pending.push_back([=](ssm_ctx* cc) -> int {
            int32_t n; memcpy(&n, h_out, 4);
            *n_out = n;
            if (n > cap) FAIL(cc, SSM_E_CAPACITY, "match buffer too small (need " + std::to_string(n) + ")");
            memcpy(out, h_out + 64, sizeof(ssm_dmatch) * (size_t)n);
            return SSM_OK;
        });
        return SSM_OK;
    }
    if (c->match_mfma) {
        // the matrix-core matcher of the sequence path on a two-row "sequence": row 0 = the query set (reference frame), row 1 = the train set
        const int capm = nq > nt ? nq : nt, capT = (capm + 31) & ~31;
        const size_t rowb = (size_t)capm * 32, expb = (size_t)2 * capT * SSM_MATCH_DESC_BYTES;
        const size_t need = 2 * rowb + 16 + 2 * expb + (size_t)capT * 8 + (size_t)nq * sizeof(ssm_dmatch) + 64;
        int r = ensure_scratch(c, need); if (r) return r;
        uint8_t* dd = reinterpret_cast<uint8_t*>(c->d_scratch);
        int32_t* dnk = reinterpret_cast<int32_t*>(dd + 2 * rowb);
        uint8_t* eq = reinterpret_cast<uint8_t*>(dnk) + 16; uint8_t* et = eq + expb;
        uint2* knn = reinterpret_cast<uint2*>(et + expb);
        ssm_dmatch* dm = reinterpret_cast<ssm_dmatch*>(knn + capT); int32_t* dn = reinterpret_cast<int32_t*>(dm + nq);
        const int32_t hn[2] = {nq, nt};
        HIPCHK(c, hipMemcpyAsync(dd, q, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(dd + rowb, t, (size_t)nt * 32, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(dnk, hn, 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, k_match_expand(dd, dnk, 0, 2, capm, capT, eq, et, c->stream));
        HIPCHK(c, k_match_seq_mfma(eq, et, dnk, 0, 1, 1, 1, ratio, capm, capT, knn, dm, dn, c->stream));
        int n = 0;
        HIPCHK(c, hipMemcpyAsync(&n, dn, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (want_knn) {
            std::vector<uint2> hk((size_t)nq);
            HIPCHK(c, hipMemcpy(hk.data(), knn, (size_t)nq * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < nq; i++) { idx[2*i] = hk[i].x & 0xFFFF; idx[2*i+1] = hk[i].y & 0xFFFF; dist[2*i] = hk[i].x >> 16; dist[2*i+1] = hk[i].y >> 16; }
        } else {
            *n_out = n;
            if (n > cap) FAIL(c, SSM_E_CAPACITY, "match buffer too small (need " + std::to_string(n) + ")");
            HIPCHK(c, hipMemcpy(out, dm, sizeof(ssm_dmatch) * n, hipMemcpyDeviceToHost));
        }
        return SSM_OK;
    }
    const size_t need = (size_t)(nq + nt) * 32 + sizeof(MatchPair) + (size_t)nq * (16 + 16) + 64;
    int r = ensure_scratch(c, need); if (r) return r;
    uint8_t* dd = reinterpret_cast<uint8_t*>(c->d_scratch);
    ssm_dmatch* dm = reinterpret_cast<ssm_dmatch*>(dd + (size_t)(nq + nt) * 32);
    int32_t* di = reinterpret_cast<int32_t*>(dm + nq); int32_t* ds = di + 2 * (size_t)nq;
    MatchPair* dp = reinterpret_cast<MatchPair*>(ds + 2 * (size_t)nq); int32_t* dn = reinterpret_cast<int32_t*>(dp + 1);
    MatchPair p; p.qoff = 0; p.nq = nq; p.toff = nq; p.nt = nt; p.out_slot = 0;
    HIPCHK(c, hipMemcpyAsync(dd, q, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dd + (size_t)nq * 32, t, (size_t)nt * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dp, &p, sizeof(p), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, k_match_pairs(dd, dp, 1, ratio, nq, dm, dn, want_knn ? di : nullptr, want_knn ? ds : nullptr, c->stream));
    int n = 0;
    HIPCHK(c, hipMemcpyAsync(&n, dn, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (want_knn) {
        HIPCHK(c, hipMemcpy(idx, di, (size_t)nq * 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(dist, ds, (size_t)nq * 8, hipMemcpyDeviceToHost));
    } else {
        *n_out = n;
        if (n > cap) FAIL(c, SSM_E_CAPACITY, "match buffer too small (need " + std::to_string(n) + ")");
        HIPCHK(c, hipMemcpy(out, dm, sizeof(ssm_dmatch) * n, hipMemcpyDeviceToHost));
    }
    return SSM_OK;
}
extern "C" int ssm_hamming_knn2(ssm_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (nq > 0 && (!idx || !dist)) FAIL(c, SSM_E_INVAL, "null output");
    return match_host(c, q, nq, t, nt, c->cfg.knn_match_ratio, true, idx, dist, nullptr, 0, nullptr);
}
extern "C" int ssm_match(ssm_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio, ssm_dmatch* out, int cap, int* n_out)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!n_out || (cap > 0 && !out)) FAIL(c, SSM_E_INVAL, "null output");
    int r = match_host(c, q, nq, t, nt, ratio, false, nullptr, nullptr, out, cap, n_out); if (r) return r;
    return wait_pending(c);
}
// Tracker::trackRefFrame's loop `for (pFrame : refFrames) matches = orb.match(pFrame, currentFrame)` (/root/reference/src/track.cpp:150-152) as ONE call: the
// reference frames' descriptor sets and the current frame's are the rows of a short "sequence" (refs oldest first, the current frame last) for the sequence
// matcher -- one upload, one expansion, ONE matrix-core launch for all pairs, one result block; list i is exactly ssm_match(refs[i], cur).
static int match_refs_enqueue(ssm_ctx* c, const uint8_t* const* refs, const int* nrefs, int nref, const uint8_t* cur, int ncur, double ratio,
                              ssm_dmatch* const* outs, const int* caps, int* n_outs)
{
    if (nref < 0 || (nref && (!refs || !nrefs || !outs || !caps || !n_outs)) || ncur < 0 || (ncur && !cur)) FAIL(c, SSM_E_INVAL, "bad arguments");
    if (nref == 0) return SSM_OK;
    if (ncur < 2) FAIL(c, SSM_E_TOO_FEW_TRAIN, "knnMatch(k=2) needs at least 2 train descriptors");
    if (ncur > 65535) FAIL(c, SSM_E_INVAL, "at most 65535 train descriptors per call");
    int capm = ncur;
    for (int i = 0; i < nref; i++) { if (nrefs[i] < 0 || (nrefs[i] && !refs[i]) || (caps[i] > 0 && !outs[i])) FAIL(c, SSM_E_INVAL, "bad reference set"); if (nrefs[i] > capm) capm = nrefs[i]; }
    if (!c->match_mfma || nref > 16) {                                         // the VALU variant (SSM_MATCH_VARIANT=0): pair by pair through the same entry
        for (int i = 0; i < nref; i++) {
            if (nrefs[i] == 0) { n_outs[i] = 0; continue; }
            int r = match_host(c, refs[i], nrefs[i], cur, ncur, ratio, false, nullptr, nullptr, outs[i], caps[i], &n_outs[i]); if (r) return r;
        }
        return SSM_OK;
    }
    const int rows = nref + 1, capT = (capm + 31) & ~31;
    const size_t rowb = (size_t)capm * 32, inb = (size_t)rows * rowb + 128;
    const size_t expb = (size_t)rows * capT * SSM_MATCH_DESC_BYTES, knnb = ((size_t)nref * capT * 8 + 255) & ~(size_t)255;
    const size_t outb = 256 + (size_t)nref * capm * sizeof(ssm_dmatch);
    uint8_t *hp, *dp;
    int r = ring_take(c, inb + 64 + outb, ((inb + 255) & ~(size_t)255) + 2 * expb + knnb + outb, &hp, &dp); if (r) return r;
    uint8_t* h_out = hp + ((inb + 63) & ~(size_t)63);
    int32_t hn[32] = {0};
    for (int i = 0; i < nref; i++) { if (nrefs[i]) memcpy(hp + (size_t)i * rowb, refs[i], (size_t)nrefs[i] * 32); hn[i] = nrefs[i]; }
    memcpy(hp + (size_t)nref * rowb, cur, (size_t)ncur * 32); hn[nref] = ncur;
    memcpy(hp + (size_t)rows * rowb, hn, 128);
    uint8_t* dd = dp; int32_t* dnk = reinterpret_cast<int32_t*>(dd + (size_t)rows * rowb);
    uint8_t* eq = dp + ((inb + 255) & ~(size_t)255); uint8_t* et = eq + expb;
    uint8_t* knn = et + expb; uint8_t* dout = knn + knnb;
    int32_t* dn = reinterpret_cast<int32_t*>(dout); ssm_dmatch* dm = reinterpret_cast<ssm_dmatch*>(dout + 256);
    HIPCHK(c, hipMemcpyAsync(dd, hp, inb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, k_match_expand(dd, dnk, 0, rows, capm, capT, eq, et, c->stream));
    HIPCHK(c, k_match_seq_mfma(eq, et, dnk, 0, 1, nref, nref, ratio, capm, capT, knn, dm, dn, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_out, dout, outb, hipMemcpyDeviceToHost, c->stream));
    std::vector<ssm_dmatch*> vo(outs, outs + nref); std::vector<int> vc(caps, caps + nref);
    c->pending.push_back([=](ssm_ctx* cc) -> int {
        for (int i = 0; i < nref; i++) {
            int32_t n; memcpy(&n, h_out + 4 * (size_t)i, 4);
            if (n < 0) n = 0;
            n_outs[i] = n;
            if (n > vc[i]) FAIL(cc, SSM_E_CAPACITY, "match buffer too small (need " + std::to_string(n) + ")");
            memcpy(vo[i], h_out + 256 + (size_t)i * capm * sizeof(ssm_dmatch), sizeof(ssm_dmatch) * (size_t)n);
        }
        return SSM_OK;
    });
    return SSM_OK;
}
extern "C" int ssm_match_refs_async(ssm_ctx* c, const uint8_t* const* refs, const int* nrefs, int nref, const uint8_t* cur, int ncur, double ratio,
                                    ssm_dmatch* const* outs, const int* caps, int* n_outs)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    return match_refs_enqueue(c, refs, nrefs, nref, cur, ncur, ratio, outs, caps, n_outs);
}
extern "C" int ssm_match_refs(ssm_ctx* c, const uint8_t* const* refs, const int* nrefs, int nref, const uint8_t* cur, int ncur, double ratio,
                              ssm_dmatch* const* outs, const int* caps, int* n_outs)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    int r = match_refs_enqueue(c, refs, nrefs, nref, cur, ncur, ratio, outs, caps, n_outs); if (r) return r;
    return wait_pending(c);
}
extern "C" int ssm_match_async(ssm_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio, ssm_dmatch* out, int cap, int* n_out)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!n_out || (cap > 0 && !out)) FAIL(c, SSM_E_INVAL, "null output");
    return match_host(c, q, nq, t, nt, ratio, false, nullptr, nullptr, out, cap, n_out);       // the VALU variant (SSM_MATCH_VARIANT=0) completes inside the call
}

// ---------------------------------------------------------------- mapper front half, host pointers
extern "C" int ssm_moving_mask(ssm_ctx* c, const uint8_t* sem, int w, int h, int stride, uint8_t* mask)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!sem || !mask) FAIL(c, SSM_E_INVAL, "null argument");
    if (w != c->g.W || h != c->g.H) FAIL(c, SSM_E_INVAL, "frame size differs from the context configuration");
    if (stride < w * 3) FAIL(c, SSM_E_INVAL, "stride smaller than a row");
    HIPCHK(c, hipMemcpy2DAsync(c->d_in_sem, (size_t)w * 3, sem, stride, (size_t)w * 3, h, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, k_moving_mask(c->d_in_sem, 1, w, h, c->d_mask, c->stream));
    HIPCHK(c, hipMemcpyAsync(mask, c->d_mask, (size_t)w * h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SSM_OK;
}
extern "C" int ssm_backproject(ssm_ctx* c, const uint16_t* depth, const uint8_t* rgb, const uint8_t* sem, int w, int h,
                               const ssm_camera* cam, const double* T, double max_distance, ssm_point* out, int cap, int* n_out)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!depth || !rgb || !sem || !cam || !n_out || (cap > 0 && !out)) FAIL(c, SSM_E_INVAL, "null argument");
    if (w != c->g.W || h != c->g.H) FAIL(c, SSM_E_INVAL, "frame size differs from the context configuration");
    const size_t np = (size_t)w * h;
    HIPCHK(c, hipMemcpyAsync(c->d_in_depth, depth, np * 2, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_in_img, rgb, np * 3, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_in_sem, sem, np * 3, hipMemcpyHostToDevice, c->stream));
    if (T) HIPCHK(c, hipMemcpyAsync(c->d_in_pose, T, 128, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, k_moving_mask(c->d_in_sem, 1, w, h, c->d_mask, c->stream));
    HIPCHK(c, k_backproject(c->d_in_depth, c->d_in_img, c->d_in_sem, c->d_mask, T ? c->d_in_pose : nullptr, 1, w, h, *cam, max_distance,
                            c->d_chunk_cnt, c->d_chunk_off, reinterpret_cast<int32_t*>(c->d_total + 1), c->d_total, c->d_points, c->stream));
    int64_t total = 0;
    HIPCHK(c, hipMemcpyAsync(&total, c->d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n_out = (int)total;
    if (total > cap) FAIL(c, SSM_E_CAPACITY, "point buffer too small (need " + std::to_string(total) + ")");
    HIPCHK(c, hipMemcpy(out, c->d_points, sizeof(ssm_point) * (size_t)total, hipMemcpyDeviceToHost));
    return SSM_OK;
}

// ---------------------------------------------------------------- device-resident sequence path
static int ensure_seq(ssm_ctx* c, int n)
{
    if (n <= c->seq_cap) return SSM_OK;
    const OrbGeom& g = c->g; const int R = c->R;
    // keep the history rows across the re-allocation
    uint8_t* old_desc = c->d_desc_all; int32_t* old_nkp = c->d_nkp_all; const int old_prev = c->prev_n;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    void* olds[] = { c->d_kps, c->d_pos3d, c->d_matches, c->d_nmatch, c->d_npoints, c->d_match_pend, c->d_exp_q, c->d_exp_t, c->d_knn };
    for (void* p : olds) if (p) hipFree(p);
    c->d_kps = nullptr; c->d_pos3d = nullptr; c->d_matches = nullptr; c->d_nmatch = nullptr; c->d_npoints = nullptr; c->d_match_pend = nullptr;
    c->d_exp_q = nullptr; c->d_exp_t = nullptr; c->d_knn = nullptr;
    c->capT = (g.cap + 31) & ~31;
    if (c->match_mfma) {       // expanded rows are rebuilt from the bit descriptors at the start of every call (history) and after every ORB sub-batch
        DALLOC(c, c->d_exp_q, (size_t)(n + R) * c->capT * SSM_MATCH_DESC_BYTES); DALLOC(c, c->d_exp_t, (size_t)(n + R) * c->capT * SSM_MATCH_DESC_BYTES); DALLOC(c, c->d_knn, (size_t)n * R * c->capT * 8);
    }
    DALLOC(c, c->d_kps, (size_t)n * g.cap); DALLOC(c, c->d_pos3d, (size_t)n * g.cap * 3);
    DALLOC(c, c->d_matches, (size_t)n * R * g.cap); DALLOC(c, c->d_nmatch, (size_t)n * R); DALLOC(c, c->d_match_pend, (size_t)n * R); DALLOC(c, c->d_npoints, (size_t)n);
    uint8_t* nd; int32_t* nn;
    DALLOC(c, nd, (size_t)(n + R) * g.cap * 32); DALLOC(c, nn, (size_t)(n + R));
    if (!c->d_hist_tmp) DALLOC(c, c->d_hist_tmp, (size_t)R * g.cap * 32 + (size_t)R * 4);
    if (old_desc && old_prev >= 0) {
        HIPCHK(c, hipMemcpy(nd, old_desc, (size_t)(old_prev + R) * g.cap * 32, hipMemcpyDeviceToDevice));
        HIPCHK(c, hipMemcpy(nn, old_nkp, (size_t)(old_prev + R) * 4, hipMemcpyDeviceToDevice));
    }
    if (old_desc) hipFree(old_desc);
    if (old_nkp) hipFree(old_nkp);
    c->d_desc_all = nd; c->d_nkp_all = nn; c->seq_cap = n;
    return SSM_OK;
}
extern "C" int ssm_seq_process(ssm_ctx* c, const ssm_frames_dev* in, ssm_seq_out_dev* out)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!in || in->n < 0) FAIL(c, SSM_E_INVAL, "bad arguments");
    const int stages = in->stages ? in->stages : (SSM_STAGE_ORB | SSM_STAGE_MATCH | SSM_STAGE_MAP);
    if ((stages & (SSM_STAGE_ORB | SSM_STAGE_MATCH)) && !in->bgr) FAIL(c, SSM_E_INVAL, "bgr is required");
    if ((stages & SSM_STAGE_MAP) && (!in->depth || !in->bgr || (!in->sem_bgr && !(stages & SSM_STAGE_SEGNET)))) FAIL(c, SSM_E_INVAL, "bgr, depth and sem_bgr (or SSM_STAGE_SEGNET) are required for the map stage");
    const OrbGeom& g = c->g; const int R = c->R, n = in->n, W = g.W, H = g.H; hipStream_t s = c->stream;
    const size_t npix = (size_t)W * H;
    // fused map launches of an earlier call that nobody has looked at since (no ssm_sync / map read in between): their skipped blocks -- if any -- are run again NOW,
    // while the launch descriptors still point at that call's outputs (ensure_seq below may re-allocate the per-frame point counts)
    if (c->map_unexamined) { const int r0 = map_settle(c, c->map_tail ? c->map_tail : c->stream, 0); if (r0) return r0; }
    int r = ensure_seq(c, n > 0 ? n : 1); if (r) return r;
    c->recs.clear(); c->pool_used = 0;
    // history rows
    const size_t row = (size_t)g.cap * 32;
    if (in->continue_sequence && c->prev_n >= 0) {
        int32_t* tn = reinterpret_cast<int32_t*>(c->d_hist_tmp + (size_t)R * row);
        HIPCHK(c, hipMemcpyAsync(c->d_hist_tmp, c->d_desc_all + (size_t)c->prev_n * row, (size_t)R * row, hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(tn, c->d_nkp_all + c->prev_n, (size_t)R * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->d_desc_all, c->d_hist_tmp, (size_t)R * row, hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->d_nkp_all, tn, (size_t)R * 4, hipMemcpyDeviceToDevice, s));
    } else {
        HIPCHK(c, hipMemsetAsync(c->d_nkp_all, 0xFF, (size_t)R * 4, s));     // -1: no such reference frame
    }
    uint8_t* desc = c->d_desc_all + (size_t)R * row; int32_t* nkp = c->d_nkp_all + R;
    const bool mfma = c->match_mfma && (stages & SSM_STAGE_MATCH);
    if (mfma) HIPCHK(c, k_match_expand(c->d_desc_all, c->d_nkp_all, 0, R, g.cap, c->capT, c->d_exp_q, c->d_exp_t, s));      // the history rows
    // Two streams: the ORB -> match chain of a sub-batch and its (SegNet ->) map stage share no data, only the inputs, so the
    // map side runs on stream2.  The chain's latency-bound kernels (pyramid, octree, describe) then overlap VALU/MFMA-bound
    // map / SegNet work.  stream2 starts behind everything already queued on the context stream and is joined at the end.
    const bool side_work = (stages & (SSM_STAGE_MAP | SSM_STAGE_SEGNET)) != 0;
    // Two chains: without the SegNet stage (one activation workspace) and with two or more sub-batches, alternate sub-batches run
    // their whole ORB -> match -> map chain on the context stream and on stream2 with a workspace each, so that one chain's
    // latency-bound kernels (quad-tree, pyramid launches, block tails) overlap the other chain's VALU-bound ones.  The only
    // dependence between neighbours is the matcher's: the reference descriptors of sub-batch b - 1 (an event per chain).
    const bool two_chains = !c->serialize && !(stages & SSM_STAGE_SEGNET) && n > c->B && (stages & SSM_STAGE_ORB) && (W & 15) == 0;
    const bool side = side_work && !c->serialize && !two_chains;
    if (side || two_chains) { r = ensure_side_streams(c); if (r) return r; }
    if (two_chains) { r = ensure_alt(c); if (r) return r; }
    if (side || two_chains) { HIPCHK(c, hipEventRecord(c->ev_fork, c->stream)); HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0)); }
    const int nch = two_chains ? (c->nchains >= 3 && n > 2 * c->B ? 3 : 2) : 1;
    if (nch == 3) HIPCHK(c, hipStreamWaitEvent(c->stream4, c->ev_fork, 0));
    const bool map3 = two_chains && c->map_stream == 1;
    if (map3) HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_fork, 0));
    struct StreamSwap {               // the helpers below launch on c->stream; point it at stream2 for the side work
        ssm_ctx* c; bool on;
        StreamSwap(ssm_ctx* c_, bool on_) : c(c_), on(on_) { if (on) std::swap(c->stream, c->stream2); }
        ~StreamSwap() { if (on) std::swap(c->stream, c->stream2); }
    };
    int bi = 0;
    for (int f0 = 0; f0 < n; f0 += c->B, bi++) {
        const int nb = (n - f0 < c->B) ? n - f0 : c->B;
        const int chain = bi % nch;
        ChainSwap cs(c, chain);                                      // from here c->stream / c->d_* are this chain's
        auto front = [&]() -> int {                                       // ORB -> match of this sub-batch
            if (stages & SSM_STAGE_ORB) {
                r = run_orb(c, in->bgr + (size_t)f0 * npix * 3, 3, in->depth ? in->depth + (size_t)f0 * npix : nullptr, nb,
                            c->d_kps + (size_t)f0 * g.cap, desc + (size_t)f0 * row, c->d_pos3d + (size_t)f0 * g.cap * 3, nkp + f0);
                if (r) return r;
                if (mfma) { prof_begin(c, "match"); HIPCHK(c, k_match_expand(c->d_desc_all, c->d_nkp_all, R + f0, nb, g.cap, c->capT, c->d_exp_q, c->d_exp_t, c->stream)); prof_end(c); }
                if (two_chains) {
                    // The matcher of sub-batch bi reads the descriptor rows of the R preceding FRAMES, i.e. (max_batch < tracker_ref_frames) of several
                    // preceding sub-batches.  Every other chain's newest event is the ORB + expand of one of bi-1 .. bi-(nch-1); an older sub-batch sits on
                    // one of those streams (or on this one) in front of that record, so waiting on all of them orders the matcher behind every row it reads.
                    HIPCHK(c, hipEventRecord(c->ev_orb[chain], c->stream));
                    for (int k = 1; k < nch && k <= bi; k++) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_orb[(bi - k) % nch], 0));
                }
            }
            if (stages & SSM_STAGE_MATCH) {
                prof_begin(c, "match");
                if (mfma) {
                    if (!(stages & SSM_STAGE_ORB)) HIPCHK(c, k_match_expand(c->d_desc_all, c->d_nkp_all, R + f0, nb, g.cap, c->capT, c->d_exp_q, c->d_exp_t, c->stream));
                    HIPCHK(c, k_match_seq_mfma(c->d_exp_q, c->d_exp_t, c->d_nkp_all, f0, nb, R, R, c->cfg.knn_match_ratio, g.cap, c->capT, c->d_knn, c->d_matches, c->d_nmatch, c->stream));
                } else
                    HIPCHK(c, k_match_seq(c->d_desc_all, c->d_nkp_all, f0, nb, R, R, c->cfg.knn_match_ratio, g.cap, c->d_matches, c->d_nmatch, c->d_match_pend + (size_t)f0 * R, c->stream));
                prof_end(c);
            }
            return SSM_OK;
        };
        auto back = [&]() -> int {                                        // (SegNet ->) map of this sub-batch
            if (!side_work) return SSM_OK;
            StreamSwap sw(c, side);
            hipStream_t s = map3 ? c->stream3 : c->stream;                // = stream2 inside this scope (unless serialised)
            struct StreamSet { ssm_ctx* c; hipStream_t keep; StreamSet(ssm_ctx* c_, hipStream_t s_) : c(c_), keep(c_->stream) { c->stream = s_; } ~StreamSet() { c->stream = keep; } } onmap(c, s);   // the stage events follow the kernels
            const uint8_t* sem_src = in->sem_bgr ? in->sem_bgr + (size_t)f0 * npix * 3 : nullptr;
            if (stages & SSM_STAGE_SEGNET) {          // Classifier in the loop (the variant commented out at src/rgbdframe.cpp:119-136)
                r = seg_init(c); if (r) return r;
                prof_begin(c, "segnet");
                r = seg_forward_dev(c, in->bgr + (size_t)f0 * npix * 3, nb, nullptr, c->seg->d_sem_gen, 0); if (r) return r;
                prof_end(c);
                sem_src = c->seg->d_sem_gen;
            }
            // The map of a context grows (map_settle): a launch covers all nb frames when the table is large, fewer while it is small, and the table's counters
            // are looked at between launches (map_before_launch).  Exact integer sums: how the frames are cut into launches does not change the map.
            for (int q0 = 0, nq; (stages & SSM_STAGE_MAP) && q0 < nb; q0 += nq) {
                r = map_before_launch(c, s, nb - q0, &nq); if (r) return r;
                const int g0 = f0 + q0;
                const uint8_t* sem_q = sem_src + (size_t)q0 * npix * 3;
                if ((W & 15) == 0) {         // streaming fused kernels (16 pixels per thread, 16-byte loads)
                    prof_begin(c, "map_fuse");
                    { MapLaunch L; L.depth = in->depth + (size_t)g0 * npix; L.rgb = in->bgr + (size_t)g0 * npix * 3; L.sem = sem_q; L.pose = in->pose ? in->pose + (size_t)g0 * 16 : nullptr;
                      L.n = nq; L.w = W; L.h = H; L.npoints = c->d_npoints + g0; L.valid = true;
                      r = map_fuse_launch(c, s, L); if (r) return r; }
                    prof_end(c);
                } else {                     // odd widths: mask -> ordered back-projection -> insert
                    prof_begin(c, "mask");
                    HIPCHK(c, k_moving_mask(sem_q, nq, W, H, c->d_mask, s)); prof_end(c);
                    prof_begin(c, "backproject");
                    HIPCHK(c, k_backproject(in->depth + (size_t)g0 * npix, in->bgr + (size_t)g0 * npix * 3, sem_q, c->d_mask,
                                            in->pose ? in->pose + (size_t)g0 * 16 : nullptr, nq, W, H, c->cfg.camera, c->cfg.mapper_max_distance,
                                            c->d_chunk_cnt, c->d_chunk_off, c->d_npoints + g0, c->d_total, c->d_points, s)); prof_end(c);
                    prof_begin(c, "voxel_insert");
                    HIPCHK(c, k_voxel_insert(c->d_points, c->d_total, (int64_t)nq * (int64_t)npix, (float)c->cfg.mapper_resolution, c->map.tab, c->map.cap_log2, c->map.counters, s));
                    prof_end(c);
                }
                r = map_after_launch(c, s, nq, (stages & SSM_STAGE_SEGNET) != 0 && (W & 15) == 0); if (r) return r;
            }
            return SSM_OK;
        };
        // The map stage shares no data with the ORB -> match chain: in two-chain mode it runs on a third stream (all sub-batches in order, one workspace),
        // so that the chains' latency-bound kernels (pyramid, quad-tree, orientation / BRIEF gathers) always have VALU-bound map work beside them
        // (+4 % over map-after-match on the chain's own stream).  With SSM_MAP_STREAM=0 chain 1 runs it FIRST instead, which puts the two chains half a
        // sub-batch out of step (+2.4 %)
        const bool map_first = two_chains && chain == 1 && c->map_first;
        if (map_first) { r = back(); if (r) return r; r = front(); if (r) return r; }
        else           { r = front(); if (r) return r; r = back(); if (r) return r; }
    }
    if (side || two_chains) { HIPCHK(c, hipEventRecord(c->ev_join, c->stream2)); HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0)); }
    if (nch == 3) { HIPCHK(c, hipEventRecord(c->ev_join4, c->stream4)); HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join4, 0)); }
    if (map3) { HIPCHK(c, hipEventRecord(c->ev_join3, c->stream3)); HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join3, 0)); }
    // where the map's newest work sits (ssm_ctx::map_tail): one side stream, or the context stream (serialised, no map stage, or the chains' own streams in turn)
    if (stages & SSM_STAGE_MAP) c->map_tail = map3 ? c->stream3 : (side ? c->stream2 : nullptr);
    c->prev_n = n;
    if (out) {
        out->kps = c->d_kps; out->desc = desc; out->pos3d = c->d_pos3d; out->nkp = nkp; out->matches = c->d_matches; out->nmatch = c->d_nmatch;
        out->npoints = c->d_npoints; out->cap = g.cap; out->R = R;
    }
    return SSM_OK;
}


// ---------------------------------------------------------------- utilities
extern "C" int ssm_dev_alloc(ssm_ctx* c, size_t bytes, void** out)
{
    if (!c || !out) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    uint8_t* p; int r = dalloc(c, &p, bytes); if (r) return r;
    *out = p; return SSM_OK;
}
extern "C" int ssm_dev_free(ssm_ctx* c, void* p)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (p) HIPCHK(c, hipFree(p));
    return SSM_OK;
}
extern "C" int ssm_memcpy_h2d(ssm_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SSM_OK;
}
extern "C" int ssm_dev_mem_info(ssm_ctx* c, size_t* free_bytes, size_t* total_bytes)
{
    if (!c || !free_bytes || !total_bytes) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    HIPCHK(c, hipMemGetInfo(free_bytes, total_bytes));
    return SSM_OK;
}
extern "C" int ssm_memcpy_h2d_async(ssm_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return SSM_OK;
}
extern "C" int ssm_memcpy_d2h_async(ssm_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return SSM_OK;
}
extern "C" int ssm_host_alloc(size_t bytes, void** out)
{
    if (!out || bytes == 0) return SSM_E_INVAL;
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); g_create_err = std::string("hipHostMalloc: ") + hipGetErrorString(e); return SSM_E_NOMEM; }
    *out = p; return SSM_OK;
}
extern "C" int ssm_host_free(void* p) { if (p && hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return SSM_E_HIP; } return SSM_OK; }
// is [p, p + bytes) page-locked host memory the device can read (hipHostMalloc / hipHostRegister)?  (an unknown pointer makes hipPointerGetAttributes fail: that is "no")
bool host_is_pinned(const void* p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
extern "C" int ssm_memcpy_d2h(ssm_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SSM_OK;
}
extern "C" int ssm_synth_frames_dev(ssm_ctx* c, uint64_t seed, int first, int n, int w, int h,
                                    uint8_t* bgr, uint16_t* depth, uint8_t* sem, uint8_t* lab, double* pose)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (n <= 0 || !bgr || !depth || !sem) FAIL(c, SSM_E_INVAL, "bad arguments");
    HIPCHK(c, k_synth(seed, first, n, w, h, bgr, depth, sem, lab, pose, c->stream));
    return SSM_OK;
}
