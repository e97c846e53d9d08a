// ssm_abi.hip -- host side of libssm_hip.so: context, geometry, device workspace and the extern "C" entry points
// declared in include/ssm_hip.h.  No computation of the path happens on the host: this file only sizes buffers,
// moves caller data and enqueues the kernels of kernels_*.hip on the context stream.  There is NO CPU fallback: if
// HIP is unusable ssm_create fails with SSM_E_NODEVICE / SSM_E_HIP.
#include "ssm_internal.h"
#include "pnp_chain.h"
#include <rccl/rccl.h>
#include <cmath>
#include <cfloat>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <functional>
#include <string>
#include <vector>

static const int8_t k_default_pattern[1024] = {
#include "orb_pattern.inc"
};
static thread_local std::string g_create_err;

namespace {

struct VoxTable {           // tab[slots] | occ[slots] | counter block (32 bytes: count, flags, overflow records, overflow capacity, overflow list address)
    ssm_voxel* tab = nullptr; uint32_t* occ = nullptr; int32_t* counters = nullptr; int cap_log2 = 0;
    ssm_voxel* ovf = nullptr; int ovf_cap = 0;       // the overflow list of the context map (kernels_map.hip vox_overflow_slot); the temporary tables have none
    size_t bytes() const { const size_t s = (size_t)1 << cap_log2; return s * sizeof(ssm_voxel) + s * 4 + 32; }
};
static const int VOX_OVF_RECORDS = 1 << 18;          // 29 MB per context
struct StageRec { const char* name; hipEvent_t a, b; };
// SegNet driving_webdemo: 26 conv layers; op list interleaves pools / unpools
struct SegLayerDef { int cin, cout, h, w; };
static const int SEG_NW = 480, SEG_NH = 360, SEG_NCLS = 12, SEG_LAYERS = 26;
static const SegLayerDef k_seg_layers[SEG_LAYERS] = {
    {3, 64, 360, 480}, {64, 64, 360, 480},                                   // conv1_1 conv1_2 | pool1
    {64, 128, 180, 240}, {128, 128, 180, 240},                               // conv2_x         | pool2
    {128, 256, 90, 120}, {256, 256, 90, 120}, {256, 256, 90, 120},           // conv3_x         | pool3
    {256, 512, 45, 60}, {512, 512, 45, 60}, {512, 512, 45, 60},              // conv4_x         | pool4 (ceil: 23x30)
    {512, 512, 23, 30}, {512, 512, 23, 30}, {512, 512, 23, 30},              // conv5_x         | pool5 (ceil: 12x15)
    {512, 512, 23, 30}, {512, 512, 23, 30}, {512, 512, 23, 30},              // upsample5 | conv5_3_D conv5_2_D conv5_1_D
    {512, 512, 45, 60}, {512, 512, 45, 60}, {512, 256, 45, 60},              // upsample4 | conv4_x_D
    {256, 256, 90, 120}, {256, 256, 90, 120}, {256, 128, 90, 120},           // upsample3 | conv3_x_D
    {128, 128, 180, 240}, {128, 64, 180, 240},                               // upsample2 | conv2_x_D
    {64, 64, 360, 480}, {64, 12, 360, 480}                                   // upsample1 | conv1_2_D conv1_1_D (no BN/ReLU)
};

} // namespace
#define SG_FAIL_WORDS 256
struct StereoState {        // workspace of the stereo path (quad matcher, SGBM depth, stereo VO) for one image geometry, B frames per launch
    int w = 0, h = 0, maxc = 0, B = 0;
    QuadBatch qb{};                          // image slots: 2 sides x (B + 1) pyramids + Scharr derivatives
    uint8_t* pyr = nullptr; int16_t* der = nullptr;
    GfttWork gw{};                           // goodFeaturesToTrack workspace (kernels_quad.hip)
    int keycap = 0; int *overflow = nullptr, *ncorner = nullptr, *has_prev = nullptr;
    int* sg_fail = nullptr;                  // SG_FAIL_WORDS words: word (sub-batch index mod SG_FAIL_WORDS) is set by that sub-batch's sgbm_sweep when a strip hand-off times out (kernels_sgbm.hip)
    // the depth stage of the most recent sequence call, kept so that sub-batches whose sweep timed out can be repeated in form 1 once the call is known to have
    // failed (ssm_sync / check_device_flags: the caller's input buffers must stay untouched until then, as for any asynchronous call)
    struct { bool valid = false; ssm_stereo_frames_dev in{}; int B = 0; } sg_pending;
    float* pts = nullptr;                    // [5][B][maxc] (x, y): lc (GFTT corners), rc, rp, lp, lp_direct
    uint8_t* status = nullptr; float* err = nullptr;        // ssm_lk_track outputs
    double* tr_all = nullptr; int32_t *vcount = nullptr, *rand_off = nullptr, *consumed = nullptr; int vo_iters = 0;   // stereo VO scratch (B x iters hypotheses)
    void* sg_wsN[3] = {nullptr, nullptr, nullptr}; size_t sg_ws_bytesN[3] = {0, 0, 0}; int* dminN[3] = {nullptr, nullptr, nullptr};   // SGBM workspaces (sized for the frames per launch actually used): successive sub-batches of a sequence run SGBM on up to three streams, one workspace each
    // sequence outputs (seq_cap frames)
    int seq_cap = 0;
    ssm_pmatch* quad = nullptr; int32_t* nquad = nullptr; float* corners = nullptr; int32_t* ncorners = nullptr; int16_t* disp = nullptr; uint16_t* depth = nullptr;
    double* tr = nullptr; int32_t *inliers = nullptr, *vo_result = nullptr;
    bool have_prev = false;                  // slot 0 holds the last frame of the previous sequence call
    uint8_t* in_stage = nullptr; size_t in_stage_bytes = 0;     // device staging of the per-pair host-pointer entry points
};
struct SegNetState {
    bool set[SEG_LAYERS] = {};
    void* w[SEG_LAYERS] = {}; float* scale[SEG_LAYERS] = {}; float* shift[SEG_LAYERS] = {};
    int cinp[SEG_LAYERS], coutp[SEG_LAYERS], coutstore[SEG_LAYERS];
    int batch = 0;
    void *actA = nullptr, *actB = nullptr, *last_logits = nullptr; uint8_t* code[5] = {}; uint8_t* labels = nullptr;
    int32_t *pre_xofs = nullptr, *pre_yofs = nullptr, *post_xofs = nullptr, *post_yofs = nullptr;
    int16_t *pre_xa = nullptr, *pre_ya = nullptr, *post_xa = nullptr, *post_ya = nullptr;
    uint8_t* d_sem_gen = nullptr;       // generated colour labels for the sequence path (max_batch frames)
};

static void stereo_free(StereoState* q);

struct ssm_ctx {
    std::mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    bool side_ready = false;            // ensure_side_streams completed
    hipStream_t stream2 = nullptr;      // ssm_seq_process: the SegNet + map stage of a sub-batch runs here, beside the ORB + match chain
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t stream3 = nullptr; hipEvent_t ev_join3 = nullptr; uint8_t* d_mask3 = nullptr; int map_stream = 1;   // two-chain mode: the map stage on a stream of its own (SSM_MAP_STREAM=0: on the chain's stream)
    ssm_config cfg{};
    OrbGeom g{};
    std::string err;
    int B = 1, R = 5;
    // constant tables
    void* d_blur_tab = nullptr; bool blur_mfma = true;                 // blur_mfma_kernel's coefficient fragments (kernels_orb.hip); SSM_BLUR_VARIANT=0: the VALU kernel
    int8_t* d_pattern = nullptr; float* d_pattern_f = nullptr;      // the BRIEF table as given, and as floats for brief_kernel
    int32_t* d_xofs[SSM_MAX_LEVELS] = {}; int16_t* d_xa[SSM_MAX_LEVELS] = {};
    void* d_xgrp[SSM_MAX_LEVELS] = {};       // resize4_kernel's per-group constants (null: the level uses the general resize kernel)
    int32_t* d_yofs[SSM_MAX_LEVELS] = {}; int16_t* d_ya[SSM_MAX_LEVELS] = {};
    // batch workspace (B frames)
    uint8_t *d_pyr = nullptr, *d_blur = nullptr; int32_t* d_cellmax = nullptr; cand_t* d_cand = nullptr; uint16_t* d_nodeof = nullptr;
    int32_t* d_ncand = nullptr; uint32_t* d_sel = nullptr; int32_t* d_nsel = nullptr; int32_t* d_status = nullptr; uint4* d_kpaux = nullptr;
    // second ORB / map workspace: ssm_seq_process runs alternate sub-batches as two chains on two streams (allocated at first use)
    struct AltWork { uint8_t *pyr = nullptr, *blur = nullptr; int32_t* cellmax = nullptr; cand_t* cand = nullptr; uint16_t* nodeof = nullptr;
                     int32_t* ncand = nullptr; uint32_t* sel = nullptr; int32_t* nsel = nullptr; uint8_t* mask = nullptr; uint4* kpaux = nullptr; bool ready = false; } alt, alt2;
    hipEvent_t ev_orb[3] = {nullptr, nullptr, nullptr};
    hipStream_t stream4 = nullptr; hipEvent_t ev_join4 = nullptr; int nchains = 3;      // a third ORB -> match chain (workspace alt2, stream4) when a call has more than two sub-batches; SSM_CHAINS=2: two
    uint8_t* d_mask = nullptr; int32_t* d_chunk_cnt = nullptr; int64_t* d_chunk_off = nullptr; int64_t* d_total = nullptr;
    ssm_point* d_points = nullptr;
    ssm_point* d_vmap = nullptr; int vmap_n = 0; size_t vmap_cap = 0;      // Mapper::viewer's filtered map, device-resident (ssm_viewer_map_update)
    ssm_point* d_vcat = nullptr; size_t vcat_cap = 0;                        // its concatenation buffer
    struct CloudSlab { ssm_point* d = nullptr; size_t cap = 0, used = 0; int live = 0; };
    std::vector<CloudSlab> cloud_slabs;                                      // key-frame clouds (ssm_backproject_dev) are carved from slabs: no hipMalloc per cloud
    // staging for the host-pointer entry points (one frame) + generic scratch
    uint8_t *d_in_img = nullptr, *d_in_sem = nullptr; uint16_t* d_in_depth = nullptr; double* d_in_pose = nullptr;
    void* d_scratch = nullptr; size_t scratch_bytes = 0;
    void* d_scratch2 = nullptr; size_t scratch2_bytes = 0;
    // sequence outputs
    int seq_cap = 0, prev_n = -1;
    ssm_keypoint* d_kps = nullptr; uint8_t* d_desc_all = nullptr; int32_t* d_nkp_all = nullptr; float* d_pos3d = nullptr;
    ssm_dmatch* d_matches = nullptr; int32_t* d_nmatch = nullptr; int32_t* d_match_pend = nullptr; int32_t* d_npoints = nullptr; uint8_t* d_hist_tmp = nullptr;
    uint8_t* d_exp_q = nullptr; uint8_t* d_exp_t = nullptr; uint8_t* d_knn = nullptr; int capT = 0; bool match_mfma = true; bool map_first = true; bool map_compact = true;   // the matcher's expanded descriptor rows (kernels_match.hip)
    // voxel tables
    VoxTable map, tmp; bool map_full_reported = false;   // table-full already reported by check_device_flags (reset by ssm_map_clear)
    // the context map grows (map_settle); between the map launches of ssm_seq_process its counters come back through a two-slot ring of asynchronous copies
    int vox_max_log2 = 28; int32_t* h_map_snap = nullptr; hipEvent_t map_snap_ev[2] = {nullptr, nullptr}; uint64_t map_launches = 0; int map_grown = 0;
    // multi-GPU: the communicator of ssm_comm_init_rank (one rank per context / GPU) and the gathered counts
    ncclComm_t comm = nullptr; int comm_rank = 0, comm_size = 1; int32_t* d_comm_counts = nullptr; int comm_counts_cap = 0;
    // SegNet
    struct SegNetState* seg = nullptr;
    // quad matcher
    struct StereoState* stereo = nullptr; int stereo_B = 16; int stereo_sgbm_streams = 2;      // ssm_config.sgbm_streams (SSM_SGBM_STREAMS overrides: ablations)
    int sgbm_form_cfg = 0; long sgbm_fallbacks = 0;                                              // ssm_config.sgbm_form; sub-batches repeated in form 1 after a sweep time-out
    // profiling
    bool profiling = false;
    uint8_t* h_pinned = nullptr; size_t pinned_bytes = 0;   // host staging for the image-sized host-pointer calls (pageable hipMemcpy is ~1 GB/s)
    // the per-frame entry points (ssm_orb_extract[_async], ssm_match[_async]): a ring of pinned host memory (inputs staged, results landed) and a ring of
    // device memory (result blocks), bump-allocated per call and released by ssm_wait; `pending` = what ssm_wait still has to hand to the callers
    uint8_t* h_ring = nullptr; uint8_t* d_ring = nullptr; size_t ring_bytes = 0, h_ring_off = 0, d_ring_off = 0;
    std::vector<std::function<int(ssm_ctx*)>> pending;
    bool serialize = false;             // profiling mode 2: keep the side work of ssm_seq_process on the context stream (clean per-stage times)
    std::vector<StageRec> recs; std::vector<hipEvent_t> pool; size_t pool_used = 0;
    std::vector<std::string> stage_names; std::vector<float> stage_ms; std::vector<int> stage_launches;
};

#define FAIL(ctx, code, msg) do { (ctx)->err = (msg); return (code); } while (0)
#define HIPCHK(ctx, expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__); return SSM_E_HIP; } } while (0)

static inline int cv_round_f(float v) { return (int)lrint((double)v); }

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
static void resize_tables(int ssize, int dsize, std::vector<int32_t>& ofs, std::vector<int16_t>& coef)
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
template <class T> static int dalloc(ssm_ctx* c, T** p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void**)p, count * sizeof(T));
    if (e != hipSuccess) { c->err = std::string("hipMalloc(") + std::to_string(count * sizeof(T)) + "): " + hipGetErrorString(e); return SSM_E_NOMEM; }
    return SSM_OK;
}
#define DALLOC(ctx, p, n) do { int r__ = dalloc(ctx, &(p), (size_t)(n)); if (r__) return r__; } while (0)
static int ensure_scratch(ssm_ctx* c, size_t bytes)
{
    if (bytes <= c->scratch_bytes) return SSM_OK;
    if (c->d_scratch) { hipStreamSynchronize(c->stream); hipFree(c->d_scratch); c->d_scratch = nullptr; c->scratch_bytes = 0; }
    uint8_t* p; int r = dalloc(c, &p, bytes); if (r) return r;
    c->d_scratch = p; c->scratch_bytes = bytes; return SSM_OK;
}
static int ensure_pinned(ssm_ctx* c, size_t bytes)
{
    if (bytes <= c->pinned_bytes) return SSM_OK;
    if (c->h_pinned) { hipStreamSynchronize(c->stream); hipHostFree(c->h_pinned); c->h_pinned = nullptr; c->pinned_bytes = 0; }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { c->err = "hipHostMalloc failed"; return SSM_E_HIP; }
    c->h_pinned = (uint8_t*)p; c->pinned_bytes = bytes; return SSM_OK;
}
static int ensure_scratch2(ssm_ctx* c, size_t bytes)
{
    if (bytes <= c->scratch2_bytes) return SSM_OK;
    if (c->d_scratch2) { hipStreamSynchronize(c->stream); hipFree(c->d_scratch2); c->d_scratch2 = nullptr; c->scratch2_bytes = 0; }
    uint8_t* p; int r = dalloc(c, &p, bytes); if (r) return r;
    c->d_scratch2 = p; c->scratch2_bytes = bytes; return SSM_OK;
}
static int table_alloc(ssm_ctx* c, VoxTable& t, int cap_log2)
{
    t.cap_log2 = cap_log2;
    uint8_t* p; int r = dalloc(c, &p, t.bytes()); if (r) return r;
    const size_t slots = (size_t)1 << cap_log2;
    t.tab = reinterpret_cast<ssm_voxel*>(p); t.occ = reinterpret_cast<uint32_t*>(t.tab + slots); t.counters = reinterpret_cast<int32_t*>(t.occ + slots);
    HIPCHK(c, k_voxel_clear(t.tab, -cap_log2, t.counters, c->stream));
    struct { int32_t cap, pad; ssm_voxel* buf; } tail = { t.ovf ? t.ovf_cap : 0, 0, t.ovf };      // counters[3], counters[4..5]
    static_assert(sizeof(tail) == 16, "counter block tail");
    int32_t head[3] = {0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(t.counters, head, 12, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(t.counters + 3, &tail.cap, 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(t.counters + 4, &tail.buf, 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));                   // (the sources are on this stack)
    return SSM_OK;
}
// The context map has no capacity of its own (the reference's globalMap grows without limit, src/mapper.cpp:121-158): voxel_capacity_log2 is where it STARTS.
// map_settle brings the map to rest on stream s (blocking): the overflow list is merged into the table and the table is re-hashed into a larger one whenever
// 4 x (voxels + overflow records + reserve) exceeds its slots -- `reserve` = new voxels the caller is about to add at most, so that an insert / merge of a known
// size can never overflow.  SSM_E_CAPACITY only beyond 2^vox_max_log2 slots (28: the key's range), SSM_E_NOMEM when the larger table cannot be allocated; in both
// cases nothing is lost: table and list stay as they are.
static int map_settle(ssm_ctx* c, hipStream_t s, int64_t reserve)
{
    VoxTable& t = c->map;
    int lo = 0;                                                   // overflow records [0, lo) are merged already
    for (int round = 0; round < 64; round++) {
        int32_t cnt[4];
        HIPCHK(c, hipMemcpyAsync(cnt, t.counters, 16, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        const int64_t n = cnt[0], hi = cnt[2] < t.ovf_cap ? cnt[2] : t.ovf_cap, m = hi - lo;
        const int64_t slots = (int64_t)1 << t.cap_log2;
        const bool grow = 4 * (n + m + reserve) > slots && t.cap_log2 < c->vox_max_log2;
        if (!grow && 2 * (n + m + reserve) > slots) {
            // at voxel_max_capacity_log2 and more than half full.  A caller that announced its insert (reserve) is refused before anything is added; records
            // waiting in the overflow list have no table to go to: the map is incomplete from here on (flag bit 0, reported until ssm_map_clear)
            if (m > 0) { const int32_t lost[2] = { cnt[1] | 1, 0 }; HIPCHK(c, hipMemcpyAsync(t.counters + 1, lost, 8, hipMemcpyHostToDevice, s)); HIPCHK(c, hipStreamSynchronize(s)); }
            FAIL(c, SSM_E_CAPACITY, "the voxel map needs more than 2^" + std::to_string(c->vox_max_log2) + " slots (voxel_max_capacity_log2)");
        }
        if (m <= 0 && !grow) {
            if (cnt[2] != 0) { const int32_t z = 0; HIPCHK(c, hipMemcpyAsync(t.counters + 2, &z, 4, hipMemcpyHostToDevice, s)); HIPCHK(c, hipStreamSynchronize(s)); }
            return SSM_OK;
        }
        if (grow) {
            int L = t.cap_log2; while (L < c->vox_max_log2 && 4 * (n + m + reserve) > ((int64_t)1 << L)) L++;
            VoxTable nt; nt.ovf = t.ovf; nt.ovf_cap = t.ovf_cap;
            { hipStream_t keep = c->stream; c->stream = s; const int r = table_alloc(c, nt, L); c->stream = keep; if (r) return r; }
            // the flags travel with the map; the new counter block goes on counting overflow records where the old one stopped (the records [lo, hi) are still to
            // merge, and the re-hash itself appends behind them should it need the list)
            const int32_t carry[2] = { cnt[1], cnt[2] < t.ovf_cap ? cnt[2] : t.ovf_cap };
            HIPCHK(c, hipMemcpyAsync(nt.counters + 1, carry, 8, hipMemcpyHostToDevice, s));
            HIPCHK(c, k_voxel_rehash(t.tab, t.cap_log2, nt.tab, nt.cap_log2, nt.counters, s));
            HIPCHK(c, hipStreamSynchronize(s));
            hipFree(t.tab);
            t = nt; c->map_grown++;
            continue;                                             // (count again: the re-hash itself may have used the list)
        }
        HIPCHK(c, k_voxel_merge(t.ovf + lo, (int)m, t.tab, t.cap_log2, t.counters, s));
        lo = (int)hi;
        if (lo >= t.ovf_cap) {                                    // the list was full to the brim: empty it before anything can be appended again
            HIPCHK(c, hipStreamSynchronize(s));
            HIPCHK(c, hipMemcpyAsync(cnt, t.counters, 16, hipMemcpyDeviceToHost, s)); HIPCHK(c, hipStreamSynchronize(s));
            if (cnt[2] > t.ovf_cap && !(cnt[1] & 1)) FAIL(c, SSM_E_CAPACITY, "voxel map: the overflow list overflowed while it was merged");
            const int32_t z = 0; HIPCHK(c, hipMemcpyAsync(t.counters + 2, &z, 4, hipMemcpyHostToDevice, s)); HIPCHK(c, hipStreamSynchronize(s));
            lo = 0;
        }
    }
    FAIL(c, SSM_E_CAPACITY, "voxel map: the overflow list did not drain");
}
// ssm_seq_process, in front of every launch of the map stage on stream s.  A small table (< 2^20 slots) is settled exactly every time and takes only
// slots / 4096 frames per launch; a large one is checked against the counters of the launch before the previous one (a two-slot ring of asynchronous copies:
// the host never waits for the launch it has just queued) and settled when it is a quarter full or its overflow list is in use.
static int map_before_launch(ssm_ctx* c, hipStream_t s)
{
    VoxTable& t = c->map;
    if (t.cap_log2 < 20) return map_settle(c, s, 0);
    if (c->map_launches < 2) return SSM_OK;
    const int slot = (int)(c->map_launches & 1);
    HIPCHK(c, hipEventSynchronize(c->map_snap_ev[slot]));
    const int32_t* cnt = c->h_map_snap + 4 * slot;
    if (cnt[2] > 0 || 4 * (int64_t)cnt[0] > ((int64_t)1 << t.cap_log2)) { c->map_launches = 0; return map_settle(c, s, 0); }
    return SSM_OK;
}
static int map_after_launch(ssm_ctx* c, hipStream_t s)
{
    const int slot = (int)(c->map_launches & 1);
    HIPCHK(c, hipMemcpyAsync(c->h_map_snap + 4 * slot, c->map.counters, 16, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipEventRecord(c->map_snap_ev[slot], s));
    c->map_launches++;
    return SSM_OK;
}
static int map_frames_per_launch(const ssm_ctx* c, int nb) { const int f = c->map.cap_log2 >= 20 ? nb : (1 << c->map.cap_log2) >> 12; return f < 1 ? 1 : (f > nb ? nb : f); }
static void prof_begin(ssm_ctx* c, const char* name)
{
    if (!c->profiling) return;
    auto get = [&]() { if (c->pool_used == c->pool.size()) { hipEvent_t e; hipEventCreate(&e); c->pool.push_back(e); } return c->pool[c->pool_used++]; };
    StageRec r; r.name = name; r.a = get(); r.b = get();
    hipEventRecord(r.a, c->stream);
    c->recs.push_back(r);
}
static void prof_end(ssm_ctx* c) { if (c->profiling) hipEventRecord(c->recs.back().b, c->stream); }

// ORB scratch overflow (d_status) is checked after every ORB entry point; the voxel-table-full flag (counters[1]) belongs to the MAP entry points
// (ssm_sync after ssm_seq_process, ssm_map_*): it is reported once, so that one overflowing call does not fail every later call on the
// context (the map then lacks the dropped points: ssm_map_clear / a larger voxel_capacity_log2 is the remedy the message names)
static int sgbm_recover(ssm_ctx* c);
static int check_device_flags(ssm_ctx* c, bool with_map)
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
    DALLOC(c, c->d_pyr, (size_t)B * g.pyr_bytes + 16); DALLOC(c, c->d_blur, (size_t)B * g.blur_bytes); DALLOC(c, c->d_cellmax, k_fast_cellmax_ints(B, g));
    DALLOC(c, c->d_cand, (size_t)B * g.cand_total); DALLOC(c, c->d_nodeof, (size_t)B * g.cand_total);
    DALLOC(c, c->d_ncand, (size_t)B * g.nlevels); DALLOC(c, c->d_sel, (size_t)B * g.sel_total); DALLOC(c, c->d_nsel, (size_t)B * g.nlevels);
    DALLOC(c, c->d_kpaux, (size_t)B * g.sel_total * 2);          // KpAux + KpRec per slot
    DALLOC(c, c->d_status, 1); HIPCHK(c, hipMemset(c->d_status, 0, 4));
    const int chunks = backproject_chunks(W, H);
    DALLOC(c, c->d_mask, (size_t)B * W * H); DALLOC(c, c->d_chunk_cnt, (size_t)B * chunks); DALLOC(c, c->d_chunk_off, (size_t)B * chunks);
    DALLOC(c, c->d_total, 2); DALLOC(c, c->d_points, (size_t)B * W * H);
    DALLOC(c, c->d_in_img, (size_t)W * H * 3); DALLOC(c, c->d_in_sem, (size_t)W * H * 3); DALLOC(c, c->d_in_depth, (size_t)W * H); DALLOC(c, c->d_in_pose, 16);
    DALLOC(c, c->map.ovf, VOX_OVF_RECORDS); c->map.ovf_cap = VOX_OVF_RECORDS;
    { void* hp = nullptr; HIPCHK(c, hipHostMalloc(&hp, 32, hipHostMallocDefault)); c->h_map_snap = (int32_t*)hp; memset(hp, 0, 32); }
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
    { const char* e = getenv("SSM_CHAINS"); c->nchains = e ? atoi(e) : 3; }
    // the stereo path's knobs come from the configuration (two contexts of a process may differ); the environment variables remain as overrides for ablation runs
    { const char* e = getenv("SSM_STEREO_BATCH"); int b = e ? atoi(e) : cfg->stereo_batch > 0 ? cfg->stereo_batch : (cfg->max_batch > 0 ? cfg->max_batch : 1); c->stereo_B = b < 1 ? 1 : b > 128 ? 128 : b; }
    { const char* e = getenv("SSM_SGBM_STREAMS"); const int v = e ? atoi(e) : cfg->sgbm_streams; if (v > 0) c->stereo_sgbm_streams = v > 3 ? 3 : v; }
    c->sgbm_form_cfg = cfg->sgbm_form;
    { const char* e = getenv("SSM_MAP_STREAM"); c->map_stream = e ? atoi(e) : 1; }
    { const char* e = getenv("SSM_MAP_VARIANT"); c->map_compact = !(e && atoi(e) == 0); }      // 0: map_stream_kernel (every pixel through the full arithmetic)
    { const char* e = getenv("SSM_MAP_FIRST"); c->map_first = !(e && atoi(e) == 0); }
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
    void* ptrs[] = { c->d_pattern, c->d_pyr, c->d_blur, c->d_cellmax, c->d_cand, c->d_nodeof, c->d_ncand, c->d_sel, c->d_nsel, c->d_status, c->d_mask,
                     c->d_chunk_cnt, c->d_chunk_off, c->d_total, c->d_points, c->d_in_img, c->d_in_sem, c->d_in_depth, c->d_in_pose,
                     c->d_scratch, c->d_scratch2, c->d_kps, c->d_desc_all, c->d_nkp_all, c->d_pos3d, c->d_matches, c->d_nmatch, c->d_match_pend, c->d_npoints,
                     c->d_hist_tmp, c->map.tab, c->tmp.tab, c->d_kpaux, c->d_pattern_f, c->d_exp_q, c->d_exp_t, c->d_knn, c->d_blur_tab, c->d_vmap, c->d_vcat };
    for (void* p : ptrs) if (p) hipFree(p);
    if (c->map.ovf) hipFree(c->map.ovf);
    if (c->h_map_snap) hipHostFree(c->h_map_snap);
    for (int k = 0; k < 2; k++) if (c->map_snap_ev[k]) hipEventDestroy(c->map_snap_ev[k]);
    { void* ap[] = { c->alt.pyr, c->alt.blur, c->alt.cellmax, c->alt.cand, c->alt.nodeof, c->alt.ncand, c->alt.sel, c->alt.nsel, c->alt.mask, c->alt.kpaux };
      for (void* p : ap) if (p) hipFree(p); }
    for (int i = 0; i < 3; i++) if (c->ev_orb[i]) hipEventDestroy(c->ev_orb[i]);
    { void* ap2[] = { c->alt2.pyr, c->alt2.blur, c->alt2.cellmax, c->alt2.cand, c->alt2.nodeof, c->alt2.ncand, c->alt2.sel, c->alt2.nsel, c->alt2.mask, c->alt2.kpaux };
      for (void* p : ap2) if (p) hipFree(p); }
    if (c->stream4) hipStreamDestroy(c->stream4);
    if (c->ev_join4) hipEventDestroy(c->ev_join4);
    for (int l = 0; l < SSM_MAX_LEVELS; l++) { if (c->d_xofs[l]) hipFree(c->d_xofs[l]); if (c->d_xa[l]) hipFree(c->d_xa[l]); if (c->d_yofs[l]) hipFree(c->d_yofs[l]); if (c->d_ya[l]) hipFree(c->d_ya[l]); if (c->d_xgrp[l]) hipFree(c->d_xgrp[l]); }
    if (c->seg) {
        SegNetState* g = c->seg;
        void* sp[] = { g->actA, g->actB, g->labels, g->d_sem_gen, g->pre_xofs, g->pre_yofs, g->post_xofs, g->post_yofs, g->pre_xa, g->pre_ya, g->post_xa, g->post_ya,
                       g->code[0], g->code[1], g->code[2], g->code[3], g->code[4] };
        for (void* p : sp) if (p) hipFree(p);
        for (int l = 0; l < SEG_LAYERS; l++) { if (g->w[l]) hipFree(g->w[l]); if (g->scale[l]) hipFree(g->scale[l]); if (g->shift[l]) hipFree(g->shift[l]); }
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
    if (c->d_mask3) hipFree(c->d_mask3);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    delete c;
}
void ssm_internal_get_config(const ssm_ctx* c, ssm_config* out) { *out = c->cfg; }
int ssm_internal_get_device(const ssm_ctx* c) { return c->device; }
extern "C" int ssm_orb_capacity(const ssm_ctx* c) { return c ? c->g.cap : 0; }
extern "C" void* ssm_stream(ssm_ctx* c) { return c ? (void*)c->stream : nullptr; }
static int wait_pending(ssm_ctx* c);
extern "C" int ssm_sync(ssm_ctx* c)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    hipSetDevice(c->device);
    { int r = wait_pending(c); if (r) return r; }                 // asynchronous per-frame calls still in flight are completed (their results delivered) too
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
static int ensure_side_streams(ssm_ctx* c)
{
    if (c->side_ready) return SSM_OK;
    // each handle is created only if it is still missing: a call that failed half-way leaves side_ready false and the next call resumes
    // SSM_MAP_CUS=N (ablation): the map stage's stream is confined to N compute units (hipExtStreamCreateWithCUMask; SSM_MAP_CUS_SPREAD=1: every
    // (256 / N)-th unit instead of the first N), and with SSM_CHAIN_CUS=1 the ORB chains' side streams get the complement -- a static split of the machine in place of
    // the hardware's block-by-block arbitration between kernels that each fill a CU on their own
    auto cu_mask = [&](bool map_side, uint32_t (&m)[8]) -> bool {
        const char* e = getenv("SSM_MAP_CUS"); const int n = e ? atoi(e) : 0;
        if (n <= 0 || n >= 256) return false;
        const char* sp = getenv("SSM_MAP_CUS_SPREAD"); const bool spread = sp && atoi(sp) != 0;
        for (int k = 0; k < 8; k++) m[k] = 0;
        for (int i = 0; i < n; i++) { const int bit = spread ? (int)((long)i * 256 / n) : i; m[bit >> 5] |= 1u << (bit & 31); }
        if (!map_side) { const char* ce = getenv("SSM_CHAIN_CUS"); if (!(ce && atoi(ce) != 0)) return false; for (int k = 0; k < 8; k++) m[k] = ~m[k]; }
        return true;
    };
    auto mk_stream = [&](hipStream_t* st, int role = 0) -> hipError_t {      // role 1: the map stream, 2: a chain's side stream
        if (*st) return hipSuccess;
        uint32_t m[8];
        if (role && cu_mask(role == 1, m)) return hipExtStreamCreateWithCUMask(st, 8, m);
        return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    };
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
    DALLOC(c, a.pyr, (size_t)B * g.pyr_bytes + 16); DALLOC(c, a.blur, (size_t)B * g.blur_bytes); DALLOC(c, a.cellmax, k_fast_cellmax_ints(B, g));
    DALLOC(c, a.cand, (size_t)B * g.cand_total); DALLOC(c, a.nodeof, (size_t)B * g.cand_total);
    DALLOC(c, a.ncand, (size_t)B * g.nlevels); DALLOC(c, a.sel, (size_t)B * g.sel_total); DALLOC(c, a.nsel, (size_t)B * g.nlevels);
    DALLOC(c, a.mask, (size_t)B * g.W * g.H); DALLOC(c, a.kpaux, (size_t)B * g.sel_total * 2);
    a.ready = true;
    return SSM_OK;
}
static int ensure_alt(ssm_ctx* c)
{
    int r = ensure_alt_ws(c, c->alt); if (r) return r;
    if (c->nchains >= 3) { r = ensure_alt_ws(c, c->alt2); if (r) return r; }
    if (!c->d_mask3) DALLOC(c, c->d_mask3, (size_t)c->B * c->g.W * c->g.H);
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
static int wait_pending(ssm_ctx* c)
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
static int orb_extract_enqueue(ssm_ctx* c, const uint8_t* img, int w, int h, int stride, int channels, const uint16_t* depth,
                               ssm_keypoint* kps, uint8_t* desc, float* pos3d, int cap, int* n_out)
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
    if ((size_t)stride == row) memcpy(h_in, img, ib);
    else for (int y = 0; y < h; y++) memcpy(h_in + (size_t)y * row, img + (size_t)y * stride, row);
    if (depth) memcpy(h_in + ib, depth, db);
    HIPCHK(c, hipMemcpyAsync(c->d_in_img, h_in, ib, hipMemcpyHostToDevice, c->stream));
    if (depth) HIPCHK(c, hipMemcpyAsync(c->d_in_depth, h_in + ib, db, hipMemcpyHostToDevice, c->stream));
    int32_t* dn = reinterpret_cast<int32_t*>(dp);
    ssm_keypoint* dk = reinterpret_cast<ssm_keypoint*>(dp + 64);
    uint8_t* dd = reinterpret_cast<uint8_t*>(dk + ocap);
    float* dps = reinterpret_cast<float*>(dd + (size_t)ocap * 32);
    r = run_orb(c, c->d_in_img, channels, depth ? c->d_in_depth : nullptr, 1, dk, dd, dps, dn); if (r) return r;
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
    int r = orb_extract_enqueue(c, img, w, h, stride, channels, depth, kps, desc, pos3d, cap, n_out); if (r) return r;
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

// ---------------------------------------------------------------- voxel map
static int table_count(ssm_ctx* c, VoxTable& t, int* n)
{
    int32_t cnt[2];
    if (&t == &c->map) { const int r = map_settle(c, c->stream, 0); if (r) return r; }
    HIPCHK(c, hipMemcpyAsync(cnt, t.counters, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (cnt[1] & 1) FAIL(c, SSM_E_CAPACITY, "voxel map incomplete (contributions were dropped): ssm_map_clear and start from a larger voxel_capacity_log2");
    *n = cnt[0];
    return SSM_OK;
}
// sorts the table's voxels by key; leaves compact array + order in scratch2.  returns pointers
static int table_sorted(ssm_ctx* c, VoxTable& t, int* n_out, ssm_voxel** compact, uint32_t** order)
{
    int n = 0; int r = table_count(c, t, &n); if (r) return r;
    *n_out = n; *compact = nullptr; *order = nullptr;
    if (n == 0) return SSM_OK;
    size_t tmp_bytes = 0;
    HIPCHK(c, voxel_sort_pairs(nullptr, &tmp_bytes, nullptr, n, nullptr, nullptr, nullptr, nullptr, c->stream));
    const size_t a = ((size_t)n * sizeof(ssm_voxel) + 255) & ~(size_t)255, kb = ((size_t)n * 8 + 255) & ~(size_t)255, ib = ((size_t)n * 4 + 255) & ~(size_t)255;
    r = ensure_scratch2(c, a + 2 * kb + 2 * ib + tmp_bytes + 512); if (r) return r;
    uint8_t* p = reinterpret_cast<uint8_t*>(c->d_scratch2);
    ssm_voxel* comp = reinterpret_cast<ssm_voxel*>(p); p += a;
    uint64_t* ka = reinterpret_cast<uint64_t*>(p); p += kb; uint64_t* kbuf = reinterpret_cast<uint64_t*>(p); p += kb;
    uint32_t* ia = reinterpret_cast<uint32_t*>(p); p += ib; uint32_t* ibuf = reinterpret_cast<uint32_t*>(p); p += ib;
    int32_t* dn = reinterpret_cast<int32_t*>(p); p += 256;
    HIPCHK(c, k_voxel_compact(t.tab, t.cap_log2, comp, dn, c->stream));
    HIPCHK(c, voxel_sort_pairs(p, &tmp_bytes, comp, n, ka, kbuf, ia, ibuf, c->stream));
    *compact = comp; *order = ibuf;
    return SSM_OK;
}
static int table_export_points(ssm_ctx* c, VoxTable& t, ssm_point* out, int cap, int* n_out)
{
    int n; ssm_voxel* comp; uint32_t* order;
    int r = table_sorted(c, t, &n, &comp, &order); if (r) return r;
    *n_out = n;
    if (n > cap) FAIL(c, SSM_E_CAPACITY, "point buffer too small (need " + std::to_string(n) + ")");
    if (n == 0) return SSM_OK;
    r = ensure_scratch(c, (size_t)n * sizeof(ssm_point)); if (r) return r;
    HIPCHK(c, k_voxel_gather_points(comp, order, n, reinterpret_cast<ssm_point*>(c->d_scratch), c->stream));
    HIPCHK(c, hipMemcpyAsync(out, c->d_scratch, (size_t)n * sizeof(ssm_point), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SSM_OK;
}
extern "C" int ssm_map_clear(ssm_ctx* c)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    HIPCHK(c, k_voxel_clear(c->map.tab, c->map.cap_log2, c->map.counters, c->stream));       // (the capacity it has grown to stays)
    c->map_full_reported = false; c->map_launches = 0;
    return SSM_OK;
}
extern "C" int ssm_map_insert(ssm_ctx* c, const ssm_point* pts, int n)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (n < 0 || (n && !pts)) FAIL(c, SSM_E_INVAL, "bad arguments");
    if (n == 0) return SSM_OK;
    int r = ensure_scratch(c, (size_t)n * sizeof(ssm_point)); if (r) return r;
    HIPCHK(c, hipMemcpyAsync(c->d_scratch, pts, (size_t)n * sizeof(ssm_point), hipMemcpyHostToDevice, c->stream));
    // in chunks the table is grown for beforehand (every point of a chunk may open a voxel): nothing can overflow
    for (int a = 0; a < n; ) {
        int64_t chunk = ((int64_t)1 << c->map.cap_log2) / 8; if (chunk < 4096) chunk = 4096; if (chunk > n - a) chunk = n - a;
        r = map_settle(c, c->stream, chunk); if (r) return r;
        HIPCHK(c, k_voxel_insert(reinterpret_cast<ssm_point*>(c->d_scratch) + a, nullptr, chunk, (float)c->cfg.mapper_resolution, c->map.tab, c->map.cap_log2, c->map.counters, c->stream));
        a += (int)chunk;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_device_flags(c, true);
}
extern "C" int ssm_map_size(ssm_ctx* c, int* n)
{
    if (!c || !n) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    return table_count(c, c->map, n);
}
extern "C" int ssm_map_export(ssm_ctx* c, ssm_point* out, int cap, int* n_out)
{
    if (!c || !n_out) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    return table_export_points(c, c->map, out, cap, n_out);
}
extern "C" int ssm_map_export_table(ssm_ctx* c, ssm_voxel* out, int cap, int* n_out)
{
    if (!c || !n_out) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    int n; ssm_voxel* comp; uint32_t* order;
    int r = table_sorted(c, c->map, &n, &comp, &order); if (r) return r;
    *n_out = n;
    if (n > cap) FAIL(c, SSM_E_CAPACITY, "table buffer too small (need " + std::to_string(n) + ")");
    if (n == 0) return SSM_OK;
    r = ensure_scratch(c, (size_t)n * sizeof(ssm_voxel)); if (r) return r;
    HIPCHK(c, k_voxel_gather_table(comp, order, n, reinterpret_cast<ssm_voxel*>(c->d_scratch), c->stream));
    HIPCHK(c, hipMemcpyAsync(out, c->d_scratch, (size_t)n * sizeof(ssm_voxel), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SSM_OK;
}
extern "C" int ssm_map_merge_table(ssm_ctx* c, const ssm_voxel* tab, int n)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (n < 0 || (n && !tab)) FAIL(c, SSM_E_INVAL, "bad arguments");
    if (n == 0) return SSM_OK;
    int r = ensure_scratch(c, (size_t)n * sizeof(ssm_voxel)); if (r) return r;
    HIPCHK(c, hipMemcpyAsync(c->d_scratch, tab, (size_t)n * sizeof(ssm_voxel), hipMemcpyHostToDevice, c->stream));
    r = map_settle(c, c->stream, n); if (r) return r;               // room for n new voxels first
    HIPCHK(c, k_voxel_merge(reinterpret_cast<ssm_voxel*>(c->d_scratch), n, c->map.tab, c->map.cap_log2, c->map.counters, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SSM_OK;
}
extern "C" int ssm_map_export_table_dev(ssm_ctx* c, ssm_voxel* out, int cap, int* n_out)
{
    if (!c || !n_out) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    int n; ssm_voxel* comp; uint32_t* order;
    int r = table_sorted(c, c->map, &n, &comp, &order); if (r) return r;
    *n_out = n;
    if (n > cap) FAIL(c, SSM_E_CAPACITY, "table buffer too small (need " + std::to_string(n) + ")");
    if (n == 0) return SSM_OK;
    if (!out) FAIL(c, SSM_E_INVAL, "null output");
    HIPCHK(c, k_voxel_gather_table(comp, order, n, out, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SSM_OK;
}
extern "C" int ssm_map_merge_table_dev(ssm_ctx* c, const ssm_voxel* tab, int n)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (n < 0 || (n && !tab)) FAIL(c, SSM_E_INVAL, "bad arguments");
    { const int r = map_settle(c, c->stream, n); if (r) return r; }
    HIPCHK(c, k_voxel_merge(tab, n, c->map.tab, c->map.cap_log2, c->map.counters, c->stream));
    return SSM_OK;
}
// ---------------------------------------------------------------- multi-GPU: one process per GPU, the voxel-map merge is the only collective
#define NCCLCHK(ctx, expr) do { ncclResult_t e__ = (expr); if (e__ != ncclSuccess) { (ctx)->err = std::string(#expr) + ": " + ncclGetErrorString(e__); return SSM_E_COMM; } } while (0)
extern "C" int ssm_comm_get_unique_id(void* id)
{
    static_assert(sizeof(ncclUniqueId) == SSM_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id) return SSM_E_INVAL;
    ncclUniqueId u;
    ncclResult_t e = ncclGetUniqueId(&u);
    if (e != ncclSuccess) { g_create_err = std::string("ncclGetUniqueId: ") + ncclGetErrorString(e); return SSM_E_COMM; }
    memcpy(id, &u, sizeof(u));
    return SSM_OK;
}
extern "C" int ssm_comm_init_rank(ssm_ctx* c, int nranks, int rank, const void* id)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) FAIL(c, SSM_E_INVAL, "bad communicator arguments");
    if (c->comm) FAIL(c, SSM_E_INVAL, "the context already has a communicator (ssm_comm_finalize first)");
    ncclUniqueId u; memcpy(&u, id, sizeof(u));
    NCCLCHK(c, ncclCommInitRank(&c->comm, nranks, u, rank));
    c->comm_rank = rank; c->comm_size = nranks;
    return SSM_OK;
}
extern "C" int ssm_comm_finalize(ssm_ctx* c)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (c->comm) { HIPCHK(c, hipStreamSynchronize(c->stream)); NCCLCHK(c, ncclCommDestroy(c->comm)); c->comm = nullptr; }
    c->comm_rank = 0; c->comm_size = 1;
    return SSM_OK;
}
extern "C" int ssm_comm_rank(const ssm_ctx* c) { return c ? c->comm_rank : 0; }
extern "C" int ssm_comm_size(const ssm_ctx* c) { return c ? c->comm_size : 1; }
// SURVEY.md s.8e "collective": (1) all-gather of the per-rank voxel counts, (2) ONE all-gather of the tables padded to the longest
// (in place: a rank compacts its own table straight into its slot of the receive buffer), (3) every rank re-inserts the nranks-1
// remote tables.  Everything runs on the context stream; the one host wait is for the counts (they size the buffer).  Exact integer
// sums (DESIGN.md "voxel sums") make the result independent of rank order: every rank ends with the bit-identical 1-GPU map.
extern "C" int ssm_voxel_allgather(ssm_ctx* c, void* rccl_comm)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    ncclComm_t comm = rccl_comm ? reinterpret_cast<ncclComm_t>(rccl_comm) : c->comm;
    if (!comm) FAIL(c, SSM_E_INVAL, "no communicator: pass a ncclComm_t or call ssm_comm_init_rank");
    int world = 0, rank = 0;
    NCCLCHK(c, ncclCommCount(comm, &world)); NCCLCHK(c, ncclCommUserRank(comm, &rank));
    if (world > c->comm_counts_cap) {     // 2 ints per rank + one word of this rank's own flag
        if (c->d_comm_counts) { HIPCHK(c, hipStreamSynchronize(c->stream)); hipFree(c->d_comm_counts); c->d_comm_counts = nullptr; c->comm_counts_cap = 0; }
        DALLOC(c, c->d_comm_counts, (size_t)2 * world + 4); c->comm_counts_cap = world;
    }
    hipStream_t s = c->stream;
    // the local map at rest first (overflow list merged).  A rank that cannot settle must not leave before the collectives: it raises its map's LOST flag, which the
    // count all-gather below carries to every rank
    const int r_settle = map_settle(c, s, 0);
    if (r_settle) { const int32_t one = 1; HIPCHK(c, hipMemcpyAsync(c->map.counters + 1, &one, 4, hipMemcpyHostToDevice, s)); HIPCHK(c, hipStreamSynchronize(s)); }
    prof_begin(c, "allgather");
    // Every decision that can end the call is taken COLLECTIVELY: a rank that returned between two collectives would leave its peers blocked in the
    // next one.  (1) all-gather {voxel count, flag word} per rank -- counters[0..1] of the map table, already on the device.
    NCCLCHK(c, ncclAllGather(c->map.counters, c->d_comm_counts, 2, ncclInt32, comm, s));
    std::vector<int32_t> cf((size_t)2 * world);
    HIPCHK(c, hipMemcpyAsync(cf.data(), c->d_comm_counts, (size_t)world * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    std::vector<int32_t> counts(world);
    int mx = 1, bad_rank = -1, neg_rank = -1;
    for (int q = 0; q < world; q++) { counts[q] = cf[2 * q]; if (cf[2 * q + 1] & 1) bad_rank = q; if (counts[q] < 0) neg_rank = q; if (counts[q] > mx) mx = counts[q]; }
    if (bad_rank >= 0) { prof_end(c); FAIL(c, SSM_E_CAPACITY, "voxel table of rank " + std::to_string(bad_rank) + " is incomplete (contributions were dropped, or it could not be settled); no rank merged"); }
    if (neg_rank >= 0) { prof_end(c); FAIL(c, SSM_E_COMM, "negative voxel count received from rank " + std::to_string(neg_rank)); }
    // (2) the receive buffer: slot r = rank r's voxels, mx entries each.  An allocation failure on one rank is agreed on by a second tiny all-gather.
    const size_t slot = (size_t)mx * sizeof(ssm_voxel);
    int r_alloc = ensure_scratch2(c, slot * world + 256);
    if (r_alloc == SSM_OK) { int64_t remote = 0; for (int q = 0; q < world; q++) if (q != rank) remote += counts[q]; r_alloc = map_settle(c, s, remote); }   // room for every remote voxel: the merges below cannot overflow
    {
        const int32_t ok = r_alloc == SSM_OK ? 0 : 1;
        HIPCHK(c, hipMemcpyAsync(c->d_comm_counts + 2 * world, &ok, 4, hipMemcpyHostToDevice, s));
        NCCLCHK(c, ncclAllGather(c->d_comm_counts + 2 * world, c->d_comm_counts, 1, ncclInt32, comm, s));
        HIPCHK(c, hipMemcpyAsync(cf.data(), c->d_comm_counts, (size_t)world * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        for (int q = 0; q < world; q++) if (cf[q]) {
            prof_end(c);
            if (r_alloc) return r_alloc;
            FAIL(c, SSM_E_NOMEM, "rank " + std::to_string(q) + " could not allocate the all-gather buffer; no rank merged");
        }
    }
    uint8_t* recv = reinterpret_cast<uint8_t*>(c->d_scratch2);
    int32_t* dn = reinterpret_cast<int32_t*>(recv + slot * world);
    HIPCHK(c, k_voxel_compact(c->map.tab, c->map.cap_log2, reinterpret_cast<ssm_voxel*>(recv + slot * rank), dn, s));
    NCCLCHK(c, ncclAllGather(recv + slot * rank, recv, slot, ncclUint8, comm, s));
    // (3) merge the remote tables into the local map
    for (int q = 0; q < world; q++) {
        if (q == rank) continue;
        HIPCHK(c, k_voxel_merge(reinterpret_cast<const ssm_voxel*>(recv + slot * q), counts[q], c->map.tab, c->map.cap_log2, c->map.counters, s));
    }
    prof_end(c);
    return SSM_OK;
}
static inline float ord2f(int i) { i = i >= 0 ? i : i ^ 0x7FFFFFFF; float f; memcpy(&f, &i, 4); return f; }
extern "C" int ssm_voxel_filter(ssm_ctx* c, const ssm_point* pts, int n, float leaf, ssm_point* out, int cap, int* n_out)
{
    if (!c || !n_out) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (n < 0 || (n && !pts) || !(leaf > 0)) FAIL(c, SSM_E_INVAL, "bad arguments");
    *n_out = 0;
    if (n == 0) return SSM_OK;
    int r;
    if (!c->tmp.tab) { r = table_alloc(c, c->tmp, c->cfg.voxel_capacity_log2); if (r) return r; }
    else HIPCHK(c, k_voxel_clear(c->tmp.tab, c->tmp.cap_log2, c->tmp.counters, c->stream));
    r = ensure_scratch(c, (size_t)n * sizeof(ssm_point) + 64); if (r) return r;
    ssm_point* dp = reinterpret_cast<ssm_point*>(c->d_scratch);
    float* mm = reinterpret_cast<float*>(dp + n);
    HIPCHK(c, hipMemcpyAsync(dp, pts, (size_t)n * sizeof(ssm_point), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, k_voxel_bounds(dp, n, mm, c->stream));
    int ord[6];
    HIPCHK(c, hipMemcpyAsync(ord, mm, 24, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {   // pcl::VoxelGrid::applyFilter overflow guard: (dx*dy*dz) > INT_MAX -> warning, output = input
        const float inv = 1.0f / leaf;
        const int64_t dx = (int64_t)((ord2f(ord[3]) - ord2f(ord[0])) * inv) + 1, dy = (int64_t)((ord2f(ord[4]) - ord2f(ord[1])) * inv) + 1,
                      dz = (int64_t)((ord2f(ord[5]) - ord2f(ord[2])) * inv) + 1;
        if (dx * dy * dz > (int64_t)2147483647) FAIL(c, SSM_E_VOXEL_RANGE, "leaf size too small for the cloud extent (PCL would return the input unfiltered)");
    }
    // pcl::VoxelGrid has no table to overflow: when the temporary table fills up, re-allocate it four times as large and insert again
    for (;;) {
        HIPCHK(c, k_voxel_insert(dp, nullptr, n, leaf, c->tmp.tab, c->tmp.cap_log2, c->tmp.counters, c->stream));
        int32_t cnt[2];
        HIPCHK(c, hipMemcpyAsync(cnt, c->tmp.counters, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!(cnt[1] & 1)) break;
        const int bigger = c->tmp.cap_log2 + 2;
        if (bigger > 28) FAIL(c, SSM_E_CAPACITY, "voxel_filter: more than 2^28 voxels");
        hipFree(c->tmp.tab); c->tmp.tab = nullptr;
        r = table_alloc(c, c->tmp, bigger); if (r) return r;
    }
    return table_export_points(c, c->tmp, out, cap, n_out);
}

// ---------------------------------------------------------------- device-resident Mapper (ssm_backproject_dev, ssm_viewer_map_*)
struct ssm_cloud { ssm_point* d = nullptr; int n = 0; int device = 0; int slab = -1; };
extern "C" int ssm_backproject_dev(ssm_ctx* c, const uint16_t* depth, const uint8_t* rgb, const uint8_t* sem, int w, int h,
                                   const ssm_camera* cam, double max_distance, ssm_cloud** cloud_out)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!depth || !rgb || !sem || !cam || !cloud_out) FAIL(c, SSM_E_INVAL, "null argument");
    if (w != c->g.W || h != c->g.H) FAIL(c, SSM_E_INVAL, "frame size differs from the context configuration");
    *cloud_out = nullptr;
    const size_t np = (size_t)w * h;
    // one pinned staging area, one host-to-device copy for the three images (a pageable copy is staged by the runtime in small pieces)
    int r = ensure_pinned(c, np * 8); if (r) return r;
    memcpy(c->h_pinned, depth, np * 2); memcpy(c->h_pinned + np * 2, rgb, np * 3); memcpy(c->h_pinned + np * 5, sem, np * 3);
    r = ensure_scratch(c, np * 8); if (r) return r;
    uint8_t* din = reinterpret_cast<uint8_t*>(c->d_scratch);
    HIPCHK(c, hipMemcpyAsync(din, c->h_pinned, np * 8, hipMemcpyHostToDevice, c->stream));
    const uint16_t* dd = reinterpret_cast<const uint16_t*>(din); const uint8_t* drgb = din + np * 2; const uint8_t* dsem = din + np * 5;
    // the cloud is written straight into a slab of device memory (room for the worst case, w h points; only the n points made are kept): no allocation, no
    // device-to-device copy and ONE wait per key-frame
    int si = -1;
    for (size_t i = 0; i < c->cloud_slabs.size(); i++) if (c->cloud_slabs[i].cap - c->cloud_slabs[i].used >= np) { si = (int)i; break; }
    if (si < 0) {
        ssm_ctx::CloudSlab sl; sl.cap = np * 8 > ((size_t)2 << 20) ? np * 8 : ((size_t)2 << 20);          // >= 64 MB of points
        if (hipMalloc((void**)&sl.d, sl.cap * sizeof(ssm_point)) != hipSuccess) FAIL(c, SSM_E_HIP, "hipMalloc of a key-frame cloud slab failed");
        c->cloud_slabs.push_back(sl); si = (int)c->cloud_slabs.size() - 1;
    }
    ssm_ctx::CloudSlab& sl = c->cloud_slabs[si];
    ssm_point* dst = sl.d + sl.used;
    HIPCHK(c, k_moving_mask(dsem, 1, w, h, c->d_mask, c->stream));
    HIPCHK(c, k_backproject(dd, drgb, dsem, c->d_mask, nullptr, 1, w, h, *cam, max_distance,
                            c->d_chunk_cnt, c->d_chunk_off, reinterpret_cast<int32_t*>(c->d_total + 1), c->d_total, dst, c->stream));
    int64_t* h_total = reinterpret_cast<int64_t*>(c->h_pinned);                  // (the staged images at the front of the pinned area are consumed by then: stream order)
    HIPCHK(c, hipMemcpyAsync(h_total, c->d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int64_t total = *h_total;
    ssm_cloud* cl = new ssm_cloud(); cl->n = (int)total; cl->device = c->device; cl->slab = si; cl->d = dst;
    sl.used += ((size_t)total + 7) & ~(size_t)7; sl.live++;
    *cloud_out = cl;
    return SSM_OK;
}
extern "C" int ssm_cloud_size(const ssm_cloud* cl) { return cl ? cl->n : 0; }
extern "C" void ssm_cloud_free(ssm_ctx* c, ssm_cloud* cl)
{
    if (!cl) return;
    if (c) {                                                                   // a slab whose clouds are all freed is reused from its start
        std::lock_guard<std::mutex> lk(c->mu);
        if (cl->slab >= 0 && cl->slab < (int)c->cloud_slabs.size()) { ssm_ctx::CloudSlab& sl = c->cloud_slabs[cl->slab]; if (--sl.live == 0) { hipSetDevice(c->device); hipStreamSynchronize(c->stream); sl.used = 0; } }
    }
    delete cl;                                                                  // (without a context the slab goes with ssm_destroy)
}
extern "C" int ssm_cloud_fetch(ssm_ctx* c, const ssm_cloud* cl, const double* T, ssm_point* out, int cap, int* n_out)
{
    if (!c || !cl || !n_out) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    *n_out = cl->n;
    if (cl->n > cap) FAIL(c, SSM_E_CAPACITY, "point buffer too small (need " + std::to_string(cl->n) + ")");
    if (cl->n == 0) return SSM_OK;
    if (!out) FAIL(c, SSM_E_INVAL, "null argument");
    int r = ensure_scratch(c, (size_t)cl->n * sizeof(ssm_point)); if (r) return r;
    HIPCHK(c, k_cloud_transform(cl->d, cl->n, T, reinterpret_cast<ssm_point*>(c->d_scratch), c->stream));
    HIPCHK(c, hipMemcpyAsync(out, c->d_scratch, (size_t)cl->n * sizeof(ssm_point), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SSM_OK;
}
static int grow_points(ssm_ctx* c, ssm_point*& p, size_t& cap, size_t need, size_t keep)
{
    if (need <= cap) return SSM_OK;
    const size_t ncap = need + need / 2 + 1024;
    ssm_point* q = nullptr;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (hipMalloc(&q, ncap * sizeof(ssm_point)) != hipSuccess) FAIL(c, SSM_E_HIP, "hipMalloc of the viewer map failed");
    if (p && keep) HIPCHK(c, hipMemcpy(q, p, keep * sizeof(ssm_point), hipMemcpyDeviceToDevice));
    if (p) hipFree(p);
    p = q; cap = ncap;
    return SSM_OK;
}
extern "C" int ssm_viewer_map_update(ssm_ctx* c, int rebuild, ssm_cloud* const* clouds, const double* poses, int n, float leaf, int* n_map_out)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (n < 0 || (n && (!clouds || !poses)) || !(leaf > 0)) FAIL(c, SSM_E_INVAL, "bad arguments");
    size_t total = rebuild ? 0 : (size_t)c->vmap_n;
    for (int i = 0; i < n; i++) { if (!clouds[i]) FAIL(c, SSM_E_INVAL, "null cloud"); if (clouds[i]->device != c->device) FAIL(c, SSM_E_INVAL, "cloud of another device"); total += (size_t)clouds[i]->n; }
    if (total > (size_t)0x7FFFFFFF) FAIL(c, SSM_E_CAPACITY, "more than 2^31 points in one map update");
    if (total == 0) { c->vmap_n = 0; if (n_map_out) *n_map_out = 0; return SSM_OK; }
    int r = grow_points(c, c->d_vcat, c->vcat_cap, total + 8, 0); if (r) return r;       // (+ 8 points: the bounds words behind the data)
    // previous centroids, then every cloud transformed by its pose: the viewer's `*map += *generatePointCloud(kf)`
    size_t off = 0;
    if (!rebuild && c->vmap_n) { HIPCHK(c, hipMemcpyAsync(c->d_vcat, c->d_vmap, (size_t)c->vmap_n * sizeof(ssm_point), hipMemcpyDeviceToDevice, c->stream)); off = (size_t)c->vmap_n; }
    for (int i = 0; i < n; i++) { HIPCHK(c, k_cloud_transform(clouds[i]->d, clouds[i]->n, poses + (size_t)16 * i, c->d_vcat + off, c->stream)); off += (size_t)clouds[i]->n; }
    const int N = (int)total;
    if (!c->tmp.tab) { r = table_alloc(c, c->tmp, c->cfg.voxel_capacity_log2); if (r) return r; }
    else HIPCHK(c, k_voxel_clear(c->tmp.tab, c->tmp.cap_log2, c->tmp.counters, c->stream));
    float* mm = reinterpret_cast<float*>(c->d_vcat + total);
    HIPCHK(c, k_voxel_bounds(c->d_vcat, N, mm, c->stream));
    int ord[6];
    HIPCHK(c, hipMemcpyAsync(ord, mm, 24, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {   // pcl::VoxelGrid::applyFilter overflow guard (as in ssm_voxel_filter): the map is then the unfiltered concatenation
        const float inv = 1.0f / leaf;
        const int64_t dx = (int64_t)((ord2f(ord[3]) - ord2f(ord[0])) * inv) + 1, dy = (int64_t)((ord2f(ord[4]) - ord2f(ord[1])) * inv) + 1,
                      dz = (int64_t)((ord2f(ord[5]) - ord2f(ord[2])) * inv) + 1;
        if (dx * dy * dz > (int64_t)2147483647) {
            r = grow_points(c, c->d_vmap, c->vmap_cap, total, 0); if (r) return r;
            HIPCHK(c, hipMemcpyAsync(c->d_vmap, c->d_vcat, total * sizeof(ssm_point), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            c->vmap_n = N; if (n_map_out) *n_map_out = N;
            return SSM_OK;
        }
    }
    for (;;) {
        HIPCHK(c, k_voxel_insert(c->d_vcat, nullptr, N, leaf, c->tmp.tab, c->tmp.cap_log2, c->tmp.counters, c->stream));
        int32_t cnt[2];
        HIPCHK(c, hipMemcpyAsync(cnt, c->tmp.counters, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!(cnt[1] & 1)) break;
        const int bigger = c->tmp.cap_log2 + 2;
        if (bigger > 28) FAIL(c, SSM_E_CAPACITY, "viewer map: more than 2^28 voxels");
        hipFree(c->tmp.tab); c->tmp.tab = nullptr;
        r = table_alloc(c, c->tmp, bigger); if (r) return r;
    }
    int nv; ssm_voxel* comp; uint32_t* order;
    r = table_sorted(c, c->tmp, &nv, &comp, &order); if (r) return r;
    r = grow_points(c, c->d_vmap, c->vmap_cap, (size_t)nv, 0); if (r) return r;
    if (nv) HIPCHK(c, k_voxel_gather_points(comp, order, nv, c->d_vmap, c->stream));
    c->vmap_n = nv;
    if (n_map_out) *n_map_out = nv;
    return SSM_OK;
}
extern "C" int ssm_viewer_map_fetch(ssm_ctx* c, ssm_point* out, int cap, int* n_out)
{
    if (!c || !n_out) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    *n_out = c->vmap_n;
    if (c->vmap_n > cap) FAIL(c, SSM_E_CAPACITY, "point buffer too small (need " + std::to_string(c->vmap_n) + ")");
    if (c->vmap_n == 0) return SSM_OK;
    if (!out) FAIL(c, SSM_E_INVAL, "null argument");
    HIPCHK(c, hipMemcpyAsync(out, c->d_vmap, (size_t)c->vmap_n * sizeof(ssm_point), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SSM_OK;
}

// ---------------------------------------------------------------- device-resident sequence path
static int seg_init(ssm_ctx* c);
static int seg_forward_dev(ssm_ctx* c, const uint8_t* bgr, int n, uint8_t* labels_net, uint8_t* sem_bgr, int flags);
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
            uint8_t* mask_ws = map3 ? c->d_mask3 : c->d_mask;
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
                r = map_before_launch(c, s); if (r) return r;
                nq = map_frames_per_launch(c, nb - q0);
                const int g0 = f0 + q0;
                const uint8_t* sem_q = sem_src + (size_t)q0 * npix * 3;
                if ((W & 15) == 0) {         // streaming fused kernels (16 pixels per thread, 16-byte loads)
                    prof_begin(c, "map_fuse");
                    HIPCHK(c, k_map_fuse(in->depth + (size_t)g0 * npix, in->bgr + (size_t)g0 * npix * 3, sem_q,
                                         in->pose ? in->pose + (size_t)g0 * 16 : nullptr, nq, W, H, c->cfg.camera, c->cfg.mapper_max_distance,
                                         (float)c->cfg.mapper_resolution, reinterpret_cast<uint16_t*>(mask_ws), reinterpret_cast<uint16_t*>(mask_ws) + (size_t)nq * (W >> 4) * H,
                                         c->map.tab, c->map.cap_log2, c->map.counters, c->d_npoints + g0, s, c->map_compact));
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
                r = map_after_launch(c, s); if (r) return r;
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
    c->prev_n = n;
    if (out) {
        out->kps = c->d_kps; out->desc = desc; out->pos3d = c->d_pos3d; out->nkp = nkp; out->matches = c->d_matches; out->nmatch = c->d_nmatch;
        out->npoints = c->d_npoints; out->cap = g.cap; out->R = R;
    }
    return SSM_OK;
}


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
static int seg_init(ssm_ctx* c)
{
    if (c->seg) return SSM_OK;
    SegNetState* g = new SegNetState();
    c->seg = g;
    for (int l = 0; l < SEG_LAYERS; l++) {
        g->cinp[l] = k_seg_layers[l].cin <= 8 ? 8 : (k_seg_layers[l].cin + 63) & ~63;   // <= 8 channels: the first-layer kernel ([H][W][8] input)
        g->coutp[l] = (k_seg_layers[l].cout + 63) & ~63;
        g->coutstore[l] = (k_seg_layers[l].cout + 31) & ~31;            // activations live in 32-channel chunks: [C/32][H][W][32]
    }
    {   // frames per SegNet launch: 64 by default (more tiles per launch: better balance in the small layers; SSM_SEGNET_BATCH for experiments; the 64-channel layers bound it to 96 by their 2^31-byte buffers)
        const char* e = getenv("SSM_SEGNET_BATCH"); int sb = e ? atoi(e) : 64; if (sb < 1) sb = 1; if (sb > 96) sb = 96;
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
        HIPCHK(c, k_segnet_conv(cur, g->w[l], g->scale[l], g->shift[l], nxt, nb, d.h, d.w, g->cinp[l], d.cout, l != SEG_LAYERS - 1, s));
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
static int seg_forward_dev(ssm_ctx* c, const uint8_t* bgr, int n, uint8_t* labels_net, uint8_t* sem_bgr, int flags)
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
        HIPCHK(c, k_segnet_conv(g->actA, g->w[arg], g->scale[arg], g->shift[arg], g->actB, 1, H, W, g->cinp[arg], k_seg_layers[arg].cout, arg != SEG_LAYERS - 1, s));
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


// ---------------------------------------------------------------- stereo path: QuadFeatureMatch, StereoSGBM depth, VisualOdometryStereo
static void stereo_free(StereoState* q)
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
    const size_t need = k_sgbm_workspace_bytes(q->w, q->h, p, nb);
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
static int stereo_seq_run(ssm_ctx* c, const ssm_stereo_frames_dev* in, ssm_stereo_out_dev* out)
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
            HIPCHK(c, k_quad_gftt(qb, nb, maxc, 0.04, 8.0, q->gw, q->pts, maxc, q->ncorner, sq));      // quadmatcher.cpp:301-308
            HIPCHK(c, k_quad_track(qb, nb, q->pts, maxc, q->ncorner, q->has_prev, q->quad + (size_t)f0 * maxc, q->nquad + f0, sq));
            HIPCHK(c, hipMemcpyAsync(q->corners + (size_t)f0 * maxc * 2, q->pts, (size_t)nb * maxc * 8, hipMemcpyDeviceToDevice, sq));
            HIPCHK(c, hipMemcpyAsync(q->ncorners + f0, q->ncorner, (size_t)nb * 4, hipMemcpyDeviceToDevice, sq));
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
            HIPCHK(c, k_sgbm(in->left + (size_t)f0 * np, in->right + (size_t)f0 * np, w, h, nb, in->sgbm, q->sg_wsN[alt], q->disp + (size_t)f0 * np, 0, sg, q->sg_fail + (f0 / B) % SG_FAIL_WORDS,
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
    r = stereo_seq_run(c, &in, &o); if (r) return r;
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
    HIPCHK(c, k_sgbm(dev, dev + (size_t)w * h, w, h, 1, *params, q->sg_wsN[0], q->disp, stage, c->stream, q->sg_fail, form, 1));
    prof_end(c);
    q->sg_pending.valid = false;                                  // (the staged pair is this call's: the host-pointer entry points repeat a timed-out sweep themselves)
    *d_disp_out = q->disp; *d_depth_out = q->depth;
    return SSM_OK;
}
// cv::StereoSGBM cannot fail (src/stereo.cpp:11-30); form 2's sweep can: its strips wait for each other, and when a hand-off exceeds its spin bound every block
// leaves mid-image with the sub-batch's fail word set.  Called with the streams drained: every sub-batch of the last sequence call whose word is set is computed again
// with form 1 (independent paths, no cross-block waits; the workspace holds its volumes anyway) from the caller's input images, so that the call's disparities and depths
// are the oracle's after all.  Reported through ssm_last_error (a note, the call succeeds) and counted in sgbm_fallbacks.
static int sgbm_recover(ssm_ctx* c)
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
        HIPCHK(c, k_sgbm(in.left + (size_t)f0 * np, in.right + (size_t)f0 * np, w, h, nb, in.sgbm, q->sg_wsN[0], q->disp + (size_t)f0 * np, 0, c->stream, q->sg_fail + bi % SG_FAIL_WORDS, 1, 1));
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
    int16_t* dd; uint16_t* ddepth;
    const size_t np = (size_t)w * h;
    for (int attempt = 0; ; attempt++) {                      // a sweep whose hand-off timed out is repeated once, in form 1 (no cross-block waits)
        int r = sgbm_run(c, left, right, w, h, stride, params, stage, &dd, &ddepth, attempt ? 1 : c->sgbm_form_cfg); if (r) return r;
        HIPCHK(c, hipMemcpyAsync(c->h_pinned + 2 * np, dd, np * 2, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->stereo->sg_fail, 4, hipMemcpyDeviceToHost, c->stream));      // (the staged input images at the front of the pinned area are consumed)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!sgbm_failed(c)) break;
        if (attempt) FAIL(c, SSM_E_HIP, "SGBM: the time-out word is set after a form-1 run");
        c->sgbm_fallbacks++;
    }
    memcpy(disp, c->h_pinned + 2 * np, np * 2);
    if (c->sgbm_fallbacks) c->err = "note: " + std::to_string(c->sgbm_fallbacks) + " SGBM sweep(s) of this context timed out in a strip hand-off and were repeated with form 1 (results complete)";
    return SSM_OK;
}
extern "C" int ssm_stereo_depth(ssm_ctx* c, const uint8_t* left, const uint8_t* right, int w, int h, int stride, const ssm_sgbm_params* params,
                                double baseline, double cu, double cv, double f, double roix, double roiy, double roiz, double scale,
                                uint16_t* depth, int16_t* disp)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (!depth) FAIL(c, SSM_E_INVAL, "null argument");
    int16_t* dd; uint16_t* ddepth;
    const size_t np = (size_t)w * h;
    for (int attempt = 0; ; attempt++) {
        int r = sgbm_run(c, left, right, w, h, stride, params, 0, &dd, &ddepth, attempt ? 1 : c->sgbm_form_cfg); if (r) return r;
        HIPCHK(c, k_sgbm_depth(dd, w, h, 1, baseline, cu, cv, f, roix, roiy, roiz, scale, c->stereo->dminN[0], ddepth, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_pinned + 4 * np, ddepth, np * 2, hipMemcpyDeviceToHost, c->stream));
        if (disp) HIPCHK(c, hipMemcpyAsync(c->h_pinned + 2 * np, dd, np * 2, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->stereo->sg_fail, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!sgbm_failed(c)) break;
        if (attempt) FAIL(c, SSM_E_HIP, "SGBM: the time-out word is set after a form-1 run");
        c->sgbm_fallbacks++;
    }
    memcpy(depth, c->h_pinned + 4 * np, np * 2);
    if (disp) memcpy(disp, c->h_pinned + 2 * np, np * 2);
    if (c->sgbm_fallbacks) c->err = "note: " + std::to_string(c->sgbm_fallbacks) + " SGBM sweep(s) of this context timed out in a strip hand-off and were repeated with form 1 (results complete)";
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

// PnPSolver::solvePnP (reference src/pnp.cpp:5-118) for one correspondence list: the block of kernels_pnp.hip that the pose chain runs per frame
extern "C" int ssm_pnp_solve(ssm_ctx* c, const float* img, const float* obj, int n, const double cam[4], int min_inliers, double T[16],
                             uint8_t* inliers, int* n_inliers, int* success)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (n < 0 || !cam || !T || !n_inliers || (n && (!img || !obj))) FAIL(c, SSM_E_INVAL, "bad arguments");
    if (n > 65535) FAIL(c, SSM_E_CAPACITY, "at most 65535 correspondences");
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t ne = (size_t)(n > 0 ? n : 1);
    const size_t o_img = 0, o_obj = o_img + al(ne * 8), o_T = o_obj + al(ne * 12), o_inl = o_T + 256, o_dec = o_inl + al(ne), o_le = o_dec + al(ne),
                 o_err = o_le + al(ne * k_pnp_edge_bytes()), o_n = o_err + al(ne * 16), total = o_n + 256;
    int r = ensure_scratch(c, total); if (r) return r;
    uint8_t* p = (uint8_t*)c->d_scratch; hipStream_t s = c->stream;
    if (n) { HIPCHK(c, hipMemcpyAsync(p + o_img, img, (size_t)n * 8, hipMemcpyHostToDevice, s)); HIPCHK(c, hipMemcpyAsync(p + o_obj, obj, (size_t)n * 12, hipMemcpyHostToDevice, s)); }
    HIPCHK(c, hipMemcpyAsync(p + o_T, T, 128, hipMemcpyHostToDevice, s));
    PnpSolveArgs a; a.img = (const float*)(p + o_img); a.obj = (const float*)(p + o_obj); a.n = n;
    a.cam.fx = cam[0]; a.cam.fy = cam[1]; a.cam.cx = cam[2]; a.cam.cy = cam[3];
    a.T = (double*)(p + o_T); a.inl = p + o_inl; a.dec = p + o_dec; a.ledges = (LEdge*)(p + o_le); a.err = (double2*)(p + o_err); a.n_inliers = (int32_t*)(p + o_n); a.edges_in_lds = 0;
    if (c->profiling) { c->recs.clear(); c->pool_used = 0; }
    prof_begin(c, "pnp");
    HIPCHK(c, k_pnp_solve(a, s));
    prof_end(c);
    int32_t m = 0;
    HIPCHK(c, hipMemcpyAsync(T, p + o_T, 128, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&m, p + o_n, 4, hipMemcpyDeviceToHost, s));
    if (inliers && n) HIPCHK(c, hipMemcpyAsync(inliers, p + o_inl, (size_t)n, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    *n_inliers = m;
    if (success) *success = n > min_inliers;                   // pnp.cpp:115 tests the flag vector's LENGTH (quirk 14)
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
