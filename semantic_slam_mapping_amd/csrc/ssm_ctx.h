// ssm_ctx.h -- the context of libssm_hip.so and what its host-side translation units share (ssm_abi.hip: lifecycle, ORB / matcher / sequence path; ssm_map.hip: voxel map,
// multi-GPU merge, device-resident Mapper; ssm_segnet_abi.hip: Classifier; ssm_stereo_abi.hip: quad matcher, SGBM depth, stereo VO, solvePnP).  Not installed.
#pragma once
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
#define SSM_HIDDEN __attribute__((visibility("hidden")))


struct VoxTable {           // tab[slots] | occ[slots] | counter block (32 bytes: count, flags, overflow records, overflow capacity, overflow list address)
    ssm_voxel* tab = nullptr; uint32_t* occ = nullptr; int32_t* counters = nullptr; int cap_log2 = 0;
    ssm_voxel* ovf = nullptr; int ovf_cap = 0;       // the overflow list of the context map (kernels_map.hip vox_overflow_slot); the temporary tables have none
    int32_t* skip = nullptr;                         // the context map's skip list (kernels_map.hip map_stream2_kernel): blocks of the fused map stage that found the overflow list beyond its high-water mark
    size_t bytes() const { const size_t s = (size_t)1 << cap_log2; return s * sizeof(ssm_voxel) + s * 4 + 32; }
};
static const int VOX_OVF_RECORDS = 1 << 18;          // 29 MB per context; a context that runs the fused map stage (ssm_seq_process) trades it for the large list of map_ensure_stream_list
// one launch of the fused map stage, kept so that blocks it skipped can be run again (ssm_map.hip map_redo)
struct MapLaunch { const uint16_t* depth; const uint8_t* rgb; const uint8_t* sem; const double* pose; int n, w, h; int32_t* npoints; bool valid; };
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
    void* ww[SEG_LAYERS] = {};           // the layer's weights in Winograd F(2, 3) form (kernels_segnet.hip conv3x3_wino_kernel), or null: the direct kernel only
    int cinp[SEG_LAYERS], coutp[SEG_LAYERS], coutstore[SEG_LAYERS];
    int batch = 0;
    void *actA = nullptr, *actB = nullptr, *last_logits = nullptr; uint8_t* code[5] = {}; uint8_t* labels = nullptr;
    int32_t *pre_xofs = nullptr, *pre_yofs = nullptr, *post_xofs = nullptr, *post_yofs = nullptr;
    int16_t *pre_xa = nullptr, *pre_ya = nullptr, *post_xa = nullptr, *post_ya = nullptr;
    uint8_t* d_sem_gen = nullptr;       // generated colour labels for the sequence path (max_batch frames)
};


struct ssm_ctx {
    std::mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    bool side_ready = false;            // ensure_side_streams completed
    hipStream_t stream2 = nullptr;      // ssm_seq_process: the SegNet + map stage of a sub-batch runs here, beside the ORB + match chain
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t stream3 = nullptr; hipEvent_t ev_join3 = nullptr; int map_stream = 1;   // two-chain mode: the map stage on a stream of its own (SSM_MAP_STREAM=0: on the chain's stream)
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
    bool viewer_fail_next = false;                                           // tests: the next ssm_viewer_map_update fails (ssm_viewer_map_release)
    std::vector<CloudSlab> cloud_slabs;                                      // key-frame clouds (ssm_backproject_dev) are carved from slabs: no hipMalloc per cloud
    // staging for the host-pointer entry points (one frame) + generic scratch
    uint8_t *d_in_img = nullptr, *d_in_sem = nullptr; uint16_t* d_in_depth = nullptr; double* d_in_pose = nullptr;
    void* d_scratch = nullptr; size_t scratch_bytes = 0;
    unsigned long long* d_pnp_xchg = nullptr; unsigned pnp_epoch = 0;    // ssm_pnp_solve's cluster: the exchange ring (persistent) and the launch number its pass tags start from
    bool pnp_solve_one_block = false;                        // ssm_pnp_solve: a cluster of eight blocks timed out once -> one block per solve from then on
    void* d_scratch2 = nullptr; size_t scratch2_bytes = 0;
    // the stream that holds the newest work on the context map when that is a side stream of ssm_seq_process (joined into `stream` by an event, so everything queued
    // on `stream` afterwards is ordered behind it): ssm_map_size / ssm_map_export_table_dev read the map there and wait for THAT stream only -- the ORB -> match chain of
    // the call's last sub-batch keeps running.  nullptr: the map's newest work is on `stream`.
    hipStream_t map_tail = nullptr;
    // sequence outputs
    int seq_cap = 0, prev_n = -1;
    ssm_keypoint* d_kps = nullptr; uint8_t* d_desc_all = nullptr; int32_t* d_nkp_all = nullptr; float* d_pos3d = nullptr;
    ssm_dmatch* d_matches = nullptr; int32_t* d_nmatch = nullptr; int32_t* d_match_pend = nullptr; int32_t* d_npoints = nullptr; uint8_t* d_hist_tmp = nullptr;
    uint8_t* d_exp_q = nullptr; uint8_t* d_exp_t = nullptr; uint8_t* d_knn = nullptr; int capT = 0; bool match_mfma = true; bool map_first = true;   // the matcher's expanded descriptor rows (kernels_match.hip)
    // voxel tables
    MapLaunch map_ring[8] = {}; unsigned map_ring_next = 0; int32_t* d_redo = nullptr; std::vector<int32_t> map_skipped;    // launches that may still have skipped blocks; ids to run again (host)
    std::vector<hipStream_t> map_launch_streams; long map_redone = 0;      // streams fused launches were queued on; blocks run again so far
    bool map_unexamined = false;                            // a fused map launch was queued since the counters were last read on a drained stream
    VoxTable map, tmp; bool map_full_reported = false;   // table-full already reported by check_device_flags (reset by ssm_map_clear)
    // the context map grows (map_settle); between the map launches of ssm_seq_process its counters come back through a two-slot ring of asynchronous copies
    int vox_max_log2 = 28; int32_t* h_map_snap = nullptr; hipEvent_t map_snap_ev[2] = {nullptr, nullptr}; uint64_t map_launches = 0; int map_grown = 0;
    double map_vpf = -1.0;                                  // voxels (+ overflow records) per fused frame, the largest rate seen on this context; < 0: none yet
    int64_t map_frames = 0, map_snap_frames[2] = {0, 0}, map_known_total = 0, map_known_frames = 0;     // frames fused since the last clear; at the ring's snapshots; the last count the host has seen and when
    // multi-GPU: the communicator of ssm_comm_init_rank (one rank per context / GPU) and the gathered counts
    ncclComm_t comm = nullptr; int comm_rank = 0, comm_size = 1; int32_t* d_comm_counts = nullptr; int comm_counts_cap = 0;
    // SegNet
    struct SegNetState* seg = nullptr;
    // quad matcher
    struct StereoState* stereo = nullptr; int stereo_B = 16; int stereo_sgbm_streams = 2;      // ssm_config.sgbm_streams 
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

inline int cv_round_f(float v) { return (int)lrint((double)v); }

#define DALLOC(ctx, p, n) do { int r__ = dalloc(ctx, &(p), (size_t)(n)); if (r__) return r__; } while (0)
#define NCCLCHK(ctx, expr) do { ncclResult_t e__ = (expr); if (e__ != ncclSuccess) { (ctx)->err = std::string(#expr) + ": " + ncclGetErrorString(e__); return SSM_E_COMM; } } while (0)
template <class T> inline int dalloc(ssm_ctx* c, T** p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void**)p, count * sizeof(T));
    if (e != hipSuccess) { c->err = std::string("hipMalloc(") + std::to_string(count * sizeof(T)) + "): " + hipGetErrorString(e); return SSM_E_NOMEM; }
    return SSM_OK;
}
// ssm_abi.hip
extern SSM_HIDDEN thread_local std::string g_create_err;
SSM_HIDDEN void resize_tables(int ssize, int dsize, std::vector<int32_t>& ofs, std::vector<int16_t>& coef);
SSM_HIDDEN int ensure_scratch(ssm_ctx* c, size_t bytes);
SSM_HIDDEN int ensure_pinned(ssm_ctx* c, size_t bytes);
SSM_HIDDEN int ensure_scratch2(ssm_ctx* c, size_t bytes);
SSM_HIDDEN void prof_begin(ssm_ctx* c, const char* name);
SSM_HIDDEN void prof_end(ssm_ctx* c);
SSM_HIDDEN int check_device_flags(ssm_ctx* c, bool with_map);
SSM_HIDDEN int wait_pending(ssm_ctx* c);
SSM_HIDDEN bool host_is_pinned(const void* p);             // page-locked host memory (ssm_host_alloc, hipHostRegister)?
SSM_HIDDEN int ensure_side_streams(ssm_ctx* c);
// ssm_map.hip
SSM_HIDDEN int table_alloc(ssm_ctx* c, VoxTable& t, int cap_log2);
SSM_HIDDEN int map_settle(ssm_ctx* c, hipStream_t s, int64_t reserve, int32_t* counters_out = nullptr);
SSM_HIDDEN int map_before_launch(ssm_ctx* c, hipStream_t s, int remaining, int* nq);
SSM_HIDDEN int map_after_launch(ssm_ctx* c, hipStream_t s, int frames, bool inputs_volatile);
SSM_HIDDEN int map_fuse_launch(ssm_ctx* c, hipStream_t s, const MapLaunch& L);      // one launch of the fused map stage on the context map, recorded for a redo
// ssm_segnet_abi.hip
SSM_HIDDEN int seg_init(ssm_ctx* c);
SSM_HIDDEN int seg_forward_dev(ssm_ctx* c, const uint8_t* bgr, int n, uint8_t* labels_net, uint8_t* sem_bgr, int flags);
// ssm_stereo_abi.hip
SSM_HIDDEN void stereo_free(StereoState* q);
SSM_HIDDEN int sgbm_recover(ssm_ctx* c);
