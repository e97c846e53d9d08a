// ssm_track.hip -- ssm_tracker_*: rgbd_tutor::Tracker::updateFrame (reference src/track.cpp:8-36, 140-212) for all frames of an ssm_seq_process call.
// Host orchestration of the pose chain (the chain is serial by nature: frame f's initial value and its reference poses are frame f-1's results); the PnP
// arithmetic is include/ssm/pnp_core.h, the code the per-frame host class (include/ssm/pnp.h) runs, so both give the same bits.  Written against the
// public C ABI (ssm_match for the on-demand pairs, ssm_memcpy_d2h) -- no access to the context's internals.
#include "pnp_chain.h"
#include <deque>
#include <string>
#include <vector>
#include <cstring>

namespace {
struct RefFrame {                         // a member of Tracker::refFrames: what trackRefFrame reads of it
    int64_t gidx = 0; int nkp = 0; double pose[16];
    std::vector<float> pos3d; std::vector<uint8_t> desc;
};
}
struct ssm_tracker {
    ssm_ctx* ctx = nullptr; ssm_tracker_params prm{}; ssm_camera cam{}; double ratio = 0.8;
    std::string err;
    int state = 0, cnt_lost = 0;          // Tracker::trackerState: 0 NOT_READY, 1 OK, 2 LOST
    double speed[16], last_pose[16];
    std::deque<RefFrame> refs;
    int64_t next_gidx = 0;
    // host copies of one call's outputs
    std::vector<int32_t> nkp, nmatch; std::vector<ssm_keypoint> kps; std::vector<float> pos3d; std::vector<uint8_t> desc; std::vector<ssm_dmatch> matches;
    std::vector<float> img, obj; std::vector<unsigned char> inl; std::vector<ssm_pnp::Edge> edges; std::vector<ssm_dmatch> tmp_matches;
    std::vector<uint8_t> have;            // per frame of the current call: bit 0 = features on the host, bit 1 = match tables on the host
    // device chain (use_device): scratch + the state block, allocated at first use
    PnpState* d_state = nullptr; double* d_pose = nullptr; ssm_track_info* d_info = nullptr; float *d_img = nullptr, *d_obj = nullptr, *d_hist = nullptr;
    uint8_t *d_inl = nullptr, *d_dec = nullptr; void* d_edges = nullptr; double2* d_err = nullptr; int d_cap = 0, d_R = 0, d_n = 0;
    unsigned long long* d_xchg = nullptr; int blocks = 1;      // the cluster form of the device chain (SSM_PNP_BLOCKS, kernels_pnp.hip): blocks per chain, their exchange ring
    long device_frames = 0, host_frames = 0;
    bool downgraded = false;              // the cluster form timed out once: one block per chain since (reported by ssm_tracker_last_error)
    int64_t work[4] = {0, 0, 0, 0};       // the device chain's passes over the edges (ssm_tracker_work)
    hipStream_t own = nullptr; hipEvent_t ev = nullptr;      // own_stream: the chain's stream and the event that orders it behind the context's stream
};
static void iso_identity(double* T) { for (int k = 0; k < 16; k++) T[k] = (k % 5 == 0) ? 1.0 : 0.0; }

extern "C" void ssm_tracker_params_default(ssm_tracker_params* p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->max_lost_frame = 10; p->ref_frames = 5; p->pnp_min_inliers = 10; p->use_device = 0;
    iso_identity(p->first_pose);
}
extern "C" int ssm_tracker_create(ssm_ctx* ctx, const ssm_tracker_params* p, ssm_tracker** out)
{
    if (!ctx || !p || !out) return SSM_E_INVAL;
    *out = nullptr;
    if (p->ref_frames < 1 || p->ref_frames > 64 || p->max_lost_frame < 0) return SSM_E_INVAL;
    ssm_tracker* t = new ssm_tracker();
    t->ctx = ctx; t->prm = *p;
    ssm_config cfg; ssm_internal_get_config(ctx, &cfg);
    t->cam = cfg.camera; t->ratio = cfg.knn_match_ratio;
    if (cfg.tracker_ref_frames != p->ref_frames) { delete t; return SSM_E_INVAL; }
    // blocks per device chain (kernels_pnp.hip, the cluster form): eight for a chain that has the GPU to itself (latency: -6.5 % per frame; four: -3.4 %); ONE for an
    // own_stream tracker -- those exist to run many chains side by side, where a CU per chain is the efficient form and the blocks of several clusters would
    // have to be resident together.  SSM_PNP_BLOCKS = 1 | 2 | 4 | 8 overrides (same bits in every form).
    { const char* e = getenv("SSM_PNP_BLOCKS"); const int d = p->own_stream ? 1 : 8, g = e ? atoi(e) : (p->blocks > 0 ? p->blocks : d); t->blocks = (g == 1 || g == 2 || g == 4 || g == 8) ? g : d; }
    (void)hipSetDevice(ssm_internal_get_device(ctx));              // the raw HIP calls of this file act on the context's device, whatever the calling thread used last
    if (p->own_stream && (hipStreamCreateWithFlags(&t->own, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&t->ev, hipEventDisableTiming) != hipSuccess)) {
        if (t->own) hipStreamDestroy(t->own);
        delete t; return SSM_E_HIP;
    }
    ssm_tracker_reset(t);
    *out = t;
    return SSM_OK;
}
static void tracker_free_device(ssm_tracker* t)
{
    void* p[] = { t->d_state, t->d_pose, t->d_info, t->d_img, t->d_obj, t->d_hist, t->d_inl, t->d_dec, t->d_edges, t->d_err, t->d_xchg };
    t->d_xchg = nullptr;
    for (void* x : p) if (x) hipFree(x);
    t->d_state = nullptr; t->d_pose = nullptr; t->d_info = nullptr; t->d_img = t->d_obj = t->d_hist = nullptr; t->d_inl = t->d_dec = nullptr; t->d_edges = nullptr; t->d_err = nullptr;
    t->d_cap = t->d_R = t->d_n = 0;
}
extern "C" void ssm_tracker_destroy(ssm_tracker* t)
{
    if (!t) return;
    (void)hipSetDevice(ssm_internal_get_device(t->ctx));
    if (t->own) { hipStreamSynchronize(t->own); hipStreamDestroy(t->own); }
    if (t->ev) hipEventDestroy(t->ev);
    tracker_free_device(t); delete t;
}
extern "C" const char* ssm_tracker_last_error(const ssm_tracker* t) { return t ? t->err.c_str() : "null tracker"; }
extern "C" int ssm_tracker_reset(ssm_tracker* t)
{
    if (!t) return SSM_E_INVAL;
    t->state = 0; t->cnt_lost = 0; t->refs.clear(); t->next_gidx = 0;
    iso_identity(t->speed); iso_identity(t->last_pose);
    return SSM_OK;
}
#define TFAIL(t, code, msg) do { (t)->err = (msg); return (code); } while (0)
#define TCHK(t, expr) do { int r__ = (expr); if (r__ != SSM_OK) { (t)->err = std::string(#expr) + ": " + ssm_last_error((t)->ctx); return r__; } } while (0)

static int tracker_ensure_device(ssm_tracker* t, int cap, int R, int n)
{
    if (t->d_state && t->d_cap == cap && t->d_R == R && t->d_n >= n) return SSM_OK;
    ssm_sync(t->ctx);
    tracker_free_device(t);
    const size_t mc = (size_t)R * cap, G = (size_t)t->blocks;          // G private slices of the state and of every scratch array (the cluster form)
    bool ok = hipMalloc((void**)&t->d_state, G * sizeof(PnpState)) == hipSuccess && hipMalloc((void**)&t->d_pose, (size_t)n * 128) == hipSuccess &&
              hipMalloc((void**)&t->d_info, (size_t)n * sizeof(ssm_track_info)) == hipSuccess && hipMalloc((void**)&t->d_img, G * mc * 8) == hipSuccess &&
              hipMalloc((void**)&t->d_obj, G * mc * 12) == hipSuccess && hipMalloc((void**)&t->d_hist, mc * 12) == hipSuccess && hipMalloc((void**)&t->d_inl, G * mc) == hipSuccess &&
              hipMalloc((void**)&t->d_dec, G * mc) == hipSuccess && hipMalloc(&t->d_edges, G * mc * k_pnp_edge_bytes()) == hipSuccess && hipMalloc((void**)&t->d_err, G * mc * sizeof(double2)) == hipSuccess &&
              hipMalloc((void**)&t->d_xchg, k_pnp_xchg_bytes()) == hipSuccess;
    if (!ok) { tracker_free_device(t); t->err = "device allocation for the pose chain failed"; return SSM_E_NOMEM; }
    t->d_cap = cap; t->d_R = R; t->d_n = n;
    return SSM_OK;
}
extern "C" int ssm_tracker_run(ssm_tracker* t, const ssm_seq_out_dev* seq, int n, double* pose_out, ssm_track_info* info_out)
{
    if (!t) return SSM_E_INVAL;
    if (!seq || n < 0 || (n && !pose_out)) TFAIL(t, SSM_E_INVAL, "bad arguments");
    if (n == 0) return SSM_OK;
    (void)hipSetDevice(ssm_internal_get_device(t->ctx));
    const int cap = seq->cap, R = seq->R;
    if (R != t->prm.ref_frames || R > SSM_TRACK_MAXREF) TFAIL(t, SSM_E_INVAL, "the sequence was matched with another tracker_ref_frames");
    const bool on_device = t->prm.use_device != 0;
    // ---- the call's outputs on the host: counts always; features and match tables in bulk (host chain) or per frame when the host path needs one
    t->nkp.resize(n); t->nmatch.resize((size_t)n * R); t->kps.resize((size_t)n * cap); t->pos3d.resize((size_t)n * cap * 3); t->desc.resize((size_t)n * cap * 32);
    t->matches.resize((size_t)n * R * cap); t->have.assign(n, 0);
    TCHK(t, ssm_sync(t->ctx));
    TCHK(t, ssm_memcpy_d2h(t->ctx, t->nkp.data(), seq->nkp, (size_t)n * 4));
    TCHK(t, ssm_memcpy_d2h(t->ctx, t->nmatch.data(), seq->nmatch, (size_t)n * R * 4));
    if (!on_device) {
        TCHK(t, ssm_memcpy_d2h(t->ctx, t->kps.data(), seq->kps, (size_t)n * cap * sizeof(ssm_keypoint)));
        TCHK(t, ssm_memcpy_d2h(t->ctx, t->pos3d.data(), seq->pos3d, (size_t)n * cap * 12));
        TCHK(t, ssm_memcpy_d2h(t->ctx, t->desc.data(), seq->desc, (size_t)n * cap * 32));
        TCHK(t, ssm_memcpy_d2h(t->ctx, t->matches.data(), seq->matches, (size_t)n * R * cap * sizeof(ssm_dmatch)));
        t->have.assign(n, 3);
    }
    auto need_features = [&](int f) -> int {
        if (t->have[f] & 1) return SSM_OK;
        const size_t k = (size_t)(t->nkp[f] > 0 ? t->nkp[f] : 0);
        if (k) {
            TCHK(t, ssm_memcpy_d2h(t->ctx, t->kps.data() + (size_t)f * cap, seq->kps + (size_t)f * cap, k * sizeof(ssm_keypoint)));
            TCHK(t, ssm_memcpy_d2h(t->ctx, t->pos3d.data() + (size_t)f * cap * 3, seq->pos3d + (size_t)f * cap * 3, k * 12));
            TCHK(t, ssm_memcpy_d2h(t->ctx, t->desc.data() + (size_t)f * cap * 32, seq->desc + (size_t)f * cap * 32, k * 32));
        }
        t->have[f] |= 1; return SSM_OK;
    };
    auto need_matches = [&](int f) -> int {
        if (t->have[f] & 2) return SSM_OK;
        TCHK(t, ssm_memcpy_d2h(t->ctx, t->matches.data() + (size_t)f * R * cap, seq->matches + (size_t)f * R * cap, (size_t)R * cap * sizeof(ssm_dmatch)));
        t->have[f] |= 2; return SSM_OK;
    };
    const size_t maxcorr = (size_t)R * cap;
    t->img.resize(2 * maxcorr + 2); t->obj.resize(3 * maxcorr + 3); t->inl.resize(maxcorr + 1); t->edges.resize(maxcorr + 1); t->tmp_matches.resize(cap);
    ssm_pnp::Camera cam; cam.fx = t->cam.fx; cam.fy = t->cam.fy; cam.cx = t->cam.cx; cam.cy = t->cam.cy;

    auto push_ref = [&](int f, const double* pose) -> int {  // refFrames.push_back(currentFrame); while (size > refFramesSize) pop_front()
        int r_ = need_features(f); if (r_) return r_;
        RefFrame r; r.gidx = t->next_gidx + f; r.nkp = t->nkp[f]; memcpy(r.pose, pose, sizeof(r.pose));
        r.pos3d.assign(t->pos3d.begin() + (size_t)f * cap * 3, t->pos3d.begin() + (size_t)f * cap * 3 + (size_t)r.nkp * 3);
        r.desc.assign(t->desc.begin() + (size_t)f * cap * 32, t->desc.begin() + (size_t)f * cap * 32 + (size_t)r.nkp * 32);
        t->refs.push_back(std::move(r));
        while ((int)t->refs.size() > t->prm.ref_frames) t->refs.pop_front();
        return SSM_OK;
    };
    // the deque is REGULAR at frame f when it is the run of frames directly in front of it: every member then has its precomputed match-table slot
    auto regular = [&](int f) {
        if (t->state != 1 || t->refs.empty()) return false;
        const int64_t G = t->next_gidx + f; const int k = (int)t->refs.size();
        for (int r = 0; r < k; r++) { if (t->refs[r].gidx != G - k + r) return false; if (t->nmatch[(size_t)f * R + (R - (k - r))] < 0 && t->nkp[f] >= 2) return false; }
        return true;
    };
    int f = 0;
    while (f < n) {
        if (on_device && regular(f)) {
            // ---- a run of frames on the device: state up, one launch, state and the run's poses down
            int rc = tracker_ensure_device(t, cap, R, n); if (rc) return rc;
            PnpState hs; memset(&hs, 0, sizeof(hs));
            memcpy(hs.speed, t->speed, 128); memcpy(hs.last_pose, t->last_pose, 128);
            hs.nref = (int)t->refs.size(); hs.cnt_lost = t->cnt_lost; hs.stopped_at = n;
            hipStream_t st = (hipStream_t)ssm_stream(t->ctx);
            if (t->own) {                                     // behind everything the context's stream holds now (the call that made `seq`), then on its own
                // (an idle context stream needs no hand-over -- and a marker in its hardware queue would wait behind another tracker's chain whenever the
                // runtime multiplexes the two streams onto one queue)
                if (hipStreamQuery(st) != hipSuccess && (hipEventRecord(t->ev, st) != hipSuccess || hipStreamWaitEvent(t->own, t->ev, 0) != hipSuccess)) TFAIL(t, SSM_E_HIP, "stream hand-over failed");
                st = t->own;
            }
            for (int r = 0; r < hs.nref; r++) {
                const int idx = (int)(t->refs[r].gidx - t->next_gidx);
                hs.ref_idx[r] = idx; memcpy(hs.ref_pose[r], t->refs[r].pose, 128);
                if (idx < 0 && t->refs[r].nkp > 0)            // a frame of the previous call: its positions are no longer on the device
                    if (hipMemcpyAsync(t->d_hist + (size_t)(idx + R) * cap * 3, t->refs[r].pos3d.data(), (size_t)t->refs[r].nkp * 12, hipMemcpyHostToDevice, st) != hipSuccess) TFAIL(t, SSM_E_HIP, "upload of the reference positions failed");
            }
            for (int b = 0; b < t->blocks; b++)
                if (hipMemcpyAsync(t->d_state + b, &hs, sizeof(hs), hipMemcpyHostToDevice, st) != hipSuccess) TFAIL(t, SSM_E_HIP, "upload of the tracker state failed");
            PnpChainArgs a; a.kps = seq->kps; a.pos3d = seq->pos3d; a.matches = seq->matches; a.nmatch = seq->nmatch; a.hist_pos3d = t->d_hist;
            a.cap = cap; a.R = R; a.f_begin = f; a.f_end = n; a.max_lost = t->prm.max_lost_frame; a.cam = cam;
            a.state = t->d_state; a.pose_out = t->d_pose; a.info_out = t->d_info; a.img = t->d_img; a.obj = t->d_obj; a.inl = t->d_inl; a.dec = t->d_dec; a.ledges = (LEdge*)t->d_edges; a.err = t->d_err; a.edges_in_lds = 0;
            a.blocks = t->blocks; a.xchg = t->d_xchg; a.xfail = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(t->d_xchg) + k_pnp_xchg_bytes() - 64);
            if (k_pnp_chain(a, st) != hipSuccess) TFAIL(t, SSM_E_HIP, "pose chain launch failed");
            if (hipMemcpyAsync(&hs, t->d_state, sizeof(hs), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) TFAIL(t, SSM_E_HIP, "pose chain failed");
            if (hs.stopped_at == -1 && t->blocks > 1) {
                // the blocks of the cluster did not meet within the spin bound (they need CUs at the same time: a device kept full by other work for seconds).
                // Nothing of the tracker's host state has changed yet: the same range again with one block per chain -- same kernel arithmetic, same bits
                t->blocks = 1; t->downgraded = true;
                continue;
            }
            const int stop = hs.stopped_at;
#ifdef SSM_PNP_PROF
            fprintf(stderr, "pnp chain %d frames: clocks gather %lld fused %lld algebra %lld chi %lld update %lld solve(total) %lld; fused passes %lld chi passes %lld\n", stop - f, hs.prof[0], hs.prof[1], hs.prof[2], hs.prof[3], hs.prof[4], hs.prof[5], hs.prof[6], hs.prof[7]);
            fprintf(stderr, "  fused: edges + group tree %lld | finish: barrier %lld publish %lld poll %lld barrier %lld sum %lld;  chi: items %lld barrier %lld sums %lld barrier %lld | finish: barrier %lld publish %lld poll %lld row sum %lld barrier %lld\n",
                    hs.prof[24], hs.prof[8], hs.prof[9], hs.prof[10], hs.prof[11], hs.prof[12], hs.prof[25], hs.prof[26], hs.prof[27], hs.prof[28], hs.prof[16], hs.prof[17], hs.prof[18], hs.prof[19], hs.prof[20]);
            fprintf(stderr, "  algebra: ldlt %lld exp map %lld publish + barrier %lld\n", hs.prof[29], hs.prof[30], hs.prof[31]);
#endif
            if (stop <= f || stop > n) TFAIL(t, SSM_E_HIP, "pose chain returned an invalid frame range");
            if (hipMemcpy(pose_out + (size_t)f * 16, t->d_pose + (size_t)f * 16, (size_t)(stop - f) * 128, hipMemcpyDeviceToHost) != hipSuccess) TFAIL(t, SSM_E_HIP, "pose download failed");
            std::vector<ssm_track_info> inf(stop - f);
            if (hipMemcpy(inf.data(), t->d_info + f, (size_t)(stop - f) * sizeof(ssm_track_info), hipMemcpyDeviceToHost) != hipSuccess) TFAIL(t, SSM_E_HIP, "info download failed");
            if (info_out) memcpy(info_out + f, inf.data(), inf.size() * sizeof(ssm_track_info));
            // the host copy of the state: speed, lastPose, cntLost, state, and the deque (the features of its new members come down now)
            memcpy(t->speed, hs.speed, 128); memcpy(t->last_pose, hs.last_pose, 128); t->cnt_lost = hs.cnt_lost; t->state = inf.back().state;
            std::deque<RefFrame> nd;
            for (int r = 0; r < hs.nref; r++) {
                const int idx = hs.ref_idx[r]; const int64_t g = t->next_gidx + idx;
                bool found = false;
                for (RefFrame& o : t->refs) if (o.gidx == g) { nd.push_back(std::move(o)); found = true; break; }
                if (!found) {
                    rc = need_features(idx); if (rc) return rc;
                    RefFrame nr; nr.gidx = g; nr.nkp = t->nkp[idx]; memcpy(nr.pose, hs.ref_pose[r], 128);
                    nr.pos3d.assign(t->pos3d.begin() + (size_t)idx * cap * 3, t->pos3d.begin() + (size_t)idx * cap * 3 + (size_t)nr.nkp * 3);
                    nr.desc.assign(t->desc.begin() + (size_t)idx * cap * 32, t->desc.begin() + (size_t)idx * cap * 32 + (size_t)nr.nkp * 32);
                    nd.push_back(std::move(nr));
                }
            }
            t->refs.swap(nd);
            t->device_frames += stop - f;
            for (int k = 0; k < 4; k++) t->work[k] += hs.work[k];
            f = stop;
            continue;
        }
        // ---- one frame on the host (the general case: first frame, lostRecover, a deque that reaches behind the match-table window)
        double* T_frame = pose_out + (size_t)f * 16;
        ssm_track_info info; info.state = 1; info.tracked = 0; info.n_matches = -1; info.n_inliers = 0;
        const int64_t G = t->next_gidx + f;
        t->host_frames++;
        if (t->state == 0) {                                 // initFirstFrame (track.cpp:30-36)
            memcpy(T_frame, t->prm.first_pose, 128);                    // the frame keeps the T_f_w it arrived with; lastPose is not touched (nor by lostRecover)
            int rc = push_ref(f, T_frame); if (rc) return rc;
            iso_identity(t->speed);
            t->state = 1; info.tracked = 1;
        } else if (t->state == 2) {                          // lostRecover (track.cpp:202-212)
            memcpy(T_frame, t->refs.back().pose, 128);
            t->refs.clear();
            int rc = push_ref(f, T_frame); if (rc) return rc;
            t->state = 1; t->cnt_lost = 0; info.tracked = 1;
        } else {                                             // trackRefFrame (track.cpp:140-200)
            int rc = need_features(f); if (rc) return rc;
            ssm_pnp::iso_mul(t->speed, t->refs.back().pose, T_frame);          // currentFrame->setTransform(speed * refFrames.back()->getTransform())
            int nc = 0;
            for (const RefFrame& ref : t->refs) {
                // orb->match(pFrame, currentFrame): the precomputed table when pFrame is one of the R frames in front of the current one
                const ssm_dmatch* m = nullptr; int nm = 0;
                const int64_t back = G - ref.gidx;           // 1 .. R: slot R - back
                if (back >= 1 && back <= R && t->nmatch[(size_t)f * R + (R - back)] >= 0) {
                    rc = need_matches(f); if (rc) return rc;
                    nm = t->nmatch[(size_t)f * R + (R - back)]; m = t->matches.data() + ((size_t)f * R + (R - back)) * cap;
                } else if (ref.nkp >= 1 && t->nkp[f] >= 2) {  // an older reference frame (the deque after tracking failures): match the pair now
                    TCHK(t, ssm_match(t->ctx, ref.desc.data(), ref.nkp, t->desc.data() + (size_t)f * cap * 32, t->nkp[f], t->ratio, t->tmp_matches.data(), cap, &nm));
                    m = t->tmp_matches.data();
                }
                double inv[16]; ssm_pnp::iso_inverse(ref.pose, inv);
                for (int k = 0; k < nm; k++) {
                    const float* p = ref.pos3d.data() + (size_t)m[k].queryIdx * 3;
                    if (p[0] == 0.f && p[1] == 0.f && p[2] == 0.f) continue;
                    double v[3]; ssm_pnp::iso_apply(inv, (double)p[0], (double)p[1], (double)p[2], v);
                    t->obj[3 * nc] = (float)v[0]; t->obj[3 * nc + 1] = (float)v[1]; t->obj[3 * nc + 2] = (float)v[2];
                    const ssm_keypoint& kp = t->kps[(size_t)f * cap + m[k].trainIdx];
                    t->img[2 * nc] = kp.x; t->img[2 * nc + 1] = kp.y;
                    nc++;
                }
            }
            info.n_matches = nc;
            bool ok = nc >= 15;
            double T[16];
            if (ok) {
                ssm_pnp::iso_mul(t->speed, t->last_pose, T);                    // T = speed * lastPose
                int success = 0;
                info.n_inliers = ssm_pnp::solve(t->img.data(), t->obj.data(), nc, cam, t->prm.pnp_min_inliers, T, t->inl.data(), t->edges.data(), &success);
                ok = info.n_inliers >= 15;
            }
            if (!ok) { t->cnt_lost++; if (t->cnt_lost > t->prm.max_lost_frame) t->state = 2; }
            else {
                memcpy(T_frame, T, 128);
                t->cnt_lost = 0;
                double linv[16]; ssm_pnp::iso_inverse(t->last_pose, linv);
                ssm_pnp::iso_mul(T, linv, t->speed);                            // speed = T * lastPose.inverse()
                memcpy(t->last_pose, T, 128);
                rc = push_ref(f, T); if (rc) return rc;
                info.tracked = 1;
            }
        }
        info.state = t->state;
        if (info_out) info_out[f] = info;
        f++;
    }
    t->next_gidx += n;
    // (a note, the call succeeded: ssm_tracker_last_error is how a downgrade of the device chain shows)
    t->err = t->downgraded ? "note: the pose chain's cluster of blocks timed out in an exchange; this tracker continues with one block per chain (same poses)" : "";
    return SSM_OK;
}
extern "C" int ssm_tracker_stats(const ssm_tracker* t, int64_t* device_frames, int64_t* host_frames)
{
    if (!t) return SSM_E_INVAL;
    if (device_frames) *device_frames = t->device_frames;
    if (host_frames) *host_frames = t->host_frames;
    return SSM_OK;
}
extern "C" int ssm_tracker_work(const ssm_tracker* t, int64_t work[4])
{
    if (!t || !work) return SSM_E_INVAL;
    for (int k = 0; k < 4; k++) work[k] = t->work[k];
    return SSM_OK;
}
