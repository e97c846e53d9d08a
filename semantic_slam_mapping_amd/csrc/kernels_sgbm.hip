// kernels_sgbm.hip -- depth from stereo for the KITTI path: cv::StereoSGBM as calDisparity_SGBM configures it
// (/root/reference/src/stereo.cpp:11-30) and the disparity -> depth conversion of FrameReader
// (/root/reference/src/rgbdframe.cpp:81-116).  SURVEY.md s.8(f) rank 2.  The contract is oracle/sgbm.c (OpenCV 2.4's
// computeDisparitySGBM in single-pass mode, medianBlur 3, filterSpeckles), all int16 arithmetic, bit-exact.
//
// OpenCV walks the image row by row with ring buffers; here every stage is a volume kernel over (y, x, d):
//   sgbm_prefilter   per pixel of both images: the clipped x-Sobel and the raw intensity, each with the Birchfield-Tomasi
//                    half-sample interval [v0, v1] (6 byte planes per image)
//   sgbm_pixcost     BT cost of (y, x, d), gradient plane + raw plane / 4                              -> u8 volume
//   sgbm_hbox/vbox   the SAD window as a separable box sum with OpenCV's replicate borders (+ P2, + the two 2.4 quirks:
//                    column 0 keeps row 0's cost, rows past height-1-SH2 keep the last full window)     -> C, u16 volume
//   sgbm_path<K,M>   one scan direction r: L_r(p,d) = C(p,d) + min(L_r(p-r,d), L_r(p-r,d+-1) + P1, min_k L_r(p-r,k) + P2)
//                    - min_k L_r(p-r,k).  A PATH (a row for r = (-1,0) and (+1,0); a column or diagonal for the three directions
//                    that come from the previous row) is owned by 16 lanes = one DPP row, each lane K = D/16 consecutive
//                    disparities: the d+-1 neighbours cross lanes by row_shr/row_shl, min_k by four row_ror steps -- no LDS,
//                    no barrier in the recurrence.  Paths are independent and every direction writes its own L volume, so the
//                    five directions are five launches on five streams, each (#paths x 16) threads running its own loop.
//   sgbm_wta         one pixel per 16 lanes, all pixels in parallel: S = min(32767, sum of the five L) (terms >= 0: equal to
//                    OpenCV's two saturate_casts), first minimum, uniqueness ratio, sub-pixel parabola; the right-image table
//                    OpenCV fills while walking right to left becomes an atomicMin on (cost, x) keys
//   sgbm_lrcheck     the left-right consistency check on both roundings of the disparity
//   sgbm_median3, sgbm_speckle_* (connected components by union-find), sgbm_depth
// HBM-bound by design (about a GB of volume traffic per 1241x376x80 frame); no MFMA.
#include "ssm_internal.h"
#include <climits>
#include <mutex>
#include <tuple>
#include <map>
#include <set>
#include <utility>
#include <type_traits>

#define SG_MAXC 32767
#define SG_DISP_SHIFT 4
#define SG_DISP_SCALE 16

// The kernels live in four included files (round 6: one 2 000-line file before): cost volume, the per-direction path kernels of forms 0 / 1, form 2's row and sweep
// kernels, and the post-processing; this file keeps the shared definitions above and the launchers below.
#include "sgbm_cost.inc"
#include "sgbm_paths.inc"
#include "sgbm_sweep.inc"
#include "sgbm_post.inc"
// ------------------------------------------------------------------ launcher
// side streams for the five concurrent scan directions: one set per caller stream (a context), created at its first call;
// every call forks them from and joins them into the caller's stream by events, so calls on one context stay ordered and
// contexts used from different host threads never share an event
struct SgStreams { hipStream_t s[4] = {}; hipEvent_t fork = nullptr, done[4] = {}; bool ok = false; };
static std::mutex g_sg_mu;
static std::map<std::pair<int, hipStream_t>, SgStreams*> g_sg_sets;
// ssm_destroy: the side streams / events of a context's stream go with it (and a recycled stream handle can never find a stale set)
void k_sgbm_release_stream(hipStream_t caller)
{
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_sg_mu);
    auto it = g_sg_sets.find({dev, caller});
    if (it == g_sg_sets.end()) return;
    SgStreams* st = it->second;
    for (int i = 0; i < 4; i++) { if (st->s[i]) { (void)hipStreamSynchronize(st->s[i]); (void)hipStreamDestroy(st->s[i]); } if (st->done[i]) (void)hipEventDestroy(st->done[i]); }
    if (st->fork) (void)hipEventDestroy(st->fork);
    delete st;
    g_sg_sets.erase(it);
}
static SgStreams& sg_streams(hipStream_t caller)
{
    std::mutex& mu = g_sg_mu;
    auto& sets = g_sg_sets;
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    SgStreams*& st = sets[{dev, caller}];
    if (!st) {
        st = new SgStreams;
        bool ok = hipEventCreateWithFlags(&st->fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 4 && ok; i++)
            ok = hipStreamCreateWithFlags(&st->s[i], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&st->done[i], hipEventDisableTiming) == hipSuccess;
        st->ok = ok;
    }
    return *st;
}
// dynamic LDS above the 64 KB default needs the attribute on the current device's copy of the function: once per (device, kernel)
static hipError_t sg_allow_lds(const void* fn, size_t bytes)
{
    if (bytes <= 48 * 1024) return hipSuccess;
    static std::mutex mu; static std::set<std::pair<int, const void*>> done;
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, fn})) return hipSuccess;
    int lim = 0;
    if (hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lim <= 0) lim = 160 * 1024;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lim - 1024);
    if (e == hipSuccess) done.insert({dev, fn});
    return e;
}
// SSM_SGBM_FORM (read once per process): 2 = sgbm_rows + sgbm_sweep (default), 1 = four L volumes + sgbm_col_wta (round 3), 0 = five L volumes + sgbm_wta
// SSM_SGBM_STRIP: columns per sweep strip (tests: many seams on small images).
static int sgbm_form()
{
    static const int form = [] {
        const char* v = getenv("SSM_SGBM_FORM"); if (v) { const int f = atoi(v); return f < 0 ? 0 : f > 2 ? 2 : f; }
        return 2;
    }();
    return form;
}
#define SGS_CPG 2
// per frame: mailbox granules of the sweep (every seam x 2 directions x SGS_SLOTS x (NP + 1) x 16 lanes); strips are at least 2 SGS_CPG columns wide
static size_t sgbm_mbox_bytes_per_frame(int w1, int D)
{
    const int K = D / 16, NG = (K + 1) / 2 + 1, maxNS = (w1 + 2 * SGS_CPG - 1) / (2 * SGS_CPG);
    return (size_t)(maxNS > 1 ? maxNS - 1 : 0) * 2 * SGS_SLOTS * NG * 16 * 8;
}
static int sg_num_cus()
{
    static std::mutex mu; static std::map<int, int> cus;
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    int& n = cus[dev];
    if (n <= 0 && (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)) n = 256;
    return n;
}
template <int K>
static hipError_t sgbm_aggregate2(const uint16_t* C, uint16_t* S04, uint16_t* ck, unsigned* flags, int w, int w1, int h, int nb, const ssm_sgbm_params& p, int minX1, int P1, int P2,
                                  bool costs_below_2_15, int16_t* disp_tmp, unsigned* disp2key, int16_t* disp1, int* fail_out, int concurrent, hipStream_t s)
{
    constexpr int NG = (K + 1) / 2 + 1, D = 16 * K;
    const size_t np = (size_t)w * h, npb = np * nb;
    const int uniq = p.uniquenessRatio >= 0 ? p.uniquenessRatio : 10;
    static const int strip_env = [] { const char* v = getenv("SSM_SGBM_STRIP"); return v ? atoi(v) : 0; }();
    // strips: at most 64 groups x SGS_CPG columns per block, and as few strips as that allows (the widest blocks: a row costs every block the same barrier and
    // hand-off whatever its width).  Measured at 64 pairs per launch: 10 strips of 118 columns 5.6 ms, 19 of 62 (two blocks per CU) the same, 12 of 98 7.4 ms --
    // 768 blocks are exactly three rounds on 256 CUs, but the strips of the frame that straddles a round boundary wait a whole round for their neighbours to
    // start, and the blocks they displace make a fourth round.  A launch that would leave most CUs empty (few frames) takes narrower strips, down to 32 columns.
    int cap = 64 * SGS_CPG;
    if (strip_env >= 2 * SGS_CPG && strip_env < cap) cap = strip_env / SGS_CPG * SGS_CPG;
    else { const int cus = sg_num_cus(); while (cap > 32 && (long)nb * ((w1 + cap - 1) / cap) * 2 <= cus) cap /= 2; }
    int NS = (w1 + cap - 1) / cap;
    const int TX = ((w1 + NS - 1) / NS + SGS_CPG - 1) / SGS_CPG * SGS_CPG;
    NS = (w1 + TX - 1) / TX;
    // eight lanes per pixel (sgbm_sweep8) unless its exchange buffers do not fit a CU's LDS (D = 128: the 16-lane kernel sgbm_sweep)
    const int threads8 = (TX * 8 + 63) / 64 * 64, ng8 = threads8 / 8;
    const size_t lds8 = (size_t)4 * ng8 * ((K + 1) * 8 + SGS8_XPAD) * 4 + (size_t)ng8 * D * 2;
    const bool use8 = lds8 <= 150 * 1024;
    const int threads = use8 ? threads8 : (TX / SGS_CPG * 16 + 63) / 64 * 64, ng = threads / 16;
    const size_t lds = use8 ? lds8 : (size_t)4 * ng * NG * 16 * 4 + (size_t)ng * D * 2;
    auto sweep = use8 ? (costs_below_2_15 ? sgbm_sweep8<K, true> : sgbm_sweep8<K, false>) : (costs_below_2_15 ? sgbm_sweep<K, SGS_CPG, true> : sgbm_sweep<K, SGS_CPG, false>);
    hipError_t e = sg_allow_lds(reinterpret_cast<const void*>(sweep), lds);
    if (e != hipSuccess) return e;
    // Forward progress of the strips' hand-offs (an ordinary launch: nothing guarantees co-residency): the blocks that run hold the lowest tickets, so a frame
    // advances as soon as all NS strips of the lowest unfinished frame are resident -- which needs NS block slots for this launch even when `concurrent` sweeps
    // (the SGBM streams of the batched path) share the device.  Checked against the occupancy the runtime reports for this kernel, block size and LDS; when it
    // does not hold the caller takes form 1 (no cross-block waits).  SSM_SGBM_TEST_TIMEOUT=2 (tests) pretends it does not.
    {
        static std::mutex mu; static std::map<std::tuple<int, const void*, int, size_t>, int> occ;
        int dev = 0; (void)hipGetDevice(&dev);
        int per_cu = 0;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto key = std::make_tuple(dev, reinterpret_cast<const void*>(sweep), threads, lds);
            auto it = occ.find(key);
            if (it == occ.end()) {
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(sweep), threads, lds) != hipSuccess) per_cu = 0;
                occ[key] = per_cu;
            } else per_cu = it->second;
        }
        static const int test_hook = [] { const char* v = getenv("SSM_SGBM_TEST_TIMEOUT"); return v ? atoi(v) : 0; }();
        if (test_hook == 2 || (long)per_cu * sg_num_cus() < (long)NS * (concurrent > 0 ? concurrent : 1)) return hipErrorCooperativeLaunchTooLarge;
    }
    // checkpoints every 12 columns (measured at 8 / 12 / 16: 0.1966 / 0.1933 / 0.1918 ms per pair for the SGBM stage, but 4.32 / 4.38 / 4.33 k pairs/s for the whole
    // path -- 16 columns of costs in registers leave the kernels of the other streams less room beside it)
    if (costs_below_2_15) sgbm_rows8<12, true><<<dim3((h * 16 + 255) / 256, nb), 256, 0, s>>>(C, S04, ck, w1, h, D, P1, P2);
    else sgbm_rows8<12, false><<<dim3((h * 16 + 255) / 256, nb), 256, 0, s>>>(C, S04, ck, w1, h, D, P1, P2);
    sgbm_fill<<<(unsigned)((npb + 255) / 256), 256, 0, s>>>(disp_tmp, (int)npb, (int16_t)((p.minDisparity - 1) * SG_DISP_SCALE));
    e = hipMemsetAsync(disp2key, 0xFF, npb * 4, s);
    if (e != hipSuccess) return e;
    const size_t mbytes = (size_t)nb * (NS - 1) * 2 * SGS_SLOTS * NG * 16 * 8;
    e = hipMemsetAsync(flags, 0, 256 + mbytes, s);           // ticket counter, time-out word, every granule's tag
    if (e != hipSuccess) return e;
    // (Round 5 measured a SECOND launch of half-width strips for the frames of an under-filled last round of blocks, SSM_SGBM_TAIL_SPLIT: 0.1862 vs 0.1723 ms per pair --
    // a launch's time is proportional to its pairs, not to its rounds of blocks: strips of a frame advance in lock step and frames start as tickets are drawn, so the
    // "empty half of the last round" is filled by the frames still in flight.  Removed in round 6)
    sweep<<<nb * NS, threads, lds, s>>>(C, S04, w, w1, h, P1, P2, p.minDisparity, minX1, uniq, NS, TX, disp_tmp, disp2key, flags,
                                       reinterpret_cast<sg_u64*>(reinterpret_cast<uint8_t*>(flags) + 256), fail_out);
    {   // SSM_SGBM_TEST_TIMEOUT=1 (tests): report a hand-off time-out whatever happened, so that the caller's repeat in form 1 runs
        static const int test_hook = [] { const char* v = getenv("SSM_SGBM_TEST_TIMEOUT"); return v ? atoi(v) : 0; }();
        if (test_hook == 1 && fail_out) { e = hipMemsetAsync(fail_out, 1, 4, s); if (e != hipSuccess) return e; }
    }
#ifdef SGS8_PROBE
    if (use8) {
        (void)hipStreamSynchronize(s);
        unsigned long long pr[8]; (void)hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_sgs8_probe), sizeof(pr));
        fprintf(stderr, "[sgs8 probe] nb %d NS %d TX %d | inner wave: rows %llu, cycles/row %.0f, barrier wait/row %.0f | edge waves: rows %llu, cycles/row %.0f, barrier wait/row %.0f, slow-path rows %llu (%.3f), poll cycles per slow row %.0f\n",
                nb, NS, TX, pr[0], (double)pr[1] / (pr[0] ? pr[0] : 1), (double)pr[2] / (pr[0] ? pr[0] : 1), pr[3], (double)pr[4] / (pr[3] ? pr[3] : 1), (double)pr[5] / (pr[3] ? pr[3] : 1), pr[6], (double)pr[6] / (pr[3] ? pr[3] : 1), (double)pr[7] / (pr[6] ? pr[6] : 1));
        unsigned long long z[8] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sgs8_probe), z, sizeof(z));
        unsigned long long ph[3][12]; (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_sgs8_phase), sizeof(ph));
        const char* cn[3] = {"inner wave 5", "mailbox-edge wave, direct row", "mailbox-edge wave, slow row"};
        for (int c = 0; c < 3; c++) { const double n = ph[c][0] ? (double)ph[c][0] : 1.0;
            fprintf(stderr, "[sgs8 phase] %-30s rows %9llu | L2 + neighbours + tag check %.0f | L1, L3, publish %.0f | slow path %.0f | exchange writes %.0f | winner pass %.0f | barrier %.0f || first step %.0f, publish %.0f, take %.0f (cumulative cycles)\n", cn[c], ph[c][0], ph[c][1] / n, ph[c][2] / n, ph[c][3] / n, ph[c][4] / n, ph[c][5] / n, ph[c][6] / n, ph[c][7] / n, ph[c][8] / n, ph[c][9] / n); }
        unsigned long long zz[3][12] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sgs8_phase), zz, sizeof(zz));
    }
#endif
    sgbm_lrcheck<<<dim3((w + 255) / 256, h, nb), 256, 0, s>>>(disp_tmp, disp2key, w, h, w1, p.minDisparity, minX1, p.disp12MaxDiff > 0 ? p.disp12MaxDiff : 1, disp1);
    return hipGetLastError();
}
template <int K>
static hipError_t sgbm_aggregate(const uint16_t* C, uint16_t* const* Lv, int w, int w1, int h, int nb, const ssm_sgbm_params& p, int minX1, int P1, int P2,
                                 int16_t* disp_tmp, unsigned* disp2key, int16_t* disp1, int form, hipStream_t s)
{
    auto blocks = [](int paths) { return (paths * 16 + 255) / 256; };
    SgStreams& st = sg_streams(s);
    if (!st.ok) return hipErrorUnknown;
    hipError_t e = hipEventRecord(st.fork, s);
    for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipStreamWaitEvent(st.s[i], st.fork, 0);
    if (e != hipSuccess) return e;
    const size_t np = (size_t)w * h, npb = np * nb;
    const long long npix = (long long)w1 * h;
    const int uniq = p.uniquenessRatio >= 0 ? p.uniquenessRatio : 10;
    if (form == 1) {
        // four directions on the four side streams; the column direction follows on `s` with the winner pass inside (sgbm_col_wta)
        sgbm_path<K, 0><<<dim3(blocks(h), nb), 256, 0, st.s[0]>>>(C, Lv[0], w1, h, P1, P2);
        sgbm_path<K, 4><<<dim3(blocks(h), nb), 256, 0, st.s[1]>>>(C, Lv[4], w1, h, P1, P2);
        sgbm_path<K, 1><<<dim3(blocks(w1 + h - 1), nb), 256, 0, st.s[2]>>>(C, Lv[1], w1, h, P1, P2);
        sgbm_path<K, 3><<<dim3(blocks(w1 + h - 1), nb), 256, 0, st.s[3]>>>(C, Lv[3], w1, h, P1, P2);
        sgbm_fill<<<(unsigned)((npb + 255) / 256), 256, 0, s>>>(disp_tmp, (int)npb, (int16_t)((p.minDisparity - 1) * SG_DISP_SCALE));
        e = hipMemsetAsync(disp2key, 0xFF, npb * 4, s);
        for (int i = 0; i < 4 && e == hipSuccess; i++) { e = hipEventRecord(st.done[i], st.s[i]); if (e == hipSuccess) e = hipStreamWaitEvent(s, st.done[i], 0); }
        if (e != hipSuccess) return e;
        sgbm_col_wta<K><<<dim3(blocks(w1), nb), 256, 0, s>>>(C, Lv[0], Lv[1], Lv[3], Lv[4], w, w1, h, P1, P2, p.minDisparity, minX1, uniq, disp_tmp, disp2key);
    } else {
        sgbm_path<K, 0><<<dim3(blocks(h), nb), 256, 0, s>>>(C, Lv[0], w1, h, P1, P2);
        sgbm_path<K, 4><<<dim3(blocks(h), nb), 256, 0, st.s[0]>>>(C, Lv[4], w1, h, P1, P2);
        sgbm_path<K, 1><<<dim3(blocks(w1 + h - 1), nb), 256, 0, st.s[1]>>>(C, Lv[1], w1, h, P1, P2);
        sgbm_path<K, 2><<<dim3(blocks(w1), nb), 256, 0, st.s[2]>>>(C, Lv[2], w1, h, P1, P2);
        sgbm_path<K, 3><<<dim3(blocks(w1 + h - 1), nb), 256, 0, st.s[3]>>>(C, Lv[3], w1, h, P1, P2);
        for (int i = 0; i < 4 && e == hipSuccess; i++) { e = hipEventRecord(st.done[i], st.s[i]); if (e == hipSuccess) e = hipStreamWaitEvent(s, st.done[i], 0); }
        if (e != hipSuccess) return e;
        sgbm_fill<<<(unsigned)((npb + 255) / 256), 256, 0, s>>>(disp_tmp, (int)npb, (int16_t)((p.minDisparity - 1) * SG_DISP_SCALE));
        e = hipMemsetAsync(disp2key, 0xFF, npb * 4, s);
        if (e != hipSuccess) return e;
        sgbm_wta<K><<<dim3((unsigned)((npix + 16 * WTA_PX - 1) / (16 * WTA_PX)), nb), 256, 0, s>>>(Lv[0], Lv[1], Lv[2], Lv[3], Lv[4], w, w1, h, p.minDisparity, minX1, uniq, disp_tmp, disp2key);
    }
    sgbm_lrcheck<<<dim3((w + 255) / 256, h, nb), 256, 0, s>>>(disp_tmp, disp2key, w, h, w1, p.minDisparity, minX1, p.disp12MaxDiff > 0 ? p.disp12MaxDiff : 1, disp1);
    return hipGetLastError();
}
// strip width of sgbm_cost_kernel and its dynamic LDS: the widest TX whose ring fits (and whose per-thread column run fits the register arrays)
bool sgbm_cost_geometry(int D, int SW, int* TX_out, size_t* lds_out)
{
    const int SW2 = SW / 2, nchunk = SGC_THREADS / D;
    for (int TX = 32; TX >= 4; TX >>= 1) {
        const int AW = TX + 2 * SW2, RW = AW + D - 1;
        const size_t lds = (size_t)SW * TX * D * 2 + 2 * (((size_t)AW * D + 15) & ~(size_t)15) + 2 * (size_t)(AW + RW) * 16;
        // the interior (clamp-free) strips must not touch the volume's border columns: SW2 <= TX
        if (lds <= 150 * 1024 && (TX + nchunk - 1) / nchunk <= 16 && AW + RW <= SGC_THREADS && SW2 <= TX) { *TX_out = TX; *lds_out = lds; return true; }
    }
    return false;
}
// workspace for nb frames per launch.  Cost-volume-sized buffers by formulation: form 2 (default) C + S04 + checkpoints = 3, form 1 C + four path volumes = 5, the
// round-3 form C + five = 6 (round 6: sized by the CONFIGURED form -- it was always six, 0.45 GB per pair and 115 GB for the bench's two workspaces of 128 pairs;
// a sub-batch that has to be repeated in form 1 runs in pieces that fit, see k_sgbm).  Never less than one frame in the largest form.
static int sgbm_form_volumes(int form) { return form == 2 ? 3 : form == 1 ? 5 : 6; }
static size_t sgbm_ws_bytes(int w, int h, const ssm_sgbm_params& p, int nb, int nvol)
{
    const int maxD = p.minDisparity + p.numberOfDisparities, minX1 = maxD > 0 ? maxD : 0, maxX1 = w + (p.minDisparity < 0 ? p.minDisparity : 0);
    const size_t w1 = maxX1 > minX1 ? (size_t)(maxX1 - minX1) : 0, vol = w1 * h * p.numberOfDisparities * nb, np = (size_t)w * h * nb;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return al(32 * np) + (size_t)nvol * al(vol * 2) + 2 * al(np * 2) + 3 * al(np * 4) + 256 + al(256 + (size_t)nb * sgbm_mbox_bytes_per_frame((int)w1, p.numberOfDisparities));
}
static int sgbm_resolve_form(int form_cfg) { return form_cfg == 0 ? sgbm_form() : form_cfg == 3 ? 0 : form_cfg == 1 ? 1 : 2; }
size_t k_sgbm_workspace_bytes(int w, int h, const ssm_sgbm_params& p, int nb, int form_cfg)
{
    const size_t a = sgbm_ws_bytes(w, h, p, nb, sgbm_form_volumes(sgbm_resolve_form(form_cfg))), b = sgbm_ws_bytes(w, h, p, 1, 6);
    return a > b ? a : b;
}
// left / right: device u8 images [nb][h][w]; disp_out: device int16 [nb][h][w] (x16 fixed point, (minD-1)*16 = invalid)
// form: 0 = the process default (2 unless SSM_SGBM_FORM says otherwise), 1 / 2 / 3 = ssm_config.sgbm_form (3: the five-volume form, SSM_SGBM_FORM=0); concurrent: launches
// of this function that may be in flight on other streams at the same time (the occupancy check of form 2)
// ws_bytes: what `workspace` holds.  A formulation whose volumes for nb frames do not fit (form 1 as the repeat of a timed-out form-2 sub-batch in a workspace
// sized for form 2) runs in pieces of frames that do, one after the other on the same stream.
hipError_t k_sgbm(const uint8_t* left, const uint8_t* right, int w, int h, int nb, const ssm_sgbm_params& p, void* workspace, size_t ws_bytes, int16_t* disp_out, int raw_only, hipStream_t s, int* fail_flag,
                  int form_cfg, int concurrent)
{
    int form = sgbm_resolve_form(form_cfg);
    if (nb <= 0) return hipSuccess;
    if (sgbm_ws_bytes(w, h, p, nb, sgbm_form_volumes(form)) > ws_bytes) {
        if (nb == 1) return hipErrorOutOfMemory;
        const int half = (nb + 1) / 2; const size_t np1_ = (size_t)w * h;
        hipError_t e1 = k_sgbm(left, right, w, h, half, p, workspace, ws_bytes, disp_out, raw_only, s, fail_flag, form_cfg, concurrent);
        if (e1 != hipSuccess) return e1;
        return k_sgbm(left + (size_t)half * np1_, right + (size_t)half * np1_, w, h, nb - half, p, workspace, ws_bytes, disp_out + (size_t)half * np1_, raw_only, s, fail_flag, form_cfg, concurrent);
    }
    const int minD = p.minDisparity, D = p.numberOfDisparities, maxD = minD + D;
    const int SW = p.SADWindowSize > 0 ? p.SADWindowSize : 5, SW2 = SW / 2;
    const int ftzero = (p.preFilterCap > 15 ? p.preFilterCap : 15) | 1;
    const int P1 = p.P1 > 0 ? p.P1 : 2, P2 = (p.P2 > 0 ? p.P2 : 5) > P1 + 1 ? (p.P2 > 0 ? p.P2 : 5) : P1 + 1;
    const int minX1 = maxD > 0 ? maxD : 0, maxX1 = w + (minD < 0 ? minD : 0), w1 = maxX1 - minX1;
    const int INVALID = (minD - 1) * SG_DISP_SCALE;
    const size_t np1 = (size_t)w * h, np = np1 * nb;
    if (w1 <= 0) {                                            // no valid column: everything invalid (OpenCV's early return)
        sgbm_fill<<<(unsigned)((np + 255) / 256), 256, 0, s>>>(disp_out, (int)np, (int16_t)INVALID);
        return hipGetLastError();
    }
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    uint8_t* q = (uint8_t*)workspace;
    uint3* planes = (uint3*)q; q += al(32 * np);                 // (12 bytes per pixel and image are used)
    const size_t vol = (size_t)w1 * h * D * nb;
    uint16_t* C = (uint16_t*)q; q += al(vol * 2);
    uint16_t* Lv[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    // (the layout sgbm_ws_bytes sizes: C + the form's volumes.  Form 2: S04 and the checkpoints in Lv[0], Lv[1]; form 1: the four path volumes 0, 1, 3, 4 -- the column
    // direction lives inside sgbm_col_wta --; the round-3 form: all five)
    for (int i = 0; i < 5; i++) { if (form == 2 ? i < 2 : form == 1 ? i != 2 : true) { Lv[i] = (uint16_t*)q; q += al(vol * 2); } }
    int16_t* d_raw = (int16_t*)q; q += al(np * 2);
    int16_t* d_tmp = (int16_t*)q; q += al(np * 2);
    unsigned* d2key = (unsigned*)q; q += al(np * 4);
    int* parent = (int*)q; q += al(np * 4);
    int* count = (int*)q; q += al(np * 4);
    unsigned* sweep_flags = (unsigned*)q;                     // 256 bytes of flags, then the sweep's mailboxes
    const dim3 gimg((w + 255) / 256, h, nb);
    {
        int TX = 0; size_t lds = 0;
        if (!sgbm_cost_geometry(D, SW, &TX, &lds)) return hipErrorInvalidValue;
        const int nstrips = (w1 + TX - 1) / TX;
        const int tail = nstrips > 1 ? ((w1 - (nstrips - 1) * TX < SW2 && nstrips > 2) ? 2 : 1) : 0;     // a last strip narrower than the half window: the one before it reaches the border too
        auto launch = [&](auto kern) {
            (void)sg_allow_lds(reinterpret_cast<const void*>(kern), lds);       // the ring of a wide window needs more than the 64 KB default of dynamic LDS
            kern<<<dim3(nstrips, nb), SGC_THREADS, lds, s>>>(planes, w, h, minD, D, minX1, w1, SW2, P2, TX, tail, C);
        };
        const long cmax_c = (long)P2 + (long)SW * SW * (2 * ftzero + 63);
        if (D == 80 && SW2 == 5 && TX == 32 && cmax_c < 65536 && w >= 8) {
            // src/stereo.cpp:16-27: the ring in registers (13 KB of LDS per block) and the pre-filter records made inside the kernel's staging (no plane pass)
            constexpr int AWc = 32 + 10, RWc = AWc + 79;
            lds = 2 * ((((size_t)(AWc + 6) * 80) + 15) & ~(size_t)15) + 2 * (size_t)(AWc + RWc) * 16;
            auto kern = sgbm_cost_reg_kernel<80, 5, 32, 6>;
            (void)sg_allow_lds(reinterpret_cast<const void*>(kern), lds);
            kern<<<dim3(nstrips, nb), SGC_THREADS, lds, s>>>(left, right, ftzero, w, h, minD, minX1, w1, P2, tail, C);
        } else {
            sgbm_prefilter<<<dim3((w + 255) / 256, h, nb * 2), 256, 0, s>>>(left, right, w, h, ftzero, planes);
            if (D == 80 && SW2 == 5 && TX == 32) launch(sgbm_cost_kernel<80, 5, 32, 6>);
            else launch(sgbm_cost_kernel<0, 0, 0, 16>);
        }
    }
    int16_t* wta_out = raw_only == 1 ? disp_out : d_raw;
    hipError_t e;
    // the largest value C can take: P2 + SADWindowSize^2 x (gradient term <= 2 ftzero, raw term <= 255 / 4); every L is <= its C
    const long cmax = (long)P2 + (long)SW * SW * (2 * ftzero + 63);
    e = hipSuccess;
    if (form == 2) {
        switch (D / 16) {
#define SG_AGG2(KK) case KK: e = sgbm_aggregate2<KK>(C, Lv[0], Lv[1], sweep_flags, w, w1, h, nb, p, minX1, P1, P2, cmax < 32768, d_tmp, d2key, wta_out, fail_flag, concurrent, s); break;
            SG_AGG2(1) SG_AGG2(2) SG_AGG2(3) SG_AGG2(4) SG_AGG2(5) SG_AGG2(6) SG_AGG2(8)
#undef SG_AGG2
            default: return hipErrorInvalidValue;
        }
        if (e == hipErrorCooperativeLaunchTooLarge) {       // the sweep's strips cannot all be resident: the form without cross-block waits (nothing was launched; the cost volume is recomputed)
            return k_sgbm(left, right, w, h, nb, p, workspace, ws_bytes, disp_out, raw_only, s, fail_flag, 1, concurrent);
        }
    }
    if (form != 2) switch (D / 16) {
#define SG_AGG1(KK) case KK: e = sgbm_aggregate<KK>(C, Lv, w, w1, h, nb, p, minX1, P1, P2, d_tmp, d2key, wta_out, form, s); break;
        SG_AGG1(1) SG_AGG1(2) SG_AGG1(3) SG_AGG1(4) SG_AGG1(5) SG_AGG1(6) SG_AGG1(8)
#undef SG_AGG1
        default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess || raw_only == 1) return e;
    sgbm_median3<<<gimg, 256, 0, s>>>(d_raw, w, h, disp_out);
    if (p.speckleWindowSize > 0 && raw_only != 2) {
        const int n = (int)np1; const dim3 gb((n + 255) / 256, nb);
        sgbm_speckle_tile<<<dim3((w + SPK_TW - 1) / SPK_TW, (h + SPK_TH - 1) / SPK_TH, nb), 256, 0, s>>>(disp_out, w, h, INVALID, SG_DISP_SCALE * p.speckleRange, parent, count);
        const int nedge = ((w - 1) / SPK_TW) * h + ((h - 1) / SPK_TH) * w;
        if (nedge > 0) sgbm_speckle_edges<<<dim3((nedge + 255) / 256, nb), 256, 0, s>>>(disp_out, w, h, INVALID, SG_DISP_SCALE * p.speckleRange, parent);
        sgbm_speckle_count<<<gb, 256, 0, s>>>(n, parent, count);
        sgbm_speckle_apply<<<gb, 256, 0, s>>>(disp_out, n, INVALID, p.speckleWindowSize, parent, count);
    }
    return hipGetLastError();
}
// min_scratch: nb ints
hipError_t k_sgbm_depth(const int16_t* disp, int w, int h, int nb, double baseline, double cu, double cv, double f, double roix, double roiy, double roiz, double scale,
                        int* min_scratch, uint16_t* depth, hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(min_scratch, 0x7F, 4 * (size_t)nb, s);         // 0x7F7F7F7F: above any int16
    if (e != hipSuccess) return e;
    sgbm_min_kernel<<<dim3(64, nb), 256, 0, s>>>(disp, w * h, min_scratch);
    sgbm_depth<<<dim3((w + 255) / 256, h, nb), 256, 0, s>>>(disp, w, h, min_scratch, baseline, cu, cv, f, roix, roiy, roiz, scale, depth);
    return hipGetLastError();
}
