// ssm_map.hip -- the context's voxel map behind the C ABI: the growable table (map_settle), ssm_map_*, ssm_voxel_filter, the multi-GPU merge
// (ssm_comm_*, ssm_voxel_allgather: RCCL) and the device-resident Mapper (ssm_backproject_dev, ssm_viewer_map_*).  Kernels: kernels_map.hip, voxel_sort.hip.
#include "ssm_ctx.h"
#include <algorithm>

int table_alloc(ssm_ctx* c, VoxTable& t, int cap_log2)
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
// every stream of the context that may hold work on the context map (ssm_seq_process with SSM_MAP_STREAM=0 runs the map stage of alternate sub-batches on the
// chains' own streams): waited for before the table is replaced or skipped blocks are run again (a launch on another stream could still be
// inserting into the old table while it is re-hashed)
static int map_drain_other_streams(ssm_ctx* c, hipStream_t s)
{
    for (hipStream_t st : c->map_launch_streams) if (st && st != s) HIPCHK(c, hipStreamSynchronize(st));      // (the streams fused launches were queued on: one, normally)
    return SSM_OK;
}
// The fused map stage needs an overflow list that can take what every resident block may still append after the last block passed its high-water check
// (kernels_map.hip map_stream2_kernel): 2^18 records of head room + resident blocks x records per block (1024 x 12288 records of 112 B: 1.4 GB of the device's 288).
// The context trades its small list for it at the first fused launch; the map is at rest (list empty) when this runs.
static int64_t map_stream_list_records()
{
    const char* e = getenv("SSM_MAP_TEST_SMALL_LIST"); const bool small = e && atoi(e) != 0;      // tests (read per launch): keep the 2^18-record list -- every block then skips itself and is run again, 21 at a time
    return small ? (int64_t)VOX_OVF_RECORDS : (int64_t)VOX_OVF_RECORDS + (int64_t)k_map_fuse_resident_blocks() * k_map_fuse_block_records();
}
static int map_ensure_stream_list(ssm_ctx* c, hipStream_t s)
{
    VoxTable& t = c->map;
    if (t.skip && t.ovf_cap >= map_stream_list_records()) return SSM_OK;
    int r = map_settle(c, s, 0); if (r) return r;
    if (!t.skip) { DALLOC(c, t.skip, (size_t)k_map_fuse_skip_cap()); DALLOC(c, c->d_redo, (size_t)k_map_fuse_skip_cap()); }
    if (t.ovf_cap < map_stream_list_records()) {
        ssm_voxel* big = nullptr;
        DALLOC(c, big, (size_t)map_stream_list_records());
        if (t.ovf) hipFree(t.ovf);
        t.ovf = big; t.ovf_cap = (int)map_stream_list_records();
        struct { int32_t cap, pad; ssm_voxel* buf; } tail = { t.ovf_cap, 0, t.ovf };
        HIPCHK(c, hipMemcpyAsync(t.counters + 3, &tail.cap, 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(t.counters + 4, &tail.buf, 8, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    return SSM_OK;
}
// One launch of the fused map stage on the context map.  Its arguments go into a ring of eight (the host looks at the counters at most a few launches late:
// map_before_launch's snapshots, the start of the next call); the ring slot is the tag the launch's skipped blocks carry.
int map_fuse_launch(ssm_ctx* c, hipStream_t s, const MapLaunch& L)
{
    VoxTable& t = c->map;
    int r = map_ensure_stream_list(c, s); if (r) return r;
    const int gx = k_map_fuse_blocks_per_frame(L.w, L.h);
    if ((int64_t)gx * L.n > k_map_fuse_skip_cap() / 4) FAIL(c, SSM_E_INVAL, "map stage: more than " + std::to_string(k_map_fuse_skip_cap() / 4) + " blocks in one launch (frames per launch x frame size)");
    const int64_t resident = std::min<int64_t>(k_map_fuse_resident_blocks(), (int64_t)gx * L.n);
    const int64_t hw = (int64_t)t.ovf_cap - resident * k_map_fuse_block_records();
    const int tag = (int)(c->map_ring_next++ & 7u);
    c->map_ring[tag] = L; c->map_ring[tag].valid = true;
    c->map_unexamined = true;
    if (std::find(c->map_launch_streams.begin(), c->map_launch_streams.end(), s) == c->map_launch_streams.end()) c->map_launch_streams.push_back(s);
    HIPCHK(c, k_map_fuse(L.depth, L.rgb, L.sem, L.pose, L.n, L.w, L.h, c->cfg.camera, c->cfg.mapper_max_distance, (float)c->cfg.mapper_resolution,
                         t.tab, t.cap_log2, t.counters, L.npoints, s, t.skip, (int)hw, tag, nullptr, 0));
    return SSM_OK;
}
// The context map has no capacity of its own (the reference's globalMap grows without limit, src/mapper.cpp:121-158): voxel_capacity_log2 is where it STARTS.
// map_settle brings the map to rest on stream s (blocking): blocks of the fused map stage that skipped themselves are run again (in batches the overflow list can
// take whole), the overflow list is merged into the table, and the table is re-hashed into a larger one whenever 4 x (voxels + overflow records + reserve) exceeds
// its slots -- `reserve` = new voxels the caller is about to add at most, so that an insert / merge of a known size can never overflow.  SSM_E_CAPACITY only when
// something cannot be placed beyond 2^vox_max_log2 slots (28: the key's range), SSM_E_NOMEM when the larger table cannot be allocated; in both cases nothing is
// lost: table and list stay as they are.
int map_settle(ssm_ctx* c, hipStream_t s, int64_t reserve, int32_t* counters_out)       // counters_out: the table's counter block as it is when the map is at rest
{
    VoxTable& t = c->map;
    int lo = 0;                                                   // overflow records [0, lo) are merged already
    for (int round = 0; round < 4096; round++) {
        int32_t cnt[8];
        HIPCHK(c, hipMemcpyAsync(cnt, t.counters, 32, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (c->map_unexamined || cnt[6] > 0) {
            // fused launches may sit on the context's other streams: their skip entries (and their table traffic) are complete only when those have drained
            { const int r = map_drain_other_streams(c, s); if (r) return r; }
            HIPCHK(c, hipMemcpyAsync(cnt, t.counters, 32, hipMemcpyDeviceToHost, s)); HIPCHK(c, hipStreamSynchronize(s));
            c->map_unexamined = false;
        }
        if (cnt[6] > 0) {                                          // blocks that skipped themselves: take their ids over, the device list starts again
            const int ns = cnt[6] < k_map_fuse_skip_cap() ? cnt[6] : k_map_fuse_skip_cap();
            const size_t at = c->map_skipped.size(); c->map_skipped.resize(at + (size_t)ns);
            const int32_t z = 0;
            HIPCHK(c, hipMemcpyAsync(c->map_skipped.data() + at, t.skip, (size_t)ns * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(t.counters + 6, &z, 4, hipMemcpyHostToDevice, s)); HIPCHK(c, hipStreamSynchronize(s));
            cnt[6] = 0;
        }
        const int64_t n = cnt[0], hi = cnt[2] < t.ovf_cap ? cnt[2] : t.ovf_cap, m = hi - lo;
        const int64_t slots = (int64_t)1 << t.cap_log2;
        const bool grow = 4 * (n + m + reserve) > slots && t.cap_log2 < c->vox_max_log2;
        if (!grow && 2 * (n + m + reserve) > slots && (m > 0 || reserve > 0)) {
            // at voxel_max_capacity_log2 and more than half full with something still to place.  A caller that announced its insert (reserve) is refused before
            // anything is added; records waiting in the overflow list have no table to go to: the map is incomplete from here on (flag bit 0, reported until
            // ssm_map_clear).  (A complete map that is merely more than half full at the cap stays readable: nothing to place, nothing refused -- ADVICE r05.)
            if (m > 0) { const int32_t lost[2] = { cnt[1] | 1, 0 }; HIPCHK(c, hipMemcpyAsync(t.counters + 1, lost, 8, hipMemcpyHostToDevice, s)); HIPCHK(c, hipStreamSynchronize(s)); }
            FAIL(c, SSM_E_CAPACITY, "the voxel map needs more than 2^" + std::to_string(c->vox_max_log2) + " slots (voxel_max_capacity_log2)");
        }
        if (m <= 0 && !grow) {
            if (cnt[2] != 0) { const int32_t z = 0; HIPCHK(c, hipMemcpyAsync(t.counters + 2, &z, 4, hipMemcpyHostToDevice, s)); HIPCHK(c, hipStreamSynchronize(s)); cnt[2] = 0; }
            lo = 0;
            if (!c->map_skipped.empty()) {
                // the map is at rest and its list is empty: run the next batch of skipped blocks again -- as many as the list can take whole, all of one launch
                // (the ids of a launch carry its ring slot).  They add to the table what they can and the rest to the list; the loop then merges, grows, goes on.
                const int tag = (c->map_skipped.back() >> 24) & 7;
                const MapLaunch L = c->map_ring[tag];
                if (!L.valid) FAIL(c, SSM_E_HIP, "voxel map: skipped blocks of a launch whose arguments are no longer known");
                const size_t room = (size_t)(t.ovf_cap / k_map_fuse_block_records());
                std::vector<int32_t> batch;
                for (size_t i = c->map_skipped.size(); i-- > 0 && batch.size() < room; ) if (((c->map_skipped[i] >> 24) & 7) == tag) { batch.push_back(c->map_skipped[i]); c->map_skipped.erase(c->map_skipped.begin() + (long)i); }
                HIPCHK(c, hipMemcpyAsync(c->d_redo, batch.data(), batch.size() * 4, hipMemcpyHostToDevice, s)); HIPCHK(c, hipStreamSynchronize(s));
                HIPCHK(c, k_map_fuse(L.depth, L.rgb, L.sem, L.pose, L.n, L.w, L.h, c->cfg.camera, c->cfg.mapper_max_distance, (float)c->cfg.mapper_resolution,
                                     t.tab, t.cap_log2, t.counters, L.npoints, s, t.skip, 0, tag, c->d_redo, (int)batch.size()));
                c->map_redone += (long)batch.size();
                continue;
            }
            if (counters_out) memcpy(counters_out, cnt, 16);
            return SSM_OK;
        }
        if (grow) {
            { const int r = map_drain_other_streams(c, s); if (r) return r; }      // nobody may still be inserting into the table that is replaced
            int L = t.cap_log2; while (L < c->vox_max_log2 && 4 * (n + m + reserve) > ((int64_t)1 << L)) L++;
            VoxTable nt; nt.ovf = t.ovf; nt.ovf_cap = t.ovf_cap; nt.skip = t.skip;
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
// ssm_seq_process, in front of every launch of the map stage on stream s: decides how many of the `remaining` frames of the sub-batch the launch takes (*nq) and
// makes room for what they are expected to add.  The expectation is the stream's own rate: voxels (+ overflow records) per frame fused since the last clear, the
// largest value seen (map_vpf); x 1.5 is what a launch is granted.
//   * a context that has no rate yet takes ONE frame and is settled behind it (the one wait of its lifetime): a first launch of a whole sub-batch at a leaf far below
//     the pixel footprint would add more than table + overflow list hold, and what is dropped cannot be recovered;
//   * a small table (< 2^20 slots) is settled exactly in front of every launch, grown for the launch's expectation, and takes at most slots / 4096 frames;
//   * a large one is looked at through the counters of the launch BEFORE the previous one (a two-slot ring of asynchronous copies: the host never waits for the
//     launch it has just queued) and settled -- grown for the expectation -- when it is a quarter full, when its overflow list is in use, or when the frames queued
//     since that snapshot plus this launch are expected to fill it beyond a half.
static void map_learn_rate(ssm_ctx* c, int64_t total, int64_t frames) { if (frames > 0 && total > 0) { const double v = (double)total / (double)frames; if (v > c->map_vpf) c->map_vpf = v; } }
static int map_settle_exact(ssm_ctx* c, hipStream_t s, int64_t reserve)
{
    int r = map_settle(c, s, reserve); if (r) return r;
    int32_t cnt[4];
    HIPCHK(c, hipMemcpyAsync(cnt, c->map.counters, 16, hipMemcpyDeviceToHost, s)); HIPCHK(c, hipStreamSynchronize(s));
    map_learn_rate(c, cnt[0], c->map_frames);
    c->map_launches = 0; c->map_known_total = cnt[0]; c->map_known_frames = c->map_frames;      // (the count just read is the newest the host knows: the estimates start from it)
    return SSM_OK;
}
int map_before_launch(ssm_ctx* c, hipStream_t s, int remaining, int* nq)
{
    VoxTable& t = c->map;
    if (c->map_vpf < 0) { *nq = 1; return map_settle(c, s, 0); }             // no rate yet (map_after_launch learns it behind this frame)
    const double grant = c->map_vpf * 1.5;
    if (t.cap_log2 < 20) {
        int f = (1 << t.cap_log2) >> 12; f = f < 1 ? 1 : (f > remaining ? remaining : f);
        *nq = f;
        return map_settle_exact(c, s, (int64_t)(f * grant));
    }
    *nq = remaining;
    if (c->map_launches < 2) {
        // no snapshot of this run of launches yet: the table was settled when the run began (map_frames counts what has been queued since)
        if (2.0 * ((double)c->map_known_total + (double)(c->map_frames - c->map_known_frames + remaining) * grant) > (double)((int64_t)1 << t.cap_log2))
            return map_settle_exact(c, s, (int64_t)(remaining * grant));
        return SSM_OK;
    }
    const int slot = (int)(c->map_launches & 1);
    HIPCHK(c, hipEventSynchronize(c->map_snap_ev[slot]));
    const int32_t* cnt = c->h_map_snap + 8 * slot;
    const int64_t total = (int64_t)cnt[0] + cnt[2], at = c->map_snap_frames[slot];
    map_learn_rate(c, total, at);
    c->map_known_total = total; c->map_known_frames = at;
    const double expect = (double)total + (double)(c->map_frames - at + remaining) * grant;
    if (cnt[2] > 0 || cnt[6] > 0 || 4 * (int64_t)cnt[0] > ((int64_t)1 << t.cap_log2) || 2.0 * expect > (double)((int64_t)1 << t.cap_log2))      // (cnt[6]: blocks skipped themselves)
        return map_settle_exact(c, s, (int64_t)(remaining * grant));
    return SSM_OK;
}
int map_after_launch(ssm_ctx* c, hipStream_t s, int frames, bool inputs_volatile)
{
    c->map_frames += frames;
    // inputs_volatile: the launch read a buffer that the next work on this stream overwrites (the labels SegNet has just generated): blocks that skipped themselves
    // must run again before that, so the map is brought to rest behind the launch (one wait per SegNet sub-batch, next to ~25 ms of SegNet)
    if (inputs_volatile && c->map_vpf >= 0) { const int r = map_settle(c, s, 0); if (r) return r; }
    if (c->map_vpf < 0) {                                         // the context's first frames: wait for them once and take the rate
        int r = map_settle_exact(c, s, 0); if (r) return r;
        if (c->map_vpf < 0) c->map_vpf = 1.0;                     // (nothing was fused: every pixel gated; the next launches are granted little and watched)
        int32_t cnt[4]; HIPCHK(c, hipMemcpy(cnt, c->map.counters, 16, hipMemcpyDeviceToHost));
        c->map_known_total = cnt[0]; c->map_known_frames = c->map_frames;
        return SSM_OK;
    }
    const int slot = (int)(c->map_launches & 1);
    HIPCHK(c, hipMemcpyAsync(c->h_map_snap + 8 * slot, c->map.counters, 32, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipEventRecord(c->map_snap_ev[slot], s));
    c->map_snap_frames[slot] = c->map_frames;
    c->map_launches++;
    return SSM_OK;
}

// ---------------------------------------------------------------- voxel map
// the stream the context map is read on (ssm_ctx::map_tail) -- the scratch table lives on the context stream
static hipStream_t table_stream(ssm_ctx* c, VoxTable& t) { return (&t == &c->map && c->map_tail) ? c->map_tail : c->stream; }
static int table_count(ssm_ctx* c, VoxTable& t, int* n)
{
    int32_t cnt[4];
    hipStream_t s = table_stream(c, t);
    if (&t == &c->map) { const int r = map_settle(c, s, 0, cnt); if (r) return r; }      // (its last look at the counters is the count: one round trip, not two)
    else { HIPCHK(c, hipMemcpyAsync(cnt, t.counters, 8, hipMemcpyDeviceToHost, s)); HIPCHK(c, hipStreamSynchronize(s)); }
    if (cnt[1] & 1) FAIL(c, SSM_E_CAPACITY, "voxel map incomplete (contributions were dropped): ssm_map_clear and start from a larger voxel_capacity_log2");
    *n = cnt[0];
    return SSM_OK;
}
// sorts the table's voxels by key; leaves compact array + order in scratch2.  returns pointers
static int table_sorted(ssm_ctx* c, VoxTable& t, int* n_out, ssm_voxel** compact, uint32_t** order)
{
    hipStream_t st = table_stream(c, t);
    int n = 0; int r = table_count(c, t, &n); if (r) return r;
    *n_out = n; *compact = nullptr; *order = nullptr;
    if (n == 0) return SSM_OK;
    size_t tmp_bytes = 0;
    HIPCHK(c, voxel_sort_pairs(nullptr, &tmp_bytes, nullptr, n, nullptr, nullptr, nullptr, nullptr, st));
    const size_t a = ((size_t)n * sizeof(ssm_voxel) + 255) & ~(size_t)255, kb = ((size_t)n * 8 + 255) & ~(size_t)255, ib = ((size_t)n * 4 + 255) & ~(size_t)255;
    r = ensure_scratch2(c, a + 2 * kb + 2 * ib + tmp_bytes + 512); if (r) return r;
    uint8_t* p = reinterpret_cast<uint8_t*>(c->d_scratch2);
    ssm_voxel* comp = reinterpret_cast<ssm_voxel*>(p); p += a;
    uint64_t* ka = reinterpret_cast<uint64_t*>(p); p += kb; uint64_t* kbuf = reinterpret_cast<uint64_t*>(p); p += kb;
    uint32_t* ia = reinterpret_cast<uint32_t*>(p); p += ib; uint32_t* ibuf = reinterpret_cast<uint32_t*>(p); p += ib;
    int32_t* dn = reinterpret_cast<int32_t*>(p); p += 256;
    HIPCHK(c, k_voxel_compact(t.tab, t.cap_log2, comp, dn, st));
    HIPCHK(c, voxel_sort_pairs(p, &tmp_bytes, comp, n, ka, kbuf, ia, ibuf, st));
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
    c->map_tail = nullptr;                                        // (the context stream has joined the map's side stream: from here the map's newest work is on it)
    HIPCHK(c, k_voxel_clear(c->map.tab, c->map.cap_log2, c->map.counters, c->stream));       // (the capacity it has grown to stays)
    c->map_full_reported = false; c->map_launches = 0; c->map_frames = 0; c->map_known_total = 0; c->map_known_frames = 0;      // (the rate map_vpf is the stream's: kept)
    c->map_skipped.clear(); for (MapLaunch& L : c->map_ring) L.valid = false;            // (blocks that skipped themselves belong to the map that is cleared)
    return SSM_OK;
}
extern "C" int ssm_map_insert(ssm_ctx* c, const ssm_point* pts, int n)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    c->map_tail = nullptr;                                        // (the context stream has joined the map's side stream: from here the map's newest work is on it)
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
extern "C" int ssm_map_stats(ssm_ctx* c, int64_t stats[4])
{
    if (!c || !stats) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    stats[0] = c->map.cap_log2; stats[1] = c->map_grown; stats[2] = c->map_redone; stats[3] = c->map.ovf_cap;
    return SSM_OK;
}
extern "C" int ssm_map_export(ssm_ctx* c, ssm_point* out, int cap, int* n_out)
{
    if (!c || !n_out) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    c->map_tail = nullptr;                                        // (the context stream has joined the map's side stream: from here the map's newest work is on it)
    return table_export_points(c, c->map, out, cap, n_out);
}
extern "C" int ssm_map_export_table(ssm_ctx* c, ssm_voxel* out, int cap, int* n_out)
{
    if (!c || !n_out) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    c->map_tail = nullptr;                                        // (the context stream has joined the map's side stream: from here the map's newest work is on it)
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
    c->map_tail = nullptr;                                        // (the context stream has joined the map's side stream: from here the map's newest work is on it)
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
    hipStream_t st = table_stream(c, c->map);                            // (waits for the map's stream only: ssm_ctx::map_tail)
    HIPCHK(c, k_voxel_gather_table(comp, order, n, out, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return SSM_OK;
}
extern "C" int ssm_map_merge_table_dev(ssm_ctx* c, const ssm_voxel* tab, int n)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    c->map_tail = nullptr;                                        // (the context stream has joined the map's side stream: from here the map's newest work is on it)
    if (n < 0 || (n && !tab)) FAIL(c, SSM_E_INVAL, "bad arguments");
    { const int r = map_settle(c, c->stream, n); if (r) return r; }
    HIPCHK(c, k_voxel_merge(tab, n, c->map.tab, c->map.cap_log2, c->map.counters, c->stream));
    return SSM_OK;
}
// ---------------------------------------------------------------- multi-GPU: one process per GPU, the voxel-map merge is the only collective
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
    c->map_tail = nullptr;                                        // (the context stream has joined the map's side stream: from here the map's newest work is on it)
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
extern "C" int ssm_viewer_map_release(ssm_ctx* c, int test_fail_next)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (test_fail_next) { c->viewer_fail_next = true; return SSM_OK; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (ssm_ctx::CloudSlab& sl : c->cloud_slabs) if (sl.live == 0 && sl.d) { hipFree(sl.d); sl.d = nullptr; sl.cap = 0; sl.used = 0; }      // (an emptied slab is skipped by the allocator: cap 0)
    if (c->d_vcat) { hipFree(c->d_vcat); c->d_vcat = nullptr; c->vcat_cap = 0; }
    if (c->d_vmap) { hipFree(c->d_vmap); c->d_vmap = nullptr; c->vmap_cap = 0; c->vmap_n = 0; }
    return SSM_OK;
}
extern "C" int ssm_viewer_map_update(ssm_ctx* c, int rebuild, ssm_cloud* const* clouds, const double* poses, int n, float leaf, int* n_map_out)
{
    if (!c) return SSM_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu); hipSetDevice(c->device);
    if (n < 0 || (n && (!clouds || !poses)) || !(leaf > 0)) FAIL(c, SSM_E_INVAL, "bad arguments");
    if (c->viewer_fail_next) { c->viewer_fail_next = false; FAIL(c, SSM_E_NOMEM, "ssm_viewer_map_update: failure requested by ssm_viewer_map_release(test_fail_next)"); }
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

