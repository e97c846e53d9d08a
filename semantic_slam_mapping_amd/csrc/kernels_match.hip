// kernels_match.hip -- K6: OrbFeature::match (/root/reference/src/orb.cpp:16-29) = cv::BFMatcher(NORM_HAMMING) knnMatch(k=2) + Lowe ratio test, for gfx950.
// Two formulations live here, bit-equal to each other and to oracle/match.c (strict '<' scan in train order: equal distances resolve to the lower trainIdx):
//   * the DEFAULT (second half of the file: match_expand_kernel / match_mfma_kernel / match_compact_kernel): the Hamming distance matrix as an exact matrix
//     product on the matrix cores -- descriptors expanded to +-1 FP4 elements, v_mfma_scale_f32_32x32x64_f8f6f4 tiles with the (distance, trainIdx) key built in
//     the accumulator, VALU top-2 tracking;
//   * the VALU form (first half: match_pairs / match_seq_kernel, SSM_MATCH_VARIANT=0 and the knn entry points): one block per (query frame, train frame) pair, train
//     descriptors staged in LDS (32 B each, read back as wave-wide broadcasts), every lane owns query descriptors in VGPRs, distance = 8 x (v_xor + v_bcnt
//     accumulate), matches compacted in ascending queryIdx with wave ballots.
#include "ssm_internal.h"

#define MT 512           // threads per block; each lane owns 2 queries -> 1024 queries per pass
#define TCH 1024         // train descriptors staged per chunk (32 KiB)

// Two queries per lane (QPL): halves the LDS broadcast traffic per pair and gives the VALU two independent chains.
// Best two kept as packed keys (distance << 16 | trainIdx): min/max/min on the packed key == "strict < in scan order"
// (equal distances order by index), 2 VALU ops (v_min, v_med3) instead of a compare-select ladder.  trainIdx < 65536.
#define QPL 2
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc)      // popcount(x) + acc in one instruction
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c)   // median of three (no clang builtin for the unsigned form)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// T threads work on the queries [qb, qe).  pend == nullptr: the block owns the whole pair.  Otherwise the pair is split over two
// blocks (the sequence kernel: 5 pairs per frame are too few, and too coarse, to balance 256 CUs) and NEITHER block waits for the
// other: a query yields at most one match, so role 1 (queries [0, split)) writes its matches from slot 0 and role 2 (queries
// [split, nq)) from slot split without overlap.  Each block publishes its count with one atomic exchange on *pend (-1 before); the
// block that finds the other's count there arrived second and closes the gap (moves role 2's segment down to role 1's count) and
// writes the total.  Forward progress needs no dispatch order; visibility follows the guide's release / acquire recipe.
template <int T>
__device__ __forceinline__ void match_pair(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt,
                                           double ratio, int cap, ssm_dmatch* __restrict__ out, int32_t* __restrict__ nout,
                                           int32_t* __restrict__ knn_idx, int32_t* __restrict__ knn_dist,
                                           int qb = 0, int qe = -1, int32_t* __restrict__ pend = nullptr, int role = 0)
{
    __shared__ uint4 tr[TCH * 2];
    __shared__ int wcnt[T / 64];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (qe < 0) qe = nq;
    if (tid == 0) base = role == 2 ? qb : 0;        // role 2 writes behind the slots role 1 can fill
    for (int q0 = qb; q0 < qe; q0 += T * QPL) {
        uint4 a[QPL], b[QPL]; uint32_t k0[QPL], k1[QPL];
#pragma unroll
        for (int u = 0; u < QPL; u++) {
            const int qi = q0 + u * T + tid;
            a[u] = make_uint4(0, 0, 0, 0); b[u] = a[u]; k0[u] = 0xFFFFFFFFu; k1[u] = 0xFFFFFFFFu;
            if (qi < qe) { const uint4* p = reinterpret_cast<const uint4*>(q + (size_t)qi * 32); a[u] = p[0]; b[u] = p[1]; }
        }
        for (int t0 = 0; t0 < nt; t0 += TCH) {
            const int m = min(TCH, nt - t0);
            __syncthreads();
            for (int i = tid; i < m * 2; i += T) tr[i] = reinterpret_cast<const uint4*>(t + (size_t)t0 * 32)[i];
            __syncthreads();
            // one train descriptor (two ds_read_b128 broadcasts) against the lane's QPL queries
            auto one_train = [&](int j) {
                const uint4 x = tr[2 * j], y = tr[2 * j + 1];
#pragma unroll
                for (int u = 0; u < QPL; u++) {
                    // 8 x (v_xor + accumulating v_bcnt): one dependent chain per (query, train) pair -- the unrolled loop keeps 8
                    // such chains in flight.  (Left to itself the compiler breaks the chain with three extra v_add3.)
                    uint32_t d = bcnt_acc(a[u].x ^ x.x, 0u);
                    d = bcnt_acc(a[u].y ^ x.y, d); d = bcnt_acc(a[u].z ^ x.z, d); d = bcnt_acc(a[u].w ^ x.w, d);
                    d = bcnt_acc(b[u].x ^ y.x, d); d = bcnt_acc(b[u].y ^ y.y, d); d = bcnt_acc(b[u].z ^ y.z, d); d = bcnt_acc(b[u].w ^ y.w, d);
                    const uint32_t key = (d << 16) | (uint32_t)(t0 + j);
                    // k0 <= k1 are the two smallest keys so far: the new pair is (min, median) of {key, k0, k1}
                    k1[u] = umed3(key, k0[u], k1[u]);
                    k0[u] = min(key, k0[u]);
                }
            };
            const int m4 = m & ~3;
            for (int j = 0; j < m4; j += 4) {
#pragma unroll
                for (int jj = 0; jj < 4; jj++) one_train(j + jj);
            }
            for (int j = m4; j < m; j++) one_train(j);
        }
#pragma unroll
        for (int u = 0; u < QPL; u++) {
            const int qi = q0 + u * T + tid;
            const int d0 = k0[u] >> 16, i0 = k0[u] & 0xFFFF, d1 = k1[u] >> 16, i1 = k1[u] & 0xFFFF;
            if (knn_idx && qi < qe) { knn_idx[2*qi] = i0; knn_idx[2*qi+1] = i1; knn_dist[2*qi] = d0; knn_dist[2*qi+1] = d1; }
            // ratio test exactly as orb.cpp:25: float distance < double ratio * float distance, compared in double
            const bool keep = (qi < qe) && ((double)(float)d0 < ratio * (double)(float)d1);
            const unsigned long long bal = __ballot(keep);
            __syncthreads();
            if (lane == 0) wcnt[wv] = __popcll(bal);
            __syncthreads();
            int off = base;
            for (int w = 0; w < wv; w++) off += wcnt[w];
            if (keep) {
                const int k = off + __popcll(bal & ((1ull << lane) - 1ull));
                if (out && k < cap) { ssm_dmatch mm; mm.queryIdx = qi; mm.trainIdx = i0; mm.imgIdx = 0; mm.distance = (float)d0; out[k] = mm; }
            }
            __syncthreads();
            if (tid == 0) { int s = 0; for (int w = 0; w < T / 64; w++) s += wcnt[w]; base += s; }
        }
    }
    __syncthreads();                                    // (every wave's stores are complete: the barrier's fence waits for them)
    if (role == 0) { if (tid == 0 && nout) *nout = base; return; }
    __shared__ int other;
    const int split = role == 1 ? qe : qb, mine = base - (role == 2 ? qb : 0);
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");         // this block's matches reach memory before its count
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        other = __hip_atomic_exchange(pend, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (other >= 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();
    if (other < 0) return;                              // arrived first: the other block finishes the pair
    const int n0 = role == 1 ? mine : other, n1 = role == 1 ? other : mine;
    if (n0 != split) {                                  // close the gap; dst < src, chunks in ascending order, read-all-then-write-all
        uint4* o = reinterpret_cast<uint4*>(out);
        for (int i0 = 0; i0 < n1; i0 += T) {
            const int i = i0 + tid;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (i < n1) v = o[split + i];
            __syncthreads();
            if (i < n1) o[n0 + i] = v;
            __syncthreads();
        }
    }
    if (tid == 0 && nout) *nout = n0 + n1;
}

__global__ void __launch_bounds__(MT)
match_pairs_kernel(const uint8_t* __restrict__ desc, const MatchPair* __restrict__ pairs, double ratio, int cap,
                   ssm_dmatch* __restrict__ out, int32_t* __restrict__ nout, int32_t* __restrict__ knn_idx, int32_t* __restrict__ knn_dist)
{
    const MatchPair p = pairs[blockIdx.x];
    match_pair<MT>(desc + (size_t)p.qoff * 32, p.nq, desc + (size_t)p.toff * 32, p.nt, ratio, cap,
               out ? out + (size_t)p.out_slot * cap : nullptr, nout ? nout + p.out_slot : nullptr, knn_idx, knn_dist);
}
hipError_t k_match_pairs(const uint8_t* desc, const MatchPair* pairs, int npairs, double ratio, int cap,
                         ssm_dmatch* out, int32_t* nout, int32_t* knn_idx, int32_t* knn_dist, hipStream_t s)
{
    match_pairs_kernel<<<npairs, MT, 0, s>>>(desc, pairs, ratio, cap, out, nout, knn_idx, knn_dist);
    return hipGetLastError();
}

// sequence mode: desc/nkp hold `hist` history frames followed by the frames of the call; frame f (0-based in the call)
// sits at row hist+f.  Block (r, f): query = ref frame row f+r+hist-R ... i.e. the r-th of the R frames preceding f,
// oldest first (std::deque order of Tracker::refFrames, src/track.cpp:150); train = frame f (orb->match(pFrame, cur)).
// Two blocks of MT / 2 threads per pair (blockIdx.x = 2 r + half): R n pairs of ~150 us each on 256 CUs leave a third of the chip
// idle in the last round; halves give twice the blocks at half the length.  pend[pair] starts at -1 (the launcher's memset).
__global__ void __launch_bounds__(MT / 2)
match_seq_kernel(const uint8_t* __restrict__ desc, const int32_t* __restrict__ nkp, int f0, int R, int hist, double ratio, int cap,
                 ssm_dmatch* __restrict__ out, int32_t* __restrict__ nout, int32_t* __restrict__ pend)
{
    const int f = f0 + blockIdx.y, r = blockIdx.x >> 1, half = blockIdx.x & 1;
    const int cur = hist + f, ref = cur - R + r;
    const int slot = f * R + r;
    const int nq = ref >= 0 ? nkp[ref] : -1, nt = nkp[cur];
    if (nq < 0 || nt < 2) { if (threadIdx.x == 0 && half == 0) nout[slot] = -1; return; }
    const int split = min(nq, (((nq + 1) >> 1) + 63) & ~63);
    match_pair<MT / 2>(desc + (size_t)ref * cap * 32, nq, desc + (size_t)cur * cap * 32, nt, ratio, cap,
                       out + (size_t)slot * cap, nout + slot, nullptr, nullptr, half ? split : 0, half ? nq : split, pend + blockIdx.y * R + r, half + 1);
}
hipError_t k_match_seq(const uint8_t* desc, const int32_t* nkp, int f0, int n, int R, int hist, double ratio, int cap,
                       ssm_dmatch* out, int32_t* nout, int32_t* pend, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(pend, 0xFF, sizeof(int32_t) * (size_t)n * R, s);
    if (e != hipSuccess) return e;
    match_seq_kernel<<<dim3(2 * R, n), MT / 2, 0, s>>>(desc, nkp, f0, R, hist, ratio, cap, out, nout, pend);
    return hipGetLastError();
}

// ------------------------------------------------------------------ the matcher on the matrix cores (sequence path)
// The Hamming distance matrix of two descriptor sets IS a matrix product: with the query bits expanded to q = -1 / +1 (bit set / clear) and the
// train bits to t = +16 / -16, sum_k q_k t_k = 16 (H - (256 - H)) = 32 H - 4096.  The elements are FP4 (E2M1: +-1.0 are the nibbles 0x2 / 0xA; the
// train side's 16 is its block scale 2^4) and the product runs on v_mfma_scale_f32_32x32x64_f8f6f4: 4 K-steps per 256-bit descriptor at the cycles
// of a bf16 32x32x16, half the matrix time and half the operand bytes of the i8 form (v_mfma_i32_32x32x32_i8, 8 K-steps) this kernel started with;
// every partial sum is an integer of magnitude <= 4096 + 31, exact in the f32 accumulator whatever the summation order.  The accumulator is not
// started at zero but at C[row][col] = 4096 + row, so that a finished 32 x 32 tile holds 32 H + (train index inside the tile): a key whose order is
// "smaller distance first, equal distances by the lower trainIdx", i.e. the strict '<' scan of oracle/match.c, with no VALU work to build it -- and
// as the keys are non-negative floats, their BIT PATTERNS order like the values: the VALU keeps the two smallest with unsigned integer min / max on
// the raw registers, and only a tile's two winners are converted (v_cvt_u32_f32), widened to (H << 16 | trainIdx) and merged into the running
// pair.  Bit-exact with the VALU matcher above.
//
// Layout of an expanded descriptor row (one frame): capT = cap rounded up to 32 descriptors, 128 B each, stored per tile of 32 descriptors in MFMA
// fragment order [K-step s: 4][lane half h: 2][descriptor r: 32][16 B] -- lane (h, r) of a wave takes the 32 nibbles k = 64 s + 32 h .. + 31 of
// descriptor r as its A (train) or B (query) fragment, so a tile is one contiguous 4 KB block and a fragment load is lane * 16 B (A and B use
// the same k order, whatever the hardware's, so the sum is over matching k).  Descriptors at or past nkp expand from zero bits; padded trains are
// masked in the last tile's tracking pass, padded queries are never read back.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define MM_QB 256                       // queries per block: 4 waves x 64 (two 32-column accumulators per wave)
#define MM_DB SSM_MATCH_DESC_BYTES      // bytes of one expanded descriptor (256 FP4 elements)
#define MM_KS 4                         // K-steps (MFMAs) per 32 x 32 tile
#define MM_TILE (32 * MM_DB)            // bytes of one expanded 32-descriptor tile
#ifndef MM_ABLATE
#define MM_ABLATE 0                     // scripts/ubench/match_bench.hip only (wrong results): 1 no key tracking, 2 no LDS staging / barrier, 4 no train prefetch, 8 no MFMA
#endif

__global__ void __launch_bounds__(256)
match_expand_kernel(const uint8_t* __restrict__ desc, const int32_t* __restrict__ nkp, int row0, int cap, int capT,
                    uint8_t* __restrict__ eq, uint8_t* __restrict__ et)
{
    const int row = row0 + blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;                       // one 16-byte fragment (32 bits of a descriptor): i = (tile * 8 + s * 2 + h) * 32 + r
    if (i >= capT * 8) return;
    const int r = i & 31, c = (i >> 5) & 7, d = (i >> 8) * 32 + r;
    uint32_t bits = 0;
    if (d < nkp[row]) bits = *reinterpret_cast<const uint32_t*>(desc + ((size_t)row * cap + d) * 32 + 4 * c);
    uint32_t q[4], t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t v = (bits >> (8 * k)) & 255u;                           // 8 bits -> the low bit of 8 nibbles
        v = (v | (v << 12)) & 0x000F000Fu;
        v = (v | (v << 6)) & 0x03030303u;
        v = (v | (v << 3)) & 0x11111111u;
        q[k] = (v << 3) + 0x22222222u;                                   // 1 -> 0xA (-1.0), 0 -> 0x2 (+1.0)
        t[k] = (v << 3) ^ 0xAAAAAAAAu;                                   // 1 -> 0x2 (+1.0), 0 -> 0xA (-1.0); x 16 by the block scale
    }
    const size_t o = (size_t)row * capT * MM_DB + (size_t)i * 16;
    *reinterpret_cast<uint4*>(eq + o) = make_uint4(q[0], q[1], q[2], q[3]);
    *reinterpret_cast<uint4*>(et + o) = make_uint4(t[0], t[1], t[2], t[3]);
}

// the two smallest of a finished tile (keys 32 H + row), widened and merged into the running pair g0 <= g1 of keys (H << 16 | trainIdx).
// Compiler-visible instructions only (v_min_f32 / v_med3_f32; the Makefile builds this file with -fno-honor-nans, otherwise every operand is
// canonicalised by an extra v_max first): the compiler then places the wait states an MFMA result needs before its first VALU read, and the
// scheduler can interleave these instructions with the next tile's MFMAs.
__device__ __forceinline__ void mm_merge2(uint32_t& a0, uint32_t& a1, uint32_t b0, uint32_t b1)     // (a0 <= a1), (b0 <= b1) -> the two smallest of the four
{
    const uint32_t m = max(a0, b0);
    a0 = min(a0, b0);
    a1 = min(min(m, a1), b1);
}
__device__ __forceinline__ void mm_track(const v16f& acc, uint32_t t32, uint32_t& g0, uint32_t& g1)
{
    // the two smallest of 16 non-negative floats by the sequential form: (l0, l1) <- (min(x, l0), med3(x, l0, l1)), 2 instructions per value
    float l0 = __builtin_fminf(acc[0], acc[1]), l1 = __builtin_fmaxf(acc[0], acc[1]);
#pragma unroll
    for (int i = 2; i < 16; i++) { l1 = __builtin_amdgcn_fmed3f(acc[i], l0, l1); l0 = __builtin_fminf(acc[i], l0); }
    const uint32_t k0 = (uint32_t)l0, k1 = (uint32_t)l1;               // v_cvt_u32_f32 (a masked row's 2^20 widens to distance 0x8000)
    const uint32_t G0 = ((k0 >> 5) << 16) | ((k0 & 31u) | t32), G1 = ((k1 >> 5) << 16) | ((k1 & 31u) | t32);
    mm_merge2(g0, g1, G0, G1);
}

// Block (frame, ref r, query block qb): queries [256 qb, +256) of reference frame `ref` against every descriptor of the current frame.
// XCD-aware 1-D grid: workgroup i runs on XCD i % 8, and the i / 8-th workgroup of an XCD is block (frame = XCD's first frame + (i / 8) / Y, y = (i / 8) % Y),
// Y = R x query blocks: an XCD owns a contiguous run of frames and walks it frame by frame, so the Y blocks that stream one frame's train tiles are
// resident together, and the reference rows (queries) a frame needs were the previous frames' -- both come from that XCD's L2, not from HBM again
// (with the frame as the fast grid index the kernel moved 1.6 GB per 250 frames and was memory bound at half the matrix rate).
__global__ void __launch_bounds__(256, 2)
match_mfma_kernel(const uint8_t* __restrict__ eq, const uint8_t* __restrict__ et, const int32_t* __restrict__ nkp,
                  int f0, int n, int R, int hist, int capT, int qblocks, int frames_per_xcd, uint2* __restrict__ knn)
{
    __shared__ __attribute__((aligned(16))) uint4 ring0[MM_TILE / 16];       // the ring as four objects: the compiler then knows that a DMA into one
    __shared__ __attribute__((aligned(16))) uint4 ring1[MM_TILE / 16];       // slot does not alias the fragment reads of another
    __shared__ __attribute__((aligned(16))) uint4 ring2[MM_TILE / 16];
    __shared__ __attribute__((aligned(16))) uint4 ring3[MM_TILE / 16];
    const int Y = R * qblocks, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int frame = xcd * frames_per_xcd + j / Y, y = j % Y;
    if (frame >= n) return;
    const int r = y / qblocks, qb = y - r * qblocks;
    const int f = f0 + frame, cur = hist + f, ref = cur - R + r;
    const int nq = ref >= 0 ? nkp[ref] : -1, nt = nkp[cur];
    if (nq < 0 || nt < 2 || qb * MM_QB >= nq) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = lane >> 5;
    const int q0 = qb * MM_QB + wv * 64;
    // B fragments: the wave's 64 queries, all 4 K-steps, stay in registers
    v4i b[2][MM_KS];
    {
        const uint4* qp = reinterpret_cast<const uint4*>(eq + ((size_t)ref * capT + q0) * MM_DB) + lane;
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int s = 0; s < MM_KS; s++) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (q0 + u * 32 < capT) v = qp[(u * MM_TILE + s * 1024) / 16];
                b[u][s] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
            }
    }
    v16f cc;                                                            // C of every tile: 4096 + the train row inside the tile
#pragma unroll
    for (int i = 0; i < 16; i++) cc[i] = (float)(4096 + (i & 3) + 8 * (i >> 2) + 4 * h);
    const int ntile = (nt + 31) >> 5;
    uint32_t g0[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, g1[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    // Train tiles go global -> LDS by LDS-DMA (no staging registers, no ds_write) through a ring of four 4 KB slots, up to four tiles ahead of the MFMAs:
    // with a register-staged double buffer the L2 latency of the next tile was exposed at every barrier and the kernel ran at half the matrix
    // rate.  Every thread issues exactly one DMA instruction per tile (clamped to the last tile past the end: same bytes, a slot nobody reads),
    // so "tile T has landed" is always vmcnt(2) + the barrier.
    const auto rsT = __builtin_amdgcn_make_buffer_rsrc((void*)(et + (size_t)cur * capT * MM_DB), 0, (unsigned)capT * MM_DB, 0x00020000);
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    auto dma = [&](int T, uint4* ring) {
        if (MM_ABLATE & 4) return;
        const unsigned o = (unsigned)min(T, ntile - 1) * MM_TILE + (unsigned)(wvu * 64 + lane) * 16u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsT, (lds_ptr_t)&ring[wvu * 64], 16, o, 0, 0, 0);
    };
    // the four A fragments of a tile, LDS -> registers
    auto frags = [&](const uint4* buf, v4i (&a)[MM_KS]) {
#pragma unroll
        for (int s = 0; s < MM_KS; s++) { const uint4 v = buf[s * 64 + lane]; a[s] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w}; }
    };
    // the 8 MFMAs of a tile whose fragments are in registers, into (x0, x1): A = train nibbles with block scale 2^4 (E8M0 131), B = query nibbles, scale 1 (127)
    auto mfma4 = [](const v4i& a, const v4i& bq, const v16f& c) {
        const v8i a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {bq[0], bq[1], bq[2], bq[3], 0, 0, 0, 0};          // FP4 operands are the low four registers
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 0x83838383, 0, 0x7F7F7F7F);
    };
    auto issue = [&](const v4i (&a)[MM_KS], v16f& x0, v16f& x1) {
        if (MM_ABLATE & 8) { x0 = cc; x1 = cc; for (int s = 0; s < MM_KS; s++) { x0[s] += (float)a[s][0]; x1[s] += (float)a[s][1]; } return; }
        x0 = mfma4(a[0], b[0][0], cc);
        x1 = mfma4(a[0], b[1][0], cc);
#pragma unroll
        for (int s = 1; s < MM_KS; s++) {
            x0 = mfma4(a[s], b[0][s], x0);
            x1 = mfma4(a[s], b[1][s], x1);
        }
    };
    // the MFMAs of a tile are issued first and the key tracking of the tile before it (VALU only, other registers) runs under them
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < 2 * MM_KS; i++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 12, 0); }
    };
    auto track = [&](const v16f& x0, const v16f& x1, int T) {
        if (MM_ABLATE & 1) { g0[0] = min(g0[0], __float_as_uint(x0[0])); g0[1] = min(g0[1], __float_as_uint(x1[0])); return; }
        mm_track(x0, (uint32_t)T << 5, g0[0], g1[0]); mm_track(x1, (uint32_t)T << 5, g0[1], g1[1]);
    };
    auto track_last = [&](v16f x0, v16f x1, int T) {                    // the last tile: its padded train rows (>= nt) never win
        const int valid = nt - T * 32;
#pragma unroll
        for (int i = 0; i < 16; i++) if ((i & 3) + 8 * (i >> 2) + 4 * h >= valid) { x0[i] = 1048576.0f; x1[i] = 1048576.0f; }       // a key no descriptor reaches (and finite: this file is built with -fno-honor-nans)
        track(x0, x1, T);
    };
    // Step T: tile T + 1 has landed (vmcnt(2): the DMAs of T + 2 and T + 3 stay in flight) and this wave holds the fragments of tile T
    // (lgkmcnt(0)); after the barrier that is true of every wave, so slot T & 3 is free for the DMA of tile T + 4.  The fragments of tile T + 1
    // are read into the other register set, then the MFMAs of tile T are issued from registers -- no LDS latency in front of them -- with the key
    // tracking of tile T - 1 under them.
#define MM_STEP(T_, rd_, wr_, acur, anxt, xn0, xn1, TRACK) { \
        __builtin_amdgcn_s_waitcnt((MM_ABLATE & 4) ? 0x007F : 0x0072);    /* vmcnt(2) lgkmcnt(0) */ \
        if (!(MM_ABLATE & 2)) __builtin_amdgcn_s_barrier();              /* bare: __syncthreads' fence would drain every DMA in flight */ \
        frags(rd_, anxt); \
        dma((T_) + 4, wr_); \
        __builtin_amdgcn_sched_barrier(0); \
        issue(acur, xn0, xn1); TRACK; interleave(); \
        __builtin_amdgcn_sched_barrier(0); }
    v16f A0, A1, B0, B1;
    v4i aA[MM_KS], aB[MM_KS];
    dma(0, ring0); dma(1, ring1); dma(2, ring2); dma(3, ring3);
    if (!(MM_ABLATE & 4)) __builtin_amdgcn_s_waitcnt(0x0F73);           // vmcnt(3): tile 0
    if (!(MM_ABLATE & 2)) __builtin_amdgcn_s_barrier();
    frags(ring0, aA);
    MM_STEP(0, ring1, ring0, aA, aB, A0, A1, (void)0)
    int T = 1;                                                          // the next tile; T = 1 (mod 4) here and after the loop
    for (; T + 3 < ntile; T += 4) {
        MM_STEP(T, ring2, ring1, aB, aA, B0, B1, track(A0, A1, T - 1))
        MM_STEP(T + 1, ring3, ring2, aA, aB, A0, A1, track(B0, B1, T))
        MM_STEP(T + 2, ring0, ring3, aB, aA, B0, B1, track(A0, A1, T + 1))
        MM_STEP(T + 3, ring1, ring0, aA, aB, A0, A1, track(B0, B1, T + 2))
    }
    if (T < ntile) MM_STEP(T, ring2, ring1, aB, aA, B0, B1, track(A0, A1, T - 1))
    if (T + 1 < ntile) MM_STEP(T + 1, ring3, ring2, aA, aB, A0, A1, track(B0, B1, T))
    if (T + 2 < ntile) MM_STEP(T + 2, ring0, ring3, aB, aA, B0, B1, track(A0, A1, T + 1))
#undef MM_STEP
    if ((ntile - 1) & 1) track_last(B0, B1, ntile - 1); else track_last(A0, A1, ntile - 1);
    __builtin_amdgcn_s_waitcnt(0x0070);                                 // the surplus DMA and fragment reads of the tail are done before the block's LDS is released
    // the lane halves hold different train rows of the same query: merge lane l with lane l ^ 32, then lanes 0..31 write accumulator 0's
    // queries and lanes 32..63 accumulator 1's
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const uint32_t o0 = __shfl_xor(g0[u], 32), o1 = __shfl_xor(g1[u], 32);
        const uint32_t m = max(g0[u], o0);
        g0[u] = min(g0[u], o0);
        g1[u] = min(min(m, g1[u]), o1);
    }
    const uint32_t k0 = h ? g0[1] : g0[0], k1 = h ? g1[1] : g1[0];
    const int qi = q0 + lane;                                            // lane 32 + j: column j of accumulator 1 = query q0 + 32 + j
    const int slot = f * R + r;
    if (qi < nq) knn[(size_t)slot * capT + qi] = make_uint2(k0, k1);
}

// ratio test + compaction in ascending queryIdx of one (frame, ref) pair from its knn keys (one block per pair)
__global__ void __launch_bounds__(256)
match_compact_kernel(const uint2* __restrict__ knn, const int32_t* __restrict__ nkp, int f0, int R, int hist, int capT, double ratio, int cap,
                     ssm_dmatch* __restrict__ out, int32_t* __restrict__ nout)
{
    __shared__ int wcnt[4];
    const int slot = (f0 + blockIdx.x / R) * R + blockIdx.x % R;
    const int cur = hist + f0 + blockIdx.x / R, ref = cur - R + blockIdx.x % R;
    const int nq = ref >= 0 ? nkp[ref] : -1, nt = nkp[cur];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (nq < 0 || nt < 2) { if (tid == 0) nout[slot] = -1; return; }
    const uint2* kk = knn + (size_t)slot * capT;
    ssm_dmatch* o = out + (size_t)slot * cap;
    int base = 0;
    for (int q0 = 0; q0 < nq; q0 += 256) {
        const int qi = q0 + tid;
        uint2 k = make_uint2(0, 0);
        if (qi < nq) k = kk[qi];
        const int d0 = k.x >> 16, i0 = k.x & 0xFFFF, d1 = k.y >> 16;
        const bool keep = (qi < nq) && ((double)(float)d0 < ratio * (double)(float)d1);      // orb.cpp:25
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wcnt[wv] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wv; w++) off += wcnt[w];
        if (keep) {
            const int kpos = off + __popcll(bal & ((1ull << lane) - 1ull));
            if (kpos < cap) { ssm_dmatch mm; mm.queryIdx = qi; mm.trainIdx = i0; mm.imgIdx = 0; mm.distance = (float)d0; o[kpos] = mm; }
        }
        base += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    if (tid == 0) nout[slot] = base;
}
hipError_t k_match_expand(const uint8_t* desc, const int32_t* nkp, int row0, int nrows, int cap, int capT, uint8_t* eq, uint8_t* et, hipStream_t s)
{
    if (nrows <= 0) return hipSuccess;
    match_expand_kernel<<<dim3((capT * 8 + 255) / 256, nrows), 256, 0, s>>>(desc, nkp, row0, cap, capT, eq, et);
    return hipGetLastError();
}
hipError_t k_match_seq_mfma(const uint8_t* eq, const uint8_t* et, const int32_t* nkp, int f0, int n, int R, int hist, double ratio, int cap, int capT,
                            void* knn, ssm_dmatch* out, int32_t* nout, hipStream_t s)
{
    const int qblocks = (capT + MM_QB - 1) / MM_QB;
    const int fpx = (n + 7) >> 3;
    match_mfma_kernel<<<8 * fpx * R * qblocks, 256, 0, s>>>(eq, et, nkp, f0, n, R, hist, capT, qblocks, fpx, reinterpret_cast<uint2*>(knn));
    match_compact_kernel<<<n * R, 256, 0, s>>>(reinterpret_cast<const uint2*>(knn), nkp, f0, R, hist, capT, ratio, cap, out, nout);
    return hipGetLastError();
}
