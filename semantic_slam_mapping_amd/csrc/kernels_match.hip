// kernels_match.hip -- K6: OrbFeature::match (/root/reference/src/orb.cpp:16-29) = cv::BFMatcher(NORM_HAMMING)
// knnMatch(k=2) + Lowe ratio test, for gfx950.  One block per (query frame, train frame) pair (two in the sequence kernel): the train
// descriptors are staged in LDS (32 B each, read back as wave-wide broadcasts), every lane owns one query descriptor
// in 8 VGPRs, distance = 8 x (v_xor + v_bcnt accumulate); the kept matches are compacted in ascending queryIdx with
// wave ballots.  Integer work, VALU-bound: no MFMA by design.  Tie rule = oracle/match.c (strict '<' scan in train
// order, so equal distances resolve to the lower trainIdx).
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
