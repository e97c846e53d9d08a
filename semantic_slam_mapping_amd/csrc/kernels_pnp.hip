// kernels_pnp.hip -- the RGB-D pose chain on the device: Tracker::trackRefFrame (reference src/track.cpp:140-200) + PnPSolver::solvePnP (src/pnp.cpp:5-118)
// for a run of consecutive frames of an ssm_seq_process call, ONE block of 1024 threads walking the frames in order.
// The chain is serial by construction (frame f starts from frame f-1's pose and un-projects its reference frames' features with their solved poses);
// what parallelism there is lives inside a frame: the correspondences of the <= tracker_ref_frames match tables (ordered compaction by block scans), and
// every pass over the edges of the Levenberg iterations (chi2 pass; chi2 + normal equations in one pass) -- thread i owns edges i, i + 1024, ... and the sums
// follow the LANE ORDER of include/ssm/pnp_core.h (64-lane neighbour-first tree, then the 16 group totals in order), which is what makes the result the same bits as the host
// class and the oracle.  The small dense algebra (6 x 6 L D L^T, exp map, Levenberg bookkeeping) is wave 0's; the estimate, the system and the tracker's
// state live in LDS, and so does the edge list (24 bytes per edge: the measurement and the point are floats to begin with) when it fits.
// All matches / features stay on the device; the host orchestrator (ssm_track.hip) only moves the tracker state.
// The kernel handles the REGULAR case -- state OK and the refFrames deque = the frames directly in front of the current one, all inside the match-table
// window -- and stops behind the first frame that fails to track; the host path (same arithmetic) takes over until the deque is regular again.
#include "pnp_chain.h"
#include <cfloat>
#include <mutex>
#include <set>
#include <utility>

using namespace ssm_pnp;
#define PC_T 1024
#define PC_SPEC 8                 // trials of a rejected streak evaluated together (pc_optimize)
#define PC_RBUF 3072              // doubles (24 KB of LDS)
#ifndef PC_SB_ROWSCAN
#define PC_SB_ROWSCAN false     // (the DPP-row sum for the 36-value exchange of pc_chi_spec_build: 148 ms against 141)
#endif
#define PC_NVAL 64                // sums a block keeps per contract group
#define PC_SYS2 (NACC + 1 + PC_SPEC)   // first column of the speculative system
#ifndef PC_FIRST_SPEC
#define PC_FIRST_SPEC 3           // cluster form: trials solved and evaluated with an iteration's first trial (1 = the first trial alone).  Measured, ms of chain per 400
                                  // frames, before the first round also built the next system: 1: 168.4 - 170.9, 2: 166.5, 4: 167.3, 8: 184.7 -- the candidates behind an
                                  // accepted first trial are wasted work; with pc_chi_spec_build (their walks run on waves the system's parts leave idle): 2: 141.5, 3: 137.7, 4: 139.7
#endif
// an edge as the passes read it: 24 bytes (the Edge of pnp_core.h is 72; its error lives in a separate global array).  meta = id | level << 16 | robust << 17
struct LEdge { float X[3], u, v; uint32_t meta; };
#define LE_LEVEL (1u << 16)
#define LE_ROBUST (1u << 17)
// the edge list of a block: in LDS when it fits (address space 3: ds_read / ds_write), else in global scratch (1).  A pointer chosen at run time is a generic one and every
// access a flat_ instruction (both counters, the texture path's latency in front of the LDS); the kernels are instantiated per address space instead (round 5)
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
template <int AS>
struct EdgeMem {
    typedef __attribute__((address_space(AS))) uint32_t W;
    typedef __attribute__((address_space(AS))) uint2v W2;
    W* p;
    __device__ __forceinline__ LEdge ld(int i) const
    {   // 24 bytes at 8-byte alignment: 8 + 16 or 16 + 8
        const W2* q = reinterpret_cast<const W2*>(p + (size_t)i * 6);
        const uint2v a = q[0], b = q[1], c = q[2];
        LEdge l; l.X[0] = __uint_as_float(a.x); l.X[1] = __uint_as_float(a.y); l.X[2] = __uint_as_float(b.x); l.u = __uint_as_float(b.y); l.v = __uint_as_float(c.x); l.meta = c.y;
        return l;
    }
    __device__ __forceinline__ void st(int i, const LEdge& l) const
    {
        W2* q = reinterpret_cast<W2*>(p + (size_t)i * 6);
        q[0] = uint2v{__float_as_uint(l.X[0]), __float_as_uint(l.X[1])}; q[1] = uint2v{__float_as_uint(l.X[2]), __float_as_uint(l.u)}; q[2] = uint2v{__float_as_uint(l.v), l.meta};
    }
    __device__ __forceinline__ uint32_t meta(int i) const { return p[(size_t)i * 6 + 5]; }
    __device__ __forceinline__ void set_meta(int i, uint32_t m) const { p[(size_t)i * 6 + 5] = m; }
};
__device__ __forceinline__ Edge pc_expand(const LEdge& l)
{
    Edge e; e.id = (int32_t)(l.meta & 0xFFFFu); e.level = (l.meta & LE_LEVEL) ? 1 : 0; e.robust = (l.meta & LE_ROBUST) ? 1 : 0; e.pad = 0;
    e.X[0] = l.X[0]; e.X[1] = l.X[1]; e.X[2] = l.X[2]; e.u = l.u; e.v = l.v; e.e0 = e.e1 = 0;
    return e;
}

#ifdef SSM_PNP_PROF
#define PROF_T0 long long pt_ = clock64();
#define PROF(k) { const long long n_ = clock64(); if (threadIdx.x == 0) sh.prof[k] += n_ - pt_; pt_ = n_; }
#define PROF_CNT(k) { if (threadIdx.x == 0) sh.prof[k] += 1; }
#define PROF2_T0 long long p2_ = clock64();
#define PROF2(k) { const long long n2_ = clock64(); if (threadIdx.x == 0) sh.prof[k] += n2_ - p2_; p2_ = n2_; }
#else
#define PROF2_T0
#define PROF2(k)
#define PROF_T0
#define PROF(k)
#define PROF_CNT(k)
#endif
struct PcShared {
#ifdef SSM_PNP_PROF
    long long prof[32];
#endif
    double red[NGROUP][PC_NVAL];       // [group][value]: 0 .. 27 the system at P (H, b, chi2), 28 .. 35 the chi2 of a round's candidates, 36 .. 63 the system at the first candidate (pc_chi_spec_build)
    double tot[PC_NVAL];
    // a rejected Levenberg trial is followed by trials whose damping is known in advance (lambda <- lambda nu, nu <- 2 nu until one is accepted): wave c solves
    // candidate c of such a streak and ONE pass over the edges evaluates all of them (pc_chi_spec)
    struct { double x[6]; Pose P; int solved, pad; } spec[PC_SPEC];
    double spec_lambda, spec_nu; int spec_n;
    double rbuf[PC_RBUF];         // cluster form: the chi2 terms of (candidate, edge slot) items computed by the waves whose lanes another block owns (pc_chi_spec)
    Pose P, saved, init;          // the estimate wave 0 publishes for the next pass; the one before the trial; the round's start value
    double speed[16], last[16], Tpred[16], T[16];   // the tracker's state and the frame's transforms (thread 0 writes them)
    int wcnt[NGROUP];
    int cont, term, acc0;         // loop controls of the Levenberg iteration, decided by wave 0 (acc0: the iteration's first trial was accepted)
    long long work[4];            // thread 0: fused passes, chi2 passes, active edges evaluated by each kind (ssm_tracker_work: the numerator of bench.py's pose-loop roofline)
    // the cluster form (round 4, gridDim.x = G > 1 blocks per chain): this block evaluates the edges of lanes [b 1024 / G, (b + 1) 1024 / G) only and the G blocks
    // trade their groups' partial sums through tagged granules in global memory (pc_lane_sum)
    unsigned long long* xmb; unsigned* xfail; unsigned xseq;
};
// exclusive position of `flag` among the block's threads in thread order, and the block total
__device__ __forceinline__ int pc_scan(bool flag, PcShared& sh, int& total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    if (lane == 0) sh.wcnt[wv] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NGROUP; k++) { const int c = sh.wcnt[k]; if (k < wv) off += c; tot += c; }
    __syncthreads();
    total = tot;
    return off + __popcll(bal & ((1ull << lane) - 1ull));
}
// the value `v` holds in lane ^ S.  S = 1, 2: DPP quad_perm (a VALU move, no LDS round trip).  TREE: the caller is the full butterfly, where every lane
// of an aligned 4- (8-) lane group already holds the same sum, so any lane of the sibling group serves: row_half_mirror (lane ^ 7) / row_mirror (lane ^ 15)
template <int S, bool TREE>
__device__ __forceinline__ double pc_xor_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (S == 1) { lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false); }
    else if constexpr (S == 2) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, false); }
    else if constexpr (S == 4 && TREE) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, false); }
    else if constexpr (S == 8 && TREE) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, false); }
    else { lo = __shfl_xor(lo, S, 64); hi = __shfl_xor(hi, S, 64); }
    return __hiloint2double(hi, lo);
}
// one halving step of the reduce-scatter: lanes with bit S clear keep a[0 .. H), the others a[H .. 2H), and add the partner's copy of what they keep
template <int S, int H>
__device__ __forceinline__ void pc_rs_step(double* a, bool up)
{
#pragma unroll
    for (int j = 0; j < H; j++) {
        const double keep = up ? a[j + H] : a[j], send = up ? a[j] : a[j + H];
        a[j] = keep + pc_xor_f64<S, false>(send);
    }
}
// lane sums of pnp_core.h: this thread's partial sums acc[NV] -> sh.tot[OFF .. OFF + NV), visible to the whole block when the function returns.
// The group (= wave) tree of one value adds sibling sub-tree sums level by level; floating-point addition commutes, so WHICH lane of a sub-tree does an
// addition does not matter.  One value: the plain butterfly.  28 values: a reduce-scatter -- at level k a lane keeps only the half of its values that its
// bit k selects (28 -> 14 -> 7 (+ 1 pad) -> 4 -> 2 -> 1) -- 29 additions and 31 exchanges per lane instead of 168 and 168 (the butterfly was 45 % of the
// fused pass's instructions and kept the LDS pipe busy with 336 ds_bpermute per wave).  Lane 0 of each wave (one value) / the lane that ends up with value v
// publishes the group sum, threads 0 .. NV-1 add the 16 group sums in group order.
template <int NV, int OFF, bool CL, bool ROWSCAN = (NV <= 8)> __device__ __forceinline__ void pc_lane_finish(PcShared& sh, int nv = NV, int skip_lo = 0, int skip_hi = 0);
template <int NV, int OFF, bool CL>
__device__ __forceinline__ void pc_lane_sum(double (&acc)[NV], PcShared& sh)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if constexpr (NV == 8) {
        // eight values: reduce-scatter over the first three levels (8 -> 4 -> 2 -> 1 per lane), the plain exchange for the rest; lane l < 8 ends with value
        // (l & 1) 4 + ((l >> 1) & 1) 2 + ((l >> 2) & 1)
        double a[8];
#pragma unroll
        for (int v = 0; v < 8; v++) a[v] = acc[v];
        pc_rs_step<1, 4>(a, lane & 1);
        pc_rs_step<2, 2>(a, lane & 2);
        pc_rs_step<4, 1>(a, lane & 4);
        double t = a[0];
        t = t + pc_xor_f64<8, false>(t); t = t + pc_xor_f64<16, false>(t); t = t + pc_xor_f64<32, false>(t);
        if (lane < 8) sh.red[wv][OFF + (lane & 1) * 4 + ((lane >> 1) & 1) * 2 + ((lane >> 2) & 1)] = t;
    } else if constexpr (NV == 1) {
        double a = acc[0];
        a = a + pc_xor_f64<1, true>(a); a = a + pc_xor_f64<2, true>(a); a = a + pc_xor_f64<4, true>(a);
        a = a + pc_xor_f64<8, true>(a); a = a + pc_xor_f64<16, true>(a); a = a + pc_xor_f64<32, true>(a);
        if (lane == 0) sh.red[wv][OFF] = a;
    } else {
        static_assert(NV == 28, "the halving sequences are written for 1, 8 and 28 values");
        double a[28];
#pragma unroll
        for (int v = 0; v < 28; v++) a[v] = acc[v];
        pc_rs_step<1, 14>(a, lane & 1);
        pc_rs_step<2, 7>(a, lane & 2);
        a[7] = 0.0;                                                              // pad: 7 -> 8
        pc_rs_step<4, 4>(a, lane & 4);
        pc_rs_step<8, 2>(a, lane & 8);
        pc_rs_step<16, 1>(a, lane & 16);
        const double t = a[0] + pc_xor_f64<32, false>(a[0]);
        const int sub = ((lane >> 2) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 4) & 1);
        const int v = (lane & 1) * 14 + ((lane >> 1) & 1) * 7 + sub;
        if (lane < 32 && sub < 7) sh.red[wv][OFF + v] = t;
    }
    pc_lane_finish<NV, OFF, CL>(sh);
}
// the block part of pc_lane_sum: sh.red[group][OFF .. OFF + NV) hold the group sums of this block's lanes (written by whoever computed them)
template <int NV, int OFF, bool CL, bool ROWSCAN>
__device__ __forceinline__ void pc_lane_finish(PcShared& sh, int nv, int skip_lo, int skip_hi)      // nv <= NV: the values in use (block-uniform; the others, and those in [skip_lo, skip_hi), are neither exchanged nor summed)
{
    PROF2_T0
    constexpr int PB = NV == 28 ? 8 : 16;
    constexpr int VS = NV > 32 ? 6 : 5;                                          // thread (g, v) = (tid >> VS, tid & (2^VS - 1)) where a group's values share a wave
    static_assert(NV <= 64, "a group's values: at most 64 (the exchange ring's stride)");
    // ROWSCAN (the passes of few values: 16 nv <= 128 threads): thread (v, g) = (tid / 16, tid % 16) takes group g's sum of value v and the row adds them (below);
    // otherwise thread (g, v) = (tid / 32, tid % 32) brings the sum to LDS and thread v walks the sixteen.  (The row form for the 28-value pass too: measured slower)
    static_assert(NGROUP == 16, "a DPP row is the sixteen groups");
    __syncthreads();
    PROF2(PB + 0)
    const int g = ROWSCAN ? (int)(threadIdx.x & 15) : (int)(threadIdx.x >> VS), v = ROWSCAN ? (int)(threadIdx.x >> 4) : (int)(threadIdx.x & ((1 << VS) - 1));
    const bool pollv = g < NGROUP && v < nv && !(v >= skip_lo && v < skip_hi);
    double r = 0.0; unsigned seq = 0;
    if constexpr (CL) {
        // every block holds the group sums of its own waves; a sum is published as two 8-byte {pass tag, 32 bits} granules (one agent-scope store each: the data is the
        // flag, cdna_hip_programming.md G16 R2) and every block polls all sixteen groups' granules, so that the ordered sum below sees the same sixteen numbers everywhere.
        // Four ring slots by pass number; a block can be at most one pass ahead of another (it needs everybody's sums to finish a pass).
        const int gpb = NGROUP / (int)gridDim.x, pg = threadIdx.x >> VS, pv = threadIdx.x & ((1 << VS) - 1);
        seq = sh.xseq; const unsigned long long tag = (unsigned long long)(seq + 1) << 32;
        typedef __attribute__((address_space(1))) unsigned long long gu64;          // (global_load / global_store, not flat: the ring is device memory)
        gu64* slot = (gu64*)sh.xmb + (size_t)(seq & 3u) * NGROUP * 64 * 2;
        if (pg < gpb && pv < nv && !(pv >= skip_lo && pv < skip_hi)) {
            const int gg = (int)blockIdx.x * gpb + pg;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(sh.red[gg][OFF + pv]);
                        const size_t gi_ = (size_t)(gg * 64 + pv);
            __hip_atomic_store(slot + gi_ * 2, tag | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(slot + gi_ * 2 + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        PROF2(PB + 1)
        if (pollv) {
            // two polls in flight, half a round trip apart (a poll that leaves just before the granule lands costs a whole round trip of ~1.2 k clocks otherwise):
            // the second leaves ~600 clocks behind the first, after that each is re-issued when its answer is in, which keeps the spacing
            const gu64* q = slot + (size_t)(g * 64 + v) * 2;
            unsigned long long lo = 0, hi = 0; bool ok = false;
            unsigned long long lo_a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi_a = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_sleep(9);
            for (unsigned spins = 0; spins < (1u << 21); ++spins) {
                const unsigned long long lo_b = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi_b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((lo_a >> 32) == (tag >> 32) && (hi_a >> 32) == (tag >> 32)) { lo = lo_a; hi = hi_a; ok = true; break; }
                lo_a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); hi_a = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((lo_b >> 32) == (tag >> 32) && (hi_b >> 32) == (tag >> 32)) { lo = lo_b; hi = hi_b; ok = true; break; }
                if ((spins & 1023u) == 1023u && __hip_atomic_load(sh.xfail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            }
            if (!ok) __hip_atomic_store(sh.xfail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (bounded: the chain then reports an invalid range instead of hanging)
            const double got = __longlong_as_double((long long)((lo & 0xFFFFFFFFull) | (hi << 32)));
            if constexpr (ROWSCAN) r = got; else sh.red[g][OFF + v] = got;
        }
        PROF2(PB + 2)
        if constexpr (!ROWSCAN) {
            __syncthreads();
            PROF2(PB + 3)
            if (threadIdx.x == 0) sh.xseq = seq + 1;
        }
    } else if constexpr (ROWSCAN) { if (pollv) r = sh.red[g][OFF + v]; }
    if constexpr (ROWSCAN) {
        // the sixteen group sums of a value sit in the sixteen lanes of a DPP row: p <- row_shr:1(p) + r, fifteen times, adds them in group order (lane g is final
        // after g steps; lane 0 receives -0.0, and -0.0 + r == r for every r) -- no trip through LDS and no barrier between the poll and the sum
        if (pollv) {
            double p = r;
#pragma unroll
            for (int k = 1; k < NGROUP; k++) {
                const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(p), 0x111, 0xF, 0xF, false);                   // row_shr:1; lane 0 of the row keeps `old`
                const int hi = __builtin_amdgcn_update_dpp((int)0x80000000, __double2hiint(p), 0x111, 0xF, 0xF, false);
                p = __hiloint2double(hi, lo) + r;
            }
            if (g == NGROUP - 1) sh.tot[OFF + v] = p;
        }
        PROF2(PB + 3)
        __syncthreads();
        if constexpr (CL) { if (threadIdx.x == 0) sh.xseq = seq + 1; }            // (read again behind the next call's first barrier)
    } else {
        if ((int)threadIdx.x < nv && !((int)threadIdx.x >= skip_lo && (int)threadIdx.x < skip_hi)) { const int vv = OFF + threadIdx.x; double s = sh.red[0][vv]; for (int gq = 1; gq < NGROUP; gq++) s = s + sh.red[gq][vv]; sh.tot[vv] = s; }
        __syncthreads();
    }
    PROF2(PB + 4)
}
// the lanes whose edges this block evaluates (all of them in the one-block form)
template <bool CL> __device__ __forceinline__ bool pc_mine() { if constexpr (CL) return (int)(threadIdx.x / (PC_T / gridDim.x)) == (int)blockIdx.x; else return true; }
// the robustified chi2 of the active edges at P (leaves every active edge's error in the edge) -> sh.tot[NACC] (H and b in sh.tot[0 .. 26] stay)
template <bool CL, class EM>
__device__ __forceinline__ void pc_chi(const EM L, double2* err, int ne, const Pose& P, const Camera& k, double delta, PcShared& sh)
{
    double acc[1] = {0.0};
    if (pc_mine<CL>())
    for (int i = threadIdx.x; i < ne; i += PC_T) {
        const LEdge l = L.ld(i);
        if (!(l.meta & LE_LEVEL)) { Edge e = pc_expand(l); acc[0] += edge_rho(e, P, k, delta); err[i] = make_double2(e.e0, e.e1); }
    }
    pc_lane_sum<1, NACC, CL>(acc, sh);
}
// the robustified chi2 of the active edges at the poses of candidates 0 .. n-1 (sh.spec[c].P) -> sh.tot[NACC + 1 + c].  The edges' stored errors are NOT touched: every
// lm_optimize ends with a pc_chi at its final estimate, and nothing reads an error before that (edge_accumulate follows an edge_rho at the same estimate)
template <bool CL, class EM>
__device__ __forceinline__ void pc_chi_spec(const EM L, int ne, int n, const Camera& k, double delta, PcShared& sh)
{
    if constexpr (CL) {
        // The cluster form: this block owns the lanes of gpb = 16 / G contract groups, i.e. gpb of its sixteen waves have edges and the others would idle.  Here
        // all of them work: wave (set, gi) stands for group gi of the block and takes the items (candidate c, edge slot j) with (c J + j) mod G = set -- ONE edge
        // of ONE candidate at a time -- and leaves the edge's chi2 term in LDS; then the set of candidate c adds a lane's terms in slot order (0 + r0 + r1 ...:
        // the lane's running sum of pnp_core.h; an inactive or missing edge contributes +0, which changes no sum of non-negative terms) and runs the group tree.
        const int G = (int)gridDim.x, gpb = NGROUP / G;
        const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, set = wv / gpb, gi = wv - set * gpb;
        const int gg = (int)blockIdx.x * gpb + gi, lane0 = gg * GROUP + lane, bl = gi * GROUP + lane;       // contract group / lane; lane index inside the block
        const int J = (ne + PC_T - 1) / PC_T, per = gpb * GROUP, room = PC_RBUF / per;                      // edge slots per lane; items the buffer holds
        const int CC = J <= room ? room / J : 0;                                                            // candidates per filling of the buffer
        PROF2_T0
        for (int c0 = 0; c0 < n; c0 += (CC > 0 ? CC : 1)) {
            if (CC > 0) {
                const int nc = n - c0 < CC ? n - c0 : CC;
                for (int it = set; it < nc * J; it += G) {
                    const int c = it / J, j = it - c * J, i = lane0 + j * PC_T;
                    double r = 0.0;
                    if (i < ne) { const LEdge l = L.ld(i); if (!(l.meta & LE_LEVEL)) { Edge e = pc_expand(l); r = edge_rho(e, sh.spec[c0 + c].P, k, delta); } }
                    sh.rbuf[it * per + bl] = r;
                }
                PROF2(25)
                __syncthreads();
                PROF2(26)
                for (int c = set; c < nc; c += G) {
                    double a = 0.0;
                    for (int j = 0; j < J; j++) a += sh.rbuf[(c * J + j) * per + bl];
                    a = a + pc_xor_f64<1, true>(a); a = a + pc_xor_f64<2, true>(a); a = a + pc_xor_f64<4, true>(a);
                    a = a + pc_xor_f64<8, true>(a); a = a + pc_xor_f64<16, true>(a); a = a + pc_xor_f64<32, true>(a);
                    if (lane == 0) sh.red[gg][NACC + 1 + c0 + c] = a;
                }
                PROF2(27)
                __syncthreads();                                                    // (the buffer is refilled by the next group of candidates)
                PROF2(28)
            } else {                                                                // a list too long for the buffer: the owning waves walk their edges, one candidate at a time
                if (set == 0) {
                    double a = 0.0;
                    for (int i = lane0; i < ne; i += PC_T) { const LEdge l = L.ld(i); if (!(l.meta & LE_LEVEL)) { Edge e = pc_expand(l); a += edge_rho(e, sh.spec[c0].P, k, delta); } }
                    a = a + pc_xor_f64<1, true>(a); a = a + pc_xor_f64<2, true>(a); a = a + pc_xor_f64<4, true>(a);
                    a = a + pc_xor_f64<8, true>(a); a = a + pc_xor_f64<16, true>(a); a = a + pc_xor_f64<32, true>(a);
                    if (lane == 0) sh.red[gg][NACC + 1 + c0] = a;
                }
            }
        }
        pc_lane_finish<PC_SPEC, NACC + 1, CL>(sh, n);
        return;
    }
    double acc[PC_SPEC];
#pragma unroll
    for (int c = 0; c < PC_SPEC; c++) acc[c] = 0.0;
    {
        // one block: every wave has edges.  Two candidates per walk over them (two independent chains, few enough registers to stay out of scratch); a walk
        // whose candidates do not exist is skipped (n is block-uniform)
#pragma unroll
        for (int c0 = 0; c0 < PC_SPEC; c0 += 2) {
            if (c0 >= n) continue;
            const Pose Pa = sh.spec[c0].P, Pb = sh.spec[c0 + 1 < n ? c0 + 1 : c0].P;
            const bool two = c0 + 1 < n;
            double a0 = 0.0, a1 = 0.0;
            for (int i = threadIdx.x; i < ne; i += PC_T) {
                const LEdge l = L.ld(i);
                if (!(l.meta & LE_LEVEL)) {
                    Edge e = pc_expand(l);
                    a0 += edge_rho(e, Pa, k, delta);
                    if (two) a1 += edge_rho(e, Pb, k, delta);
                }
            }
            acc[c0] = a0; acc[c0 + 1] = a1;
        }
    }
    pc_lane_sum<PC_SPEC, NACC + 1, CL>(acc, sh);
}
// chi2 and the normal equations at P in ONE pass over the edges (the host evaluates active_chi2 and build_system one after the other at the same
// estimate: the same per-edge values, the same lane sums) -> sh.tot[0 .. 26] = H (lower triangle) and b, sh.tot[27] = chi2
//
// The cluster form splits the 28 sums over four waves per contract group (round 5).  A block owns gpb = 16 / G groups, so gpb of its sixteen waves had edges and
// walked them with 28 accumulators each (56 registers of the 128 a wave of a 1024-thread block may hold: the walk spilled) while the others idled.  Every sum is its
// own chain of additions -- edge by edge in slot order, row 0's term then row 1's -- so WHICH wave owns a sum changes no bit: wave (part, group) walks the group's
// edges for the sums of its part only (the error, the Jacobian and the weight are recomputed by each, the 40 products and additions of edge_accumulate are shared
// out), and runs the eight-value reduce-scatter instead of the 28-value one.  PC_PART_Q[part][slot] = index into acc / sh.tot (-1: unused slot):
//   part 0: H rows 0-1 + b0 b1 b2;  part 1: H rows 2-3 + b3;  part 2: H row 4 + b4 b5 + chi2 (and the edge's stored error);  part 3: H row 5
// (balanced by the products and additions each costs: row 3 only has row 0's terms and row 4 only row 1's -- the literal zeros of J -- ; 42 / 40 / 28 / 42 with the
// Jacobian entries each needs; rows 0-2 + b0 b1 | row 3 + b2 b3 + chi2 | row 4 + b4 | row 5 + b5 was 52 / 28 / 23 / 46)
// PC_NPARTS = 2 (a wave's 14 sums: H rows 0-3 + b0 .. b3 | H rows 4-5 + b4 b5 + chi2; sixteen-value group tree) re-computes an edge's error, Jacobian and weight twice
// instead of four times: the walks are bound by instruction issue once the first round's candidate walks share the SIMDs (pc_chi_spec_build).
#ifndef PC_NPARTS
#define PC_NPARTS 2
#endif
#if PC_NPARTS == 4
#define PC_PSLOTS 8
#define PC_CHI_PART 2
__device__ constexpr int PC_PART_Q[4][8] = {{0, 1, 2, 21, 22, 23, -1, -1}, {3, 4, 5, 6, 7, 8, 9, 24}, {10, 11, 12, 13, 14, 25, 26, NACC}, {15, 16, 17, 18, 19, 20, -1, -1}};
#else
#define PC_PSLOTS 16
#define PC_CHI_PART 1
__device__ constexpr int PC_PART_Q[2][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 21, 22, 23, 24, -1, -1}, {10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 25, 26, NACC, -1, -1}};
#endif
constexpr int pc_part_slot(int part, int q) { for (int k = 0; k < PC_PSLOTS; k++) if (PC_PART_Q[part][k] == q) return k; return -1; }
// edge_accumulate of pnp_core.h restricted to the sums of PART: the same expressions in the same order (the loops unroll, the tests on q fold away and the
// Jacobian entries a part does not use are never computed)
template <int PART>
__device__ __forceinline__ void pc_edge_accumulate_part(const Edge& e, const Pose& P, const Camera& k, double delta, double (&acc)[PC_PSLOTS])
{
    double p[3]; edge_map(e, P, p);
    const double x = p[0], y = p[1], iz = 1.0 / p[2], iz2 = iz * iz;
    const double J[2][6] = {{x * y * iz2 * k.fx, -(1 + (x * x * iz2)) * k.fx, y * iz * k.fx, -iz * k.fx, 0, x * iz2 * k.fx},
                            {(1 + y * y * iz2) * k.fy, -x * y * iz2 * k.fy, -x * iz * k.fy, 0, -iz * k.fy, y * iz2 * k.fy}};
    double w = 1.0;
    if (e.robust) { double r0; huber(edge_chi2(e), delta, r0, w); }
    const double er[2] = {e.e0, e.e1};
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const double wr = -er[r] * w;
        const int z = r == 0 ? 4 : 3;
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            if (a == z) { q += a + 1; continue; }
            { const int sb = pc_part_slot(PART, 21 + a); if (sb >= 0) acc[sb] += J[r][a] * wr; }
            const double jw = J[r][a] * w;
#pragma unroll
            for (int c = 0; c <= a; c++) { if (c != z) { const int sq = pc_part_slot(PART, q); if (sq >= 0) acc[sq] += jw * J[r][c]; } q++; }
        }
    }
}
// (the edge's error is not stored: in the cluster form pc_solve recomputes every active edge's error after lm_optimize, and nothing reads one before)
template <int PART, class EM>
__device__ __forceinline__ void pc_build_part(const EM L, int ne, int lane0, const Pose& P, const Camera& k, double delta, double (&acc)[PC_PSLOTS])
{
    for (int i = lane0; i < ne; i += PC_T) {
        const LEdge l = L.ld(i);
        if (!(l.meta & LE_LEVEL)) {
            Edge e = pc_expand(l);
            const double rho = edge_rho(e, P, k, delta);
            if constexpr (PART == PC_CHI_PART) acc[pc_part_slot(PC_CHI_PART, NACC)] += rho;
            pc_edge_accumulate_part<PART>(e, P, k, delta, acc);
        }
    }
}
// the sums of `part` at P for this wave's group gg -> sh.red[gg][OFF + q] (the part's walk, then the group tree of its values: the halving of pc_lane_sum)
template <int OFF, class EM>
__device__ __forceinline__ void pc_build_group(const EM L, int ne, int part, int gg, const Pose& P, const Camera& k, double delta, PcShared& sh)
{
    const int lane = threadIdx.x & 63, lane0 = gg * GROUP + lane;
    double a[PC_PSLOTS];
#pragma unroll
    for (int q = 0; q < PC_PSLOTS; q++) a[q] = 0.0;
#if PC_NPARTS == 4
    switch (part) {
        case 0: pc_build_part<0>(L, ne, lane0, P, k, delta, a); break;
        case 1: pc_build_part<1>(L, ne, lane0, P, k, delta, a); break;
        case 2: pc_build_part<2>(L, ne, lane0, P, k, delta, a); break;
        default: pc_build_part<3>(L, ne, lane0, P, k, delta, a); break;
    }
    // lane l < 8 ends with slot (l & 1) 4 + ((l >> 1) & 1) 2 + ((l >> 2) & 1)
    pc_rs_step<1, 4>(a, lane & 1);
    pc_rs_step<2, 2>(a, lane & 2);
    pc_rs_step<4, 1>(a, lane & 4);
    double t = a[0];
    t = t + pc_xor_f64<8, false>(t); t = t + pc_xor_f64<16, false>(t); t = t + pc_xor_f64<32, false>(t);
    if (lane < 8) { const int q = PC_PART_Q[part][(lane & 1) * 4 + ((lane >> 1) & 1) * 2 + ((lane >> 2) & 1)]; if (q >= 0) sh.red[gg][OFF + q] = t; }
#else
    if (part == 0) pc_build_part<0>(L, ne, lane0, P, k, delta, a); else pc_build_part<1>(L, ne, lane0, P, k, delta, a);
    // lane l < 16 ends with slot (l & 1) 8 + ((l >> 1) & 1) 4 + ((l >> 2) & 1) 2 + ((l >> 3) & 1)
    pc_rs_step<1, 8>(a, lane & 1);
    pc_rs_step<2, 4>(a, lane & 2);
    pc_rs_step<4, 2>(a, lane & 4);
    pc_rs_step<8, 1>(a, lane & 8);
    double t = a[0];
    t = t + pc_xor_f64<16, false>(t); t = t + pc_xor_f64<32, false>(t);
    if (lane < 16) { const int q = PC_PART_Q[part][(lane & 1) * 8 + ((lane >> 1) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 3) & 1)]; if (q >= 0) sh.red[gg][OFF + q] = t; }
#endif
}
template <bool CL, class EM>
__device__ __forceinline__ void pc_chi_build(const EM L, double2* err, int ne, const Pose& P, const Camera& k, double delta, PcShared& sh)
{
    PROF2_T0
    if constexpr (CL) {
        const int G = (int)gridDim.x, gpb = NGROUP / G;
        const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, set = wv / gpb, gi = wv - set * gpb;
        const int gg = (int)blockIdx.x * gpb + gi;                                                          // contract group of this wave's edges
        for (int part = set; part < PC_NPARTS; part += G) pc_build_group<0>(L, ne, part, gg, P, k, delta, sh);       // (G = 2: two parts per wave)
        PROF2(24)
        pc_lane_finish<NACC + 1, 0, CL>(sh);
        return;
    }
    double acc[NACC + 1];
#pragma unroll
    for (int q = 0; q < NACC + 1; q++) acc[q] = 0.0;
    for (int i = threadIdx.x; i < ne; i += PC_T) {
        const LEdge l = L.ld(i);
        if (!(l.meta & LE_LEVEL)) { Edge e = pc_expand(l); acc[NACC] += edge_rho(e, P, k, delta); edge_accumulate(e, P, k, delta, acc); err[i] = make_double2(e.e0, e.e1); }
    }
    pc_lane_sum<NACC + 1, 0, CL>(acc, sh);
}
// The first round of an iteration in the eight-block cluster form (round 5): the chi2 of its candidates AND the normal equations at the FIRST candidate's estimate in one
// pass.  The first trial is accepted in two iterations out of three (68 % on the bench's stream), and the iteration that follows an accepted trial starts with exactly
// this system -- chi2_build at the estimate the trial produced -- so building it here, on waves that a chi2 pass of two candidates leaves idle, takes a whole fused pass
// and its exchange off the critical path of those iterations; a rejected first trial drops it.  Waves (part, group) walk the group's edges at sh.spec[0].P for
// the sums of their part -> columns PC_SYS2 + q (the part with chi2 also fills the first candidate's column NACC + 1: the same sum); the next sets of waves walk them
// for the chi2 of candidates 1, 2 .. (a lane's terms in slot order, then the group tree); ONE exchange carries all of it.  Same sums, same order, same bits as the two passes.
template <class EM>
__device__ __forceinline__ void pc_chi_spec_build(const EM L, int ne, int n, const Camera& k, double delta, PcShared& sh)
{
    PROF2_T0
    const int gpb = NGROUP / 8;                                                  // (the caller has checked gridDim.x == 8)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, set = wv / gpb, gi = wv - set * gpb;
    const int gg = (int)blockIdx.x * gpb + gi;
    if (set < PC_NPARTS) {
        pc_build_group<PC_SYS2>(L, ne, set, gg, sh.spec[0].P, k, delta, sh);
        if (set == PC_CHI_PART && lane == 0) sh.red[gg][NACC + 1] = sh.red[gg][PC_SYS2 + NACC];      // (a lane of this wave wrote it a moment ago: LDS is in order within a wave)
    } else {
        const int c = set - (PC_NPARTS - 1);
        if (c < n) {
            double a = 0.0;
            for (int i = gg * GROUP + lane; i < ne; i += PC_T) { const LEdge l = L.ld(i); if (!(l.meta & LE_LEVEL)) { Edge e = pc_expand(l); a += edge_rho(e, sh.spec[c].P, k, delta); } }
            a = a + pc_xor_f64<1, true>(a); a = a + pc_xor_f64<2, true>(a); a = a + pc_xor_f64<4, true>(a);
            a = a + pc_xor_f64<8, true>(a); a = a + pc_xor_f64<16, true>(a); a = a + pc_xor_f64<32, true>(a);
            if (lane == 0) sh.red[gg][NACC + 1 + c] = a;
        }
    }
    PROF2(25)
    pc_lane_finish<PC_SPEC + NACC + 1, NACC + 1, true, PC_SB_ROWSCAN>(sh, PC_SPEC + NACC + 1, n, PC_SPEC);      // values 0 .. n-1 (the candidates) and 8 .. 35 (the system)
}
// solve_ldlt of pnp_core.h by the first six lanes of a wave, ONE ROW of L each: the same operations in the same order -- lane i forms
// s = A[i][j] - sum_k (L[i][k] L[j][k]) D[k] for every column j (lane j's s is the pivot d_j: the diagonal's formula is the off-diagonal one with i = j), divides by the
// broadcast pivot, and the forward substitution subtracts column by column, which is each row's ascending-k order; the back substitution's order (x[i] needs x[i+1]
// before x[i+2] ...) is a chain, so it runs uniformly on broadcast values.  15 + 6 divisions and 35 multiply-subtract terms of the scalar form become 7 and 15 wave
// instructions: ~375 instead of ~600 on a wave that is alone on its SIMD (every instruction a ~7-clock step).  Hl / b: LDS (uniform addresses per lane); x: uniform.
__device__ __forceinline__ double pc_lane_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ bool pc_solve_ldlt_wave(const double* Hl, double lambda, const double* b, double (&x)[6])
{
    const int lane = threadIdx.x & 63, i = lane < 6 ? lane : 5;
    double arow[6], Lrow[6], D[6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const int hi = i > c ? i : c, lo = i > c ? c : i;
        const double a = Hl[hi * (hi + 1) / 2 + lo];
        arow[c] = c == i ? a + lambda : a;
        Lrow[c] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double sacc = arow[j];
#pragma unroll
        for (int k = 0; k < j; k++) { const double Ljk = pc_lane_bcast(Lrow[k], j); sacc -= Lrow[k] * Ljk * D[k]; }
        const double d = pc_lane_bcast(sacc, j);
        if (!(d > 0)) return false;
        D[j] = d;
        Lrow[j] = sacc / d;
    }
    double yv = b[i];
#pragma unroll
    for (int k = 0; k < 5; k++) { const double yk = pc_lane_bcast(yv, k); if (i > k) yv -= Lrow[k] * yk; }
    double Di = D[0];
#pragma unroll
    for (int c = 1; c < 6; c++) Di = i == c ? D[c] : Di;
    yv = yv / Di;
    double yu[6], Lu[6][6];
#pragma unroll
    for (int c = 0; c < 6; c++) yu[c] = pc_lane_bcast(yv, c);
#pragma unroll
    for (int k = 1; k < 6; k++)
#pragma unroll
        for (int c = 0; c < k; c++) Lu[k][c] = pc_lane_bcast(Lrow[c], k);
#pragma unroll
    for (int r = 5; r >= 0; r--) { double sacc = yu[r]; for (int k = r + 1; k < 6; k++) sacc -= Lu[k][r] * x[k]; x[r] = sacc; }
    return true;
}
// lm_optimize of pnp_core.h.  The passes over the edges are the whole block's; the 6 x 6 algebra between them (L D L^T, exp map, Levenberg's bookkeeping)
// is WAVE 0's alone -- run by all sixteen waves it cost four times as much, since four waves share a SIMD -- which publishes the next estimate and the loop
// controls through LDS.
template <bool CL, class EM>
__device__ __forceinline__ void pc_optimize(const EM L, double2* err, int ne, int nact, const Camera& k, double delta, int iterations, PcShared& sh)
{
    const Pose& P = sh.P;                                                        // in / out: the estimate lives in LDS (uniform reads; 24 registers saved)
    int any = 0;
    for (int i = threadIdx.x; i < ne; i += PC_T) any |= !(L.meta(i) & LE_LEVEL);
    if (!__syncthreads_or(any)) return;
    const bool w0 = threadIdx.x < 64;
    LmState st; st.lambda = 0; st.nu = 2;                                        // (meaningful in wave 0 only)
    double chi = 0, x[6], gain = 0; int trials = 0; bool solved = false;
    const double* Hl = sh.tot; const double* b = sh.tot + 21;                 // the system stays in LDS over the trials (registers are short here)
    // (SPECB: the first round of an iteration also builds the system at its first candidate -- pc_chi_spec_build; have_sys: sh.tot[0 .. 27] already hold the system at P)
    const bool SPECB = CL && gridDim.x == 8 && PC_FIRST_SPEC <= 9 - PC_NPARTS;
    bool have_sys = false;
    PROF_T0
    for (int it = 0; it < iterations; it++) {
        if (!have_sys) {
            pc_chi_build<CL>(L, err, ne, P, k, delta, sh);
            if (threadIdx.x == 0) { sh.work[0] += 1; sh.work[2] += nact; }
        }
        have_sys = false;
        PROF(1) PROF_CNT(6)
        if (w0) {
            chi = sh.tot[NACC];
            if (it == 0) { double mx = 0; for (int j = 0; j < 6; j++) { const double dg = fabs(Hl[j * (j + 1) / 2 + j]); if (dg > mx) mx = dg; } st.lambda = 1e-5 * mx; st.nu = 2; }
            gain = 0; trials = 0;
        }
        // The trials of an iteration.  The first one runs alone (it is accepted more often than not).  A rejected trial is followed by trials whose damping is
        // known in advance (lm_update's reject branch: lambda <- lambda nu, nu <- 2 nu), so after a rejection wave c solves trial c of the coming streak (up to
        // PC_SPEC, never beyond the tenth trial) and all their estimates go through ONE pass over the edges; wave 0 then walks the results in order with lm_update --
        // the first accepted trial ends the walk (later candidates are dropped), a stop condition ends it too: the same decisions and the same numbers as one
        // trial at a time.  (round 5: a converged optimize spends its time in streaks of seven or more rejections: ~54 solve -> pass -> update rounds per frame became ~36)
        // (the cluster form has idle waves for the algebra and for the pass -- see pc_chi_spec -- so it solves the trials that WOULD follow a rejection together with the
        // first one: PC_FIRST_SPEC candidates; the one-block form pays for every candidate with a walk over its edges and keeps the first trial alone)
        if (threadIdx.x == 0) { sh.spec_lambda = st.lambda; sh.spec_nu = st.nu; sh.spec_n = CL ? PC_FIRST_SPEC : 1; sh.acc0 = 0; }
        __syncthreads();
        bool first = true, built = false;
        for (;;) {
            const int nsp = sh.spec_n;
            const int wv = threadIdx.x >> 6;
            PROF2_T0
            if (wv < nsp) {
                double lam = sh.spec_lambda, nu = sh.spec_nu;
                for (int c = 0; c < wv; c++) { lam *= nu; nu *= 2; }
                for (int q = 0; q < 6; q++) x[q] = 0;
                solved = pc_solve_ldlt_wave(Hl, lam, b, x);
                PROF2(29)
                Pose Pn = sh.P;
                pose_oplus(Pn, x);
                PROF2(30)
                if ((threadIdx.x & 63) == 0) { for (int q = 0; q < 6; q++) sh.spec[wv].x[q] = x[q]; sh.spec[wv].P = Pn; sh.spec[wv].solved = solved ? 1 : 0; }
            }
            __syncthreads();
            PROF2(31)
            PROF(2)
            if constexpr (CL) {
                if (SPECB && first && it + 1 < iterations) {
                    pc_chi_spec_build(L, ne, nsp, k, delta, sh); built = true;
                    if (threadIdx.x == 0) { sh.work[0] += 1; sh.work[2] += nact; sh.work[1] += nsp - 1; sh.work[3] += (long long)(nsp - 1) * nact; }
                } else {
                    pc_chi_spec<CL>(L, ne, nsp, k, delta, sh);
                    if (threadIdx.x == 0) { sh.work[1] += nsp; sh.work[3] += (long long)nsp * nact; }
                }
            } else {
                pc_chi_spec<CL>(L, ne, nsp, k, delta, sh);
                if (threadIdx.x == 0) { sh.work[1] += nsp; sh.work[3] += (long long)nsp * nact; }
            }
            PROF(3) PROF_CNT(7)
            if (w0) {
                bool cont = true, brk = false;
                for (int c = 0; c < nsp && cont; c++) {
                    for (int q = 0; q < 6; q++) x[q] = sh.spec[c].x[q];
                    const double chi_new = sh.tot[NACC + 1 + c];
                    if (lm_update(st, chi, chi_new, sh.spec[c].solved != 0, x, b, gain)) { chi = chi_new; if (threadIdx.x == 0) { sh.P = sh.spec[c].P; if (first && c == 0) sh.acc0 = 1; } }
                    else if (!isfinite(st.lambda)) brk = true;
                    if (!brk) trials++;
                    cont = !brk && gain < 0 && trials < 10;
                }
                if (threadIdx.x == 0) { sh.cont = cont ? 1 : 0; sh.term = (!cont && (trials == 10 || gain == 0)) ? 1 : 0;
                                        sh.spec_lambda = st.lambda; sh.spec_nu = st.nu; sh.spec_n = (10 - trials) < PC_SPEC ? (10 - trials) : PC_SPEC; }
            }
            __syncthreads();
            PROF(4)
            first = false;
            if (!sh.cont) break;
        }
        const int term = sh.term;
        const bool take = built && sh.acc0 != 0;                                 // the first trial was accepted: the system built beside it is the next iteration's
        __syncthreads();                                                         // (sh.cont / sh.term / sh.acc0 are rewritten by the next iteration)
        if (term) break;
        if (take) {
            if (threadIdx.x < NACC + 1) sh.tot[threadIdx.x] = sh.tot[PC_SYS2 + threadIdx.x];
            __syncthreads();
            have_sys = true;
        }
    }
    // "the active edges carry the error at the final estimate": in the cluster form pc_solve recomputes every active edge's error itself (a block has only
    // stored the errors of its own lanes), so the pass -- whose sum nobody reads -- is the one-block form's only
    if constexpr (!CL) {
        pc_chi<CL>(L, err, ne, P, k, delta, sh);
        if (threadIdx.x == 0) { sh.work[1] += 1; sh.work[3] += nact; }
    }
}
// ssm_pnp::solve for the block: img / obj (nc correspondences) in global scratch, T in / out (in LDS; thread 0 writes it); returns the number of set flags
template <bool CL, class EM>
__device__ __forceinline__ int pc_solve(const float* img, const float* obj, int n, const Camera& cam, double* T, uint8_t* inl, const EM L, double2* err, uint8_t* dec, PcShared& sh)
{
    const double delta = (double)(float)sqrt(5.991);
    // edge list: the correspondences with depth, in order
    int ne = 0;
    for (int i0 = 0; i0 < n; i0 += PC_T) {
        const int i = i0 + threadIdx.x;
        const bool has = i < n && !(obj[3 * i] == 0.f && obj[3 * i + 1] == 0.f && obj[3 * i + 2] == 0.f);
        int tot; const int pos = ne + pc_scan(has, sh, tot);
        if (i < n) inl[i] = has ? 1 : 0;
        if (has) { LEdge l; l.meta = (uint32_t)i | LE_ROBUST; l.X[0] = obj[3 * i]; l.X[1] = obj[3 * i + 1]; l.X[2] = obj[3 * i + 2]; l.u = img[2 * i]; l.v = img[2 * i + 1]; L.st(pos, l); err[pos] = make_double2(0.0, 0.0); }
        ne += tot;
    }
    __syncthreads();
    int good = ne, nact = ne;                                                    // nact: edges at level 0 in the coming round
    if (threadIdx.x == 0) { Pose i0; pose_from_iso(T, i0); sh.init = i0; }
    const Pose& P = sh.P;
    for (int it = 0; it < 4; it++) {
        __syncthreads();
        if (threadIdx.x == 0) sh.P = sh.init;
        __syncthreads();
        pc_optimize<CL>(L, err, ne, nact, cam, delta, 10, sh);
        // pnp.cpp:74-93 for all edges at once: the reads of inliers[e->id()] see the flags of before this loop (an earlier edge's writes never land on a
        // later edge's id: ids are unique and a position never exceeds its id); of the writes, a passing edge's inliers[position] = true comes after the
        // failing write of the edge whose id equals that position (position <= id), so: decide, clear, then set
        // (the decisions of a thread's edges -- at most 64: ne <= 65535 -- stay in a register between the three steps; `dec` is no longer written)
        int nout = 0; unsigned long long decm = 0ull;
        for (int i = threadIdx.x, j = 0; i < ne; i += PC_T, j++) {
            LEdge l = L.ld(i);
            Edge e = pc_expand(l);
            // (an ACTIVE edge's stored error is edge_error at this P already -- lm_optimize ends with a chi2 pass over the active edges -- so recomputing it changes
            // nothing in the one-block form; in the cluster form only the block that owns the edge's lane has stored it)
            const bool redo = CL ? (inl[e.id] || !(l.meta & LE_LEVEL)) : (inl[e.id] != 0);
            if (redo) { edge_error(e, P, cam); err[i] = make_double2(e.e0, e.e1); }
            else { const double2 er = err[i]; e.e0 = er.x; e.e1 = er.y; }
            const bool out = edge_chi2(e) > 5.991;
            l.meta = (l.meta & ~LE_LEVEL) | (out ? LE_LEVEL : 0u);
            if (it == 2) l.meta &= ~LE_ROBUST;
            L.set_meta(i, l.meta); decm |= (unsigned long long)(out ? 1 : 0) << j; nout += out;
        }
        __syncthreads();
        for (int i = threadIdx.x, j = 0; i < ne; i += PC_T, j++) if ((decm >> j) & 1ull) inl[L.meta(i) & 0xFFFFu] = 0;
        __syncthreads();
        for (int i = threadIdx.x, j = 0; i < ne; i += PC_T, j++) if (!((decm >> j) & 1ull)) inl[i] = 1;
        __syncthreads();
        // good -= the number of failing edges (block sum of nout)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nout += __shfl_xor(nout, o, 64);
        if ((threadIdx.x & 63) == 0) sh.wcnt[threadIdx.x >> 6] = nout;
        __syncthreads();
        int allout = 0; for (int k = 0; k < NGROUP; k++) allout += sh.wcnt[k];
        __syncthreads();
        good -= allout; nact = ne - allout;
        if (good < 5) break;
    }
    if (threadIdx.x == 0) { const Pose Pf = sh.P; pose_to_iso(Pf, T); }
    int m = 0;
    for (int i = threadIdx.x; i < n; i += PC_T) m += inl[i] != 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o, 64);
    if ((threadIdx.x & 63) == 0) sh.wcnt[threadIdx.x >> 6] = m;
    __syncthreads();
    int mm = 0; for (int k = 0; k < NGROUP; k++) mm += sh.wcnt[k];
    __syncthreads();
    return mm;
}

template <bool CL, int AS>
__global__ void __launch_bounds__(PC_T)
pnp_chain_kernel(PnpChainArgs a)
{
    __shared__ PcShared sh;
    extern __shared__ __align__(16) unsigned char pc_dyn[];
    // the cluster form: G blocks run this SAME program on the same inputs -- state, scratch and results are private to a block (slice blockIdx.x of every array), so
    // the blocks agree by construction; they differ only in which lanes' edges they evaluate in a pass, and trade the partial sums (pc_lane_sum)
    if constexpr (CL) {
        const size_t mc = (size_t)a.R * a.cap, b = blockIdx.x;
        a.state += b; a.img += b * mc * 2; a.obj += b * mc * 3; a.inl += b * mc; a.dec += b * mc; a.err += b * mc;
        a.ledges = reinterpret_cast<LEdge*>(reinterpret_cast<unsigned char*>(a.ledges) + b * mc * sizeof(LEdge)); }
    const bool lead = !CL || blockIdx.x == 0;                                          // its copies of pose_out / info_out are the call's results
    EdgeMem<AS> L; if constexpr (AS == 3) L.p = (typename EdgeMem<AS>::W*)pc_dyn; else L.p = (typename EdgeMem<AS>::W*)a.ledges;
    const int tid = threadIdx.x;
    if (CL && tid == 0) { sh.xmb = a.xchg; sh.xfail = a.xfail; sh.xseq = 0; }
    // the tracker state: in LDS, written by thread 0 (sixteen waves holding five 4 x 4 transforms each in registers spilled most of them)
    if (tid == 0) { for (int k = 0; k < 16; k++) { sh.speed[k] = a.state->speed[k]; sh.last[k] = a.state->last_pose[k]; } for (int k = 0; k < 4; k++) sh.work[k] = 0; }
#ifdef SSM_PNP_PROF
    if (tid == 0) for (int k = 0; k < 32; k++) sh.prof[k] = 0;
#endif
    int nref = a.state->nref, cnt_lost = a.state->cnt_lost;
    // the deque lives in global memory (a.state->ref_idx / ref_pose); every thread tracks nref
    int f = a.f_begin;
    int stopped = a.f_end;
    __syncthreads();
    for (; f < a.f_end; f++) {
        ssm_track_info info; info.state = 1; info.tracked = 0; info.n_matches = -1; info.n_inliers = 0;
        PROF_T0
        if (tid == 0) iso_mul(sh.speed, a.state->ref_pose[nref - 1], sh.Tpred); // currentFrame->setTransform(speed * refFrames.back()->getTransform())
        // ---- the correspondences of every reference frame, in deque order then match order (track.cpp:150-163)
        // ONE walk over the concatenated match lists (round 5; before, one walk per reference frame: five trips of ~540 matches through the dependent global loads and the
        // scan's barriers instead of three of 1024).  The inverses of the references' poses and the lists' offsets sit in sh.rbuf, which only the chi2 passes use.
        int nc = 0;
        double (*invs)[16] = reinterpret_cast<double (*)[16]>(sh.rbuf);
        int* roff = reinterpret_cast<int*>(sh.rbuf + SSM_TRACK_MAXREF * 16);      // [nref + 1] offsets, then [nref] list lengths
        static_assert(SSM_TRACK_MAXREF * 16 + SSM_TRACK_MAXREF + 2 <= PC_RBUF, "the references' inverses and offsets fit the chi2 passes' buffer");
        __syncthreads();
        if (tid < nref) {
            iso_inverse(a.state->ref_pose[tid], invs[tid]);
            const int slot = a.R - (f - a.state->ref_idx[tid]);
            roff[SSM_TRACK_MAXREF + 1 + tid] = max(a.nmatch[(size_t)f * a.R + slot], 0);
        }
        __syncthreads();
        if (tid == 0) { int o = 0; for (int r = 0; r < nref; r++) { roff[r] = o; o += roff[SSM_TRACK_MAXREF + 1 + r]; } roff[nref] = o; }
        __syncthreads();
        const int total = roff[nref];
        for (int k0 = 0; k0 < total; k0 += PC_T) {
            const int idx = k0 + tid;
            bool keep = false; float p0 = 0, p1 = 0, p2 = 0; int ti = 0, r = 0;
            if (idx < total) {
                while (idx >= roff[r + 1]) r++;
                const int k = idx - roff[r];
                const int ridx = a.state->ref_idx[r];                          // local frame index of the reference (negative: a frame of the previous call)
                const int slot = a.R - (f - ridx);
                const ssm_dmatch* m = a.matches + ((size_t)f * a.R + slot) * a.cap;
                const float* rpos = ridx >= 0 ? a.pos3d + (size_t)ridx * a.cap * 3 : a.hist_pos3d + (size_t)(ridx + a.R) * a.cap * 3;
                const ssm_dmatch d = m[k]; const float* p = rpos + (size_t)d.queryIdx * 3; p0 = p[0]; p1 = p[1]; p2 = p[2]; ti = d.trainIdx; keep = !(p0 == 0.f && p1 == 0.f && p2 == 0.f);
            }
            int tot; const int pos = nc + pc_scan(keep, sh, tot);
            if (keep) {
                double v[3]; iso_apply(invs[r], (double)p0, (double)p1, (double)p2, v);
                a.obj[3 * pos] = (float)v[0]; a.obj[3 * pos + 1] = (float)v[1]; a.obj[3 * pos + 2] = (float)v[2];
                const ssm_keypoint kp = a.kps[(size_t)f * a.cap + ti];
                a.img[2 * pos] = kp.x; a.img[2 * pos + 1] = kp.y;
            }
            nc += tot;
        }
        __syncthreads();
        PROF(0)
        info.n_matches = nc;
        bool ok = nc >= 15;
        if (ok) {
            if (tid == 0) iso_mul(sh.speed, sh.last, sh.T);                     // T = speed * lastPose
            __syncthreads();
            info.n_inliers = pc_solve<CL>(a.img, a.obj, nc, a.cam, sh.T, a.inl, L, a.err, a.dec, sh);
            ok = info.n_inliers >= 15;
            PROF(5)
        }
        if (!ok) {
            cnt_lost++;
            if (cnt_lost > a.max_lost) info.state = 2;
            if (tid == 0 && lead) { for (int k = 0; k < 16; k++) a.pose_out[(size_t)f * 16 + k] = sh.Tpred[k]; a.info_out[f] = info; }
            stopped = f + 1;                                                    // the deque now trails behind the match-table window: the host path goes on
            break;
        }
        cnt_lost = 0;
        info.tracked = 1;
        __syncthreads();                                                        // every thread has read the deque of this frame; sh.T is written
        if (tid == 0) {
            double linv[16], sp[16]; iso_inverse(sh.last, linv);
            iso_mul(sh.T, linv, sp);                                            // speed = T * lastPose.inverse()
            for (int k = 0; k < 16; k++) { sh.speed[k] = sp[k]; sh.last[k] = sh.T[k]; }
            if (lead) { for (int k = 0; k < 16; k++) a.pose_out[(size_t)f * 16 + k] = sh.T[k]; a.info_out[f] = info; }
            // refFrames.push_back(currentFrame); pop_front beyond tracker_ref_frames
            if (nref == a.R) { for (int r = 1; r < nref; r++) { a.state->ref_idx[r - 1] = a.state->ref_idx[r]; for (int k = 0; k < 16; k++) a.state->ref_pose[r - 1][k] = a.state->ref_pose[r][k]; } }
            const int slot = nref == a.R ? nref - 1 : nref;
            a.state->ref_idx[slot] = f;
            for (int k = 0; k < 16; k++) a.state->ref_pose[slot][k] = sh.T[k];
        }
        if (nref < a.R) nref++;
        __threadfence_block();
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) {
        for (int k = 0; k < 16; k++) { a.state->speed[k] = sh.speed[k]; a.state->last_pose[k] = sh.last[k]; }
        a.state->nref = nref; a.state->cnt_lost = cnt_lost; a.state->stopped_at = stopped;
        for (int k = 0; k < 4; k++) a.state->work[k] = sh.work[k];
        if (CL && __hip_atomic_load(a.xfail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) a.state->stopped_at = -1;     // an exchange timed out: the host rejects the range
#ifdef SSM_PNP_PROF
        for (int k = 0; k < 32; k++) a.state->prof[k] = sh.prof[k];
#endif
    }
}
// ---- PnPSolver::solvePnP alone (ssm_pnp_solve): the same pc_solve on a caller's correspondence list
template <bool CL, int AS>
__global__ void __launch_bounds__(PC_T)
pnp_solve_kernel(PnpSolveArgs a)
{
    __shared__ PcShared sh;
    extern __shared__ __align__(16) unsigned char pc_dyn[];
    if constexpr (CL) {                                        // block b's private scratch (the blocks run the same program and agree by construction, like the chain's cluster)
        const size_t b = blockIdx.x;
        a.inl += b * a.slice; a.dec += b * a.slice; a.err += b * a.slice;
        a.ledges = reinterpret_cast<LEdge*>(reinterpret_cast<unsigned char*>(a.ledges) + b * a.slice * sizeof(LEdge));
        if (threadIdx.x == 0) { sh.xmb = a.xchg; sh.xfail = a.xfail; sh.xseq = a.seq_base; }
    }
    EdgeMem<AS> L; if constexpr (AS == 3) L.p = (typename EdgeMem<AS>::W*)pc_dyn; else L.p = (typename EdgeMem<AS>::W*)a.ledges;
    if (threadIdx.x < 16) sh.T[threadIdx.x] = a.T[threadIdx.x];
    if (threadIdx.x < 4) sh.work[threadIdx.x] = 0;
    __syncthreads();
    const int m = pc_solve<CL>(a.img, a.obj, a.n, a.cam, sh.T, a.inl, L, a.err, a.dec, sh);
    __syncthreads();
    if (!CL || blockIdx.x == 0) {
        if (threadIdx.x < 16) a.T[threadIdx.x + 16] = sh.T[threadIdx.x];          // T out sits behind T in (a block of the cluster may still be reading T in when block 0 is done)
        if (threadIdx.x == 0) *a.n_inliers = m;
    }
}
static hipError_t pc_dyn_size(size_t nedges, int* in_lds, size_t* dyn, const void* fn)
{
    if (nedges > 65535) return hipErrorInvalidValue;                             // ids are 16-bit in LEdge::meta
    const size_t need = nedges * sizeof(LEdge);
    *in_lds = need + sizeof(PcShared) + 1024 <= 160 * 1024 ? 1 : 0;              // the edge list in LDS when it fits beside the static part (160 KB per CU)
    *dyn = *in_lds ? need : 0;
    // the attribute belongs to the current device's copy of the function (a process may hold contexts on several devices): set once per device and kernel, to
    // the largest size any list in LDS can have -- not per launch: the call can serialise against kernels in flight, and trackers on their own streams launch
    // while other chains run
    if (*dyn <= 48 * 1024) return hipSuccess;
    static std::mutex mu; static std::set<std::pair<int, const void*>> done;
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, fn})) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)sizeof(PcShared) - 1024);
    if (e == hipSuccess) done.insert({dev, fn});
    return e;
}
hipError_t k_pnp_solve(const PnpSolveArgs& a_in, hipStream_t s)
{
    PnpSolveArgs a = a_in; size_t dyn;
    if (a.blocks > 1) {                                        // the cluster form (the caller owns the ring's zeroing and the time-out word: PnpSolveArgs::seq_base)
        hipError_t e = pc_dyn_size((size_t)(a.n > 0 ? a.n : 1), &a.edges_in_lds, &dyn, reinterpret_cast<const void*>(pnp_solve_kernel<true, 3>));
        if (e != hipSuccess) return e;
        if (a.edges_in_lds) pnp_solve_kernel<true, 3><<<a.blocks, PC_T, dyn, s>>>(a); else pnp_solve_kernel<true, 1><<<a.blocks, PC_T, 0, s>>>(a);
        return hipGetLastError();
    }
    hipError_t e = pc_dyn_size((size_t)(a.n > 0 ? a.n : 1), &a.edges_in_lds, &dyn, reinterpret_cast<const void*>(pnp_solve_kernel<false, 3>));
    if (e != hipSuccess) return e;
    if (a.edges_in_lds) pnp_solve_kernel<false, 3><<<1, PC_T, dyn, s>>>(a); else pnp_solve_kernel<false, 1><<<1, PC_T, 0, s>>>(a);
    return hipGetLastError();
}
size_t k_pnp_edge_bytes(void) { return sizeof(LEdge); }
size_t k_pnp_xchg_bytes(void) { return (size_t)4 * NGROUP * 64 * 2 * 8 + 64; }      // four ring slots x 16 groups x 64 values x two granules, + the time-out word
hipError_t k_pnp_chain(const PnpChainArgs& a_in, hipStream_t s)
{
    PnpChainArgs a = a_in; size_t dyn;
    const int G = a.blocks > 0 ? a.blocks : 1;
    hipError_t e = pc_dyn_size((size_t)a.R * a.cap, &a.edges_in_lds, &dyn, G > 1 ? reinterpret_cast<const void*>(pnp_chain_kernel<true, 3>) : reinterpret_cast<const void*>(pnp_chain_kernel<false, 3>));
    if (e != hipSuccess) return e;
    if (G > 1) {                                               // the exchange ring and the time-out word: zero before every launch (tags count passes within a launch)
        e = hipMemsetAsync(a.xchg, 0, k_pnp_xchg_bytes(), s);
        if (e != hipSuccess) return e;
        const char* tv = getenv("SSM_PNP_TEST_TIMEOUT");                                        // tests: the time-out word set from the start -> the host's retry path
        if (tv && atoi(tv) != 0) { e = hipMemsetAsync(a.xfail, 1, 4, s); if (e != hipSuccess) return e; }
        if (a.edges_in_lds) pnp_chain_kernel<true, 3><<<G, PC_T, dyn, s>>>(a); else pnp_chain_kernel<true, 1><<<G, PC_T, 0, s>>>(a);
    } else { if (a.edges_in_lds) pnp_chain_kernel<false, 3><<<1, PC_T, dyn, s>>>(a); else pnp_chain_kernel<false, 1><<<1, PC_T, 0, s>>>(a); }
    return hipGetLastError();
}
