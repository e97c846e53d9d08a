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

// ------------------------------------------------------------------ pre-filter + BT intervals
// per image one plane of three words per pixel: .x = the value, .y = min(value, half-sample neighbours), .z = max(...), each word (clipped x-Sobel | raw
// intensity << 16).  The two cost terms of a pixel pair are then the two 16-bit halves of the same packed subtract / max / min instructions,
// and a pixel's three words are ONE 16-byte LDS read in the cost kernel.
__global__ void __launch_bounds__(256)
sgbm_prefilter(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right, int w, int h, int ftzero, uint3* __restrict__ planes_all)
{
    // blockIdx.z = frame * 2 + side; a frame's 2 planes: left, right
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const size_t np = (size_t)w * h;
    const int f = blockIdx.z >> 1, side = blockIdx.z & 1;
    const uint8_t* img = (side ? right : left) + (size_t)f * np;
    uint3* planes = planes_all + ((size_t)f * 2 + (size_t)side) * np;
    const uint8_t* row = img + (size_t)y * w;
    const int n1 = y > 0 ? -w : 0, s1 = y < h - 1 ? w : 0;
    auto grad = [&](int xx) -> int {            // prow[x]: tab[...] for 1 <= x <= w-2, tab[0] = ftzero at the two border columns
        if (xx < 1 || xx > w - 2) return ftzero;
        const int g = (row[xx + 1] - row[xx - 1]) * 2 + row[xx + n1 + 1] - row[xx + n1 - 1] + row[xx + s1 + 1] - row[xx + s1 - 1];
        return min(max(g, -ftzero), ftzero) + ftzero;
    };
    auto raw = [&](int xx) -> int { return (xx < 1 || xx > w - 2) ? ftzero : (int)row[xx]; };     // the border columns of the raw plane hold tab[0] too
    uint32_t pv = 0, pmin = 0, pmax = 0;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int v = c ? raw(x) : grad(x);
        const int vl = x > 0 ? (v + (c ? raw(x - 1) : grad(x - 1))) / 2 : v, vr = x < w - 1 ? (v + (c ? raw(x + 1) : grad(x + 1))) / 2 : v;
        pv |= (uint32_t)v << (16 * c); pmin |= (uint32_t)min(min(vl, vr), v) << (16 * c); pmax |= (uint32_t)max(max(vl, vr), v) << (16 * c);
    }
    planes[(size_t)y * w + x] = make_uint3(pv, pmin, pmax);      // 12 bytes per pixel in memory; the cost kernel widens them to the 16-byte LDS words
}
// The matching cost C(y, x, d) = P2 + sum over the SAD window of the Birchfield-Tomasi pixel cost, in ONE streaming kernel (it was three volume kernels:
// pixel cost -> u8 volume, horizontal box -> u16 volume, vertical box -> C; 8 bytes of HBM traffic per volume entry instead of 2).
// A block owns a strip of TX cost-volume columns x all D disparities of one frame and walks the image rows top to bottom, like OpenCV's row loop:
//   stage   the pixel words of row r that the strip needs (left: TX + 2 SW2 columns, right: D - 1 more) into LDS -- fetched into registers one row
//           ahead, so the loads of row r + 1 fly during the arithmetic of row r;
//   pixel   thread (d, chunk) computes the BT cost of its disparity for every column of the strip + apron (the left pixel is an LDS broadcast, the right
//           pixels of consecutive d are consecutive 16-byte words; both cost terms in one set of packed 16-bit operations) -> u8 row in LDS;
//   hbox    the same thread slides the SW-wide window over its run of columns (replicate borders at the ends of the cost volume) -> hs(r, x, d);
//   vbox    a ring of the last SW hs rows in LDS (each (x, d) is read and written by its owner only: no barrier) gives the running vertical sum:
//           C(y) = C(y - 1) + hs(y + SH2) - hs(max(y - SH2 - 1, 0)), C(0) = P2 + (SH2 + 1) hs(0) + hs(1) + ... + hs(SH2) (replicated top border).
// The two OpenCV 2.4 quirks of the contract: cost-volume column 0 keeps row 0's value, rows below h - 1 - SH2 repeat the last full window.
// Two barriers per row.  Dynamic LDS: ring u16 [SW][TX][D] | pixrow u8 [TX + 2 SW2][D] | lrow uint4 [TX + 2 SW2] | rrow uint4 [TX + 2 SW2 + D - 1].
// CD / CSW2 / CTX: compile-time D, SW2, TX of the instantiation for stereo.cpp's configuration (80 disparities, SAD 11); 0 = run-time values.
#define SGC_THREADS 512
typedef unsigned short us2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (us2v)(__builtin_bit_cast(us2v, a) + __builtin_bit_cast(us2v, b))); }
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (us2v)(__builtin_bit_cast(us2v, a) - __builtin_bit_cast(us2v, b))); }
__device__ __forceinline__ uint32_t pk_mul16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (us2v)(__builtin_bit_cast(us2v, a) * __builtin_bit_cast(us2v, b))); }
__device__ __forceinline__ uint32_t pk_min16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2v, a), __builtin_bit_cast(us2v, b))); }
template <bool EDGE, int CD, int CSW2, int CTX, int MAXCW>
__device__ __forceinline__ void sgbm_cost_strip(const uint3* __restrict__ planes_all, int w, int h, int minD, int Drt, int minX1, int w1, int SW2rt, int P2, int TXrt, uint16_t* __restrict__ C_all)
{
    extern __shared__ __align__(16) uint8_t sg_smem[];
    const int D = CD ? CD : Drt, SW2 = CD ? CSW2 : SW2rt, TX = CD ? CTX : TXrt;
    const int f = blockIdx.y, xs = blockIdx.x * TX, tid = threadIdx.x;
    const int SW = 2 * SW2 + 1, AW = TX + 2 * SW2, RW = AW + D - 1;
    uint16_t* ring = reinterpret_cast<uint16_t*>(sg_smem);
    const size_t pixb = ((size_t)AW * D + 15) & ~(size_t)15;
    uint8_t* pixrow2 = sg_smem + (size_t)SW * TX * D * 2;                // two pixel-cost rows and two staged pixel rows (row parity): ONE barrier per row
    uint4* lrow2 = reinterpret_cast<uint4*>(pixrow2 + 2 * pixb);
    const size_t np = (size_t)w * h;
    // (d, chunk) decomposition of the block
    const int nchunk = SGC_THREADS / D, d = tid % D, chunk = tid / D;
    const bool active = chunk < nchunk;
    const int cw = (TX + nchunk - 1) / nchunk;                  // <= MAXCW by the launcher's choice of TX
    const int cx0 = chunk * cw, cx1 = min(min(cx0 + cw, TX), w1 - xs);      // this thread's strip columns [cx0, cx1)
    const int ncol = active ? max(cx1 - cx0, 0) : 0;
    // staging: thread k < AW + RW fetches one pixel word of the left / right row
    const uint3* src = nullptr; uint4 pre = make_uint4(0, 0, 0, 0);
    auto wide = [](const uint3 t) { return make_uint4(t.x, t.y, t.z, 0u); };
    if (tid < AW + RW) {
        const bool isl = tid < AW;
        const int xi = min(max(isl ? xs - SW2 + minX1 + tid : xs - SW2 + minX1 - minD - (D - 1) + (tid - AW), 0), w - 1);   // columns outside the image are never used by a valid cost
        src = planes_all + ((size_t)f * 2 + (isl ? 0 : 1)) * np + xi;
    }
    int Cacc[MAXCW]; uint32_t hs0[MAXCW];
#pragma unroll
    for (int k = 0; k < MAXCW; k++) { Cacc[k] = P2; hs0[k] = 0; }
    const int SH2 = SW2, ylast = h - 1 - SH2;
    uint16_t* Cp = C_all + ((size_t)f * w1 * h + xs + cx0) * D + d;           // C(0, xs + cx0, d); + y * w1 * D per row, + D per column
    const size_t crow = (size_t)w1 * D;
    const int lo = SW2 - xs, hi = w1 - 1 - xs + SW2;            // pixrow index of cost-volume columns 0 and w1 - 1 (replicate beyond them; only in EDGE strips)
    // one image row: pixel costs of row r (staged by the previous trip), stage row r + 1, ONE barrier, horizontal sums -> sum[k] of this thread's columns
    // (PHASE 0: r = 0, 1: 1 .. SH2, 2: beyond).  Everything a trip writes for others lives in the buffers of ITS row parity: the next trip's pixel pass
    // (other parity) may start while slower waves still sum this row, and a buffer is rewritten two trips later, behind the barrier in between.
    auto row = [&](int r, auto phase) {
        constexpr int PHASE = decltype(phase)::value;
        uint8_t* pixrow = pixrow2 + (size_t)(r & 1) * pixb;
        const uint4* lrow = lrow2 + (size_t)(r & 1) * (AW + RW); const uint4* rrow = lrow + AW;
        const uint8_t* pxd = pixrow + d;
        auto px = [&](int x) -> int { const int i = x + SW2; return pxd[(EDGE ? min(max(i, lo), hi) : i) * D]; };       // strip column x (may be negative: apron)
        if (src && r + 1 < h) { lrow2[(size_t)((r + 1) & 1) * (AW + RW) + tid] = pre; if (r + 2 < h) pre = wide(src[(size_t)(r + 2) * w]); }      // row r + 1 for the next trip; row r + 2 on its way
        {
            // pixel costs, one thread per (strip column i, eight consecutive disparities): the left pixel's word is read ONCE for the eight, the right pixels are
            // eight consecutive words, and the eight cost bytes leave as one 8-byte write (round 5: the (d, column chunk) layout read a left and a right word
            // and wrote one byte per cost -- 107 KB of LDS traffic per row and block against 60 KB now; the kernel's LDS pipe was busy 0.64 of the time)
            const int NO = D >> 3;
            for (int it = tid; it < AW * NO; it += SGC_THREADS) {
                const int o = it / AW, i = it - o * AW;
                const uint4 L = lrow[i];
                const us2v u = __builtin_bit_cast(us2v, L.x), u0 = __builtin_bit_cast(us2v, L.y), u1 = __builtin_bit_cast(us2v, L.z);
                const uint4* rp = rrow + (i + (D - 1) - 8 * o);      // right pixel of disparity 8 o + dd: image column of i minus the disparity
                uint32_t pk[2] = {0u, 0u};
#pragma unroll
                for (int dd = 0; dd < 8; dd++) {
                    const uint4 R = rp[-dd];
                    const us2v v = __builtin_bit_cast(us2v, R.x), v0 = __builtin_bit_cast(us2v, R.y), v1 = __builtin_bit_cast(us2v, R.z);
                    // (saturating differences: see sgbm_cost_strip_reg)
                    const us2v c0 = __builtin_elementwise_max(__builtin_elementwise_sub_sat(u, v1), __builtin_elementwise_sub_sat(v0, u));
                    const us2v c1 = __builtin_elementwise_max(__builtin_elementwise_sub_sat(v, u1), __builtin_elementwise_sub_sat(u0, v));
                    const us2v m = __builtin_elementwise_min(c0, c1);
                    pk[dd >> 2] |= (uint32_t)(uint8_t)((int)m.x + ((int)m.y >> 2)) << (8 * (dd & 3));
                }
                *reinterpret_cast<uint2*>(pixrow + (size_t)i * D + 8 * o) = make_uint2(pk[0], pk[1]);
            }
        }
        __syncthreads();
        if (ncol > 0) {
            int sum = 0;
            for (int j = -SW2; j <= SW2; j++) sum += px(cx0 + j);
            uint16_t* rg = ring + ((size_t)(r % SW) * TX + cx0) * D + d;
            uint16_t* Cr = Cp + (ptrdiff_t)(r - SH2) * (ptrdiff_t)crow;
#pragma unroll
            for (int k = 0; k < MAXCW; k++) {
                if (k < ncol) {
                    if (k > 0) sum += px(cx0 + k + SW2) - px(cx0 + k - SW2 - 1);
                    if (PHASE == 0) { hs0[k] = (uint32_t)sum; Cacc[k] += (SH2 + 1) * sum; rg[k * D] = (uint16_t)sum; }
                    else if (PHASE == 1) { Cacc[k] += sum; rg[k * D] = (uint16_t)sum; }
                    else {
                        const int old = r - SW >= 0 ? (int)rg[k * D] : (int)hs0[k];      // hs(r - SW, x, d): the slot this row overwrites
                        rg[k * D] = (uint16_t)sum;
                        const int delta = sum - old;
                        Cacc[k] += (EDGE && xs + cx0 + k == 0) ? 0 : delta;              // cost-volume column 0 keeps C(0)
                    }
                    if (PHASE == 2 || r == SH2) Cr[k * D] = (uint16_t)Cacc[k];
                }
            }
        }
    };
    if (src) { lrow2[tid] = wide(src[0]); if (h > 1) pre = wide(src[(size_t)w]); }      // row 0 staged, row 1 in registers
    __syncthreads();
    row(0, std::integral_constant<int, 0>());
    for (int r = 1; r <= SH2; r++) row(r, std::integral_constant<int, 1>());
    for (int r = SH2 + 1; r < h; r++) row(r, std::integral_constant<int, 2>());
    // rows below h - 1 - SH2 repeat the last full window
    if (ncol > 0) {
        for (int y = ylast + 1; y < h; y++)
#pragma unroll
            for (int k = 0; k < MAXCW; k++) if (k < ncol) Cp[(size_t)y * crow + k * D] = (uint16_t)Cacc[k];
    }
}
// sgbm_prefilter's record of pixel (x, y) computed from the raw image (sgbm_cost_strip_reg: the records are made where they are staged, the 12-byte-per-pixel
// planes are neither written nor read).  p[j][i] = img(row y - 1 + j clamped into the image, column x - 2 + i); columns outside the image may hold anything
// (every use is guarded by the same border tests as sgbm_prefilter's).
__device__ __forceinline__ uint4 sg_prefilter_record(const uint2 (&wr)[3], int x, int w, int ftzero)      // wr[j]: .x = the row's bytes at columns x - 2 .. x + 1, .y = column x + 2
{
    auto px = [&](int j, int i) -> int { return i < 4 ? (int)((wr[j].x >> (8 * i)) & 0xFFu) : (int)(wr[j].y & 0xFFu); };
    auto grad = [&](int i) -> int {             // i = 1, 2, 3: columns x - 1, x, x + 1
        const int xx = x - 2 + i;
        if (xx < 1 || xx > w - 2) return ftzero;
        const int g = (px(1, i + 1) - px(1, i - 1)) * 2 + px(0, i + 1) - px(0, i - 1) + px(2, i + 1) - px(2, i - 1);
        return min(max(g, -ftzero), ftzero) + ftzero;
    };
    auto raw = [&](int i) -> int { const int xx = x - 2 + i; return (xx < 1 || xx > w - 2) ? ftzero : px(1, i); };
    uint32_t pv = 0, pmin = 0, pmax = 0;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int v = c ? raw(2) : grad(2);
        const int vl = x > 0 ? (v + (c ? raw(1) : grad(1))) / 2 : v, vr = x < w - 1 ? (v + (c ? raw(3) : grad(3))) / 2 : v;
        pv |= (uint32_t)v << (16 * c); pmin |= (uint32_t)min(min(vl, vr), v) << (16 * c); pmax |= (uint32_t)max(max(vl, vr), v) << (16 * c);
    }
    return make_uint4(pv, pmin, pmax, 0u);
}
// The same strip with the ring of the last SW horizontal sums in REGISTERS (round 5; stereo.cpp's configuration only: compile-time D, SW2, TX, every C below 2^16).
// The LDS ring (SW x TX x D x 2 bytes = 56 KB of the block's 68) held a block to one per CU, i.e. eight waves = two per SIMD for a kernel that alternates between an
// LDS-heavy and a VALU-heavy phase with a barrier per row; without it a block needs 13 KB and two (or three) blocks share a CU, one's barrier wait under the other's
// work.  A thread's MAXCW columns travel as MAXCW / 2 packed u16 pairs (sum, ring entry, running C: one v_pk_add_u16 / v_pk_sub_u16 per pair), the ring slot
// of a row is a compile-time index because the row loop is unrolled SW rows at a time.  Same arithmetic, same bits as sgbm_cost_strip.
template <bool EDGE, int CD, int CSW2, int CTX, int MAXCW>
__device__ __forceinline__ void sgbm_cost_strip_reg(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right, int ftzero, int w, int h, int minD, int minX1, int w1, int P2, uint16_t* __restrict__ C_all)
{
    static_assert(CD > 0 && (MAXCW & 1) == 0, "compile-time geometry, an even number of columns per thread");
    extern __shared__ __align__(16) uint8_t sg_smem[];
    constexpr int D = CD, SW2 = CSW2, TX = CTX, SW = 2 * SW2 + 1, AW = TX + 2 * SW2, RW = AW + D - 1, NPAIR = MAXCW / 2, SH2 = SW2;
    const int f = blockIdx.y, xs = blockIdx.x * TX, tid = threadIdx.x;
    constexpr size_t pixb = (((size_t)(AW + MAXCW) * D) + 15) & ~(size_t)15;      // MAXCW columns of slack: a thread whose run ends before MAXCW reads (and drops) what lies behind the apron
    uint8_t* pixrow2 = sg_smem;
    uint4* lrow2 = reinterpret_cast<uint4*>(pixrow2 + 2 * pixb);
    const size_t np = (size_t)w * h;
    constexpr int nchunk = SGC_THREADS / D, cw = (TX + nchunk - 1) / nchunk;
    static_assert(cw <= MAXCW, "columns per thread");
    const int d = tid % D, chunk = tid / D;
    const bool active = chunk < nchunk;
    const int cx0 = chunk * cw, cx1 = min(min(cx0 + cw, TX), w1 - xs);
    const int ncol = active ? max(cx1 - cx0, 0) : 0;
    // staging: thread k < AW + RW owns one pixel column of the left / right image and makes its pre-filter record (value | min | max of the clipped x-Sobel and of the raw
    // intensity, sgbm_prefilter's arithmetic) for one row per trip from a rolling 3 x 5 window of raw pixels: one unaligned 8-byte load per row
    const uint8_t* src = nullptr; uint4 pre = make_uint4(0, 0, 0, 0);
    int xi = 0, wofs = 0; uint2 win[3]; unsigned long long nxt = 0ull;      // (one 64-bit value: as a uint2 the compiler copied the freshly loaded pair with its halves swapped at the branch's join -- and waited for the load there)
    auto row_bytes = [&](int y) -> unsigned long long {             // the 8 bytes at column wstart of image row y (clamped)
        const uint8_t* q = src + (size_t)min(max(y, 0), h - 1) * w;
        unsigned long long v; __builtin_memcpy(&v, q, 8); return v;
    };
    auto unpack = [&](const unsigned long long b, uint2& o) {      // window position i = byte wofs + i of the 8 loaded (wofs -2 .. 5: positions outside the 8 bytes are columns outside the image)
        const unsigned long long t = wofs >= 0 ? b >> (8 * wofs) : b << (8 * -wofs);
        o.x = (uint32_t)t; o.y = (uint32_t)(t >> 32);
    };
    if (tid < AW + RW) {
        const bool isl = tid < AW;
        xi = min(max(isl ? xs - SW2 + minX1 + tid : xs - SW2 + minX1 - minD - (D - 1) + (tid - AW), 0), w - 1);
        const int wstart = min(max(xi - 2, 0), w - 8);            // an 8-byte window inside the row that holds columns xi - 2 .. xi + 2 where they exist
        wofs = xi - 2 - wstart;                                   // -2 .. 5: below 0 / above 3 at the image's first / last columns, where the window reaches outside
        src = (isl ? left : right) + (size_t)f * np + wstart;
    }
    // (columns xi - 2 + i that fall outside the image read a neighbouring in-row byte instead: sg_prefilter_record ignores them)
    auto advance = [&](int ynew) {                      // the window moves one row down; row ynew (clamped) arrives from `nxt`, the load of the row behind it is issued
        win[0] = win[1]; win[1] = win[2];
        unpack(nxt, win[2]);
        __builtin_amdgcn_sched_barrier(0);                // (the next load stays BEHIND the use of the previous one: hoisted above it, the use's vmcnt(0) waits for the new load)
        nxt = row_bytes(ynew + 1);
    };
    uint32_t ring[SW][NPAIR], Cacc[NPAIR], hs0[NPAIR];
#pragma unroll
    for (int q = 0; q < NPAIR; q++) { Cacc[q] = (uint32_t)P2 * 0x00010001u; hs0[q] = 0; }
    const int ylast = h - 1 - SH2;
    uint16_t* Cp = C_all + ((size_t)f * w1 * h + xs + cx0) * D + d;
    const size_t crow = (size_t)w1 * D;
    const int lo = SW2 - xs, hi = w1 - 1 - xs + SW2;
    const uint32_t col0_mask = (EDGE && xs + cx0 == 0) ? 0xFFFF0000u : 0xFFFFFFFFu;      // cost-volume column 0 keeps C(0): the low half of this thread's first pair
    auto store_row = [&](int y) {
        uint16_t* Cr = Cp + (size_t)y * crow;
#pragma unroll
        for (int k = 0; k < MAXCW; k++) if (k < ncol) Cr[k * D] = (uint16_t)(Cacc[k >> 1] >> (16 * (k & 1)));
    };
    auto row = [&](int r, auto phase, auto slot_c) {
        constexpr int PHASE = decltype(phase)::value, SLOT = decltype(slot_c)::value;
        uint8_t* pixrow = pixrow2 + (size_t)(r & 1) * pixb;
        const uint4* lrow = lrow2 + (size_t)(r & 1) * (AW + RW); const uint4* rrow = lrow + AW;
        const uint8_t* pxd = pixrow + d;
        auto px = [&](int x) -> int { const int i = x + SW2; return pxd[(EDGE ? min(max(i, lo), hi) : i) * D]; };
        if (src && r + 1 < h) lrow2[(size_t)((r + 1) & 1) * (AW + RW) + tid] = pre;      // the record of row r + 1, made during the previous row
        {
            constexpr int NO = D >> 3;
            for (int it = tid; it < AW * NO; it += SGC_THREADS) {
                const int o = it / AW, i = it - o * AW;
                const uint4 L = lrow[i];
                const us2v u = __builtin_bit_cast(us2v, L.x), u0 = __builtin_bit_cast(us2v, L.y), u1 = __builtin_bit_cast(us2v, L.z);
                const uint4* rp = rrow + (i + (D - 1) - 8 * o);
                uint32_t pk[2] = {0u, 0u};
#pragma unroll
                for (int dd = 0; dd < 8; dd++) {
                    const uint4 R = rp[-dd];
                    const us2v v = __builtin_bit_cast(us2v, R.x), v0 = __builtin_bit_cast(us2v, R.y), v1 = __builtin_bit_cast(us2v, R.z);
                    // max(0, a - b) of values below 2^15 is the unsigned saturating difference (v_pk_sub_u16 clamp), and max(0, p, q) = max(max(0, p), max(0, q)):
                    // seven packed instructions per cost instead of nine (round 6)
                    const us2v c0 = __builtin_elementwise_max(__builtin_elementwise_sub_sat(u, v1), __builtin_elementwise_sub_sat(v0, u));
                    const us2v c1 = __builtin_elementwise_max(__builtin_elementwise_sub_sat(v, u1), __builtin_elementwise_sub_sat(u0, v));
                    const us2v m = __builtin_elementwise_min(c0, c1);
                    pk[dd >> 2] |= (uint32_t)(uint8_t)((int)m.x + ((int)m.y >> 2)) << (8 * (dd & 3));
                }
                *reinterpret_cast<uint2*>(pixrow + (size_t)i * D + 8 * o) = make_uint2(pk[0], pk[1]);
            }
        }
        __syncthreads();
        // The staging threads move their window BEHIND the barrier (round 6): using the row loaded a row ago is a vmcnt(0) (the stores of a row are conditional,
        // no count of younger operations is guaranteed), i.e. it also waits for the acknowledgement of this wave's cost stores of the previous row.  At the top of
        // the row those had just been issued and the other waves sat at the barrier until the staging waves came; here the stores are half a row old.
        if (src && r + 2 < h) { advance(r + 3); pre = sg_prefilter_record(win, xi, w, ftzero); }      // the window now holds rows r + 1 .. r + 3: the record of row r + 2
        if (ncol > 0) {
            int sum = 0;
#pragma unroll
            for (int j = -SW2; j <= SW2; j++) sum += px(cx0 + j);
            uint32_t sp[NPAIR];
#pragma unroll
            for (int k = 0; k < MAXCW; k++) {
                if (k > 0) sum += px(cx0 + k + SW2) - px(cx0 + k - SW2 - 1);
                if (k & 1) sp[k >> 1] |= (uint32_t)sum << 16; else sp[k >> 1] = (uint32_t)sum & 0xFFFFu;
            }
#pragma unroll
            for (int q = 0; q < NPAIR; q++) {
                if (PHASE == 0) { hs0[q] = sp[q]; Cacc[q] = pk_add16(Cacc[q], pk_mul16(sp[q], (uint32_t)(SH2 + 1) * 0x00010001u)); ring[SLOT][q] = sp[q]; }
                else if (PHASE == 1) { Cacc[q] = pk_add16(Cacc[q], sp[q]); ring[SLOT][q] = sp[q]; }
                else {
                    const uint32_t old = PHASE == 2 ? hs0[q] : ring[SLOT][q];      // hs(r - SW, x, d): the slot this row overwrites (PHASE 2: rows SH2 + 1 .. SW - 1, above the image: row 0's)
                    ring[SLOT][q] = sp[q];
                    uint32_t delta = pk_sub16(sp[q], old);
                    if (q == 0) delta &= col0_mask;
                    Cacc[q] = pk_add16(Cacc[q], delta);
                }
            }
            if (PHASE >= 2 || r == SH2) store_row(r - SH2);
        }
    };
    if (src) {
        // window = rows -1 (= 0), 0, 1 -> the record of row 0; then rows 0, 1, 2 -> row 1's, kept in `pre`
        const unsigned long long t0 = row_bytes(0), t1 = row_bytes(1);
        unpack(t0, win[0]); unpack(t0, win[1]); unpack(t1, win[2]);
        nxt = row_bytes(2);
        lrow2[tid] = sg_prefilter_record(win, xi, w, ftzero);
        if (h > 1) { advance(2); pre = sg_prefilter_record(win, xi, w, ftzero); }
    }
    __syncthreads();
    // rows 0 .. SW - 1 (ring slots 0 .. SW - 1): row 0, the rows whose window still reaches above the image, the first rows that drop row 0's replicas
    auto first_rows = [&](auto self, auto rc) -> void {
        constexpr int R = decltype(rc)::value;
        if constexpr (R < SW) {
            if (R < h) {
                row(R, std::integral_constant<int, (R == 0 ? 0 : R <= SH2 ? 1 : 2)>(), std::integral_constant<int, R>());
                self(self, std::integral_constant<int, R + 1>());
            }
        }
    };
    first_rows(first_rows, std::integral_constant<int, 0>());
    auto block_rows = [&](auto self, int base, auto sc) -> void {
        constexpr int S = decltype(sc)::value;
        if constexpr (S < SW) {
            if (base + S < h) {
                row(base + S, std::integral_constant<int, 3>(), std::integral_constant<int, S>());
                self(self, base, std::integral_constant<int, S + 1>());
            }
        }
    };
    for (int base = SW; base < h; base += SW) block_rows(block_rows, base, std::integral_constant<int, 0>());
    if (ncol > 0) for (int y = ylast + 1; y < h; y++) store_row(y);          // rows below h - 1 - SH2 repeat the last full window
}
// the first strip and the last `tail` strips replicate the cost-volume border columns (and strip 0 holds the frozen column 0): they run the EDGE
// instantiation; the strips between them read their apron without clamps.  One launch for all strips (block-uniform branch).
template <int CD, int CSW2, int CTX, int MAXCW>
__global__ void __launch_bounds__(SGC_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
sgbm_cost_kernel(const uint3* __restrict__ planes_all, int w, int h, int minD, int D, int minX1, int w1, int SW2, int P2, int TX, int tail, uint16_t* __restrict__ C_all)
{
    if (blockIdx.x == 0 || (int)blockIdx.x >= (int)gridDim.x - tail) sgbm_cost_strip<true, CD, CSW2, CTX, MAXCW>(planes_all, w, h, minD, D, minX1, w1, SW2, P2, TX, C_all);
    else sgbm_cost_strip<false, CD, CSW2, CTX, MAXCW>(planes_all, w, h, minD, D, minX1, w1, SW2, P2, TX, C_all);
}
template <int CD, int CSW2, int CTX, int MAXCW>
__global__ void __launch_bounds__(SGC_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
sgbm_cost_reg_kernel(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right, int ftzero, int w, int h, int minD, int minX1, int w1, int P2, int tail, uint16_t* __restrict__ C_all)
{
    if (blockIdx.x == 0 || (int)blockIdx.x >= (int)gridDim.x - tail) sgbm_cost_strip_reg<true, CD, CSW2, CTX, MAXCW>(left, right, ftzero, w, h, minD, minX1, w1, P2, C_all);
    else sgbm_cost_strip_reg<false, CD, CSW2, CTX, MAXCW>(left, right, ftzero, w, h, minD, minX1, w1, P2, C_all);
}

// ------------------------------------------------------------------ one aggregation step on a 16-lane row (K disparities per lane)
template <int CTRL>
__device__ __forceinline__ int sg_dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, 0xF, false); }
// a lane permutation in which every lane has a source (rotations, quad_perm, mirrors): with bound_ctrl and no `old` the compiler folds the move into the
// instruction that uses it (v_min_i32_dpp: ONE instruction per reduction stage; with `old = v` it was v_mov + v_mov_dpp + v_min -- round 6)
template <int CTRL>
__device__ __forceinline__ int sg_dpp_all(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ int sg_rowmin(int v)
{
    v = min(v, sg_dpp_all<0x128>(v));     // row_ror:8
    v = min(v, sg_dpp_all<0x124>(v));     // row_ror:4
    v = min(v, sg_dpp_all<0x122>(v));     // row_ror:2
    v = min(v, sg_dpp_all<0x121>(v));     // row_ror:1
    return v;
}
template <int K>
__device__ __forceinline__ void sg_step(int (&L)[K], int& minPrev, const int (&Cp)[K], int P1, int P2)
{
    const int left = sg_dpp<0x111>(SG_MAXC, L[K - 1]);        // row_shr:1 -- L(d-1) of the lane's first disparity; d = -1 is MAX_COST
    const int right = sg_dpp<0x101>(SG_MAXC, L[0]);           // row_shl:1 -- L(d+1) of the lane's last disparity;  d = D  is MAX_COST
    const int delta = minPrev + P2;
    int Ln[K], m = INT_MAX;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int lm = k > 0 ? L[k - 1] : left, lp = k < K - 1 ? L[k + 1] : right;
        Ln[k] = Cp[k] + min(L[k], min(lm + P1, min(lp + P1, delta))) - delta;
        m = min(m, Ln[k]);
    }
    minPrev = sg_rowmin(m);
#pragma unroll
    for (int k = 0; k < K; k++) L[k] = Ln[k];
}
// The same step on PACKED pairs: a lane's K disparities live two per dword as u16 (every quantity of the recurrence stays below 2^16: costs < 2^15,
// MAX_COST + P1 = 33251), so one v_pk_add_u16 / v_pk_min_u16 / v_pk_sub_u16 serves two disparities, the neighbour pairs (d-1, d+1) are one v_alignbyte /
// v_perm each, and the costs arrive from memory already in this form (K u16 = one 8-byte load + one 2-byte load for K = 5).  A scan path is ONE wave
// alone on its SIMD -- it issues an instruction every ~6 cycles whatever the instruction does (profiles/r02_valu_rate.md, column "@1 wave/SIMD") -- so
// a path's time is its instruction count: ~45 per step here against ~95 for the 32-bit form.  Odd K: the pad slot of the last pair is kept at 0xFFFF.
template <int K>
__device__ __forceinline__ void sg_step_pk(uint32_t (&L)[(K + 1) / 2], int& minPrev, const uint32_t (&Cp)[(K + 1) / 2], uint32_t P1P1, int P2)
{
    constexpr int NP = (K + 1) / 2;
    constexpr bool ODD = (K & 1) != 0;
    constexpr uint32_t MAXMAX = (uint32_t)SG_MAXC | ((uint32_t)SG_MAXC << 16);
    // the left lane's last pair and the right lane's first pair (MAX_COST beyond the ends of the disparity range)
    const uint32_t lft = (uint32_t)sg_dpp<0x111>((int)MAXMAX, (int)L[NP - 1]);        // row_shr:1
    const uint32_t rgt = (uint32_t)sg_dpp<0x101>((int)MAXMAX, (int)L[0]);             // row_shl:1
    const uint32_t dd = (uint32_t)(minPrev + P2) * 0x00010001u;
    uint32_t Ln[NP], m = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        // (slot 2j-1, slot 2j) and (slot 2j+1, slot 2j+2)
        const uint32_t lm = j > 0 ? __builtin_amdgcn_alignbyte(L[j], L[j - 1], 2)
                                  : (ODD ? __builtin_amdgcn_perm(L[0], lft, 0x05040100u) : __builtin_amdgcn_alignbyte(L[0], lft, 2));
        const uint32_t lp = j < NP - 1 ? __builtin_amdgcn_alignbyte(L[j + 1], L[j], 2)
                                       : (ODD ? rgt : __builtin_amdgcn_alignbyte(rgt, L[j], 2));
        const uint32_t t = pk_min16(pk_min16(L[j], pk_add16(lm, P1P1)), pk_min16(pk_add16(lp, P1P1), dd));
        Ln[j] = pk_sub16(pk_add16(Cp[j], t), dd);
        if (ODD && j == NP - 1) Ln[j] |= 0xFFFF0000u;             // the pad slot never wins a minimum
        m = pk_min16(m, Ln[j]);
    }
    minPrev = sg_rowmin((int)min(m & 0xFFFFu, m >> 16));
#pragma unroll
    for (int j = 0; j < NP; j++) L[j] = Ln[j];
}
// K consecutive u16 <-> packed pairs (exact sizes: nothing beyond the lane's own K values is touched)
template <int K> __device__ __forceinline__ void sg_load_pk(uint32_t (&d)[(K + 1) / 2], const uint16_t* p)
{
#pragma unroll
    for (int j = 0; j < (K + 1) / 2; j++) d[j] = 0;
    __builtin_memcpy(d, p, 2 * K);
}
template <int K> __device__ __forceinline__ void sg_store_pk(uint16_t* p, const uint32_t (&d)[(K + 1) / 2]) { __builtin_memcpy(p, d, 2 * K); }
// MODE 0: r = (-1, 0): path = row y, steps x = 0 .. w1-1        MODE 4: r = (+1, 0): path = row y, steps x = w1-1 .. 0
// MODE 1..3: r = (-1,-1), (0,-1), (+1,-1): path = diagonal / column, steps y = 0 .. h-1
// Every direction writes its own L volume: the five launches share nothing but C and run concurrently on five streams.
template <int K, int MODE>
__global__ void __launch_bounds__(256)
sgbm_path(const uint16_t* __restrict__ C_all, uint16_t* __restrict__ Lout_all, int w1, int h, int P1, int P2)
{
    constexpr int D = 16 * K;
    const uint16_t* C = C_all + (size_t)blockIdx.y * w1 * h * D; uint16_t* Lout = Lout_all + (size_t)blockIdx.y * w1 * h * D;      // blockIdx.y = frame
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, li = threadIdx.x & 15;
    constexpr bool ROW = MODE == 0 || MODE == 4;
    const int npaths = ROW ? h : (MODE == 2 ? w1 : w1 + h - 1);
    const bool live = g < npaths;                             // dead groups run the loop too (DPP wants the whole wave), clamped to path 0
    const int gp = live ? g : 0;
    const int rx = MODE == 1 ? 1 : MODE == 3 ? -1 : 0;        // x(y) = o + rx * y
    const int o = MODE == 1 ? gp - (h - 1) : gp;
    constexpr int NP = (K + 1) / 2;
    constexpr uint32_t LZERO_LAST = (K & 1) ? 0xFFFF0000u : 0u;   // L = 0 with the pad slot parked at 0xFFFF
    uint32_t L[NP]; int minPrev = 0;
#pragma unroll
    for (int j = 0; j < NP; j++) L[j] = j == NP - 1 ? LZERO_LAST : 0u;
    const uint32_t P1P1 = (uint32_t)P1 * 0x00010001u;
    const int steps = ROW ? w1 : h;
    // a diagonal is inside the image only for part of the rows (x(y) = o + rx y in [0, w1)): the wave walks the union of its four paths' ranges -- neighbouring
    // offsets, so nearly the same range -- instead of all h rows (1616 diagonals x 376 rows is 30 % more steps than the image has pixels); a path still
    // outside the image inside that range starts from the zeroed border as before
    int t_lo = 0, t_hi = steps;
    if (MODE == 1 || MODE == 3) {
        const int g0 = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4;      // the wave's first path
        const int oa = (MODE == 1 ? g0 - (h - 1) : g0), ob = oa + 3;              // offsets of its first and last path (paths beyond npaths: clamped later, any range is fine)
        if (MODE == 1) { t_lo = max(0, -ob); t_hi = min(h, w1 - oa); }
        else { t_lo = max(0, oa - (w1 - 1)); t_hi = min(h, ob + 1); }
        if (g0 + 3 >= npaths) { t_lo = 0; t_hi = steps; }                         // a wave with clamped (dead) groups: path 0's range
        t_lo = __builtin_amdgcn_readfirstlane(t_lo); t_hi = __builtin_amdgcn_readfirstlane(max(t_hi, t_lo));
    }
    // the costs of SG_UN steps are loaded together and the next group's loads are issued before the current group's steps run (two register sets):
    // a path pays the memory latency once per 2 SG_UN steps at most, and the addresses of every path are known in advance (x(t) = o + rx t)
    constexpr int SG_UN = 16;
    uint32_t Ca[SG_UN][NP], Cb[SG_UN][NP];
    auto pix_of = [&](int t, int& x, int& y) { if (ROW) { y = gp; x = MODE == 0 ? t : w1 - 1 - t; } else { y = t; x = o + rx * t; } };
    auto load_group = [&](uint32_t (&Cq)[SG_UN][NP], int t0) {
#pragma unroll
        for (int u = 0; u < SG_UN; u++) {
            int x, y; pix_of(min(t0 + u, steps - 1), x, y);
            sg_load_pk<K>(Cq[u], C + ((size_t)y * w1 + (x >= 0 && x < w1 ? x : 0)) * D + li * K);
        }
    };
    auto run_group = [&](const uint32_t (&Cq)[SG_UN][NP], int t0) {
#pragma unroll
        for (int u = 0; u < SG_UN; u++) {
            const int t = t0 + u;
            if (t >= t_hi) break;                                 // wave-uniform
            int x, y; pix_of(t, x, y);
            const bool in = ROW || (x >= 0 && x < w1);
            if (!in) {                                            // outside the image the predecessor is OpenCV's zeroed border
#pragma unroll
                for (int j = 0; j < NP; j++) L[j] = j == NP - 1 ? LZERO_LAST : 0u;
                minPrev = 0;
            }
            uint32_t Lc[NP]; int mp = minPrev;
#pragma unroll
            for (int j = 0; j < NP; j++) Lc[j] = L[j];
            sg_step_pk<K>(Lc, mp, Cq[u], P1P1, P2);
            if (in) {
#pragma unroll
                for (int j = 0; j < NP; j++) L[j] = Lc[j];
                minPrev = mp;
                if (live) sg_store_pk<K>(Lout + ((size_t)y * w1 + x) * D + li * K, Lc);
            }
        }
    };
    load_group(Ca, t_lo);
    for (int t0 = t_lo; t0 < t_hi; t0 += 2 * SG_UN) {
        load_group(Cb, t0 + SG_UN);
        run_group(Ca, t0);
        if (t0 + SG_UN >= t_hi) break;
        load_group(Ca, t0 + 2 * SG_UN);
        run_group(Cb, t0 + SG_UN);
    }
}
// ------------------------------------------------------------------ the column direction with the winner pass inside
// sgbm_path<K, 2> that does not write its L volume: it runs behind the other four directions, reads their L at the pixel it has just aggregated, and does
// sgbm_wta's work for that pixel on the spot (every pixel of the volume lies on exactly one column path).  The L volume of this direction is never written
// and the winner pass reads four volumes instead of five: 140 MB of 1.22 GB per 1241 x 376 x 80 frame.
#define SGW_UN 6
template <int K>
__global__ void __launch_bounds__(256)
sgbm_col_wta(const uint16_t* __restrict__ C_all, const uint16_t* __restrict__ L0, const uint16_t* __restrict__ L1, const uint16_t* __restrict__ L3,
             const uint16_t* __restrict__ L4, int w, int w1, int h, int P1, int P2, int minD, int minX1, int uniquenessRatio,
             int16_t* __restrict__ disp1, unsigned* __restrict__ disp2key)
{
    constexpr int D = 16 * K, NP = (K + 1) / 2;
    {   const size_t fv = (size_t)blockIdx.y * w1 * h * D, fp = (size_t)blockIdx.y * w * h;      // blockIdx.y = frame
        C_all += fv; L0 += fv; L1 += fv; L3 += fv; L4 += fv; disp1 += fp; disp2key += fp; }
    __shared__ uint16_t srow[16][D];                          // S of the group's pixel, for the three sub-pixel taps
    const int gl = threadIdx.x >> 4, li = threadIdx.x & 15;
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool live = g < w1;
    const int x = live ? g : 0;
    constexpr uint32_t LZERO_LAST = (K & 1) ? 0xFFFF0000u : 0u;
    uint32_t L[NP]; int minPrev = 0;
#pragma unroll
    for (int j = 0; j < NP; j++) L[j] = j == NP - 1 ? LZERO_LAST : 0u;
    const uint32_t P1P1 = (uint32_t)P1 * 0x00010001u;
    uint32_t Va[SGW_UN][5][NP], Vb[SGW_UN][5][NP];           // [step][C, L0, L1, L3, L4][pairs]
    auto load_group = [&](uint32_t (&V)[SGW_UN][5][NP], int t0) {
#pragma unroll
        for (int u = 0; u < SGW_UN; u++) {
            const size_t off = ((size_t)min(t0 + u, h - 1) * w1 + x) * D + li * K;
            sg_load_pk<K>(V[u][0], C_all + off); sg_load_pk<K>(V[u][1], L0 + off); sg_load_pk<K>(V[u][2], L1 + off);
            sg_load_pk<K>(V[u][3], L3 + off); sg_load_pk<K>(V[u][4], L4 + off);
        }
    };
    auto run_group = [&](const uint32_t (&V)[SGW_UN][5][NP], int t0) {
#pragma unroll
        for (int u = 0; u < SGW_UN; u++) {
            const int y = t0 + u;
            if (y >= h) break;                                    // wave-uniform
            sg_step_pk<K>(L, minPrev, V[u][0], P1P1, P2);
            // ---- the winner pass for pixel (y, x): S = min(32767, L0 + L1 + L2 + L3 + L4)
            int Sv[K], best = INT_MAX;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int sh = 16 * (k & 1), j = k >> 1;
                Sv[k] = min((int)((L[j] >> sh) & 0xFFFFu) + (int)((V[u][1][j] >> sh) & 0xFFFFu) + (int)((V[u][2][j] >> sh) & 0xFFFFu) +
                            (int)((V[u][3][j] >> sh) & 0xFFFFu) + (int)((V[u][4][j] >> sh) & 0xFFFFu), SG_MAXC);
                best = min(best, (Sv[k] << 8) | (li * K + k));
                srow[gl][li * K + k] = (uint16_t)Sv[k];
            }
            best = sg_rowmin(best);
            const int minS = best >> 8, bestDisp = best & 255;
            bool bad = false;
#pragma unroll
            for (int k = 0; k < K; k++) bad |= Sv[k] * (100 - uniquenessRatio) < minS * 100 && abs(bestDisp - (li * K + k)) > 1;
            const unsigned long long bal = __ballot(bad);
            const bool rejected = ((bal >> (threadIdx.x & 48)) & 0xFFFFull) != 0;       // any lane of my 16-lane group
            if (live && !rejected && li == 0) {
                int d = bestDisp;
                const int x2 = x + minX1 - d - minD;
                if (minS < SG_MAXC) atomicMin(&disp2key[(size_t)y * w + x2], ((unsigned)minS << 16) | (unsigned)(65535 - x));
                if (0 < d && d < D - 1) {
                    const int sm = srow[gl][d - 1], s0 = srow[gl][d], sp = srow[gl][d + 1];
                    const int denom2 = max(sm + sp - 2 * s0, 1);
                    d = d * SG_DISP_SCALE + ((sm - sp) * SG_DISP_SCALE + denom2) / (denom2 * 2);
                } else d *= SG_DISP_SCALE;
                disp1[(size_t)y * w + x + minX1] = (int16_t)(d + minD * SG_DISP_SCALE);
            }
            __builtin_amdgcn_wave_barrier();                   // (the next step overwrites srow: LDS operations of a wave execute in order)
        }
    };
    load_group(Va, 0);
    for (int t0 = 0; t0 < h; t0 += 2 * SGW_UN) {
        load_group(Vb, t0 + SGW_UN);
        run_group(Va, t0);
        if (t0 + SGW_UN >= h) break;
        load_group(Va, t0 + 2 * SGW_UN);
        run_group(Vb, t0 + SGW_UN);
    }
}
// ------------------------------------------------------------------ round 4: TWO volumes (C and the row sum) instead of C + four L volumes
// The five directions as two kernels that leave no per-direction volume behind (1.11 GB -> ~0.55 GB of HBM traffic per 1241 x 376 x 80 pair):
//   sgbm_rows   both horizontal directions of a row by the same 16 lanes -> S04 = min(32767, L0 + L4), ONE volume.  L0 of a whole row (w1 x D x 2 B =
//               186 KB) fits neither registers nor a CU's LDS, so the row is cut into segments of SGR_SEG columns: pass 1 walks the row left to right and
//               keeps only L0 at the segment starts (a checkpoint volume of 1 / SGR_SEG of C); pass 2 walks the segments right to left -- the segment's
//               costs are loaded once into registers, L0 is re-run forward from the checkpoint into registers, then L4 runs backward over the same
//               registers and the saturated sum is stored.  Three recurrence steps per pixel instead of two, no L volume written or read.
//   sgbm_sweep  the three directions that come from the previous row, (-1,-1), (0,-1), (+1,-1), AND the winner pass in one top-down sweep: a block owns a
//               strip of TX columns of one frame, a 16-lane group CPG neighbouring columns, and the states L1, L2, L3 of the previous row live in registers.
//               The diagonal predecessors cross groups through LDS (one barrier per row) and cross STRIPS through mailboxes in global memory: 8-byte
//               {row tag, two costs} granules written by one agent-scope (sc1) store each -- the data is the flag, no fence (cdna_hip_programming.md G16
//               R2) -- that the edge group of the neighbouring strip polls at the start of the next row; they are published right after the row's steps,
//               before its winner pass, so a hand-off has most of a row time to land.  S = min(32767, S04 + L1 + L2 + L3) never leaves registers.
//               Forward progress: the strips of a frame wait for each other, so a block takes its (frame, strip) from a TICKET (atomic counter) -- blocks
//               that run hold the lowest tickets whatever order the hardware starts them in, so every frame whose strips all run finishes and frees its CUs;
//               every spin is bounded and a time-out sets a flag the host reports (SSM_E_HIP) instead of hanging.
// Saturation: every L is >= 0, so min(32767, a + b + ...) may be taken after every addition (exact), and partial sums of two values < 2^16 ... are kept
// below 2^16 by clamping each operand to 32767 first.
typedef unsigned long long sg_u64;
#define SG_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define SGR_SEG 8
#define SGS_SLOTS 4
#define SGS_SPIN_LIMIT (1u << 20)
template <int NP> __device__ __forceinline__ int sg_min_of(const uint32_t (&L)[NP])
{
    uint32_t m = L[0];
#pragma unroll
    for (int j = 1; j < NP; j++) m = pk_min16(m, L[j]);
    return sg_rowmin((int)min(m & 0xFFFFu, m >> 16));
}
// min(32767, a + b) on packed pairs with a, b <= 32767: ONE instruction, v_pk_add_i16 with the clamp bit (both are non-negative as i16, the sum saturates at 32767)
typedef short sg_i2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_addsat_i15(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(sg_i2v, a), __builtin_bit_cast(sg_i2v, b)));
}
// min(32767, a + b) on packed pairs, a and b any u16
__device__ __forceinline__ uint32_t pk_addsat15(uint32_t a, uint32_t b)
{
    constexpr uint32_t MM = 0x7FFF7FFFu;
    return pk_min16(pk_add16(pk_min16(a, MM), pk_min16(b, MM)), MM);
}
// (the 16-lane form of this kernel, sgbm_rows<K, SEG>: D / 16 disparities per lane -- measured slower at every D (profiles/r04_stereo_*), removed in round 6)
// sgbm_rows with EIGHT disparities per lane: a row is owned by D / 8 of a DPP row's 16 lanes (10 at D = 80), a lane's costs are ONE aligned 16-byte word.
// sgbm_rows' five u16 per lane are an 8-byte + a 2-byte access at 2-byte alignment, for all 16 lanes: the kernel sat at 0.80 (busiest CU 0.94) of its
// texture-address units' time (profiles/r04_stereo_ta_busy.md) with HBM at 4 TB/s; here a column is 10 lane-accesses instead of 32.  The idle lanes run along
// (DPP wants the wave): their minimum is parked at 0xFFFF and the last active lane's d + 1 neighbour is MAX_COST, like a group's edge in sg_step8.
__device__ __forceinline__ void sg_step_pk8(uint32_t (&L)[4], int& minPrev, const uint32_t (&Cp)[4], uint32_t P1P1, int P2, bool act, bool last)
{
    constexpr uint32_t MAXMAX = (uint32_t)SG_MAXC | ((uint32_t)SG_MAXC << 16);
    const uint32_t MAXP = pk_add16(MAXMAX, P1P1);                                   // L + P1 once per pair: sg_step8
    const uint32_t dd = (uint32_t)(minPrev + P2) * 0x00010001u;
    uint32_t Ln[4], LP[4], m = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 4; j++) LP[j] = pk_add16(L[j], P1P1);
    const uint32_t lftp = (uint32_t)sg_dpp<0x111>((int)MAXP, (int)LP[3]);           // row_shr:1 (lane 0 of the row: MAX_COST + P1)
    uint32_t rgtp = (uint32_t)sg_dpp<0x101>((int)MAXP, (int)LP[0]);                 // row_shl:1
    rgtp = last ? MAXP : rgtp;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t lm = __builtin_amdgcn_alignbyte(LP[j], j > 0 ? LP[j - 1] : lftp, 2);
        const uint32_t lp = __builtin_amdgcn_alignbyte(j < 3 ? LP[j + 1] : rgtp, LP[j], 2);
        const uint32_t t = pk_min16(pk_min16(L[j], lm), pk_min16(lp, dd));
        Ln[j] = pk_sub16(pk_add16(Cp[j], t), dd);
        m = pk_min16(m, Ln[j]);
    }
    m = act ? m : 0xFFFFFFFFu;
    minPrev = sg_rowmin((int)min(m & 0xFFFFu, m >> 16));
#pragma unroll
    for (int j = 0; j < 4; j++) L[j] = Ln[j];
}
template <int SEG, bool FAST>
__global__ void __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(2, 2)))      // (167 registers would allow three waves per SIMD: measured, the stereo path as a whole loses 3 % -- 5.47 k vs 5.68 k pairs/s -- with the kernel's own time unchanged: it is HBM-bound and the third wave only takes bandwidth from the kernels of the other streams earlier)
sgbm_rows8(const uint16_t* __restrict__ C_all, uint16_t* __restrict__ S_all, uint16_t* __restrict__ ck_all, int w1, int h, int D, int P1, int P2)
{
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, li = threadIdx.x & 15, NL = D >> 3;
    const bool act = li < NL, last = li == NL - 1;
    const bool live = g < h && act;                           // dead groups run row 0 without stores (DPP wants the whole wave); idle lanes read lane 0's words
    const int y = g < h ? g : 0, lc = act ? li : 0;
    const int nseg = (w1 + SEG - 1) / SEG;
    const size_t rowi = (size_t)blockIdx.y * h + y;           // blockIdx.y = frame
    const uint4* Crow = reinterpret_cast<const uint4*>(C_all + rowi * w1 * D) + lc;       // a column = NL words
    uint4* Srow = reinterpret_cast<uint4*>(S_all + rowi * w1 * D) + lc;
    uint4* ck = reinterpret_cast<uint4*>(ck_all + rowi * nseg * D) + lc;                  // ck[s]: L0 in front of segment s (s >= 1)
    const uint32_t P1P1 = (uint32_t)P1 * 0x00010001u;
    uint32_t Ca[SEG][4], Cb[SEG][4];
    auto ld = [&](uint32_t (&d)[4], const uint4* p) { const uint4 t = *p; d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; };
    auto load_seg = [&](uint32_t (&Cq)[SEG][4], int s) {
#pragma unroll
        for (int u = 0; u < SEG; u++) ld(Cq[u], Crow + (size_t)min(s * SEG + u, w1 - 1) * NL);
    };
    // ---- pass 1: L0 left to right, checkpoints only
    if (nseg > 1) {
        uint32_t L[4] = {0u, 0u, 0u, 0u}; int mp = 0;
        auto fwd_seg = [&](const uint32_t (&Cq)[SEG][4], int s) {
#pragma unroll
            for (int u = 0; u < SEG; u++) sg_step_pk8(L, mp, Cq[u], P1P1, P2, act, last);
            if (live) ck[(size_t)(s + 1) * NL] = make_uint4(L[0], L[1], L[2], L[3]);
        };
        // The next segment's loads are UNCONDITIONAL (the index clamped, a row's last prefetch repeats a segment): behind a conditional prefetch the wait-count pass
        // has no guaranteed number of younger loads and waits for the current segment with vmcnt(0), i.e. for the prefetch it has just issued -- the double
        // buffer never overlapped a load with a step (round 6, found in the ISA: L x 12, s_waitcnt vmcnt(0)).
        load_seg(Ca, 0);
        for (int s = 0; s < nseg - 1; s += 2) {
            load_seg(Cb, min(s + 1, nseg - 1));
            fwd_seg(Ca, s);
            if (s + 1 >= nseg - 1) break;
            load_seg(Ca, min(s + 2, nseg - 1));
            fwd_seg(Cb, s + 1);
        }
    }
    // ---- pass 2: segments right to left; L0 forward from the checkpoint, L4 backward, the sum out
    uint32_t R[4] = {0u, 0u, 0u, 0u}; int mpr = 0;
    uint32_t Fa[4], Fb[4];
    auto load_ck = [&](uint32_t (&F)[4], int s) { ld(F, ck + (size_t)min(max(s, 1), nseg - 1) * NL); };      // (unconditional like the segments; segment 0 starts from zero: seg_run)
    auto seg_run = [&](const uint32_t (&Cq)[SEG][4], uint32_t (&F)[4], int s) {
        if (s <= 0) { F[0] = F[1] = F[2] = F[3] = 0u; }                           // (wave-uniform)
        uint32_t m = pk_min16(pk_min16(F[0], F[1]), pk_min16(F[2], F[3]));
        m = act ? m : 0xFFFFFFFFu;
        int mpf = sg_rowmin((int)min(m & 0xFFFFu, m >> 16));
        uint32_t L0[SEG][4];
#pragma unroll
        for (int u = 0; u < SEG; u++) {
            if (s * SEG + u < w1) sg_step_pk8(F, mpf, Cq[u], P1P1, P2, act, last);          // (wave-uniform; false only in the last segment)
#pragma unroll
            for (int j = 0; j < 4; j++) L0[u][j] = F[j];
        }
#pragma unroll
        for (int u = SEG - 1; u >= 0; u--) {
            const int x = s * SEG + u;
            if (x < w1) {
                sg_step_pk8(R, mpr, Cq[u], P1P1, P2, act, last);
                if (live) {
                    if (FAST) Srow[(size_t)x * NL] = make_uint4(pk_addsat_i15(L0[u][0], R[0]), pk_addsat_i15(L0[u][1], R[1]), pk_addsat_i15(L0[u][2], R[2]), pk_addsat_i15(L0[u][3], R[3]));      // every L < 2^15 (the launcher's bound on the costs)
                    else Srow[(size_t)x * NL] = make_uint4(pk_addsat15(L0[u][0], R[0]), pk_addsat15(L0[u][1], R[1]), pk_addsat15(L0[u][2], R[2]), pk_addsat15(L0[u][3], R[3]));
                }
            }
        }
    };
    load_seg(Ca, nseg - 1); load_ck(Fa, nseg - 1);
    for (int s = nseg - 1; s >= 0; s -= 2) {
        load_seg(Cb, max(s - 1, 0)); load_ck(Fb, s - 1);
        seg_run(Ca, Fa, s);
        if (s - 1 < 0) break;
        load_seg(Ca, max(s - 2, 0)); load_ck(Fa, s - 2);
        seg_run(Cb, Fb, s - 1);
    }
}
// mailbox of one (frame, seam, direction): [SGS_SLOTS][NP + 1][16 lanes] granules; granule j < NP = the lane's packed pair j, granule NP = the path's minimum
template <int NG> __device__ __forceinline__ bool sg_mbox_wait(const sg_u64* g, unsigned epoch, uint32_t (&v)[NG], unsigned* flags)
{
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < NG; j++) { const sg_u64 x = __hip_atomic_load(g + j * 16, SG_RLX_AGENT); v[j] = (uint32_t)x; ok &= (unsigned)(x >> 32) == epoch; }
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) return true;                   // (the lanes of the polling group only: the others are masked off)
        if (spins >= SGS_SPIN_LIMIT || ((spins & 255u) == 255u && __hip_atomic_load(flags + 1, SG_RLX_AGENT) != 0u)) return false;
        __builtin_amdgcn_s_sleep(4);
    }
}
#define SGS_UN 1
// exact n / d (C's truncating division) for |n| < 2^22, 0 < d < 2^22 without the ~40-instruction integer division sequence: the float quotient is within one
// of the true one, the remainder test repairs it
__device__ __forceinline__ int sg_div_small(int n, int d)
{
    const int an = abs(n);
    int q = (int)((float)an * __builtin_amdgcn_rcpf((float)d));
    int r = an - q * d;
    if (r < 0) { q--; r += d; }
    if (r >= d) q++;
    return n < 0 ? -q : q;
}
// FAST: the launcher has checked that no cost can reach 2^15 (C <= P2 + SADWindowSize^2 x the largest pixel cost), so every L is below 2^15 and a sum of two
// cannot wrap: three packed operations per addition less
template <int K, int CPG, bool FAST>
__global__ void __launch_bounds__(1024)
sgbm_sweep(const uint16_t* __restrict__ C_all, const uint16_t* __restrict__ S_all, int w, int w1, int h, int P1, int P2, int minD, int minX1, int uniquenessRatio,
           int NS, int TX, int16_t* __restrict__ disp1, unsigned* __restrict__ disp2key, unsigned* flags /* [0] ticket counter, [1] time-out */, sg_u64* mbox_all, int* fail_out)
{
    constexpr int D = 16 * K, NP = (K + 1) / 2, NG = NP + 1;
    constexpr uint32_t LZERO_LAST = (K & 1) ? 0xFFFF0000u : 0u;
    extern __shared__ __align__(16) uint8_t sw_smem[];
    __shared__ unsigned s_ticket; __shared__ int s_fail[2];
    const int ng = blockDim.x >> 4, g = threadIdx.x >> 4, li = threadIdx.x & 15;
    uint32_t* xch = reinterpret_cast<uint32_t*>(sw_smem);     // [parity 2][direction 2][ng][NG][16]: the state a group hands to its right (dir 0: L1 of its last column) / left (dir 1: L3 of its first column) neighbour
    uint16_t* srow = reinterpret_cast<uint16_t*>(xch + (size_t)4 * ng * NG * 16) + (size_t)g * D;      // [ng][D]: S of the group's pixel for the sub-pixel taps
    auto xslot = [&](int par, int dir, int gg) -> uint32_t* { return xch + ((size_t)((par * 2 + dir) * ng + gg) * NG) * 16 + li; };
    if (threadIdx.x == 0) { s_ticket = atomicAdd(&flags[0], 1u); s_fail[0] = 0; s_fail[1] = 0; }
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {                       // "row -1": OpenCV's zeroed border
        uint32_t* p = xslot(1, dir, g);
#pragma unroll
        for (int j = 0; j < NP; j++) p[j * 16] = j == NP - 1 ? LZERO_LAST : 0u;
        p[NP * 16] = 0u;
    }
    __syncthreads();
    const int t = (int)s_ticket, f = t / NS, strip = t - f * NS;
    const int ngu = TX / CPG;                                 // groups of the block that own columns
    const int x0 = strip * TX + g * CPG, xend = min((strip + 1) * TX, w1);
    const uint16_t* Cf = C_all + (size_t)f * w1 * h * D + li * K; const uint16_t* Sf = S_all + (size_t)f * w1 * h * D + li * K;
    disp1 += (size_t)f * w * h; disp2key += (size_t)f * w * h;
    sg_u64* mb = mbox_all + (size_t)f * (NS - 1) * 2 * SGS_SLOTS * NG * 16 + li;
    auto mslot = [&](int seam, int dir, int slot) -> sg_u64* { return mb + (size_t)(((seam * 2 + dir) * SGS_SLOTS + slot) * NG) * 16; };
    const bool usedg = g < ngu;
    const bool edgeL = g == 0 && strip > 0, edgeR = g == ngu - 1 && strip < NS - 1;
    const uint32_t P1P1 = (uint32_t)P1 * 0x00010001u;
    const int udiv = 100 - uniquenessRatio;
    uint32_t L1[CPG][NP], L2[CPG][NP], L3[CPG][NP]; int m1[CPG], m2[CPG], m3[CPG];
#pragma unroll
    for (int c = 0; c < CPG; c++) {
#pragma unroll
        for (int j = 0; j < NP; j++) L1[c][j] = L2[c][j] = L3[c][j] = j == NP - 1 ? LZERO_LAST : 0u;
        m1[c] = m2[c] = m3[c] = 0;
    }
    uint32_t Va[SGS_UN][CPG][2][NP], Vb[SGS_UN][CPG][2][NP];       // [row][column][C, S04][pairs]
    auto load_rows = [&](uint32_t (&V)[SGS_UN][CPG][2][NP], int y0) {
#pragma unroll
        for (int u = 0; u < SGS_UN; u++)
#pragma unroll
            for (int c = 0; c < CPG; c++) {
                const size_t off = ((size_t)min(y0 + u, h - 1) * w1 + min(x0 + c, w1 - 1)) * D;
                sg_load_pk<K>(V[u][c][0], Cf + off); sg_load_pk<K>(V[u][c][1], Sf + off);
            }
    };
    bool stop = false;
    auto run_rows = [&](const uint32_t (&V)[SGS_UN][CPG][2][NP], int y0) {
#pragma unroll
        for (int u = 0; u < SGS_UN; u++) {
            const int y = y0 + u;
            if (y >= h || stop) break;                            // block-uniform
            // ---- the steps whose predecessor lives in this group's registers first: L1 of columns 1 .. CPG-1 (from the column to the left), L3 of columns
            // 0 .. CPG-2 (from the column to the right), L2 of every column.  L1 and L3 are independent recurrences -- L1 flows left to right, L3 right to left --
            // so what a strip hands to its right neighbour (L1 of its last column) never depends on what it receives from it (L3), and is published BEFORE
            // the incoming mailboxes are polled: a hand-off has a whole row time to land instead of gating the neighbour's next publish (polling first made
            // every row a round trip, poll -> steps -> publish -> latency: 5.5 - 6.9 us per row against 3.8 us of VALU issue)
            static_assert(CPG >= 2, "the strip edges publish states computed from the group's own registers");
#pragma unroll
            for (int c = CPG - 1; c >= 1; c--) {
#pragma unroll
                for (int j = 0; j < NP; j++) L1[c][j] = L1[c - 1][j];
                m1[c] = m1[c - 1];
                sg_step_pk<K>(L1[c], m1[c], V[u][c][0], P1P1, P2);
            }
#pragma unroll
            for (int c = 0; c < CPG - 1; c++) {
#pragma unroll
                for (int j = 0; j < NP; j++) L3[c][j] = L3[c + 1][j];
                m3[c] = m3[c + 1];
                sg_step_pk<K>(L3[c], m3[c], V[u][c][0], P1P1, P2);
            }
            if (edgeR) {                                          // (all columns of a strip that has a right neighbour are inside the image)
                sg_u64* o = mslot(strip, 0, y & (SGS_SLOTS - 1));
#pragma unroll
                for (int j = 0; j < NP; j++) __hip_atomic_store(o + j * 16, ((sg_u64)(unsigned)(y + 1) << 32) | L1[CPG - 1][j], SG_RLX_AGENT);
                __hip_atomic_store(o + NP * 16, ((sg_u64)(unsigned)(y + 1) << 32) | (uint32_t)m1[CPG - 1], SG_RLX_AGENT);
            }
            if (edgeL) {
                sg_u64* o = mslot(strip - 1, 1, y & (SGS_SLOTS - 1));
#pragma unroll
                for (int j = 0; j < NP; j++) __hip_atomic_store(o + j * 16, ((sg_u64)(unsigned)(y + 1) << 32) | L3[0][j], SG_RLX_AGENT);
                __hip_atomic_store(o + NP * 16, ((sg_u64)(unsigned)(y + 1) << 32) | (uint32_t)m3[0], SG_RLX_AGENT);
            }
#pragma unroll
            for (int c = 0; c < CPG; c++) sg_step_pk<K>(L2[c], m2[c], V[u][c][0], P1P1, P2);
            // ---- the diagonal predecessors from outside the group (row y - 1): the neighbouring groups' through LDS, the neighbouring strips' through the mailboxes
            uint32_t nl[NG], nr[NG];
            const int pp = (y + 1) & 1;
            {   const uint32_t* p = xslot(pp, 0, g > 0 ? g - 1 : 0);
#pragma unroll
                for (int j = 0; j < NG; j++) nl[j] = p[j * 16];
                const uint32_t* q = xslot(pp, 1, g < ng - 1 ? g + 1 : g);
#pragma unroll
                for (int j = 0; j < NG; j++) nr[j] = q[j * 16];
            }
            if (g == 0 || g >= ngu - 1) {                         // strip borders: the neighbouring strip's mailbox, or the zeroed image border
                bool okl = true, okr = true;
                if (g == 0) {
                    if (edgeL && y > 0) okl = sg_mbox_wait<NG>(mslot(strip - 1, 0, (y - 1) & (SGS_SLOTS - 1)), (unsigned)y, nl, flags);
                    else {
#pragma unroll
                        for (int j = 0; j < NG; j++) nl[j] = j == NP - 1 ? LZERO_LAST : 0u;
                    }
                }
                if (g >= ngu - 1) {
                    if (edgeR && y > 0) okr = sg_mbox_wait<NG>(mslot(strip, 1, (y - 1) & (SGS_SLOTS - 1)), (unsigned)y, nr, flags);
                    else {
#pragma unroll
                        for (int j = 0; j < NG; j++) nr[j] = j == NP - 1 ? LZERO_LAST : 0u;
                    }
                }
                if (!(okl && okr)) { s_fail[y & 1] = 1; __hip_atomic_store(flags + 1, 1u, SG_RLX_AGENT); if (fail_out) atomicOr(fail_out, 1); }
            }
            {
#pragma unroll
                for (int j = 0; j < NP; j++) { L1[0][j] = nl[j]; L3[CPG - 1][j] = nr[j]; }
                m1[0] = (int)nl[NP]; m3[CPG - 1] = (int)nr[NP];
                sg_step_pk<K>(L1[0], m1[0], V[u][0][0], P1P1, P2);
                sg_step_pk<K>(L3[CPG - 1], m3[CPG - 1], V[u][CPG - 1][0], P1P1, P2);
            }
#pragma unroll
            for (int c = 0; c < CPG; c++)
                if (!(usedg && x0 + c < xend)) {                  // a column outside the strip / image: its neighbours see the zeroed border
#pragma unroll
                    for (int j = 0; j < NP; j++) L1[c][j] = L2[c][j] = L3[c][j] = j == NP - 1 ? LZERO_LAST : 0u;
                    m1[c] = m2[c] = m3[c] = 0;
                }
            // ---- the border states for the neighbouring groups
            {   uint32_t* p = xslot(y & 1, 0, g);
#pragma unroll
                for (int j = 0; j < NP; j++) p[j * 16] = L1[CPG - 1][j];
                p[NP * 16] = (uint32_t)m1[CPG - 1];
                uint32_t* q = xslot(y & 1, 1, g);
#pragma unroll
                for (int j = 0; j < NP; j++) q[j * 16] = L3[0][j];
                q[NP * 16] = (uint32_t)m3[0];
            }
            // ---- the winner pass of the group's pixels (sgbm_wta's arithmetic on S = min(32767, S04 + L1 + L2 + L3))
#pragma unroll
            for (int c = 0; c < CPG; c++) {
                const bool live = usedg && x0 + c < xend;
                const int x = x0 + c;
                uint32_t sp2[NP];
#pragma unroll
                for (int j = 0; j < NP; j++) {
                    if (FAST) {                                   // L < 2^15, S04 <= 32767: pair sums stay below 2^16
                        constexpr uint32_t MM = 0x7FFF7FFFu;
                        sp2[j] = pk_min16(pk_add16(pk_min16(pk_add16(L1[c][j], L2[c][j]), MM), pk_min16(pk_add16(L3[c][j], V[u][c][1][j]), MM)), MM);
                    } else sp2[j] = pk_addsat15(pk_addsat15(pk_addsat15(L1[c][j], L2[c][j]), L3[c][j]), V[u][c][1][j]);
                }
                int Sv[K], best = INT_MAX;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const int sh = 16 * (k & 1), j = k >> 1;
                    Sv[k] = (int)((sp2[j] >> sh) & 0xFFFFu);
                    best = min(best, (Sv[k] << 8) | (li * K + k));
                    srow[li * K + k] = (uint16_t)Sv[k];
                }
                best = sg_rowmin(best);
                const int minS = best >> 8, bestDisp = best & 255;
                // "S (100 - u) < 100 minS and |d - best| > 1": for 0 <= u < 100 the first test is S <= floor((100 minS - 1) / (100 - u)), one division per pixel
                bool bad = false;
                if (udiv > 0) {                                   // (uniform)
                    const int uth = minS > 0 ? sg_div_small(100 * minS - 1, udiv) : -1;
#pragma unroll
                    for (int k = 0; k < K; k++) bad |= Sv[k] <= uth && (unsigned)(li * K + k - bestDisp + 1) > 2u;
                } else {
#pragma unroll
                    for (int k = 0; k < K; k++) bad |= Sv[k] * udiv < minS * 100 && (unsigned)(li * K + k - bestDisp + 1) > 2u;
                }
                const unsigned long long bal = __ballot(bad);
                const bool rejected = ((bal >> (threadIdx.x & 48)) & 0xFFFFull) != 0;       // any lane of my 16-lane group
                if (live && !rejected && li == 0) {
                    int d = bestDisp;
                    const int x2 = x + minX1 - d - minD;
                    if (minS < SG_MAXC) atomicMin(&disp2key[(size_t)y * w + x2], ((unsigned)minS << 16) | (unsigned)(65535 - x));
                    if (0 < d && d < D - 1) {
                        const int sm = srow[d - 1], s0 = srow[d], sp = srow[d + 1];
                        const int denom2 = max(sm + sp - 2 * s0, 1);
                        d = d * SG_DISP_SCALE + sg_div_small((sm - sp) * SG_DISP_SCALE + denom2, denom2 * 2);
                    } else d *= SG_DISP_SCALE;
                    disp1[(size_t)y * w + x + minX1] = (int16_t)(d + minD * SG_DISP_SCALE);
                }
                __builtin_amdgcn_wave_barrier();               // (the next pixel overwrites srow: LDS operations of a wave execute in order)
            }
            __syncthreads();
            if (s_fail[y & 1]) stop = true;                       // a hand-off timed out: every wave leaves at the same row
        }
    };
    load_rows(Va, 0);
    for (int y0 = 0; y0 < h && !stop; y0 += 2 * SGS_UN) {
        load_rows(Vb, y0 + SGS_UN);
        run_rows(Va, y0);
        if (y0 + SGS_UN >= h || stop) break;
        load_rows(Va, y0 + 2 * SGS_UN);
        run_rows(Vb, y0 + SGS_UN);
    }
}
// ------------------------------------------------------------------ the sweep with EIGHT lanes per pixel (default)
// sgbm_sweep gives a pixel 16 lanes (one DPP row) with D / 16 disparities each -- 45 instructions per recurrence step for the 4 pixels of a wave at D = 80.
// With 8 lanes per pixel a lane holds D / 8 disparities = K packed pairs (never an odd count: no pad slot) and a wave 8 pixels: 14 + 10 K - 1 = 63
// instructions per step for twice the pixels (the d +- 1 neighbours cross lanes by row_shr / row_shl with the lane at a group's edge repaired by a select,
// min over d by quad_perm x 2 + row_half_mirror = three steps instead of four), and the fixed part of the winner pass (selection, uniqueness vote, division,
// atomics: per wave instruction, whatever the number of pixels) is paid once per 8 pixels.  A block is 128 groups = 128 columns (CPG = 1), every column's
// diagonal predecessors come from the neighbouring groups through LDS; the strip hand-off (mailboxes, tickets, publish-before-poll) is sgbm_sweep's.
template <int K>
__device__ __forceinline__ void sg_step8(uint32_t (&L)[K], int& minPrev, const uint32_t (&Cp)[K], uint32_t P1P1, int P2, bool first, bool last)
{
    constexpr uint32_t MAXMAX = (uint32_t)SG_MAXC | ((uint32_t)SG_MAXC << 16);
    // L + P1 once per pair (round 6): the d - 1 / d + 1 neighbours are byte-aligned out of these sums (v_pk_add_u16 works on the halves separately, so adding before
    // or after the alignment is the same bits); the group's edges see MAX_COST + P1
    const uint32_t MAXP = pk_add16(MAXMAX, P1P1);
    const uint32_t dd = (uint32_t)(minPrev + P2) * 0x00010001u;
    uint32_t Ln[K], LP[K], m = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < K; j++) LP[j] = pk_add16(L[j], P1P1);
    uint32_t lft = (uint32_t)sg_dpp<0x111>((int)MAXP, (int)LP[K - 1]);           // row_shr:1: the left lane's last pair
    uint32_t rgt = (uint32_t)sg_dpp<0x101>((int)MAXP, (int)LP[0]);               // row_shl:1: the right lane's first pair
    lft = first ? MAXP : lft;                                                     // lane 0 of the GROUP (lane 8 of the DPP row would see the other group's lane 7)
    rgt = last ? MAXP : rgt;
#pragma unroll
    for (int j = 0; j < K; j++) {
        const uint32_t lm = __builtin_amdgcn_alignbyte(LP[j], j > 0 ? LP[j - 1] : lft, 2);           // (slot 2j-1, slot 2j) + P1
        const uint32_t lp = __builtin_amdgcn_alignbyte(j < K - 1 ? LP[j + 1] : rgt, LP[j], 2);       // (slot 2j+1, slot 2j+2) + P1
        const uint32_t t = pk_min16(pk_min16(L[j], lm), pk_min16(lp, dd));
        Ln[j] = pk_sub16(pk_add16(Cp[j], t), dd);
        m = pk_min16(m, Ln[j]);
    }
    int v = (int)min(m & 0xFFFFu, m >> 16);
    v = min(v, sg_dpp_all<0xB1>(v));      // quad_perm [1,0,3,2]
    v = min(v, sg_dpp_all<0x4E>(v));      // quad_perm [2,3,0,1]
    v = min(v, sg_dpp_all<0x141>(v));     // row_half_mirror: lane i <-> 7 - i of its 8-lane half
    minPrev = v;
#pragma unroll
    for (int j = 0; j < K; j++) L[j] = Ln[j];
}
__device__ __forceinline__ int sg_min8(int v)
{
    v = min(v, sg_dpp_all<0xB1>(v)); v = min(v, sg_dpp_all<0x4E>(v)); v = min(v, sg_dpp_all<0x141>(v));
    return v;
}
template <int K> __device__ __forceinline__ void sg_load8(uint32_t (&d)[K], const uint16_t* p) { __builtin_memcpy(d, p, 4 * K); }      // 2 K u16, 4-byte aligned
// exchange slots of a column: NG x 8 dwords, padded to SGS8_XSTRIDE so that the eight columns of a wave fall into different LDS banks (48 dwords: columns g and g + 4 collide)
#define SGS8_XPAD 8
#ifdef SGS8_PROBE
// a clock read no instruction is scheduled across (the first probes were moved behind the row's first wait by the scheduler and under-counted the row)
__device__ __forceinline__ unsigned long long sgs8_clock() { __builtin_amdgcn_sched_barrier(0); const unsigned long long t = clock64(); __builtin_amdgcn_sched_barrier(0); return t; }
__device__ unsigned long long g_sgs8_probe[8]; __device__ unsigned long long g_sgs8_phase[3][12];   // [class: inner wave 5, mailbox-edge wave direct row, slow row][0 rows, 1.. cumulative cycles at the phase marks]
//    // [0] rows of inner wave, [1] its cycles, [2] its barrier cycles, [3] rows of edge waves, [4] cycles, [5] barrier cycles, [6] slow-path rows, [7] poll cycles
#endif
template <int K, bool FAST>
__global__ void __launch_bounds__(1024)
sgbm_sweep8(const uint16_t* __restrict__ C_all, const uint16_t* __restrict__ S_all, int w, int w1, int h, int P1, int P2, int minD, int minX1, int uniquenessRatio,
            int NS, int TX, int16_t* __restrict__ disp1, unsigned* __restrict__ disp2key, unsigned* flags /* [0] ticket counter, [1] time-out */, sg_u64* mbox_all, int* fail_out)
{
    constexpr int D = 16 * K, NG = K + 1;                      // a lane: 2 K disparities = K pairs; a message: K pairs + the path minimum
    extern __shared__ __align__(16) uint8_t sw_smem[];
    __shared__ unsigned s_ticket; __shared__ int s_fail[2];
    __shared__ __align__(32) uint32_t s_keep[(2 * K + 4) * 8];          // the uniqueness test's masks: row rel + 3, dword j = pair j (see the winner pass)
    const int ng = blockDim.x >> 3, g = threadIdx.x >> 3, li = threadIdx.x & 7;
    for (int i = threadIdx.x; i < (2 * K + 4) * 8; i += blockDim.x) {
        const int rel = (i >> 3) - 3, j = i & 7;
        const uint32_t excl = rel >= 0 ? (7u << rel) : (7u >> -rel);  // slots rel .. rel + 2
        s_keep[i] = ((excl >> (2 * j)) & 1u ? 0u : 0x8000u) | ((excl >> (2 * j + 1)) & 1u ? 0u : 0x80000000u);
    }
    uint32_t idxr[(2 * K + 3) / 4];                                     // the lane's disparity numbers li 2K + k as bytes (the winner pass builds its keys with v_perm_b32)
#pragma unroll
    for (int q = 0; q < (2 * K + 3) / 4; q++) idxr[q] = (uint32_t)(li * 2 * K + 4 * q) * 0x01010101u + 0x03020100u;
    uint32_t* xch = reinterpret_cast<uint32_t*>(sw_smem);     // [parity 2][direction 2][ng][NG][8]: dir 0 = L1 of the column (for its right neighbour), dir 1 = L3 (for its left neighbour)
    constexpr int XS = NG * 8 + SGS8_XPAD;                     // dwords per column slot
    uint16_t* srow = reinterpret_cast<uint16_t*>(xch + (size_t)4 * ng * XS) + (size_t)g * D;
    auto xslot = [&](int par, int dir, int gg) -> uint32_t* { return xch + (size_t)((par * 2 + dir) * ng + gg) * XS + li; };
    if (threadIdx.x == 0) { s_ticket = atomicAdd(&flags[0], 1u); s_fail[0] = 0; s_fail[1] = 0; }
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {                       // "row -1": OpenCV's zeroed border
        uint32_t* p = xslot(1, dir, g);
#pragma unroll
        for (int j = 0; j < NG; j++) p[j * 8] = 0u;
    }
    __syncthreads();
    const int t = (int)s_ticket, f = t / NS, strip = t - f * NS;
    const int x = strip * TX + g, xend = min((strip + 1) * TX, w1);
    const bool live = g < TX && x < xend;
    const uint16_t* Cf = C_all + (size_t)f * w1 * h * D + li * 2 * K; const uint16_t* Sf = S_all + (size_t)f * w1 * h * D + li * 2 * K;
    disp1 += (size_t)f * w * h; disp2key += (size_t)f * w * h;
    sg_u64* mb = mbox_all + (size_t)f * (NS - 1) * 2 * SGS_SLOTS * NG * 8 + li;
    auto mslot = [&](int seam, int dir, int slot) -> sg_u64* { return mb + (size_t)(((seam * 2 + dir) * SGS_SLOTS + slot) * NG) * 8; };
    const bool edgeL = g == 0 && strip > 0, edgeR = g == TX - 1 && strip < NS - 1;
    const bool first = li == 0, last = li == 7;
    const uint32_t P1P1 = (uint32_t)P1 * 0x00010001u;
    const int udiv = 100 - uniquenessRatio;
    const unsigned umagic = udiv > 1 ? (unsigned)(0x100000000ull / (unsigned)udiv) + 1u : 0u;
    uint32_t L1[K], L2[K], L3[K]; int m1 = 0, m2 = 0, m3 = 0;
#pragma unroll
    for (int j = 0; j < K; j++) L1[j] = L2[j] = L3[j] = 0u;
    uint32_t Va[2][K], Vb[2][K];                              // [C, S04][pairs] of the row in work and of the next one
    auto load_row = [&](uint32_t (&V)[2][K], int y) {
        const size_t off = ((size_t)min(y, h - 1) * w1 + min(x, w1 - 1)) * D;
        sg_load8<K>(V[0], Cf + off); sg_load8<K>(V[1], Sf + off);
    };
    bool stop = false;
    auto mbox_wait8 = [&](const sg_u64* gq, unsigned epoch, uint32_t (&v)[NG]) -> bool {
        for (unsigned spins = 0;; ++spins) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < NG; j++) { const sg_u64 q = __hip_atomic_load(gq + j * 8, SG_RLX_AGENT); v[j] = (uint32_t)q; ok &= (unsigned)(q >> 32) == epoch; }
            if (__builtin_amdgcn_ballot_w64(!ok) == 0) return true;
            if (spins >= SGS_SPIN_LIMIT || ((spins & 255u) == 255u && __hip_atomic_load(flags + 1, SG_RLX_AGENT) != 0u)) return false;
            __builtin_amdgcn_s_sleep(4);
        }
    };
    // the mailbox granules an outermost column will need in the NEXT row are fetched ahead (a poll is a round trip to the memory side, ~1.2 k clocks, whether the
    // data is there or not): pf = the incoming message of row y - 1 -- L1 from the left strip for the lanes of column 0, L3 from the right strip for the lanes of the
    // last column (a lane is never both) -- valid when every tag says y.  ONE array: with one per side the kernel spilled registers, and a spill's reload waits
    // with vmcnt(0), i.e. for the NEXT row's cost loads as well: every row paid a trip to HBM (round 6, found with -DSGS8_PROBE)
    sg_u64 pf[NG];
#pragma unroll
    for (int j = 0; j < NG; j++) pf[j] = 0;
    const bool wave_has_edge = ((threadIdx.x & ~63) == 0) || ((int)(threadIdx.x | 63) >> 3) >= TX - 1;      // wave-uniform
    const bool wave_mbox_l = (threadIdx.x >> 6) == 0 && strip > 0, wave_mbox_r = (int)(threadIdx.x >> 6) == ((TX - 1) >> 3) && strip < NS - 1;      // it holds edgeL / edgeR lanes
    // the waves that hold a strip's outermost columns do more per row (publish, poll / prefetch) and everybody waits for them at the row's barrier: they issue first
    if (wave_has_edge) __builtin_amdgcn_s_setprio(3);
    // the neighbouring columns' states of the previous row (LDS)
    uint32_t nl[NG], nr[NG];
#pragma unroll
    for (int j = 0; j < NG; j++) nl[j] = nr[j] = 0u;              // "row -1": OpenCV's zeroed border
#ifdef SGS8_PROBE
    unsigned long long pr_rows = 0, pr_cyc = 0, pr_bar = 0, pr_slow = 0, pr_poll = 0, pr_ph[3][12] = {};
#endif
    auto run_row = [&](uint32_t (&V)[2][K], int y) {
#ifdef SGS8_PROBE
        const unsigned long long pr_t0 = sgs8_clock(); unsigned long long pr_m[9] = {};
#define PR_MARK(i_) pr_m[i_] = sgs8_clock() - pr_t0
#else
#define PR_MARK(i_)
#endif
        // ---- the neighbouring columns' states of row y - 1 (LDS; the strip's outermost columns: the mailbox or the zeroed border, below)
        const int pp = (y + 1) & 1;
        {   const uint32_t* p = xslot(pp, 0, g > 0 ? g - 1 : 0);
#pragma unroll
            for (int j = 0; j < NG; j++) nl[j] = p[j * 8];
            const uint32_t* q = xslot(pp, 1, g < ng - 1 ? g + 1 : g);
#pragma unroll
            for (int j = 0; j < NG; j++) nr[j] = q[j * 8];
        }
        const bool outL = g == 0, outR = g >= TX - 1;
        // the vertical direction needs no neighbour: it runs first, in front of the first use of the prefetched mailbox words (their loads were issued at the end of
        // the previous row: SGS8_PF_POS)
        sg_step8<K>(L2, m2, V[0], P1P1, P2, first, last);
        // ---- L1 and L3.  A strip's outermost column takes the predecessor of ONE direction from outside the strip: the mailbox of the neighbour strip (its message of
        // row y - 1 usually sits in the prefetch registers) or the image's zeroed border.  What a strip hands on never depends on what it receives in the same row
        // (L1 flows right, L3 flows left), so a wave with a mailbox edge steps its OUTGOING direction first and publishes it at once -- the neighbour strip has
        // almost a whole row to see it -- and only then looks at the incoming message (round 6; before, both directions were stepped, published behind the second
        // one, and stepped again whenever the prefetch had missed).
        bool direct = true; (void)direct;                         // (read by the -DSGS8_PROBE build only)
        if (wave_has_edge) {
            if (outL && !(edgeL && y > 0)) {
#pragma unroll
                for (int j = 0; j < NG; j++) nl[j] = 0u;
            }
            if (outR && !(edgeR && y > 0)) {
#pragma unroll
                for (int j = 0; j < NG; j++) nr[j] = 0u;
            }
        }
        const bool wL = wave_mbox_l && y > 0, wR = wave_mbox_r && y > 0;      // (wave-uniform) lanes of this wave take a predecessor from a mailbox
        auto step_l1 = [&]() {
#pragma unroll
            for (int j = 0; j < K; j++) L1[j] = nl[j];
            m1 = (int)nl[K];
            sg_step8<K>(L1, m1, V[0], P1P1, P2, first, last);
        };
        auto step_l3 = [&]() {
#pragma unroll
            for (int j = 0; j < K; j++) L3[j] = nr[j];
            m3 = (int)nr[K];
            sg_step8<K>(L3, m3, V[0], P1P1, P2, first, last);
        };
        auto publish = [&](bool mine, const uint32_t (&L)[K], int mn, sg_u64* o) {
            if (mine) {
#pragma unroll
                for (int j = 0; j < K; j++) __hip_atomic_store(o + j * 8, ((sg_u64)(unsigned)(y + 1) << 32) | L[j], SG_RLX_AGENT);
                __hip_atomic_store(o + K * 8, ((sg_u64)(unsigned)(y + 1) << 32) | (uint32_t)mn, SG_RLX_AGENT);
            }
        };
        auto pub_l1 = [&]() { publish(edgeR, L1, m1, mslot(strip, 0, y & (SGS_SLOTS - 1))); };            // its L1 came from LDS: final
        auto pub_l3 = [&]() { publish(edgeL, L3, m3, mslot(strip - 1, 1, y & (SGS_SLOTS - 1))); };
        // the incoming message of row y - 1 for the lanes `mine`: the prefetched words when every tag says y, otherwise poll (bounded; a time-out stops the block)
        auto take = [&](bool mine, uint32_t (&n)[NG], const sg_u64* slot) {
            bool have = true;
            if (mine) {
#pragma unroll
                for (int j = 0; j < NG; j++) have &= (unsigned)(pf[j] >> 32) == (unsigned)y;
            }
            if (__builtin_amdgcn_ballot_w64(mine && !have) == 0) {
                if (mine) {
#pragma unroll
                    for (int j = 0; j < NG; j++) n[j] = (uint32_t)pf[j];
                }
            } else {
                direct = false;
#ifdef SGS8_PROBE
                const unsigned long long pr_p0 = sgs8_clock();
#endif
                bool ok = true;
                if (mine) ok = mbox_wait8(slot, (unsigned)y, n);
#ifdef SGS8_PROBE
                pr_poll += sgs8_clock() - pr_p0;
#endif
                if (!ok) { s_fail[y & 1] = 1; __hip_atomic_store(flags + 1, 1u, SG_RLX_AGENT); if (fail_out) atomicOr(fail_out, 1); }
            }
        };
        auto take_l = [&]() { take(edgeL, nl, mslot(strip - 1, 0, (y - 1) & (SGS_SLOTS - 1))); };
        auto take_r = [&]() { take(edgeR, nr, mslot(strip, 1, (y - 1) & (SGS_SLOTS - 1))); };
        PR_MARK(0);
        if (wL && wR) {                                           // a strip of one wave (tests): both directions with placeholders, publish, then the real ones
            step_l1(); step_l3(); pub_l1(); pub_l3(); take_l(); take_r(); step_l1(); step_l3();
        } else if (wR) { step_l1(); PR_MARK(6); pub_l1(); PR_MARK(7); take_r(); PR_MARK(8); step_l3(); pub_l3(); }
        else if (wL) { step_l3(); PR_MARK(6); pub_l3(); PR_MARK(7); take_l(); PR_MARK(8); step_l1(); pub_l1(); }
        else { step_l1(); step_l3(); pub_l1(); pub_l3(); }        // inner waves, the image's border, row 0
        PR_MARK(1);
#ifdef SGS8_PROBE
        if (!direct) pr_slow++;
#endif
        // the neighbours' messages of THIS row (for row y + 1)
        auto prefetch_mbox = [&]() {
        if (wave_has_edge && y + 1 < h) {
            if (edgeL || edgeR) {
                const sg_u64* gq = edgeL ? mslot(strip - 1, 0, y & (SGS_SLOTS - 1)) : mslot(strip, 1, y & (SGS_SLOTS - 1));
#pragma unroll
                for (int j = 0; j < NG; j++) pf[j] = __hip_atomic_load(gq + j * 8, SG_RLX_AGENT);
            }
        }
        };
        PR_MARK(2);
        if (!live) {                                              // a column outside the strip / image: its neighbours see the zeroed border
#pragma unroll
            for (int j = 0; j < K; j++) L1[j] = L2[j] = L3[j] = 0u;
            m1 = m2 = m3 = 0;
        }
        {   uint32_t* p = xslot(y & 1, 0, g);
#pragma unroll
            for (int j = 0; j < K; j++) p[j * 8] = L1[j];
            p[K * 8] = (uint32_t)m1;
            uint32_t* q = xslot(y & 1, 1, g);
#pragma unroll
            for (int j = 0; j < K; j++) q[j * 8] = L3[j];
            q[K * 8] = (uint32_t)m3;
        }
        PR_MARK(3);
        // ---- the winner pass of the group's pixel: S = min(32767, S04 + L1 + L2 + L3)
        {
            uint32_t sp2[K];
#pragma unroll
            for (int j = 0; j < K; j++) {
                if (FAST) sp2[j] = pk_addsat_i15(pk_addsat_i15(L1[j], L2[j]), pk_addsat_i15(L3[j], V[1][j]));      // every L < 2^15, S04 <= 32767: three clamped adds (six instructions before)
                else sp2[j] = pk_addsat15(pk_addsat15(pk_addsat15(L1[j], L2[j]), L3[j]), V[1][j]);
            }
            // the key (S << 8 | disparity) of a value is ONE v_perm_b32: bytes 1 - 2 from the pair, byte 0 from the lane's table of disparity numbers (round 6; before:
            // mask / shift + v_lshl_or per value)
            int best = INT_MAX;
#pragma unroll
            for (int j = 0; j < K; j++) {
                const int klo = (int)__builtin_amdgcn_perm(sp2[j], idxr[(2 * j) >> 2], 0x0C050400u | (uint32_t)((2 * j) & 3));
                const int khi = (int)__builtin_amdgcn_perm(sp2[j], idxr[(2 * j + 1) >> 2], 0x0C070600u | (uint32_t)((2 * j + 1) & 3));
                best = min(best, min(klo, khi));
            }
#pragma unroll
            for (int j = 0; j < K; j++) reinterpret_cast<uint32_t*>(srow)[li * K + j] = sp2[j];       // (every value <= 32767: the pairs are the u16 table as it stands)
            best = sg_min8(best);
            const int minS = best >> 8, bestDisp = best & 255;
            bool bad = false;
            if (udiv > 0) {                                       // (uniform)
                // (100 minS - 1) / udiv by a multiply-high: n < 2^22 and udiv <= 100, so floor(n m / 2^32) with m = floor(2^32 / udiv) + 1 is the exact quotient
                // (n = q d + r: the product's excess over q is (r + n e / 2^32) / d with e <= d, and n e < 2^32); udiv = 1 has no 32-bit m
                const int un = 100 * minS - 1;
                const int uth = minS > 0 ? (udiv == 1 ? un : (int)__umulhi((unsigned)un, umagic)) : -1;
                // "S <= uth" for a pair at once: S - (uth + 1) borrows into bit 15 of its half exactly when S <= uth (S <= 32767; uth + 1 clamped to 32768 keeps the
                // subtraction inside 16 bits); the three disparities around the winner are masked out by a per-lane bit pattern: 2 + 2 instructions per pair instead of 5 per value
                const uint32_t U2 = (uint32_t)min(uth + 1, 32768) * 0x00010001u;
                const int rel = min(max(bestDisp - 1 - li * 2 * K, -3), 2 * K);     // this lane's slot of disparity bestDisp - 1 (clamped: outside -2 .. 2 K - 1 nothing of the triple is this lane's)
                // the K masks of the lane (bit 15 / 31 of a pair unless the slot is one of rel .. rel + 2) come from a table in LDS, one 32-byte row per value of rel
                // (round 6: two ds_read instead of five shift / mask instructions per pair)
                const uint32_t* kp = s_keep + (rel + 3) * 8;
                uint32_t acc = 0u;
#pragma unroll
                for (int j = 0; j < K; j++) acc |= pk_sub16(sp2[j], U2) & kp[j];
                bad = acc != 0u;
            } else {
#pragma unroll
                for (int k = 0; k < 2 * K; k++) bad |= (int)((sp2[k >> 1] >> (16 * (k & 1))) & 0xFFFFu) * udiv < minS * 100 && (unsigned)(li * 2 * K + k - bestDisp + 1) > 2u;
            }
            const unsigned long long bal = __ballot(bad);
            const bool rejected = ((bal >> (threadIdx.x & 56)) & 0xFFull) != 0;          // any lane of my 8-lane group
            if (live && !rejected && li == 0) {
                int d = bestDisp;
                const int x2 = x + minX1 - d - minD;
                if (minS < SG_MAXC) atomicMin(&disp2key[(size_t)y * w + x2], ((unsigned)minS << 16) | (unsigned)(65535 - x));
                if (0 < d && d < D - 1) {
                    const int sm = srow[d - 1], s0 = srow[d], sp = srow[d + 1];
                    const int denom2 = max(sm + sp - 2 * s0, 1);
                    d = d * SG_DISP_SCALE + sg_div_small((sm - sp) * SG_DISP_SCALE + denom2, denom2 * 2);
                } else d *= SG_DISP_SCALE;
                disp1[(size_t)y * w + x + minX1] = (int16_t)(d + minD * SG_DISP_SCALE);
            }
        }
        // The end of the row: the mailbox loads for row y + 1, then the costs of row y + 2 into the registers this row has finished with.  (Measured in round 6,
        // stage time of the bench's serial pass: the mailbox loads behind the own publish / in the winner pass / here, the cost loads in the winner pass / here:
        // 0.1607 - 0.1617 ms per pair for five of the six combinations, 0.1667 for mailbox loads right behind the publish with the cost loads here.)
        prefetch_mbox();
        load_row(V, y + 2);
#ifdef SGS8_PROBE
        const unsigned long long pr_b0 = sgs8_clock(); PR_MARK(4);
#endif
        __syncthreads();
#ifdef SGS8_PROBE
        { const unsigned long long pr_t1 = sgs8_clock(); pr_bar += pr_t1 - pr_b0; pr_cyc += pr_t1 - pr_t0; pr_rows++; PR_MARK(5);
          const int wvp = threadIdx.x >> 6; const bool mb_edge = (wvp == 0 && strip > 0) || (wvp == (int)((TX - 1) >> 3) && strip < NS - 1);
          const int cls = wvp == 5 ? 0 : mb_edge ? (direct ? 1 : 2) : -1;
          if (cls >= 0 && y > 0) { pr_ph[cls][0]++; for (int i = 0; i < 9; i++) pr_ph[cls][1 + i] += pr_m[i]; } }
#endif
        if (s_fail[y & 1]) stop = true;                           // a hand-off timed out: every wave leaves at the same row
    };
    load_row(Va, 0); load_row(Vb, 1);
    // The first two rows are waited for HERE.  Otherwise the scheduler is free to issue Va's loads behind Vb's, the wait-count pass merges "Va is the youngest load" from this entry edge into the
    // loop header, and every second row starts with vmcnt(0) -- which for an outermost wave means the mailbox loads it issued a moment ago.
    // (The wait is made by USING the loaded words -- an empty asm with every register as an input: a bare s_waitcnt builtin is hoisted above the loads.)
#pragma unroll
    for (int j = 0; j < K; j++) asm volatile("" :: "v"(Va[0][j]), "v"(Va[1][j]), "v"(Vb[0][j]), "v"(Vb[1][j]));
    for (int y = 0; y < h && !stop; y += 2) {
        run_row(Va, y);
        if (y + 1 >= h || stop) break;
        run_row(Vb, y + 1);
    }
#ifdef SGS8_PROBE
    if ((threadIdx.x & 63) == 0) {
        for (int c = 0; c < 3; c++) for (int i = 0; i < 10; i++) if (pr_ph[c][i]) atomicAdd(&g_sgs8_phase[c][i], pr_ph[c][i]);
        const int wv = threadIdx.x >> 6;
        if (wv == 5) { atomicAdd(&g_sgs8_probe[0], pr_rows); atomicAdd(&g_sgs8_probe[1], pr_cyc); atomicAdd(&g_sgs8_probe[2], pr_bar); }
        if (wave_has_edge && (edgeL || edgeR || true)) { atomicAdd(&g_sgs8_probe[3], pr_rows); atomicAdd(&g_sgs8_probe[4], pr_cyc); atomicAdd(&g_sgs8_probe[5], pr_bar); atomicAdd(&g_sgs8_probe[6], pr_slow); atomicAdd(&g_sgs8_probe[7], pr_poll); }
    }
#endif
}
// ------------------------------------------------------------------ winner-takes-all, one pixel per 16 lanes, all pixels in parallel
// S(p, d) = min(32767, sum of the five L_r) (all terms >= 0: equal to OpenCV's two saturating steps).  disp2 (the right-image
// disparity table OpenCV fills while walking x from right to left, replacing an entry only by a strictly smaller cost) becomes
// an atomicMin on the key (cost << 16 | 65535 - x): smallest cost, then the larger x, i.e. the entry the walk would have kept.
// A 16-lane group takes WTA_PX consecutive pixels; D / 8 of its lanes are active, each with EIGHT disparities = one aligned 16-byte load per volume (a pixel's
// D costs are D / 8 such words; the K = D / 16 u16 per lane of the path kernels would be 10-byte loads at 2-byte alignment for D = 80).  The 5 x WTA_PX loads of a
// lane are issued together, then the pixels are reduced one after the other.
#define WTA_PX 4
template <int K>
__global__ void __launch_bounds__(256)
sgbm_wta(const uint16_t* __restrict__ L0, const uint16_t* __restrict__ L1, const uint16_t* __restrict__ L2, const uint16_t* __restrict__ L3,
         const uint16_t* __restrict__ L4, int w, int w1, int h, int minD, int minX1, int uniquenessRatio, int16_t* __restrict__ disp1, unsigned* __restrict__ disp2key)
{
    constexpr int D = 16 * K, NL = D / 8;                     // NL active lanes of a group
    {   const size_t fv = (size_t)blockIdx.y * w1 * h * D, fp = (size_t)blockIdx.y * w * h;      // blockIdx.y = frame
        L0 += fv; L1 += fv; L2 += fv; L3 += fv; L4 += fv; disp1 += fp; disp2key += fp; }
    __shared__ uint16_t srow[16][D];                          // S of the group's pixel, for the three sub-pixel taps
    const int gl = threadIdx.x >> 4, li = threadIdx.x & 15;
    const bool act = li < NL;
    const int lc = act ? li : 0;
    const long long npix = (long long)w1 * h;
    const long long p0 = ((long long)blockIdx.x * 16 + gl) * WTA_PX;
    uint4 v[WTA_PX][5];
#pragma unroll
    for (int q = 0; q < WTA_PX; q++) {
        const long long pid = p0 + q < npix ? p0 + q : npix - 1;
        const size_t base = (size_t)pid * D + lc * 8;
        v[q][0] = *reinterpret_cast<const uint4*>(L0 + base); v[q][1] = *reinterpret_cast<const uint4*>(L1 + base); v[q][2] = *reinterpret_cast<const uint4*>(L2 + base);
        v[q][3] = *reinterpret_cast<const uint4*>(L3 + base); v[q][4] = *reinterpret_cast<const uint4*>(L4 + base);
    }
#pragma unroll
    for (int q = 0; q < WTA_PX; q++) {
        const bool live = p0 + q < npix;
        const long long pid = live ? p0 + q : npix - 1;
        const int y = (int)(pid / w1), x = (int)(pid - (long long)y * w1);
        int Sv[8], best = INT_MAX;                             // (S << 8 | d): smallest S, then smallest d ("Sval < minS" scanning d upwards)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int sh = 16 * (k & 1), j = k >> 1;
            auto word = [&](const uint4& u) { return j == 0 ? u.x : j == 1 ? u.y : j == 2 ? u.z : u.w; };
            Sv[k] = min((int)((word(v[q][0]) >> sh) & 0xFFFFu) + (int)((word(v[q][1]) >> sh) & 0xFFFFu) + (int)((word(v[q][2]) >> sh) & 0xFFFFu) +
                        (int)((word(v[q][3]) >> sh) & 0xFFFFu) + (int)((word(v[q][4]) >> sh) & 0xFFFFu), SG_MAXC);
            if (act) { best = min(best, (Sv[k] << 8) | (li * 8 + k)); srow[gl][li * 8 + k] = (uint16_t)Sv[k]; }
        }
        best = sg_rowmin(best);
        const int minS = best >> 8, bestDisp = best & 255;
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 8; k++) bad |= Sv[k] * (100 - uniquenessRatio) < minS * 100 && abs(bestDisp - (li * 8 + k)) > 1;
        const unsigned long long bal = __ballot(bad && act);
        const bool rejected = ((bal >> (threadIdx.x & 48)) & 0xFFFFull) != 0;           // any lane of my 16-lane group
        if (live && !rejected && li == 0) {
            int d = bestDisp;
            const int x2 = x + minX1 - d - minD;
            if (minS < SG_MAXC) atomicMin(&disp2key[(size_t)y * w + x2], ((unsigned)minS << 16) | (unsigned)(65535 - x));     // "disp2cost > minS" from MAX_COST
            if (0 < d && d < D - 1) {
                const int sm = srow[gl][d - 1], s0 = srow[gl][d], sp = srow[gl][d + 1];
                const int denom2 = max(sm + sp - 2 * s0, 1);
                d = d * SG_DISP_SCALE + ((sm - sp) * SG_DISP_SCALE + denom2) / (denom2 * 2);
            } else d *= SG_DISP_SCALE;
            disp1[(size_t)y * w + x + minX1] = (int16_t)(d + minD * SG_DISP_SCALE);
        }
        __builtin_amdgcn_wave_barrier();                       // (the group's next pixel overwrites srow: LDS operations of a wave execute in order)
    }
}
// left-right check: the disparity rounded down and up must both disagree with the right-image table to be dropped
__global__ void __launch_bounds__(256)
sgbm_lrcheck(const int16_t* __restrict__ disp1, const unsigned* __restrict__ disp2key, int w, int h, int w1, int minD, int minX1, int disp12MaxDiff, int16_t* __restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    { const size_t fp = (size_t)blockIdx.z * w * h; disp1 += fp; disp2key += fp; out += fp; }
    const int INVALID = (minD - 1) * SG_DISP_SCALE;
    const unsigned* k2 = disp2key + (size_t)y * w;
    auto disp2 = [&](int xx) -> int {                          // the disparity the winning pixel assigned to right-image column xx
        const unsigned key = k2[xx];
        if (key == 0xFFFFFFFFu) return INVALID;
        const int xw = 65535 - (int)(key & 0xFFFFu);
        return xw + minX1 - xx;                               // x2 = x + minX1 - d - minD  =>  d + minD = x + minX1 - x2
    };
    int v = disp1[(size_t)y * w + x];
    if (x >= minX1 && x < minX1 + w1 && v != INVALID) {
        const int _d = v >> SG_DISP_SHIFT, d_ = (v + SG_DISP_SCALE - 1) >> SG_DISP_SHIFT;
        const int _x = x - _d, x_ = x - d_;
        if (0 <= _x && _x < w && disp2(_x) >= minD && abs(disp2(_x) - _d) > disp12MaxDiff &&
            0 <= x_ && x_ < w && disp2(x_) >= minD && abs(disp2(x_) - d_) > disp12MaxDiff)
            v = INVALID;
    }
    out[(size_t)y * w + x] = (int16_t)v;
}
// ------------------------------------------------------------------ cv::medianBlur 3x3 (int16, replicate border)
__global__ void __launch_bounds__(256)
sgbm_median3(const int16_t* __restrict__ src, int w, int h, int16_t* __restrict__ dst)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    { const size_t fp = (size_t)blockIdx.z * w * h; src += fp; dst += fp; }
    int v[9];
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) v[(dy + 1) * 3 + dx + 1] = src[(size_t)min(max(y + dy, 0), h - 1) * w + min(max(x + dx, 0), w - 1)];
#define SG_CE(a, b) { const int lo_ = min(v[a], v[b]), hi_ = max(v[a], v[b]); v[a] = lo_; v[b] = hi_; }
    SG_CE(1, 2) SG_CE(4, 5) SG_CE(7, 8) SG_CE(0, 1) SG_CE(3, 4) SG_CE(6, 7) SG_CE(1, 2) SG_CE(4, 5) SG_CE(7, 8)
    SG_CE(0, 3) SG_CE(5, 8) SG_CE(4, 7) SG_CE(3, 6) SG_CE(1, 4) SG_CE(2, 5) SG_CE(4, 7) SG_CE(4, 2) SG_CE(6, 4) SG_CE(4, 2)
#undef SG_CE
    dst[(size_t)y * w + x] = (int16_t)v[4];
}
// ------------------------------------------------------------------ cv::filterSpeckles by union-find
// components: 4-neighbours, both != newVal, |difference| <= maxDiff.  label = smallest pixel index of the component.
__device__ __forceinline__ int uf_find(int* parent, int i)
{
    // parents always point to a smaller index (roots are hooked under smaller roots) and only ever decrease, so pointing a
    // traversed node at the root found -- with atomicMin, never a plain store -- keeps every entry an ancestor of its node
    int root = i;
    while (true) { const int p = __atomic_load_n(&parent[root], __ATOMIC_RELAXED); if (p == root) break; root = p; }
    while (i > root) {                                        // (a concurrent compression may already point past `root`: indices only fall)
        const int p = __atomic_load_n(&parent[i], __ATOMIC_RELAXED);
        if (p > root) atomicMin(&parent[i], root);
        i = p;
    }
    return root;
}
__device__ __forceinline__ void uf_union(int* parent, int a, int b)
{
    while (true) {
        a = uf_find(parent, a); b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }         // hook the larger root under the smaller, only while it still is a root
        if (atomicCAS(&parent[a], a, b) == a) return;          // (otherwise somebody hooked a first: find again and retry)
    }
}
// Two levels: a block first resolves its 64 x 16 tile in LDS (local forest, LDS atomics), flattens it and writes every pixel's parent = the GLOBAL index of
// its tile-local root (the smallest index of the local component: row-major order is the same inside the tile and in the image, so the invariant "parents
// point to smaller indices" holds globally); then only the pixels on tile edges are united across the edges in global memory.  The global forest sees
// ~1/13 of the unions the one-level version made, and none of the long hot chains inside large planes.
#define SPK_TW 64
#define SPK_TH 16
__device__ __forceinline__ int lds_find(int* parent, int i)
{
    int root = i;
    while (true) { const int p = parent[root]; if (p == root) break; root = p; }
    while (i > root) { const int p = parent[i]; if (p > root) atomicMin(&parent[i], root); i = p; }
    return root;
}
__device__ __forceinline__ void lds_union(int* parent, int a, int b)
{
    while (true) {
        a = lds_find(parent, a); b = lds_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        if (atomicCAS(&parent[a], a, b) == a) return;
    }
}
__global__ void __launch_bounds__(256)
sgbm_speckle_tile(const int16_t* __restrict__ img, int w, int h, int newVal, int maxDiff, int* __restrict__ parent, int* __restrict__ count)
{
    __shared__ int lp[SPK_TW * SPK_TH];
    __shared__ int lcnt[SPK_TW * SPK_TH];
    __shared__ int16_t val[SPK_TH][SPK_TW];
    { const size_t fp = (size_t)blockIdx.z * w * h; img += fp; parent += fp; count += fp; }        // blockIdx.z = frame: every frame has its own forest (indices inside the frame)
    const int tx0 = blockIdx.x * SPK_TW, ty0 = blockIdx.y * SPK_TH;
    for (int i = threadIdx.x; i < SPK_TW * SPK_TH; i += 256) {
        const int ly = i / SPK_TW, lx = i - ly * SPK_TW, gx = tx0 + lx, gy = ty0 + ly;
        val[ly][lx] = (gx < w && gy < h) ? img[(size_t)gy * w + gx] : (int16_t)newVal;
        lp[i] = i;
    }
    __syncthreads();
    // Rows first, without a single atomic: a wave owns whole tile rows (64 pixels = its lanes), a pixel is linked to its left neighbour or starts a RUN, and the
    // run's first pixel -- the highest start at or below the lane in the ballot of starts -- is every member's parent.  Then only runs are united downwards:
    // a pixel with a link to the pixel below it does the union unless the column to its left has such a link too AND both pixels continue their left
    // neighbours' runs (the same two runs: that union is the left column's).  A smooth plane is one run per row and fifteen unions per tile; the version that
    // united every pixel with its right and lower neighbour made 2 x 1024 (LDS compare-and-swap chains: 0.61 ms per 64 frame pairs).
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        static_assert(SPK_TW == 64 && SPK_TH % 4 == 0, "a wave per tile row");
        bool linkL[SPK_TH / 4]; int start[SPK_TH / 4];
#pragma unroll
        for (int k = 0; k < SPK_TH / 4; k++) {
            const int ly = wv * (SPK_TH / 4) + k;
            const int v = val[ly][lane];
            linkL[k] = lane > 0 && v != newVal && val[ly][lane - 1] != newVal && abs(v - val[ly][lane - 1]) <= maxDiff;
            const unsigned long long starts = __ballot(!linkL[k]);                       // (lane 0 always starts a run)
            start[k] = 63 - __clzll((long long)(starts & ((2ull << lane) - 1ull)));
            lp[ly * SPK_TW + lane] = ly * SPK_TW + start[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SPK_TH / 4; k++) {
            const int ly = wv * (SPK_TH / 4) + k;
            if (ly >= SPK_TH - 1) continue;                                                  // (wave-uniform)
            const int v = val[ly][lane], b = val[ly + 1][lane];
            const bool linkV = v != newVal && b != newVal && abs(v - b) <= maxDiff;
            const bool belowL = lane > 0 && b != newVal && val[ly + 1][lane - 1] != newVal && abs(b - val[ly + 1][lane - 1]) <= maxDiff;
            const unsigned long long lv = __ballot(linkV);
            const bool implied = lane > 0 && ((lv >> (lane - 1)) & 1ull) && linkL[k] && belowL;
            if (linkV && !implied) lds_union(lp, ly * SPK_TW + start[k], lp[(ly + 1) * SPK_TW + lane]);      // (the lower pixel's entry: its run's first pixel, or already an ancestor of it)
        }
    }
    __syncthreads();
    // flatten; the size of every tile-local component is counted HERE (LDS atomics) and lands on its root pixel: the global pass that follows the edge
    // unions then moves one number per tile-local root instead of chasing and counting every pixel
    int myroot[(SPK_TW * SPK_TH) / 256];
#pragma unroll
    for (int k = 0; k < (SPK_TW * SPK_TH) / 256; k++) myroot[k] = lds_find(lp, threadIdx.x + 256 * k);
    __syncthreads();
    for (int i = threadIdx.x; i < SPK_TW * SPK_TH; i += 256) lcnt[i] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (SPK_TW * SPK_TH) / 256; k++) {
        const int i = threadIdx.x + 256 * k, ly = i / SPK_TW, lx = i - ly * SPK_TW;
        if (tx0 + lx < w && ty0 + ly < h && val[ly][lx] != newVal) atomicAdd(&lcnt[myroot[k]], 1);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (SPK_TW * SPK_TH) / 256; k++) {
        const int i = threadIdx.x + 256 * k, ly = i / SPK_TW, lx = i - ly * SPK_TW, gx = tx0 + lx, gy = ty0 + ly;
        if (gx >= w || gy >= h) continue;
        const int root = myroot[k], ry = root / SPK_TW, rx = root - ry * SPK_TW;
        const int g = gy * w + gx;
        parent[g] = (ty0 + ry) * w + tx0 + rx; count[g] = root == i ? lcnt[i] : 0;          // > 0 exactly on the tile-local roots of valid components
    }
}
// the unions across tile edges: thread = one pixel of a tile's last column (links to x + 1) or last row (links to y + 1)
__global__ void __launch_bounds__(256)
sgbm_speckle_edges(const int16_t* __restrict__ img, int w, int h, int newVal, int maxDiff, int* __restrict__ parent)
{
    { const size_t fp = (size_t)blockIdx.y * w * h; img += fp; parent += fp; }
    const int ncx = (w - 1) / SPK_TW, ncy = (h - 1) / SPK_TH;          // interior vertical / horizontal edge lines
    const int nv = ncx * h, nh = ncy * w;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nv + nh) return;
    int x, y, dx, dy;
    if (t < nv) { const int e = t / h; y = t - e * h; x = (e + 1) * SPK_TW - 1; dx = 1; dy = 0; }
    else { const int u = t - nv, e = u / w; x = u - e * w; y = (e + 1) * SPK_TH - 1; dx = 0; dy = 1; }
    const int i = y * w + x, j = (y + dy) * w + x + dx;
    const int v = img[i], q = img[j];
    if (!(v != newVal && q != newVal && abs(v - q) <= maxDiff)) return;
    // Along an edge line most links repeat their neighbour's: if the previous pixel pair of the line (same two tiles) is linked too and both of this pair's
    // pixels are connected to the previous pair's inside their tiles (adjacent along the line: the tile kernel has united them), this union is implied by that
    // one.  Only the first link of every such run goes to the global forest (the long CAS / find chains on large planes came from the repeats).
    const int along = dy ? x % SPK_TW : y % SPK_TH;           // position inside the tile along the line: 0 = the previous pair belongs to other tiles
    if (along != 0) {
        const int ip = dy ? i - 1 : i - w, jp = dy ? j - 1 : j - w;
        const int vp = img[ip], qp = img[jp];
        if (vp != newVal && qp != newVal && abs(vp - qp) <= maxDiff && abs(v - vp) <= maxDiff && abs(q - qp) <= maxDiff) return;
    }
    uf_union(parent, i, j);
}
// after the edge unions: a tile-local root that was hooked under another root hands its count to the component's final root (only final roots receive, so a
// hooked root's own count never changes while it is read), and points straight at it
__global__ void __launch_bounds__(256)
sgbm_speckle_count(int n, int* __restrict__ parent, int* __restrict__ count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    { const size_t fp = (size_t)blockIdx.y * n; parent += fp; count += fp; }
    if (i >= n) return;
    const int c = count[i];
    if (c == 0) return;                                       // not a tile-local root
    const int r = uf_find(parent, i);
    if (r != i) atomicAdd(&count[r], c);
}
__global__ void __launch_bounds__(256)
sgbm_speckle_apply(int16_t* __restrict__ img, int n, int newVal, int maxSpeckleSize, const int* __restrict__ parent, const int* __restrict__ count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    { const size_t fp = (size_t)blockIdx.y * n; img += fp; parent += fp; count += fp; }
    if (i >= n || img[i] == newVal) return;
    int r = parent[i];                                        // the tile-local root, then (at most a few hops, compressed by the pass above) the final one
    while (true) { const int p = parent[r]; if (p == r) break; r = p; }
    if (count[r] <= maxSpeckleSize) img[i] = (int16_t)newVal;
}
// ------------------------------------------------------------------ disparity -> depth (rgbdframe.cpp:81-116)
__global__ void __launch_bounds__(256)
sgbm_min_kernel(const int16_t* __restrict__ disp, int n, int* __restrict__ out)
{
    disp += (size_t)blockIdx.y * n; out += blockIdx.y;
    int m = INT_MAX;
    // eight disparities per 16-byte load (a frame starts at any 2-byte boundary: unaligned loads), the tail one by one: with a 2-byte load per trip a thread
    // walked 28 dependent round trips for 56 bytes
    const int n8 = n >> 3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += gridDim.x * blockDim.x) {
        uint4 t; __builtin_memcpy(&t, disp + (size_t)8 * i, 16);
        const uint32_t wv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; k++) m = min(m, min((int)(int16_t)(wv[k] & 0xFFFFu), (int)(int16_t)(wv[k] >> 16)));
    }
    for (int i = 8 * n8 + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = min(m, (int)disp[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o, 64));
    __shared__ int wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMin(out, min(min(wm[0], wm[1]), min(wm[2], wm[3])));
}
__global__ void __launch_bounds__(256)
sgbm_depth(const int16_t* __restrict__ disp, int w, int h, const int* __restrict__ min_disp, double baseline, double cu, double cv, double f,
           double roix, double roiy, double roiz, double scale, uint16_t* __restrict__ depth)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
    if (u >= w) return;
    { const size_t fp = (size_t)blockIdx.z * w * h; disp += fp; depth += fp; min_disp += blockIdx.z; }
    const int d = disp[(size_t)v * w + u];
    uint16_t out = 0;
    if (d != 0 && d != *min_disp) {                           // |d| > FLT_EPSILON and |d - min| > FLT_EPSILON on integers
        const double pw = baseline / (1.0 * (double)d);
        const double px = (((double)u - cu) * pw) * 16.0, py = (((double)v - cv) * pw) * 16.0, pz = (f * pw) * 16.0;
        if (fabs(px) < roix && fabs(py) < roiy && fabs(pz) < roiz && pz > 0) out = (uint16_t)(pz * scale);
    }
    depth[(size_t)v * w + u] = out;
}

__global__ void __launch_bounds__(256)
sgbm_fill(int16_t* __restrict__ p, int n, int16_t v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
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
