// kernels_quad.hip -- K7/K8: the stereo quad-matcher of /root/reference/src/quadmatcher.cpp on gfx950:
//   cv::goodFeaturesToTrack (detectFeature :388-417, GFTT q=0.04 minDistance 8)    -> mineig / collect / select kernels
//   cv::calcOpticalFlowPyrLK x4 (circularMatching :548-588, win 11, 3 levels)      -> pyrdown / scharr / lk kernels
//   filteringTracks (:420-503)                                                      -> filter kernel
//   matching()/caldistance (:41-83, 525-544), the windowed brute-force Hamming NN   -> window_match kernel
// Contracts = oracle/quad.c (exact integer window sums, (value desc, raster asc) corner order).  Integer / float VALU
// work with LDS staging: no MFMA by design.
#include "ssm_internal.h"
#include <cfloat>
#include <cmath>

__device__ __forceinline__ int refl101d(int i, int n) { i = i < 0 ? -i : i; i = i >= n ? 2 * n - 2 - i : i; return min(max(i, 0), n - 1); }
__device__ __forceinline__ int f2ordq(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }

// Every kernel below takes a FRAME dimension (blockIdx.y or .z): the batched stereo path (ssm_stereo_seq_process) runs them over the nb frame pairs of a
// sub-batch in one launch, the per-pair entry points (ssm_quad_track, ssm_gftt, ssm_lk_track) are the same launches with nb = 1.
// Images live in SLOTS: slot s of side 0 (left) / side 1 (right) holds the 4-level LK pyramid of one frame (level l at byte off[l], packed rows) and,
// in `der`, its Scharr derivatives at the same element offsets.  Slot 0 is the carried previous frame, slot 1 + f is frame f of the sub-batch.
__device__ __forceinline__ const uint8_t* q_img(const QuadBatch& q, int side, int slot, int level) { return q.pyr + (size_t)(side * q.B1 + slot) * q.slot_elems + q.off[level]; }
__device__ __forceinline__ const short2* q_der(const QuadBatch& q, int side, int slot, int level) { return reinterpret_cast<const short2*>(q.der) + (size_t)(side * q.B1 + slot) * q.slot_elems + q.off[level]; }

// ------------------------------------------------------------------ cornerMinEigenVal(block 3, ksize 3) on exact integers
#define ME_W 64
#define ME_H 16
__global__ void __launch_bounds__(256)
mineig_kernel(QuadBatch q, float* __restrict__ eig_all, int* __restrict__ maxord_all)
{
    __shared__ __attribute__((aligned(16))) uint8_t px[ME_H + 4][ME_W + 4];
    __shared__ __attribute__((aligned(16))) int16_t dx[ME_H + 2][ME_W + 2], dy[ME_H + 2][ME_W + 2];
    __shared__ int smax;
    const int w = q.w[0], h = q.h[0], stride = w, f = blockIdx.y;
    const uint8_t* img = q_img(q, 0, 1 + f, 0);
    float* eig = eig_all + (size_t)f * w * h; int* maxord = maxord_all + f;
    const int tiles_x = (w + ME_W - 1) / ME_W;
    const int tx0 = (blockIdx.x % tiles_x) * ME_W, ty0 = (blockIdx.x / tiles_x) * ME_H;
    if (threadIdx.x == 0) smax = (int)0x80000000;
    // a tile whose staged pixels all lie inside the image (five tiles of six at 1241 x 376) needs no reflection anywhere: the staged pixel of (ly, lx) is the
    // image pixel, a derivative's neighbours are its LDS neighbours.  (Block-uniform; the border tiles keep the general path: six reflections per derivative.)
    const bool interior = tx0 >= 2 && ty0 >= 2 && tx0 + ME_W + 2 <= w && ty0 + ME_H + 2 <= h;
    if (interior) {
        const uint8_t* base = img + (size_t)(ty0 - 2) * stride + (tx0 - 2);
        for (int i = threadIdx.x; i < (ME_H + 4) * ((ME_W + 4) / 4); i += 256) {           // 68 = 17 dwords per staged row (any alignment: unaligned loads)
            const int ly = i / ((ME_W + 4) / 4), q4 = i - ly * ((ME_W + 4) / 4);
            uint32_t t; __builtin_memcpy(&t, base + (size_t)ly * stride + 4 * q4, 4);
            *reinterpret_cast<uint32_t*>(&px[ly][4 * q4]) = t;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < (ME_H + 2) * (ME_W + 2); i += 256) {
            const int ly = i / (ME_W + 2), lx = i - ly * (ME_W + 2);                        // derivative position (ty0 + ly - 1, tx0 + lx - 1) = staged (ly + 1, lx + 1)
            const int vx = (px[ly][lx + 2] - px[ly][lx]) + 2 * (px[ly + 1][lx + 2] - px[ly + 1][lx]) + (px[ly + 2][lx + 2] - px[ly + 2][lx]);
            const int vy = (px[ly + 2][lx] - px[ly][lx]) + 2 * (px[ly + 2][lx + 1] - px[ly][lx + 1]) + (px[ly + 2][lx + 2] - px[ly][lx + 2]);
            dx[ly][lx] = (int16_t)vx; dy[ly][lx] = (int16_t)vy;
        }
    } else {
    for (int i = threadIdx.x; i < (ME_H + 4) * (ME_W + 4); i += 256) {
        const int ly = i / (ME_W + 4), lx = i - ly * (ME_W + 4);
        px[ly][lx] = img[(size_t)refl101d(ty0 + ly - 2, h) * stride + refl101d(tx0 + lx - 2, w)];
    }
    __syncthreads();
    // derivative maps on tile + 1; positions outside the image hold the REFLECT_101 neighbour's derivative (boxFilter border)
    for (int i = threadIdx.x; i < (ME_H + 2) * (ME_W + 2); i += 256) {
        const int ly = i / (ME_W + 2), lx = i - ly * (ME_W + 2);
        const int gx = refl101d(tx0 + lx - 1, w), gy = refl101d(ty0 + ly - 1, h);
        // where does (gx, gy) sit in the staged pixels?  its 3x3 neighbourhood must be staged: true for positions within tile+-1 after reflection
        const int sx = gx - tx0 + 2, sy = gy - ty0 + 2;
        int vx = 0, vy = 0;
        if (sx >= 1 && sx <= ME_W + 2 && sy >= 1 && sy <= ME_H + 2) {
            // the staged pixels are themselves REFLECT_101 of the image, so plain neighbours give the reflected Sobel
            const int xm = refl101d(gx - 1, w) - tx0 + 2, xp = refl101d(gx + 1, w) - tx0 + 2, ym = refl101d(gy - 1, h) - ty0 + 2, yp = refl101d(gy + 1, h) - ty0 + 2;
            vx = (px[ym][xp] - px[ym][xm]) + 2 * (px[sy][xp] - px[sy][xm]) + (px[yp][xp] - px[yp][xm]);
            vy = (px[yp][xm] - px[ym][xm]) + 2 * (px[yp][sx] - px[ym][sx]) + (px[yp][xp] - px[ym][xp]);
        }
        dx[ly][lx] = (int16_t)vx; dy[ly][lx] = (int16_t)vy;
    }
    }
    __syncthreads();
    const float s = (float)(1.0 / (255.0 * 4.0 * 3.0)), s2 = s * s;
    int lmax = (int)0x80000000;
    // four neighbouring pixels per thread: their 3 x 6 derivatives come as nine dword reads per map (the kernel ran at 0.83 of the LDS's cycles on eighteen
    // 2-byte reads per pixel)
    {
        static_assert(ME_W * ME_H == 256 * 4 && ((ME_W + 2) & 1) == 0, "one pass: 16 rows x 16 groups of four");
        const int ly = threadIdx.x >> 4, lx0 = (threadIdx.x & 15) << 2, gy = ty0 + ly;
        int a[3][6], b[3][6];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint32_t* ra = reinterpret_cast<const uint32_t*>(&dx[ly + j][lx0]); const uint32_t* rb = reinterpret_cast<const uint32_t*>(&dy[ly + j][lx0]);
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const uint32_t wa = ra[t], wb = rb[t];
                a[j][2 * t] = (int)(int16_t)(wa & 0xFFFFu); a[j][2 * t + 1] = (int)(int16_t)(wa >> 16);
                b[j][2 * t] = (int)(int16_t)(wb & 0xFFFFu); b[j][2 * t + 1] = (int)(int16_t)(wb >> 16);
            }
        }
        if (gy < h) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int gx = tx0 + lx0 + k;
                if (gx >= w) break;
                int sxx = 0, sxy = 0, syy = 0;                                   // |Sobel| <= 4 * 255: nine products stay below 2^24
#pragma unroll
                for (int j = 0; j < 3; j++)
#pragma unroll
                    for (int kk = 0; kk < 3; kk++) { const int av = a[j][k + kk], bv = b[j][k + kk]; sxx += av * av; sxy += av * bv; syy += bv * bv; }
                const float fa = (float)sxx * s2 * 0.5f, fb = (float)sxy * s2, fc = (float)syy * s2 * 0.5f;
                const float d = fa - fc;
                const float e = (fa + fc) - sqrtf(d * d + fb * fb);
                eig[(size_t)gy * w + gx] = e;
                lmax = max(lmax, f2ordq(e));
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = max(lmax, __shfl_xor(lmax, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&smax, lmax);
    __syncthreads();
    // (a block whose maximum is not above the frame's running maximum has nothing to say: the atomics on that one word are a serial chain of ~0.4 us steps,
    // 480 blocks per 1241 x 376 frame)
    if (threadIdx.x == 0 && smax > __atomic_load_n(maxord, __ATOMIC_RELAXED)) atomicMax(maxord, smax);
}
// candidates: v > thr and v equals the 3x3 max of the thresholded map, interior pixels only.  key = value bits << 32 | ~index: a larger key is a
// STRONGER corner in cv::goodFeaturesToTrack's walk (value descending, then raster index ascending)
// Round 4: a block takes 1024 consecutive pixels (a wave 256: lane l looks at pixels l, l + 64, l + 128, l + 192 of them, so each of its four ballots is
// 64 consecutive pixels = two whole words of the bit image), row / column of a pixel by a reciprocal multiplication: the thread-per-pixel version spent its
// time on three barriers, one division sequence and two atomics per 256 pixels of almost no work (0.017 VALU instructions per clock, 19 us per 1241 x 376 frame)
#define GC_PX 16      // (round 4, later: 16 pixels per thread = 4096 per block -- the block's one atomic on the frame's counter is a ~0.4 us step of a serial chain, 456 of them per 1241 x 376 frame with 1024-pixel blocks)
__global__ void __launch_bounds__(256)
gftt_collect_kernel(const float* __restrict__ eig_all, int w, int h, const int* __restrict__ maxord_all, double quality, unsigned long long* __restrict__ keys_all,
                    int* __restrict__ count_all, int cap, int* __restrict__ cand_at_all, uint32_t* __restrict__ bits_all, int bits_words, uint32_t mul_w)
{
    const int f = blockIdx.y;
    const float* eig = eig_all + (size_t)f * w * h; unsigned long long* keys = keys_all + (size_t)f * cap; int* count = count_all + f;
    int* cand_at = cand_at_all + (size_t)f * w * h;          // per-pixel candidate index + 1 (0 = none; zero on entry, gftt_finish_kernel zeroes it again)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * 4 + wv) * (64 * GC_PX) + lane;         // the lane's first pixel; + 64 j
    const int np = w * h;
    const int mo = maxord_all[f]; const float mx = __int_as_float(mo >= 0 ? mo : mo ^ 0x7FFFFFFF);
    const float thr = (float)((double)fmaxf(mx, 0.f) * quality);
    bool keep[GC_PX]; float v[GC_PX];
#pragma unroll
    for (int j = 0; j < GC_PX; j++) {
        const int i = i0 + 64 * j;
        keep[j] = false; v[j] = 0.f;
        if (i < np) {
            int y = (int)__umulhi((uint32_t)i, mul_w); int x = i - y * w;        // floor(i / w): mul_w = ceil(2^32 / w) is exact for i w < 2^32 up to one correction step
            if (x < 0) { y--; x += w; }
            if (y >= 1 && y < h - 1 && x >= 1 && x < w - 1) {
                v[j] = eig[i];
                if (v[j] > thr) {
                    float m = 0.f;
#pragma unroll
                    for (int a = -1; a <= 1; a++)
#pragma unroll
                        for (int k = -1; k <= 1; k++) { float qv = eig[i + a * w + k]; qv = qv > thr ? qv : 0.f; m = fmaxf(m, qv); }
                    keep[j] = v[j] == m;
                }
            }
        }
    }
    // one reservation per BLOCK (the list order is free: the selection compares keys, it does not walk a sorted list): waves take their offsets from an LDS
    // counter, one global atomic per block instead of one per wave on the frame's single counter
    __shared__ int s_cnt, s_base;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned long long bal[GC_PX]; int wcount = 0;
#pragma unroll
    for (int j = 0; j < GC_PX; j++) { bal[j] = __ballot(keep[j]); wcount += __popcll(bal[j]); }
    int woff = 0;
    if (wcount && lane == 0) woff = atomicAdd(&s_cnt, wcount);
    woff = __shfl(woff, 0, 64);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) s_base = atomicAdd(count, s_cnt);
    __syncthreads();
    int before = 0;                                                             // kept pixels of the wave's earlier 64-pixel groups
#pragma unroll
    for (int j = 0; j < GC_PX; j++) {
        const int i = i0 + 64 * j;
        bool listed = false;
        if (keep[j]) { const int k = s_base + woff + before + __popcll(bal[j] & ((1ull << lane) - 1ull)); if (k < cap) { keys[k] = ((unsigned long long)__float_as_uint(v[j]) << 32) | (0xFFFFFFFFu - (unsigned)i); cand_at[i] = k + 1; listed = true; } }
        before += __popcll(bal[j]);
        // the same map as one bit per pixel (bit i of the frame's bit image; a ballot covers 64 consecutive pixels = two whole words, written whether set or not):
        // gftt_deps_kernel looks at 17 x 17 pixels per candidate and ~97 % of them hold nothing
        const unsigned long long lb = __ballot(listed);
        if (lane == 0) { uint32_t* bw = bits_all + (size_t)f * bits_words + ((i0 + 64 * j) >> 5); bw[0] = (uint32_t)lb; bw[1] = (uint32_t)(lb >> 32); }
    }
}
// minDistance selection (cv::goodFeaturesToTrack: walk the corners from the strongest, keep one unless an already kept corner lies closer than
// minDistance, stop at maxCorners).  The sequential walk is equivalent to rounds of local decisions, because a corner's fate depends only on
// STRONGER corners within minDistance:
//   rejected  as soon as a kept corner is that close,
//   kept      once no stronger corner that close is still undecided (and none is kept),
// so every round decides at least the strongest undecided corner and most corners settle in the first few rounds.  States (1 undecided, 2 kept,
// 3 rejected) only move undecided -> kept / rejected and a decision never reads a weaker corner, so updating in place during a round is safe.  The
// first maxCorners kept corners in strength order are the reference's result (a kept corner never depends on weaker ones).  A neighbour's strength is
// its key (value bits, then position), so no sorted list exists anywhere.
//   gftt_deps_kernel    thread per candidate: scans the (2 rad + 1)^2 window of the per-pixel candidate map ONCE and stores the stronger candidates
//                       within minDistance (<= GFTT_DEPS; more: the count says so and the finish kernel re-scans the window); a candidate without any
//                       is kept at once (round 1)
//   gftt_round_kernel   thread per undecided candidate, GFTT_ROUNDS launches: reads the states of its few dependencies
//   gftt_finish_kernel  one block per frame: loops further rounds until nothing is pending (normally zero iterations), collects the kept keys and
//                       zeroes the candidate map for the next call
//   gftt_rank_kernel    thread per kept corner: its output slot = the number of kept corners with a larger key (counting rank, keys read as broadcasts)
#define GFTT_DEPS 32
#define GFTT_ROUNDS 12
__global__ void __launch_bounds__(256)
gftt_deps_kernel(int w, int h, const unsigned long long* __restrict__ keys_all, const int* __restrict__ count_all, int cap, float min_distance,
                 const int* __restrict__ cand_at_all, const uint32_t* __restrict__ bits_all, int bits_words, uint32_t* __restrict__ deps_all,
                 uint8_t* __restrict__ depn_all, uint8_t* __restrict__ state_all)
{
    const int f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nc = min(count_all[f], cap);
    if (i >= nc) return;
    const int* cand_at = cand_at_all + (size_t)f * w * h; const unsigned long long* keys = keys_all + (size_t)f * cap;
    const uint32_t* bits = bits_all + (size_t)f * bits_words;
    uint32_t* deps = deps_all + ((size_t)f * cap + i) * GFTT_DEPS;
    const unsigned long long key = keys[i];
    const unsigned idx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu);
    const int y = idx / w, x = idx - y * w;
    const float md2 = min_distance * min_distance;
    const int rad = (int)ceilf(min_distance);                 // |dx|, |dy| < minDistance
    int n = 0;
    if (rad <= 15) {
        // a row of the window is <= 31 consecutive bits of the frame's bit image (two words); pixels outside the row or the disc are masked off, the set bits
        // are visited in ascending dx: the same candidates in the same order as the full scan below
        for (int dy = -rad; dy <= rad; dy++) {
            const int yy = y + dy;
            if (yy < 0 || yy >= h) continue;
            const int x0 = x - rad, base = yy * w + x0;           // (may be negative at the first pixels of the image: those bits are masked)
            const int wq = base >> 5, sh = base & 31;
            unsigned long long w01;                                // the two words as ONE 8-byte access (the kernel spends 0.57 of its time in the texture-address units)
            if (wq >= 0 && wq + 1 < bits_words) __builtin_memcpy(&w01, bits + wq, 8);
            else { const uint32_t w0 = wq >= 0 ? bits[wq] : 0u, w1 = wq + 1 >= 0 && wq + 1 < bits_words ? bits[wq + 1] : 0u; w01 = ((unsigned long long)w1 << 32) | w0; }
            uint32_t m = (uint32_t)(w01 >> sh);
            m &= (2 * rad + 1 >= 32) ? 0xFFFFFFFFu : ((1u << (2 * rad + 1)) - 1u);
            while (m) {
                const int b = __ffs((int)m) - 1; m &= m - 1;
                const int dx = b - rad, xx = x + dx;
                if (xx < 0 || xx >= w || (dx == 0 && dy == 0)) continue;
                if ((float)dx * (float)dx + (float)dy * (float)dy >= md2) continue;
                const int j = cand_at[yy * w + xx] - 1;
                if (j < 0 || keys[j] < key) continue;         // (j < 0 cannot happen: the bit says there is one) a weaker one
                if (n < GFTT_DEPS) deps[n] = (uint32_t)j;
                n++;
            }
        }
    } else {
        for (int dy = -rad; dy <= rad; dy++) {
            const int yy = y + dy;
            if (yy < 0 || yy >= h) continue;
            for (int dx = -rad; dx <= rad; dx++) {
                const int xx = x + dx;
                if (xx < 0 || xx >= w || (dx == 0 && dy == 0)) continue;
                if ((float)dx * (float)dx + (float)dy * (float)dy >= md2) continue;
                const int j = cand_at[yy * w + xx] - 1;
                if (j < 0 || keys[j] < key) continue;         // no candidate there, or a weaker one
                if (n < GFTT_DEPS) deps[n] = (uint32_t)j;
                n++;
            }
        }
    }
    depn_all[(size_t)f * cap + i] = (uint8_t)min(n, 255);
    state_all[(size_t)f * cap + i] = n == 0 ? 2 : 1;
}
__global__ void __launch_bounds__(256)
gftt_round_kernel(const int* __restrict__ count_all, int cap, const uint32_t* __restrict__ deps_all, const uint8_t* __restrict__ depn_all, uint8_t* __restrict__ state_all)
{
    const int f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nc = min(count_all[f], cap);
    if (i >= nc) return;
    uint8_t* state = state_all + (size_t)f * cap;
    if (__atomic_load_n(&state[i], __ATOMIC_RELAXED) != 1) return;
    const int n = depn_all[(size_t)f * cap + i];
    if (n > GFTT_DEPS) return;                               // list truncated: the finish kernel decides this one
    const uint32_t* deps = deps_all + ((size_t)f * cap + i) * GFTT_DEPS;
    bool blocked = false, rejected = false;
    for (int k = 0; k < n; k++) {
        const int sj = __atomic_load_n(&state[deps[k]], __ATOMIC_RELAXED);
        if (sj == 2) { rejected = true; break; }
        if (sj == 1) blocked = true;
    }
    if (rejected) __atomic_store_n(&state[i], (uint8_t)3, __ATOMIC_RELAXED);
    else if (!blocked) __atomic_store_n(&state[i], (uint8_t)2, __ATOMIC_RELAXED);
}
__global__ void __launch_bounds__(1024)
gftt_finish_kernel(int w, int h, const unsigned long long* __restrict__ keys_all, const int* __restrict__ count_all, int cap, float min_distance,
                   int* __restrict__ cand_at_all, const uint32_t* __restrict__ deps_all, const uint8_t* __restrict__ depn_all, uint8_t* __restrict__ state_all,
                   unsigned long long* __restrict__ kept_all, int* __restrict__ nkept_all, int* __restrict__ overflow)
{
    __shared__ int s_pending, s_nkept;
    const int f = blockIdx.x, tid = threadIdx.x;
    int* cand_at = cand_at_all + (size_t)f * w * h; uint8_t* state = state_all + (size_t)f * cap;
    const unsigned long long* keys = keys_all + (size_t)f * cap; unsigned long long* kept = kept_all + (size_t)f * cap;
    int nc = count_all[f];
    if (nc > cap) { nc = cap; if (tid == 0) atomicOr(overflow, 1); }       // candidate buffer too small: reported by the host
    if (tid == 0) { s_pending = 0; s_nkept = 0; }
    __syncthreads();
    const float md2 = min_distance * min_distance;
    const int rad = (int)ceilf(min_distance);
    for (;;) {
        for (int i = tid; i < nc; i += 1024) {
            if (__atomic_load_n(&state[i], __ATOMIC_RELAXED) != 1) continue;
            const int n = depn_all[(size_t)f * cap + i];
            bool blocked = false, rejected = false;
            if (n <= GFTT_DEPS) {
                const uint32_t* deps = deps_all + ((size_t)f * cap + i) * GFTT_DEPS;
                for (int k = 0; k < n; k++) {
                    const int sj = __atomic_load_n(&state[deps[k]], __ATOMIC_RELAXED);
                    if (sj == 2) { rejected = true; break; }
                    if (sj == 1) blocked = true;
                }
            } else {                                          // more stronger neighbours than the list holds: scan the window
                const unsigned long long key = keys[i];
                const unsigned idx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu);
                const int y = idx / w, x = idx - y * w;
                for (int dy = -rad; dy <= rad && !rejected; dy++) {
                    const int yy = y + dy;
                    if (yy < 0 || yy >= h) continue;
                    for (int dx = -rad; dx <= rad; dx++) {
                        const int xx = x + dx;
                        if (xx < 0 || xx >= w || (dx == 0 && dy == 0)) continue;
                        if ((float)dx * (float)dx + (float)dy * (float)dy >= md2) continue;
                        const int j = cand_at[yy * w + xx] - 1;
                        if (j < 0 || keys[j] < key) continue;
                        const int sj = __atomic_load_n(&state[j], __ATOMIC_RELAXED);
                        if (sj == 2) { rejected = true; break; }
                        if (sj == 1) blocked = true;
                    }
                }
            }
            if (rejected) __atomic_store_n(&state[i], (uint8_t)3, __ATOMIC_RELAXED);
            else if (!blocked) __atomic_store_n(&state[i], (uint8_t)2, __ATOMIC_RELAXED);
            else s_pending = 1;
        }
        __syncthreads();
        const int pending = s_pending;
        __syncthreads();
        if (!pending) break;
        if (tid == 0) s_pending = 0;
        __syncthreads();
    }
    // kept corners (any order) -> list; the map goes back to zero
    for (int i = tid; i < nc; i += 1024) {
        const unsigned long long key = keys[i];
        cand_at[0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu)] = 0;
        if (state[i] == 2) kept[atomicAdd(&s_nkept, 1)] = key;
    }
    __syncthreads();
    if (tid == 0) nkept_all[f] = s_nkept;
}
__global__ void __launch_bounds__(256)
gftt_rank_kernel(const unsigned long long* __restrict__ kept_all, const int* __restrict__ nkept_all, int cap, int w, int max_corners,
                 float* __restrict__ pts_all, int pts_stride, int* __restrict__ nout_all)
{
    const int f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nk = nkept_all[f];
    if (i == 0) nout_all[f] = (max_corners > 0 && nk > max_corners) ? max_corners : nk;
    if ((int)(blockIdx.x * blockDim.x) >= nk) return;                        // (block-uniform)
    const unsigned long long* kept = kept_all + (size_t)f * cap;
    const unsigned long long key = i < nk ? kept[i] : ~0ull;
    // the kept keys pass through LDS 256 at a time (a compare per broadcast read; the plain loop over global memory waited for one 8-byte load per key: 0.35 ms per 64 frames)
    __shared__ unsigned long long tile[256];
    int r = 0;
    for (int j0 = 0; j0 < nk; j0 += 256) {
        __syncthreads();
        tile[threadIdx.x] = j0 + (int)threadIdx.x < nk ? kept[j0 + threadIdx.x] : 0ull;      // 0 is below every key (never counted)
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < 256; j++) r += tile[j] > key;
    }
    if (i < nk && (max_corners <= 0 || r < max_corners)) {
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu);
        float* pts = pts_all + (size_t)f * pts_stride * 2;
        pts[2 * r] = (float)(idx % (unsigned)w); pts[2 * r + 1] = (float)(idx / (unsigned)w);
    }
}

// ------------------------------------------------------------------ pyramid + Scharr for LK (all frames and both sides of a sub-batch per launch)
// Both kernels make FOUR neighbouring outputs per thread from a few wide loads: with one output and 25 / 9 byte loads per thread they spent 0.68 / 0.56 of their
// time in the texture-address units (profiles/r04_stereo_ta_busy.md).  Outputs next to the left / right border (reflection) take the one-by-one path.
// blockIdx.y = f * 2 + side
__device__ __forceinline__ int pyrdown_px(const uint8_t* __restrict__ src, int w, int h, int x, int y)
{
    const int k[5] = {1, 4, 6, 4, 1};
    int s = 0;
#pragma unroll
    for (int j = -2; j <= 2; j++) {
        const uint8_t* r = src + (size_t)refl101d(2 * y + j, h) * w;
        int rs = 0;
#pragma unroll
        for (int qq = -2; qq <= 2; qq++) rs += k[qq + 2] * r[refl101d(2 * x + qq, w)];
        s += k[j + 2] * rs;
    }
    return (s + 128) >> 8;
}
__global__ void __launch_bounds__(256)
pyrdown_kernel(QuadBatch q, uint8_t* __restrict__ pyr, int level)
{
    const int f = blockIdx.y >> 1, side = blockIdx.y & 1;
    const int w = q.w[level - 1], h = q.h[level - 1], dw = q.w[level], dh = q.h[level];
    const size_t sl = (size_t)(side * q.B1 + 1 + f) * q.slot_elems;
    const uint8_t* src = pyr + sl + q.off[level - 1]; uint8_t* dst = pyr + sl + q.off[level];
    const int groups = (dw + 3) >> 2;                        // four outputs of a row per thread
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * dh) return;
    const int y = i / groups, x = (i - y * groups) << 2;
    if (x + 3 < dw && 2 * x - 2 >= 0 && 2 * x + 9 <= w - 1) {
        // source columns 2x - 2 .. 2x + 8 of five rows: 12 bytes per row
        const int k[5] = {1, 4, 6, 4, 1};
        int s4[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = -2; j <= 2; j++) {
            const uint8_t* r = src + (size_t)refl101d(2 * y + j, h) * w + (2 * x - 2);
            uint32_t d[3]; __builtin_memcpy(d, r, 12);
            int b[12];
#pragma unroll
            for (int t = 0; t < 12; t++) b[t] = (int)((d[t >> 2] >> (8 * (t & 3))) & 255u);
#pragma unroll
            for (int o = 0; o < 4; o++) s4[o] += k[j + 2] * (b[2 * o] + 4 * b[2 * o + 1] + 6 * b[2 * o + 2] + 4 * b[2 * o + 3] + b[2 * o + 4]);
        }
        const uint32_t out = (uint32_t)((s4[0] + 128) >> 8) | ((uint32_t)((s4[1] + 128) >> 8) << 8) | ((uint32_t)((s4[2] + 128) >> 8) << 16) | ((uint32_t)((s4[3] + 128) >> 8) << 24);
        __builtin_memcpy(dst + (size_t)y * dw + x, &out, 4);
    } else {
        for (int o = 0; o < 4 && x + o < dw; o++) dst[(size_t)y * dw + x + o] = (uint8_t)pyrdown_px(src, w, h, x + o, y);
    }
}
// all four levels of a slot in one launch: thread i = elements 4 i .. 4 i + 3 of the slot (a level starts at a multiple of 16 elements)
__device__ __forceinline__ short2 scharr_px(const uint8_t* __restrict__ src, int w, int h, int x, int y)
{
    const uint8_t *r0 = src + (size_t)refl101d(y - 1, h) * w, *r1 = src + (size_t)y * w, *r2 = src + (size_t)refl101d(y + 1, h) * w;
    const int xm = refl101d(x - 1, w), xp = refl101d(x + 1, w);
    return make_short2((short)(3 * (r0[xp] - r0[xm]) + 10 * (r1[xp] - r1[xm]) + 3 * (r2[xp] - r2[xm])),
                       (short)(3 * (r2[xm] - r0[xm]) + 10 * (r2[x] - r0[x]) + 3 * (r2[xp] - r0[xp])));
}
__global__ void __launch_bounds__(256)
scharr_kernel(QuadBatch q, short2* __restrict__ der)
{
    const int f = blockIdx.y >> 1, side = blockIdx.y & 1;
    const int e = (blockIdx.x * blockDim.x + threadIdx.x) << 2;
    if (e >= (int)q.slot_elems) return;
    const int level = e >= q.off[3] ? 3 : e >= q.off[2] ? 2 : e >= q.off[1] ? 1 : 0;
    const int w = q.w[level], h = q.h[level], i = e - q.off[level];
    const size_t sl = (size_t)(side * q.B1 + 1 + f) * q.slot_elems;
    const uint8_t* src = q.pyr + sl + q.off[level];
    const int y = i / w, x = i - y * w;
    if (y < h && x >= 1 && x + 6 <= w - 1) {                 // the four outputs and their neighbours x - 1 .. x + 4 lie in one row, and the 8-byte windows stay inside it
        unsigned long long a0, a1, a2;
        __builtin_memcpy(&a0, src + (size_t)refl101d(y - 1, h) * w + x - 1, 8); __builtin_memcpy(&a1, src + (size_t)y * w + x - 1, 8); __builtin_memcpy(&a2, src + (size_t)refl101d(y + 1, h) * w + x - 1, 8);
        short2 o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int m0 = (int)((a0 >> (8 * k)) & 255u), c0 = (int)((a0 >> (8 * k + 8)) & 255u), p0 = (int)((a0 >> (8 * k + 16)) & 255u);
            const int m1 = (int)((a1 >> (8 * k)) & 255u), p1 = (int)((a1 >> (8 * k + 16)) & 255u);
            const int m2 = (int)((a2 >> (8 * k)) & 255u), c2 = (int)((a2 >> (8 * k + 8)) & 255u), p2 = (int)((a2 >> (8 * k + 16)) & 255u);
            o[k] = make_short2((short)(3 * (p0 - m0) + 10 * (p1 - m1) + 3 * (p2 - m2)), (short)(3 * (m2 - m0) + 10 * (c2 - c0) + 3 * (p2 - p0)));
        }
        uint4 out; __builtin_memcpy(&out, o, 16);
        *reinterpret_cast<uint4*>(der + sl + e) = out;      // (sl and e are multiples of 4 elements: 16-byte aligned)
    } else {
        // (elements in the padding behind a level are computed like the one-element kernel did: rows past the level, values nobody reads)
        for (int k = 0; k < 4; k++) {
            const int ik = i + k, yk = ik / w, xk = ik - yk * w;
            const uint8_t *r0 = src + (size_t)refl101d(yk - 1, h) * w, *r1 = src + (size_t)yk * w, *r2 = src + (size_t)refl101d(yk + 1, h) * w;
            const int xm = refl101d(xk - 1, w), xp = refl101d(xk + 1, w);
            der[sl + e + k] = make_short2((short)(3 * (r0[xp] - r0[xm]) + 10 * (r1[xp] - r1[xm]) + 3 * (r2[xp] - r2[xm])),
                                          (short)(3 * (r2[xm] - r0[xm]) + 10 * (r2[xk] - r0[xk]) + 3 * (r2[xp] - r0[xp])));
        }
    }
}

// ------------------------------------------------------------------ pyramidal LK, one wave per point, all levels in one launch
#define LKW 11
#define LKL 4
// Wave sums of the window's integer terms (integers: any order gives the sum).  |Ix|, |Iy| <= 16 * 255 = 4080 (Scharr) and |I|, |J - I| <= 255 * 32, so a sum of
// 121 derivative products fits 32 bits (121 * 4080^2 < 2^31) and so does a sum of the 64 mismatch products of lanes 0 .. 31 (64 * 8160 * 4080 < 2^31); the
// mismatch sums of the whole window can need 33 bits: two half-wave sums, added as doubles (exact).  DPP prefix sums inside the rows of 16, then row_bcast.
#define LK_DPP(v, ctrl, rows) __builtin_amdgcn_update_dpp(0, (v), (ctrl), (rows), 0xF, false)
__device__ __forceinline__ int lk_sum_i32(int v)
{
    v += LK_DPP(v, 0x111, 0xF); v += LK_DPP(v, 0x112, 0xF); v += LK_DPP(v, 0x114, 0xF); v += LK_DPP(v, 0x118, 0xF);
    v += LK_DPP(v, 0x142, 0xA); v += LK_DPP(v, 0x143, 0xC);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double lk_sum_wide(int v)
{
    v += LK_DPP(v, 0x111, 0xF); v += LK_DPP(v, 0x112, 0xF); v += LK_DPP(v, 0x114, 0xF); v += LK_DPP(v, 0x118, 0xF);
    v += LK_DPP(v, 0x142, 0xA);
    return (double)__builtin_amdgcn_readlane(v, 31) + (double)__builtin_amdgcn_readlane(v, 63);
}
#define DESCALE(v, n) (((v) + (1 << ((n) - 1))) >> (n))
typedef short lk_s2 __attribute__((ext_vector_type(2)));
// a * w.x + b * w.y + c for two bytes (a in bits 0 .. 7, b in bits 8 .. 15 of `pair`) and two 16-bit weights: v_perm_b32 + v_dot2_i32_i16
__device__ __forceinline__ int lk_dot_bytes(uint32_t pair, lk_s2 w, int c)
{
    const uint32_t sp = __builtin_amdgcn_perm(0u, pair, 0x0c010c00u);
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(lk_s2, sp), w, c, false);
}
__device__ __forceinline__ uint32_t lk_load16(const uint8_t* p) { unsigned short t; __builtin_memcpy(&t, p, 2); return t; }
// cv::calcOpticalFlowPyrLK for ONE point by one wave: previous image (ps, pslot), next image (ns, nslot), derivatives of the previous image.
// A window that lies inside the image with its +1 taps (wave-uniform, and the usual case) needs no border reflection and no tap tests; its bilinear taps are
// two 2-byte loads (8-byte loads for the derivative pairs) and two dot products with the packed 15-bit weights.
// NT next images tracked from the SAME previous image and point (lk_quad_kernel: the current-left corner into the current-right and into the previous-left image):
// the per-level set-up -- the window of the previous image and of its derivatives, the 2 x 2 matrix, the minimum-eigenvalue test -- depends on neither and is made once.
template <int NT>
__device__ __forceinline__ void lk_point(const QuadBatch& q, int ps, int pslot, const int (&ns)[NT], const int (&nslot)[NT], float p0x, float p0y, int lane,
                                         int max_count, float eps2, float min_eig_thr, float (&nxs)[NT], float (&nys)[NT], int (&sts)[NT], float& er)
{
    const float FLT_SCALE = 1.f / (1 << 20);
    const float half = (LKW - 1) * 0.5f;
#pragma unroll
    for (int tr = 0; tr < NT; tr++) { nxs[tr] = 0.f; nys[tr] = 0.f; sts[tr] = 1; }
    er = 0.f;
    // this lane's two window pixels: e0 = lane, e1 = lane + 64 (valid when < 121)
    const int e0 = lane, e1 = lane + 64;
    const int wy0 = e0 / LKW, wx0 = e0 - wy0 * LKW, wy1 = e1 / LKW, wx1 = e1 - wy1 * LKW;
    const bool v1 = e1 < LKW * LKW;
    for (int level = LKL - 1; level >= 0; level--) {
        const int W = q.w[level], H = q.h[level];
        const uint8_t* P = q_img(q, ps, pslot, level); const short2* D = q_der(q, ps, pslot, level);
        float ppx = p0x * (float)(1. / (1 << level)), ppy = p0y * (float)(1. / (1 << level));
#pragma unroll
        for (int tr = 0; tr < NT; tr++) { if (level == LKL - 1) { nxs[tr] = ppx; nys[tr] = ppy; } else { nxs[tr] *= 2.f; nys[tr] *= 2.f; } }
        ppx -= half; ppy -= half;
        const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
        if (ipx < -LKW || ipx >= W || ipy < -LKW || ipy >= H) { if (level == 0) { for (int tr = 0; tr < NT; tr++) sts[tr] = 0; er = 0.f; } continue; }
        float a = ppx - ipx, b = ppy - ipy;
        int iw00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << 14)), iw01 = __float2int_rn(a * (1.f - b) * (1 << 14)), iw10 = __float2int_rn((1.f - a) * b * (1 << 14));
        int iw11 = (1 << 14) - iw00 - iw01 - iw10;
        const uint32_t woff0 = (uint32_t)(wy0 * W + wx0), woff1 = (uint32_t)(wy1 * W + wx1);      // this lane's pixels relative to the window's corner
        int I[2], Ix[2], Iy[2];
        int sA11 = 0, sA12 = 0, sA22 = 0;
        if (ipx >= 0 && ipy >= 0 && ipx + LKW < W && ipy + LKW < H) {
            const lk_s2 w0 = { (short)iw00, (short)iw01 }, w1 = { (short)iw10, (short)iw11 };
            const uint32_t base = (uint32_t)(ipy * W + ipx);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                I[t] = Ix[t] = Iy[t] = 0;
                if (t == 1 && !v1) continue;
                const uint32_t o = base + (t ? woff1 : woff0);
                const int iv = lk_dot_bytes(lk_load16(P + o + W), w1, lk_dot_bytes(lk_load16(P + o), w0, 1 << (14 - 5 - 1))) >> (14 - 5);
                uint2 r0, r1; __builtin_memcpy(&r0, D + o, 8); __builtin_memcpy(&r1, D + o + W, 8);       // (dx, dy) of two neighbours each
                const lk_s2 x0 = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm(r0.y, r0.x, 0x05040100u)), y0 = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm(r0.y, r0.x, 0x07060302u));
                const lk_s2 x1 = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm(r1.y, r1.x, 0x05040100u)), y1 = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm(r1.y, r1.x, 0x07060302u));
                const int ix = __builtin_amdgcn_sdot2(x1, w1, __builtin_amdgcn_sdot2(x0, w0, 1 << 13, false), false) >> 14;
                const int iy = __builtin_amdgcn_sdot2(y1, w1, __builtin_amdgcn_sdot2(y0, w0, 1 << 13, false), false) >> 14;
                I[t] = (short)iv; Ix[t] = (short)ix; Iy[t] = (short)iy;
                sA11 += Ix[t] * Ix[t]; sA12 += Ix[t] * Iy[t]; sA22 += Iy[t] * Iy[t];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; t++) {
                I[t] = Ix[t] = Iy[t] = 0;
                if (t == 1 && !v1) continue;
                const int gx = ipx + (t ? wx1 : wx0), gy = ipy + (t ? wy1 : wy0);
                const int x0 = refl101d(gx, W), x1 = refl101d(gx + 1, W), y0 = refl101d(gy, H), y1 = refl101d(gy + 1, H);
                const int iv = DESCALE(P[(size_t)y0 * W + x0] * iw00 + P[(size_t)y0 * W + x1] * iw01 + P[(size_t)y1 * W + x0] * iw10 + P[(size_t)y1 * W + x1] * iw11, 14 - 5);
                const bool i00 = gx >= 0 && gx < W && gy >= 0 && gy < H, i01 = gx + 1 >= 0 && gx + 1 < W && gy >= 0 && gy < H;
                const bool i10 = gx >= 0 && gx < W && gy + 1 >= 0 && gy + 1 < H, i11 = gx + 1 >= 0 && gx + 1 < W && gy + 1 >= 0 && gy + 1 < H;
                const short2 z = make_short2(0, 0);
                const short2 d00 = i00 ? D[(size_t)gy * W + gx] : z, d01 = i01 ? D[(size_t)gy * W + gx + 1] : z;
                const short2 d10 = i10 ? D[(size_t)(gy + 1) * W + gx] : z, d11 = i11 ? D[(size_t)(gy + 1) * W + gx + 1] : z;
                const int ix = DESCALE(d00.x * iw00 + d01.x * iw01 + d10.x * iw10 + d11.x * iw11, 14);
                const int iy = DESCALE(d00.y * iw00 + d01.y * iw01 + d10.y * iw10 + d11.y * iw11, 14);
                I[t] = (short)iv; Ix[t] = (short)ix; Iy[t] = (short)iy;
                sA11 += Ix[t] * Ix[t]; sA12 += Ix[t] * Iy[t]; sA22 += Iy[t] * Iy[t];
            }
        }
        const float A11 = (float)lk_sum_i32(sA11) * FLT_SCALE, A12 = (float)lk_sum_i32(sA12) * FLT_SCALE, A22 = (float)lk_sum_i32(sA22) * FLT_SCALE;
        float Dt = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * LKW * LKW);
        er = minEig;
        if (minEig < min_eig_thr || Dt < FLT_EPSILON) { if (level == 0) { for (int tr = 0; tr < NT; tr++) sts[tr] = 0; } continue; }
        Dt = 1.f / Dt;
#pragma unroll
        for (int tr = 0; tr < NT; tr++) {
        const uint8_t* N = q_img(q, ns[tr], nslot[tr], level);
        float& nx = nxs[tr]; float& ny = nys[tr]; int& st = sts[tr];
        float npx = nx - half, npy = ny - half;
        float pdx = 0.f, pdy = 0.f;
        // the window's taps of the next image stay in registers while the window's integer corner does not move (a converging track moves by fractions of a
        // pixel: only the weights change) -- the loads were the start of every iteration's dependent chain and half of the kernel's texture-address work
        int tap_x = INT_MIN, tap_y = INT_MIN; uint32_t tapA[2] = {0u, 0u}, tapB[2] = {0u, 0u};
        for (int j = 0; j < max_count; j++) {
            const int inx = (int)floorf(npx), iny = (int)floorf(npy);
            if (inx < -LKW || inx >= W || iny < -LKW || iny >= H) { if (level == 0) st = 0; break; }
            a = npx - inx; b = npy - iny;
            iw00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << 14)); iw01 = __float2int_rn(a * (1.f - b) * (1 << 14)); iw10 = __float2int_rn((1.f - a) * b * (1 << 14));
            iw11 = (1 << 14) - iw00 - iw01 - iw10;
            int sb1 = 0, sb2 = 0;
            if (inx >= 0 && iny >= 0 && inx + LKW < W && iny + LKW < H) {
                const lk_s2 w0 = { (short)iw00, (short)iw01 }, w1 = { (short)iw10, (short)iw11 };
                if (inx != tap_x || iny != tap_y) {                   // (wave-uniform)
                    const uint32_t base = (uint32_t)(iny * W + inx);
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        if (t == 1 && !v1) continue;
                        const uint32_t o = base + (t ? woff1 : woff0);
                        tapA[t] = lk_load16(N + o); tapB[t] = lk_load16(N + o + W);
                    }
                    tap_x = inx; tap_y = iny;
                }
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    if (t == 1 && !v1) continue;
                    const int diff = (lk_dot_bytes(tapB[t], w1, lk_dot_bytes(tapA[t], w0, 1 << (14 - 5 - 1))) >> (14 - 5)) - I[t];
                    sb1 += __mul24(diff, Ix[t]); sb2 += __mul24(diff, Iy[t]);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    if (t == 1 && !v1) continue;
                    const int gx = inx + (t ? wx1 : wx0), gy = iny + (t ? wy1 : wy0);
                    const int x0 = refl101d(gx, W), x1 = refl101d(gx + 1, W), y0 = refl101d(gy, H), y1 = refl101d(gy + 1, H);
                    const int diff = DESCALE(N[(size_t)y0 * W + x0] * iw00 + N[(size_t)y0 * W + x1] * iw01 + N[(size_t)y1 * W + x0] * iw10 + N[(size_t)y1 * W + x1] * iw11, 14 - 5) - I[t];
                    sb1 += diff * Ix[t]; sb2 += diff * Iy[t];
                }
            }
            const float b1 = (float)lk_sum_wide(sb1) * FLT_SCALE, b2 = (float)lk_sum_wide(sb2) * FLT_SCALE;
            const float ddx = (A12 * b2 - A22 * b1) * Dt, ddy = (A12 * b1 - A11 * b2) * Dt;
            npx += ddx; npy += ddy;
            nx = npx + half; ny = npy + half;
            if (ddx * ddx + ddy * ddy <= eps2) break;
            if (j > 0 && fabsf(ddx + pdx) < 0.01f && fabsf(ddy + pdy) < 0.01f) { nx -= ddx * 0.5f; ny -= ddy * 0.5f; break; }
            pdx = ddx; pdy = ddy;
        }
        }
    }
}
// one pass (ssm_lk_track): previous = (side 0, slot 1), next = (side 1, slot 1)
__global__ void __launch_bounds__(256)
lk_kernel(QuadBatch q, const float* __restrict__ prev_pts, int n, float* __restrict__ next_pts, uint8_t* __restrict__ status, float* __restrict__ err,
          int max_count, float eps2, float min_eig_thr)
{
    const int lane = threadIdx.x & 63;
    const int pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pi >= n) return;
    float nx[1], ny[1], er; int st[1];
    const int ns1[1] = {1}, sl1[1] = {1};
    lk_point<1>(q, 0, 1, ns1, sl1, prev_pts[2*pi], prev_pts[2*pi+1], lane, max_count, eps2, min_eig_thr, nx, ny, st, er);
    if (lane == 0) { next_pts[2*pi] = nx[0]; next_pts[2*pi+1] = ny[0]; status[pi] = (uint8_t)st[0]; if (err) err[pi] = er; }
}
// the four passes of QuadFeatureMatch::circularMatching in tracking mode (quadmatcher.cpp:566-576) for one GFTT corner of frame f by one wave:
// lc -> rc, rc -> rp, rp -> lp and lc -> lp (direct); current frame = slot 1 + f, previous = slot f.  LK status vectors are ignored downstream
// (SURVEY.md quirk 10), so only the positions leave the kernel.  pts: [5][nb][stride] (x, y) pairs = lc, rc, rp, lp, lp_direct.
__global__ void __launch_bounds__(256)
lk_quad_kernel(QuadBatch q, float* __restrict__ pts, int stride, const int* __restrict__ ncorner, const int* __restrict__ has_prev, int nb,
               int max_count, float eps2, float min_eig_thr)
{
    const int lane = threadIdx.x & 63, f = blockIdx.y;
    const int pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (!has_prev[f] || pi >= ncorner[f]) return;
    const size_t set = (size_t)nb * stride * 2;
    float* p = pts + ((size_t)f * stride + pi) * 2;
    const float x0 = p[0], y0 = p[1];
    float er;
    float xa[2], ya[2]; int sa[2];                           // lc -> rc and lc -> lp (direct): one set-up per level for both
    { const int nsd[2] = {1, 0}, nsl[2] = {1 + f, f}; lk_point<2>(q, 0, 1 + f, nsd, nsl, x0, y0, lane, max_count, eps2, min_eig_thr, xa, ya, sa, er); }
    const float x1 = xa[0], y1 = ya[0], x4 = xa[1], y4 = ya[1];
    float x2[1], y2[1], x3[1], y3[1]; int s1[1];
    { const int nsd[1] = {1}, nsl[1] = {f}; lk_point<1>(q, 1, 1 + f, nsd, nsl, x1, y1, lane, max_count, eps2, min_eig_thr, x2, y2, s1, er); }          // rc -> rp
    { const int nsd[1] = {0}, nsl[1] = {f}; lk_point<1>(q, 1, f, nsd, nsl, x2[0], y2[0], lane, max_count, eps2, min_eig_thr, x3, y3, s1, er); }        // rp -> lp
    if (lane == 0) { p[set] = x1; p[set + 1] = y1; p[2 * set] = x2[0]; p[2 * set + 1] = y2[0]; p[3 * set] = x3[0]; p[3 * set + 1] = y3[0]; p[4 * set] = x4; p[4 * set + 1] = y4; }
}

// ------------------------------------------------------------------ filteringTracks (quadmatcher.cpp:420-503), ordered compaction, one block per frame
struct QPmatch { float u1p, v1p; int i1p; float u2p, v2p; int i2p; float u1c, v1c; int i1c; float u2c, v2c; int i2c; short dis_c, dis_p; };
__device__ __forceinline__ bool within_region(float x, float y) { return x < 1280 && x > 0.0f && y < 960 && y > 0.0f; }
__global__ void __launch_bounds__(1024)
filter_tracks_kernel(const float* __restrict__ pts, int stride, const int* __restrict__ ncorner, const int* __restrict__ has_prev, int nb,
                     QPmatch* __restrict__ out_all, int* __restrict__ nout_all)
{
    __shared__ int wcnt[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, f = blockIdx.x;
    if (!has_prev[f]) { if (tid == 0) nout_all[f] = -1; return; }      // no previous frame: no quad matches (first frame of a sequence)
    const int n = ncorner[f];
    const size_t set = (size_t)nb * stride * 2;
    const float *lc = pts + (size_t)f * stride * 2, *rc = lc + set, *rp = lc + 2 * set, *lp = lc + 3 * set, *ld = lc + 4 * set;
    QPmatch* out = out_all + (size_t)f * stride;
    if (tid == 0) base = 0;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        bool keep = false; QPmatch r;
        if (i < n) {
            const float lcx = lc[2*i], lcy = lc[2*i+1], rcx = rc[2*i], rcy = rc[2*i+1], lpx = lp[2*i], lpy = lp[2*i+1], rpx = rp[2*i], rpy = rp[2*i+1];
            const float ldx = ld[2*i], ldy = ld[2*i+1];
            const int dh1 = __float2int_rn(fabsf(lcy - rcy)), dh2 = __float2int_rn(fabsf(lpy - rpy));
            const int dh11 = __float2int_rn(fabsf(lcy - lpy)), dh22 = __float2int_rn(fabsf(rcy - rpy));
            const int dw1 = __float2int_rn(fabsf(lcx - lpx)), dw2 = __float2int_rn(fabsf(rcx - rpx));
            const int disp1 = __float2int_rn(fabsf(lcx - rcx)), disp2 = __float2int_rn(fabsf(lpx - rpx));
            const int dfx = __float2int_rn(fabsf(lpx - ldx)), dfy = __float2int_rn(fabsf(lpy - ldy));
            keep = within_region(lcx, lcy) && within_region(lpx, lpy) && within_region(rcx, rcy) && within_region(rpx, rpy) &&
                   dh1 < 20 && dh2 < 20 && dh11 < 30 && dh22 < 30 && dw1 < 200 && dw2 < 200 && disp1 > 3 && disp2 > 3 && dfx < 1 && dfy < 1;
            r.u1c = lcx; r.v1c = lcy; r.u1p = lpx; r.v1p = lpy; r.u2c = rcx; r.v2c = rcy; r.u2p = rpx; r.v2p = rpy;
            r.i1c = r.i1p = r.i2c = r.i2p = i; r.dis_c = 0; r.dis_p = 0;
        }
        const unsigned long long bal = __ballot(keep);
        __syncthreads();
        if (lane == 0) wcnt[wv] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int k = 0; k < wv; k++) off += wcnt[k];
        if (keep) out[off + __popcll(bal & ((1ull << lane) - 1ull))] = r;
        __syncthreads();
        if (tid == 0) { int s = 0; for (int k = 0; k < 16; k++) s += wcnt[k]; base += s; }
    }
    __syncthreads();
    if (tid == 0) nout_all[f] = base;
}

// ------------------------------------------------------------------ QuadFeatureMatch::matching on binary descriptors (:41-83)
__global__ void __launch_bounds__(256)
window_match_kernel(const float* __restrict__ kp1, const uint8_t* __restrict__ d1, int n1, const float* __restrict__ kp2, const uint8_t* __restrict__ d2, int n2,
                    float sw, float sh, float thr, ssm_dmatch* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    const uint4* q = reinterpret_cast<const uint4*>(d1 + (size_t)i * 32);
    const uint4 a = q[0], b = q[1];
    const float x1 = kp1[2*i], y1 = kp1[2*i+1];
    int id = 0; float mind = 999999999.9f;
    for (int j = 0; j < n2; j++) {
        if (fabsf(kp2[2*j] - x1) < sw && fabsf(kp2[2*j+1] - y1) < sh) {
            const uint4* t = reinterpret_cast<const uint4*>(d2 + (size_t)j * 32);
            const uint4 x = t[0], y = t[1];
            const int d = __popc(a.x ^ x.x) + __popc(a.y ^ x.y) + __popc(a.z ^ x.z) + __popc(a.w ^ x.w) + __popc(b.x ^ y.x) + __popc(b.y ^ y.y) + __popc(b.z ^ y.z) + __popc(b.w ^ y.w);
            if ((float)d < mind) { mind = (float)d; id = j; }
        }
    }
    if (mind > thr) id = -1;
    ssm_dmatch m; m.queryIdx = i; m.trainIdx = id; m.imgIdx = -1; m.distance = mind;
    out[i] = m;
}

// ------------------------------------------------------------------ launchers (nb frames per launch)
// level 0 of frames [0, nb) of both sides is in place (slots 1 .. nb): the three pyrDown levels, then the Scharr derivatives of every level
hipError_t k_quad_pyramids(const QuadBatch& q, int nb, hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    uint8_t* pyr = const_cast<uint8_t*>(q.pyr);
    for (int l = 1; l < 4; l++) pyrdown_kernel<<<dim3((((q.w[l] + 3) >> 2) * q.h[l] + 255) / 256, nb * 2), 256, 0, s>>>(q, pyr, l);
    scharr_kernel<<<dim3(((int)(q.slot_elems >> 2) + 255) / 256, nb * 2), 256, 0, s>>>(q, reinterpret_cast<short2*>(const_cast<int16_t*>(q.der)));
    return hipGetLastError();
}
// cv::goodFeaturesToTrack on the left image of frames [0, nb): pts[f][stride] (x, y), ncorner[f].  Workspace (GfttWork): eig nb*w*h floats; cand_at
// nb*w*h ints, ZERO on entry (left zeroed); keys / kept nb*cap u64; deps nb*cap*GFTT_DEPS u32; depn / state nb*cap bytes; maxord / count / nkept nb ints;
// overflow 1 int (set when a frame has more than cap candidates)
size_t k_quad_gftt_bits_words(int w, int h) { return ((size_t)w * h + 256 * GC_PX - 1) / (256 * GC_PX) * (8 * GC_PX) + 2; }      // whole blocks of gftt_collect_kernel (256 x GC_PX pixels) + the word a window may read past the end
size_t k_quad_gftt_deps_per_candidate() { return GFTT_DEPS; }
hipError_t k_quad_gftt(const QuadBatch& q, int nb, int max_corners, double quality, double min_distance, const GfttWork& g, float* pts, int stride, int* ncorner, hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    const int w = q.w[0], h = q.h[0], cap = g.cap;
    hipError_t e = hipMemsetAsync(g.maxord, 0x80, 4 * (size_t)nb, s);          // 0x80808080: below any real value in the ordered-int encoding
    if (e == hipSuccess) e = hipMemsetAsync(g.count, 0, 4 * (size_t)nb, s);
    if (e != hipSuccess) return e;
    const int tx = (w + ME_W - 1) / ME_W, ty = (h + ME_H - 1) / ME_H;
    mineig_kernel<<<dim3(tx * ty, nb), 256, 0, s>>>(q, g.eig, g.maxord);
    const int bw = (int)k_quad_gftt_bits_words(w, h);
    gftt_collect_kernel<<<dim3((w * h + 256 * GC_PX - 1) / (256 * GC_PX), nb), 256, 0, s>>>(g.eig, w, h, g.maxord, quality, g.keys, g.count, cap, g.cand_at, g.cand_bits, bw, (uint32_t)(((1ull << 32) + w - 1) / w));
    const dim3 gc((cap + 255) / 256, nb);
    gftt_deps_kernel<<<gc, 256, 0, s>>>(w, h, g.keys, g.count, cap, (float)min_distance, g.cand_at, g.cand_bits, bw, g.deps, g.depn, g.state);
    for (int r = 0; r < GFTT_ROUNDS; r++) gftt_round_kernel<<<gc, 256, 0, s>>>(g.count, cap, g.deps, g.depn, g.state);
    gftt_finish_kernel<<<nb, 1024, 0, s>>>(w, h, g.keys, g.count, cap, (float)min_distance, g.cand_at, g.deps, g.depn, g.state, g.kept, g.nkept, g.overflow);
    gftt_rank_kernel<<<gc, 256, 0, s>>>(g.kept, g.nkept, cap, w, max_corners, pts, stride, ncorner);
    return hipGetLastError();
}
hipError_t k_quad_lk(const QuadBatch& q, const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err, int max_count, float eps2, float min_eig_thr, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    lk_kernel<<<(n + 3) / 4, 256, 0, s>>>(q, prev_pts, n, next_pts, status, err, max_count, eps2, min_eig_thr);
    return hipGetLastError();
}
// the four LK passes + filteringTracks for frames [0, nb): pts = [5][nb][stride] (set 0 = the GFTT corners), out[f][stride], nout[f] (-1 without a previous frame)
hipError_t k_quad_track(const QuadBatch& q, int nb, float* pts, int stride, const int* ncorner, const int* has_prev, void* out, int* nout, hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    lk_quad_kernel<<<dim3((stride + 3) / 4, nb), 256, 0, s>>>(q, pts, stride, ncorner, has_prev, nb, 200, (float)(0.01 * 0.01), 1e-6f);
    filter_tracks_kernel<<<nb, 1024, 0, s>>>(pts, stride, ncorner, has_prev, nb, reinterpret_cast<QPmatch*>(out), nout);
    return hipGetLastError();
}
hipError_t k_quad_window_match(const float* kp1, const uint8_t* d1, int n1, const float* kp2, const uint8_t* d2, int n2, int sw, int sh, float thr,
                               ssm_dmatch* out, hipStream_t s)
{
    if (n1 <= 0) return hipSuccess;
    window_match_kernel<<<(n1 + 255) / 256, 256, 0, s>>>(kp1, d1, n1, kp2, d2, n2, (float)sw, (float)sh, thr, out);
    return hipGetLastError();
}
