// kernels_orb.hip -- K1..K5 of the hot path for gfx950: what OrbFeature::detectFeatures
// (/root/reference/include/orb.h:32-53) does through cv::cvtColor (orb.h:39) and ORB_SLAM2::ORBextractor (orb.h:44),
// batched over frames (grid.y = frame).  Integer / byte work, HBM- and LDS-bound: no MFMA here by design.
// Contracts (rounding rules, tie-breaks) are those of oracle/orb.c; compile with -ffp-contract=off.
#include "ssm_internal.h"
#include <cstdlib>
#include <type_traits>

#define WAVE 64

// ------------------------------------------------------------------ K1: BGR -> gray (level 0 of the pyramid)
// 4 pixels per thread: 3 dword loads (12 B) -> 1 dword store when rows are 4-aligned.
template <bool VEC>
__global__ void gray_kernel(const uint8_t* __restrict__ img, int W, int H, int stride0, uint8_t* __restrict__ pyr, int pyr_bytes)
{
    const int quads = stride0 >> 2;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= quads * H) return;
    const int y = q / quads, x = (q - y * quads) << 2;
    const uint8_t* src = img + (size_t)blockIdx.y * W * H * 3 + ((size_t)y * W + x) * 3;
    uint8_t* dst = pyr + (size_t)blockIdx.y * pyr_bytes + (size_t)y * stride0 + x;
    uint32_t out = 0;
    if (x >= W) { *reinterpret_cast<uint32_t*>(dst) = 0; return; }      // row padding
    if (VEC) {
        const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
        uint32_t a = s[0], b = s[1], c = s[2];
        uint32_t p0 = ((a & 255) * 1868 + ((a >> 8) & 255) * 9617 + ((a >> 16) & 255) * 4899 + 8192) >> 14;
        uint32_t p1 = ((a >> 24) * 1868 + (b & 255) * 9617 + ((b >> 8) & 255) * 4899 + 8192) >> 14;
        uint32_t p2 = (((b >> 16) & 255) * 1868 + (b >> 24) * 9617 + (c & 255) * 4899 + 8192) >> 14;
        uint32_t p3 = (((c >> 8) & 255) * 1868 + ((c >> 16) & 255) * 9617 + (c >> 24) * 4899 + 8192) >> 14;
        out = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
    } else {
        for (int i = 0; i < 4; i++)
            if (x + i < W) out |= ((src[3*i] * 1868u + src[3*i+1] * 9617u + src[3*i+2] * 4899u + 8192u) >> 14) << (8 * i);
    }
    *reinterpret_cast<uint32_t*>(dst) = out;
}
__global__ void graycopy_kernel(const uint8_t* __restrict__ img, int in_stride, size_t frame_bytes, int W, int H, int stride0,
                                uint8_t* __restrict__ pyr, int pyr_bytes)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= stride0 * H) return;
    const int y = q / stride0, x = q - y * stride0;
    pyr[(size_t)blockIdx.y * pyr_bytes + q] = x < W ? img[(size_t)blockIdx.y * frame_bytes + (size_t)y * in_stride + x] : 0;
}
hipError_t k_gray(const uint8_t* img, int channels, int n, const OrbGeom& g, uint8_t* pyr, hipStream_t s)
{
    const int stride0 = g.L[0].stride;
    if (channels == 3) {
        dim3 grid(((stride0 >> 2) * g.H + 255) / 256, n);
        if ((g.W & 3) == 0 && (reinterpret_cast<uintptr_t>(img) & 3) == 0)
            gray_kernel<true><<<grid, 256, 0, s>>>(img, g.W, g.H, stride0, pyr, g.pyr_bytes);
        else
            gray_kernel<false><<<grid, 256, 0, s>>>(img, g.W, g.H, stride0, pyr, g.pyr_bytes);
    } else {
        dim3 grid((stride0 * g.H + 255) / 256, n);
        graycopy_kernel<<<grid, 256, 0, s>>>(img, g.W, (size_t)g.W * g.H, g.W, g.H, stride0, pyr, g.pyr_bytes);
    }
    return hipGetLastError();
}
hipError_t k_copy_gray_strided(const uint8_t* img, int stride, const OrbGeom& g, uint8_t* pyr, hipStream_t s)
{
    dim3 grid((g.L[0].stride * g.H + 255) / 256, 1);
    graycopy_kernel<<<grid, 256, 0, s>>>(img, stride, 0, g.W, g.H, g.L[0].stride, pyr, g.pyr_bytes);
    return hipGetLastError();
}

// ------------------------------------------------------------------ K2: pyramid level from the previous level
// cv::resize INTER_LINEAR 8u fixed point (coefficient tables built on the host, oracle/orb.c sso_resize_tables).
// The per-CU texture-address unit spends >= 16 cycles on every vector-memory instruction whatever its width, so the
// source rows a block needs are staged in LDS with 16-byte loads and the results leave as 16-byte stores:
// one block = PY_ROWS output rows x the full output width; 16 output pixels per thread-iteration.
// a block makes `prows` output rows; the launcher picks prows = floor(256 / (16-pixel groups per row)) so that the block's
// work items fill its 256 threads in ONE pass (a fixed 8 rows left a second pass with 16 active threads at level 1)
__global__ void __launch_bounds__(256)
resize_kernel(uint8_t* __restrict__ pyr, int pyr_bytes, int src_off, int sw, int sh, int sstride,
              int dst_off, int dw, int dh, int dstride, int prows,
              const int32_t* __restrict__ xofs, const int16_t* __restrict__ xa,
              const int32_t* __restrict__ yofs, const int16_t* __restrict__ ya)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];      // [xtab: dw x u32 (ofs | a0 << 16)] [xa1: dw x u16] [rows: nrows x sstride]
    uint32_t* xt = reinterpret_cast<uint32_t*>(smem);
    uint16_t* x1 = reinterpret_cast<uint16_t*>(smem + (size_t)dstride * 4);
    uint8_t* rows = smem + (size_t)dstride * 6;                         // dstride % 16 == 0 keeps this 16-B aligned
    const int y0 = blockIdx.x * prows, y1 = min(y0 + prows, dh);
    const int sy_first = yofs[y0], sy_last = min(yofs[y1 - 1] + 1, sh - 1);
    const int nrows = sy_last - sy_first + 1;
    const uint8_t* src = pyr + (size_t)blockIdx.y * pyr_bytes + src_off + (size_t)sy_first * sstride;
    const int nvec = (nrows * sstride) >> 4;                            // strides are multiples of 16
    for (int i = threadIdx.x; i < nvec; i += 256) reinterpret_cast<uint4*>(rows)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (int x = threadIdx.x; x < dw; x += 256) { xt[x] = (uint32_t)xofs[x] | ((uint32_t)(uint16_t)xa[2*x] << 16); x1[x] = (uint16_t)xa[2*x+1]; }
    __syncthreads();
    uint8_t* dst = pyr + (size_t)blockIdx.y * pyr_bytes + dst_off;
    const int groups = dstride >> 4;                                    // 16-pixel groups per output row
    for (int i = threadIdx.x; i < groups * (y1 - y0); i += 256) {
        const int ry = i / groups, g = i - ry * groups, y = y0 + ry, x0 = g << 4;
        const int syA = yofs[y], syB = min(syA + 1, sh - 1);
        const int b0 = ya[2*y], b1 = ya[2*y+1];
        const uint8_t* r0 = rows + (syA - sy_first) * sstride;
        const uint8_t* r1 = rows + (syB - sy_first) * sstride;
        uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int x = x0 + k;
            if (x < dw) {
                const uint32_t t = xt[x];
                const int sx0 = t & 0xFFFF, sx1 = min(sx0 + 1, sw - 1);
                const int a0 = t >> 16, a1 = x1[x];
                const int h0 = r0[sx0] * a0 + r0[sx1] * a1;
                const int h1 = r1[sx0] * a0 + r1[sx1] * a1;
                const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                o[k >> 2] |= (uint32_t)(v & 255) << (8 * (k & 3));
            }
        }
        *reinterpret_cast<uint4*>(dst + (size_t)y * dstride + x0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
// Streaming form (scale factors up to ~1.3, i.e. ORB's 1.2): one thread = 4 consecutive output pixels = one dword store.  Their
// source bytes lie inside 12 bytes from the 4-byte-aligned start of the first one, so a thread reads three dwords from each
// of the two source rows straight from global memory (the source level was written just before: L2 / Infinity-Cache
// resident) -- no LDS staging, no barrier, no byte-granular LDS reads.  Per pixel: pick the dword pair, v_alignbyte to bring
// the two neighbours to the low bytes, v_perm to spread them into 16-bit lanes, v_dot2_u32_u16 with the (a0, a1) pair, then
// the vertical blend with umulhi on coefficients pre-shifted by 16.  The per-group constants (XGroup) come from the host.
struct XGroup { uint32_t a[4]; uint32_t base, offs, pad0, pad1; };   // a[k] = a0 | a1 << 16; base = byte offset of pixel 0's left neighbour (any alignment); offs = 4 bits per pixel: its left neighbour's offset from base (<= 6)
__global__ void __launch_bounds__(256)
resize4_kernel(uint8_t* __restrict__ pyr, int pyr_bytes, int src_off, int sh, int sstride, int dst_off, int dh, int dstride,
               const XGroup* __restrict__ xg, const int32_t* __restrict__ yofs, const int16_t* __restrict__ ya, uint32_t mul_groups)
{
    // one thread = one 4-pixel column group x FOUR consecutive output rows: the group's constants and the rows' tables are loaded once (two + two 16-byte
    // loads; the y tables are padded to a multiple of four rows by the host), then the eight source windows are all in flight before the first is used --
    // with one row per thread every output dword paid two dependent memory latencies (tables, then pixels) and the kernel ran at 2 TB/s
    const int groups = dstride >> 2;
    const int frame = blockIdx.y;                                       // (frame-fastest order measured 5 % slower here: pure streaming, nothing to share)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= groups * ((dh + 3) >> 2)) return;
    const int y4 = (int)__umulhi((uint32_t)i, mul_groups), g = i - y4 * groups, y0 = 4 * y4;       // i / groups by the host's reciprocal (exact: i * groups < 2^32)
    const uint4 A = reinterpret_cast<const uint4*>(xg)[2 * g], Q = reinterpret_cast<const uint4*>(xg)[2 * g + 1];
    const uint4 YO = *reinterpret_cast<const uint4*>(yofs + y0), YA = *reinterpret_cast<const uint4*>(ya + 2 * y0);
    const uint32_t yo[4] = {YO.x, YO.y, YO.z, YO.w}, yc[4] = {YA.x, YA.y, YA.z, YA.w};
    const uint8_t* src = pyr + (size_t)frame * pyr_bytes + src_off + Q.x;
    // the eight bytes that hold the four pixels' neighbour pairs, one (unaligned) 8-byte load per source row; a pixel's pair then comes out of ONE v_perm
    // whose selector is its offset replicated
    uint2 r0[4], r1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int syA = (int)yo[j], syB = min(syA + 1, sh - 1);
        __builtin_memcpy(&r0[j], src + (size_t)syA * sstride, 8);
        __builtin_memcpy(&r1[j], src + (size_t)syB * sstride, 8);
    }
    const uint32_t av[4] = {A.x, A.y, A.z, A.w};
    uint32_t sel[4];
#pragma unroll
    for (int k = 0; k < 4; k++) sel[k] = ((Q.y >> (4 * k)) & 15u) * 0x00010001u + 0x0C010C00u;      // bytes (off, zero, off + 1, zero)
    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
    uint8_t* dst = pyr + (size_t)frame * pyr_bytes + dst_off + (size_t)y0 * dstride + 4 * g;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t b0 = yc[j] << 16, b1 = yc[j] & 0xFFFF0000u;      // (b0, b1) of the row as u16, pre-shifted by 16
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t p0 = __builtin_amdgcn_perm(r0[j].y, r0[j].x, sel[k]), p1 = __builtin_amdgcn_perm(r1[j].y, r1[j].x, sel[k]);
            ushort2v c2, x0, x1; memcpy(&c2, &av[k], 4); memcpy(&x0, &p0, 4); memcpy(&x1, &p1, 4);
            const uint32_t h0 = __builtin_amdgcn_udot2(x0, c2, 0u, false), h1 = __builtin_amdgcn_udot2(x1, c2, 0u, false);
            const uint32_t v = (__umulhi(b0, h0 >> 4) + __umulhi(b1, h1 >> 4) + 2u) >> 2;
            o |= (v & 255u) << (8 * k);
        }
        if (y0 + j < dh) *reinterpret_cast<uint32_t*>(dst + (size_t)j * dstride) = o;
    }
}
hipError_t k_pyramid(int n, const OrbGeom& g, uint8_t* pyr, const int32_t* const* xofs, const int16_t* const* xa,
                     const int32_t* const* yofs, const int16_t* const* ya, const void* const* xgroups, hipStream_t s)
{
    for (int l = 1; l < g.nlevels; l++) {
        const LevelGeom& a = g.L[l-1]; const LevelGeom& b = g.L[l];
        if (xgroups && xgroups[l]) {                                             // streaming form (the host found every window inside 8 bytes)
            const int items = (b.stride >> 2) * ((b.h + 3) >> 2);
            const int groups = b.stride >> 2;
            resize4_kernel<<<dim3((items + 255) / 256, n), 256, 0, s>>>(pyr, g.pyr_bytes, a.img_off, a.h, a.stride, b.img_off, b.h, b.stride,
                                                                        reinterpret_cast<const XGroup*>(xgroups[l]), yofs[l], ya[l], (uint32_t)(((1ull << 32) + groups - 1) / groups));
            continue;
        }
        int prows = 256 / (b.stride >> 4); if (prows < 1) prows = 1; if (prows > 24) prows = 24;
        const int max_rows = (int)((prows - 1) * ((double)a.h / b.h)) + 4;       // source rows one block can touch
        dim3 grid((b.h + prows - 1) / prows, n);
        resize_kernel<<<grid, 256, (size_t)max_rows * a.stride + (size_t)b.stride * 6, s>>>(pyr, g.pyr_bytes, a.img_off, a.w, a.h, a.stride, b.img_off, b.w, b.h, b.stride, prows,
                                                                     xofs[l], xa[l], yofs[l], ya[l]);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ K5a: 7x7 sigma-2 Gaussian, fixed point
// taps {18,34,49,55,49,34,18}; row pass fits u16 (<= 257*255); (v + 2^15) >> 16, saturate; BORDER_REFLECT_101.
// One block per 128x32 tile of one level of one frame (all levels in one launch, same tiling as the FAST kernel):
// tile + apron staged in LDS by dword loads (all issued before first use), 4 pixels per thread in both passes, row-pass
// intermediate kept in LDS as u16 (the two rows of a pair interleaved per pixel), output written as dwords.
#define BT_W 128
#define BT_H 32
#define BT_PW (BT_W + 16)     // staged row: [tx0-4, tx0+140) = nine 16-byte words (the passes read [tx0-4, tx0+132))
#define BT_PH (BT_H + 6)
__device__ __forceinline__ int reflect101(int i, int n) { i = i < 0 ? -i : i; i = i >= n ? 2 * n - 2 - i : i; return min(max(i, 0), n - 1); }
__global__ void __launch_bounds__(256)
blur_kernel(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, OrbGeom g, int nframes)
{
    // XCD-aware launch order (the tile kernels of this file share it): grid = (frames rounded up to 8, tiles), FRAME fastest.  Workgroups are dealt
    // round-robin over the 8 XCDs, so with the frame as the fast index all tiles of a frame run on ONE XCD, and tiles t, t+1, ... of a frame are
    // dispatched close in time: the apron rows / columns that neighbouring tiles share are then served by that XCD's L2 instead of HBM again
    const int frame = blockIdx.x, tile_id = blockIdx.y;
    if (frame >= nframes) return;
    __shared__ __attribute__((aligned(16))) uint8_t  in[BT_PH * BT_PW];
    __shared__ uint4 hp2[(BT_PH / 2) * (BT_W / 4)];             // row-pass sums: [row pair][pixel] = (even row | odd row << 16)
    const int tid = threadIdx.x;
    int l = 0;
    while (l + 1 < g.nlevels && tile_id >= g.L[l+1].tile_off) l++;
    const LevelGeom& L = g.L[l];
    const int t = tile_id - L.tile_off;
    const int trow = L.tiles_x == 1 ? t : (int)__umulhi((uint32_t)t, L.mulTX);   // t / tiles_x without the division sequence (every wave would run it); 2^32 / 1 does not fit the multiplier
    const int tx0 = (t - trow * L.tiles_x) * BT_W, ty0 = trow * BT_H;
    const int w = L.w, h = L.h, stride = L.stride;
    const uint8_t* src = pyr + (size_t)frame * g.pyr_bytes + L.img_off;
    // nine 16-byte words per staged row; a word that lies inside the image is one (unaligned) load, a word that crosses the left or right
    // border (two per row, in edge tiles only) is assembled from reflected bytes; rows reflect as a whole
    for (int i = tid; i < BT_PH * (BT_PW / 16); i += 256) {
        const int ly = (i * 7282) >> 16, c = i - ly * (BT_PW / 16);                 // i / 9 for i < 342
        const int gx = tx0 - 4 + 16 * c, gy = reflect101(ty0 - 3 + ly, h);
        const uint8_t* row = src + (size_t)gy * stride;
        uint4 v;
        if (gx >= 0 && gx + 16 <= w) __builtin_memcpy(&v, row + gx, 16);
        else {
            uint32_t d[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int x = gx + 4 * q;
                d[q] = (uint32_t)row[reflect101(x, w)] | ((uint32_t)row[reflect101(x + 1, w)] << 8) | ((uint32_t)row[reflect101(x + 2, w)] << 16) | ((uint32_t)row[reflect101(x + 3, w)] << 24);
            }
            v = make_uint4(d[0], d[1], d[2], d[3]);
        }
        reinterpret_cast<uint4*>(in)[i] = v;
    }
    __syncthreads();
    // row pass, two rows (2j, 2j+1) x four pixels per work item; the two rows' sums of a pixel share one dword of hp (low half = even
    // row), so that the column pass reads whole row pairs and feeds them to v_dot2_u32_u16 without re-pairing them
    for (int i = tid; i < (BT_PH / 2) * (BT_W / 4); i += 256) {
        const int j = i >> 5, lq = i & 31;
        const uint32_t TA = 18u | (34u << 8) | (49u << 16) | (55u << 24), TB = 49u | (34u << 8) | (18u << 16);
        uint32_t o[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const uint32_t* r = reinterpret_cast<const uint32_t*>(in + (2 * j + rr) * BT_PW) + lq;     // dwords at x-4, x, x+4
            const uint32_t d0 = r[0], d1 = r[1], d2 = r[2];
            // output k (pixel x + k) is the 7-tap dot product of the bytes x+k-3 .. x+k+3 = bytes k+1 .. k+7 of (d0, d1, d2):
            // two v_dot4_u32_u8 on the byte-aligned dwords (taps 18 34 49 55 | 49 34 18 0)
            o[rr][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), TA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 1), TB, 0u, false), false);
            o[rr][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), TA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), TB, 0u, false), false);
            o[rr][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), TA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 3), TB, 0u, false), false);
            o[rr][3] = __builtin_amdgcn_udot4(d1, TA, __builtin_amdgcn_udot4(d2, TB, 0u, false), false);
        }
        hp2[j * (BT_W / 4) + lq] = make_uint4(o[0][0] | (o[1][0] << 16), o[0][1] | (o[1][1] << 16), o[0][2] | (o[1][2] << 16), o[0][3] | (o[1][3] << 16));
    }
    __syncthreads();
    uint8_t* dst = blur + (size_t)frame * g.blur_bytes;                // tiled layout: blur_off()
    for (int i = tid; i < BT_H * (BT_W / 4); i += 256) {
        const int ly = i >> 5, lq = i & 31;
        const int gx = tx0 + 4 * lq, gy = ty0 + ly;
        if (gx >= stride || gy >= h) continue;
        // column pass over the staged rows ly .. ly+6 = the pairs m .. m+3 (m = ly >> 1): an even ly takes (18,34) (49,55) (49,34) (18,0) of them,
        // an odd ly (0,18) (34,49) (55,49) (34,18): four v_dot2_u32_u16 per pixel, the rounding constant rides in as the first accumulator
        typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
        auto dot2 = [](uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, a), __builtin_bit_cast(ushort2v, b), c, false); };
        const bool odd = ly & 1;
        const uint32_t T0 = odd ? (18u << 16) : (18u | (34u << 16)), T1 = odd ? (34u | (49u << 16)) : (49u | (55u << 16)),
                       T2 = odd ? (55u | (49u << 16)) : (49u | (34u << 16)), T3 = odd ? (34u | (18u << 16)) : 18u;
        const uint4* hq = hp2 + (ly >> 1) * (BT_W / 4) + lq;
        const uint4 p0 = hq[0], p1 = hq[BT_W / 4], p2 = hq[2 * (BT_W / 4)], p3 = hq[3 * (BT_W / 4)];
        const uint32_t a0 = dot2(p3.x, T3, dot2(p2.x, T2, dot2(p1.x, T1, dot2(p0.x, T0, 32768u))));
        const uint32_t a1 = dot2(p3.y, T3, dot2(p2.y, T2, dot2(p1.y, T1, dot2(p0.y, T0, 32768u))));
        const uint32_t a2 = dot2(p3.z, T3, dot2(p2.z, T2, dot2(p1.z, T1, dot2(p0.z, T0, 32768u))));
        const uint32_t a3 = dot2(p3.w, T3, dot2(p2.w, T2, dot2(p1.w, T1, dot2(p0.w, T0, 32768u))));
        // (acc >> 16) of two pixels side by side as u16, saturated to 255 by one packed min, then the four low bytes into one dword
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        const us2 lim = {255, 255};
        const us2 q01 = __builtin_elementwise_min(__builtin_bit_cast(us2, __builtin_amdgcn_perm(a1, a0, 0x07060302u)), lim);
        const us2 q23 = __builtin_elementwise_min(__builtin_bit_cast(us2, __builtin_amdgcn_perm(a3, a2, 0x07060302u)), lim);
        uint32_t out = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, q23), __builtin_bit_cast(uint32_t, q01), 0x06040200u);
        if (gx + 3 >= w) out &= gx >= w ? 0u : (0xFFFFFFFFu >> (8 * (gx + 4 - w)));       // keep the padding columns zero
        *reinterpret_cast<uint32_t*>(dst + blur_off(L.boff, stride, gx, gy)) = out;
    }
}
hipError_t k_blur(int n, const OrbGeom& g, const uint8_t* pyr, uint8_t* blur, hipStream_t s)
{
    blur_kernel<<<dim3((n + 7) & ~7, g.tiles_total), 256, 0, s>>>(pyr, blur, g, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------ K5a on the matrix cores
// The same 7x7 fixed-point Gaussian as two exact integer matrix products per 32-column unit (v_mfma_i32_32x32x32_i8):
//   row pass     R[y][x'] = sum_k (p[y][c0 - 16 + k] - 128) * T[k][x'] + 128 * 257        A = pixels (lane = image row, 16 consecutive bytes per K half),
//                                                                                       B = a banded matrix of the taps (constant per unit column; it also folds
//                                                                                       BORDER_REFLECT_101 of the columns in: a tap that falls outside the image
//                                                                                       is added to the coefficient of the column it reflects to) -> R is the exact
//                                                                                       u16 row sum, lane = column x', registers = rows
//   column pass  Z[x'][y'] = sum_i R[i][x'] * tc[i - y']                                 the accumulator tile is the A operand as it stands (X^T B form: the sum
//                                                                                       runs over its register index), split into its low and high bytes (two
//                                                                                       products, v = 256 ZH + ZL), B = the banded tap matrix in the accumulator's
//                                                                                       own row order -> lane = output row, registers = 4 x 4 consecutive columns
// Rows reflect as whole rows when the block stages its 64 input rows (58 output rows + 6) in LDS (coalesced 16-byte loads, p - 128 applied there); the
// output leaves through an LDS tile as whole 128-byte rows.  Per pixel the VALU only repacks bytes (v_perm) and shifts / saturates the result: about a
// quarter of the instructions of blur_kernel, which is VALU-issue bound.  Bit-exact with it (all sums are exact integers).
// Coefficient table (built on the host by blur_mfma_tables): [F_same: 64 x 16 B][F_next: 64 x 16 B] then per level and 32-column unit [K half s: 2][lane: 64] x 16 B.
#define BM_IN_RS 176      // bytes between staged input rows (160 used: columns [X0 - 16, X0 + 144))
#define BM_OUT_RS 144     // bytes between rows of the output tile (128 used)
typedef int bm_v4i __attribute__((ext_vector_type(4)));
typedef int bm_v16i __attribute__((ext_vector_type(16)));
size_t blur_mfma_table_bytes(const OrbGeom& g) { return (size_t)(128 + 128 * g.bt_units_total) * 16; }
void blur_mfma_tables(const OrbGeom& g, void* host_out)
{
    static const int tc[7] = {18, 34, 49, 55, 49, 34, 18};
    int8_t* o = reinterpret_cast<int8_t*>(host_out);
    auto refl = [](int i, int n) { i = i < 0 ? -i : i; i = i >= n ? 2 * n - 2 - i : i; return i < 0 ? 0 : (i >= n ? n - 1 : i); };
    for (int which = 0; which < 2; which++)                       // column pass: B[k][n], k in the accumulator's row order: element j of lane half h = row 8 (j >> 2) + 4 h + (j & 3)
        for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 16; j++) {
                const int n = lane & 31, h = lane >> 5, k = 8 * (j >> 2) + 4 * h + (j & 3) + 32 * which, d = k - n;
                o[(which * 64 + lane) * 16 + j] = (int8_t)((d >= 0 && d <= 6) ? tc[d] : 0);
            }
    for (int l = 0; l < g.nlevels; l++) {
        const LevelGeom& L = g.L[l];
        for (int u = 0; u < (L.stride + 31) / 32; u++)
            for (int s = 0; s < 2; s++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 16; j++) {
                        const int n = lane & 31, h = lane >> 5, k = 32 * s + 16 * h + j;
                        const int x_in = 32 * u - 16 + k, x_out = 32 * u + n;
                        int coef = 0;
                        if (x_out < L.w) for (int t = 0; t < 7; t++) if (refl(x_out + t - 3, L.w) == x_in) coef += tc[t];
                        o[((size_t)(128 + (L.bt_units_off + u) * 128 + s * 64 + lane)) * 16 + j] = (int8_t)coef;
                    }
    }
}
__global__ void __launch_bounds__(256, 4)                    // 128 registers: four blocks per CU
blur_mfma_kernel(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, OrbGeom g, const uint4* __restrict__ tab, int nframes)
{
    // One block = one 128-column strip of one level of one frame, walked top to bottom in steps of 58 output rows: the 64 input rows of the next step are
    // loaded (into registers) before the current step computes, so only the first step of a strip waits for memory.  blockIdx.y = strip over all levels.
    const int frame = blockIdx.x, strip = blockIdx.y;            // frame-fastest launch order: see blur_kernel
    if (frame >= nframes) return;
    __shared__ __attribute__((aligned(16))) uint8_t sin[64 * BM_IN_RS];
    __shared__ __attribute__((aligned(16))) uint8_t sout[64 * BM_OUT_RS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, hh = lane >> 5;
    int l = 0;
    while (l + 1 < g.nlevels && strip >= g.L[l+1].bt_off) l++;
    const LevelGeom& L = g.L[l];
    const int X0 = (strip - L.bt_off) * 128;
    const int w = L.w, h = L.h, stride = L.stride;
    const uint8_t* src = pyr + (size_t)frame * g.pyr_bytes + L.img_off;
    uint8_t* dst = blur + (size_t)frame * g.blur_bytes;                  // tiled layout: blur_off()
    auto ld = [](const uint4* p) { const uint4 v = *p; return bm_v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w}; };
    const int c0 = X0 + 32 * wv;
    const bool active = c0 < stride;
    const uint4* T1 = tab + 128 + (size_t)(L.bt_units_off + (active ? (c0 >> 5) : 0)) * 128 + lane;
    const bm_v4i B0 = ld(T1), B1 = ld(T1 + 64), Fs = ld(tab + lane), Fn = ld(tab + 64 + lane);
    uint32_t cmask[4];                                                   // the padding columns [w, stride) stay zero: dword q of a lane holds columns c0 + 8 q + 4 hh .. + 3
#pragma unroll
    for (int q = 0; q < 4; q++) { const int left = w - (c0 + 8 * q + 4 * hh); cmask[q] = left >= 4 ? 0xFFFFFFFFu : (left <= 0 ? 0u : (0xFFFFFFFFu >> (8 * (4 - left)))); }
    // a thread's staging slots: words i = tid, tid + 256, tid + 512 (< 640) of the 64 x 10 input words; (row, q) are the same in every step
    int srow[3], sq[3]; bool sok[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = tid + 256 * k;
        srow[k] = (i * 6554) >> 16; sq[k] = i - srow[k] * 10;             // i / 10 for i < 640
        const int gx = X0 - 16 + 16 * sq[k];
        sok[k] = i < 640 && gx >= 0 && gx < stride;                        // words outside [0, stride) are never multiplied by a non-zero coefficient
    }
    uint4 pre[3];
    auto fetch = [&](int y0) {                                            // rows y0 - 3 .. y0 + 60, reflected as whole rows
#pragma unroll
        for (int k = 0; k < 3; k++) {
            pre[k] = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
            // (a 32-bit offset from the block's uniform base: one register per staging slot instead of a 64-bit pointer.  With pointers the kernel spilled one, and a
            // spill's reload waits with vmcnt(0), i.e. for the load issued in front of it: the three prefetches of a step went out one HBM round trip apart -- round 6)
            if (sok[k]) pre[k] = *reinterpret_cast<const uint4*>(src + (uint32_t)(reflect101(y0 - 3 + srow[k], h) * stride + (X0 - 16 + 16 * sq[k])));
        }
    };
    fetch(0);
    // The coefficient tables are waited for HERE, by using them (an empty asm with the registers as inputs).  Otherwise their loads are still pending on the loop's
    // entry edge, the wait-count pass merges that into the loop header, and every step's first MFMAs wait with vmcnt(3) .. vmcnt(0) -- for the three prefetch
    // loads the step has just issued: the prefetch never ran ahead (round 6; the same pattern as sgbm_sweep8's).
    asm volatile("" :: "v"(B0), "v"(B1), "v"(Fs), "v"(Fn));
    for (int y0 = 0; y0 < h; y0 += BLUR_ROWS) {
        // stage p - 128 (the fetch was issued one step ago), then start the next step's loads
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (tid + 256 * k < 640)
                *reinterpret_cast<uint4*>(sin + srow[k] * BM_IN_RS + 16 * sq[k]) = make_uint4(pre[k].x ^ 0x80808080u, pre[k].y ^ 0x80808080u, pre[k].z ^ 0x80808080u, pre[k].w ^ 0x80808080u);
        if (y0 + BLUR_ROWS < h) fetch(y0 + BLUR_ROWS);
        __syncthreads();
        if (active) {
            const uint8_t* ap = sin + r * BM_IN_RS + 32 * wv + 16 * hh;
            bm_v16i CL, CH;                                               // 32 registers of constants: the row pass shares CH (its padding columns are masked at the end instead)
#pragma unroll
            for (int i = 0; i < 16; i++) { CL[i] = 128 * 257 + 32768; CH[i] = 128 * 257; }
            // the row sums (< 65536) as two planes of signed bytes, 16 registers -> 4 + 4: element j of the fragment = register j
            auto planes = [](const bm_v16i& R, bm_v4i& lo, bm_v4i& hi) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t a = (uint32_t)R[4*q], b = (uint32_t)R[4*q+1], c = (uint32_t)R[4*q+2], d = (uint32_t)R[4*q+3];
                    const uint32_t l01 = __builtin_amdgcn_perm(b, a, 0x0C0C0400u), l23 = __builtin_amdgcn_perm(d, c, 0x0C0C0400u);
                    const uint32_t h01 = __builtin_amdgcn_perm(b, a, 0x0C0C0501u), h23 = __builtin_amdgcn_perm(d, c, 0x0C0C0501u);
                    lo[q] = (int)(__builtin_amdgcn_perm(l23, l01, 0x05040100u) ^ 0x80808080u);
                    hi[q] = (int)(__builtin_amdgcn_perm(h23, h01, 0x05040100u) ^ 0x80808080u);
                }
            };
            // one 32-row tile of row sums at a time (fences: the second tile's fragments and accumulator re-use the first one's registers)
            bm_v4i L0, H0, L1, H1;
            {   const bm_v4i A0 = ld(reinterpret_cast<const uint4*>(ap)), A1 = ld(reinterpret_cast<const uint4*>(ap + 32));
                bm_v16i R = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, B0, CH, 0, 0, 0);
                R = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, B1, R, 0, 0, 0);
                planes(R, L0, H0); }
            __builtin_amdgcn_sched_barrier(0);
            {   const bm_v4i A0 = ld(reinterpret_cast<const uint4*>(ap + 32 * BM_IN_RS)), A1 = ld(reinterpret_cast<const uint4*>(ap + 32 * BM_IN_RS + 32));
                bm_v16i R = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, B0, CH, 0, 0, 0);
                R = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, B1, R, 0, 0, 0);
                planes(R, L1, H1); }
            __builtin_amdgcn_sched_barrier(0);
            // v = 256 ZH + ZL (the rounding constant rode in with CL); (v >> 16) saturated = byte 2 of min(v, 0xFFFFFF); registers 4q .. 4q + 3 of a lane
            // are four consecutive columns of output row (lane & 31): one dword of the output tile
            auto emit = [&](const bm_v16i& ZL, const bm_v16i& ZH, int tile) {
                uint8_t* o = sout + (32 * tile + r) * BM_OUT_RS + 32 * wv + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t v[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) v[k] = min(((uint32_t)ZH[4*q+k] << 8) + (uint32_t)ZL[4*q+k], 0xFFFFFFu);
                    const uint32_t p01 = __builtin_amdgcn_perm(v[1], v[0], 0x0C0C0602u), p23 = __builtin_amdgcn_perm(v[3], v[2], 0x0C0C0602u);
                    *reinterpret_cast<uint32_t*>(o + 8 * q) = __builtin_amdgcn_perm(p23, p01, 0x05040100u) & cmask[q];
                }
            };
            // output rows 0..31 take input rows 0..37 (both tiles), output rows 32..57 input rows 32..63 (the second tile only)
            bm_v16i ZL = __builtin_amdgcn_mfma_i32_32x32x32_i8(L0, Fs, CL, 0, 0, 0);
            bm_v16i ZH = __builtin_amdgcn_mfma_i32_32x32x32_i8(H0, Fs, CH, 0, 0, 0);
            ZL = __builtin_amdgcn_mfma_i32_32x32x32_i8(L1, Fn, ZL, 0, 0, 0);
            ZH = __builtin_amdgcn_mfma_i32_32x32x32_i8(H1, Fn, ZH, 0, 0, 0);
            emit(ZL, ZH, 0);
            __builtin_amdgcn_sched_barrier(0);                           // the second output tile re-uses the first one's registers
            ZL = __builtin_amdgcn_mfma_i32_32x32x32_i8(L1, Fs, CL, 0, 0, 0);
            ZH = __builtin_amdgcn_mfma_i32_32x32x32_i8(H1, Fs, CH, 0, 0, 0);
            emit(ZL, ZH, 1);
        }
        __syncthreads();
        for (int i = tid; i < BLUR_ROWS * 8; i += 256) {
            const int row = i >> 3, q = i & 7, gx = X0 + 16 * q, gy = y0 + row;
            if (gx < stride && gy < h) *reinterpret_cast<uint4*>(dst + blur_off(L.boff, stride, gx, gy)) = *reinterpret_cast<const uint4*>(sout + row * BM_OUT_RS + 16 * q);
        }
    }
}
hipError_t k_blur_mfma(int n, const OrbGeom& g, const uint8_t* pyr, uint8_t* blur, const void* tab, hipStream_t s)
{
    blur_mfma_kernel<<<dim3((n + 7) & ~7, g.bt_total), 256, 0, s>>>(pyr, blur, g, reinterpret_cast<const uint4*>(tab), n);
    return hipGetLastError();
}

// ------------------------------------------------------------------ K3: per-cell FAST-9/16 + NMS
// ORBextractor runs cv::FAST on every cell of a 30-px grid (sub-image = cell + 3-px margin): S = max over the 16 arcs of
// 9 of the arc-min of (ring - v) / (v - ring); corner at t iff S > t; response = S - 1; NMS keeps a corner whose
// response beats its 8 neighbours INSIDE the cell's detection region; iniThFAST first, minThFAST if the cell is empty.
// Restated for the GPU without per-cell blocks:
//   * keep(p, t)  <=>  S(p) > t  and no same-cell neighbour n with S(n) >= S(p)      (for t in {ini, min}: a neighbour
//     with S(n) >= S(p) > t is itself a corner at t), so ONE local-maximum flag serves both thresholds;
//   * a 9-arc always contains two adjacent compass points, so pixels failing that test at minThFAST are dropped before
//     scoring (about 3/4 of a textured image); survivors are compacted in LDS and scored densely;
//   * the consumer (quad-tree kernel) keeps a maximum iff S > (cellmax > iniThFAST ? iniThFAST : minThFAST)  ==  "retry the cell at minThFAST";
//   * TWO PASSES (round 3).  In a cell that has a corner at iniThFAST only maxima with S > ini survive, and a neighbour with S <= ini can neither be
//     one nor suppress one (suppression needs S(n) >= S(p) > ini), so such a cell only needs the positions that pass the quick test AT ini: pass 1
//     quick-tests, scores and emits at iniThFAST everywhere (about a third of the positions the min threshold lets through) and leaves max S per cell;
//     pass 2 -- a second launch, because a cell spans tiles -- looks at the cells its tile touches, returns at once unless one of them stayed empty
//     (cellmax <= ini), and otherwise runs the same steps at minThFAST on the positions of the empty cells only (their same-cell neighbours are in
//     the same empty cell).  What pass 2 adds to cellmax is <= ini, so the consumer's rule and the emptiness test of other pass-2 tiles are unaffected.
// One block per 128x32 tile of one level of one frame (all levels in one launch), tile + 4-px apron staged in LDS by
// dword loads.
// inclusive scan of a u32 across a wave64 (DPP Hillis-Steele inside the 16-lane rows, then row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);     // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);     // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);     // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);     // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
    return v;
}
// three-input min / max in one instruction (the compiler shares pairwise minima between neighbouring arcs instead, which costs
// twice the instructions on this pattern)
__device__ __forceinline__ int imin3(int a, int b, int c) { int r; asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int imax3(int a, int b, int c) { int r; asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int fast_S(const uint8_t* p, int st)
{
    const int v = p[0];
    int d[16];
    d[0] = p[3*st] - v;      d[1] = p[3*st+1] - v;   d[2] = p[2*st+2] - v;   d[3] = p[st+3] - v;
    d[4] = p[3] - v;         d[5] = p[-st+3] - v;    d[6] = p[-2*st+2] - v;  d[7] = p[-3*st+1] - v;
    d[8] = p[-3*st] - v;     d[9] = p[-3*st-1] - v;  d[10] = p[-2*st-2] - v; d[11] = p[-st-3] - v;
    d[12] = p[-3] - v;       d[13] = p[st-3] - v;    d[14] = p[2*st-2] - v;  d[15] = p[3*st-1] - v;
    // min / max over every 9-arc as (3 of 3): 4 x 16 three-input ops, then the max over the 16 arcs
    int mn3[16], mx3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn3[k] = imin3(d[k], d[(k+1)&15], d[(k+2)&15]); mx3[k] = imax3(d[k], d[(k+1)&15], d[(k+2)&15]); }
    int mn9[16], mx9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn9[k] = imin3(mn3[k], mn3[(k+3)&15], mn3[(k+6)&15]); mx9[k] = imax3(mx3[k], mx3[(k+3)&15], mx3[(k+6)&15]); }
    // bright ring: max over arcs of min(ring - v); dark ring: max over arcs of min(v - ring) = -(min over arcs of max(ring - v))
    int bright = imax3(mn9[0], mn9[1], mn9[2]), dark = imin3(mx9[0], mx9[1], mx9[2]);
#pragma unroll
    for (int k = 3; k < 15; k += 2) { bright = imax3(bright, mn9[k], mn9[k+1]); dark = imin3(dark, mx9[k], mx9[k+1]); }
    bright = max(bright, mn9[15]); dark = min(dark, mx9[15]);
    return max(bright, -dark);
}
// The same score for TWO positions per lane.  The ring differences of position A live in the low half of a dword and those of B in the high half,
// as f16 DENORMALS: the bit pattern n (0..255) is the half-precision number n * 2^-24, sums and differences of such numbers are exact (|n| < 1024) and
// kernels run with f16 denormals enabled, so v_pk_add_f16 is an exact packed 16-bit subtract whose results order like the integers -- and gfx950 has the
// three-input packed v_pk_minimum3_f16 / v_pk_maximum3_f16 (there is no three-input packed INTEGER min / max): one instruction does what two v_min3_i32
// did.  Result: S_A | S_B << 16 (each clamped at 0 like fast_S's callers do).
__device__ __forceinline__ uint32_t pk_min3h(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ uint32_t pk_max3h(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ uint32_t pk_addh(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t pk_maxh(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t pk_minh(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_min_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t fast_S2(const uint8_t* pa, const uint8_t* pb, int st)
{
#define RING2(o) ((uint32_t)pa[o] | ((uint32_t)pb[o] << 16))
    const uint32_t negc = RING2(0) ^ 0x80008000u;                      // -(centre) in both halves
    uint32_t d[16];
    d[0] = pk_addh(RING2(3*st), negc);      d[1] = pk_addh(RING2(3*st+1), negc);   d[2] = pk_addh(RING2(2*st+2), negc);   d[3] = pk_addh(RING2(st+3), negc);
    d[4] = pk_addh(RING2(3), negc);         d[5] = pk_addh(RING2(-st+3), negc);    d[6] = pk_addh(RING2(-2*st+2), negc);  d[7] = pk_addh(RING2(-3*st+1), negc);
    d[8] = pk_addh(RING2(-3*st), negc);     d[9] = pk_addh(RING2(-3*st-1), negc);  d[10] = pk_addh(RING2(-2*st-2), negc); d[11] = pk_addh(RING2(-st-3), negc);
    d[12] = pk_addh(RING2(-3), negc);       d[13] = pk_addh(RING2(st-3), negc);    d[14] = pk_addh(RING2(2*st-2), negc);  d[15] = pk_addh(RING2(3*st-1), negc);
#undef RING2
    uint32_t mn3[16], mx3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn3[k] = pk_min3h(d[k], d[(k+1)&15], d[(k+2)&15]); mx3[k] = pk_max3h(d[k], d[(k+1)&15], d[(k+2)&15]); }
    uint32_t mn9[16], mx9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn9[k] = pk_min3h(mn3[k], mn3[(k+3)&15], mn3[(k+6)&15]); mx9[k] = pk_max3h(mx3[k], mx3[(k+3)&15], mx3[(k+6)&15]); }
    uint32_t bright = pk_max3h(mn9[0], mn9[1], mn9[2]), dark = pk_min3h(mx9[0], mx9[1], mx9[2]);
#pragma unroll
    for (int k = 3; k < 15; k += 2) { bright = pk_max3h(bright, mn9[k], mn9[k+1]); dark = pk_min3h(dark, mx9[k], mx9[k+1]); }
    bright = pk_maxh(bright, mn9[15]); dark = pk_minh(dark, mx9[15]);
    // max(bright, -dark, +0): a positive denormal's bits ARE the integer
    return pk_max3h(bright, dark ^ 0x80008000u, 0u);
}
#define FT_W 128
#define FT_H 32
#define FT_PW (FT_W + 16)     // staged pixel row [tx0 - 4, tx0 + 140) = nine 16-byte words: 4-px apron each side (3 for the ring + 1 for the NMS neighbours)
#define FT_PH (FT_H + 8)
#define FT_SW (FT_W + 2)      // scored positions: tile + 1
#define FT_SH (FT_H + 2)
#define FT_SST 132
#define FT_STAGE ((FT_PH * FT_PW) / 8)   // candidates staged per tile: as many as fit in the pixel tile they replace (680)
template <int PASS>
__device__ __forceinline__ void fast_tile(const int frame, const int tile_id, const uint8_t* __restrict__ pyr, const OrbGeom& g, cand_t* __restrict__ cand,
                                          int32_t* __restrict__ ncand, int32_t* __restrict__ cellmax, int stage_cap)
{
    __shared__ uint32_t emptyrow[8];                             // pass 2: bit j of word i = cell (cy0 + i, cx0 + j) had no corner at iniThFAST
    __shared__ __attribute__((aligned(16))) uint8_t px[FT_PH * FT_PW];
    __shared__ __attribute__((aligned(16))) uint8_t sc[(FT_SH * FT_SST + 15) / 16 * 16];
    __shared__ uint16_t list[FT_SW * FT_SH];
    __shared__ int16_t cellx[FT_SW], celly[FT_SH];
    __shared__ int lmax[64];
    // occupancy is what this kernel lives on (LDS-limited: 27.5 KB gave 5 blocks per CU, 19.4 KB gives 8 = the 32-wave limit; 3.42 -> 2.7 us/frame),
    // so the candidate staging area reuses the pixel tile: px is dead once every position is scored, and the NMS pass that fills
    // `stage` starts behind the barrier that ends the scoring loop
    cand_t* stage = reinterpret_cast<cand_t*>(px);
    __shared__ int nlist, nsurv, nstage, gbase;
    const int tid = threadIdx.x, lane = tid & 63;
    int l = 0;
    while (l + 1 < g.nlevels && tile_id >= g.L[l+1].ftile_off) l++;
    const LevelGeom& L = g.L[l];
    const int t = tile_id - L.ftile_off;
    const int trow = L.ftiles_x == 1 ? t : (int)__umulhi((uint32_t)t, L.fmulTX);
    const int tx0 = SSM_EDGE + (t - trow * L.ftiles_x) * FT_W, ty0 = SSM_EDGE + trow * FT_H;      // the grid starts at the first position FAST may report
    const int w = L.w, h = L.h, stride = L.stride;
    const uint8_t* im = pyr + (size_t)frame * g.pyr_bytes + L.img_off;
    if (PASS == 2) {
        // the cells this tile touches: cells of its first and last scored position in x and y (at most 8 x 8: build_geometry)
        auto cell_of = [](int gpos, int origin, uint32_t mul) { const int v = gpos - origin - 3; return v >= 0 ? (int)__umulhi((uint32_t)v, mul) : 0; };
        const int cx0 = cell_of(tx0, L.minBX, L.mulW), cy0 = cell_of(ty0, L.minBY, L.mulH);
        const int cx1 = min(cell_of(min(tx0 + FT_W - 1, w - 1), L.minBX, L.mulW), L.nCols - 1), cy1 = min(cell_of(min(ty0 + FT_H - 1, h - 1), L.minBY, L.mulH), L.nRows - 1);
        bool mine = false;
        if (tid < 64) {
            const int ci = cy0 + (tid >> 3), cj = cx0 + (tid & 7);
            if (ci <= cy1 && cj <= cx1) mine = cellmax[(size_t)frame * g.cells_total + L.cell_off + ci * L.nCols + cj] <= g.ini_th;
        }
        const unsigned long long bal = __ballot(mine);           // wave 0 holds all 64 cells
        if (tid < 8) emptyrow[tid] = (uint32_t)((bal >> (8 * tid)) & 0xFFull);
        if (!__syncthreads_or(mine ? 1 : 0)) return;             // every cell of the tile has its corners from pass 1 (block-uniform)
    }
    // ---- stage the tile: nine (unaligned) 16-byte loads per row, 360 per tile.  Rows / words outside the image are CLAMPED into it instead of
    // zero-filled: they then hold shifted pixels, which no valid position ever looks at (valid positions sit >= 19 px from every border, the ring and
    // the NMS neighbours reach 4), and no address leaves the level image
    for (int i = tid; i < FT_PH * (FT_PW / 16); i += 256) {
        const int ly = (i * 7282) >> 16, c = i - ly * (FT_PW / 16);                 // i / 9 for i < 360
        const int gy = min(max(ty0 - 4 + ly, 0), h - 1), gx = min(max(tx0 - 4 + 16 * c, 0), stride - 16);
        uint4 v; __builtin_memcpy(&v, im + (size_t)gy * stride + gx, 16);
        reinterpret_cast<uint4*>(px)[i] = v;
    }
    for (int i = tid; i < (FT_SH * FT_SST + 15) / 16; i += 256) reinterpret_cast<uint4*>(sc)[i] = make_uint4(0, 0, 0, 0);
    // (cell index by multiplication with the host's reciprocal: the two integer divisions cost every wave ~70 instructions)
    if (tid < FT_SW) { const int gx = tx0 + tid - 1 - L.minBX - 3; cellx[tid] = (int16_t)(gx >= 0 ? (int)__umulhi((uint32_t)gx, L.mulW) : -1); }
    if (tid < FT_SH) { const int gy = ty0 + tid - 1 - L.minBY - 3; celly[tid] = (int16_t)(gy >= 0 ? (int)__umulhi((uint32_t)gy, L.mulH) : -1); }
    if (tid < 64) lmax[tid] = 0;
    if (tid == 0) { nlist = 0; nsurv = 0; nstage = 0; gbase = 0; }
    __syncthreads();
    // ---- quick reject + compaction of the positions worth scoring.  A position passes when two ADJACENT compass points of
    // the ring (N, E, S, W at distance 3) are both brighter than v + t or both darker than v - t (necessary for a 9-arc).
    // Four horizontally adjacent positions per thread: five dword LDS reads (the centre dword, its left/right neighbours and
    // the dwords 3 rows up/down), bytes widened to packed u16 pairs, then packed 16-bit min/max:
    //   bright = max over adjacent pairs of min(x - v, y - v),  dark = max over pairs of min(v - x, v - y) = -min over pairs of max
    // thread = (group of 4 columns g = tid & 31, row tid >> 5 + 8k): no division, conflict-free rows.
    const int min_th = PASS == 1 ? g.ini_th : g.min_th;         // the threshold of this pass
    const int pcx0 = max((int)cellx[1], 0), pcy0 = max((int)celly[1], 0);
    {
        const uint32_t* pxw = reinterpret_cast<const uint32_t*>(px);
        const int gq = tid & 31, r0 = tid >> 5;
        int cjs[4];                                                     // pass 2: the cell columns of this thread's four positions, relative to the tile's first cell
#pragma unroll
        for (int j = 0; j < 4; j++) cjs[j] = PASS == 2 ? ((int)cellx[4 * gq + 1 + j] - pcx0) & 7 : 0;
        const uint32_t t1 = (uint32_t)(min_th + 1) * 0x00010001u;
        const int gx0 = tx0 + 4 * gq;                                   // first of the 4 positions; sx = 4 gq + 1 + j
        unsigned xvalid = 0;                                            // positions inside the FAST window of the level, in x (the same for every row)
#pragma unroll
        for (int j = 0; j < 4; j++) if (gx0 + j >= SSM_EDGE && gx0 + j < w - SSM_EDGE) xvalid |= 1u << j;
#pragma unroll 1
        for (int sy = r0; sy < FT_SH; sy += 8) {
            const int gy = ty0 + sy - 1;
            const int rowc = (sy + 3) * (FT_PW / 4) + gq + 1;
            const uint32_t C = pxw[rowc], P = pxw[rowc - 1], Nx = pxw[rowc + 1];
            const uint32_t U = pxw[rowc - 3 * (FT_PW / 4)], D = pxw[rowc + 3 * (FT_PW / 4)];
            const uint32_t Lw = __builtin_amdgcn_alignbyte(C, P, 1);     // pixels x-3 of the four positions
            const uint32_t Rw = __builtin_amdgcn_alignbyte(Nx, C, 3);    // pixels x+3
            unsigned passbits = 0;
#pragma unroll
            for (int hpair = 0; hpair < 2; hpair++) {
                const uint32_t sel = hpair ? 0x0C030C02u : 0x0C010C00u;  // bytes (2h, 2h+1) zero-extended to u16 pairs
                typedef short short2v __attribute__((ext_vector_type(2)));
                auto widen = [&](uint32_t w) { uint32_t r = __builtin_amdgcn_perm(0u, w, sel); short2v o; __builtin_memcpy(&o, &r, 4); return o; };
                const short2v c = widen(C);
                const short2v a = widen(D) - c, bq = widen(Rw) - c, cq = widen(U) - c, d = widen(Lw) - c;
                // max over the four adjacent pairs of the cycle a-b-c-d of min(x, y) = min(max(a, c), max(b, d)): "some adjacent pair is
                // above the threshold" is (A or C) and (B or D); likewise the min over pairs of max = max(min(a, c), min(b, d))
                const short2v br = __builtin_elementwise_min(__builtin_elementwise_max(a, cq), __builtin_elementwise_max(bq, d));
                const short2v dk = __builtin_elementwise_max(__builtin_elementwise_min(a, cq), __builtin_elementwise_min(bq, d));
                short2v t1v; __builtin_memcpy(&t1v, &t1, 4);
                const short2v e = __builtin_elementwise_max(br, (short2v)(0 - dk)) - t1v;      // >= 0  <=>  pass
                uint32_t eb; __builtin_memcpy(&eb, &e, 4);
                passbits |= ((~eb >> 15) & 1u) << (2 * hpair);
                passbits |= ((~eb >> 31) & 1u) << (2 * hpair + 1);
            }
            // positions outside the FAST window of the level never pass
            passbits &= (gy >= SSM_EDGE && gy < h - SSM_EDGE) ? xvalid : 0u;
            if (PASS == 2) {                                            // only the cells that stayed empty at iniThFAST are retried
                const uint32_t er = emptyrow[((int)celly[sy] - pcy0) & 7];
                unsigned m = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) m |= ((er >> cjs[j]) & 1u) << j;
                passbits &= m;
            }
            // compaction: wave scan of the per-thread counts, one LDS reservation per wave
            const uint32_t cntp = __popc(passbits);
            const uint32_t incl = wave_incl_scan_u32(cntp);
            const uint32_t tot = __shfl(incl, 63, 64);
            if (tot) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&nlist, (int)tot);
                base = __shfl(base, 0, 64) + (int)(incl - cntp);
#pragma unroll
                for (int j = 0; j < 4; j++) if (passbits & (1u << j)) list[base++] = (uint16_t)((sy << 8) | (4 * gq + 1 + j));
            }
        }
        // the two apron columns sx = 0 and sx = FT_W + 1 (scored only as NMS neighbours)
        if (tid < 2 * FT_SH) {
            const int sy = tid >> 1, sx = (tid & 1) ? FT_W + 1 : 0;
            const int gx = tx0 + sx - 1, gy = ty0 + sy - 1;
            if (gx >= SSM_EDGE && gx < w - SSM_EDGE && gy >= SSM_EDGE && gy < h - SSM_EDGE &&
                (PASS == 1 || ((emptyrow[((int)celly[sy] - pcy0) & 7] >> (((int)cellx[sx] - pcx0) & 7)) & 1u))) {
                const uint8_t* p = &px[(sy + 3) * FT_PW + sx + 3];
                const int v = p[0];
                const int a = p[3 * FT_PW] - v, b = p[3] - v, c = p[-3 * FT_PW] - v, d = p[-3] - v;
                const int br = min(max(a, c), max(b, d));
                const int dk = max(min(a, c), min(b, d));
                if (max(br, -dk) > min_th) list[atomicAdd(&nlist, 1)] = (uint16_t)((sy << 8) | sx);
            }
        }
    }
    __syncthreads();
    const int n = nlist;
    // ---- score the listed positions, two per lane (fast_S2), and compact the ones that can become keypoints (S > minThFAST, inside the tile and the
    // image) to the front of the same list: a chunk's entries are all read before the barrier, survivors are written behind it into slots below the
    // chunk's end (there are never more survivors than entries processed), so the NMS pass below walks ~10 % of the positions instead of ~23 %
    for (int e0 = 0; e0 < n; e0 += 512) {
        const int ea = e0 + tid, eb = e0 + 256 + tid;
        const int ia = list[min(ea, n - 1)], ib = list[min(eb, n - 1)];
        const int sya = ia >> 8, sxa = ia & 255, syb = ib >> 8, sxb = ib & 255;
        const uint32_t S2 = fast_S2(&px[(sya + 3) * FT_PW + sxa + 3], &px[(syb + 3) * FT_PW + sxb + 3], FT_PW);
        const int Sa = (int)(S2 & 0xFFFFu), Sb = (int)(S2 >> 16);
        if (ea < n) sc[sya * FT_SST + sxa] = (uint8_t)Sa;
        if (eb < n) sc[syb * FT_SST + sxb] = (uint8_t)Sb;
        const bool ka = ea < n && Sa > min_th && sxa >= 1 && sxa <= FT_W && sya >= 1 && sya <= FT_H && tx0 + sxa - 1 < w && ty0 + sya - 1 < h;
        const bool kb = eb < n && Sb > min_th && sxb >= 1 && sxb <= FT_W && syb >= 1 && syb <= FT_H && tx0 + sxb - 1 < w && ty0 + syb - 1 < h;
        __syncthreads();                                                // every entry of the chunk has been read
        const unsigned long long ba = __ballot(ka), bb = __ballot(kb);
        const int ca = __popcll(ba), cb = __popcll(bb);
        if (ca + cb) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&nsurv, ca + cb);
            base = __shfl(base, 0, 64);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (ka) list[base + __popcll(ba & below)] = (uint16_t)ia;
            if (kb) list[base + ca + __popcll(bb & below)] = (uint16_t)ib;
        }
    }
    __syncthreads();
    const int n2 = nsurv;
    // ---- local maxima among same-cell neighbours; staged in LDS, ONE global reservation per tile
    const int cx0 = max((int)cellx[1], 0), cy0 = max((int)celly[1], 0);
    cand_t* out = cand + (size_t)frame * g.cand_total + L.cand_off;
    int32_t* nc = ncand + frame * g.nlevels + l;
    for (int e0 = 0; e0 < n2; e0 += 256) {
        const int e = e0 + tid;
        bool keep = false; int S = 0, sx = 0, sy = 0;
        if (e < n2) {
            const int i = list[e]; sy = i >> 8; sx = i & 255;
            {
                const uint8_t* q = &sc[sy * FT_SST + sx];
                S = q[0];
                {
                    const int cx = cellx[sx], cy = celly[sy];
                    const bool xl = cellx[sx-1] == cx, xr = cellx[sx+1] == cx, yu = celly[sy-1] == cy, yd = celly[sy+1] == cy;
                    int nb = 0;
                    if (yu) { nb = max(nb, (int)q[-FT_SST]); if (xl) nb = max(nb, (int)q[-FT_SST-1]); if (xr) nb = max(nb, (int)q[-FT_SST+1]); }
                    if (yd) { nb = max(nb, (int)q[FT_SST]);  if (xl) nb = max(nb, (int)q[FT_SST-1]);  if (xr) nb = max(nb, (int)q[FT_SST+1]); }
                    if (xl) nb = max(nb, (int)q[-1]);
                    if (xr) nb = max(nb, (int)q[1]);
                    keep = nb < S;
                }
            }
        }
        const unsigned long long bal = __ballot(keep);
        if (bal) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&nstage, __popcll(bal));
            base = __shfl(base, 0, 64);
            if (keep) {
                const int k = base + __popcll(bal & ((1ull << lane) - 1ull));
                const int gx = tx0 + sx - 1, gy = ty0 + sy - 1;
                const int cj = cellx[sx], ci = celly[sy];
                atomicMax(&lmax[((ci - cy0) & 7) * 8 + ((cj - cx0) & 7)], S);
                cand_t c;
                c.x = (uint32_t)(gx - L.minBX) | ((uint32_t)(gy - L.minBY) << 12) | ((uint32_t)(S - 1) << 24);
                c.y = ((uint32_t)(ci * L.nCols + cj) << 14) | ((uint32_t)(gy - L.minBY - ci * L.hCell) << 7) | (uint32_t)(gx - L.minBX - cj * L.wCell);
                if (k < stage_cap) stage[k] = c;
                else { const int kg = atomicAdd(nc, 1); if (kg < L.cand_cap) out[kg] = c; }     // tile with > FT_STAGE maxima: rare
            }
        }
    }
    __syncthreads();
    const int ns = min(nstage, stage_cap);
    if (tid == 0 && ns) gbase = atomicAdd(nc, ns);
    __syncthreads();
    for (int k = tid; k < ns; k += 256) if (gbase + k < L.cand_cap) out[gbase + k] = stage[k];
    if (tid < 64 && lmax[tid] > 0) {
        const int ci = cy0 + (tid >> 3), cj = cx0 + (tid & 7);
        atomicMax(&cellmax[(size_t)frame * g.cells_total + L.cell_off + ci * L.nCols + cj], lmax[tid]);
    }
}
// pass 1: one block per (frame, tile)
__global__ void __launch_bounds__(256)
fast_kernel(const uint8_t* __restrict__ pyr, OrbGeom g, cand_t* __restrict__ cand, int32_t* __restrict__ ncand, int32_t* __restrict__ cellmax, int stage_cap, int nframes)
{
    const int frame = blockIdx.x, tile_id = blockIdx.y;          // frame-fastest launch order: see blur_kernel
    if (frame >= nframes) return;
    fast_tile<1>(frame, tile_id, pyr, g, cand, ncand, cellmax, stage_cap);
}
// which (frame, tile) pairs touch a cell that pass 1 left empty?  One thread per pair -> work list (tile << 16 | frame), count in work[-1].
// A pair costs a thread here instead of a block in the retry launch: on a textured stream most tiles need no retry.
__global__ void __launch_bounds__(256)
fast_need_kernel(OrbGeom g, const int32_t* __restrict__ cellmax, int nframes, int32_t* __restrict__ work)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int frame = i % ((nframes + 7) & ~7), tile_id = i / ((nframes + 7) & ~7);      // frame-fastest, like the tile launch
    bool need = false;
    if (frame < nframes && tile_id < g.ftiles_total) {
        int l = 0;
        while (l + 1 < g.nlevels && tile_id >= g.L[l+1].ftile_off) l++;
        const LevelGeom& L = g.L[l];
        const int t = tile_id - L.ftile_off;
        const int trow = L.ftiles_x == 1 ? t : (int)__umulhi((uint32_t)t, L.fmulTX);
        const int tx0 = SSM_EDGE + (t - trow * L.ftiles_x) * FT_W, ty0 = SSM_EDGE + trow * FT_H;      // the grid starts at the first position FAST may report
        auto cell_of = [](int gpos, int origin, uint32_t mul) { const int v = gpos - origin - 3; return v >= 0 ? (int)__umulhi((uint32_t)v, mul) : 0; };
        const int cx0 = cell_of(tx0, L.minBX, L.mulW), cy0 = cell_of(ty0, L.minBY, L.mulH);
        const int cx1 = min(cell_of(min(tx0 + FT_W - 1, L.w - 1), L.minBX, L.mulW), L.nCols - 1), cy1 = min(cell_of(min(ty0 + FT_H - 1, L.h - 1), L.minBY, L.mulH), L.nRows - 1);
        const int32_t* cm = cellmax + (size_t)frame * g.cells_total + L.cell_off;
        for (int ci = cy0; ci <= cy1 && !need; ci++) for (int cj = cx0; cj <= cx1; cj++) if (cm[ci * L.nCols + cj] <= g.ini_th) { need = true; break; }
    }
    const unsigned long long bal = __ballot(need);
    if (bal) {
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0) base = atomicAdd(work - 1, __popcll(bal));
        base = __shfl(base, 0, 64);
        if (need) work[base + __popcll(bal & ((1ull << lane) - 1ull))] = (tile_id << 16) | frame;
    }
}
// pass 2: a fixed grid walks the work list
__global__ void __launch_bounds__(256)
fast_retry_kernel(const uint8_t* __restrict__ pyr, OrbGeom g, cand_t* __restrict__ cand, int32_t* __restrict__ ncand, int32_t* __restrict__ cellmax, int stage_cap,
                  const int32_t* __restrict__ work)
{
    const int nwork = work[-1];
    for (int i = blockIdx.x; i < nwork; i += gridDim.x) {
        const int item = work[i];
        fast_tile<2>(item & 65535, item >> 16, pyr, g, cand, ncand, cellmax, stage_cap);
        __syncthreads();                                         // the next item re-initialises the tile's LDS state
    }
}
size_t k_fast_cellmax_ints(int nframes, const OrbGeom& g) { return (size_t)nframes * (g.cells_total + g.ftiles_total) + 16; }
size_t k_fast_ncand_pad(int nframes, const OrbGeom& g) { return ((size_t)nframes * g.nlevels + 63) & ~(size_t)63; }
hipError_t k_fast(int n, const OrbGeom& g, const uint8_t* pyr, cand_t* cand, int32_t* ncand, int32_t* cellmax, hipStream_t s)
{
    // zero: the candidate counters, the n frames' cell maxima and the retry list's length (the word at work - 1, right behind the maxima).  The context allocates the
    // counters in FRONT of the maxima (k_fast_ncand_pad ints, one allocation), so that this is one fill; any other layout takes the two fills
    hipError_t e;
    const ptrdiff_t gap = cellmax - ncand;
    // (a whole number of 16-byte words: the runtime splits any other size into two fill kernels, 4.7 us each in a per-frame call; the up to three extra ints are the
    // head of the retry list, which fast_need_kernel writes afterwards)
    if (gap >= (ptrdiff_t)n * g.nlevels && gap <= (ptrdiff_t)1 << 24) e = hipMemsetAsync(ncand, 0, sizeof(int32_t) * (((size_t)gap + (size_t)n * g.cells_total + 4 + 3) & ~(size_t)3), s);
    else { e = hipMemsetAsync(ncand, 0, sizeof(int32_t) * n * g.nlevels, s); if (e == hipSuccess) e = hipMemsetAsync(cellmax, 0, sizeof(int32_t) * ((size_t)n * g.cells_total + 4), s); }
    if (e != hipSuccess) return e;
    // SSM_FAST_STAGE_CAP (tests): a smaller staging area forces the per-candidate global path that tiles with more than FT_STAGE maxima take
    static const int stage_cap = [] { const char* e = getenv("SSM_FAST_STAGE_CAP"); const int v = e ? atoi(e) : FT_STAGE; return v < 0 ? 0 : (v > FT_STAGE ? FT_STAGE : v); }();
    // the retry work list lives behind the n frames' cell maxima (the buffer is sized for it: k_fast_cellmax_ints)
    int32_t* work = cellmax + (size_t)n * g.cells_total + 4;
    const int n8 = (n + 7) & ~7;
    fast_kernel<<<dim3(n8, g.ftiles_total), 256, 0, s>>>(pyr, g, cand, ncand, cellmax, stage_cap, n);
    fast_need_kernel<<<(n8 * g.ftiles_total + 255) / 256, 256, 0, s>>>(g, cellmax, n, work);
    const int pairs = n * g.ftiles_total;
    fast_retry_kernel<<<pairs < 4096 ? pairs : 4096, 256, 0, s>>>(pyr, g, cand, ncand, cellmax, stage_cap, work);
    return hipGetLastError();
}

// ------------------------------------------------------------------ K4: ORBextractor::DistributeOctTree
// One block per (level, frame).  Keys stay in global scratch (node id per key in node_of); the node LIST lives in LDS as
// an array in std::list order and is rebuilt once per pass by lane 0, while the per-key work (child counting,
// re-labelling, best-response selection) is spread over the block.  Equivalence with the literal std::list code of
// oracle/orb.c: a pass splits parents p_1..p_m in processing order (list order in the first phase, (size desc, newest
// first) in the second, stopping when the list reaches N); push_front of n1..n4 then erase(parent) leaves
//   [n4..n1 of p_m] ... [n4..n1 of p_1] ++ (old list without the split parents).
struct QNode { short x0, y0, x1, y1; };
#define DROPPED 0xFFFF
// NODES = LDS node capacity (512 covers 2000 features / 8 levels).  The node id of every key (2 B) lives in LDS when the level has <= OT_KCAP candidates;
// the key words are read from global memory (coalesced, L2-resident, four in flight per thread).  What is kept in LDS is sized so that EIGHT blocks fit a CU at
// NODES = 256 (20.0 KB): the blocks of a launch all take about the same time whatever their level (a split pass is a chain of barriers and LDS latencies:
// 40 - 52 us per block, scripts/octree_prof.py), so a launch lasts ceil(blocks / resident slots) block times -- with the key words and 4096 node ids in
// LDS (40.5 KB, 4 blocks per CU) the 2000 blocks of a 250-frame batch took two rounds.
#define OT_KCAP 2048
// exclusive scan over the OT_T threads of a block (thread order), `total` = the block sum; sw = OT_T/64 words of LDS.
// Contains two barriers; every thread of the block must call it.
#define OT_T 256                              // threads per octree block (1024 measured 6 % slower: the passes are barrier chains)
#define OT_UN 4                               // keys per thread and trip of the two sweeps of a split pass
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* sw, uint32_t& total)
{
    const uint32_t inc = wave_incl_scan_u32(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) sw[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < OT_T / 64; k++) { const uint32_t x = sw[k]; tot += x; if (k < w) base += x; }
    __syncthreads();
    total = tot;
    return inc - v + base;
}
#ifdef SSM_OT_PROF
__device__ unsigned long long ot_prof[8][8];              // [level][section]: shader clocks of thread 0, summed over blocks; section 7 = passes
#define OTP_T0 long long otp_ = clock64();
#define OTP(k) { const long long n_ = clock64(); if (threadIdx.x == 0) atomicAdd(&ot_prof[blockIdx.y & 7][k], (unsigned long long)(n_ - otp_)); otp_ = n_; }
#define OTP_CNT(k) { if (threadIdx.x == 0) atomicAdd(&ot_prof[blockIdx.y & 7][k], 1ull); }
#else
#define OTP_T0
#define OTP(k)
#define OTP_CNT(k)
#endif
template <int NODES>
__global__ void __launch_bounds__(OT_T, NODES == 256 ? 8 : 1)      // NODES = 256: eight waves per SIMD (64 registers), so that eight 20 KB blocks share a CU
octree_kernel(OrbGeom g, const cand_t* __restrict__ cand, const int32_t* __restrict__ ncand, const int32_t* __restrict__ cellmax,
              uint16_t* __restrict__ node_of, uint32_t* __restrict__ sel, int32_t* __restrict__ nsel, int32_t* __restrict__ status)
{
    __shared__ QNode    nd[2][NODES];
    __shared__ uint32_t cnt[2][NODES];
    __shared__ uint32_t sq[2][NODES];
    __shared__ __attribute__((aligned(8))) uint32_t cc[NODES][4];
    __shared__ short    newpos[NODES];
    __shared__ __attribute__((aligned(8))) short childpos[NODES][4];
    __shared__ short    order[NODES];
    unsigned long long* best = reinterpret_cast<unsigned long long*>(&cc[0][0]);   // final selection only, when the split passes (cc) are over
    __shared__ uint16_t lnof[OT_KCAP];
    __shared__ uint16_t cumn[NODES];
    __shared__ uint32_t sscan[OT_T / 64];
    __shared__ int sL, sFinish, sMode, sErr, sValid, sFirst;
    const int l = blockIdx.y, f = blockIdx.x, tid = threadIdx.x;        // level-major launch order: the long level-0 blocks start first
    OTP_T0
    const LevelGeom& L = g.L[l];
    const int N = L.nfeat;
    int nc = ncand[f * g.nlevels + l];
    if (nc > L.cand_cap) { nc = L.cand_cap; if (tid == 0) atomicOr(status, 1); }
    const cand_t* gkeys = cand + (size_t)f * g.cand_total + L.cand_off;
    uint16_t* gnof = node_of + (size_t)f * g.cand_total + L.cand_off;
    const bool in_lds = nc <= OT_KCAP;                      // block-uniform
    // the position + response word of the thread's own keys (i = tid + u * OT_T) stays in registers over the passes: a sweep then makes no global load
    constexpr int KPT = OT_KCAP / OT_T;
    uint32_t kxr[KPT];
#pragma unroll
    for (int u = 0; u < KPT; u++) kxr[u] = 0u;
#define KEY(i) (gkeys[i])                                /* both words: key load and final selection only (global, coalesced) */
#define KEYX(i) (gkeys[i].x)                             /* the word the split passes work on (position + response) */
#define NOF(i) (in_lds ? lnof[i] : gnof[i])
#define SETNOF(i, v) do { if (in_lds) lnof[i] = (uint16_t)(v); else gnof[i] = (uint16_t)(v); } while (0)
    uint32_t* out = sel + (size_t)f * g.sel_total + L.sel_off;
    if (nc == 0) { if (tid == 0) nsel[f * g.nlevels + l] = 0; return; }
    // ---- root nodes
    const int nIni = L.nIni; const float hX = L.hX;
    for (int i = tid; i < nIni; i += OT_T) {
        QNode q; q.x0 = (short)(int)(hX * (float)i); q.y0 = 0; q.x1 = (short)(int)(hX * (float)(i + 1)); q.y1 = (short)(L.maxBY - L.minBY);
        nd[0][i] = q; cnt[0][i] = 0; sq[0][i] = i;
    }
    if (tid == 0) { sErr = 0; sValid = 0; }
    __syncthreads();
    // keys are the local maxima at minThFAST; a cell that has one above iniThFAST keeps only those (DROPPED otherwise)
    const int32_t* cm = cellmax + (size_t)f * g.cells_total + L.cell_off;
    int nvalid = 0;
    // (four keys per trip: their key loads, then their cell-maximum loads, issued together -- two dependent global latencies per trip, not per key)
    auto init_trip = [&](auto lds_tag, int i0, cand_t (&k)[OT_UN]) {
        constexpr bool in_lds = decltype(lds_tag)::value;      // (shadows the block-uniform flag: a pointer chosen at run time made every node-id access a flat_ instruction)
        int cmv[OT_UN];
#pragma unroll
        for (int u = 0; u < OT_UN; u++) { const int i = i0 + u * OT_T; k[u] = i < nc ? KEY(i) : make_uint2(0u, 0u); }
#pragma unroll
        for (int u = 0; u < OT_UN; u++) cmv[u] = i0 + u * OT_T < nc ? cm[k[u].y >> 14] : 0;
#pragma unroll
        for (int u = 0; u < OT_UN; u++) {
            const int i = i0 + u * OT_T;
            if (i >= nc) continue;
            const int S = (int)(k[u].x >> 24) + 1;
            const int th = cmv[u] > g.ini_th ? g.ini_th : g.min_th;
            if (S <= th) { SETNOF(i, DROPPED); continue; }
            const int x = k[u].x & 4095;
            int b = (int)((float)x / hX); b = min(b, nIni - 1);
            SETNOF(i, b); atomicAdd(&cnt[0][b], 1u); nvalid++;
        }
    };
    if (in_lds) {
#pragma unroll
        for (int t = 0; t < KPT / OT_UN; t++) {
            const int i0 = tid + t * OT_UN * OT_T;
            if (i0 < nc) { cand_t k[OT_UN]; init_trip(std::true_type{}, i0, k); for (int u = 0; u < OT_UN; u++) kxr[t * OT_UN + u] = k[u].x; }
        }
    } else {
        for (int i0 = tid; i0 < nc; i0 += OT_UN * OT_T) { cand_t k[OT_UN]; init_trip(std::false_type{}, i0, k); }
    }
    if (nvalid) atomicAdd(&sValid, nvalid);
    __syncthreads();
    if (sValid == 0) { if (tid == 0) nsel[f * g.nlevels + l] = 0; return; }
    if (tid == 0) {
        int pos = 0;
        for (int i = 0; i < nIni; i++) {
            if (cnt[0][i] == 0) { newpos[i] = -1; continue; }
            nd[1][pos] = nd[0][i]; cnt[1][pos] = cnt[0][i]; sq[1][pos] = sq[0][i]; newpos[i] = (short)pos; pos++;
        }
        sL = pos; sFinish = 0; sMode = 0;
    }
    __syncthreads();
    if (in_lds) { for (int i = tid; i < nc; i += OT_T) { const int o = lnof[i]; if (o != DROPPED) lnof[i] = (uint16_t)newpos[o]; } }
    else        { for (int i = tid; i < nc; i += OT_T) { const int o = gnof[i]; if (o != DROPPED) gnof[i] = (uint16_t)newpos[o]; } }
    int cur = 1;      // buffer holding the current list
    __syncthreads();
    OTP(0)
    // ---- split passes
    while (!sFinish) {
        const int Lsz = sL;
        QNode* cn = nd[cur]; uint32_t* ccnt = cnt[cur]; uint32_t* csq = sq[cur];
        QNode* nn = nd[cur ^ 1]; uint32_t* ncnt = cnt[cur ^ 1]; uint32_t* nsq = sq[cur ^ 1];
        // (okey: the second phase's ordering key of every node -- size, then creation sequence, + 1; 0 = not expandable -- in the bytes of childpos, which
        // is dead between the relabel sweep of one pass and the rebuild of the next)
        unsigned long long* okey = reinterpret_cast<unsigned long long*>(&childpos[0][0]);
        for (int i = tid; i < Lsz; i += OT_T) {
            cc[i][0] = cc[i][1] = cc[i][2] = cc[i][3] = 0;
            const uint32_t ci = ccnt[i];
            okey[i] = ci > 1 ? (((unsigned long long)ci << 32) | csq[i]) + 1ull : 0ull;
        }
        __syncthreads();
        // (four keys per trip, each step of the dependent chain key -> node id -> node -> counter issued for all four before the next: a block is a chain
        // of LDS latencies, and one key per trip paid every one of them in full)
        auto count_trip = [&](auto lds_tag, int i0, const uint32_t (&kx)[OT_UN]) {
            constexpr bool in_lds = decltype(lds_tag)::value;
            int ni[OT_UN]; uint32_t cn_[OT_UN]; QNode q[OT_UN];
#pragma unroll
            for (int u = 0; u < OT_UN; u++) { const int i = i0 + u * OT_T; ni[u] = i < nc ? (int)NOF(i) : DROPPED; }
#pragma unroll
            for (int u = 0; u < OT_UN; u++) cn_[u] = ni[u] != DROPPED ? ccnt[ni[u]] : 0u;
#pragma unroll
            for (int u = 0; u < OT_UN; u++) q[u] = cn[ni[u] != DROPPED ? ni[u] : 0];
#pragma unroll
            for (int u = 0; u < OT_UN; u++) {
                if (cn_[u] > 1) {
                    const int x = kx[u] & 4095, y = (kx[u] >> 12) & 4095;
                    const int mx = q[u].x0 + ((q[u].x1 - q[u].x0 + 1) >> 1), my = q[u].y0 + ((q[u].y1 - q[u].y0 + 1) >> 1);
                    const int qd = (x < mx) ? (y < my ? 0 : 2) : (y < my ? 1 : 3);
                    atomicAdd(&cc[ni[u]][qd], 1u);
                }
            }
        };
        if (in_lds) {
#pragma unroll
            for (int t = 0; t < KPT / OT_UN; t++) {
                const int i0 = tid + t * OT_UN * OT_T;
                if (i0 < nc) { uint32_t kx[OT_UN]; for (int u = 0; u < OT_UN; u++) kx[u] = kxr[t * OT_UN + u]; count_trip(std::true_type{}, i0, kx); }
            }
        } else {
            for (int i0 = tid; i0 < nc; i0 += OT_UN * OT_T) {
                uint32_t kx[OT_UN];
#pragma unroll
                for (int u = 0; u < OT_UN; u++) { const int i = i0 + u * OT_T; kx[u] = i < nc ? KEYX(i) : 0u; }
                count_trip(std::false_type{}, i0, kx);
            }
        }
        OTP(1)
        // processing order for the second phase only: rank among expandable nodes by (size desc, creation seq desc)
        if (sMode == 1) for (int i = tid; i < Lsz; i += OT_T) {
            const unsigned long long ki = okey[i];
            if (ki) {
                int r = 0;
#pragma unroll 8
                for (int j = 0; j < Lsz; j++) r += okey[j] > ki ? 1 : 0;      // (one broadcast LDS read per node, no branch: the loop was a chain of LDS latencies)
                order[r] = (short)i;
            }
        }
        __syncthreads();
        OTP(2)
        // ---- rebuild of the node list, all threads (thread t owns nodes / ranks t*IPT .. t*IPT+IPT-1, so block scans
        // run in list order).  The list semantics are ORB-SLAM2's std::list with push_front: the children of the split nodes
        // come first -- parents in reverse processing order, each as n4, n3, n2, n1 -- then the nodes that were not split,
        // in their previous order.  A child's creation sequence number is 4 * (its parent's processing index) + quadrant.
        // emit the children of parent i (processing index e) at positions cb, cb+1, ...
        #define EMIT_CHILDREN(i, e, cb)                                                                     \
            {   const QNode q = cn[i];                                                                      \
                const int mx = q.x0 + ((q.x1 - q.x0 + 1) >> 1), my = q.y0 + ((q.y1 - q.y0 + 1) >> 1);       \
                int p_ = (cb);                                                                              \
                for (int qd = 3; qd >= 0; qd--) {                                                           \
                    const uint32_t c = cc[i][qd];                                                           \
                    if (c == 0) { childpos[i][qd] = -1; continue; }                                         \
                    if (p_ < NODES) {                                                                       \
                        QNode ch;                                                                           \
                        ch.x0 = (qd & 1) ? (short)mx : q.x0; ch.x1 = (qd & 1) ? q.x1 : (short)mx;           \
                        ch.y0 = (qd & 2) ? (short)my : q.y0; ch.y1 = (qd & 2) ? q.y1 : (short)my;           \
                        nn[p_] = ch; ncnt[p_] = c; nsq[p_] = 4u * (uint32_t)(e) + (uint32_t)qd;             \
                        childpos[i][qd] = (short)p_;                                                        \
                    } else childpos[i][qd] = 0;                                                             \
                    p_++;                                                                                   \
                }                                                                                           \
            }
        constexpr int IPT = NODES > OT_T ? NODES / OT_T : 1;
        const int mode = sMode;
        // scan A over nodes: expandable count (low half) | kept-as-is count (high half); scan B: children | children with > 1 key
        uint32_t aloc[IPT], bloc[IPT], asum = 0, bsum = 0;
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const int i = tid * IPT + k;
            aloc[k] = 0; bloc[k] = 0;
            if (i < Lsz) {
                if (ccnt[i] > 1) {
                    aloc[k] = 1u;
                    for (int qd = 0; qd < 4; qd++) { const uint32_t c = cc[i][qd]; bloc[k] += (c > 0 ? 1u : 0u) + (c > 1 ? 0x10000u : 0u); }
                } else aloc[k] = 0x10000u;
            }
            asum += aloc[k]; bsum += bloc[k];
        }
        uint32_t aT, bT;
        uint32_t aP = block_excl_scan_u32(asum, sscan, aT);
        uint32_t bP = block_excl_scan_u32(bsum, sscan, bT);
        const int E = (int)(aT & 0xFFFFu);
        int pos_total;
        if (mode == 0) {
            const int totalCh = (int)(bT & 0xFFFFu);
            pos_total = totalCh + (int)(aT >> 16);
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                const int i = tid * IPT + k;
                if (i < Lsz) {
                    if (aloc[k] & 1u) {                 // split: children go in front, in reverse parent order
                        const int e = (int)(aP & 0xFFFFu), cb = totalCh - (int)((bP & 0xFFFFu) + (bloc[k] & 0xFFFFu));
                        EMIT_CHILDREN(i, e, cb); newpos[i] = -2;
                    } else {
                        const int p_ = totalCh + (int)(aP >> 16);
                        if (p_ < NODES) { nn[p_] = cn[i]; ncnt[p_] = ccnt[i]; nsq[p_] = csq[i]; newpos[i] = (short)p_; } else newpos[i] = 0;
                    }
                }
                aP += aloc[k]; bP += bloc[k];
            }
            if (tid == 0) {
                if (pos_total >= N || pos_total == Lsz) sFinish = 1;
                else if (pos_total + 3 * (int)(bT >> 16) > N) sMode = 1;
            }
        } else {
            // second phase: nodes are taken in `order` (size desc, creation desc) until the list would reach N
            if (tid == 0) sFirst = E - 1;
            uint32_t nloc[IPT], nsum = 0;
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                const int r = tid * IPT + k;
                nloc[k] = 0;
                if (r < E) { const int i = order[r]; for (int qd = 0; qd < 4; qd++) nloc[k] += cc[i][qd] > 0 ? 1u : 0u; }
                nsum += nloc[k];
            }
            uint32_t nT;
            uint32_t nP = block_excl_scan_u32(nsum, sscan, nT);           // (the barriers inside also publish sFirst)
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                const int r = tid * IPT + k;
                nP += nloc[k];                                             // inclusive: children of ranks 0..r
                if (r < E) {
                    cumn[r] = (uint16_t)nP;
                    if (Lsz + (int)nP - (r + 1) >= N) atomicMin(&sFirst, r);
                }
            }
            for (int i = tid; i < Lsz; i += OT_T) newpos[i] = 0;
            __syncthreads();
            const int nsplit = E > 0 ? sFirst + 1 : 0, totalCh = nsplit > 0 ? (int)cumn[nsplit - 1] : 0;
            pos_total = totalCh + Lsz - nsplit;
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                const int r = tid * IPT + k;
                if (r < nsplit) { const int i = order[r]; EMIT_CHILDREN(i, r, totalCh - (int)cumn[r]); newpos[i] = -2; }
            }
            __syncthreads();
            uint32_t kloc[IPT], ksum = 0;
#pragma unroll
            for (int k = 0; k < IPT; k++) { const int i = tid * IPT + k; kloc[k] = (i < Lsz && newpos[i] != -2) ? 1u : 0u; ksum += kloc[k]; }
            uint32_t kT;
            uint32_t kP = block_excl_scan_u32(ksum, sscan, kT);
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                const int i = tid * IPT + k;
                if (kloc[k]) {
                    const int p_ = totalCh + (int)kP;
                    if (p_ < NODES) { nn[p_] = cn[i]; ncnt[p_] = ccnt[i]; nsq[p_] = csq[i]; newpos[i] = (short)p_; } else newpos[i] = 0;
                }
                kP += kloc[k];
            }
            if (tid == 0 && (pos_total >= N || pos_total == Lsz)) sFinish = 1;
        }
        #undef EMIT_CHILDREN
        if (tid == 0) {
            if (pos_total > NODES) { sErr = 1; sFinish = 1; pos_total = NODES; }
            sL = pos_total;
        }
        __syncthreads();
        OTP(3)
        auto relabel_trip = [&](auto lds_tag, int i0, const uint32_t (&kx)[OT_UN]) {
            constexpr bool in_lds = decltype(lds_tag)::value;
            int ni[OT_UN]; int np_[OT_UN]; QNode q[OT_UN];
#pragma unroll
            for (int u = 0; u < OT_UN; u++) { const int i = i0 + u * OT_T; ni[u] = i < nc ? (int)NOF(i) : DROPPED; }
#pragma unroll
            for (int u = 0; u < OT_UN; u++) { const int n_ = ni[u] != DROPPED ? ni[u] : 0; np_[u] = newpos[n_]; q[u] = cn[n_]; }
#pragma unroll
            for (int u = 0; u < OT_UN; u++) {
                if (ni[u] == DROPPED) continue;
                const int i = i0 + u * OT_T;
                if (np_[u] == -2) {
                    const int x = kx[u] & 4095, y = (kx[u] >> 12) & 4095;
                    const int mx = q[u].x0 + ((q[u].x1 - q[u].x0 + 1) >> 1), my = q[u].y0 + ((q[u].y1 - q[u].y0 + 1) >> 1);
                    const int qd = (x < mx) ? (y < my ? 0 : 2) : (y < my ? 1 : 3);
                    SETNOF(i, childpos[ni[u]][qd]);
                } else SETNOF(i, np_[u]);
            }
        };
        if (in_lds) {
#pragma unroll
            for (int t = 0; t < KPT / OT_UN; t++) {
                const int i0 = tid + t * OT_UN * OT_T;
                if (i0 < nc) { uint32_t kx[OT_UN]; for (int u = 0; u < OT_UN; u++) kx[u] = kxr[t * OT_UN + u]; relabel_trip(std::true_type{}, i0, kx); }
            }
        } else {
            for (int i0 = tid; i0 < nc; i0 += OT_UN * OT_T) {
                uint32_t kx[OT_UN];
#pragma unroll
                for (int u = 0; u < OT_UN; u++) { const int i = i0 + u * OT_T; kx[u] = i < nc ? KEYX(i) : 0u; }
                relabel_trip(std::false_type{}, i0, kx);
            }
        }
        cur ^= 1;
        __syncthreads();
        OTP(4) OTP_CNT(7)
    }
    // ---- best response per node (ties: first in detection order == lowest rank)
    const int Lf = sL;
    if (sErr && tid == 0) atomicOr(status, 2);
    for (int i = tid; i < Lf; i += OT_T) best[i] = 0ull;
    __syncthreads();
    for (int i0 = tid; i0 < nc; i0 += OT_UN * OT_T) {
        int ni[OT_UN]; cand_t k[OT_UN];
#pragma unroll
        for (int u = 0; u < OT_UN; u++) { const int i = i0 + u * OT_T; ni[u] = i < nc ? (in_lds ? (int)lnof[i] : (int)gnof[i]) : DROPPED; k[u] = i < nc ? KEY(i) : make_uint2(0u, 0u); }
#pragma unroll
        for (int u = 0; u < OT_UN; u++) {
            if (ni[u] == DROPPED) continue;
            const unsigned long long v = ((unsigned long long)(k[u].x >> 24) << 56) | ((unsigned long long)(0xFFFFFFFFu - k[u].y) << 24) | (k[u].x & 0xFFFFFFu);
            atomicMax(&best[ni[u]], v);
        }
    }
    __syncthreads();
    for (int i = tid; i < Lf && i < L.sel_cap; i += OT_T) {
        const unsigned long long v = best[i];
        const uint32_t x = (uint32_t)(v & 4095) + L.minBX, y = (uint32_t)((v >> 12) & 4095) + L.minBY, s = (uint32_t)(v >> 56);
        out[i] = x | (y << 12) | (s << 24);
    }
    if (tid == 0) { nsel[f * g.nlevels + l] = min(Lf, L.sel_cap); if (Lf > L.sel_cap) atomicOr(status, 4); }
    OTP(5)
}
#ifdef SSM_OT_PROF
extern "C" void ssm_debug_octree_prof(void)
{
    unsigned long long h[8][8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(ot_prof), sizeof(h)) != hipSuccess) return;
    for (int l = 0; l < 8; l++) fprintf(stderr, "octree level %d: init %llu count-sweep %llu order %llu rebuild %llu relabel %llu best %llu | passes %llu\n", l, h[l][0], h[l][1], h[l][2], h[l][3], h[l][4], h[l][5], h[l][7]);
    for (auto& r_ : h) for (auto& v_ : r_) v_ = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ot_prof), h, sizeof(h));
}
#endif
hipError_t k_octree(int n, const OrbGeom& g, const cand_t* cand, const int32_t* ncand, const int32_t* cellmax, uint16_t* node_of,
                    uint32_t* sel, int32_t* nsel, int32_t* status, hipStream_t s)
{
    int need = 0;
    for (int l = 0; l < g.nlevels; l++) need = need > g.L[l].nfeat + 3 ? need : g.L[l].nfeat + 3;
    for (int l = 0; l < g.nlevels; l++) need = need > 4 * g.L[l].nIni + 8 ? need : 4 * g.L[l].nIni + 8;
    if (need <= 256 - 8) octree_kernel<256><<<dim3(n, g.nlevels), OT_T, 0, s>>>(g, cand, ncand, cellmax, node_of, sel, nsel, status);
    else if (need <= 512 - 8) octree_kernel<512><<<dim3(n, g.nlevels), OT_T, 0, s>>>(g, cand, ncand, cellmax, node_of, sel, nsel, status);
    else                 octree_kernel<1024><<<dim3(n, g.nlevels), OT_T, 0, s>>>(g, cand, ncand, cellmax, node_of, sel, nsel, status);
    return hipGetLastError();
}

// ------------------------------------------------------------------ K5b: IC_Angle + steered BRIEF + 3-D position
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)2.2204460492503131e-16); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else          { c = ax / (ay + (float)2.2204460492503131e-16); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}
__device__ __forceinline__ void contract_sincos(float angle_rad, float* s, float* c)
{
    const double PIO2_HI = 1.57079632673412561417e+00, PIO2_LO = 6.07710050650619224932e-11;
    const double x = (double)angle_rad;
    const double kd = rint(x * 0.63661977236758134308);
    const int k = (int)kd;
    const double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    const double r2 = r * r;
    double ps = -1.0 / 1307674368000.0;
    ps = ps * r2 + 1.0 / 6227020800.0;
    ps = ps * r2 - 1.0 / 39916800.0;
    ps = ps * r2 + 1.0 / 362880.0;
    ps = ps * r2 - 1.0 / 5040.0;
    ps = ps * r2 + 1.0 / 120.0;
    ps = ps * r2 - 1.0 / 6.0;
    const double sn = r + r * (r2 * ps);
    double pc = 1.0 / 20922789888000.0;
    pc = pc * r2 - 1.0 / 87178291200.0;
    pc = pc * r2 + 1.0 / 479001600.0;
    pc = pc * r2 - 1.0 / 3628800.0;
    pc = pc * r2 + 1.0 / 40320.0;
    pc = pc * r2 - 1.0 / 720.0;
    pc = pc * r2 + 1.0 / 24.0;
    pc = pc * r2 - 0.5;
    const double cs = 1.0 + r2 * pc;
    double S, C;
    switch (k & 3) { case 0: S = sn; C = cs; break; case 1: S = cs; C = -sn; break; case 2: S = -sn; C = -cs; break; default: S = -cs; C = sn; break; }
    *s = (float)S; *c = (float)C;
}
// inclusive wave sum by DPP (no LDS traffic); the total is in lane 63
__device__ __forceinline__ int wave_total(int v) { return __builtin_amdgcn_readlane((int)wave_incl_scan_u32((uint32_t)v), 63); }
// K5 runs as three launches (it was one kernel with one wave per keypoint doing everything: 773 VALU instructions per wave, of which
// ~110 were the wave-uniform atan2 + double-precision sin / cos and the depth unprojection executed by all 64 lanes for one value):
//   orient_kernel : one WAVE per selected keypoint -- intensity-centroid moments m10, m01 over the radius-15 disc
//   angle_kernel  : one THREAD per selected keypoint -- fastAtan2, the contract sin / cos, the cv::KeyPoint record, project2dTo3d
//   brief_kernel  : one WAVE per selected keypoint -- steered BRIEF on the blurred level
// The per-CU texture-address unit charges >= 16 cycles per vector-memory instruction whatever its width, and byte-granular patch reads
// made this stage TA-bound; so a wave stages its patch in LDS with ALIGNED 16-BYTE loads (rows of the level images are 16-B aligned:
// 31 rows x 48 B for the moments, 37 rows x 64 B for the steered BRIEF reach of +-18) and then works on LDS bytes.
struct KpAux { int32_t m10, m01; float sn, cs; };          // per (frame, slot): moments -> (sin, cos) of the keypoint angle
#define DP_ROWS_O 31
#define DP_QW_O 3           // 16-byte words per staged row of the orientation patch: [(x-15) & ~15, +48) covers x-15 .. x+15
#define DP_ROWS_B 37
#define DP_QW_B 4           // blurred patch: [(x-18) & ~15, +64) covers x-18 .. x+18
// slot -> (level, index in level, output index); false when the slot is empty
__device__ __forceinline__ bool kp_slot(const OrbGeom& g, const int32_t* __restrict__ ns, int slot, int& l, int& oidx)
{
    if (slot >= g.sel_total) return false;
    l = 0;
    while (l + 1 < g.nlevels && slot >= g.L[l+1].sel_off) l++;
    const int i = slot - g.L[l].sel_off;
    if (i >= ns[l]) return false;
    oidx = i;
    for (int k = 0; k < l; k++) oidx += ns[k];
    return true;
}
// kp_prepare_kernel: one THREAD per slot resolves the slot once -- level, output index, byte offset of the keypoint's pixel inside the frame's pyramid -- so
// that the wave-per-keypoint kernels below start from ONE 16-byte record instead of each walking the level table and the per-level counts (that walk
// was a chain of three dependent memory round trips and ~40 % of their instructions)
struct KpRec { uint32_t off; uint32_t stride_level; int32_t oidx; uint32_t pk; };      // off = img_off + y stride + x; stride | level << 16; oidx < 0: empty slot
__global__ void __launch_bounds__(256)
kp_prepare_kernel(OrbGeom g, const uint32_t* __restrict__ sel, const int32_t* __restrict__ nsel, KpRec* __restrict__ recs, int32_t* __restrict__ nkp)
{
    const int f = blockIdx.y, slot = blockIdx.x * 256 + threadIdx.x;
    const int32_t* ns = nsel + f * g.nlevels;
    if (slot == 0) { int t = 0; for (int l = 0; l < g.nlevels; l++) t += ns[l]; nkp[f] = t; }
    if (slot >= g.sel_total) return;
    int l, oidx;
    KpRec r; r.off = 0; r.stride_level = 0; r.oidx = -1; r.pk = 0;
    if (kp_slot(g, ns, slot, l, oidx)) {
        const LevelGeom& L = g.L[l];
        r.pk = sel[(size_t)f * g.sel_total + slot];
        const int x = r.pk & 4095, y = (r.pk >> 12) & 4095;
        r.off = (uint32_t)(L.img_off + y * L.stride + x); r.stride_level = (uint32_t)L.stride | ((uint32_t)l << 16); r.oidx = oidx;
    }
    recs[(size_t)f * g.sel_total + slot] = r;
}
// orient_kernel: lane = (row v = (lane >> 1) - 15, half): the right half-row is the 16 pixels [x, x + 16), the left one [x - 16, x); ONE (unaligned)
// 16-byte load per lane fetches the whole disc of a keypoint in a single vector-memory instruction; the disc's extent in that row (umax) becomes a
// per-lane byte mask, the sums are v_dot4_u32_u8 with constant weights: si = sum p, sk = sum k p (k = byte index 0..15; u = k on the right,
// u = k - 16 on the left).  A wave walks OR_KPW consecutive slots with all their loads in flight first: the stage is latency-bound, not VALU-bound.
#define OR_KPW 4
// (the builtin, not inline asm: hipcc cannot see an opcode inside an asm statement and then omits the wait state a dependent VALU instruction
// needs behind a dot instruction -- scripts/ubench/orient_check.hip shows the wrong sums that gives)
__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_udot4(a, b, acc, false); }
__global__ void __launch_bounds__(256)
orient_kernel(OrbGeom g, unsigned long long umax_pack, const uint8_t* __restrict__ pyr, const KpRec* __restrict__ recs, KpAux* __restrict__ aux, int nframes)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, f = blockIdx.x;       // frame-fastest launch order: see blur_kernel
    const int slot0 = (blockIdx.y * 4 + wv) * OR_KPW;
    if (f >= nframes || slot0 >= g.sel_total) return;
    // per-lane constants
    const int r = lane >> 1, left = !(lane & 1);
    const int v = r <= 2 * SSM_HALF_PATCH ? r - SSM_HALF_PATCH : 0;
    const int d = r <= 2 * SSM_HALF_PATCH ? (int)((umax_pack >> (4 * (v < 0 ? -v : v))) & 15ull) : -1;      // -1: lanes 62, 63 hold no row
    uint32_t M[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int kk = 4 * j + k;
            const bool keep = left ? (kk >= 16 - d && d > 0) : (kk <= d);       // left: u = kk - 16 in [-d, -1]; right: u = kk in [0, d]
            m |= keep ? (0xFFu << (8 * k)) : 0u;
        }
        M[j] = m;
    }
    const int koff = left ? -16 : 0;
    const uint8_t* fr = pyr + (size_t)f * g.pyr_bytes;
    // The OR_KPW records first, then the OR_KPW pixel loads, each group UNCONDITIONAL (an empty or missing slot reads a harmless address): with the loads
    // inside `if (rec.oidx >= 0)` the compiler issued record, wait, pixels, next record, vmcnt(0) (which waits for the pixels too) ... -- eight memory round trips
    // one after the other per wave instead of two (round 6, found in the kernel's load / wait listing).
    uint4 px[OR_KPW]; bool ok[OR_KPW]; KpRec rec[OR_KPW];
#pragma unroll
    for (int q = 0; q < OR_KPW; q++) rec[q] = recs[(size_t)f * g.sel_total + min(slot0 + q, g.sel_total - 1)];
#pragma unroll
    for (int q = 0; q < OR_KPW; q++) {
        ok[q] = slot0 + q < g.sel_total && rec[q].oidx >= 0;
        // keypoints sit >= 19 px from every border: [x - 16, x + 16) x [y - 15, y + 15] lies inside the level
        const uint8_t* p = ok[q] ? fr + rec[q].off + v * (int)(rec[q].stride_level & 0xFFFFu) + koff : fr;
        uint4 t; __builtin_memcpy(&t, p, 16);                            // unaligned 16-byte load
        px[q] = t;
    }
#pragma unroll
    for (int q = 0; q < OR_KPW; q++) {
        if (!ok[q]) continue;                                           // wave-uniform
        const uint32_t a0 = px[q].x & M[0], a1 = px[q].y & M[1], a2 = px[q].z & M[2], a3 = px[q].w & M[3];
        uint32_t si = udot4(a0, 0x01010101u, 0u); si = udot4(a1, 0x01010101u, si); si = udot4(a2, 0x01010101u, si); si = udot4(a3, 0x01010101u, si);
        uint32_t sk = udot4(a0, 0x03020100u, 0u); sk = udot4(a1, 0x07060504u, sk); sk = udot4(a2, 0x0B0A0908u, sk); sk = udot4(a3, 0x0F0E0D0Cu, sk);
        const int sui = (int)sk - (left ? 16 * (int)si : 0);
        const int m10 = wave_total(sui), m01 = wave_total(v * (int)si);
        if (lane == 0) { KpAux a; a.m10 = m10; a.m01 = m01; a.sn = 0.f; a.cs = 0.f; aux[(size_t)f * g.sel_total + slot0 + q] = a; }
    }
}
__global__ void __launch_bounds__(256)
angle_kernel(OrbGeom g, const KpRec* __restrict__ recs, const uint16_t* __restrict__ depth, ssm_camera cam,
             KpAux* __restrict__ aux, ssm_keypoint* __restrict__ kps, float* __restrict__ pos3d)
{
    const int f = blockIdx.y, slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= g.sel_total) return;
    const KpRec rec = recs[(size_t)f * g.sel_total + slot];
    if (rec.oidx < 0) return;
    const int l = (int)(rec.stride_level >> 16), oidx = rec.oidx;
    const LevelGeom& L = g.L[l];
    const uint32_t pk = rec.pk;
    const int x = pk & 4095, y = (pk >> 12) & 4095, score = pk >> 24;
    KpAux a = aux[(size_t)f * g.sel_total + slot];
    const float angle = fast_atan2_deg((float)a.m01, (float)a.m10);
    contract_sincos(angle * (float)(3.14159265358979323846 / 180.f), &a.sn, &a.cs);
    aux[(size_t)f * g.sel_total + slot] = a;
    ssm_keypoint kp;
    kp.x = (float)x; kp.y = (float)y;
    if (l != 0) { kp.x *= L.sf; kp.y *= L.sf; }
    kp.size = (float)SSM_PATCH * L.sf; kp.angle = angle; kp.response = (float)score; kp.octave = l; kp.class_id = -1;
    kps[(size_t)f * g.cap + oidx] = kp;
    if (pos3d) {
        float px = 0.f, py = 0.f, pz = 0.f;
        if (depth) {
            const int u = (int)kp.x, v = (int)kp.y;            // include/orb.h:50 float -> int truncation
            const uint16_t d = depth[(size_t)f * g.W * g.H + (size_t)v * g.W + u];
            if (d != 0) {
                pz = (float)((double)d / cam.scale);
                px = (float)(((double)u - cam.cx) * (double)pz / cam.fx);
                py = (float)(((double)v - cam.cy) * (double)pz / cam.fy);
            }
        }
        float* o = pos3d + ((size_t)f * g.cap + oidx) * 3;
        o[0] = px; o[1] = py; o[2] = pz;
    }
}
// value of lane + n inside the 16-lane row (0 beyond the row): the 16 nibbles of one 64-bit descriptor word sit in one row
#define ROW_SHL(v, n) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x100 + (n), 0xF, 0xF, true))
#define BR_KPW 2               // keypoints per step: the lane's 64 bytes of pattern serve both, and both patches are in flight together (4 measured slower: LDS halves the occupancy)
#define BR_NIT 4               // steps per wave: the patches of step k + 1 are loaded (into registers) while step k computes
#define BR_PW ((DP_ROWS_B * DP_QW_B + 63) / 64)      // 16-byte patch words per lane per keypoint (3)
__global__ void __launch_bounds__(256)
brief_kernel(OrbGeom g, const uint8_t* __restrict__ blur, const KpRec* __restrict__ recs,
             const float* __restrict__ pattern_f, const KpAux* __restrict__ aux, uint8_t* __restrict__ desc, int nframes)
{
    __shared__ uint4 pb[4][BR_KPW][DP_ROWS_B * DP_QW_B];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), f = blockIdx.x;       // frame-fastest launch order: see blur_kernel
    const int slot0 = (blockIdx.y * 4 + wv) * (BR_KPW * BR_NIT);
    if (f >= nframes || slot0 >= g.sel_total) return;
    const KpRec* rp = recs + (size_t)f * g.sel_total;
    const KpAux* ap = aux + (size_t)f * g.sel_total;
    // steered BRIEF: lane -> 4 of the 256 comparisons.  The pattern arrives as floats (converted once on the host); the rotation runs on packed
    // pairs -- (x0, x1) and (y0, y1) of a comparison through v_pk_mul_f32 / v_pk_add_f32: the same IEEE multiplies and adds in the same order as the
    // scalar form (no contraction) -- and cvRound is the 1.5 * 2^23 trick: adding 12582912.0f rounds to nearest-even in the adder and leaves the
    // integer in the low mantissa bits (|value| <= 18.4)
    typedef float f2 __attribute__((ext_vector_type(2)));
    const float4* pf = reinterpret_cast<const float4*>(pattern_f) + lane * 4;
    const float4 pq[4] = {pf[0], pf[1], pf[2], pf[3]};
    // a step's patch words, global -> registers (a wave walks BR_NIT steps: the next step's words are in flight while this one computes, so a wave
    // pays the record -> patch latency chain once, not per keypoint)
    uint4 pw[BR_KPW][BR_PW];
    KpRec rec[BR_KPW], nrec[BR_KPW];
    auto load_recs = [&](int it, KpRec (&r)[BR_KPW]) {
#pragma unroll
        for (int j = 0; j < BR_KPW; j++) {
            const int slot = slot0 + it * BR_KPW + j;
            r[j].oidx = -1;
            if (it < BR_NIT && slot < g.sel_total) r[j] = rp[slot];
        }
    };
    auto fetch = [&](const KpRec (&r)[BR_KPW]) {
#pragma unroll
        for (int j = 0; j < BR_KPW; j++) {
            if (r[j].oidx < 0) continue;
            const int stride = (int)(r[j].stride_level & 0xFFFFu), x = r[j].pk & 4095, y = (r[j].pk >> 12) & 4095;
            const uint8_t* lv = blur + (size_t)f * g.blur_bytes;                 // the blurred pyramid is tiled (blur_off): a 37-row patch covers ~25 lines, not ~47
            const int boff = g.L[r[j].stride_level >> 16].boff;
            const int xb0 = (x - 18) & ~15;
#pragma unroll
            for (int q = 0; q < BR_PW; q++) {
                const int e = lane + 64 * q, rr = e >> 2, c = e & 3, gx = xb0 + 16 * c;
                pw[j][q] = (e < DP_ROWS_B * DP_QW_B && gx < stride) ? *reinterpret_cast<const uint4*>(lv + blur_off(boff, stride, gx, y + rr - 18)) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    load_recs(0, rec);
    fetch(rec);
    for (int it = 0; it < BR_NIT; it++) {
        load_recs(it + 1, nrec);
        KpAux a[BR_KPW];
#pragma unroll
        for (int j = 0; j < BR_KPW; j++) if (rec[j].oidx >= 0) a[j] = ap[slot0 + it * BR_KPW + j];
        // this step's words into the wave's LDS patches (every lane is past the previous step's reads: a wave runs in lock step between the barriers)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < BR_KPW; j++) {
            if (rec[j].oidx < 0) continue;
#pragma unroll
            for (int q = 0; q < BR_PW; q++) { const int e = lane + 64 * q; if (e < DP_ROWS_B * DP_QW_B) pb[wv][j][e] = pw[j][q]; }
        }
        fetch(nrec);                                                    // the next step's patches are in flight during this step's comparisons
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < BR_KPW; j++) {
            if (rec[j].oidx < 0) continue;
            const int x = rec[j].pk & 4095, xb0 = (x - 18) & ~15;
            const uint8_t* bb = reinterpret_cast<const uint8_t*>(pb[wv][j]) + 18 * (DP_QW_B * 16) + (x - xb0);      // centre pixel of the blurred patch
            const float sb = a[j].sn, ca = a[j].cs;
            const f2 s2 = {sb, sb}, c2 = {ca, ca}, magic = {12582912.0f, 12582912.0f};
            uint32_t nib = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const f2 X = {pq[k].x, pq[k].z}, Y = {pq[k].y, pq[k].w};
                const f2 xs = X * s2, yc = Y * c2, xc = X * c2, ys = Y * s2;
                const f2 yy = (xs + yc) + magic, xx = (xc - ys) + magic;
                // index = yy * 64 + xx with both integers still biased by 0x4B400000: one shift-add, the bias leaves as a constant (mod 2^32)
                const uint32_t i0 = (__float_as_uint(yy.x) << 6) + __float_as_uint(xx.x) - 0x4B400000u * 65u;
                const uint32_t i1 = (__float_as_uint(yy.y) << 6) + __float_as_uint(xx.y) - 0x4B400000u * 65u;
                const int t0 = bb[(int)i0], t1 = bb[(int)i1];
                nib |= (uint32_t)(t0 < t1) << k;
            }
            // 16 nibbles (lanes 16j..16j+15) -> one 64-bit word, by DPP inside the row
            uint32_t b = nib | (ROW_SHL(nib, 1) << 4);                   // even lanes: one byte
            b |= ROW_SHL(b, 2) << 8;                                     // lanes %4==0: 2 bytes
            b |= ROW_SHL(b, 4) << 16;                                    // lanes %8==0: 4 bytes
            const uint32_t hi = ROW_SHL(b, 8);
            if ((lane & 15) == 0)
                reinterpret_cast<uint2*>(desc + ((size_t)f * g.cap + rec[j].oidx) * 32)[lane >> 4] = make_uint2(b, hi);
        }
#pragma unroll
        for (int j = 0; j < BR_KPW; j++) rec[j] = nrec[j];
    }
}
hipError_t k_describe(int n, const OrbGeom& g, const uint8_t* pyr, const uint8_t* blur, const uint32_t* sel,
                      const int32_t* nsel, const float* pattern_f, const uint16_t* depth, ssm_camera cam, void* kpaux,
                      ssm_keypoint* kps, uint8_t* desc, float* pos3d, int32_t* nkp, hipStream_t s)
{
    unsigned long long um = 0;
    for (int v = 0; v <= SSM_HALF_PATCH; v++) um |= (unsigned long long)(g.umax[v] & 15) << (4 * v);
    KpAux* aux = reinterpret_cast<KpAux*>(kpaux);
    KpRec* recs = reinterpret_cast<KpRec*>(aux + (size_t)n * g.sel_total);       // second half of the buffer
    kp_prepare_kernel<<<dim3((g.sel_total + 255) / 256, n), 256, 0, s>>>(g, sel, nsel, recs, nkp);
    orient_kernel<<<dim3((n + 7) & ~7, (g.sel_total + 4 * OR_KPW - 1) / (4 * OR_KPW)), 256, 0, s>>>(g, um, pyr, recs, aux, n);
    angle_kernel<<<dim3((g.sel_total + 255) / 256, n), 256, 0, s>>>(g, recs, depth, cam, aux, kps, pos3d);
    brief_kernel<<<dim3((n + 7) & ~7, (g.sel_total + 4 * BR_KPW * BR_NIT - 1) / (4 * BR_KPW * BR_NIT)), 256, 0, s>>>(g, blur, recs, pattern_f, aux, desc, n);
    return hipGetLastError();
}
