// kernels_segnet.hip -- K9: the SegNet forward the reference runs through Caffe (Classifier::Predict,
// /root/reference/src/segnet.cpp:87-108: net_->ForwardPrefilled()) for the driving_webdemo network: VGG-16 encoder
// (13 conv3x3 pad 1 + BN + ReLU, 5 max-pool 2x2 s2 CEIL with arg-max mask), mirrored decoder (5 mask-driven Upsample,
// 13 conv), last conv -> 12 classes, ArgMax.  This is the ONLY place on the path where MFMA is used:
//   conv = implicit GEMM  M = pixels, N = Cout, K = 9 * Cin  on v_mfma_f32_32x32x16_f16 (fp16 storage, fp32 accumulate),
//   activations fp16 in channel-chunked layout [C/32][H][W][32] and weights pre-packed per (Cout tile, Cin chunk) so that
//   every staging load of the LDS-tiled kernel is a contiguous 16-byte read; a lane's 8-element K fragment is one
//   ds_read_b128; BN (folded scale/shift) + ReLU fused into the epilogue.
// Also here: Classifier::Preprocess (segnet.cpp:130-167; cv::resize to 480x360, planar float, mean 0) and the label
// colouring of experiment/segnet.cpp:80-83,131-146 (Pavement->Road remap, cv::resize back to the frame size, cv::LUT).
#include "ssm_internal.h"
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------ pre-processing: BGR u8 frame -> 480x360 NHWC fp16 (C padded to 16)
// cv::resize INTER_LINEAR 8u per channel (same fixed-point contract as the ORB pyramid), then float (exact in fp16), mean 0
__global__ void __launch_bounds__(256)
segnet_prep_kernel(const uint8_t* __restrict__ bgr, int sw, int sh, int dw, int dh,
                   const int32_t* __restrict__ xofs, const int16_t* __restrict__ xa, const int32_t* __restrict__ yofs, const int16_t* __restrict__ ya,
                   _Float16* __restrict__ out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= dw * dh) return;
    const int y = p / dw, x = p - y * dw;
    const uint8_t* src = bgr + (size_t)blockIdx.y * sw * sh * 3;
    const int sy0 = yofs[y], sy1 = min(sy0 + 1, sh - 1), b0 = ya[2*y], b1 = ya[2*y+1];
    const int sx0 = xofs[x], sx1 = min(sx0 + 1, sw - 1), a0 = xa[2*x], a1 = xa[2*x+1];
    half8 lo, z;
#pragma unroll
    for (int k = 0; k < 8; k++) { lo[k] = (_Float16)0.f; z[k] = (_Float16)0.f; }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int h0 = src[((size_t)sy0 * sw + sx0) * 3 + c] * a0 + src[((size_t)sy0 * sw + sx1) * 3 + c] * a1;
        const int h1 = src[((size_t)sy1 * sw + sx0) * 3 + c] * a0 + src[((size_t)sy1 * sw + sx1) * 3 + c] * a1;
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        lo[c] = (_Float16)(float)(v & 255);
    }
    half8* o = reinterpret_cast<half8*>(out + ((size_t)blockIdx.y * dw * dh + p) * 32);     // one 32-channel chunk: B, G, R, 29 zeros
    o[0] = lo; o[1] = z; o[2] = z; o[3] = z;
}

// ------------------------------------------------------------------ conv3x3 pad 1 (+ scale/shift + ReLU): implicit GEMM on MFMA, LDS-staged operands
// A/B operand maps (cdna guide s.3): lane l holds A[row l&31][k 8(l>>5)..+7] and B[k 8(l>>5)..+7][col l&31];
// C/D: col = l&31, row = (reg&3) + 8(reg>>2) + 4(l>>5).
// block = 256 threads = 4 waves; output tile = 32 px wide x 8 rows (256 pixels) x 64 output channels.  Per stage of
// KC = 32 input channels the block stages, with 16-byte loads issued together:
//   * the input halo tile (34 x 10 pixels) as 4 planes of 8 channels: plane[c8][pixel][8 x f16]  (21.8 KB)
//   * the weights [tap][c8][cout 64][8 x f16]                                                     (36.9 KB)
// and then runs 9 taps x 2 K-steps x 4 MFMAs per wave with every operand fragment coming from ONE ds_read_b128:
// a wave owns two output rows (2 M-tiles of 32 contiguous pixels -> conflict-free 512-byte reads) x 2 N-tiles.
// 58.7 KB LDS per block -> two blocks per CU overlap each other's staging and MFMA phases.
#define CT_W 32
#define CT_H 8
#define CT_N 64
#define CT_KC 32
#define CT_PW (CT_W + 2)
#define CT_PH (CT_H + 2)
template <bool RELU>
__global__ void __launch_bounds__(256, 2)
conv3x3_lds_kernel(const _Float16* __restrict__ in, const _Float16* __restrict__ wt, const float* __restrict__ scale, const float* __restrict__ shift,
                   _Float16* __restrict__ out, int H, int W, int Cin, int Cout, int tiles_x)
{
    __shared__ __attribute__((aligned(16))) half8 sa[4 * CT_PH * CT_PW];     // [c8][py][px]
    __shared__ __attribute__((aligned(16))) half8 sb[9 * 4 * CT_N];          // [tap][c8][cout]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int tx0 = (blockIdx.x % tiles_x) * CT_W, ty0 = (blockIdx.x / tiles_x) * CT_H;
    const int n0 = blockIdx.y * CT_N;
    const _Float16* inf = in + (size_t)blockIdx.z * H * W * Cin;          // [Cin/32][H][W][32]
    const int nchunks = Cin / CT_KC;
    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;
    half8 zero;
#pragma unroll
    for (int k = 0; k < 8; k++) zero[k] = (_Float16)0.f;
    // software pipeline: the next stage's global data is fetched into registers (15 x 16 B per thread, all loads issued
    // back to back) while the current stage's MFMAs run; it is written to LDS after the barrier that retires the stage.
    constexpr int NA = (CT_PH * CT_PW * 4 + 255) / 256;             // 6 input chunks per thread (last one partial)
    constexpr int NB = 9 * 4 * CT_N / 256;                          // 9 weight chunks per thread
    half8 ra[NA], rb[NB];
    int a_dst[NA]; const _Float16* a_src[NA];                       // per-thread staging slots are the same for every stage
#pragma unroll
    for (int k = 0; k < NA; k++) {
        const int i = tid + 256 * k;
        a_dst[k] = -1; a_src[k] = nullptr;
        if (i < CT_PH * CT_PW * 4) {
            const int c8 = i & 3, p = i >> 2;
            const int py = p / CT_PW, px = p - py * CT_PW;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1;
            a_dst[k] = c8 * (CT_PH * CT_PW) + p;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) a_src[k] = inf + ((size_t)gy * W + gx) * CT_KC + 8 * c8;
        }
    }
    const half8* wbase = reinterpret_cast<const half8*>(wt) + (size_t)blockIdx.y * nchunks * (9 * 4 * CT_N) + tid;
    const size_t chunk_stride = (size_t)H * W * CT_KC;
#define CT_FETCH(ck)                                                                                   \
    {   _Pragma("unroll") for (int k = 0; k < NA; k++) ra[k] = a_src[k] ? *reinterpret_cast<const half8*>(a_src[k] + (size_t)(ck) * chunk_stride) : zero; \
        _Pragma("unroll") for (int k = 0; k < NB; k++) rb[k] = wbase[(size_t)(ck) * (9 * 4 * CT_N) + 256 * k]; }
    CT_FETCH(0)
    for (int ck = 0; ck < nchunks; ck++) {
        __syncthreads();                                             // previous stage fully consumed
#pragma unroll
        for (int k = 0; k < NA; k++) if (a_dst[k] >= 0) sa[a_dst[k]] = ra[k];
#pragma unroll
        for (int k = 0; k < NB; k++) sb[tid + 256 * k] = rb[k];
        __syncthreads();
        if (ck + 1 < nchunks) CT_FETCH(ck + 1)
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int dy = tap / 3, dx = tap - dy * 3;                 // already offset by the halo (+1)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const int c8 = ks * 2 + hh;
                const half8* pa = sa + c8 * (CT_PH * CT_PW) + (2 * wv + dy) * CT_PW + r + dx;
                const half8 A0 = pa[0], A1 = pa[CT_PW];
                const half8* pbv = sb + (tap * 4 + c8) * CT_N + r;
                const half8 B0 = pbv[0], B1 = pbv[32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, B0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, B1, acc[1][1], 0, 0, 0);
            }
        }
    }
#undef CT_FETCH
    // output in the same chunked layout: chunk = channel / 32; a wave-half writes 64 contiguous bytes per pixel
    const int cout_chunks = (Cout + 31) >> 5;
    _Float16* of = out + (size_t)blockIdx.z * H * W * cout_chunks * 32;
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int ch = n0 + 32 * tn + r, chunk = (n0 >> 5) + tn;
        if (chunk >= cout_chunks) continue;
        const float sc = ch < Cout ? scale[ch] : 0.f, sh = ch < Cout ? shift[ch] : 0.f;      // padded channels store 0
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
            const int gy = ty0 + 2 * wv + tm;
            if (gy >= H) continue;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int gx = tx0 + (k & 3) + 8 * (k >> 2) + 4 * hh;
                if (gx < W) {
                    float v = acc[tm][tn][k] * sc + sh;
                    if (RELU) v = fmaxf(v, 0.f);
                    of[(((size_t)chunk * H + gy) * W + gx) * 32 + r] = (_Float16)v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ max-pool 2x2 stride 2, CEIL mode, with arg-max code
// Caffe (SegNet fork) PoolingLayer MAX with top_mask: window scanned row-major, strict '>' => first maximum wins.
// code = (dy*2 + dx) inside the window.  8 channels (16 B) per thread.
__global__ void __launch_bounds__(256)
pool2x2_kernel(const _Float16* __restrict__ in, int H, int W, int C, _Float16* __restrict__ out, uint8_t* __restrict__ code, int PH, int PW)
{
    const int c8 = C >> 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PH * PW * c8) return;
    const int pp = i / c8, cg = i - pp * c8;
    const int ph = pp / PW, pw = pp - ph * PW;
    const _Float16* src = in + (size_t)blockIdx.y * H * W * C;          // [C/32][H][W][32]
    half8 best; uint8_t bc[8];
    bool first = true;
#pragma unroll
    for (int dy = 0; dy < 2; dy++)
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
            const int y = ph * 2 + dy, x = pw * 2 + dx;
            if (y >= H || x >= W) continue;
            const half8 v = *reinterpret_cast<const half8*>(src + (((size_t)(cg >> 2) * H + y) * W + x) * 32 + (cg & 3) * 8);
            if (first) { best = v; for (int k = 0; k < 8; k++) bc[k] = (uint8_t)(dy * 2 + dx); first = false; }
            else {
#pragma unroll
                for (int k = 0; k < 8; k++) if (v[k] > best[k]) { best[k] = v[k]; bc[k] = (uint8_t)(dy * 2 + dx); }
            }
        }
    const size_t o = (size_t)blockIdx.y * PH * PW * C + ((size_t)(cg >> 2) * PH * PW + pp) * 32 + (cg & 3) * 8;
    *reinterpret_cast<half8*>(out + o) = best;
    uint2 pk; pk.x = bc[0] | (bc[1] << 8) | (bc[2] << 16) | (bc[3] << 24); pk.y = bc[4] | (bc[5] << 8) | (bc[6] << 16) | (bc[7] << 24);
    *reinterpret_cast<uint2*>(code + o) = pk;
}
// SegNet Upsample layer: out[argmax position] = pooled value, everything else 0; explicit output size H x W
__global__ void __launch_bounds__(256)
unpool2x2_kernel(const _Float16* __restrict__ in, const uint8_t* __restrict__ code, int PH, int PW, int C, _Float16* __restrict__ out, int H, int W)
{
    const int c8 = C >> 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W * c8) return;
    const int p = i / c8, cg = i - p * c8;
    const int y = p / W, x = p - y * W;
    const int ph = y >> 1, pw = x >> 1, my = (y & 1) * 2 + (x & 1);
    const size_t s = (size_t)blockIdx.y * PH * PW * C + ((size_t)(cg >> 2) * PH * PW + (size_t)ph * PW + pw) * 32 + (cg & 3) * 8;
    const half8 v = *reinterpret_cast<const half8*>(in + s);
    const uint2 pk = *reinterpret_cast<const uint2*>(code + s);
    half8 o;
#pragma unroll
    for (int k = 0; k < 8; k++) { const int cd = ((k < 4 ? pk.x : pk.y) >> (8 * (k & 3))) & 255; o[k] = cd == my ? v[k] : (_Float16)0.f; }
    *reinterpret_cast<half8*>(out + (size_t)blockIdx.y * H * W * C + ((size_t)(cg >> 2) * H * W + p) * 32 + (cg & 3) * 8) = o;
}
// ArgMax over the class logits (Softmax is monotone): first maximum wins, like caffe ArgMaxLayer's partial_sort on (value, index)
__global__ void __launch_bounds__(256)
argmax_kernel(const _Float16* __restrict__ logits, int npix, int Cstore, int ncls, uint8_t* __restrict__ labels)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const _Float16* l = logits + ((size_t)blockIdx.y * npix + p) * Cstore;
    int best = 0; float bv = (float)l[0];
    for (int c = 1; c < ncls; c++) { const float v = (float)l[c]; if (v > bv) { bv = v; best = c; } }
    labels[(size_t)blockIdx.y * npix + p] = (uint8_t)best;
}
// experiment/segnet.cpp:80-83,131-146: ids -> (Pavement 5 -> Road 4) -> 3-channel id image -> cv::resize to the frame size
// (bilinear ON THE IDS, as the reference does; nearest when `nearest` != 0) -> cv::LUT(color.png) -> BGR class colours
__constant__ uint8_t c_seg_palette[12][3] = {
    {128,128,128}, {0,0,128}, {128,192,192}, {0,69,255}, {128,64,128}, {222,40,60},
    {0,128,128}, {128,128,192}, {128,64,64}, {128,0,64}, {0,64,64}, {192,128,0}
};
__global__ void __launch_bounds__(256)
label_color_kernel(const uint8_t* __restrict__ ids, int sw, int sh, int dw, int dh,
                   const int32_t* __restrict__ xofs, const int16_t* __restrict__ xa, const int32_t* __restrict__ yofs, const int16_t* __restrict__ ya,
                   int pavement_to_road, int nearest, uint8_t* __restrict__ sem_bgr, uint8_t* __restrict__ ids_out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= dw * dh) return;
    const int y = p / dw, x = p - y * dw;
    const uint8_t* src = ids + (size_t)blockIdx.y * sw * sh;
    auto id_at = [&](int yy, int xx) { int v = src[(size_t)yy * sw + xx]; return (pavement_to_road && v == 5) ? 4 : v; };
    int v;
    if (nearest) {
        const int sy = min((int)((y * (long long)sh) / dh), sh - 1), sx = min((int)((x * (long long)sw) / dw), sw - 1);
        v = id_at(sy, sx);
    } else {
        const int sy0 = yofs[y], sy1 = min(sy0 + 1, sh - 1), b0 = ya[2*y], b1 = ya[2*y+1];
        const int sx0 = xofs[x], sx1 = min(sx0 + 1, sw - 1), a0 = xa[2*x], a1 = xa[2*x+1];
        const int h0 = id_at(sy0, sx0) * a0 + id_at(sy0, sx1) * a1;
        const int h1 = id_at(sy1, sx0) * a0 + id_at(sy1, sx1) * a1;
        v = ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 255;
    }
    const size_t o = (size_t)blockIdx.y * dw * dh + p;
    if (ids_out) ids_out[o] = (uint8_t)v;
    if (sem_bgr) {
        uint8_t b = 0, g = 0, r = 0;
        if (v < 12) { b = c_seg_palette[v][0]; g = c_seg_palette[v][1]; r = c_seg_palette[v][2]; }
        sem_bgr[3*o] = b; sem_bgr[3*o+1] = g; sem_bgr[3*o+2] = r;
    }
}

// ------------------------------------------------------------------ launchers
hipError_t k_segnet_prep(const uint8_t* bgr, int n, int sw, int sh, int dw, int dh, const int32_t* xofs, const int16_t* xa,
                         const int32_t* yofs, const int16_t* ya, void* out_f16, hipStream_t s)
{
    segnet_prep_kernel<<<dim3((dw * dh + 255) / 256, n), 256, 0, s>>>(bgr, sw, sh, dw, dh, xofs, xa, yofs, ya, (_Float16*)out_f16);
    return hipGetLastError();
}
hipError_t k_segnet_conv(const void* in, const void* wt, const float* scale, const float* shift, void* out, int n, int H, int W,
                         int CinPad, int Cout, int relu, hipStream_t s)
{
    const int tx = (W + CT_W - 1) / CT_W, ty = (H + CT_H - 1) / CT_H;
    dim3 grid(tx * ty, (Cout + CT_N - 1) / CT_N, n);
    if (relu) conv3x3_lds_kernel<true><<<grid, 256, 0, s>>>((const _Float16*)in, (const _Float16*)wt, scale, shift, (_Float16*)out, H, W, CinPad, Cout, tx);
    else      conv3x3_lds_kernel<false><<<grid, 256, 0, s>>>((const _Float16*)in, (const _Float16*)wt, scale, shift, (_Float16*)out, H, W, CinPad, Cout, tx);
    return hipGetLastError();
}
hipError_t k_segnet_pool(const void* in, int n, int H, int W, int C, void* out, uint8_t* code, hipStream_t s)
{
    const int PH = (H + 1) / 2, PW = (W + 1) / 2;
    pool2x2_kernel<<<dim3((PH * PW * (C / 8) + 255) / 256, n), 256, 0, s>>>((const _Float16*)in, H, W, C, (_Float16*)out, code, PH, PW);
    return hipGetLastError();
}
hipError_t k_segnet_unpool(const void* in, const uint8_t* code, int n, int PH, int PW, int C, void* out, int H, int W, hipStream_t s)
{
    unpool2x2_kernel<<<dim3((H * W * (C / 8) + 255) / 256, n), 256, 0, s>>>((const _Float16*)in, code, PH, PW, C, (_Float16*)out, H, W);
    return hipGetLastError();
}
hipError_t k_segnet_argmax(const void* logits, int n, int npix, int Cstore, int ncls, uint8_t* labels, hipStream_t s)
{
    argmax_kernel<<<dim3((npix + 255) / 256, n), 256, 0, s>>>((const _Float16*)logits, npix, Cstore, ncls, labels);
    return hipGetLastError();
}
hipError_t k_segnet_color(const uint8_t* ids, int n, int sw, int sh, int dw, int dh, const int32_t* xofs, const int16_t* xa,
                          const int32_t* yofs, const int16_t* ya, int pavement_to_road, int nearest, uint8_t* sem_bgr, uint8_t* ids_out, hipStream_t s)
{
    label_color_kernel<<<dim3((dw * dh + 255) / 256, n), 256, 0, s>>>(ids, sw, sh, dw, dh, xofs, xa, yofs, ya, pavement_to_road, nearest, sem_bgr, ids_out);
    return hipGetLastError();
}
