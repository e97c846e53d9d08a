// kernels_segnet.hip -- K9: the SegNet forward the reference runs through Caffe (Classifier::Predict,
// /root/reference/src/segnet.cpp:87-108: net_->ForwardPrefilled()) for the driving_webdemo network: VGG-16 encoder
// (13 conv3x3 pad 1 + BN + ReLU, 5 max-pool 2x2 s2 CEIL with arg-max mask), mirrored decoder (5 mask-driven Upsample,
// 13 conv), last conv -> 12 classes, ArgMax.  This is the ONLY place on the path where MFMA is used:
//   conv = implicit GEMM  M = pixels, N = Cout, K = 9 * Cin  on v_mfma_f32_32x32x16_f16 (fp16 storage, fp32 accumulate),
//   activations NHWC fp16 so that a lane's 8-element K fragment is one 16-byte load,
//   BN (folded scale/shift) + ReLU fused into the epilogue.
// Also here: Classifier::Preprocess (segnet.cpp:130-167; cv::resize to 480x360, planar float, mean 0) and the label
// colouring of experiment/segnet.cpp:80-83,131-146 (Pavement->Road remap, cv::resize back to the frame size, cv::LUT).
#include "ssm_internal.h"

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
    half8 lo, hi;
#pragma unroll
    for (int k = 0; k < 8; k++) { lo[k] = (_Float16)0.f; hi[k] = (_Float16)0.f; }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int h0 = src[((size_t)sy0 * sw + sx0) * 3 + c] * a0 + src[((size_t)sy0 * sw + sx1) * 3 + c] * a1;
        const int h1 = src[((size_t)sy1 * sw + sx0) * 3 + c] * a0 + src[((size_t)sy1 * sw + sx1) * 3 + c] * a1;
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        lo[c] = (_Float16)(float)(v & 255);
    }
    half8* o = reinterpret_cast<half8*>(out + ((size_t)blockIdx.y * dw * dh + p) * 16);
    o[0] = lo; o[1] = hi;
}

// ------------------------------------------------------------------ conv3x3 pad 1 (+ scale/shift + ReLU), implicit GEMM on MFMA
// block = 4 waves along M; a wave owns 64 consecutive (flattened) output pixels x 64 output channels = 2x2 tiles of
// 32x32, 64 fp32 accumulators.  Per K step of 16 (one tap, 16 input channels): two 16-byte A loads (pixel rows) and two
// 16-byte B loads (weight rows) feed 4 MFMAs.  A/B operand maps (cdna guide s.3): lane l holds A[row l&31][k 8(l>>5)..+7]
// and B[k 8(l>>5)..+7][col l&31]; C/D: col = l&31, row = (reg&3) + 8(reg>>2) + 4(l>>5).
// weights: [CoutPad][9][Cin] fp16 (CoutPad multiple of 64, zero rows beyond Cout).  Cin multiple of 16.
template <bool RELU>
__global__ void __launch_bounds__(256)
conv3x3_mfma_kernel(const _Float16* __restrict__ in, const _Float16* __restrict__ wt, const float* __restrict__ scale, const float* __restrict__ shift,
                    _Float16* __restrict__ out, int H, int W, int Cin, int Cout, int CoutStore)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int npix = H * W;
    const int p0 = (blockIdx.x * 4 + wv) * 64;
    const int n0 = blockIdx.y * 64;
    const _Float16* inf = in + (size_t)blockIdx.z * npix * Cin;
    // the two pixel rows this lane feeds (tile 0: p0 + r, tile 1: p0 + 32 + r)
    int py[2], pxx[2]; bool pv[2];
#pragma unroll
    for (int t = 0; t < 2; t++) { const int p = p0 + 32 * t + r; pv[t] = p < npix; py[t] = pv[t] ? p / W : 0; pxx[t] = pv[t] ? p - py[t] * W : 0; }
    const _Float16* w0 = wt + ((size_t)(n0 + r) * 9) * Cin + 8 * hh;
    const _Float16* w1 = wt + ((size_t)(n0 + 32 + r) * 9) * Cin + 8 * hh;
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
    for (int tap = 0; tap < 9; tap++) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const _Float16* a_ptr[2]; bool a_ok[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int yy = py[t] + dy, xx = pxx[t] + dx;
            a_ok[t] = pv[t] && yy >= 0 && yy < H && xx >= 0 && xx < W;
            a_ptr[t] = inf + ((size_t)(a_ok[t] ? yy * W + xx : 0)) * Cin + 8 * hh;
        }
        const _Float16* b0p = w0 + (size_t)tap * Cin;
        const _Float16* b1p = w1 + (size_t)tap * Cin;
#pragma unroll 2
        for (int c0 = 0; c0 < Cin; c0 += 16) {
            const half8 A0 = a_ok[0] ? *reinterpret_cast<const half8*>(a_ptr[0] + c0) : zero;
            const half8 A1 = a_ok[1] ? *reinterpret_cast<const half8*>(a_ptr[1] + c0) : zero;
            const half8 B0 = *reinterpret_cast<const half8*>(b0p + c0);
            const half8 B1 = *reinterpret_cast<const half8*>(b1p + c0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, B0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, B1, acc[1][1], 0, 0, 0);
        }
    }
    // epilogue: y = acc * scale[ch] + shift[ch] (conv bias and BatchNorm folded), ReLU, fp16 NHWC store
    _Float16* of = out + (size_t)blockIdx.z * npix * CoutStore;
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int ch = n0 + 32 * tn + r;
        if (ch >= Cout) continue;
        const float sc = scale[ch], sh = shift[ch];
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int p = p0 + 32 * tm + (k & 3) + 8 * (k >> 2) + 4 * hh;
                if (p < npix) {
                    float v = acc[tm][tn][k] * sc + sh;
                    if (RELU) v = fmaxf(v, 0.f);
                    of[(size_t)p * CoutStore + ch] = (_Float16)v;
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
    const _Float16* src = in + (size_t)blockIdx.y * H * W * C;
    half8 best; uint8_t bc[8];
    bool first = true;
#pragma unroll
    for (int dy = 0; dy < 2; dy++)
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
            const int y = ph * 2 + dy, x = pw * 2 + dx;
            if (y >= H || x >= W) continue;
            const half8 v = *reinterpret_cast<const half8*>(src + ((size_t)y * W + x) * C + cg * 8);
            if (first) { best = v; for (int k = 0; k < 8; k++) bc[k] = (uint8_t)(dy * 2 + dx); first = false; }
            else {
#pragma unroll
                for (int k = 0; k < 8; k++) if (v[k] > best[k]) { best[k] = v[k]; bc[k] = (uint8_t)(dy * 2 + dx); }
            }
        }
    const size_t o = ((size_t)blockIdx.y * PH * PW + pp) * C + cg * 8;
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
    const size_t s = ((size_t)blockIdx.y * PH * PW + (size_t)ph * PW + pw) * C + cg * 8;
    const half8 v = *reinterpret_cast<const half8*>(in + s);
    const uint2 pk = *reinterpret_cast<const uint2*>(code + s);
    half8 o;
#pragma unroll
    for (int k = 0; k < 8; k++) { const int cd = ((k < 4 ? pk.x : pk.y) >> (8 * (k & 3))) & 255; o[k] = cd == my ? v[k] : (_Float16)0.f; }
    *reinterpret_cast<half8*>(out + ((size_t)blockIdx.y * H * W + p) * C + cg * 8) = o;
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
                         int Cin, int Cout, int CoutPad, int CoutStore, int relu, hipStream_t s)
{
    dim3 grid((H * W + 255) / 256, CoutPad / 64, n);
    if (relu) conv3x3_mfma_kernel<true><<<grid, 256, 0, s>>>((const _Float16*)in, (const _Float16*)wt, scale, shift, (_Float16*)out, H, W, Cin, Cout, CoutStore);
    else      conv3x3_mfma_kernel<false><<<grid, 256, 0, s>>>((const _Float16*)in, (const _Float16*)wt, scale, shift, (_Float16*)out, H, W, Cin, Cout, CoutStore);
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
