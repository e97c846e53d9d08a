// kernels_synth.hip -- device generator of the synthetic 640x480 RGB-D + 12-class stream (BASELINE.json configs[1],
// SURVEY.md s.8d C2).  Bench / test input, not part of the mapping path.  Integer-only; the definition is
// oracle/synth.c and tests/test_synth.py checks the two bit for bit.
#include "ssm_internal.h"

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t hash3(uint64_t seed, uint64_t tag, int64_t a, int64_t b)
{
    return mix64(seed ^ mix64(tag ^ mix64((uint64_t)a * 0x9E3779B1ULL ^ mix64((uint64_t)b))));
}
__device__ __forceinline__ int32_t isin_q15(uint32_t p)
{
    const int32_t x = (int32_t)(p & 0x7FFF);
    const int32_t y = (x * (32768 - x)) >> 13;
    return (p & 0x8000) ? -y : y;
}
__constant__ uint8_t c_palette[12][3] = {
    {128,128,128}, {0,0,128}, {128,192,192}, {0,69,255}, {128,64,128}, {222,40,60},
    {0,128,128}, {128,128,192}, {128,64,64}, {128,0,64}, {0,64,64}, {192,128,0}
};
__global__ void __launch_bounds__(256)
synth_kernel(uint64_t seed, int first, int w, int h, uint8_t* __restrict__ bgr, uint16_t* __restrict__ depth, uint8_t* __restrict__ sem,
             uint8_t* __restrict__ lab, double* __restrict__ pose)
{
    const int frame_id = first + blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix == 0 && pose) {
        double* T = pose + (size_t)blockIdx.y * 16;
        for (int i = 0; i < 16; i++) T[i] = 0.0;
        T[0] = T[5] = T[10] = T[15] = 1.0; T[12] = 0.01 * (double)frame_id;
    }
    if (pix >= w * h) return;
    const int v = pix / w, u = pix - v * w;
    const int64_t wx = (int64_t)u + 2 * (int64_t)frame_id, wy = v;
    const int64_t lx = wx >> 4, ly = wy >> 4; const int fx = (int)(wx & 15), fy = (int)(wy & 15);
    const int a = (int)(hash3(seed, 1, lx, ly) & 255), b = (int)(hash3(seed, 1, lx + 1, ly) & 255);
    const int c = (int)(hash3(seed, 1, lx, ly + 1) & 255), d = (int)(hash3(seed, 1, lx + 1, ly + 1) & 255);
    const int vn = ((a * (16 - fx) + b * fx) * (16 - fy) + (c * (16 - fx) + d * fx) * fy + 128) >> 8;
    int B = 64 + (vn >> 1), G = B, R = B;
    const int64_t bx = wx >> 5, by = wy >> 5;
    for (int dy = -1; dy <= 0; dy++)
        for (int dx = -1; dx <= 0; dx++)
            for (int k = 0; k < 4; k++) {
                const uint64_t H = hash3(seed, 2 + (uint64_t)k, bx + dx, by + dy);
                const int64_t x0 = ((bx + dx) << 5) + (int64_t)(H & 31), y0 = ((by + dy) << 5) + (int64_t)((H >> 5) & 31);
                const int rw = 4 + (int)((H >> 10) % 21), rh = 4 + (int)((H >> 20) % 21);
                if (wx >= x0 && wx < x0 + rw && wy >= y0 && wy < y0 + rh) { B = (int)((H >> 32) & 255); G = (int)((H >> 40) & 255); R = (int)((H >> 48) & 255); }
            }
    const uint64_t N = hash3(seed, 7, (int64_t)frame_id, (int64_t)pix);
    const int nb = (int)(N & 3), ng = (int)((N >> 2) & 3), nr = (int)((N >> 4) & 3);
    B += (nb == 0) ? -1 : (nb == 1 ? 1 : 0); G += (ng == 0) ? -1 : (ng == 1 ? 1 : 0); R += (nr == 0) ? -1 : (nr == 1 ? 1 : 0);
    B = min(max(B, 0), 255); G = min(max(G, 0), 255); R = min(max(R, 0), 255);
    const size_t o = (size_t)blockIdx.y * w * h + pix;
    bgr[3*o] = (uint8_t)B; bgr[3*o+1] = (uint8_t)G; bgr[3*o+2] = (uint8_t)R;
    const int32_t s = isin_q15(((uint32_t)u * 197u) & 0xFFFF), cc = isin_q15(((uint32_t)v * 254u + 16384u) & 0xFFFF);
    const int64_t prod = (int64_t)600 * s * cc;
    int dd = 1000 + (int)((prod + ((int64_t)1 << 29)) >> 30) + (200 * (frame_id & 15)) / 16;
    if (((N >> 8) & 1023) < 51) dd = 0;
    depth[o] = (uint16_t)dd;
    const int cl = (int)(hash3(seed, 9, bx, by) % 12);
    if (lab) lab[o] = (uint8_t)cl;
    sem[3*o] = c_palette[cl][0]; sem[3*o+1] = c_palette[cl][1]; sem[3*o+2] = c_palette[cl][2];
}
hipError_t k_synth(uint64_t seed, int first, int n, int w, int h, uint8_t* bgr, uint16_t* depth, uint8_t* sem,
                   uint8_t* lab, double* pose, hipStream_t s)
{
    synth_kernel<<<dim3((w * h + 255) / 256, n), 256, 0, s>>>(seed, first, w, h, bgr, depth, sem, lab, pose);
    return hipGetLastError();
}
