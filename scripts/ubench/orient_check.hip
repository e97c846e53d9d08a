// checks the lane scheme of orient_kernel (kernels_orb.hip) against a scalar loop: unaligned 16-byte loads, per-lane byte masks, v_dot4_u32_u8, DPP wave totals
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ uint32_t scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}
__device__ __forceinline__ int wave_total(int v) { return __builtin_amdgcn_readlane((int)scan((uint32_t)v), 63); }
template <bool ASM> __device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t acc)
{
    if (ASM) { uint32_t r; asm("v_dot4_u32_u8 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc)); return r; }
    return __builtin_amdgcn_udot4(a, b, acc, false);
}
template <bool ASM>
__global__ void k(const unsigned char* img, int stride, const int* xy, int n, unsigned long long umax_pack, int* out, int* dbg)
{
    const int lane = threadIdx.x & 63, kp = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (kp >= n) return;
    const int r = lane >> 1, left = !(lane & 1);
    const int v = r <= 30 ? r - 15 : 0;
    const int d = r <= 30 ? (int)((umax_pack >> (4 * (v < 0 ? -v : v))) & 15ull) : -1;
    uint32_t M[4];
    for (int j = 0; j < 4; j++) { uint32_t m = 0; for (int kq = 0; kq < 4; kq++) { const int kk = 4 * j + kq; const bool keep = left ? (kk >= 16 - d && d > 0) : (kk <= d); m |= keep ? (0xFFu << (8 * kq)) : 0u; } M[j] = m; }
    const int x = xy[2 * kp], y = xy[2 * kp + 1];
    const unsigned char* p = img + (size_t)(y + v) * stride + (x + (left ? -16 : 0));
    uint4 t; __builtin_memcpy(&t, p, 16);
    const uint32_t a0 = t.x & M[0], a1 = t.y & M[1], a2 = t.z & M[2], a3 = t.w & M[3];
    uint32_t si = udot4<ASM>(a0, 0x01010101u, 0u); si = udot4<ASM>(a1, 0x01010101u, si); si = udot4<ASM>(a2, 0x01010101u, si); si = udot4<ASM>(a3, 0x01010101u, si);
    uint32_t sk = udot4<ASM>(a0, 0x03020100u, 0u); sk = udot4<ASM>(a1, 0x07060504u, sk); sk = udot4<ASM>(a2, 0x0B0A0908u, sk); sk = udot4<ASM>(a3, 0x0F0E0D0Cu, sk);
    const int sui = (int)sk - (left ? 16 * (int)si : 0);
    if (kp == 0) { dbg[lane * 4] = (int)si; dbg[lane * 4 + 1] = (int)sk; dbg[lane * 4 + 2] = sui; dbg[lane * 4 + 3] = (int)M[0]; }
    const int m10 = wave_total(sui), m01 = wave_total(v * (int)si);
    if (lane == 0) { out[2 * kp] = m10; out[2 * kp + 1] = m01; }
}
int main()
{
    const int W = 256, H = 128, n = 512, umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    std::vector<unsigned char> img((size_t)W * H); srand(1);
    for (auto& b : img) b = (unsigned char)(rand() >> 7);
    std::vector<int> xy(2 * n);
    for (int i = 0; i < n; i++) { xy[2 * i] = 19 + rand() % (W - 38); xy[2 * i + 1] = 19 + rand() % (H - 38); }
    unsigned long long um = 0; for (int v = 0; v < 16; v++) um |= (unsigned long long)umax[v] << (4 * v);
    unsigned char* dimg; int *dxy, *dout, *ddbg;
    hipMalloc(&dimg, img.size()); hipMalloc(&dxy, xy.size() * 4); hipMalloc(&dout, n * 8); hipMalloc(&ddbg, 64 * 16);
    hipMemcpy(dimg, img.data(), img.size(), hipMemcpyHostToDevice); hipMemcpy(dxy, xy.data(), xy.size() * 4, hipMemcpyHostToDevice);
    int rc = 0;
    for (int variant = 0; variant < 2; variant++) {
        if (variant) k<true><<<n / 4, 256>>>(dimg, W, dxy, n, um, dout, ddbg); else k<false><<<n / 4, 256>>>(dimg, W, dxy, n, um, dout, ddbg);
        std::vector<int> out(2 * n), dbg(256);
        hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost); hipMemcpy(dbg.data(), ddbg, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; i++) {
            int m10 = 0, m01 = 0; const int x = xy[2 * i], y = xy[2 * i + 1];
            for (int v = -15; v <= 15; v++) { const int d = umax[v < 0 ? -v : v]; for (int u = -d; u <= d; u++) { const int p = img[(size_t)(y + v) * W + x + u]; m10 += u * p; m01 += v * p; } }
            if (m10 != out[2 * i] || m01 != out[2 * i + 1]) { if (bad < 3) printf("  kp %d: want (%d, %d) got (%d, %d)\n", i, m10, m01, out[2 * i], out[2 * i + 1]); bad++; }
        }
        printf("%s: %d of %d keypoints wrong\n", variant ? "inline-asm v_dot4_u32_u8" : "__builtin_amdgcn_udot4", bad, n);
        if (bad) { rc = 1; for (int l = 28; l < 34; l++) printf("    lane %d: si %d sk %d sui %d M0 %08x\n", l, dbg[4 * l], dbg[4 * l + 1], dbg[4 * l + 2], (unsigned)dbg[4 * l + 3]); }
    }
    return rc;
}
