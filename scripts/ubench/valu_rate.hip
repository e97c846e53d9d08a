// micro-benchmark: issue rate of v_bcnt_u32_b32 vs v_xor_b32 vs v_add3 on gfx950 (decides the matcher's VALU roofline)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(unsigned* out, int iters)
{
    unsigned a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] = __builtin_popcount(a[i] ^ 0x5bd1e995u) + a[(i + 1) & 7];     // xor + bcnt(acc)
                if (OP == 1) a[i] = (a[i] ^ a[(i + 1) & 7]) ^ 0x5bd1e995u;                         // 2 xor (or one xor3?)
                if (OP == 2) a[i] = __builtin_popcount(a[i]) + a[(i + 3) & 7];                      // bcnt(acc) only
                if (OP == 3) a[i] = max(a[i] + 7u, a[(i + 1) & 7]) ;                                // add + max
            }
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int vops_per_inner)
{
    unsigned* d; hipMalloc(&d, 256 * 8 * 4 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, blocks = 256 * 8;
    k<OP><<<blocks, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a); k<OP><<<blocks, 256>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double laneops = (double)blocks * 256 * iters * 64.0 * vops_per_inner;
    printf("%-28s %8.3f ms  %7.2f T lane-op/s (counting %d VALU per element)\n", name, ms, laneops / ms / 1e9, vops_per_inner);
    hipFree(d);
}
int main()
{
    run<0>("xor + bcnt_acc", 2);
    run<1>("xor + xor", 2);
    run<2>("bcnt_acc", 1);
    run<3>("add + max", 2);
    return 0;
}
