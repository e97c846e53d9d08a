// xcd_pingpong.hip -- round-trip time of a tagged 8-byte granule (agent-scope relaxed store / load, cdna_hip_programming.md G16 R2) between two workgroups, as a function of
// WHERE the two run: the same XCD (shared L2) or different XCDs (through the fabric).  Decides how the blocks of the pose chain's cluster form (kernels_pnp.hip) are placed:
// the hand-off is correct wherever the blocks land (agent scope), only its latency depends on placement.
// Grid: 64 one-wave blocks; block 0 plays ping with block `peer`, everybody else exits.  Every block records its XCC_ID so the table shows the real placement.
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_pingpong xcd_pingpong.hip ; run: ./xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void pingpong(unsigned long long* box, int peer, int rounds, long long* clocks, int* xcc, int* fail)
{
    const int b = blockIdx.x;
    if (threadIdx.x == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[b] = (int)(id & 0xF); }
    if (b != 0 && b != peer) return;
    if (threadIdx.x != 0) return;
    unsigned long long* mine = box + (b == 0 ? 0 : 16), *theirs = box + (b == 0 ? 16 : 0);      // 128 bytes apart
    const long long t0 = clock64();
    for (int r = 1; r <= rounds; r++) {
        if (b == 0) __hip_atomic_store(mine, ((unsigned long long)r << 32) | 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long v = 0; unsigned spins = 0;
        for (; spins < (1u << 22); ++spins) {
            v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 32) == (unsigned long long)r) break;
        }
        if (spins == (1u << 22)) { *fail = 1; return; }
        if (b != 0) __hip_atomic_store(mine, ((unsigned long long)r << 32) | 9u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (b == 0) *clocks = clock64() - t0;
}

int main()
{
    unsigned long long* box; long long* clocks; int *xcc, *fail;
    CK(hipMalloc((void**)&box, 4096)); CK(hipMalloc((void**)&clocks, 8)); CK(hipMalloc((void**)&xcc, 64 * 4)); CK(hipMalloc((void**)&fail, 4));
    const int rounds = 2000;
    int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    printf("| peer block | XCC of block 0 | XCC of the peer | clocks per round trip | ns per round trip (s_memtime at %d kHz) |\n|---:|---:|---:|---:|---:|\n", clk_khz);
    for (int peer : {1, 2, 4, 7, 8, 16, 24, 9, 32, 40}) {
        CK(hipMemset(box, 0, 4096)); CK(hipMemset(fail, 0, 4)); CK(hipMemset(clocks, 0, 8));
        for (int rep = 0; rep < 2; rep++) {         // (the second launch is the warm one)
            CK(hipMemset(box, 0, 4096));
            hipLaunchKernelGGL(pingpong, dim3(64), dim3(64), 0, 0, box, peer, rounds, clocks, xcc, fail);
            CK(hipDeviceSynchronize());
        }
        long long c = 0; int x[64], f = 0;
        CK(hipMemcpy(&c, clocks, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(x, xcc, sizeof(x), hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        if (f) { printf("| %d | timed out |\n", peer); continue; }
        printf("| %d | %d | %d | %.0f | %.0f |\n", peer, x[0], x[peer], (double)c / rounds, (double)c / rounds * 1e6 / clk_khz);
    }
    return 0;
}
