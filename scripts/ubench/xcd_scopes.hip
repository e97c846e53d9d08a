// Round trip of a flag between two workgroups, by placement (same XCD / different XCDs) and by the scope of the loads and stores (agent: sc1, workgroup: sc0).
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_pingpong scripts/ubench/xcd_scopes.hip && /tmp/xcd_pingpong
// Background for kernels_pnp.hip (pc_lane_finish): the pose chain's blocks trade group sums through such flags ~55 times per frame.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(1))) unsigned long long gu64;
// SCOPE: a __HIP_MEMORY_SCOPE_*, or 100 = workgroup-scope store + (buffer_inv sc0, then a workgroup-scope load: the L1 is emptied, the load is served by the XCD's L2),
// 101 = workgroup-scope store + a returning workgroup-scope atomic add of 0 (performed in the XCD's L2)
template <int SCOPE>
__device__ __forceinline__ unsigned long long ld(gu64* p)
{
    if constexpr (SCOPE == 100) { asm volatile("buffer_inv sc0" ::: "memory"); return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    else if constexpr (SCOPE == 101) return __hip_atomic_fetch_add(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE);
}
template <int SCOPE>
__device__ __forceinline__ void st(gu64* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE >= 100 ? __HIP_MEMORY_SCOPE_WORKGROUP : SCOPE); }
template <int SCOPE>
__global__ void __launch_bounds__(64) pingpong(unsigned long long* flags, unsigned* xcc, long long* out, int ida, int idb, int rounds)
{
    unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) xcc[blockIdx.x] = x & 15u;
    if ((int)blockIdx.x != ida && (int)blockIdx.x != idb) return;
    if (threadIdx.x != 0) return;
    gu64* f = (gu64*)flags;
    const bool isa = (int)blockIdx.x == ida;
    // meet first (agent scope: always valid)
    st<__HIP_MEMORY_SCOPE_AGENT>(f + (isa ? 16 : 24), 1ull);
    for (unsigned s = 0; s < (1u << 24) && ld<__HIP_MEMORY_SCOPE_AGENT>(f + (isa ? 24 : 16)) == 0ull; s++) {}
    const long long t0 = clock64();
    int fail = 0;
    for (int i = 1; i <= rounds && !fail; i++) {
        if (isa) {
            st<SCOPE>(f, (unsigned long long)i);
            unsigned s = 0; while (ld<SCOPE>(f + 8) != (unsigned long long)i) if (++s > (1u << 18)) { fail = 1; break; }
        } else {
            unsigned s = 0; while (ld<SCOPE>(f) != (unsigned long long)i) if (++s > (1u << 18)) { fail = 1; break; }
            st<SCOPE>(f + 8, (unsigned long long)i);
        }
    }
    const long long t1 = clock64();
    if (isa) { out[0] = t1 - t0; out[1] = fail; }
    if (fail) { st<__HIP_MEMORY_SCOPE_AGENT>(f, ~0ull); st<__HIP_MEMORY_SCOPE_AGENT>(f + 8, ~0ull); }
}
int main()
{
    unsigned long long* flags; unsigned* xcc; long long* out;
    hipMalloc((void**)&flags, 4096); hipMalloc((void**)&xcc, 64 * 4); hipMalloc((void**)&out, 64);
    const int rounds = 2000;
    unsigned hx[64]; long long ho[2];
    const int pairs[3][2] = {{0, 8}, {0, 1}, {3, 59}};     // same XCD if workgroups go round-robin by id; different; same
    const char* names[4] = {"agent (sc1)", "workgroup (sc0)", "workgroup store, buffer_inv sc0 + load", "workgroup store, returning atomic add 0"};
    for (int sc = 0; sc < 4; sc++)
        for (int p = 0; p < 3; p++) {
            hipMemset(flags, 0, 4096); hipMemset(out, 0, 64);
            if (sc == 0) hipLaunchKernelGGL(pingpong<__HIP_MEMORY_SCOPE_AGENT>, dim3(64), dim3(64), 0, 0, flags, xcc, out, pairs[p][0], pairs[p][1], rounds);
            else if (sc == 1) hipLaunchKernelGGL(pingpong<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(64), dim3(64), 0, 0, flags, xcc, out, pairs[p][0], pairs[p][1], rounds);
            else if (sc == 2) hipLaunchKernelGGL(pingpong<100>, dim3(64), dim3(64), 0, 0, flags, xcc, out, pairs[p][0], pairs[p][1], rounds);
            else              hipLaunchKernelGGL(pingpong<101>, dim3(64), dim3(64), 0, 0, flags, xcc, out, pairs[p][0], pairs[p][1], rounds);
            hipDeviceSynchronize();
            hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost); hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
            printf("%s scope, workgroups %2d (XCC %u) and %2d (XCC %u): %s %.0f clocks per round trip\n", names[sc], pairs[p][0], hx[pairs[p][0]], pairs[p][1], hx[pairs[p][1]],
                   ho[1] ? "TIMED OUT," : "", (double)ho[0] / rounds);
        }
    printf("XCC_ID of workgroups 0..15:"); for (int i = 0; i < 16; i++) printf(" %u", hx[i]); printf("\n");
    return 0;
}
