// mfma_valu -- can VALU instructions issue under v_mfma_i32_32x32x32_i8 on gfx950?  A loop of [MFMA acc0; N VALU; MFMA acc1; N VALU] from inline asm.
// mode 0: the VALU ops are v_min_u32 (4-cycle class) on registers no MFMA touches; mode 2: v_and_b32 (2-cycle class).  Prints ns per MFMA at 1 and 2 waves per SIMD.  If the time stays at the N = 0 value
// the VALU work is hidden; if it grows by 4 N cycles it is serialised.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_valu.hip -o scripts/ubench/bin/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define V0 "v_min_u32 %2, %2, %6\n v_min_u32 %3, %3, %6\n"
#define V2 "v_and_b32 %2, %2, %6\n v_and_b32 %3, %3, %6\n"
template <int N2, int MODE>     // N2 = VALU pairs per MFMA
__global__ void k(int iters, int* out)
{
    v16i acc0 = {}, acc1 = {};
    v4i a = {1, 2, 3, 4}, b = {(int)threadIdx.x, 1, 2, 3};
    unsigned t0 = threadIdx.x, t1 = threadIdx.x * 3, c = 12345;
    for (int i = 0; i < iters; i++) {
#define BODY(VA, VB) \
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %4, %5, %0\n" VA "v_mfma_i32_32x32x32_i8 %1, %4, %5, %1\n" VB \
                     : "+v"(acc0), "+v"(acc1), "+v"(t0), "+v"(t1) : "v"(a), "v"(b), "v"(c));
        if (MODE == 0) { if (N2 == 0) BODY("", "") else if (N2 == 1) BODY(V0, V0) else if (N2 == 2) BODY(REP2(V0), REP2(V0)) else if (N2 == 3) BODY(REP2(V0) V0, REP2(V0) V0) else BODY(REP4(V0), REP4(V0)) }
        if (MODE == 2) { if (N2 == 0) BODY("", "") else if (N2 == 1) BODY(V2, V2) else if (N2 == 2) BODY(REP2(V2), REP2(V2)) else if (N2 == 3) BODY(REP2(V2) V2, REP2(V2) V2) else BODY(REP4(V2), REP4(V2)) }
    }
    int s = t0 + t1; for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i];
    if (s == 0x7fffffff) out[0] = s;
}
template <int N2, int MODE> void run(int threads, int* d)
{
    const int iters = 4000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<N2, MODE><<<256, threads>>>(100, d); hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) { hipEventRecord(e0); k<N2, MODE><<<256, threads>>>(iters, d); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    const int wps = threads / 256;
    printf("mode %d  VALU per MFMA %d  waves/SIMD %d : %.2f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", MODE, 2 * N2, wps, best * 1e6 / (iters * 2.0 * wps), best * 1e6 / (iters * 2.0 * wps) * 2.4);
}
int main()
{
    int* d; hipMalloc(&d, 4);
    for (int th = 256; th <= 512; th += 256) {
        run<0, 0>(th, d); run<1, 0>(th, d); run<2, 0>(th, d); run<3, 0>(th, d); run<4, 0>(th, d);
        run<2, 2>(th, d); run<4, 2>(th, d);
    }
    return 0;
}
