// conv_bench.hip -- micro-benchmark of the SegNet conv3x3 kernel on the five layer shapes of the network.
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Isemantic_slam_mapping_amd/csrc -Iinclude \
//         [-DCT_ABL_...] scripts/ubench/conv_bench.hip -o gpurun_out/conv_bench
// The kernel source is #included so that experiments can be switched with -D macros; random operands (zeros run at a higher clock).
#include "../../semantic_slam_mapping_amd/csrc/kernels_segnet.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 32, reps = argc > 2 ? atoi(argv[2]) : 10;
    struct Shape { int H, W, Cin, Cout; } shapes[] = { {360, 480, 64, 64}, {180, 240, 128, 128}, {90, 120, 256, 256}, {45, 60, 512, 512}, {23, 30, 512, 512} };
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> ud(-1.f, 1.f);
    double tot_us = 0, tot_fl = 0;
    for (const Shape& sh : shapes) {
        const size_t na = (size_t)n * sh.H * sh.W * sh.Cin, nw = (size_t)sh.Cout * sh.Cin * 9, no = (size_t)n * sh.H * sh.W * sh.Cout;
        std::vector<_Float16> ha(na), hw(nw);
        for (auto& v : ha) v = (_Float16)ud(rng);
        for (auto& v : hw) v = (_Float16)(ud(rng) * 0.05f);
        std::vector<float> sc(sh.Cout, 1.f), sf(sh.Cout, 0.f);
        _Float16 *da, *dw, *dout; float *dsc, *dsf;
        CK(hipMalloc(&da, na * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dout, no * 2)); CK(hipMalloc(&dsc, sh.Cout * 4)); CK(hipMalloc(&dsf, sh.Cout * 4));
        CK(hipMemcpy(da, ha.data(), na * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsc, sc.data(), sh.Cout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsf, sf.data(), sh.Cout * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipMemset(dout, 0xFF, no * 2));
        for (int i = 0; i < 2; i++) CK(k_segnet_conv(da, dw, dsc, dsf, dout, n, sh.H, sh.W, sh.Cin, sh.Cout, 1, 0));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; i++) CK(k_segnet_conv(da, dw, dsc, dsf, dout, n, sh.H, sh.W, sh.Cin, sh.Cout, 1, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, fl = 2.0 * n * sh.H * sh.W * (double)sh.Cin * sh.Cout * 9;
        double util = 0, ghz = 0;
#ifdef SSM_CONV_ABLATE
        {   // one more launch with the block-lifetime counter: MFMA utilisation in shader clocks, independent of the clock the chip holds
            unsigned long long zero = 0, cyc = 0;
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_cycles), &zero, 8));
            CK(k_segnet_conv(da, dw, dsc, dsf, dout, n, sh.H, sh.W, sh.Cin, sh.Cout, 1, 0));
            CK(hipDeviceSynchronize());
            CK(hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_conv_cycles), 8));
            if (cyc) { util = fl / (double)cyc / 1048576.0; ghz = (double)cyc / us * 1e-3; }
            unsigned long long ph[8][4];
            CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_conv_phase), sizeof(ph)));
            if (getenv("SSM_CONV_PHASES")) for (int b = 0; b < 8; b++) if (ph[b][2])
                printf("    probe %d: tiles %llu  mfma-phase %.0f clk/tile  other %.0f clk/tile  lifetime %llu\n", b, ph[b][2], (double)ph[b][0] / ph[b][2], (double)ph[b][1] / ph[b][2], ph[b][3]);
        }
#endif
        printf("%3dx%3d %3d->%3d n=%d : %8.1f us  %7.1f TFLOP/s  mfma-util %.3f  clock %.2f GHz\n", sh.H, sh.W, sh.Cin, sh.Cout, n, us, fl / us * 1e-6, util, ghz);
        // weights: the five shapes appear in the network with these multiplicities (Cin==Cout layers only; an approximation)
        tot_us += us; tot_fl += fl;
        hipFree(da); hipFree(dw); hipFree(dout); hipFree(dsc); hipFree(dsf);
    }
    printf("sum: %.1f us  %.1f TFLOP/s\n", tot_us, tot_fl / tot_us * 1e-6);
    return 0;
}
