// match_bench -- the MFMA matcher (kernels_match.hip) alone on random descriptors: time per 250-frame launch, with the ablation switches of
// MM_ABLATE compiled in (-DMM_ABLATE=n: 1 no key tracking, 2 no LDS staging / barrier, 4 no train prefetch, 8 no MFMA) to see where the time goes.
// Build (repo root): hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-honor-nans -Iinclude -Isemantic_slam_mapping_amd/csrc [-DMM_ABLATE=n] scripts/ubench/match_bench.hip -o scripts/ubench/bin/match_bench_n
#include "kernels_match.hip"
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv)
{
    const int n = 250, R = 5, cap = 1000, capT = 1024, rows = n + R;
    std::vector<uint8_t> h((size_t)rows * cap * 32);
    std::mt19937 rng(1); for (auto& b : h) b = (uint8_t)rng();
    std::vector<int32_t> nk(rows, cap);
    uint8_t *d, *eq, *et; int32_t *dn, *nout; uint2* knn; ssm_dmatch* out;
    CK(hipMalloc(&d, h.size())); CK(hipMalloc(&eq, (size_t)rows * capT * MM_DB)); CK(hipMalloc(&et, (size_t)rows * capT * MM_DB));
    CK(hipMalloc(&dn, rows * 4)); CK(hipMalloc(&knn, (size_t)n * R * capT * 8)); CK(hipMalloc(&out, (size_t)n * R * cap * sizeof(ssm_dmatch))); CK(hipMalloc(&nout, n * R * 4));
    CK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dn, nk.data(), rows * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    CK(k_match_expand(d, dn, 0, rows, cap, capT, eq, et, 0));
    const int qblocks = (capT + MM_QB - 1) / MM_QB, fpx = (n + 7) >> 3;
    float best = 1e9f, sum = 0;
    for (int it = 0; it < 12; it++) {
        hipEventRecord(e0, 0);
        match_mfma_kernel<<<8 * fpx * R * qblocks, 256, 0, 0>>>(eq, et, dn, 0, n, R, R, capT, qblocks, fpx, knn);
        hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double mf = (double)n * R * 4 * 4 * 32 * 2 * MM_KS;      // MFMAs per launch
    printf("MM_ABLATE=%d  best %.1f us  mean %.1f us  (%.3f us/frame)   MFMA issue at 32 cycles, 2.4 GHz: %.1f us\n", MM_ABLATE, best * 1e3, sum / 10 * 1e3, best * 1e3 / n, mf * 32 / 1024 / 2.4e9 * 1e6);
    return 0;
}
