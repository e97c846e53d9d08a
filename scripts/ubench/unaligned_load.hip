// does gfx950 (as configured on this pool) serve UNALIGNED 16-byte global loads correctly?  (orient_kernel relies on it)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const unsigned char* src, uint4* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 t; __builtin_memcpy(&t, src + i, 16);      // byte offset i: every alignment
    out[i] = t;
}
int main()
{
    const int n = 4096;
    std::vector<unsigned char> h(n + 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)(i * 37 + (i >> 8) * 11 + 5);
    unsigned char* d; uint4* o;
    hipMalloc(&d, h.size()); hipMalloc(&o, n * sizeof(uint4));
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(d, o, n);
    std::vector<uint4> r(n);
    hipMemcpy(r.data(), o, n * sizeof(uint4), hipMemcpyDeviceToHost);
    int bad = 0, bad_by_align[16] = {0};
    for (int i = 0; i < n; i++) if (memcmp(&r[i], h.data() + i, 16)) { bad++; bad_by_align[i & 15]++; }
    printf("unaligned 16-byte global loads: %d of %d wrong\n", bad, n);
    for (int a = 0; a < 16; a++) printf("  offset %% 16 == %2d: %d wrong\n", a, bad_by_align[a]);
    return bad != 0;
}
