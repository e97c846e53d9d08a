// glds_oob.hip -- what does an out-of-range `buffer_load_dwordx4 ... lds` lane leave in LDS?  (zeros or the old bytes)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, float* out, int nbytes)
{
    __shared__ __attribute__((aligned(16))) float s[256 * 4];
    const int tid = threadIdx.x;
    for (int i = 0; i < 4; i++) s[tid * 4 + i] = -7.f;
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nbytes, 0x00020000);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int voff = tid * 16; if (tid & 1) voff = 0x7FFFFFF0;          // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&s[wv * 256], 16, voff, 0, 0, 0);
    __syncthreads();
    for (int i = 0; i < 4; i++) out[tid * 4 + i] = s[tid * 4 + i];
}
int main()
{
    float h[1024], *di, *d_o;
    for (int i = 0; i < 1024; i++) h[i] = (float)i + 1;
    hipMalloc(&di, 4096); hipMalloc(&d_o, 4096); hipMemcpy(di, h, 4096, hipMemcpyHostToDevice);
    k<<<1, 256>>>(di, d_o, 4096); hipMemcpy(h, d_o, 4096, hipMemcpyDeviceToHost);
    printf("lane0: %g %g %g %g | lane1 (OOB): %g %g %g %g | lane2: %g | lane3 (OOB): %g | lane 65 (OOB): %g lane 66: %g\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[12], h[65*4], h[66*4]);
    return 0;
}
