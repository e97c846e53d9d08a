// fetch_calib -- what rocprofv3's FETCH_SIZE reports for the read patterns the kernels of this repo use, each over 1 GiB of never-touched-before data
// (so nothing comes from a cache): the ratio counter / bytes is the calibration scripts/pmc_traffic.py applies (the guide gives 0.5 for contiguous 16-B-per-lane
// reads and calls every other pattern uncalibrated).
//   k_contig16   lane i reads the uint4 at i                                  (blur / fast staging, LDS-DMA)
//   k_stride32   lane i reads 2 uint4 at 32 i                                 (map_stream: depth)
//   k_stride48   lane i reads 3 uint4 at 48 i                                 (map_stream / class_bits: rgb, semantic)
//   k_dword      lane i reads the dword at i                                  (resize4)
//   k_stride12   lane i reads 3 dwords at 12 i                                (gray)
//   k_unalign16  lane i reads one unaligned uint4 at 16 i + 4                 (orient)
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench/fetch_calib.hip -o scripts/ubench/bin/fetch_calib
// run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o run -- scripts/ubench/bin/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_contig16(const uint4* p, size_t n, uint32_t* o) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { uint4 v = p[i]; if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) o[0] = 1; } }
__global__ void k_stride32(const uint4* p, size_t n, uint32_t* o) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { uint4 a = p[2 * i], b = p[2 * i + 1]; if ((a.x ^ b.y) == 0x12345u) o[0] = 1; } }
__global__ void k_stride48(const uint4* p, size_t n, uint32_t* o) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { uint4 a = p[3 * i], b = p[3 * i + 1], c = p[3 * i + 2]; if ((a.x ^ b.y ^ c.z) == 0x12345u) o[0] = 1; } }
__global__ void k_dword(const uint32_t* p, size_t n, uint32_t* o) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { if (p[i] == 0x12345u) o[0] = 1; } }
__global__ void k_stride12(const uint32_t* p, size_t n, uint32_t* o) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { uint32_t a = p[3 * i], b = p[3 * i + 1], c = p[3 * i + 2]; if ((a ^ b ^ c) == 0x12345u) o[0] = 1; } }
__global__ void k_unalign16(const uint8_t* p, size_t n, uint32_t* o) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { uint4 v; __builtin_memcpy(&v, p + 16 * i + 4, 16); if ((v.x ^ v.w) == 0x12345u) o[0] = 1; } }
__global__ void k_fill(uint32_t* p, size_t n, uint32_t seed) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; p[i] = x; } }
__global__ void k_gray_like(const uint32_t* p, size_t n, uint32_t* dst) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { uint32_t a = p[3 * i], b = p[3 * i + 1], c = p[3 * i + 2]; dst[i] = a + b * 3 + c * 5; } }
int main()
{
    const size_t B = 1ull << 30;
    uint8_t* buf[6]; uint32_t* o; hipMalloc(&o, 4);
    for (int i = 0; i < 6; i++) { hipMalloc(&buf[i], B + 64); hipMemset(buf[i], i + 1, B + 64); }
    // push the memset data out of the Infinity Cache: stream 2 GiB of something else
    uint8_t* junk; hipMalloc(&junk, 2 * B); hipMemset(junk, 7, 2 * B); hipDeviceSynchronize();
    size_t n;
    n = B / 16; k_contig16<<<(n + 255) / 256, 256>>>((const uint4*)buf[0], n, o);
    n = B / 32; k_stride32<<<(n + 255) / 256, 256>>>((const uint4*)buf[1], n, o);
    n = B / 48; k_stride48<<<(n + 255) / 256, 256>>>((const uint4*)buf[2], n, o);
    n = B / 4;  k_dword<<<(n + 255) / 256, 256>>>((const uint32_t*)buf[3], n, o);
    n = B / 12; k_stride12<<<(n + 255) / 256, 256>>>((const uint32_t*)buf[4], n, o);
    n = B / 16; k_unalign16<<<(n + 255) / 256, 256>>>(buf[5], n, o);
    hipDeviceSynchronize();
    // the same patterns on data a kernel wrote (hashed, not a constant fill), and a read + dword-write kernel (gray's shape)
    for (int i = 0; i < 3; i++) k_fill<<<(B / 4 + 255) / 256, 256>>>((uint32_t*)buf[i], B / 4, i);
    hipMemset(junk, 9, 2 * B); hipDeviceSynchronize();
    n = B / 16; k_contig16<<<(n + 255) / 256, 256>>>((const uint4*)buf[0], n, o);
    n = B / 48; k_stride48<<<(n + 255) / 256, 256>>>((const uint4*)buf[1], n, o);
    n = B / 12; k_gray_like<<<(n + 255) / 256, 256>>>((const uint32_t*)buf[2], n, (uint32_t*)buf[3]);
    hipDeviceSynchronize();
    printf("each kernel read %zu bytes (1 GiB = 1048576 KiB)\n", B);
    return 0;
}
