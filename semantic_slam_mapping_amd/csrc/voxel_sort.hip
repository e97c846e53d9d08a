// voxel_sort.hip -- orders the occupied voxels by key for export (pcl::VoxelGrid emits centroids sorted by its linear
// voxel index, which is the (k,j,i) order of the key).  M voxels, not P points: off the per-frame hot path, so the
// library radix sort (rocPRIM, native AMD) is used rather than a hand-written one.
#include "ssm_internal.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

__global__ void vox_keys_kernel(const ssm_voxel* __restrict__ c, int n, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = (uint64_t)c[i].key; idx[i] = (uint32_t)i; }
}
hipError_t voxel_sort_pairs(void* tmp, size_t* tmp_bytes, const ssm_voxel* compact, int n, uint64_t* keys_a, uint64_t* keys_b,
                            uint32_t* idx_a, uint32_t* idx_b, hipStream_t s)
{
    if (tmp == nullptr)
        return rocprim::radix_sort_pairs(nullptr, *tmp_bytes, keys_a, keys_b, idx_a, idx_b, (size_t)n, 0, 63, s);
    if (n <= 0) return hipSuccess;
    vox_keys_kernel<<<(n + 255) / 256, 256, 0, s>>>(compact, n, keys_a, idx_a);
    return rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_a, keys_b, idx_a, idx_b, (size_t)n, 0, 63, s);
}
