// kernels_map.hip -- K10, K11, K12 of the hot path for gfx950 (HBM-bound byte/integer kernels; no MFMA by design):
//   K10 Mapper::semantic_motion_fuse    /root/reference/src/mapper.cpp:189-216
//   K11 Mapper::generatePointCloud      /root/reference/src/mapper.cpp:12-94  (+ rgbdframe.h:63-75, transformPointCloud :90-91)
//   K12 pcl::VoxelGrid in Mapper::viewer /root/reference/src/mapper.cpp:106-107,154-155
// Contracts = oracle/mapper.c.  Batched over frames.
#include "ssm_internal.h"
#include <mutex>
#include <cstring>
#include <cmath>

// ------------------------------------------------------------------ K10: moving-class mask + 5x5 box dilate
#define MK_W 64
#define MK_H 16
__global__ void __launch_bounds__(256)
mask_kernel(const uint8_t* __restrict__ sem, int w, int h, int tiles_x, uint8_t* __restrict__ mask)
{
    __shared__ uint8_t m0[MK_H + 4][MK_W + 4];
    __shared__ uint8_t m1[MK_H + 4][MK_W];
    const int tx0 = (blockIdx.x % tiles_x) * MK_W, ty0 = (blockIdx.x / tiles_x) * MK_H;
    const uint8_t* s = sem + (size_t)blockIdx.y * w * h * 3;
    for (int i = threadIdx.x; i < (MK_H + 4) * (MK_W + 4); i += 256) {
        const int ly = i / (MK_W + 4), lx = i - ly * (MK_W + 4);
        const int gx = tx0 + lx - 2, gy = ty0 + ly - 2;
        uint8_t v = 0;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
            const uint8_t* p = s + ((size_t)gy * w + gx) * 3;
            const int b = p[0], g = p[1], r = p[2];
            v = ((b == 0 && g == 64 && r == 64) || (b == 192 && g == 128 && r == 0)) ? 255 : 0;   // pedestrian | cyclist
        }
        m0[ly][lx] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (MK_H + 4) * MK_W; i += 256) {
        const int ly = i / MK_W, lx = i - ly * MK_W;
        const uint8_t* p = &m0[ly][lx];
        m1[ly][lx] = p[0] | p[1] | p[2] | p[3] | p[4];
    }
    __syncthreads();
    uint8_t* dst = mask + (size_t)blockIdx.y * w * h;
    const int lx = threadIdx.x & 63;
    for (int ly = threadIdx.x >> 6; ly < MK_H; ly += 4) {
        const int gx = tx0 + lx, gy = ty0 + ly;
        if (gx < w && gy < h) dst[(size_t)gy * w + gx] = m1[ly][lx] | m1[ly+1][lx] | m1[ly+2][lx] | m1[ly+3][lx] | m1[ly+4][lx];
    }
}
hipError_t k_moving_mask(const uint8_t* sem, int n, int w, int h, uint8_t* mask, hipStream_t s)
{
    const int tx = (w + MK_W - 1) / MK_W, ty = (h + MK_H - 1) / MK_H;
    mask_kernel<<<dim3(tx * ty, n), 256, 0, s>>>(sem, w, h, tx, mask);
    return hipGetLastError();
}

// ------------------------------------------------------------------ K11: gated back-projection, ordered compaction
// chunk = 1024 consecutive row-major pixels (256 threads x 4).  count -> global exclusive scan -> emit, so the point
// list is in the reference's row-major order (the serial loop of mapper.cpp:21-86).
#define BP_PIX 1024
int backproject_chunks(int w, int h) { return (w * h + BP_PIX - 1) / BP_PIX; }

__device__ __forceinline__ bool bp_keep(int d, int mk, int b, int g, int r, double maxd)
{
    if (d == 0) return false;
    if ((double)d > maxd) return false;                    // mapper.cpp:30  d > max_distance * camera.scale
    if (mk == 255) return false;                           // mapper.cpp:32
    if ((b == 128 && g == 128 && r == 128) || (b == 128 && g == 192 && r == 192) || (b == 192 && g == 128 && r == 0)) return false;  // :41-55
    return true;
}
__global__ void __launch_bounds__(256)
bp_count_kernel(const uint16_t* __restrict__ depth, const uint8_t* __restrict__ sem, const uint8_t* __restrict__ mask,
                int npix, double maxd, int chunks, int32_t* __restrict__ chunk_cnt)
{
    __shared__ int wsum[4];
    const size_t fo = (size_t)blockIdx.y * npix;
    const int p0 = blockIdx.x * BP_PIX + threadIdx.x * 4;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int p = p0 + i;
        if (p < npix) {
            const uint8_t* s = sem + (fo + p) * 3;
            c += bp_keep(depth[fo + p], mask[fo + p], s[0], s[1], s[2], maxd);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[blockIdx.y * chunks + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// one block: exclusive scan of all n*chunks counts (int64 offsets), per-frame totals
__global__ void __launch_bounds__(1024)
bp_scan_kernel(const int32_t* __restrict__ chunk_cnt, int n, int chunks, int64_t* __restrict__ chunk_off,
               int32_t* __restrict__ npoints, int64_t* __restrict__ total)
{
    __shared__ long long part[1024];
    const int tot = n * chunks, tid = threadIdx.x;
    const int per = (tot + 1023) / 1024;
    const int b = tid * per, e = min(b + per, tot);
    long long s = 0;
    for (int i = b; i < e; i++) s += chunk_cnt[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        long long v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    long long run = part[tid] - s;
    for (int i = b; i < e; i++) { chunk_off[i] = run; run += chunk_cnt[i]; }
    if (tid == 1023) *total = part[1023];
    __syncthreads();
    // per-frame totals
    for (int f = tid; f < n; f += 1024) {
        int t = 0;
        for (int c = 0; c < chunks; c++) t += chunk_cnt[f * chunks + c];
        npoints[f] = t;
    }
}
__global__ void __launch_bounds__(256)
bp_emit_kernel(const uint16_t* __restrict__ depth, const uint8_t* __restrict__ rgb, const uint8_t* __restrict__ sem,
               const uint8_t* __restrict__ mask, const double* __restrict__ pose, int w, int npix, ssm_camera cam, double maxd,
               int chunks, const int64_t* __restrict__ chunk_off, ssm_point* __restrict__ out)
{
    __shared__ int wsum[4];
    const size_t fo = (size_t)blockIdx.y * npix;
    const int p0 = blockIdx.x * BP_PIX + threadIdx.x * 4;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bool k[4]; int c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int p = p0 + i; k[i] = false;
        if (p < npix) {
            const uint8_t* s = sem + (fo + p) * 3;
            k[i] = bp_keep(depth[fo + p], mask[fo + p], s[0], s[1], s[2], maxd);
        }
        c += k[i];
    }
    int inc = c;                                             // inclusive wave scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int off = inc - c;
    for (int i = 0; i < wv; i++) off += wsum[i];
    if (c == 0) return;
    ssm_point* dst = out + chunk_off[blockIdx.y * chunks + blockIdx.x] + off;
    double T[12];
    const bool hasT = pose != nullptr;
    if (hasT) { const double* P = pose + (size_t)blockIdx.y * 16;
#pragma unroll
        for (int j = 0; j < 4; j++) { T[3*j] = P[4*j]; T[3*j+1] = P[4*j+1]; T[3*j+2] = P[4*j+2]; } }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (!k[i]) continue;
        const int p = p0 + i;
        const int v = p / w, u = p - v * w;
        const int d = depth[fo + p];
        const float z = (float)((double)d / cam.scale);                            // rgbdframe.h:71-73
        const float x = (float)(((double)u - cam.cx) * (double)z / cam.fx);
        const float y = (float)(((double)v - cam.cy) * (double)z / cam.fy);
        float ox = x, oy = y, oz = z;
        if (hasT) {                                                                // pcl::transformPointCloud, double, left to right
            const double X = x, Y = y, Z = z;
            ox = (float)(T[0] * X + T[3] * Y + T[6] * Z + T[9]);
            oy = (float)(T[1] * X + T[4] * Y + T[7] * Z + T[10]);
            oz = (float)(T[2] * X + T[5] * Y + T[8] * Z + T[11]);
        }
        const uint8_t* c3 = rgb + (fo + p) * 3;
        const uint8_t* s3 = sem + (fo + p) * 3;
        const int sb = s3[0], sg = s3[1], sr = s3[2];
        uint32_t label = 255;
        // 12-class palette (BGR), SegNet driving_webdemo id order
        switch ((sb << 16) | (sg << 8) | sr) {
            case (128 << 16) | (128 << 8) | 128: label = 0; break;   case (0 << 16) | (0 << 8) | 128: label = 1; break;
            case (128 << 16) | (192 << 8) | 192: label = 2; break;   case (0 << 16) | (69 << 8) | 255: label = 3; break;
            case (128 << 16) | (64 << 8) | 128: label = 4; break;    case (222 << 16) | (40 << 8) | 60: label = 5; break;
            case (0 << 16) | (128 << 8) | 128: label = 6; break;     case (128 << 16) | (128 << 8) | 192: label = 7; break;
            case (128 << 16) | (64 << 8) | 64: label = 8; break;     case (128 << 16) | (0 << 8) | 64: label = 9; break;
            case (0 << 16) | (64 << 8) | 64: label = 10; break;      case (192 << 16) | (128 << 8) | 0: label = 11; break;
        }
        uint4 lo, hi;
        lo.x = __float_as_uint(ox); lo.y = __float_as_uint(oy); lo.z = __float_as_uint(oz); lo.w = __float_as_uint(1.0f);
        hi.x = (uint32_t)c3[0] | ((uint32_t)c3[1] << 8) | ((uint32_t)c3[2] << 16); hi.y = label; hi.z = 0; hi.w = 0;
        uint4* o = reinterpret_cast<uint4*>(dst);
        o[0] = lo; o[1] = hi;
        dst++;
    }
}
hipError_t k_backproject(const uint16_t* depth, const uint8_t* rgb, const uint8_t* sem, const uint8_t* mask,
                         const double* pose, int n, int w, int h, ssm_camera cam, double max_distance,
                         int32_t* chunk_cnt, int64_t* chunk_off, int32_t* npoints, int64_t* total,
                         ssm_point* out, hipStream_t s)
{
    const int npix = w * h, chunks = backproject_chunks(w, h);
    const double maxd = max_distance * cam.scale;
    bp_count_kernel<<<dim3(chunks, n), 256, 0, s>>>(depth, sem, mask, npix, maxd, chunks, chunk_cnt);
    bp_scan_kernel<<<1, 1024, 0, s>>>(chunk_cnt, n, chunks, chunk_off, npoints, total);
    bp_emit_kernel<<<dim3(chunks, n), 256, 0, s>>>(depth, rgb, sem, mask, pose, w, npix, cam, maxd, chunks, chunk_off, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ K12: voxel table of exact integer sums
// open-addressing hash (linear probing) keyed by the 63-bit voxel key; slot layout == ssm_voxel (112 B).  Exact int64
// sums make the result independent of arrival order, so plain device-scope atomics are bit-reproducible.  Points arrive
// in row-major pixel order, so a wave's 64 points fall into a few runs of equal key: a segmented wave scan reduces each
// run in registers and only the run's last lane touches memory.
// counter block (32 bytes behind the occupied-slot list): [0] occupied slots, [1] flags (1 = contributions were LOST: the table had no room and the overflow list
// was full too; 2 = points outside the key range were skipped), [2] records appended to the overflow list, [3] its capacity, [4..5] its address.
// A table never refuses a contribution as long as its overflow list has room: when the neighbourhood of a key's home slot is taken (VOX_PROBE_LIMIT slots: a table
// that is too full, or one the host has not grown yet), the whole contribution is appended to the list as one ssm_voxel record and the host merges the list into the
// (grown) table the next time it settles the map (ssm_map.hip map_settle) -- exact integer sums: when and where a contribution is added does not matter.
#define VOX_PROBE_LIMIT 128u
__device__ __forceinline__ uint32_t vox_hash(int64_t key)
{
    uint64_t z = (uint64_t)key * 0x9E3779B97F4A7C15ULL;
    z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 32;
    return (uint32_t)z;
}
__device__ __forceinline__ ssm_voxel* vox_find_or_insert(ssm_voxel* tab, int cap_log2, int64_t key, int32_t* counters, uint32_t* occ)
{
    const uint32_t mask = (1u << cap_log2) - 1u;
    const uint32_t limit = mask < VOX_PROBE_LIMIT - 1u ? mask : VOX_PROBE_LIMIT - 1u;
    uint32_t slot = vox_hash(key) & mask;
    for (uint32_t probe = 0; probe <= limit; probe++, slot = (slot + 1) & mask) {
        unsigned long long* kp = reinterpret_cast<unsigned long long*>(&tab[slot].key);
        unsigned long long cur = __hip_atomic_load(kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == (unsigned long long)key) return &tab[slot];
        if (cur == (unsigned long long)SSM_VOX_EMPTY) {
            const unsigned long long prev = atomicCAS(kp, (unsigned long long)SSM_VOX_EMPTY, (unsigned long long)key);
            if (prev == (unsigned long long)SSM_VOX_EMPTY) { const int i = atomicAdd(&counters[0], 1); occ[i] = slot; return &tab[slot]; }
            if (prev == (unsigned long long)key) return &tab[slot];
        }
    }
    return nullptr;
}
// a fresh record of the overflow list (the caller writes all of it), or nullptr when the list is full too (flag bit 0: the contribution is lost).
// (round 5, measured: with a GENERIC list pointer the record went out as flat_store instructions, and their presence in map_stream2_kernel's run loop -- a flat
// access counts against the LDS counter as well as the memory counter -- cost the kernel 4 %: 1.60 -> 1.66 us per frame; g_voxel below is what fixed it, not the
// placement of the cold code: out of line, behind __builtin_expect or inline made no difference)
typedef __attribute__((address_space(1))) ssm_voxel g_voxel;      // the overflow list is DEVICE memory: a pointer loaded from memory would be generic, its stores `flat_store`s,
                                                                  // which wait on the LDS counter too -- inside map_stream2_kernel's loop that cost the kernel 4 %
__device__ __forceinline__ g_voxel* vox_overflow_slot(int32_t* counters)
{
    const int cap = counters[3];
    g_voxel* buf = (g_voxel*)*reinterpret_cast<ssm_voxel* const*>(counters + 4);
    const int i = cap > 0 ? atomicAdd(&counters[2], 1) : cap;
    if (i >= cap) { atomicOr(&counters[1], 1); return nullptr; }
    return buf + i;
}
__device__ __forceinline__ void vox_store_record(g_voxel* o, const ssm_voxel& sv)
{
    o->key = sv.key; o->sx = sv.sx; o->sy = sv.sy; o->sz = sv.sz; o->sr = sv.sr; o->sg = sv.sg; o->sb = sv.sb; o->n = sv.n;
    for (int c = 0; c < 12; c++) o->hist[c] = sv.hist[c];
}
// the same for an entry of a block's LDS table: label votes as six packed 16-bit pairs
__device__ __forceinline__ void vox_overflow_packed(int32_t* counters, long long key, long long sx, long long sy, long long sz, unsigned r, unsigned g, unsigned b, unsigned n, const unsigned* hist6)
{
    unsigned h[6];
    for (int c = 0; c < 6; c++) h[c] = hist6[c];
    g_voxel* o = vox_overflow_slot(counters);
    if (!o) return;
    o->key = key; o->sx = sx; o->sy = sy; o->sz = sz; o->sr = r; o->sg = g; o->sb = b; o->n = n;
    for (int c = 0; c < 12; c++) o->hist[c] = (h[c >> 1] >> (16 * (c & 1))) & 0xFFFF;
}
__device__ __forceinline__ void vox_overflow_one(int32_t* counters, long long key, long long sx, long long sy, long long sz, unsigned long long sr, unsigned long long sg,
                                              unsigned long long sb, unsigned long long n, uint32_t lab)
{
    g_voxel* o = vox_overflow_slot(counters);
    if (!o) return;
    o->key = key; o->sx = sx; o->sy = sy; o->sz = sz; o->sr = sr; o->sg = sg; o->sb = sb; o->n = n;
#pragma unroll
    for (int c = 0; c < 12; c++) o->hist[c] = (uint32_t)c == lab ? (uint32_t)n : 0u;
}
__device__ __forceinline__ void vox_add(ssm_voxel* v, long long sx, long long sy, long long sz, unsigned long long sr, unsigned long long sg,
                                        unsigned long long sb, unsigned long long n)
{
    atomicAdd(reinterpret_cast<unsigned long long*>(&v->sx), (unsigned long long)sx);
    atomicAdd(reinterpret_cast<unsigned long long*>(&v->sy), (unsigned long long)sy);
    atomicAdd(reinterpret_cast<unsigned long long*>(&v->sz), (unsigned long long)sz);
    atomicAdd(reinterpret_cast<unsigned long long*>(&v->sr), sr);
    atomicAdd(reinterpret_cast<unsigned long long*>(&v->sg), sg);
    atomicAdd(reinterpret_cast<unsigned long long*>(&v->sb), sb);
    atomicAdd(reinterpret_cast<unsigned long long*>(&v->n), n);
}
__global__ void vox_clear_kernel(ssm_voxel* __restrict__ tab, uint32_t* __restrict__ occ, int32_t* __restrict__ counters, int full, unsigned slots)
{
    // clears only the occupied slots (or everything when `full`)
    const unsigned n = full ? slots : (unsigned)counters[0];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        ssm_voxel* v = &tab[full ? i : occ[i]];
        uint4* p = reinterpret_cast<uint4*>(v);
        const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int k = 1; k < 7; k++) p[k] = z;
        p[0] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0);
    }
}
__global__ void vox_reset_counters(int32_t* counters) { counters[0] = 0; counters[1] = 0; counters[2] = 0; counters[6] = 0; counters[7] = 0; }
hipError_t k_voxel_clear(ssm_voxel* tab, int cap_log2, int32_t* counters, hipStream_t s)
{
    // cap_log2 < 0  =>  full clear of 2^-cap_log2 slots (first use)
    const int full = cap_log2 < 0; const unsigned slots = 1u << (full ? -cap_log2 : cap_log2);
    uint32_t* occ = reinterpret_cast<uint32_t*>(tab + slots);
    vox_clear_kernel<<<2048, 256, 0, s>>>(tab, occ, counters, full, slots);
    vox_reset_counters<<<1, 1, 0, s>>>(counters);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256)
vox_insert_kernel(const ssm_point* __restrict__ pts, const int64_t* __restrict__ n_dev, int64_t n_max, float inv_leaf,
                  ssm_voxel* __restrict__ tab, int cap_log2, int32_t* __restrict__ counters)
{
    const int64_t n = n_dev ? min(*n_dev, n_max) : n_max;
    uint32_t* occ = reinterpret_cast<uint32_t*>(tab + (1u << cap_log2));
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += stride) {
        const int64_t i = base + threadIdx.x;
        const bool valid = i < n;
        long long key = -2, sx = 0, sy = 0, sz = 0; uint32_t rg = 0, bn = 0, label = 255;
        if (valid) {
            const uint4* p = reinterpret_cast<const uint4*>(pts + i);
            const uint4 lo = p[0], hi = p[1];
            const float x = __uint_as_float(lo.x), y = __uint_as_float(lo.y), z = __uint_as_float(lo.z);
            const float fi = floorf(x * inv_leaf), fj = floorf(y * inv_leaf), fk = floorf(z * inv_leaf);
            const long long vi = (long long)(int)fi + (1 << 20), vj = (long long)(int)fj + (1 << 20), vk = (long long)(int)fk + (1 << 20);     // v_cvt_i32_f32 (saturating): an index the range check below rejects may be anything; float -> int64 costs ~8 instructions each
            key = (vk << 42) | (vj << 21) | vi;
            // range contract (oracle/mapper.c sso_voxel_key): index not finite or outside (-2^20, 2^20) -> the point is skipped, flag bit 1
            if (!(fabsf(fi) < 1048576.0f && fabsf(fj) < 1048576.0f && fabsf(fk) < 1048576.0f)) { key = -1; atomicOr(&counters[1], 2); }
            sx = __double2ll_rn((double)x * 16777216.0); sy = __double2ll_rn((double)y * 16777216.0); sz = __double2ll_rn((double)z * 16777216.0);
            rg = ((hi.x >> 16) & 255) | (((hi.x >> 8) & 255) << 16);       // r | g<<16
            bn = (hi.x & 255) | (1u << 16);                               // b | n<<16
            label = hi.y;
        }
        // runs of equal key inside the wave
        const long long kprev = __shfl_up(key, 1, 64);
        const bool head = lane == 0 || kprev != key;
        const unsigned long long heads = __ballot(head);
        const int start = 63 - __clzll(heads & (~0ull >> (63 - lane)));
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long ax = __shfl_up(sx, o, 64), ay = __shfl_up(sy, o, 64), az = __shfl_up(sz, o, 64);
            const uint32_t arg = __shfl_up(rg, o, 64), abn = __shfl_up(bn, o, 64);
            if (lane - o >= start) { sx += ax; sy += ay; sz += az; rg += arg; bn += abn; }
        }
        const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
        // label votes of the run: 12 ballots masked by the run
        unsigned long long lb[12];
#pragma unroll
        for (int c = 0; c < 12; c++) lb[c] = __ballot(label == (uint32_t)c);
        if (valid && tail && key >= 0) {
            ssm_voxel* v = vox_find_or_insert(tab, cap_log2, key, counters, occ);
            const unsigned long long run = (~0ull >> (63 - lane)) & (~0ull << start);
            if (v) {
                vox_add(v, sx, sy, sz, rg & 0xFFFF, rg >> 16, bn & 0xFFFF, bn >> 16);
#pragma unroll
                for (int c = 0; c < 12; c++) { const int k = __popcll(lb[c] & run); if (k) atomicAdd(&v->hist[c], (uint32_t)k); }
            } else if (g_voxel* o = vox_overflow_slot(counters)) {
                o->key = key; o->sx = sx; o->sy = sy; o->sz = sz; o->sr = rg & 0xFFFF; o->sg = rg >> 16; o->sb = bn & 0xFFFF; o->n = bn >> 16;
#pragma unroll
                for (int c = 0; c < 12; c++) o->hist[c] = (uint32_t)__popcll(lb[c] & run);
            }
        }
    }
}
hipError_t k_voxel_insert(const ssm_point* pts, const int64_t* n_dev, int64_t n_max, float leaf, ssm_voxel* tab,
                          int cap_log2, int32_t* counters, hipStream_t s)
{
    if (n_max <= 0) return hipSuccess;
    int64_t blocks = (n_max + 255) / 256; if (blocks > 8192) blocks = 8192;
    vox_insert_kernel<<<(int)blocks, 256, 0, s>>>(pts, n_dev, n_max, 1.0f / leaf, tab, cap_log2, counters);
    return hipGetLastError();
}
// ------------------------------------------------------------------ K10+K11+K12, streaming form (w % 16 == 0): map_stream2_kernel below.
// One thread = 16 consecutive pixels of a row; consecutive pixels with the same (voxel, label) are summed in registers, the run tails update a block-local LDS hash
// (256 slots for the 3 x 4096 pixels of a block) and the block flushes it with one global atomic group per voxel.  The point list of generatePointCloud is never
// written: exact integer sums make the map independent of order, so the result is bit-identical to mask -> backproject -> insert (tests/test_gpu_parity.py).
// (Rounds 1-5 also carried the first form of this fusion -- class_bits_kernel + vdilate_bits_kernel + map_stream_kernel, every pixel through the full arithmetic,
// two wave-wide segmented scans per chunk: 12 % slower, profiles/r03_*; removed in round 6, the compacting kernel covers every width the path accepts.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_mov0_ll(long long v)
{
    const int lo = dpp_mov0<CTRL, ROW_MASK>((int)(unsigned)(unsigned long long)v), hi = dpp_mov0<CTRL, ROW_MASK>((int)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
struct RunAcc { long long sx, sy, sz; uint32_t r, g, b, n; };
// inclusive segmented scan over runs that start at lane `start` (start <= lane): the run's LAST lane ends with the run total.
// DPP only (VALU): row_shr:1/2/4/8 inside rows of 16, then row_bcast:15 and row_bcast:31 across rows.
__device__ __forceinline__ void run_scan(RunAcc& a, int lane, int start)
{
#define RS_STEP(CTRL, MASK, COND)                                                                                   \
    {   const long long tx = dpp_mov0_ll<CTRL, MASK>(a.sx), ty = dpp_mov0_ll<CTRL, MASK>(a.sy), tz = dpp_mov0_ll<CTRL, MASK>(a.sz); \
        const uint32_t t1 = (uint32_t)dpp_mov0<CTRL, MASK>((int)a.r), t2 = (uint32_t)dpp_mov0<CTRL, MASK>((int)a.g);             \
        const uint32_t t3 = (uint32_t)dpp_mov0<CTRL, MASK>((int)a.b), t4 = (uint32_t)dpp_mov0<CTRL, MASK>((int)a.n);             \
        if (COND) { a.sx += tx; a.sy += ty; a.sz += tz; a.r += t1; a.g += t2; a.b += t3; a.n += t4; } }
    const int li = lane & 15, row = lane >> 4;
    RS_STEP(0x111, 0xF, (li >= 1 && lane - 1 >= start))
    RS_STEP(0x112, 0xF, (li >= 2 && lane - 2 >= start))
    RS_STEP(0x114, 0xF, (li >= 4 && lane - 4 >= start))
    RS_STEP(0x118, 0xF, (li >= 8 && lane - 8 >= start))
    RS_STEP(0x142, 0xA, ((row & 1) && start < 16 * row))
    RS_STEP(0x143, 0xC, (row >= 2 && start < 32))
#undef RS_STEP
}
// llrint for |v| < 2^51: adding 1.5 * 2^52 rounds to the nearest-even integer in the FP adder and leaves it in the low
// mantissa bits (3 instructions instead of the generic f64 -> i64 conversion sequence)
__device__ __forceinline__ long long f64_to_ll_rn(double v)
{
    const double M = 6755399441055744.0;
    return __double_as_longlong(v + M) - __double_as_longlong(M);
}
// slot k of the palette's perfect hash ((bgr * 0x7589a82b) >> 28): label << 24 | b | g << 8 | r << 16; empty slots hold label 255
__device__ __forceinline__ uint32_t label_hash_entry(int k)
{
    switch (k) {
        case 10: return (0u << 24) | 128u | (128u << 8) | (128u << 16);   case 1:  return (1u << 24) | 0u | (0u << 8) | (128u << 16);
        case 2:  return (2u << 24) | 128u | (192u << 8) | (192u << 16);   case 9:  return (3u << 24) | 0u | (69u << 8) | (255u << 16);
        case 4:  return (4u << 24) | 128u | (64u << 8) | (128u << 16);    case 13: return (5u << 24) | 222u | (40u << 8) | (60u << 16);
        case 14: return (6u << 24) | 0u | (128u << 8) | (128u << 16);     case 11: return (7u << 24) | 128u | (128u << 8) | (192u << 16);
        case 3:  return (8u << 24) | 128u | (64u << 8) | (64u << 16);     case 12: return (9u << 24) | 128u | (0u << 8) | (64u << 16);
        case 7:  return (10u << 24) | 0u | (64u << 8) | (64u << 16);      case 15: return (11u << 24) | 192u | (128u << 8) | (0u << 16);
    }
    return 0xFF000001u;
}
#define MS_SLOTS 256
#ifndef MS_MINB
#define MS_MINB 4        // blocks per CU the register allocation is held to (16 waves per CU)
#endif
#define MS_CH 3                // 4096-pixel chunks (consecutive rows of one frame) a block accumulates in its LDS table before the flush
struct LdsVox { long long key, sx, sy, sz; unsigned r, g, b, n; unsigned hist[6]; };
__device__ __forceinline__ uint32_t label_of_bgr24(uint32_t bgr)     // b | g<<8 | r<<16
{
    switch (bgr) {
        case 128u | (128u << 8) | (128u << 16): return 0;   case 0u | (0u << 8) | (128u << 16): return 1;
        case 128u | (192u << 8) | (192u << 16): return 2;   case 0u | (69u << 8) | (255u << 16): return 3;
        case 128u | (64u << 8) | (128u << 16): return 4;    case 222u | (40u << 8) | (60u << 16): return 5;
        case 0u | (128u << 8) | (128u << 16): return 6;     case 128u | (128u << 8) | (192u << 16): return 7;
        case 128u | (64u << 8) | (64u << 16): return 8;     case 128u | (0u << 8) | (64u << 16): return 9;
        case 0u | (64u << 8) | (64u << 16): return 10;      case 192u | (128u << 8) | (0u << 16): return 11;
    }
    return 255;
}
// the block table is full (a leaf far below the pixel footprint): the run goes straight to the global table, or to the context's overflow list
__device__ __forceinline__ void vox_direct(long long key, uint32_t lab, long long sx, long long sy, long long sz, unsigned r, unsigned g, unsigned b, unsigned n,
                ssm_voxel* tab, int cap_log2, int32_t* counters, uint32_t* occ)
{
    ssm_voxel* v = vox_find_or_insert(tab, cap_log2, key, counters, occ);
    if (__builtin_expect(v != nullptr, 1)) { vox_add(v, sx, sy, sz, r, g, b, n); if (lab < 12) atomicAdd(&v->hist[lab], n); }
    else vox_overflow_one(counters, key, sx, sy, sz, r, g, b, n, lab);
}
__device__ __forceinline__ void lds_vox_update(LdsVox* lt, long long key, uint32_t lab, const RunAcc& f,
                                               ssm_voxel* tab, int cap_log2, int32_t* counters, uint32_t* occ)
{
    uint32_t slot = vox_hash(key) & (MS_SLOTS - 1);
    LdsVox* e = nullptr;
    for (int probe = 0; probe < MS_SLOTS; probe++, slot = (slot + 1) & (MS_SLOTS - 1)) {
        const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&lt[slot].key), (unsigned long long)SSM_VOX_EMPTY, (unsigned long long)key);
        if (prev == (unsigned long long)SSM_VOX_EMPTY || prev == (unsigned long long)key) { e = &lt[slot]; break; }
    }
    if (__builtin_expect(e != nullptr, 1)) {
        atomicAdd(reinterpret_cast<unsigned long long*>(&e->sx), (unsigned long long)f.sx);
        atomicAdd(reinterpret_cast<unsigned long long*>(&e->sy), (unsigned long long)f.sy);
        atomicAdd(reinterpret_cast<unsigned long long*>(&e->sz), (unsigned long long)f.sz);
        atomicAdd(&e->r, f.r); atomicAdd(&e->g, f.g); atomicAdd(&e->b, f.b); atomicAdd(&e->n, f.n);
        if (lab < 12) atomicAdd(&e->hist[lab >> 1], f.n << (16 * (lab & 1)));
    } else vox_direct(key, lab, f.sx, f.sy, f.sz, f.r, f.g, f.b, f.n, tab, cap_log2, counters, occ);
}
// merge equal neighbouring runs across the wave and push the tails into the block table
__device__ __forceinline__ void wave_flush(LdsVox* lt, long long key, uint32_t lab, RunAcc f, int lane,
                                           ssm_voxel* tab, int cap_log2, int32_t* counters, uint32_t* occ)
{
    const long long kprev = dpp_mov0_ll<0x138, 0xF>(key);      // wave_shr:1
    const uint32_t lprev = (uint32_t)dpp_mov0<0x138, 0xF>((int)lab);
    const bool head = lane == 0 || kprev != key || lprev != lab;
    const unsigned long long heads = __ballot(head);
    const int start = 63 - __clzll(heads & (~0ull >> (63 - lane)));
    run_scan(f, lane, start);
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    if (key >= 0 && tail) lds_vox_update(lt, key, lab, f, tab, cap_log2, counters, occ);
}
// The three divisions of the unprojection (d / scale, n / fx, n / fy) without the IEEE division sequence and with the same bits
// (FASTDIV instantiation, chosen on the host): q = n r with r = RN(1 / f), then two Markstein corrections q += fma(-q, f, n) r.
// With a correctly rounded reciprocal the first makes q faithful and the second makes it the correctly rounded quotient
// (Markstein 1990), provided the significand of f is not all ones, which the host checks together with every d / scale.
// 5 f64 instructions instead of ~11 per division; f64 runs at half rate, and these were a quarter of the kernel's f64 work.
// Checked exhaustively on the CPU for the cameras of the tests and benches (all 16-bit depths x 1300 columns x 500 rows: 7e8
// divisions, no mismatch) and by tests/test_gpu_parity.py against the division-based ordered path (bp_emit_kernel).
struct MapDiv { double rscale, rfx, rfy; };
__device__ __forceinline__ double markstein_div(double n, double f, double r)
{
    double q = n * r;
    q = fma(fma(-q, f, n), r, q);
    return fma(fma(-q, f, n), r, q);
}
// ---- map_stream2_kernel: the same fusion with the kept pixels COMPACTED before the expensive part.
// The first form (map_stream_kernel, removed) ran unprojection / pose transform / voxel key (about 170 of its 250 VALU instructions per pixel, most of them f64) for all 16
// pixels of a lane as soon as any lane of the wave keeps that pixel; on the configs[1] stream 59 % of the pixels pass the gates, so 41 % of that work
// is masked out.  Here a wave first decides (labels, depth range, class gates, moving mask: ~15 instructions per pixel), writes its lanes' depth and
// kept pixels as 14-bit entries (source lane, pixel in the lane, label) to an LDS list in wave-scan order; then lane i takes the
// list entries [i P, (i + 1) P), P = ceil(kept / 64): every lane works on kept pixels only, still consecutive in scan order, so the register run
// accumulation works as before.  Exact integer sums: the map is bit-identical.
// Round 3: the moving-class bits are made HERE (class_bits_kernel + vdilate_bits_kernel and their bit images are gone from this path): the block first
// classifies the semantic rows of its three chunks plus two rows above and below (the vertical reach of the 5 x 5 dilate) into an LDS bit image -- the same
// cache lines its gate pass reads right after -- and the gate pass ORs five rows x three words of it instead of loading three pre-dilated words.
#define MS2_MINB 4
// Lossless under any rate (round 6).  The table is sized from the stream's own rate (ssm_map.hip map_before_launch); when a launch brings far more voxels than
// that (the camera leaves a near wall at a fine leaf), contributions the table refuses go to the context's overflow list, and a full list would DROP them.  So a
// block that STARTS while the list holds more than `hw` records adds nothing at all: it logs (launch tag, frame, block) in the skip list and exits; the host
// grows the map at its next look at the counters and launches exactly the logged blocks again (REDO: block ids from a list, no check).  A skipped block has
// contributed nothing and its inputs are still where they were, so the redo is exact.  hw = list capacity - (blocks that can be resident: 256 CUs x MS2_MINB,
// or the grid if smaller) x (records one block can append: one per pixel, MS_CH x 4096): after the last block that passed the check, only blocks resident at
// that moment can still append, so the list never overflows and nothing is ever dropped.  Cost: one cached load per block.
#define MS2_SKIP_CAP 65536
#define MS2_CB_WORDS (MS_CH * 256 + 6 * 256)      // 16-pixel words of (rows of the block + 4 halo rows + partial first / last row) for widths up to 4096
template <bool FASTDIV, bool REDO>
__global__ void __launch_bounds__(256, MS2_MINB)
map_stream2_kernel(const uint16_t* __restrict__ depth, const uint8_t* __restrict__ rgb, const uint8_t* __restrict__ sem,
                   const double* __restrict__ pose, int w, int h, ssm_camera cam, MapDiv md, double maxd,
                   float inv_leaf, ssm_voxel* __restrict__ tab, int cap_log2, int32_t* __restrict__ counters, int32_t* __restrict__ npoints, uint32_t mul_wpr,
                   int32_t* __restrict__ skip, int hw, int tag, int gx)
{
    __shared__ LdsVox lt[MS_SLOTS];
    __shared__ int s_npts, s_skip;
    // REDO: `skip` holds the ids to run again, one per block of this launch: (tag << 24) | (frame * gx + block)
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    if (REDO) { const int id = skip[blockIdx.x] & 0xFFFFFF; by_ = id / gx; bx_ = id - by_ * gx; }
    const int bx = bx_, by = by_;
    if (!REDO && threadIdx.x == 0) {
        int sk = 0;
        if (__hip_atomic_load(&counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > hw) {
            const int i = atomicAdd(&counters[6], 1);
            if (i < MS2_SKIP_CAP) skip[i] = (tag << 24) | (by * gx + bx); else atomicOr(&counters[1], 1);      // (the host keeps grid <= MS2_SKIP_CAP / 4: cannot happen)
            sk = 1;
        }
        s_skip = sk;
    }
    __shared__ uint16_t vlist[4][1024];                                 // per wave: kept pixels in scan order: lane << 4 | pixel | label << 10
    __shared__ uint16_t cbits[MS2_CB_WORDS];                            // pedestrian | cyclist bit per pixel, rows ry0 .. of this frame, wpr words per row
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wpr = w >> 4, words = wpr * h;
    // ---- labels and class bits.  The wave classifies the semantic words of its three chunks ONCE: the 4-bit labels stay in registers for the gate pass, the
    // moving-class bit of every pixel goes to the LDS bit image.  The rest of the bit image -- 2 rows above and below the block's rows and the parts of its
    // first / last row that belong to the neighbouring blocks -- is classified by all threads with the two compares the bit needs (rows outside the image: no
    // moving pixel).  (Before: every row of the block was read and classified here and read and hashed again in the gate pass.)
    const int bw0 = bx * MS_CH * 256;
    const int ry0 = (int)__umulhi((uint32_t)bw0, mul_wpr) - 2;
    const int own0 = bw0 - ry0 * wpr, own1 = min(bw0 + MS_CH * 256, words) - ry0 * wpr;      // the block's own words inside the bit image
    const uint32_t tab_entry = label_hash_entry(lane & 15);
    uint32_t labs[MS_CH][2];                                            // 4-bit labels (15 = none of the palette) of the lane's 16 pixels, per chunk
#pragma unroll
    for (int ch = 0; ch < MS_CH; ch++) {
        labs[ch][0] = labs[ch][1] = 0u;
        const int wi = (bx * MS_CH + ch) * 256 + wv * 64 + lane;
        // EVERY lane of the wave runs the classification (a lane past the frame's last word re-reads that word and stores nothing): the palette table is
        // read with a wave shuffle from lanes 0 .. 15, which must not be masked off when the frame's words end inside this wave (found by
        // tests/test_gpu_fuzz.py at 176 x 88: 968 words, the last wave has 8 live lanes and lanes 8 .. 15 returned stale table entries)
        const bool inw = wi < words;
        if ((bx * MS_CH + ch) * 256 + wv * 64 < words) {      // wave-uniform
            const uint4* ps = reinterpret_cast<const uint4*>(sem + ((size_t)by * words + (inw ? wi : words - 1)) * 48);
            const uint4 S0 = ps[0], S1 = ps[1], S2 = ps[2];
            const uint32_t ss[13] = {S0.x, S0.y, S0.z, S0.w, S1.x, S1.y, S1.z, S1.w, S2.x, S2.y, S2.z, S2.w, 0u};
            uint32_t bits = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int o = 3 * k;
                const uint32_t sbgr = ((o & 3) ? __builtin_amdgcn_alignbyte(ss[(o >> 2) + 1], ss[o >> 2], o & 3) : ss[o >> 2]) & 0xFFFFFFu;
                const uint32_t ent = (uint32_t)__shfl((int)tab_entry, (int)((sbgr * 0x7589a82bu) >> 28), 64);
                const uint32_t lab = (ent & 0xFFFFFFu) == sbgr ? ent >> 24 : 15u;          // 0..11, or 15
                labs[ch][k >> 3] |= lab << (4 * (k & 7));
                bits |= (uint32_t)(lab == 10u || lab == 11u) << k;                          // pedestrian (0,64,64) | cyclist (192,128,0) BGR
            }
            if (inw) cbits[wi - ry0 * wpr] = (uint16_t)bits;
        }
    }
    {
        const int ry1 = (int)__umulhi((uint32_t)(min(bw0 + MS_CH * 256, words) - 1), mul_wpr) + 2;
        const int ncw = (ry1 - ry0 + 1) * wpr, nhalo = ncw - (own1 - own0);
        for (int j = tid; j < nhalo; j += 256) {
            const int i = j < own0 ? j : j + (own1 - own0);              // the bit-image words that are not the block's own
            const int rr = (int)__umulhi((uint32_t)i, mul_wpr), gy = ry0 + rr;
            uint32_t bits = 0;
            if (gy >= 0 && gy < h) {
                const uint4* p = reinterpret_cast<const uint4*>(sem + ((size_t)by * words + (size_t)((long long)ry0 * wpr + i)) * 48);
                const uint4 a = p[0], b = p[1], c = p[2];
                const uint32_t d[13] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, 0u};
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int o = 3 * k;
                    const uint32_t bgr = ((o & 3) ? __builtin_amdgcn_alignbyte(d[(o >> 2) + 1], d[o >> 2], o & 3) : d[o >> 2]) & 0xFFFFFFu;
                    bits |= (uint32_t)(bgr == (0u | (64u << 8) | (64u << 16)) || bgr == (192u | (128u << 8) | (0u << 16))) << k;   // (0,64,64) | (192,128,0) BGR
                }
            }
            cbits[i] = (uint16_t)bits;
        }
    }
    for (int i = tid; i < MS_SLOTS; i += 256) {
        lt[i].key = SSM_VOX_EMPTY; lt[i].sx = 0; lt[i].sy = 0; lt[i].sz = 0; lt[i].r = lt[i].g = lt[i].b = lt[i].n = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) lt[i].hist[k] = 0;
    }
    if (tid == 0) s_npts = 0;
    __syncthreads();
    if (!REDO && s_skip) return;                                       // (block-uniform; nothing has been added anywhere yet)
    uint32_t* occ = reinterpret_cast<uint32_t*>(tab + (1u << cap_log2));
    int kept = 0; bool out_of_range = false;
    double T[12];
    const bool hasT = pose != nullptr;
    if (hasT) { const double* P = pose + (size_t)by * 16;
#pragma unroll
        for (int j = 0; j < 4; j++) { T[3*j] = P[4*j]; T[3*j+1] = P[4*j+1]; T[3*j+2] = P[4*j+2]; } }
    const int dmax = maxd >= 65535.0 ? 65535 : (int)maxd;          // integer d > maxd  <=>  d > floor(maxd)
#pragma unroll 1
    for (int ch = 0; ch < MS_CH; ch++) {
    const int wbase = (bx * MS_CH + ch) * 256 + wv * 64;    // the wave's first 16-pixel word of this frame
    const int wi = wbase + lane;
    uint32_t keepbits = 0;
    uint32_t lab4[2] = {labs[0][0], labs[0][1]};                     // (the chunk loop is not unrolled: the chunk's label registers by selects)
#pragma unroll
    for (int c_ = 1; c_ < MS_CH; c_++) { lab4[0] = ch == c_ ? labs[c_][0] : lab4[0]; lab4[1] = ch == c_ ? labs[c_][1] : lab4[1]; }
    if (wi < words) {
        const size_t gw = (size_t)by * words + wi;
        const int gy = (int)__umulhi((uint32_t)wi, mul_wpr), xw = wi - gy * wpr;
        const uint4* pd = reinterpret_cast<const uint4*>(depth + gw * 16);
        const uint4 D0 = pd[0], D1 = pd[1];
        // vertical OR of rows gy - 2 .. gy + 2 for the word and its two neighbours, then the 5-wide horizontal OR
        const uint16_t* cb = cbits + (gy - ry0 - 2) * wpr + xw;
        uint32_t vl = 0, vc = 0, vr = 0;
#pragma unroll
        for (int r5 = 0; r5 < 5; r5++) { vc |= cb[r5 * wpr]; if (xw > 0) vl |= cb[r5 * wpr - 1]; if (xw + 1 < wpr) vr |= cb[r5 * wpr + 1]; }
        const unsigned long long win = (unsigned long long)vl | ((unsigned long long)vc << 16) | ((unsigned long long)vr << 32);
        const uint32_t moving = (uint32_t)((win >> 14) | (win >> 15) | (win >> 16) | (win >> 17) | (win >> 18)) & 0xFFFFu;   // 5-wide OR
        const uint32_t dd[8] = {D0.x, D0.y, D0.z, D0.w, D1.x, D1.y, D1.z, D1.w};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int d = (dd[k >> 1] >> (16 * (k & 1))) & 0xFFFF;
            const uint32_t lab = (lab4[k >> 3] >> (4 * (k & 7))) & 15u;
            const bool gated = (0x805u >> lab) & 1u;                                  // sky 0, pole 2, cyclist 11 (bit 15 is clear)
            keepbits |= (uint32_t)(d != 0 && d <= dmax && !gated) << k;
        }
        keepbits &= ~moving;                                          // mapper.cpp:32
    }
    // the wave's kept pixels, in scan order
    const uint32_t cnt = __popc(keepbits);
    uint32_t incl = cnt;
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false);     // row_shr:1
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false);     // row_shr:2
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false);     // row_shr:4
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false);     // row_shr:8
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
    const int V = __builtin_amdgcn_readlane((int)incl, 63);
    kept += (int)cnt;
    {
        int base = (int)(incl - cnt);
#pragma unroll
        for (int k = 0; k < 16; k++)
            if ((keepbits >> k) & 1u) vlist[wv][base++] = (uint16_t)((lane << 4) | k | (((lab4[k >> 3] >> (4 * (k & 7))) & 15u) << 10));
    }
    const int P = (V + 63) >> 6;                                    // wave-uniform
    long long k0 = -2, k1 = -2; uint32_t l0 = 255, l1 = 255; RunAcc a0, a1;
    a0.sx = a0.sy = a0.sz = 0; a0.r = a0.g = a0.b = a0.n = 0; a1 = a0;
    __builtin_amdgcn_wave_barrier();                                // (LDS operations of a wave execute in order: the reads below see the writes above)
    // one kept pixel: list entry -> (voxel key, label, fixed-point sums + colour); straight-line code, so that the MS2_UN pixels of a loop trip interleave
    // (one pixel alone is a chain of ~60 dependent f64 instructions)
    // a kept pixel's list entry and its depth / colour from global memory (the wave read these lines in the gate pass: cache hits; an LDS copy would hold
    // the block to three per CU); fetched one loop trip ahead of the arithmetic
    struct PixIn { uint32_t ent; int d; uint32_t c01, c2; };
    auto fetch = [&](int e, PixIn& q) {
        q.ent = vlist[wv][min(e, V - 1)];                               // (only called with V >= 1; entries past the end repeat the last one)
        const int sl = (q.ent >> 4) & 63, k = q.ent & 15;
        const size_t gp = ((size_t)by * words + wbase + sl) * 16 + k;
        q.d = depth[gp];
        uint16_t c01; __builtin_memcpy(&c01, rgb + gp * 3, 2);
        q.c01 = c01; q.c2 = rgb[gp * 3 + 2];
    };
    auto pixel = [&](int e, const PixIn& q, long long& key, uint32_t& lab, RunAcc& p) {
        const uint32_t ent = q.ent;
        const int sl = (ent >> 4) & 63, k = ent & 15;
        lab = (ent >> 10) == 15u ? 255u : (ent >> 10);
        const int d = q.d;
        const uint32_t cbgr = q.c01 | (q.c2 << 16);
        const int wip = wbase + sl;
        const int gy = (int)__umulhi((uint32_t)wip, mul_wpr), gx = ((wip - gy * wpr) << 4) + k;
        const double yf = (double)gy - cam.cy;
        float x, y, z;
        if (FASTDIV) {
            z = (float)markstein_div((double)d, cam.scale, md.rscale);
            x = (float)markstein_div(((double)gx - cam.cx) * (double)z, cam.fx, md.rfx);
            y = (float)markstein_div(yf * (double)z, cam.fy, md.rfy);
        } else {
            z = (float)((double)d / cam.scale);
            x = (float)(((double)gx - cam.cx) * (double)z / cam.fx);
            y = (float)(yf * (double)z / cam.fy);
        }
        float ox = x, oy = y, oz = z;
        if (hasT) {
            const double X = x, Y = y, Z = z;
            ox = (float)(T[0] * X + T[3] * Y + T[6] * Z + T[9]);
            oy = (float)(T[1] * X + T[4] * Y + T[7] * Z + T[10]);
            oz = (float)(T[2] * X + T[5] * Y + T[8] * Z + T[11]);
        }
        const float fi = floorf(ox * inv_leaf), fj = floorf(oy * inv_leaf), fk = floorf(oz * inv_leaf);
        const long long vi = (long long)(int)fi + (1 << 20), vj = (long long)(int)fj + (1 << 20), vk = (long long)(int)fk + (1 << 20);     // v_cvt_i32_f32 (saturating): an index the range check below rejects may be anything; float -> int64 costs ~8 instructions each
        key = (vk << 42) | (vj << 21) | vi;
        // range contract (oracle/mapper.c sso_voxel_key): such a point still counts in npoints (generatePointCloud emits it) but is not fused
        if (!(fabsf(fi) < 1048576.0f && fabsf(fj) < 1048576.0f && fabsf(fk) < 1048576.0f)) { key = -1; if (e < V) out_of_range = true; }
        p.sx = f64_to_ll_rn((double)ox * 16777216.0); p.sy = f64_to_ll_rn((double)oy * 16777216.0); p.sz = f64_to_ll_rn((double)oz * 16777216.0);
        p.b = cbgr & 255; p.g = (cbgr >> 8) & 255; p.r = cbgr >> 16; p.n = 1;
    };
    auto accumulate = [&](long long key, uint32_t lab, const RunAcc& p) {
        if (k1 == -2 && (k0 == -2 || (k0 == key && l0 == lab))) {                 // still in the first run
            k0 = key; l0 = lab; a0.sx += p.sx; a0.sy += p.sy; a0.sz += p.sz; a0.r += p.r; a0.g += p.g; a0.b += p.b; a0.n += 1;
        } else if (k1 == -2 || (k1 == key && l1 == lab)) {                       // second run
            k1 = key; l1 = lab; a1.sx += p.sx; a1.sy += p.sy; a1.sz += p.sz; a1.r += p.r; a1.g += p.g; a1.b += p.b; a1.n += 1;
        } else {                                                                 // a third run inside the lane's share: rare
            if (k1 >= 0) lds_vox_update(lt, k1, l1, a1, tab, cap_log2, counters, occ);
            k1 = key; l1 = lab; a1 = p;
        }
    };
    constexpr int MS2_UN = 2;
    PixIn cur[MS2_UN], nxt[MS2_UN];
    if (P > 0) {
#pragma unroll
        for (int u = 0; u < MS2_UN; u++) fetch(lane * P + u, nxt[u]);
    }
#pragma unroll 1
    for (int i = 0; i < P; i += MS2_UN) {
        long long kk[MS2_UN]; uint32_t ll[MS2_UN]; RunAcc pp[MS2_UN];
#pragma unroll
        for (int u = 0; u < MS2_UN; u++) { cur[u] = nxt[u]; fetch(lane * P + i + MS2_UN + u, nxt[u]); }
#pragma unroll
        for (int u = 0; u < MS2_UN; u++) pixel(lane * P + i + u, cur[u], kk[u], ll[u], pp[u]);
#pragma unroll
        for (int u = 0; u < MS2_UN; u++) if (i + u < P && lane * P + i + u < V) accumulate(kk[u], ll[u], pp[u]);
    }
    // every lane's (at most two) runs straight into the block table: with ~10 kept pixels per lane the wave-wide segmented scans that merged neighbouring
    // lanes first (map_stream_kernel: two scans of ten dwords, ~700 instructions per chunk) cost more than the LDS atomics they saved
    if (k0 >= 0) lds_vox_update(lt, k0, l0, a0, tab, cap_log2, counters, occ);
    if (k1 >= 0) lds_vox_update(lt, k1, l1, a1, tab, cap_log2, counters, occ);
    __builtin_amdgcn_wave_barrier();                                // the next chunk overwrites this wave's pix / vlist
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o, 64);
    if (lane == 0 && kept) atomicAdd(&s_npts, kept);
    if (__ballot(out_of_range) && lane == 0) atomicOr(&counters[1], 2);
    __syncthreads();
    for (int i = tid; i < MS_SLOTS; i += 256) {
        if (lt[i].key == SSM_VOX_EMPTY) continue;
        ssm_voxel* v = vox_find_or_insert(tab, cap_log2, lt[i].key, counters, occ);
        if (__builtin_expect(!v, 0)) { vox_overflow_packed(counters, lt[i].key, lt[i].sx, lt[i].sy, lt[i].sz, lt[i].r, lt[i].g, lt[i].b, lt[i].n, lt[i].hist); continue; }
        vox_add(v, lt[i].sx, lt[i].sy, lt[i].sz, lt[i].r, lt[i].g, lt[i].b, lt[i].n);
#pragma unroll
        for (int c = 0; c < 12; c++) { const uint32_t k = (lt[i].hist[c >> 1] >> (16 * (c & 1))) & 0xFFFF; if (k) atomicAdd(&v->hist[c], k); }
    }
    if (tid == 0 && s_npts) atomicAdd(&npoints[by], s_npts);
}
// MapDiv for a camera (see the struct): the reciprocal form when the divisors allow it; the check is cached per camera
static bool map_div_for(const ssm_camera& cam, MapDiv& md)
{
    static std::mutex mu; static ssm_camera seen = {0, 0, 0, 0, 0}; static bool seen_ok = false;
    md.rscale = 1.0 / cam.scale; md.rfx = 1.0 / cam.fx; md.rfy = 1.0 / cam.fy;
    std::lock_guard<std::mutex> lk(mu);
    if (seen.scale != cam.scale || seen.fx != cam.fx || seen.fy != cam.fy) {
        auto plain = [](double f) {                       // normal, positive, significand not all ones
            uint64_t b; memcpy(&b, &f, 8);
            const uint64_t man = b & 0xFFFFFFFFFFFFFull; const int ex = (int)((b >> 52) & 0x7FF);
            return f > 0 && ex > 0 && ex < 0x7FF && man != 0xFFFFFFFFFFFFFull;
        };
        bool ok = plain(cam.scale) && plain(cam.fx) && plain(cam.fy);
        for (int d = 0; d < 65536 && ok; d++) {           // every depth value through the same arithmetic on the host
            double q = (double)d * md.rscale; q = std::fma(std::fma(-q, cam.scale, (double)d), md.rscale, q); q = std::fma(std::fma(-q, cam.scale, (double)d), md.rscale, q);
            if (q != (double)d / cam.scale) ok = false;
        }
        seen = cam; seen_ok = ok;
    }
    return seen_ok;
}
int k_map_fuse_blocks_per_frame(int w, int h) { const int words = (w >> 4) * h; return (words + 256 * MS_CH - 1) / (256 * MS_CH); }
int k_map_fuse_block_records(void) { return MS_CH * 4096; }            // overflow records one block can append at most: one per pixel
int k_map_fuse_resident_blocks(void) { return 256 * MS2_MINB; }
int k_map_fuse_skip_cap(void) { return MS2_SKIP_CAP; }
// n frames of w x h (w % 16 == 0, w <= 4096).  skip / hw / tag: see map_stream2_kernel (skip = the context's skip list, MS2_SKIP_CAP entries; counters[6] counts them).
// nredo > 0: run exactly the blocks redo_ids[0 .. nredo) (device) of an earlier launch with these arguments again (its per-frame point counts go on counting).
hipError_t k_map_fuse(const uint16_t* depth, const uint8_t* rgb, const uint8_t* sem, const double* pose, int n, int w, int h,
                      ssm_camera cam, double max_distance, float leaf,
                      ssm_voxel* tab, int cap_log2, int32_t* counters, int32_t* npoints, hipStream_t s, int32_t* skip, int hw, int tag, const int32_t* redo_ids, int nredo)
{
    const int wpr = w >> 4, words = wpr * h;
    if (!((long long)words * wpr < (1ll << 32) && wpr <= 256)) return hipErrorInvalidValue;      // frames wider than 4096 pixels: not supported by the fused map stage
    if (nredo <= 0) { const hipError_t e = hipMemsetAsync(npoints, 0, sizeof(int32_t) * n, s); if (e != hipSuccess) return e; }
    MapDiv md; const bool fast = map_div_for(cam, md);
    const int gx = (words + 256 * MS_CH - 1) / (256 * MS_CH);
    const dim3 grid(gx, n);
    const uint32_t mul_wpr = (uint32_t)(((1ull << 32) + wpr - 1) / wpr);           // floor(i / wpr) = umulhi(i, mul) for i < words (i * wpr < 2^32)
    const double maxd = max_distance * cam.scale; const float il = 1.0f / leaf;
    if (nredo > 0) {
        int32_t* ids = const_cast<int32_t*>(redo_ids);
        if (fast) map_stream2_kernel<true, true><<<nredo, 256, 0, s>>>(depth, rgb, sem, pose, w, h, cam, md, maxd, il, tab, cap_log2, counters, npoints, mul_wpr, ids, 0, tag, gx);
        else map_stream2_kernel<false, true><<<nredo, 256, 0, s>>>(depth, rgb, sem, pose, w, h, cam, md, maxd, il, tab, cap_log2, counters, npoints, mul_wpr, ids, 0, tag, gx);
    } else {
        if (fast) map_stream2_kernel<true, false><<<grid, 256, 0, s>>>(depth, rgb, sem, pose, w, h, cam, md, maxd, il, tab, cap_log2, counters, npoints, mul_wpr, skip, hw, tag, gx);
        else map_stream2_kernel<false, false><<<grid, 256, 0, s>>>(depth, rgb, sem, pose, w, h, cam, md, maxd, il, tab, cap_log2, counters, npoints, mul_wpr, skip, hw, tag, gx);
    }
    return hipGetLastError();
}

__global__ void vox_merge_kernel(const ssm_voxel* __restrict__ src, int n, ssm_voxel* __restrict__ tab, int cap_log2, int32_t* __restrict__ counters)
{
    uint32_t* occ = reinterpret_cast<uint32_t*>(tab + (1u << cap_log2));
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ssm_voxel sv = src[i];
    ssm_voxel* v = vox_find_or_insert(tab, cap_log2, sv.key, counters, occ);
    if (!v) { if (g_voxel* o = vox_overflow_slot(counters)) vox_store_record(o, sv); return; }
    vox_add(v, sv.sx, sv.sy, sv.sz, sv.sr, sv.sg, sv.sb, sv.n);
    for (int c = 0; c < 12; c++) if (sv.hist[c]) atomicAdd(&v->hist[c], sv.hist[c]);
}
// every occupied slot of one table into another (the host grows the map: ssm_map.hip map_settle)
__global__ void vox_rehash_kernel(const ssm_voxel* __restrict__ src, const uint32_t* __restrict__ src_occ, const int32_t* __restrict__ src_counters,
                                  ssm_voxel* __restrict__ tab, int cap_log2, int32_t* __restrict__ counters)
{
    uint32_t* occ = reinterpret_cast<uint32_t*>(tab + (1u << cap_log2));
    const int n = src_counters[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const ssm_voxel sv = src[src_occ[i]];
        ssm_voxel* v = vox_find_or_insert(tab, cap_log2, sv.key, counters, occ);
        if (!v) { if (g_voxel* o = vox_overflow_slot(counters)) vox_store_record(o, sv); continue; }
        vox_add(v, sv.sx, sv.sy, sv.sz, sv.sr, sv.sg, sv.sb, sv.n);
        for (int c = 0; c < 12; c++) if (sv.hist[c]) atomicAdd(&v->hist[c], sv.hist[c]);
    }
}
hipError_t k_voxel_rehash(const ssm_voxel* src, int src_cap_log2, ssm_voxel* tab, int cap_log2, int32_t* counters, hipStream_t s)
{
    const uint32_t* socc = reinterpret_cast<const uint32_t*>(src + ((size_t)1 << src_cap_log2));
    const int32_t* scnt = reinterpret_cast<const int32_t*>(socc + ((size_t)1 << src_cap_log2));
    vox_rehash_kernel<<<1024, 256, 0, s>>>(src, socc, scnt, tab, cap_log2, counters);
    return hipGetLastError();
}
hipError_t k_voxel_merge(const ssm_voxel* src, int n, ssm_voxel* tab, int cap_log2, int32_t* counters, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    vox_merge_kernel<<<(n + 255) / 256, 256, 0, s>>>(src, n, tab, cap_log2, counters);
    return hipGetLastError();
}
// gather the occupied slots (unordered) into a dense array
__global__ void vox_compact_kernel(const ssm_voxel* __restrict__ tab, const uint32_t* __restrict__ occ, const int32_t* __restrict__ counters,
                                   ssm_voxel* __restrict__ out, int32_t* __restrict__ n_out)
{
    const int n = counters[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint4* s = reinterpret_cast<const uint4*>(&tab[occ[i]]);
        uint4* d = reinterpret_cast<uint4*>(&out[i]);
#pragma unroll
        for (int k = 0; k < 7; k++) d[k] = s[k];
    }
}
hipError_t k_voxel_compact(const ssm_voxel* tab, int cap_log2, ssm_voxel* out, int32_t* n_out, hipStream_t s)
{
    const uint32_t* occ = reinterpret_cast<const uint32_t*>(tab + (1u << cap_log2));
    const int32_t* counters = reinterpret_cast<const int32_t*>(occ + (1u << cap_log2));
    vox_compact_kernel<<<1024, 256, 0, s>>>(tab, occ, counters, out, n_out);
    return hipGetLastError();
}
__global__ void vox_gather_points_kernel(const ssm_voxel* __restrict__ c, const uint32_t* __restrict__ order, int n, ssm_point* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ssm_voxel v = c[order[i]];
    const double nn = (double)v.n;
    const float x = (float)(((double)v.sx / nn) * (1.0 / 16777216.0));
    const float y = (float)(((double)v.sy / nn) * (1.0 / 16777216.0));
    const float z = (float)(((double)v.sz / nn) * (1.0 / 16777216.0));
    const uint32_t r = (uint32_t)(v.sr / v.n), g = (uint32_t)(v.sg / v.n), b = (uint32_t)(v.sb / v.n);
    uint32_t best = 0, lab = 255;
#pragma unroll
    for (int k = 0; k < 12; k++) if (v.hist[k] > best) { best = v.hist[k]; lab = k; }
    uint4 lo, hi;
    lo.x = __float_as_uint(x); lo.y = __float_as_uint(y); lo.z = __float_as_uint(z); lo.w = __float_as_uint(1.0f);
    hi.x = b | (g << 8) | (r << 16); hi.y = lab; hi.z = 0; hi.w = 0;
    uint4* o = reinterpret_cast<uint4*>(out + i);
    o[0] = lo; o[1] = hi;
}
hipError_t k_voxel_gather_points(const ssm_voxel* compact, const uint32_t* order, int n, ssm_point* out, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    vox_gather_points_kernel<<<(n + 255) / 256, 256, 0, s>>>(compact, order, n, out);
    return hipGetLastError();
}
__global__ void vox_gather_table_kernel(const ssm_voxel* __restrict__ c, const uint32_t* __restrict__ order, int n, ssm_voxel* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* s = reinterpret_cast<const uint4*>(&c[order[i]]);
    uint4* d = reinterpret_cast<uint4*>(&out[i]);
#pragma unroll
    for (int k = 0; k < 7; k++) d[k] = s[k];
}
hipError_t k_voxel_gather_table(const ssm_voxel* compact, const uint32_t* order, int n, ssm_voxel* out, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    vox_gather_table_kernel<<<(n + 255) / 256, 256, 0, s>>>(compact, order, n, out);
    return hipGetLastError();
}
// bounding box of a cloud (for pcl::VoxelGrid's index-overflow guard); minmax6 pre-set to +inf x3, -inf x3 as ordered ints
__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__global__ void vox_bounds_kernel(const ssm_point* __restrict__ pts, int n, int* __restrict__ mm)
{
    int mn[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = *reinterpret_cast<const float4*>(pts + i);
        const int a[3] = {f2ord(p.x), f2ord(p.y), f2ord(p.z)};
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = min(mn[k], a[k]); mx[k] = max(mx[k], a[k]); }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn[k] = min(mn[k], __shfl_xor(mn[k], o, 64)); mx[k] = max(mx[k], __shfl_xor(mx[k], o, 64)); }
    }
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 3; k++) { atomicMin(&mm[k], mn[k]); atomicMax(&mm[3 + k], mx[k]); }
}
__global__ void vox_bounds_init(int* mm) { mm[0] = mm[1] = mm[2] = 0x7FFFFFFF; mm[3] = mm[4] = mm[5] = (int)0x80000000; }
hipError_t k_voxel_bounds(const ssm_point* pts, int n, float* minmax6, hipStream_t s)
{
    vox_bounds_init<<<1, 1, 0, s>>>(reinterpret_cast<int*>(minmax6));
    if (n > 0) vox_bounds_kernel<<<min((n + 255) / 256, 1024), 256, 0, s>>>(pts, n, reinterpret_cast<int*>(minmax6));
    return hipGetLastError();
}

// ---- pcl::transformPointCloud on a device-resident cloud (Mapper::generatePointCloud, /root/reference/src/mapper.cpp:89-91): dst[i] = T src[i] with
// x' = float(t00 x + t01 y + t02 z + t03) in double, left to right -- the arithmetic of k_backproject's pose step and of include/ssm/mapper.h's host loop
struct PoseArg { double t[12]; };                                 // column-major 3 x 4: t[3 c + r]
__global__ void __launch_bounds__(256)
cloud_transform_kernel(const ssm_point* __restrict__ src, int n, PoseArg P, ssm_point* __restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ssm_point p = src[i];
    const double x = p.x, y = p.y, z = p.z;
    p.x = (float)(P.t[0] * x + P.t[3] * y + P.t[6] * z + P.t[9]);
    p.y = (float)(P.t[1] * x + P.t[4] * y + P.t[7] * z + P.t[10]);
    p.z = (float)(P.t[2] * x + P.t[5] * y + P.t[8] * z + P.t[11]);
    dst[i] = p;
}
// T: 16 doubles, column-major 4 x 4 (HOST pointer: the pose travels as a kernel argument); nullptr = copy
hipError_t k_cloud_transform(const ssm_point* src, int n, const double* T, ssm_point* dst, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    if (!T) return hipMemcpyAsync(dst, src, (size_t)n * sizeof(ssm_point), hipMemcpyDeviceToDevice, s);
    PoseArg P;
    for (int c = 0; c < 4; c++) for (int r = 0; r < 3; r++) P.t[3 * c + r] = T[4 * c + r];
    cloud_transform_kernel<<<(n + 255) / 256, 256, 0, s>>>(src, n, P, dst);
    return hipGetLastError();
}

