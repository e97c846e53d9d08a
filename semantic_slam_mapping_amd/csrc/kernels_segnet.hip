// kernels_segnet.hip -- K9: the SegNet forward the reference runs through Caffe (Classifier::Predict,
// /root/reference/src/segnet.cpp:87-108: net_->ForwardPrefilled()) for the driving_webdemo network: VGG-16 encoder
// (13 conv3x3 pad 1 + BN + ReLU, 5 max-pool 2x2 s2 CEIL with arg-max mask), mirrored decoder (5 mask-driven Upsample,
// 13 conv), last conv -> 12 classes, ArgMax.  This is the ONLY place on the path where MFMA is used:
//   conv = implicit GEMM  D[cout][pixel] += W[cout][k] X[k][pixel], k = (tap, Cin), on v_mfma_f32_32x32x16_f16 (fp16 storage,
//   fp32 accumulate); activations fp16 in channel-chunked layout [n][C/32][H][W][32] and weights pre-packed per (Cout tile,
//   Cin chunk) so that every staging transfer is a contiguous 16-byte piece and a lane's 8-element K fragment is one
//   ds_read_b128; operands reach LDS by buffer_load .. lds (LDS-DMA) in a persistent, double-buffered kernel; BN (folded
//   scale/shift) + ReLU, the 2x2 max-pool with its arg-max codes and the final class ArgMax are epilogue variants
//   (DESIGN.md s.4.1).
// Also here: Classifier::Preprocess (segnet.cpp:130-167; cv::resize to 480x360, planar float, mean 0) and the label
// colouring of experiment/segnet.cpp:80-83,131-146 (Pavement->Road remap, cv::resize back to the frame size, cv::LUT).
#include "ssm_internal.h"
#include <map>
#include <mutex>
#include <utility>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------ pre-processing: BGR u8 frame -> 480x360 [H][W][8] fp16 (B, G, R, zeros)
// cv::resize INTER_LINEAR 8u per channel (same fixed-point contract as the ORB pyramid), then float (exact in fp16), mean 0
__global__ void __launch_bounds__(256)
segnet_prep_kernel(const uint8_t* __restrict__ bgr, int sw, int sh, int dw, int dh,
                   const int32_t* __restrict__ xofs, const int16_t* __restrict__ xa, const int32_t* __restrict__ yofs, const int16_t* __restrict__ ya,
                   _Float16* __restrict__ out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= dw * dh) return;
    const int y = p / dw, x = p - y * dw;
    const uint8_t* src = bgr + (size_t)blockIdx.y * sw * sh * 3;
    const int sy0 = yofs[y], sy1 = min(sy0 + 1, sh - 1), b0 = ya[2*y], b1 = ya[2*y+1];
    const int sx0 = xofs[x], sx1 = min(sx0 + 1, sw - 1), a0 = xa[2*x], a1 = xa[2*x+1];
    half8 lo;
#pragma unroll
    for (int k = 0; k < 8; k++) lo[k] = (_Float16)0.f;
    // the two neighbours of a source row are six consecutive bytes: ONE 8-byte load per row instead of six byte loads (the kernel spent 0.70 of its time in the
    // texture-address units); the last pixels of the frame, whose window would end past the buffer, and the clamped right border take the byte form
    if (sx1 == sx0 + 1 && ((size_t)sy1 * sw + sx0) * 3 + 8 <= (size_t)sw * sh * 3) {
        unsigned long long r0, r1;
        __builtin_memcpy(&r0, src + ((size_t)sy0 * sw + sx0) * 3, 8); __builtin_memcpy(&r1, src + ((size_t)sy1 * sw + sx0) * 3, 8);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int h0 = (int)((r0 >> (8 * c)) & 255u) * a0 + (int)((r0 >> (8 * c + 24)) & 255u) * a1;
            const int h1 = (int)((r1 >> (8 * c)) & 255u) * a0 + (int)((r1 >> (8 * c + 24)) & 255u) * a1;
            const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            lo[c] = (_Float16)(float)(v & 255);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int h0 = src[((size_t)sy0 * sw + sx0) * 3 + c] * a0 + src[((size_t)sy0 * sw + sx1) * 3 + c] * a1;
            const int h1 = src[((size_t)sy1 * sw + sx0) * 3 + c] * a0 + src[((size_t)sy1 * sw + sx1) * 3 + c] * a1;
            const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            lo[c] = (_Float16)(float)(v & 255);
        }
    }
    *reinterpret_cast<half8*>(out + ((size_t)blockIdx.y * dw * dh + p) * 8) = lo;           // [n][H][W][8]: B, G, R, 5 zeros (conv3x3_first_kernel's input)
}

// ------------------------------------------------------------------ conv3x3 pad 1 (+ scale/shift + ReLU): implicit GEMM on MFMA
// A/B operand maps (cdna guide s.3): lane l holds A[row l&31][k 8(l>>5)..+7] and B[k 8(l>>5)..+7][col l&31];
// C/D: col = l&31, row = (reg&3) + 8(reg>>2) + 4(l>>5).
// Per stage of KC = 32 input channels a block stages in LDS
//   * the input halo tile as 4 planes of 8 channels: plane[c8][pixel][8 x f16]
//   * the weights [tap][c8][cout 64][8 x f16]
// and runs 9 taps x 2 K-steps x 4 MFMAs per wave with every operand fragment coming from ONE ds_read_b128: a wave owns two
// output rows (2 tiles of 32 contiguous pixels -> conflict-free 512-byte reads) x 2 tiles of 32 output channels.
#define CT_N 64
#define CT_KC 32
// scripts/ubench only (-DCT_ABL_MFMA16): the same FLOPs, operand registers and LDS traffic issued as two v_mfma_f32_16x16x32_f16
// per 32x32x16 (results are garbage) -- measures what clock the other MFMA shape would hold inside this kernel
#ifdef CT_ABL_MFMA16
typedef float floatx4_abl __attribute__((ext_vector_type(4)));
__device__ __forceinline__ floatx16 mfma_abl(half8 a, half8 b, floatx16 c)
{
    floatx4_abl c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
    c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3]; c[4] = c1[0]; c[5] = c1[1]; c[6] = c1[2]; c[7] = c1[3];
    return c;
}
#define CT_MFMA(a, b, c) mfma_abl(a, b, c)
#else
#define CT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif
// ------------------------------------------------------------------ conv epilogue shared by the MFMA kernels below
// The MFMAs run with the weights as the row operand, so lane (r = pixel x, hh) holds acc[tm][tn][4g+q] = channel
// 32 tn + 8 g + 4 hh + q of rows y0 + tm.  BN scale/shift (s_ss, in LDS, zero for padding channels) and ReLU are applied,
// the two halves of the wave trade quads (v_permlane32_swap) so that a lane owns 8 consecutive channels, and
//   EPI 0: the rows are stored (16-byte buffer stores),
//   EPI 1: the 2x2 max-pool of the tile and its arg-max codes are stored instead.
// Dead lanes (and a padding 32-channel chunk) get an out-of-range offset, which the buffer store drops, so every wave issues
// exactly 8 store instructions per tile (the callers' vmcnt(8) relies on it).
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
template <bool RELU, int EPI, int NTN = 2, typename RS>
__device__ __forceinline__ void conv_epilogue(const floatx16 (&acc)[2][2], const float (*s_ss)[CT_N], const RS& rsO, const RS& rsC,
                                              int f, int y0, int gx, bool live0, bool live1, int chunk0, int cout_chunks, int H, int W, int r, int hh)
{
#pragma unroll
    for (int tn = 0; tn < NTN; tn++) {                                        // NTN = 1: a 32-channel output tile
        const int chunk = chunk0 + tn;
#pragma unroll
        for (int gp = 0; gp < 2; gp++) {
            uint4v vec[2];                                                    // [row] 8 consecutive channels, fp16
#pragma unroll
            for (int tm = 0; tm < 2; tm++) {
                unsigned pk[2][2];                                            // [quad of the pair][half2]
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int g = 2 * gp + e, cl = 32 * tn + 8 * g + 4 * hh;
                    const float4 sc = *reinterpret_cast<const float4*>(&s_ss[0][cl]), sf = *reinterpret_cast<const float4*>(&s_ss[1][cl]);
                    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, sfv[4] = {sf.x, sf.y, sf.z, sf.w};
                    // one FMA per value, conversion two at a time (v_cvt_pk_f16_f32, round to nearest even), ReLU on the packed
                    // halves (v_pk_max_f16): max(cvt(x), 0) == cvt(max(x, 0)) because the conversion is monotonic and keeps 0
                    typedef float float2v __attribute__((ext_vector_type(2)));
                    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        const float2v val = {__builtin_fmaf(acc[tm][tn][4 * g + q], scv[q], sfv[q]), __builtin_fmaf(acc[tm][tn][4 * g + q + 1], scv[q + 1], sfv[q + 1])};
                        half2v h2 = __builtin_convertvector(val, half2v);
                        if (RELU) h2 = __builtin_elementwise_max(h2, (half2v){(_Float16)0, (_Float16)0});
                        memcpy(&pk[e][q >> 1], &h2, 4);
                    }
                }
                // lanes 0-31 keep quad 2gp (channels +0..3) and receive the upper half's quad 2gp (+4..7);
                // lanes 32-63 receive the lower half's quad 2gp+1 (+8..11) and keep their own (+12..15)
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                vec[tm].x = s0[0]; vec[tm].y = s1[0]; vec[tm].z = s0[1]; vec[tm].w = s1[1];
            }
            if (EPI == 0) {
#pragma unroll
                for (int tm = 0; tm < 2; tm++) {
                    const bool live = tm ? live1 : live0;
                    const unsigned ob = live && chunk < cout_chunks ? (unsigned)((((f * cout_chunks + chunk) * H + y0 + tm) * W + gx) * 64 + 16 * hh + 32 * gp) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(vec[tm], rsO, ob, 0, 0);
                }
            } else {
                // fused 2x2 / stride 2 max-pool with arg-max code (pool2x2_kernel's contract: window scanned row-major,
                // strict '>' so the first maximum wins; windows are clipped at the right / bottom edge).  The window of an
                // even lane is its own two rows and those of lane + 1 (quad_perm [1,0,3,2]).
                uint4v nb[2];
#pragma unroll
                for (int tm = 0; tm < 2; tm++) {
                    nb[tm].x = __builtin_amdgcn_mov_dpp(vec[tm].x, 0xB1, 0xF, 0xF, true); nb[tm].y = __builtin_amdgcn_mov_dpp(vec[tm].y, 0xB1, 0xF, 0xF, true);
                    nb[tm].z = __builtin_amdgcn_mov_dpp(vec[tm].z, 0xB1, 0xF, 0xF, true); nb[tm].w = __builtin_amdgcn_mov_dpp(vec[tm].w, 0xB1, 0xF, 0xF, true);
                }
                const bool right = gx + 1 < W;
                _Float16 best[8]; unsigned char bc[8];
                _Float16 c01[8], c10[8], c11[8];
                memcpy(best, &vec[0], 16); memcpy(c01, &nb[0], 16); memcpy(c10, &vec[1], 16); memcpy(c11, &nb[1], 16);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    bc[k] = 0;
                    if (right && c01[k] > best[k]) { best[k] = c01[k]; bc[k] = 1; }
                    if (live1 && c10[k] > best[k]) { best[k] = c10[k]; bc[k] = 2; }
                    if (live1 && right && c11[k] > best[k]) { best[k] = c11[k]; bc[k] = 3; }
                }
                const int PH = (H + 1) >> 1, PW = (W + 1) >> 1;
                const bool plive = live0 && !(r & 1) && chunk < cout_chunks;
                const unsigned pidx = (unsigned)(((f * cout_chunks + chunk) * PH + (y0 >> 1)) * PW + (gx >> 1)) * 32u + 8u * hh + 16u * gp;   // elements
                uint4v pv; memcpy(&pv, best, 16);
                typedef unsigned uint2v __attribute__((ext_vector_type(2)));
                uint2v cv; memcpy(&cv, bc, 8);
                __builtin_amdgcn_raw_buffer_store_b128(pv, rsO, plive ? pidx * 2u : 0x80000000u, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(cv, rsC, plive ? pidx : 0x80000000u, 0, 0);
            }
        }
    }
}

// ------------------------------------------------------------------ conv3x3, LDS-DMA staged: the scheme of the kernel the network runs on (conv3x3_dma2_kernel)
// The implicit GEMM and LDS images described above, with
//   * one 512-thread block (8 waves) per CU owns a 32 x 16 pixel tile x 64 output channels, so a staged weight slab is
//     shared by twice the pixels (L2 -> LDS bytes per flop -35 %);
//   * operands go global -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write pass): the LDS image
//     of a stage is lane-linear (wave-instruction j fills chunks 64j..64j+63), halo pixels outside the image are
//     out-of-range buffer offsets, which the DMA writes as zeros;
//   * two LDS buffers: the DMA of stage k+1 is issued, one instruction per MFMA step, under the MFMAs of stage k; one
//     barrier per stage;
//   * the frames of a batch are stacked into one virtual image of n x VH rows (VH = H rounded up to even, + >= 1 zero row,
//     so vertical halos never leak between frames); 16-row tiles run over the stack: the 23-row layers waste 4 %, not 28 %;
//   * MFMA operands are swapped (D rows = output channels, columns = pixels) so a lane ends up with 4 consecutive output
//     channels of one pixel per accumulator quad: 8-byte stores, 4 per 32 channels;
//   * blockIdx.x = pixel_tile * n_cout_tiles + cout_tile: the cout tiles of one pixel tile are neighbours in launch order,
//     i.e. spread round-robin over the 8 XCDs, so every XCD's L2 keeps re-serving the same 64-channel weight slab.
#define DT_W 32
#define DT_H 16
#define DT_PW (DT_W + 2)
#define DT_PH (DT_H + 2)
#define DT_PLANE (DT_PH * DT_PW)              // 612 pixels per 8-channel plane
#define DT_ACH (4 * DT_PLANE)                 // 2448 16-byte chunks of input per stage
#define DT_AINS 40                            // A wave-instructions per stage (5 per wave; 2560 slots, the tail is padding)
#define DT_BCH (9 * 4 * CT_N)                 // 2304 chunks of weights per stage = 36 wave-instructions
#define DT_STAGE (DT_AINS * 64 + DT_BCH)      // chunks per LDS buffer (77,824 B)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
// EPI 0: out = conv (+BN, ReLU).  EPI 1: out = the 2x2 max-pooled conv, code = the arg-max codes; the full-resolution
// activation is never written.
// EPI 2 (with NT = 1: one 32-channel output tile, half the weight staging and MFMAs): the class ArgMax of the last layer --
// out is the uint8 label image [n][H][W]; the logits are never written.
// (conv3x3_dma_kernel, the one-block-of-eight-waves form these notes were written for, was SSM_CONV_VARIANT=1 until round 6: 7 % slower than the two-block form below
// on the network -- all eight waves reach the epilogue together and the MFMA pipe idles for its length --, profiles/r03_segnet_*.  The LDS images, the DMA scheme, the
// stacked virtual image and the epilogue below are its.)
// ------------------------------------------------------------------ conv3x3, LDS-DMA staged, TWO independent 4-wave blocks per CU
// Same tile (32 x 16 pixels x 64 output channels), same LDS images, same DMA scheme and the same epilogue as
// conv3x3_dma_kernel, but
//   * a block is 4 waves and a wave owns FOUR rows x 64 channels (8 accumulators): 6 fragment reads feed 8 MFMAs (was 4 : 4);
//   * a stage is 16 input channels (one MFMA K step per tap, 9 steps of 8 MFMAs), 38 KB, so two stage buffers are 76 KB and
//     TWO blocks share a CU.  Their barriers are independent: while one block is in its epilogue (BN, pack, stores) or waits
//     at a barrier, the other block's waves keep the matrix cores busy.  With one 8-wave block all waves reach the epilogue
//     together and the MFMA pipe idles for its whole length, which costs most on the 64- and 128-channel layers (2 and 4
//     stages per tile).
//   * the second half of the grid starts half a tile late so that the two blocks of a CU run out of phase.
#ifdef SSM_CONV_ABLATE
__device__ unsigned long long g_conv_cycles;     // scripts/ubench/conv_bench.hip: longest block lifetime in shader clocks
__device__ unsigned long long g_conv_phase[8][4]; // [probe block][mfma phase, epilogue, tiles, lifetime] of wave 0
#endif
#define D2_AINS 20                            // input wave-instructions per stage (5 per wave; 1280 slots for 2 planes = 1224 chunks)
#define D2_ACH (2 * DT_PLANE)
#define D2_BCH (9 * 2 * CT_N)                 // 1152 chunks of weights per stage = 18 wave-instructions
#define D2_STAGE (D2_AINS * 64 + D2_BCH)      // 2432 chunks = 38,912 B per LDS buffer
// UP: the input is a max-pooled tensor + its arg-max codes and the convolution runs on the un-pooled (2x) image without that
// image ever being written: the DMA of a halo pixel fetches the pooled pixel (y/2, x/2), the thread that issued it also loads
// the 8 code bytes of that chunk, and once the stage has landed it zeroes, in LDS, the channels whose code is not this
// pixel's position in its 2x2 window (each thread masks exactly the chunks it fetched, before the stage's barrier).
template <bool RELU, int EPI, int NT, bool UP = false>
__global__ void __launch_bounds__(256, 2)
conv3x3_dma2_kernel(const _Float16* __restrict__ in, const _Float16* __restrict__ wt, const float* __restrict__ scale, const float* __restrict__ shift,
                    _Float16* __restrict__ out, uint8_t* __restrict__ code, int n, int H, int W, int Cin, int Cout, int tiles_x, int ncout_tiles, int total_tiles,
                    unsigned in_bytes, unsigned wt_bytes, unsigned out_bytes, int* __restrict__ queue, const uint8_t* __restrict__ ucode, int xcd_map)
{
    __shared__ __attribute__((aligned(16))) half8 lds0[D2_STAGE];
    __shared__ __attribute__((aligned(16))) half8 lds1[D2_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int VH = (H + 2) & ~1, VR = n * VH;
    const unsigned vh_magic = (0xFFFFFFFFu / (unsigned)VH) + 1u;
    const int nchunks = Cin / CT_KC;
    const int UPH = (H + 1) >> 1, UPW = (W + 1) >> 1;                             // UP: the pooled input's size
    const unsigned plane_bytes = UP ? (unsigned)UPH * UPW * 64u : (unsigned)H * W * 64u;   // one 32-channel chunk of one input frame
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)wt, 0, wt_bytes, 0x00020000);
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, out_bytes, 0x00020000);
    const auto rsC = __builtin_amdgcn_make_buffer_rsrc((void*)code, 0, EPI == 1 ? out_bytes / 2 : 0, 0x00020000);
    const auto rsU = __builtin_amdgcn_make_buffer_rsrc((void*)ucode, 0, UP ? in_bytes / 2 : 0, 0x00020000);   // codes: one byte per pooled element
    constexpr int BROW = 32 * NT;
    constexpr int BINS = 9 * 2 * BROW / 64;                                        // weight wave-instructions per stage: 18 or 9
    constexpr int BK = (BINS + 3) / 4;                                             // per wave: 5 or 3
    constexpr int NSTORE = EPI == 2 ? 4 : 8 * NT;                                  // store instructions per tile epilogue
    constexpr int WAIT_TILE = 0x0F70 | (NSTORE & 15) | ((NSTORE >> 4) << 14);      // s_waitcnt vmcnt(NSTORE) (vmcnt is split: bits 3:0 and 15:14)
    constexpr int CTW = 32 * NT;                                                   // output channels per tile: ncout_tiles counts tiles of this width
    __shared__ __attribute__((aligned(16))) float s_ss[2][CT_N];
    // dynamic tile order: a block starts with tile blockIdx.x and keeps its cout tile; the following pixel tiles come from a
    // per-cout-tile counter (queue[2 ct]), fetched one tile ahead because the pipeline prefetches across tile boundaries.
    // The two blocks of a CU do not run at the same speed (the SIMD issues the older wave first), so a static split leaves the
    // faster block idle at the end.  queue[2 ct + 1] counts the blocks that have drained the counter; the last one zeroes
    // both, so the buffer is ready for the next launch on the stream.
    __shared__ int s_next;
    // Which tiles a block takes.  Blocks are dealt round-robin over the 8 XCDs (blockIdx.x % 8 labels the blocks that share an XCD and its L2: MI355X_MICROARCH.md).
    // xcd_map (round 6): the blocks of one XCD group cover ALL cout tiles of the SAME pixel tiles (pixel tile p belongs to group p % 8), so an input tile is fetched
    // into one L2 and served to its cout tiles from there; the weights (0.6 - 4.7 MB per layer) are what every XCD streams, from MALL.  The first mapping
    // (my_ct = blockIdx.x % ncout_tiles: one cout tile per XCD group) kept a weight slab per L2 and made every XCD read the whole input: measured, the stage's reads
    // were 133 MB of inputs once + 311 MB of re-reads per frame; with the XCD order the re-reads are 107 MB (profiles/r06_segnet_mfma_util.md).  The stage's time moved
    // by 1 % only: the kernel is bound by the L2 -> LDS staging rate, not by HBM.
    const int xg = blockIdx.x & 7, xslot = blockIdx.x >> 3;
    const int my_ct = xcd_map ? xslot % ncout_tiles : blockIdx.x % ncout_tiles;
    const int blocks_per_ct = xcd_map ? (gridDim.x >> 3) / ncout_tiles : gridDim.x / ncout_tiles;      // blocks that share this block's counter
    int* const qp = queue + 2 * (xcd_map ? xg * ncout_tiles + my_ct : my_ct);                           // this block's counter pair
    // tile index of this counter's i-th pixel tile: t_base + i t_stride (two scalars: the kernel is short of scalar registers)
    const int t_base = xcd_map ? xg * ncout_tiles + my_ct : my_ct, t_stride = xcd_map ? 8 * ncout_tiles : ncout_tiles;
#define D2_TILE_OF(i_) (t_base + (i_) * t_stride)
    if (tid < 2 * CT_N) {
        const int cl = tid & (CT_N - 1), ch = my_ct * CTW + cl;
        s_ss[tid >> 6][cl] = cl < CTW && ch < Cout ? (tid < CT_N ? scale[ch] : shift[ch]) : 0.f;
    }
    unsigned a_off[5], b_off[5]; int b_j[5];
    int a_py[5], a_px[5]; unsigned a_c8[5];
    unsigned u_pos[5]; uint2v u_code[5];                                           // UP: 2x2 position of the slot's pixel (x 0x01010101), its chunk's codes
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int i = (wv + 4 * k) * 64 + lane;
        const int c8 = i / DT_PLANE, p = i - c8 * DT_PLANE;
        a_py[k] = i < D2_ACH ? p / DT_PW : -0x10000;
        a_px[k] = p - (p / DT_PW) * DT_PW; a_c8[k] = c8 * 16u;
        // tile origins are even in x and y (and so is a frame's first row in the stacked image): the parity of a halo pixel is the slot's
        u_pos[k] = (unsigned)((((p / DT_PW) - 1) & 1) * 2 + ((a_px[k] - 1) & 1)) * 0x01010101u; u_code[k] = uint2v{0u, 0u};
        b_j[k] = min(wv + 4 * k, BINS - 1);
        // packed weights: [tap][c8 of 4][cout 64][8]; a stage takes c8 = 2 half + {0, 1} (the half is in the scalar offset).
        // LDS rows are (tap, c) x BROW couts; with NT = 1 a wave-instruction fills two rows with the first 32 couts of each
        const int row = NT == 2 ? b_j[k] : 2 * b_j[k] + (lane >> 5);
        b_off[k] = (unsigned)((((row >> 1) * 4 + (row & 1)) * 64) + (NT == 2 ? lane : (lane & 31))) * 16u;
    }
#define D2_TILE_OFFSETS(tile)                                                                           \
    {   const int pt_ = (tile) / ncout_tiles;                                                           \
        const int tx_ = (pt_ % tiles_x) * DT_W, ty_ = (pt_ / tiles_x) * DT_H;                           \
        _Pragma("unroll") for (int k = 0; k < 5; k++) {                                                 \
            const int v = ty_ + a_py[k] - 1, gx = tx_ + a_px[k] - 1;                                    \
            const int f = (int)__umulhi((unsigned)v, vh_magic), y = v - f * VH;                         \
            const bool ok = (tile) < total_tiles && v >= 0 && v < VR && gx >= 0 && gx < W && y < H;      \
            a_off[k] = !ok ? 0x80000000u : UP ? ((unsigned)f * nchunks) * plane_bytes + ((unsigned)(y >> 1) * UPW + (gx >> 1)) * 64u + a_c8[k] \
                                              : ((unsigned)f * nchunks) * plane_bytes + ((unsigned)y * W + gx) * 64u + a_c8[k]; \
        } }
#ifdef CT_ABL_NODMA
#define D2_DMA_A(k, dst, so) asm volatile("" :: "v"(a_off[k]), "s"(so));
#define D2_DMA_B(k, dst, so) asm volatile("" :: "v"(b_off[k]), "s"(so));
#else
#define D2_DMA_A(k, dst, so) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)&dst[(wv + 4 * (k)) * 64], 16, a_off[k], so, 0, 0); \
                               if (UP) u_code[k] = __builtin_amdgcn_raw_buffer_load_b64(rsU, a_off[k] >> 1, (so) >> 1, 0); }   /* out of range -> 0 */
#define D2_DMA_B(k, dst, so) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)&dst[D2_AINS * 64 + b_j[k] * 64], 16, b_off[k], so, 0, 0);
#endif
    // UP: zero the channels of this thread's five chunks of `buf` whose code differs from the pixel's position; codes are 0..3, so
    // byte equality is ~(t | t >> 1) & 1 on t = code ^ position; v_perm spreads the flags to 16-bit lanes, v_pk_mul_lo_u16 applies them
#define D2_UNPOOL_MASK(buf)                                                                             \
    if (UP) {                                                                                           \
        _Pragma("unroll") for (int k = 0; k < 5; k++) {                                                 \
            const int idx_ = (wv + 4 * k) * 64 + lane;                                                  \
            uint4v d_ = *reinterpret_cast<const uint4v*>(&buf[idx_]);                                   \
            const unsigned t0_ = u_code[k].x ^ u_pos[k], t1_ = u_code[k].y ^ u_pos[k];                  \
            const unsigned e0_ = ~(t0_ | (t0_ >> 1)) & 0x01010101u, e1_ = ~(t1_ | (t1_ >> 1)) & 0x01010101u; \
            typedef unsigned short us2_ __attribute__((ext_vector_type(2)));                            \
            const unsigned m_[4] = {__builtin_amdgcn_perm(0u, e0_, 0x0C010C00u), __builtin_amdgcn_perm(0u, e0_, 0x0C030C02u), \
                                    __builtin_amdgcn_perm(0u, e1_, 0x0C010C00u), __builtin_amdgcn_perm(0u, e1_, 0x0C030C02u)}; \
            unsigned o_[4] = {d_.x, d_.y, d_.z, d_.w};                                                  \
            _Pragma("unroll") for (int q = 0; q < 4; q++) { us2_ a_, b_; memcpy(&a_, &o_[q], 4); memcpy(&b_, &m_[q], 4); a_ = a_ * b_; memcpy(&o_[q], &a_, 4); } \
            d_.x = o_[0]; d_.y = o_[1]; d_.z = o_[2]; d_.w = o_[3];                                     \
            *reinterpret_cast<uint4v*>(&buf[idx_]) = d_;                                                \
        }                                                                                               \
        __builtin_amdgcn_s_waitcnt(0xC07F);   /* lgkmcnt(0): the masked chunks are in LDS before the barrier */ \
    }
#ifdef CT_ABL_NOBAR
#define D2_BARRIER()
#else
#define D2_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#ifdef SSM_CONV_ABLATE
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    unsigned long long ph_mfma = 0, ph_tiles = 0;
#endif
    // byte offset of cout tile ct's weights for chunk 0: packed per 64-cout tile [chunk][tap][c8][cout 64][8]; a 32-cout tile is
    // the first or second half of each 64-cout row
#define D2_SLAB0(ct_) (NT == 2 ? (unsigned)(ct_) * nchunks * (DT_BCH * 16u) : (unsigned)((ct_) >> 1) * nchunks * (DT_BCH * 16u) + (unsigned)((ct_) & 1) * 512u)
    int tile = xcd_map ? D2_TILE_OF(xslot / ncout_tiles) : (int)blockIdx.x;
    // phase shift between the two blocks of a CU (blocks i and i + gridDim/2 are dispatched to the same CU when the grid is
    // 2 x CUs): about half of a tile's MFMA time
    if (blockIdx.x >= (gridDim.x >> 1)) {
        for (int i = 0; i < nchunks; i++) __builtin_amdgcn_s_sleep(36);       // 36 x 64 clk = one stage (72 MFMAs x 32 clk) per chunk
    }
    D2_TILE_OFFSETS(tile)
    {
        const unsigned bso = D2_SLAB0(tile % ncout_tiles);
#pragma unroll
        for (int k = 0; k < 5; k++) { D2_DMA_A(k, lds0, 0u) if (k < BK) D2_DMA_B(k, lds0, bso) }
        // UP: the first tile's codes are waited for here, once: left alone the compiler sinks these loads below the stores that
        // follow, and the wait it then needs at the top of the tile loop (one code path for the first and all later tiles)
        // would drain the epilogue stores of every tile
        if (UP) { _Pragma("unroll") for (int k = 0; k < 5; k++) asm volatile("" : "+v"(u_code[k].x), "+v"(u_code[k].y) :: "memory"); }
        const uint4v z4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < NSTORE; k++) __builtin_amdgcn_raw_buffer_store_b128(z4, rsO, 0x80000000u + 16u * (tid + 256 * k), 0, 0);
    }
    int next_tile = total_tiles;
    const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (; tile < total_tiles; tile = next_tile) {
        const int ct = tile % ncout_tiles, pt = tile / ncout_tiles;
        const int tx0 = (pt % tiles_x) * DT_W, ty0 = (pt / tiles_x) * DT_H;
        floatx16 acc[4][NT];                                                       // first written by the tile's first MFMAs (C = 0)
        // one stage: 9 steps (taps) of 4 NT MFMAs on buffer `rd`, column offset (dx) major: the six halo rows a wave needs at one
        // dx are read ONCE and serve the three taps (dy) of that column -- 6 pixel + 3 x NT weight fragment reads per 12 NT MFMAs
        // (a tap-by-tap walk reads 12 + 3 NT).  The next stage's 10 DMA instructions (into `wr`), the next step's weight
        // fragments and a third of the next column's pixel fragments are issued ahead of each step's MFMAs.
#ifdef CT_ABL_NOREAD
#define D2_READ_ALL 0
#else
#define D2_READ_ALL 1
#endif
#define D2_LOADW(rd, fbuf, st)      /* weight fragments of step st: tap = 3 dy + dx with dx = st / 3, dy = st % 3 */ \
        {   const int tap_ = 3 * ((st) % 3) + (st) / 3;                                                       \
            const half8* pb_ = rd + D2_AINS * 64 + (tap_ * 2 + hh) * BROW + r;                                \
            if (D2_READ_ALL || (st) < 2) { fb[fbuf][0] = pb_[0]; if (NT == 2) fb[fbuf][1] = pb_[32]; } }
#define D2_LOADX(rd, gbuf, dx_, hr_) /* pixel fragment: halo row hr_ of this wave's six, column offset dx_ */          \
        {   if (D2_READ_ALL || (dx_) == 0) fa[gbuf][hr_] = (rd + hh * DT_PLANE + (4 * wv + (hr_)) * DT_PW + r + (dx_))[0]; }
#define D2_STAGE_BODY(rd, wr, Z)                                                                        \
        {   half8 fa[2][6], fb[2][NT];                                                                  \
            _Pragma("unroll") for (int hr = 0; hr < 6; hr++) D2_LOADX(rd, 0, 0, hr)                     \
            D2_LOADW(rd, 0, 0)                                                                          \
            _Pragma("unroll") for (int st = 0; st < 9; st++) {                                          \
                const int cur = st & 1, dxs = st / 3, dys = st - 3 * dxs, grp = dxs & 1;                \
                if (st < 5) D2_DMA_A(st, wr, a_so)                                                      \
                if (st >= 5 && st - 5 < BK) D2_DMA_B(st - 5, wr, b_so)                                  \
                if (st == 0 && BK == 5) D2_DMA_B(4, wr, b_so)                                           \
                if (st + 1 < 9) D2_LOADW(rd, cur ^ 1, st + 1)                                           \
                if (dxs < 2) { D2_LOADX(rd, grp ^ 1, dxs + 1, 2 * dys) D2_LOADX(rd, grp ^ 1, dxs + 1, 2 * dys + 1) } \
                __builtin_amdgcn_sched_barrier(0);                                                      \
                _Pragma("unroll") for (int tm_ = 0; tm_ < 4; tm_++) _Pragma("unroll") for (int tn_ = 0; tn_ < NT; tn_++) \
                    acc[tm_][tn_] = CT_MFMA(fb[cur][tn_], fa[grp][tm_ + dys], (Z) && st == 0 ? zero16 : acc[tm_][tn_]); \
                __builtin_amdgcn_sched_barrier(0);                                                      \
            } }
        // the two stages of one 32-channel chunk; the first chunk of a tile is written out separately because its wait differs
        // (one merged path makes the compiler's own vmcnt bookkeeping pessimistic: it then drains the epilogue stores)
#define D2_CHUNK(ck, WAITC, FIRST)                                                                      \
        {   __builtin_amdgcn_s_waitcnt(WAITC);                                                          \
            D2_UNPOOL_MASK(lds0)                                                                        \
            D2_BARRIER();                                                               \
            /* the counter fetch is older than this stage's DMA and is covered by the stage's closing vmcnt(0); as inline    \
               assembly, because the compiler would wait for a returning atomic at the end of the branch (draining the   \
               previous tile's stores) */                                                                  \
            int fetched_;                                                                               \
            if (FIRST && tid == 0) asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=v"(fetched_) : "v"(0), "v"(1), "s"(qp) : "memory"); \
            const unsigned slab = D2_SLAB0(ct) + (unsigned)(ck) * (DT_BCH * 16u);                       \
            unsigned a_so = (unsigned)(ck) * plane_bytes + 32u, b_so = slab + 2048u;   /* channels 16..31 of this chunk */ \
            D2_STAGE_BODY(lds0, lds1, FIRST)                                                                \
            __builtin_amdgcn_s_waitcnt(0x0F70);                                                         \
            if (FIRST && tid == 0) s_next = D2_TILE_OF(blocks_per_ct + fetched_);                       \
            D2_UNPOOL_MASK(lds1)                                                                        \
            D2_BARRIER();                                                               \
            if ((ck) + 1 < nchunks) {                                                                   \
                a_so = (unsigned)((ck) + 1) * plane_bytes;                                              \
                b_so = slab + DT_BCH * 16u;                                                             \
            } else {                                                                                    \
                const int nt = __builtin_amdgcn_readfirstlane(s_next);                                  \
                next_tile = nt;                                                                         \
                D2_TILE_OFFSETS(nt)                                                                     \
                a_so = 0u;                                                                              \
                b_so = nt < total_tiles ? D2_SLAB0(nt % ncout_tiles) : 0x80000000u;                     \
            }                                                                                           \
            D2_STAGE_BODY(lds1, lds0, false) }
#ifdef SSM_CONV_ABLATE
        const unsigned long long t_a = __builtin_amdgcn_s_memtime();
#endif
        // the tile's first DMA batch is older than the NSTORE stores of the previous epilogue, which may stay in flight
        D2_CHUNK(0, WAIT_TILE, true)
        for (int ck = 1; ck < nchunks; ck++) D2_CHUNK(ck, 0x0F70, false)
#undef D2_CHUNK
#ifdef SSM_CONV_ABLATE
        const unsigned long long t_b = __builtin_amdgcn_s_memtime();
        ph_mfma += t_b - t_a; ph_tiles++;
#endif
#undef D2_STAGE_BODY
#undef D2_LOADW
#undef D2_LOADX
        const int cout_chunks = (Cout + 31) >> 5;
        const int gx = tx0 + r;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int v0 = ty0 + 4 * wv + 2 * half;                               // even row of the stacked image; VH is even, so y0 is even too
            const int f = (int)__umulhi((unsigned)v0, vh_magic), y0 = v0 - f * VH;
            const bool live0 = v0 < VR && y0 < H && gx < W, live1 = live0 && y0 + 1 < H;
            if constexpr (EPI == 2) {
                // ArgMax over the classes (channels < Cout <= 12) of the fp16-rounded logits, first maximum wins (argmax_kernel's contract).
                // acc[..][0][4g+q] is class 8g + 4hh + q: the lower half-wave owns classes 0-3 and 8-11, the upper 4-7.
#pragma unroll
                for (int tm = 0; tm < 2; tm++) {
                    _Float16 bv[2]; int bi[2];
#pragma unroll
                    for (int g = 0; g < 2; g++) {
                        const int c0 = 8 * g + 4 * hh;
                        const float4 sc = *reinterpret_cast<const float4*>(&s_ss[0][c0]), sf = *reinterpret_cast<const float4*>(&s_ss[1][c0]);
                        const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, sfv[4] = {sf.x, sf.y, sf.z, sf.w};
                        bv[g] = (_Float16)-65504.f; bi[g] = 255;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            float val = __builtin_fmaf(acc[2 * half + tm][0][4 * g + q], scv[q], sfv[q]);
                            if (RELU) val = fmaxf(val, 0.f);
                            const _Float16 hvq = (_Float16)val;
                            if (c0 + q < Cout && (bi[g] == 255 || hvq > bv[g])) { bv[g] = hvq; bi[g] = c0 + q; }
                        }
                    }
                    unsigned short vb; memcpy(&vb, &bv[0], 2);
                    const unsigned mine = ((unsigned)vb << 8) | (unsigned)bi[0];
                    const unsigned theirs = __builtin_amdgcn_permlane32_swap(mine, mine, false, false)[1];
                    unsigned short tb = (unsigned short)(theirs >> 8); _Float16 tv; memcpy(&tv, &tb, 2);
                    const int ti = (int)(theirs & 255u);
                    _Float16 best = bv[0]; int lab = bi[0];
                    if (ti != 255 && tv > best) { best = tv; lab = ti; }
                    if (bi[1] != 255 && bv[1] > best) { best = bv[1]; lab = bi[1]; }
                    const bool live = (tm ? live1 : live0) && hh == 0;
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)lab, rsO, live ? (unsigned)((f * H + y0 + tm) * W + gx) : 0x80000000u, 0, 0);
                }
            } else {
                floatx16 a2[2][2];
#pragma unroll
                for (int tm = 0; tm < 2; tm++)
#pragma unroll
                    for (int tn = 0; tn < 2; tn++) a2[tm][tn] = acc[2 * half + tm][tn < NT ? tn : 0];
                conv_epilogue<RELU, EPI, NT>(a2, s_ss, rsO, rsC, f, y0, gx, live0, live1, ct * NT, cout_chunks, H, W, r, hh);
            }
        }
    }
    // exactly one fetch per block came back past the end: the last block of this cout tile to get there resets the counters
    if (tid == 0 && atomicAdd(&qp[1], 1) == blocks_per_ct - 1) { qp[0] = 0; qp[1] = 0; }
#undef D2_TILE_OF
#ifdef SSM_CONV_ABLATE
    if (tid == 0) {
        const unsigned long long life = __builtin_amdgcn_s_memtime() - t_start;
        atomicMax(&g_conv_cycles, life);
        const int probe = blockIdx.x == 0 ? 0 : blockIdx.x == 1 ? 1 : blockIdx.x == 100 ? 2 : blockIdx.x == 255 ? 3 : blockIdx.x == 256 ? 4 : blockIdx.x == 257 ? 5 : blockIdx.x == 400 ? 6 : blockIdx.x == 511 ? 7 : -1;
        if (probe >= 0) { g_conv_phase[probe][0] = ph_mfma; g_conv_phase[probe][1] = life - ph_mfma; g_conv_phase[probe][2] = ph_tiles; g_conv_phase[probe][3] = life; }
    }
#endif
#undef D2_DMA_A
#undef D2_DMA_B
#undef D2_BARRIER
#undef D2_UNPOOL_MASK
#undef D2_TILE_OFFSETS
#undef D2_SLAB0
}

// ------------------------------------------------------------------ conv3x3 by Winograd F(2, 3) along x (round 6)
// y[2j], y[2j+1] of a row from d0 .. d3 = x[2j-1 .. 2j+2] of the three input rows dy:  V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3;
// U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2 (per dy, packed on the host in fp32, stored fp16);  M_k = sum over (dy, cin) U_k V_k;
// y[2j] = (M0 + M1) + M2, y[2j+1] = (M1 - M2) - M3.  Four MFMAs per output PAIR and dy instead of six: 1.5 x fewer than the direct form.
// Same tile (32 x 16 pixels), the same stacked virtual image, LDS-DMA staging, two 4-wave blocks per CU, tile queue and BN / ReLU / 16-byte-store epilogue as
// conv3x3_dma2_kernel with 32-cout tiles (NT = 1), but
//   * an MFMA's 32 pixel columns are 16 pairs x 2 rows: lane r = (pair j = r & 15, row select rs = r >> 4) works on rows 4 wv + rp + 2 rs, rp = 0, 1, and
//     acc[rp][k] are the four transform positions (8 accumulators, as the direct NT = 2 form has): a lane's 2 x 2 output block is (rows rp = 0, 1) x (its pair);
//   * the halo image in LDS keeps the EVEN pixels of a row in front of the odd ones (the DMA's source offset is per lane, so this costs nothing): lane j reads
//     d0 .. d3 at chunks j, 17 + j, j + 1, 18 + j of its row -- consecutive lanes, consecutive chunks -- and builds V0 .. V3 with 16 packed fp16 subtractions
//     and additions in registers (nothing goes back to LDS); a lane needs the FOUR halo rows t = rp + dy = 0 .. 3 of its row select;
//   * weights: 12 "taps" (dy, k) per stage instead of 9; the twelve fragments of a stage are read once and serve both rp.
// Per wave and stage: 24 MFMAs, 28 ds_read_b128 (the direct NT = 1 form: 36 and 18).  Integer-valued data stays exact (the constants are 1, -1, 1/2).
#define WG_TAPS 12
#define WG_BCH_FULL (WG_TAPS * 4 * CT_N)      // chunks of transformed weights per (64-cout tile, 32-cin chunk): [tap][c8 4][cout 64][8]
#define WG_BINS 12                            // weight wave-instructions per stage (12 taps x 2 c8 x 32 couts = 768 chunks)
// the halo image of a stage: 2 planes (8 channels each) of 18 rows x 40 chunks: the 17 even pixels of a row, the 17 odd ones, 6 chunks of padding -- a row stride of
// 40 chunks makes the chunks of lanes two rows apart (the row select of a lane, below) fall on the same bank groups as their own pair index: ds_read_b128 without
// bank conflicts (the first version read rows ONE apart at a stride of 34: 38 % of its LDS cycles were conflicts)
#define WG_PW 40
#define WG_ODD 17
#define WG_PLANE (DT_PH * WG_PW)              // 720
#define WG_ACH (2 * WG_PLANE)                 // 1440 chunks of input per stage
#define WG_AINS 23                            // input wave-instructions per stage (6 per wave; 1472 slots)
#define WG_AK 6
#define WG_STAGE (WG_AINS * 64 + WG_BINS * 64) // 2240 chunks = 35.8 KB per LDS buffer
template <bool RELU>
__global__ void __launch_bounds__(256, 2)
conv3x3_wino_kernel(const _Float16* __restrict__ in, const _Float16* __restrict__ wt, const float* __restrict__ scale, const float* __restrict__ shift,
                    _Float16* __restrict__ out, int n, int H, int W, int Cin, int Cout, int tiles_x, int ncout_tiles, int total_tiles,
                    unsigned in_bytes, unsigned wt_bytes, unsigned out_bytes, int* __restrict__ queue)
{
    __shared__ __attribute__((aligned(16))) half8 lds0[WG_STAGE];
    __shared__ __attribute__((aligned(16))) half8 lds1[WG_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // an MFMA's 32 pixel columns: 16 pairs x 2 rows.  Lane r = (pair pj, row select rs): rows 4 wv + rp + 2 rs for the wave's two accumulator sets rp
    const int r = lane & 31, hh = lane >> 5, pj = r & 15, rs = r >> 4;
    // (the geometry the per-lane address arithmetic uses lives in VECTOR registers on purpose: the kernel runs out of scalar registers otherwise, and a spilled
    // scalar next to the inline-assembly counter fetch below is what the first version of this kernel crashed on)
    int VH = (H + 2) & ~1, VR = n * VH, Hv = H, Wv = W;
    unsigned vh_magic = (0xFFFFFFFFu / (unsigned)VH) + 1u;
    asm volatile("" : "+v"(VH), "+v"(VR), "+v"(Hv), "+v"(Wv), "+v"(vh_magic));
    const int nchunks = Cin / CT_KC;
    const unsigned plane_bytes = (unsigned)H * W * 64u;                            // one 32-channel chunk of one input frame
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)wt, 0, wt_bytes, 0x00020000);
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, out_bytes, 0x00020000);
    constexpr int BK = 3;                                                          // weight wave-instructions per wave and stage
    constexpr int NSTORE = 8;
    constexpr int WAIT_TILE = 0x0F70 | (NSTORE & 15) | ((NSTORE >> 4) << 14);      // s_waitcnt vmcnt(NSTORE)
    __shared__ __attribute__((aligned(16))) float s_ss[2][CT_N];
    __shared__ int s_next;
    const int my_ct = blockIdx.x % ncout_tiles, blocks_per_ct = gridDim.x / ncout_tiles;
    if (tid < 2 * CT_N) {
        const int cl = tid & (CT_N - 1), ch = my_ct * 32 + cl;
        s_ss[tid >> 6][cl] = cl < 32 && ch < Cout ? (tid < CT_N ? scale[ch] : shift[ch]) : 0.f;
    }
    unsigned a_off[WG_AK], b_off[BK];
#pragma unroll
    for (int k = 0; k < BK; k++) {
        const int row = 2 * (wv + 4 * k) + (lane >> 5);                            // rows (tap, c) of the stage's weight image: wave-instruction wv + 4 k fills two of them
        b_off[k] = (unsigned)((((row >> 1) * 4 + (row & 1)) * 64) + (lane & 31)) * 16u;
    }
    // the DMA slots of this thread: slot i = (wv + 4 k) 64 + lane -> (plane c8, halo row py, chunk q of the row); q < 17: even pixel 2 q, q < 34: odd pixel 2 (q - 17) + 1,
    // else padding (an out-of-range offset: the DMA writes zeros).  One packed register per slot: py | px << 8 | c8 << 16, or -1
    int a_slot[WG_AK];
#pragma unroll
    for (int k = 0; k < WG_AK; k++) {
        const int i_ = (wv + 4 * k) * 64 + lane;
        const int c8_ = i_ >= WG_PLANE ? 1 : 0, p_ = i_ - c8_ * WG_PLANE;
        const int py_ = p_ / WG_PW, q_ = p_ - py_ * WG_PW;
        const int px_ = q_ < WG_ODD ? 2 * q_ : 2 * (q_ - WG_ODD) + 1;
        a_slot[k] = (i_ < WG_ACH && q_ < 2 * WG_ODD) ? (py_ | (px_ << 8) | (c8_ << 16)) : -1;
    }
#define WG_TILE_OFFSETS(tile)                                                                           \
    {   const int pt_ = (tile) / ncout_tiles;                                                           \
        const int tx_ = (pt_ % tiles_x) * DT_W, ty_ = (pt_ / tiles_x) * DT_H;                           \
        _Pragma("unroll") for (int k = 0; k < WG_AK; k++) {                                             \
            const int sl_ = a_slot[k];                                                                  \
            const int v = ty_ + (sl_ & 255) - 1, gx = tx_ + ((sl_ >> 8) & 255) - 1;                     \
            const int f = (int)__umulhi((unsigned)v, vh_magic), y = v - f * VH;                         \
            const bool ok = (tile) < total_tiles && sl_ >= 0 && v >= 0 && v < VR && gx >= 0 && gx < Wv && y < Hv; \
            a_off[k] = !ok ? 0x80000000u : ((unsigned)f * nchunks) * plane_bytes + ((unsigned)y * Wv + gx) * 64u + (unsigned)((sl_ >> 16) & 1) * 16u; \
        } }
#define WG_DMA_A(k, dst, so) { if (wv + 4 * (k) < WG_AINS) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)&dst[(wv + 4 * (k)) * 64], 16, a_off[k], so, 0, 0); }
#define WG_DMA_B(k, dst, so) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)&dst[WG_AINS * 64 + (wv + 4 * (k)) * 64], 16, b_off[k], so, 0, 0);
    // byte offset of 32-cout tile ct's transformed weights for chunk 0: the first or second half of each 64-cout row of its 64-cout slab
#define WG_SLAB0(ct_) ((unsigned)((ct_) >> 1) * nchunks * (WG_BCH_FULL * 16u) + (unsigned)((ct_) & 1) * 512u)
    int tile = blockIdx.x;
    if (blockIdx.x >= (gridDim.x >> 1)) {                                          // the two blocks of a CU half a tile out of phase (see conv3x3_dma2_kernel)
        for (int i = 0; i < nchunks; i++) __builtin_amdgcn_s_sleep(12);            // 12 x 64 clk = one stage (24 MFMAs x 32 clk) per chunk
    }
    WG_TILE_OFFSETS(tile)
    {
        const unsigned bso = WG_SLAB0(tile % ncout_tiles);
#pragma unroll
        for (int k = 0; k < WG_AK; k++) { WG_DMA_A(k, lds0, 0u) if (k < BK) WG_DMA_B(k, lds0, bso) }
        const uint4v z4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < NSTORE; k++) __builtin_amdgcn_raw_buffer_store_b128(z4, rsO, 0x80000000u + 16u * (tid + 256 * k), 0, 0);
    }
    int next_tile = total_tiles;
    const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // this lane's pixel chunks of halo row (4 wv + 2 rs + t), t = rp + dy = 0 .. 3: plane hh, even pixel j / odd pixel j / even j + 1 / odd j + 1
    const int x_base = hh * WG_PLANE + (4 * wv + 2 * rs) * WG_PW + pj;
    const int w_base = WG_AINS * 64 + hh * 32 + r;
    for (; tile < total_tiles; tile = next_tile) {
        const int ct = tile % ncout_tiles, pt = tile / ncout_tiles;
        const int tx0 = (pt % tiles_x) * DT_W, ty0 = (pt / tiles_x) * DT_H;
        floatx16 acc[2][4];                                                        // [rp][transform position]
#define WG_LOADX(rd, buf, t_)                                                                           \
        {   const half8* px_ = rd + x_base + (t_) * WG_PW;                                              \
            xr[buf][0] = px_[0]; xr[buf][1] = px_[WG_ODD]; xr[buf][2] = px_[1]; xr[buf][3] = px_[WG_ODD + 1]; }
#define WG_LOADW(rd, dst, dy_)                                                                          \
        {   _Pragma("unroll") for (int k_ = 0; k_ < 4; k_++) dst[k_] = (rd + w_base + ((dy_) * 4 + k_) * 64)[0]; }
#define WG_XFORM(cur)                                                                                   \
            half8 v_[4];                                                                                \
            v_[0] = xr[cur][0] - xr[cur][2]; v_[1] = xr[cur][1] + xr[cur][2];                           \
            v_[2] = xr[cur][2] - xr[cur][1]; v_[3] = xr[cur][1] - xr[cur][3];
#define WG_MFMA4(rp_, w_, ZERO)                                                                         \
            { _Pragma("unroll") for (int k_ = 0; k_ < 4; k_++) acc[rp_][k_] = CT_MFMA(w_[k_], v_[k_], (ZERO) ? zero16 : acc[rp_][k_]); }
        // one stage: halo row t serves (rp, dy) = (0, t) for t <= 2 and (1, t - 1) for t >= 1.  Two sets of weight fragments are live at a time (wa: dy 0, then dy 2 once row
        // 1's rp = 1 products have read dy 0; wb: dy 1); the pixel reads of row t + 1 are issued ahead of row t's MFMAs; the next stage's 9 DMA instructions (into `wr`)
        // are spread over the rows.
#define WG_STAGE_BODY(rd, wr, Z)                                                                        \
        {   half8 wa[4], wb[4], xr[2][4];                                                               \
            WG_LOADX(rd, 0, 0)                                                                          \
            WG_LOADW(rd, wa, 0)                                                                         \
            {   WG_DMA_A(0, wr, a_so) WG_DMA_A(1, wr, a_so) WG_DMA_A(2, wr, a_so)                       \
                WG_DMA_A(3, wr, a_so) WG_DMA_A(4, wr, a_so) WG_DMA_A(5, wr, a_so)                       \
                WG_DMA_B(0, wr, b_so) WG_DMA_B(1, wr, b_so) WG_DMA_B(2, wr, b_so)                       \
                WG_LOADX(rd, 1, 1)                                                                      \
                WG_LOADW(rd, wb, 1)                                                                     \
                WG_XFORM(0)                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                      \
                WG_MFMA4(0, wa, Z)                                                                      \
                __builtin_amdgcn_sched_barrier(0); }                                                    \
            {   WG_LOADX(rd, 0, 2)                                                                      \
                WG_XFORM(1)                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                      \
                WG_MFMA4(1, wa, Z)                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                      \
                WG_LOADW(rd, wa, 2)                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                      \
                WG_MFMA4(0, wb, false)                                                                  \
                __builtin_amdgcn_sched_barrier(0); }                                                    \
            {   WG_LOADX(rd, 1, 3)                                                                      \
                WG_XFORM(0)                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                      \
                WG_MFMA4(1, wb, false)                                                                  \
                WG_MFMA4(0, wa, false)                                                                  \
                __builtin_amdgcn_sched_barrier(0); }                                                    \
            {   WG_XFORM(1)                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                      \
                WG_MFMA4(1, wa, false)                                                                  \
                __builtin_amdgcn_sched_barrier(0); } }
#define WG_CHUNK(ck, WAITC, FIRST)                                                                      \
        {   __builtin_amdgcn_s_waitcnt(WAITC);                                                          \
            __builtin_amdgcn_s_barrier();                                                               \
            int fetched_;                                                                               \
            if (FIRST && tid == 0) fetched_ = __hip_atomic_fetch_add(queue + 2 * my_ct, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            const unsigned slab = WG_SLAB0(ct) + (unsigned)(ck) * (WG_BCH_FULL * 16u);                  \
            unsigned a_so = (unsigned)(ck) * plane_bytes + 32u, b_so = slab + 2048u;   /* channels 16..31 of this chunk */ \
            WG_STAGE_BODY(lds0, lds1, FIRST)                                                            \
            __builtin_amdgcn_s_waitcnt(0x0F70);                                                         \
            if (FIRST && tid == 0) s_next = (blocks_per_ct + fetched_) * ncout_tiles + my_ct;           \
            __builtin_amdgcn_s_barrier();                                                               \
            if ((ck) + 1 < nchunks) {                                                                   \
                a_so = (unsigned)((ck) + 1) * plane_bytes;                                              \
                b_so = slab + WG_BCH_FULL * 16u;                                                        \
            } else {                                                                                    \
                const int nt = __builtin_amdgcn_readfirstlane(s_next);                                  \
                next_tile = nt;                                                                         \
                WG_TILE_OFFSETS(nt)                                                                     \
                a_so = 0u;                                                                              \
                b_so = nt < total_tiles ? WG_SLAB0(nt % ncout_tiles) : 0x80000000u;                     \
            }                                                                                           \
            WG_STAGE_BODY(lds1, lds0, false) }
        WG_CHUNK(0, WAIT_TILE, true)
        for (int ck = 1; ck < nchunks; ck++) WG_CHUNK(ck, 0x0F70, false)
#undef WG_CHUNK
#undef WG_STAGE_BODY
#undef WG_LOADX
#undef WG_LOADW
#undef WG_XFORM
#undef WG_MFMA4
        // ---- epilogue: inverse transform (fp32), BN, ReLU, fp16; the lane holds the pixel pair (2 j, 2 j + 1) of row 4 wv + rp + 2 rs, channels 8 g + 4 hh + q
        const int cout_chunks = (Cout + 31) >> 5;
#pragma unroll
        for (int rp = 0; rp < 2; rp++) {
            const int v0 = ty0 + 4 * wv + rp + 2 * rs;
            const int f = (int)__umulhi((unsigned)v0, vh_magic), y0 = v0 - f * VH;
            const int gx = tx0 + 2 * pj;
            const bool live = v0 < VR && y0 < Hv;
#pragma unroll
            for (int px = 0; px < 2; px++) {
#pragma unroll
                for (int gp = 0; gp < 2; gp++) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const int g = 2 * gp + e, cl = 8 * g + 4 * hh;
                        const float4 sc = *reinterpret_cast<const float4*>(&s_ss[0][cl]), sf = *reinterpret_cast<const float4*>(&s_ss[1][cl]);
                        const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, sfv[4] = {sf.x, sf.y, sf.z, sf.w};
                        typedef float float2v __attribute__((ext_vector_type(2)));
                        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int q = 0; q < 4; q += 2) {
                            float yv[2];
#pragma unroll
                            for (int u = 0; u < 2; u++) {
                                const int i = 4 * g + q + u;
                                yv[u] = px == 0 ? (acc[rp][0][i] + acc[rp][1][i]) + acc[rp][2][i] : (acc[rp][1][i] - acc[rp][2][i]) - acc[rp][3][i];
                            }
                            const float2v val = {__builtin_fmaf(yv[0], scv[q], sfv[q]), __builtin_fmaf(yv[1], scv[q + 1], sfv[q + 1])};
                            half2v h2 = __builtin_convertvector(val, half2v);
                            if (RELU) h2 = __builtin_elementwise_max(h2, (half2v){(_Float16)0, (_Float16)0});
                            memcpy(&pk[e][q >> 1], &h2, 4);
                        }
                    }
                    // lanes 0-31 keep quad 2gp (channels +0..3) and receive the upper half's quad 2gp (+4..7); lanes 32-63 receive the lower half's quad 2gp+1 (+8..11) and keep their own
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                    uint4v vec; vec.x = s0[0]; vec.y = s1[0]; vec.z = s0[1]; vec.w = s1[1];
                    const bool lv = live && gx + px < Wv && ct < cout_chunks;
                    const unsigned ob = lv ? (unsigned)((((f * cout_chunks + ct) * Hv + y0) * Wv + gx + px) * 64 + 16 * hh + 32 * gp) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(vec, rsO, ob, 0, 0);
                }
            }
        }
    }
    if (tid == 0 && atomicAdd(&queue[2 * my_ct + 1], 1) == blocks_per_ct - 1) { queue[2 * my_ct] = 0; queue[2 * my_ct + 1] = 0; }
#undef WG_DMA_A
#undef WG_DMA_B
#undef WG_TILE_OFFSETS
#undef WG_SLAB0
}

// (Until round 6 a third form lived here, conv3x3_k32_kernel on v_mfma_f32_16x16x32_f16, SSM_CONV_VARIANT=3.  On dense random operands the chip is power-limited and
// holds a markedly higher clock on that MFMA shape (+18..21 % TFLOP/s in scripts/ubench/conv_bench.hip's CT_ABL_MFMA16 build) and the kernel was 5 % faster there; inside
// the network -- post-ReLU activations, half of them zero -- the chip is less power-limited, the shape buys nothing and the one-block-per-CU structure K = 32 forces cost
// 2 %: DESIGN.md s.4.1, profiles/r04_segnet_*.)
// ------------------------------------------------------------------ conv3x3 of an input with <= 8 channels (the network's first layer)
// The input is one 8-channel (16-byte) vector per pixel, [n][H][W][8].  A 16-deep MFMA K step then covers TWO taps: the
// lower half-wave (k 0..7) reads tap 2s and the upper (k 8..15) tap 2s+1 of the same LDS plane, i.e. the two halves just use
// different pixel offsets for their fragment read.  9 taps = 5 steps (the 10th half has zero weights) instead of the 18 the
// 32-channel chunk kernel would spend on a chunk that is 29/32 padding.  Weights: [cout tile][step 5][half 2][cout 64][8],
// staged once per block; the input tile (18 x 34 pixels, 9.6 KB) is double-buffered by LDS-DMA across the block's tiles as
// in conv3x3_dma_kernel, and the epilogue is shared.  The layer is bound by its 64-channel output write.
#define FT_ACH (DT_PH * DT_PW)                // 612 chunks = one plane
#define FT_ASLOTS 640                         // 10 wave-instructions
#define FT_BCH (5 * 2 * CT_N)                 // 640 chunks of weights
template <bool RELU>
__global__ void __launch_bounds__(512, 1)
conv3x3_first_kernel(const _Float16* __restrict__ in, const _Float16* __restrict__ wt, const float* __restrict__ scale, const float* __restrict__ shift,
                     _Float16* __restrict__ out, int n, int H, int W, int Cout, int tiles_x, int ncout_tiles, int total_tiles,
                     unsigned in_bytes, unsigned wt_bytes, unsigned out_bytes)
{
    __shared__ __attribute__((aligned(16))) half8 lda0[FT_ASLOTS];
    __shared__ __attribute__((aligned(16))) half8 lda1[FT_ASLOTS];
    __shared__ __attribute__((aligned(16))) half8 ldb[FT_BCH];
    __shared__ __attribute__((aligned(16))) float s_ss[2][CT_N];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int VH = (H + 2) & ~1, VR = n * VH;
    const unsigned vh_magic = (0xFFFFFFFFu / (unsigned)VH) + 1u;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)wt, 0, wt_bytes, 0x00020000);
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, out_bytes, 0x00020000);
    const int ct = blockIdx.x % ncout_tiles;                                       // constant for the block (grid is a multiple of ncout_tiles)
    if (tid < 2 * CT_N) {
        const int ch = ct * CT_N + (tid & (CT_N - 1));
        s_ss[tid >> 6][tid & (CT_N - 1)] = ch < Cout ? (tid < CT_N ? scale[ch] : shift[ch]) : 0.f;
    }
    // the block's weights, once: 10 wave-instructions, waves 0 and 1 take two
    for (int j = wv; j < FT_BCH / 64; j += 8)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)&ldb[j * 64], 16, (unsigned)(ct * FT_BCH + j * 64 + lane) * 16u, 0, 0, 0);
    // input DMA slots of this thread: chunk i = (wv + 8k) * 64 + lane of the 18 x 34 halo tile, k = 0, 1 (k = 1: waves 0, 1 only)
    int a_py[2], a_px[2]; unsigned a_off[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = (wv + 8 * k) * 64 + lane;
        a_py[k] = i < FT_ACH ? i / DT_PW : -0x10000;
        a_px[k] = i - (i / DT_PW) * DT_PW;
    }
#define FT_TILE_OFFSETS(tile)                                                                           \
    {   const int pt_ = (tile) / ncout_tiles;                                                           \
        const int tx_ = (pt_ % tiles_x) * DT_W, ty_ = (pt_ / tiles_x) * DT_H;                           \
        _Pragma("unroll") for (int k = 0; k < 2; k++) {                                                 \
            const int v = ty_ + a_py[k] - 1, gx = tx_ + a_px[k] - 1;                                    \
            const int f = (int)__umulhi((unsigned)v, vh_magic), y = v - f * VH;                         \
            const bool ok = (tile) < total_tiles && v >= 0 && v < VR && gx >= 0 && gx < W && y < H;      \
            a_off[k] = ok ? (((unsigned)f * H + y) * W + gx) * 16u : 0x80000000u;                       \
        } }
#define FT_DMA_A(dst)                                                                                   \
    {   __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)&dst[wv * 64], 16, a_off[0], 0, 0, 0); \
        if (wv < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)&dst[(wv + 8) * 64], 16, a_off[1], 0, 0, 0); }
    int tile = blockIdx.x;
    FT_TILE_OFFSETS(tile)
    FT_DMA_A(lda0)
    {
        const uint4v z4 = {0u, 0u, 0u, 0u};                                       // see conv3x3_dma_kernel: 8 dropped stores make vmcnt(8) path-invariant
#pragma unroll
        for (int k = 0; k < 8; k++) __builtin_amdgcn_raw_buffer_store_b128(z4, rsO, 0x80000000u + 16u * (tid + 512 * k), 0, 0);
    }
    // per-lane fragment offsets of the 5 steps: tap = min(2 s + hh, 8) (the 10th half multiplies zero weights)
    int fo[5];
#pragma unroll
    for (int st = 0; st < 5; st++) { const int tap = min(2 * st + hh, 8), dy = tap / 3, dx = tap - dy * 3; fo[st] = (2 * wv + dy) * DT_PW + r + dx; }
    const int cout_chunks = (Cout + 31) >> 5;
#define FT_TILE_BODY(rd, wr)                                                                            \
    {   const int pt = tile / ncout_tiles;                                                              \
        const int tx0 = (pt % tiles_x) * DT_W, ty0 = (pt / tiles_x) * DT_H;                             \
        __builtin_amdgcn_s_waitcnt(0x0F78);      /* vmcnt(8): this tile's input has landed; the last epilogue's stores may fly on */ \
        __builtin_amdgcn_s_barrier();                                                                   \
        { const int nt = tile + gridDim.x; FT_TILE_OFFSETS(nt) }                                        \
        FT_DMA_A(wr)                                                                                    \
        floatx16 acc[2][2];                                                                             \
        _Pragma("unroll") for (int a = 0; a < 2; a++) _Pragma("unroll") for (int b = 0; b < 2; b++) _Pragma("unroll") for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f; \
        _Pragma("unroll") for (int st = 0; st < 5; st++) {                                              \
            const half8 fa0 = rd[fo[st]], fa1 = rd[fo[st] + DT_PW];                                     \
            const half8 fb0 = ldb[(st * 2 + hh) * CT_N + r], fb1 = ldb[(st * 2 + hh) * CT_N + r + 32];  \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb0, fa0, acc[0][0], 0, 0, 0);           \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb1, fa0, acc[0][1], 0, 0, 0);           \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb0, fa1, acc[1][0], 0, 0, 0);           \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb1, fa1, acc[1][1], 0, 0, 0);           \
        }                                                                                               \
        const int gx = tx0 + r, v0 = ty0 + 2 * wv;                                                      \
        const int f = (int)__umulhi((unsigned)v0, vh_magic), y0 = v0 - f * VH;                          \
        const bool live0 = v0 < VR && y0 < H && gx < W, live1 = live0 && y0 + 1 < H;                    \
        conv_epilogue<RELU, 0>(acc, s_ss, rsO, rsO, f, y0, gx, live0, live1, ct * (CT_N / 32), cout_chunks, H, W, r, hh); \
    }
    for (; tile < total_tiles; tile += gridDim.x) {
        FT_TILE_BODY(lda0, lda1)
        tile += gridDim.x;
        if (tile >= total_tiles) break;
        FT_TILE_BODY(lda1, lda0)
    }
#undef FT_TILE_BODY
#undef FT_DMA_A
#undef FT_TILE_OFFSETS
}

// ------------------------------------------------------------------ max-pool 2x2 stride 2, CEIL mode, with arg-max code
// Caffe (SegNet fork) PoolingLayer MAX with top_mask: window scanned row-major, strict '>' => first maximum wins.
// code = (dy*2 + dx) inside the window.  8 channels (16 B) per thread.
__global__ void __launch_bounds__(256)
pool2x2_kernel(const _Float16* __restrict__ in, int H, int W, int C, _Float16* __restrict__ out, uint8_t* __restrict__ code, int PH, int PW)
{
    const int c8 = C >> 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PH * PW * c8) return;
    const int pp = i / c8, cg = i - pp * c8;
    const int ph = pp / PW, pw = pp - ph * PW;
    const _Float16* src = in + (size_t)blockIdx.y * H * W * C;          // [C/32][H][W][32]
    half8 best; uint8_t bc[8];
    bool first = true;
#pragma unroll
    for (int dy = 0; dy < 2; dy++)
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
            const int y = ph * 2 + dy, x = pw * 2 + dx;
            if (y >= H || x >= W) continue;
            const half8 v = *reinterpret_cast<const half8*>(src + (((size_t)(cg >> 2) * H + y) * W + x) * 32 + (cg & 3) * 8);
            if (first) { best = v; for (int k = 0; k < 8; k++) bc[k] = (uint8_t)(dy * 2 + dx); first = false; }
            else {
#pragma unroll
                for (int k = 0; k < 8; k++) if (v[k] > best[k]) { best[k] = v[k]; bc[k] = (uint8_t)(dy * 2 + dx); }
            }
        }
    const size_t o = (size_t)blockIdx.y * PH * PW * C + ((size_t)(cg >> 2) * PH * PW + pp) * 32 + (cg & 3) * 8;
    *reinterpret_cast<half8*>(out + o) = best;
    uint2 pk; pk.x = bc[0] | (bc[1] << 8) | (bc[2] << 16) | (bc[3] << 24); pk.y = bc[4] | (bc[5] << 8) | (bc[6] << 16) | (bc[7] << 24);
    *reinterpret_cast<uint2*>(code + o) = pk;
}
// SegNet Upsample layer: out[argmax position] = pooled value, everything else 0; explicit output size H x W
__global__ void __launch_bounds__(256)
unpool2x2_kernel(const _Float16* __restrict__ in, const uint8_t* __restrict__ code, int PH, int PW, int C, _Float16* __restrict__ out, int H, int W)
{
    const int c8 = C >> 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W * c8) return;
    const int p = i / c8, cg = i - p * c8;
    const int y = p / W, x = p - y * W;
    const int ph = y >> 1, pw = x >> 1, my = (y & 1) * 2 + (x & 1);
    const size_t s = (size_t)blockIdx.y * PH * PW * C + ((size_t)(cg >> 2) * PH * PW + (size_t)ph * PW + pw) * 32 + (cg & 3) * 8;
    const half8 v = *reinterpret_cast<const half8*>(in + s);
    const uint2 pk = *reinterpret_cast<const uint2*>(code + s);
    half8 o;
#pragma unroll
    for (int k = 0; k < 8; k++) { const int cd = ((k < 4 ? pk.x : pk.y) >> (8 * (k & 3))) & 255; o[k] = cd == my ? v[k] : (_Float16)0.f; }
    *reinterpret_cast<half8*>(out + (size_t)blockIdx.y * H * W * C + ((size_t)(cg >> 2) * H * W + p) * 32 + (cg & 3) * 8) = o;
}
// ArgMax over the class logits (Softmax is monotone): first maximum wins, like caffe ArgMaxLayer's partial_sort on (value, index)
__global__ void __launch_bounds__(256)
argmax_kernel(const _Float16* __restrict__ logits, int npix, int Cstore, int ncls, uint8_t* __restrict__ labels)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const _Float16* l = logits + ((size_t)blockIdx.y * npix + p) * Cstore;
    int best = 0; float bv = (float)l[0];
    for (int c = 1; c < ncls; c++) { const float v = (float)l[c]; if (v > bv) { bv = v; best = c; } }
    labels[(size_t)blockIdx.y * npix + p] = (uint8_t)best;
}
// experiment/segnet.cpp:80-83,131-146: ids -> (Pavement 5 -> Road 4) -> 3-channel id image -> cv::resize to the frame size
// (bilinear ON THE IDS, as the reference does; nearest when `nearest` != 0) -> cv::LUT(color.png) -> BGR class colours
__constant__ uint8_t c_seg_palette[12][3] = {
    {128,128,128}, {0,0,128}, {128,192,192}, {0,69,255}, {128,64,128}, {222,40,60},
    {0,128,128}, {128,128,192}, {128,64,64}, {128,0,64}, {0,64,64}, {192,128,0}
};
__global__ void __launch_bounds__(256)
label_color_kernel(const uint8_t* __restrict__ ids, int sw, int sh, int dw, int dh,
                   const int32_t* __restrict__ xofs, const int16_t* __restrict__ xa, const int32_t* __restrict__ yofs, const int16_t* __restrict__ ya,
                   int pavement_to_road, int nearest, uint8_t* __restrict__ sem_bgr, uint8_t* __restrict__ ids_out)
{
    // four consecutive pixels of a row per thread: the 12 colour bytes and the 4 ids leave as three + one aligned dword stores (one byte store per channel and
    // pixel ran at 0.85 TB/s); rows whose width is not a multiple of four keep the one-pixel form for their last pixels
    // the palette as 16 packed words in LDS (a lane-indexed read of __constant__ memory is a vector-memory access per byte: with the per-pixel id taps the
    // kernel spent 0.69 of its time in the texture-address units)
    __shared__ uint32_t pal[16];
    if (threadIdx.x < 16) pal[threadIdx.x] = threadIdx.x < 12 ? ((uint32_t)c_seg_palette[threadIdx.x][0] | ((uint32_t)c_seg_palette[threadIdx.x][1] << 8) | ((uint32_t)c_seg_palette[threadIdx.x][2] << 16)) : 0u;
    __syncthreads();
    const int qpr = (dw + 3) >> 2;                          // quads per row
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= qpr * dh) return;
    const int y = q / qpr, x0 = (q - y * qpr) << 2;
    const uint8_t* src = ids + (size_t)blockIdx.y * sw * sh;
    auto id_at = [&](int yy, int xx) { int v = src[(size_t)yy * sw + xx]; return (pavement_to_road && v == 5) ? 4 : v; };
    int v[4] = {0, 0, 0, 0};
    const int sy0 = nearest ? min((int)((y * (long long)sh) / dh), sh - 1) : yofs[y], sy1 = min(sy0 + 1, sh - 1);
    const int b0 = nearest ? 0 : ya[2*y], b1 = nearest ? 0 : ya[2*y+1];
    // the quad's taps lie in 8 consecutive source ids of two rows when its columns do not reach the clamped right border: two 8-byte loads + the quad's x
    // tables as two 16-byte loads instead of 16 + 12 scattered ones
    bool quad = false;
    if (!nearest && x0 + 4 <= dw && (dw & 3) == 0) {
        const int4 XO = *reinterpret_cast<const int4*>(xofs + x0); const uint4 XA = *reinterpret_cast<const uint4*>(xa + 2 * x0);
        const int xo[4] = {XO.x, XO.y, XO.z, XO.w}; const uint32_t xw[4] = {XA.x, XA.y, XA.z, XA.w};
        if (xo[3] + 1 <= sw - 1 && xo[3] - xo[0] <= 6 && (size_t)sy1 * sw + xo[0] + 8 <= (size_t)sw * sh) {
            quad = true;
            unsigned long long r0, r1;
            __builtin_memcpy(&r0, src + (size_t)sy0 * sw + xo[0], 8); __builtin_memcpy(&r1, src + (size_t)sy1 * sw + xo[0], 8);
            auto fix = [&](int t) { return (pavement_to_road && t == 5) ? 4 : t; };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int sh8 = 8 * (xo[k] - xo[0]);
                const int a0 = (int)(int16_t)(xw[k] & 0xFFFFu), a1 = (int)(int16_t)(xw[k] >> 16);
                const int h0 = fix((int)((r0 >> sh8) & 255u)) * a0 + fix((int)((r0 >> (sh8 + 8)) & 255u)) * a1;
                const int h1 = fix((int)((r1 >> sh8) & 255u)) * a0 + fix((int)((r1 >> (sh8 + 8)) & 255u)) * a1;
                v[k] = ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 255;
            }
        }
    }
    if (!quad) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = x0 + k;
        if (x >= dw) break;
        if (nearest) { v[k] = id_at(sy0, min((int)((x * (long long)sw) / dw), sw - 1)); continue; }
        const int sx0 = xofs[x], sx1 = min(sx0 + 1, sw - 1), a0 = xa[2*x], a1 = xa[2*x+1];
        const int h0 = id_at(sy0, sx0) * a0 + id_at(sy0, sx1) * a1;
        const int h1 = id_at(sy1, sx0) * a0 + id_at(sy1, sx1) * a1;
        v[k] = ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 255;
    }
    }
    const size_t o = (size_t)blockIdx.y * dw * dh + (size_t)y * dw + x0;
    uint8_t c[12];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t pc = pal[min(v[k], 15)];                // (ids 12 .. 255: black, like the entries 12 .. 15)
        c[3*k] = (uint8_t)(pc & 255u); c[3*k+1] = (uint8_t)((pc >> 8) & 255u); c[3*k+2] = (uint8_t)(pc >> 16);
    }
    const bool whole = x0 + 4 <= dw && ((dw & 3) == 0);     // aligned dword stores need every row to start on a multiple of four pixels
    if (whole) {
        if (ids_out) *reinterpret_cast<uint32_t*>(ids_out + o) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
        if (sem_bgr) {
            uint32_t w3[3]; __builtin_memcpy(w3, c, 12);
            uint32_t* d = reinterpret_cast<uint32_t*>(sem_bgr + 3 * o);
            d[0] = w3[0]; d[1] = w3[1]; d[2] = w3[2];
        }
    } else {
        for (int k = 0; k < 4 && x0 + k < dw; k++) {
            if (ids_out) ids_out[o + k] = (uint8_t)v[k];
            if (sem_bgr) { sem_bgr[3*(o+k)] = c[3*k]; sem_bgr[3*(o+k)+1] = c[3*k+1]; sem_bgr[3*(o+k)+2] = c[3*k+2]; }
        }
    }
}

// ------------------------------------------------------------------ launchers
hipError_t k_segnet_prep(const uint8_t* bgr, int n, int sw, int sh, int dw, int dh, const int32_t* xofs, const int16_t* xa,
                         const int32_t* yofs, const int16_t* ya, void* out_f16, hipStream_t s)
{
    segnet_prep_kernel<<<dim3((dw * dh + 255) / 256, n), 256, 0, s>>>(bgr, sw, sh, dw, dh, xofs, xa, yofs, ya, (_Float16*)out_f16);
    return hipGetLastError();
}
// tile counters of conv3x3_dma2_kernel (self-resetting, see there): one small zeroed buffer per (device, stream); launches on a
// stream are ordered, so they can share it
static std::mutex g_tq_mu; static std::map<std::pair<int, hipStream_t>, int*> g_tq_bufs;
void k_segnet_release_stream(hipStream_t s)                       // ssm_destroy: the stream's tile-counter buffer goes with the context
{
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_tq_mu);
    auto it = g_tq_bufs.find({dev, s});
    if (it != g_tq_bufs.end()) { (void)hipFree(it->second); g_tq_bufs.erase(it); }
}
#define CONV_QUEUE_INTS 1024          // a counter pair per (XCD group, cout tile): 8 x 32 x 2 at most
static int* conv_tile_queue(hipStream_t s)
{
    std::mutex& mu = g_tq_mu; auto& bufs = g_tq_bufs;
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto it = bufs.find({dev, s});
    if (it != bufs.end()) return it->second;
    int* p = nullptr;
    if (hipMalloc(&p, CONV_QUEUE_INTS * sizeof(int)) != hipSuccess || hipMemset(p, 0, CONV_QUEUE_INTS * sizeof(int)) != hipSuccess) return nullptr;
    bufs[{dev, s}] = p;
    return p;
}
// start of a forward pass: zero the stream's tile counters (they reset themselves at the end of every launch; this only
// keeps an aborted launch from poisoning the passes after it)
hipError_t k_segnet_begin(hipStream_t s)
{
    int* q = conv_tile_queue(s);
    return q ? hipMemsetAsync(q, 0, CONV_QUEUE_INTS * sizeof(int), s) : hipErrorOutOfMemory;
}
static int conv_grid_limit()
{
    static int cus = 0;
    if (!cus) { int dev = 0; hipGetDevice(&dev); if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256; }
    return cus;
}
// epi 0: conv; 1: conv + max-pool (out pooled, code); 2: conv + class ArgMax (out = uint8 labels [n][H][W]; Cout <= 12)
// ucode != nullptr: `in` is the max-pooled tensor of an H x W image and ucode its arg-max codes (un-pool on load)
static hipError_t conv_dma_launch(const void* in, const void* wt, const float* scale, const float* shift, void* out, uint8_t* code, int n, int H, int W,
                                  int CinPad, int Cout, int relu, int epi, hipStream_t s, const uint8_t* ucode = nullptr)
{
    const int nct = (Cout + CT_N - 1) / CT_N, VH = (H + 2) & ~1, cs = (Cout + 31) & ~31;
    const unsigned long long in_bytes = ucode ? (unsigned long long)n * ((H + 1) / 2) * ((W + 1) / 2) * CinPad * 2 : (unsigned long long)n * H * W * CinPad * 2, wt_bytes = (unsigned long long)nct * CT_N * CinPad * 9 * 2;
    const unsigned long long out_bytes = epi == 1 ? (unsigned long long)n * ((H + 1) / 2) * ((W + 1) / 2) * cs * 2 : epi == 2 ? (unsigned long long)n * H * W : (unsigned long long)n * H * W * cs * 2;
    if (epi == 2 && Cout > 12) return hipErrorInvalidValue;
    // 32-bit buffer offsets and a 16-bit virtual row index; callers batch below these
    if (in_bytes >= 0x80000000ull || wt_bytes >= 0x80000000ull || out_bytes >= 0x80000000ull || (long long)n * VH >= 65536) return hipErrorInvalidValue;
    const int tx = (W + DT_W - 1) / DT_W, ty = (n * VH + DT_H - 1) / DT_H, total = tx * ty * nct;
    if (ucode && (epi != 0 || !relu)) return hipErrorInvalidValue;
#ifdef SSM_CONV_ABLATE   /* scripts/ubench/conv_bench.hip only: zero-sized buffer descriptors drop the stores (1) / turn the DMA into zero fills (2) */
    static const int abl = [] { const char* e = getenv("SSM_CONV_ABL"); return e ? atoi(e) : 0; }();
    const unsigned long long in_bytes_k = (abl & 2) ? 0 : in_bytes, wt_bytes_k = (abl & 2) ? 0 : wt_bytes, out_bytes_k = (abl & 1) ? 0 : out_bytes;
#define in_bytes in_bytes_k
#define wt_bytes wt_bytes_k
#define out_bytes out_bytes_k
#endif
    {
        // two persistent 4-wave blocks per CU (conv3x3_dma2_kernel); the grid is a multiple of the cout-tile count so a block keeps
        // its weight slab.  Tiles are 64 output channels wide, or 32 when that balances the CUs better: blocks b and b + grid/2
        // share a CU and its matrix cores, so a CU's time is the work of both; a 32-wide tile costs a bit more than half
        // (the input tile is staged once per 32 channels instead of once per 64).
        const int cus = conv_grid_limit();
        auto makespan = [&](int units, int tiles_n, double cost) {
            int grid = 2 * cus; grid -= grid % tiles_n; if (grid > units) grid = units;
            const int half = cus;                                                  // blocks b and b + cus share a CU
            long worst = 0;
            for (int b = 0; b < half && b < grid; b++) {
                long c = (units - b + grid - 1) / grid;
                if (b + half < grid) c += (units - (b + half) + grid - 1) / grid;
                worst = c > worst ? c : worst;
            }
            return worst * cost;
        };
        int nt_w = 2, nct_k = nct;
        if (epi == 2) nt_w = 1;
        else if (Cout % 64 == 0 && makespan(2 * total, 2 * nct, 0.56) < makespan(total, nct, 1.0)) { nt_w = 1; nct_k = 2 * nct; }
        const int total_k = tx * ty * nct_k;
        int grid = 2 * cus; grid -= grid % nct_k; if (grid > total_k) grid = total_k;
#define D2_LAUNCH(R, E, N) conv3x3_dma2_kernel<R, E, N><<<grid, 256, 0, s>>>((const _Float16*)in, (const _Float16*)wt, scale, shift, (_Float16*)out, code, n, H, W, CinPad, Cout, tx, nct_k, total_k, (unsigned)in_bytes, (unsigned)wt_bytes, (unsigned)out_bytes, queue, ucode, xcd_map)
#define D2_LAUNCH_UP(N) conv3x3_dma2_kernel<true, 0, N, true><<<grid, 256, 0, s>>>((const _Float16*)in, (const _Float16*)wt, scale, shift, (_Float16*)out, code, n, H, W, CinPad, Cout, tx, nct_k, total_k, (unsigned)in_bytes, (unsigned)wt_bytes, (unsigned)out_bytes, queue, ucode, xcd_map)
        int* queue = conv_tile_queue(s);
        if (!queue || nct_k > 32) return hipErrorOutOfMemory;
        // the XCD-aware tile order needs whole groups of 8 x nct_k blocks (a full grid has them; a launch smaller than the grid keeps the plain order)
        // (A/B on the stage, same box: 5.04-5.12k frames/s plain, 5.14-5.16k with the XCD order; measured bytes 607 -> 386 MB per frame)
        const int xcd_map = grid % (8 * nct_k) == 0 ? 1 : 0;
        if (ucode) { if (nt_w == 1) D2_LAUNCH_UP(1); else D2_LAUNCH_UP(2); }
        else if (epi == 2) { if (relu) D2_LAUNCH(true, 2, 1); else D2_LAUNCH(false, 2, 1); }
        else if (epi == 1 && nt_w == 1) { if (relu) D2_LAUNCH(true, 1, 1); else D2_LAUNCH(false, 1, 1); }
        else if (epi == 1) { if (relu) D2_LAUNCH(true, 1, 2); else D2_LAUNCH(false, 1, 2); }
        else if (nt_w == 1) { if (relu) D2_LAUNCH(true, 0, 1); else D2_LAUNCH(false, 0, 1); }
        else { if (relu) D2_LAUNCH(true, 0, 2); else D2_LAUNCH(false, 0, 2); }
#undef D2_LAUNCH
#undef D2_LAUNCH_UP
#ifdef SSM_CONV_ABLATE
#undef in_bytes
#undef wt_bytes
#undef out_bytes
#endif
        return hipGetLastError();
    }
}
static hipError_t conv_first_launch(const void* in, const void* wt, const float* scale, const float* shift, void* out, int n, int H, int W, int Cout, int relu, hipStream_t s)
{
    const int nct = (Cout + CT_N - 1) / CT_N, VH = (H + 2) & ~1, cs = (Cout + 31) & ~31;
    const unsigned long long in_bytes = (unsigned long long)n * H * W * 16, wt_bytes = (unsigned long long)nct * FT_BCH * 16, out_bytes = (unsigned long long)n * H * W * cs * 2;
    if (in_bytes >= 0x80000000ull || out_bytes >= 0x80000000ull || (long long)n * VH >= 65536) return hipErrorInvalidValue;
    const int tx = (W + DT_W - 1) / DT_W, ty = (n * VH + DT_H - 1) / DT_H, total = tx * ty * nct;
    int grid = 2 * conv_grid_limit(); grid -= grid % nct; if (grid > total) grid = total;        // 31 KB of LDS: two blocks per CU
    if (relu) conv3x3_first_kernel<true><<<grid, 512, 0, s>>>((const _Float16*)in, (const _Float16*)wt, scale, shift, (_Float16*)out, n, H, W, Cout, tx, nct, total, (unsigned)in_bytes, (unsigned)wt_bytes, (unsigned)out_bytes);
    else      conv3x3_first_kernel<false><<<grid, 512, 0, s>>>((const _Float16*)in, (const _Float16*)wt, scale, shift, (_Float16*)out, n, H, W, Cout, tx, nct, total, (unsigned)in_bytes, (unsigned)wt_bytes, (unsigned)out_bytes);
    return hipGetLastError();
}
// CinPad == 8: the <= 8-channel first layer ([n][H][W][8] input, its own weight packing); otherwise CinPad is a multiple of 64
static hipError_t conv_wino_launch(const void* in, const void* wt, const float* scale, const float* shift, void* out, int n, int H, int W, int CinPad, int Cout, int relu, hipStream_t s)
{
    const int nct64 = (Cout + CT_N - 1) / CT_N, VH = (H + 2) & ~1;
    const unsigned long long in_bytes = (unsigned long long)n * H * W * CinPad * 2, wt_bytes = (unsigned long long)nct64 * CT_N * CinPad * WG_TAPS * 2, out_bytes = (unsigned long long)n * H * W * Cout * 2;
    if (Cout % 32 || CinPad % (2 * CT_KC)) return hipErrorInvalidValue;
    if (in_bytes >= 0x80000000ull || wt_bytes >= 0x80000000ull || out_bytes >= 0x80000000ull || (long long)n * VH >= 65536) return hipErrorInvalidValue;
    const int tx = (W + DT_W - 1) / DT_W, ty = (n * VH + DT_H - 1) / DT_H, nct = Cout / 32, total = tx * ty * nct;
    const int cus = conv_grid_limit();
    int grid = 2 * cus; grid -= grid % nct; if (grid > total) grid = total;
    int* queue = conv_tile_queue(s);
    if (!queue || nct > 32) return hipErrorOutOfMemory;
    if (relu) conv3x3_wino_kernel<true><<<grid, 256, 0, s>>>((const _Float16*)in, (const _Float16*)wt, scale, shift, (_Float16*)out, n, H, W, CinPad, Cout, tx, nct, total, (unsigned)in_bytes, (unsigned)wt_bytes, (unsigned)out_bytes, queue);
    else conv3x3_wino_kernel<false><<<grid, 256, 0, s>>>((const _Float16*)in, (const _Float16*)wt, scale, shift, (_Float16*)out, n, H, W, CinPad, Cout, tx, nct, total, (unsigned)in_bytes, (unsigned)wt_bytes, (unsigned)out_bytes, queue);
    return hipGetLastError();
}
hipError_t k_segnet_conv(const void* in, const void* wt, const float* scale, const float* shift, void* out, int n, int H, int W,
                         int CinPad, int Cout, int relu, hipStream_t s, const void* wt_wino)
{
    if (CinPad == 8) return conv_first_launch(in, wt, scale, shift, out, n, H, W, Cout, relu, s);
    if (CinPad % (2 * CT_KC)) return hipErrorInvalidValue;
    if (wt_wino) return conv_wino_launch(in, wt_wino, scale, shift, out, n, H, W, CinPad, Cout, relu, s);
    return conv_dma_launch(in, wt, scale, shift, out, nullptr, n, H, W, CinPad, Cout, relu, 0, s);
}
// conv + BN + ReLU + max-pool 2x2 (CEIL) in one pass: out is [n][Cout/32][PH][PW][32], code the arg-max codes in the same index space
hipError_t k_segnet_conv_pool(const void* in, const void* wt, const float* scale, const float* shift, void* out, uint8_t* code, int n, int H, int W,
                              int CinPad, int Cout, hipStream_t s)
{
    if ((CinPad / CT_KC) & 1) return hipErrorInvalidValue;
    return conv_dma_launch(in, wt, scale, shift, out, code, n, H, W, CinPad, Cout, 1, 1, s);
}
// last layer + ArgMax in one pass: labels [n][H][W] uint8 (the class logits are not materialised)
hipError_t k_segnet_conv_argmax(const void* in, const void* wt, const float* scale, const float* shift, uint8_t* labels, int n, int H, int W,
                                int CinPad, int Cout, hipStream_t s)
{
    if (CinPad % (2 * CT_KC)) return hipErrorInvalidValue;
    return conv_dma_launch(in, wt, scale, shift, labels, nullptr, n, H, W, CinPad, Cout, 0, 2, s);
}
// un-pool + conv + BN + ReLU in one pass: `pooled` is [n][CinPad/32][(H+1)/2][(W+1)/2][32] with its codes, out the H x W convolution
// of the un-pooled image (which is never written).  Only the default kernel has this form: k_segnet_conv_unpool_available().
int k_segnet_conv_unpool_available() { return 1; }
hipError_t k_segnet_conv_unpool(const void* pooled, const uint8_t* ucode, const void* wt, const float* scale, const float* shift, void* out, int n, int H, int W,
                                int CinPad, int Cout, hipStream_t s)
{
    if (CinPad % (2 * CT_KC) || !ucode) return hipErrorInvalidValue;
    return conv_dma_launch(pooled, wt, scale, shift, out, nullptr, n, H, W, CinPad, Cout, 1, 0, s, ucode);
}
hipError_t k_segnet_pool(const void* in, int n, int H, int W, int C, void* out, uint8_t* code, hipStream_t s)
{
    const int PH = (H + 1) / 2, PW = (W + 1) / 2;
    pool2x2_kernel<<<dim3((PH * PW * (C / 8) + 255) / 256, n), 256, 0, s>>>((const _Float16*)in, H, W, C, (_Float16*)out, code, PH, PW);
    return hipGetLastError();
}
hipError_t k_segnet_unpool(const void* in, const uint8_t* code, int n, int PH, int PW, int C, void* out, int H, int W, hipStream_t s)
{
    unpool2x2_kernel<<<dim3((H * W * (C / 8) + 255) / 256, n), 256, 0, s>>>((const _Float16*)in, code, PH, PW, C, (_Float16*)out, H, W);
    return hipGetLastError();
}
hipError_t k_segnet_argmax(const void* logits, int n, int npix, int Cstore, int ncls, uint8_t* labels, hipStream_t s)
{
    argmax_kernel<<<dim3((npix + 255) / 256, n), 256, 0, s>>>((const _Float16*)logits, npix, Cstore, ncls, labels);
    return hipGetLastError();
}
hipError_t k_segnet_color(const uint8_t* ids, int n, int sw, int sh, int dw, int dh, const int32_t* xofs, const int16_t* xa,
                          const int32_t* yofs, const int16_t* ya, int pavement_to_road, int nearest, uint8_t* sem_bgr, uint8_t* ids_out, hipStream_t s)
{
    label_color_kernel<<<dim3((((dw + 3) >> 2) * dh + 255) / 256, n), 256, 0, s>>>(ids, sw, sh, dw, dh, xofs, xa, yofs, ya, pavement_to_road, nearest, sem_bgr, ids_out);
    return hipGetLastError();
}
