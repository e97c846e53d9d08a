// ssm_internal.h -- shared between the translation units of libssm_hip.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ssm_hip.h"

#define SSM_MAX_LEVELS 12
#define SSM_EDGE 19            // ORBextractor EDGE_THRESHOLD
#define SSM_HALF_PATCH 15
#define SSM_PATCH 31
#define SSM_MAX_NODES 1024     // quad-tree nodes held in LDS per (frame, level)
#define SSM_VOX_EMPTY ((int64_t)-1)

struct LevelGeom {
    int w, h, stride;          // level image; rows padded to a multiple of 16 bytes
    int img_off;               // byte offset inside one frame's pyramid buffer (16-B aligned)
    int nCols, nRows, wCell, hCell;   // FAST cell grid (ComputeKeyPointsOctTree, W = 30)
    int cell_off;              // first flattened cell id of this level
    int tile_off, tiles_x;     // 128x32 tiles of the whole level image in the flattened grid (blur_kernel)
    int ftile_off, ftiles_x;   // FAST tiles (128x32) of this level: they cover [SSM_EDGE, w - SSM_EDGE) x [SSM_EDGE, h - SSM_EDGE) only, the positions FAST may report
    int nfeat;                 // mnFeaturesPerLevel
    int cand_off, cand_cap;    // entries inside one frame's candidate buffer
    int sel_off, sel_cap;      // slots inside one frame's selected-keypoint staging (nfeat + 3)
    int minBX, minBY, maxBX, maxBY;
    int nIni;                  // quad-tree root nodes
    float hX;                  // root node width
    float sf;                  // mvScaleFactor[level]
    uint32_t mulTX, fmulTX;    // ceil(2^32 / tiles_x), ceil(2^32 / ftiles_x): same use
    int boff;                  // byte offset of the level inside one frame's BLURRED pyramid (tiled: see blur_off)
    int bt_off, bt_x, bt_units_off;   // blur_mfma_kernel: first 128-column strip of the level, strips of the level, first 32-column unit table
    uint32_t mulW, mulH;       // ceil(2^32 / wCell), ceil(2^32 / hCell): floor(n / cell) == __umulhi(n, mul) for n < 4096 (exact: n * (mul * cell - 2^32) < 2^32)
};
// The blurred pyramid is stored in tiles of 8 rows x 16 columns (128 bytes = one cache line): its only reader, brief_kernel, gathers 37 x 37 patches,
// and a patch covers ~25 such lines instead of the ~47 it touches in a row-major image (a 37-byte row segment drags in a whole 128-byte line).
// Offset of the 16-byte word that holds (x, y): rows padded to a multiple of 8.
__host__ __device__ inline int blur_off(int boff, int stride, int x, int y) { return boff + (((y >> 3) * (stride >> 4) + (x >> 4)) << 7) + ((y & 7) << 4) + (x & 15); }
struct OrbGeom {
    int nlevels, W, H;
    int pyr_bytes;             // one frame's pyramid (all levels)
    int cells_total, cand_total, sel_total, tiles_total, ftiles_total;
    int blur_bytes;            // one frame's blurred pyramid
    int bt_total, bt_units_total;   // blur_mfma_kernel strips (= blocks) per frame, 32-column unit tables
    int cap;                   // output keypoints per frame (orb_features + 3*levels)
    int ini_th, min_th;
    int umax[SSM_HALF_PATCH + 1];
    LevelGeom L[SSM_MAX_LEVELS];
};
// candidate: lo = x | y<<12 | score<<24 (x,y relative to minBorder), hi = rank (cell-major raster order)
typedef uint2 cand_t;

// ---- launchers (each enqueues on `s`; returns hipGetLastError()) ----
hipError_t k_gray(const uint8_t* img, int channels, int n, const OrbGeom& g, uint8_t* pyr, hipStream_t s);
hipError_t k_copy_gray_strided(const uint8_t* img, int stride, const OrbGeom& g, uint8_t* pyr, hipStream_t s);
hipError_t k_pyramid(int n, const OrbGeom& g, uint8_t* pyr, const int32_t* const* xofs, const int16_t* const* xa,
                     const int32_t* const* yofs, const int16_t* const* ya, const void* const* xgroups, hipStream_t s);
hipError_t k_blur(int n, const OrbGeom& g, const uint8_t* pyr, uint8_t* blur, hipStream_t s);
// the same blur on the matrix cores; tab = blur_mfma_tables() on the device
#define BLUR_ROWS 58           // output rows of one blur_mfma block (64 input rows)
hipError_t k_blur_mfma(int n, const OrbGeom& g, const uint8_t* pyr, uint8_t* blur, const void* tab, hipStream_t s);
size_t blur_mfma_table_bytes(const OrbGeom& g);
void blur_mfma_tables(const OrbGeom& g, void* host_out);
hipError_t k_fast(int n, const OrbGeom& g, const uint8_t* pyr, cand_t* cand, int32_t* ncand, int32_t* cellmax /* k_fast_cellmax_ints(frames) ints */, hipStream_t s);
size_t k_fast_ncand_pad(int nframes, const OrbGeom& g);         // ints reserved for the per-level candidate counters in front of the cell maxima (one allocation, one fill)
size_t k_fast_cellmax_ints(int nframes, const OrbGeom& g);      // per-cell maxima of nframes frames + the retry work list of k_fast behind them
hipError_t k_octree(int n, const OrbGeom& g, const cand_t* cand, const int32_t* ncand, const int32_t* cellmax, uint16_t* node_of,
                    uint32_t* sel, int32_t* nsel, int32_t* status, hipStream_t s);
hipError_t k_describe(int n, const OrbGeom& g, const uint8_t* pyr, const uint8_t* blur, const uint32_t* sel,
                      const int32_t* nsel, const float* pattern_f /* the 256 x 4 BRIEF table as floats */, const uint16_t* depth, ssm_camera cam, void* kpaux /* n * sel_total * 32 B */,
                      ssm_keypoint* kps, uint8_t* desc, float* pos3d, int32_t* nkp, hipStream_t s);

// matcher.  pair p: query = desc + qoff[p]*32 (nq[p] rows), train = desc + toff[p]*32 (nt[p] rows)
struct MatchPair { int32_t qoff, nq, toff, nt, out_slot; };
hipError_t k_match_pairs(const uint8_t* desc, const MatchPair* pairs, int npairs, double ratio, int cap,
                         ssm_dmatch* out, int32_t* nout, int32_t* knn_idx, int32_t* knn_dist, hipStream_t s);
// as k_match_pairs but pair descriptors derived on the device from nkp[] (sequence mode)
hipError_t k_match_seq(const uint8_t* desc, const int32_t* nkp, int f0, int n, int R, int hist, double ratio, int cap,
                       ssm_dmatch* out, int32_t* nout, int32_t* pend /* n * R ints of scratch */, hipStream_t s);

// the matcher on the matrix cores: descriptors expanded to FP4 rows (capT = cap rounded up to 32 descriptors, SSM_MATCH_DESC_BYTES each, tile-fragment
// order), then v_mfma_scale_f32_32x32x64_f8f6f4 distance tiles -> knn keys -> ratio test + ordered compaction.  Same results as k_match_seq.
#define SSM_MATCH_DESC_BYTES 128
hipError_t k_match_expand(const uint8_t* desc, const int32_t* nkp, int row0, int nrows, int cap, int capT, uint8_t* eq, uint8_t* et, hipStream_t s);
hipError_t k_match_seq_mfma(const uint8_t* eq, const uint8_t* et, const int32_t* nkp, int f0, int n, int R, int hist, double ratio, int cap, int capT,
                            void* knn /* n * R * capT * 8 B */, ssm_dmatch* out, int32_t* nout, hipStream_t s);

// mapper front half
hipError_t k_moving_mask(const uint8_t* sem, int n, int w, int h, uint8_t* mask, hipStream_t s);
hipError_t k_backproject(const uint16_t* depth, const uint8_t* rgb, const uint8_t* sem, const uint8_t* mask,
                         const double* pose, int n, int w, int h, ssm_camera cam, double max_distance,
                         int32_t* chunk_cnt, int64_t* chunk_off, int32_t* npoints, int64_t* total,
                         ssm_point* out, hipStream_t s);
int backproject_chunks(int w, int h);

// voxel table
hipError_t k_voxel_clear(ssm_voxel* tab, int cap_log2, int32_t* counters, hipStream_t s);
hipError_t k_voxel_insert(const ssm_point* pts, const int64_t* n_dev, int64_t n_max, float leaf, ssm_voxel* tab,
                          int cap_log2, int32_t* counters, hipStream_t s);
// the fused map stage (kernels_map.hip map_stream2_kernel): n frames of w x h (w % 16 == 0, w <= 4096).  skip: the context's skip list (k_map_fuse_skip_cap() ints,
// counted in counters[6]); hw: a block that starts with more than hw records in the overflow list logs itself there and adds nothing; tag: the launch's slot in the
// host's ring of launch descriptors.  nredo > 0: run the blocks redo_ids[0 .. nredo) (device; entries as logged) of the launch with these arguments again.
hipError_t k_map_fuse(const uint16_t* depth, const uint8_t* rgb, const uint8_t* sem, const double* pose, int n, int w, int h,
                      ssm_camera cam, double max_distance, float leaf,
                      ssm_voxel* tab, int cap_log2, int32_t* counters, int32_t* npoints, hipStream_t s, int32_t* skip, int hw, int tag, const int32_t* redo_ids, int nredo);
int k_map_fuse_blocks_per_frame(int w, int h);
int k_map_fuse_block_records(void);        // overflow records one block can append at most
int k_map_fuse_resident_blocks(void);      // blocks of the kernel the device can hold at a time
int k_map_fuse_skip_cap(void);
hipError_t k_voxel_merge(const ssm_voxel* src, int n, ssm_voxel* tab, int cap_log2, int32_t* counters, hipStream_t s);
hipError_t k_voxel_rehash(const ssm_voxel* src, int src_cap_log2, ssm_voxel* tab, int cap_log2, int32_t* counters, hipStream_t s);
hipError_t k_voxel_compact(const ssm_voxel* tab, int cap_log2, ssm_voxel* out, int32_t* n_out, hipStream_t s);
hipError_t k_voxel_gather_points(const ssm_voxel* compact, const uint32_t* order, int n, ssm_point* out, hipStream_t s);
hipError_t k_voxel_gather_table(const ssm_voxel* compact, const uint32_t* order, int n, ssm_voxel* out, hipStream_t s);
hipError_t k_voxel_bounds(const ssm_point* pts, int n, float* minmax6, hipStream_t s);
// dst[i] = T src[i] (T: HOST pointer to a column-major 4 x 4, or nullptr = copy); pcl::transformPointCloud's arithmetic
hipError_t k_cloud_transform(const ssm_point* src, int n, const double* T, ssm_point* dst, hipStream_t s);
// sort n (key,index) pairs by key; tmp storage managed by caller via size query (tmp==nullptr)
hipError_t voxel_sort_pairs(void* tmp, size_t* tmp_bytes, const ssm_voxel* compact, int n, uint64_t* keys_a, uint64_t* keys_b,
                            uint32_t* idx_a, uint32_t* idx_b, hipStream_t s);

// synthetic stream
hipError_t k_synth(uint64_t seed, int first, int n, int w, int h, uint8_t* bgr, uint16_t* depth, uint8_t* sem,
                   uint8_t* lab, double* pose, hipStream_t s);

// SegNet (kernels_segnet.hip)
hipError_t k_segnet_prep(const uint8_t* bgr, int n, int sw, int sh, int dw, int dh, const int32_t* xofs, const int16_t* xa,
                         const int32_t* yofs, const int16_t* ya, void* out_f16, hipStream_t s);
hipError_t k_segnet_begin(hipStream_t s);
void k_segnet_release_stream(hipStream_t s);     // per-stream helper state of the SegNet / SGBM launchers, freed by ssm_destroy
void k_sgbm_release_stream(hipStream_t s);
// wt_wino != nullptr: the layer's weights transformed for the Winograd F(2, 3) kernel ([cout tile 64][cin chunk 32][tap = 4 dy + k][c8 4][cout 64][8] fp16): that kernel runs
hipError_t k_segnet_conv(const void* in, const void* wt, const float* scale, const float* shift, void* out, int n, int H, int W,
                         int CinPad, int Cout, int relu, hipStream_t s, const void* wt_wino = nullptr);
// nb frames per launch: left / right [nb][h][w], disp_out [nb][h][w]
size_t k_sgbm_workspace_bytes(int w, int h, const ssm_sgbm_params& p, int nb, int form_cfg);      // sized for the configured formulation (0 = the default), never less than one frame in the largest
bool sgbm_cost_geometry(int D, int SW, int* TX_out, size_t* lds_out);      // false: SADWindowSize too wide for the streaming cost kernel
// fail_flag: device int the sweep kernel ORs 1 into when a strip hand-off times out (never on a healthy device; the host turns it into SSM_E_HIP)
hipError_t k_sgbm(const uint8_t* left, const uint8_t* right, int w, int h, int nb, const ssm_sgbm_params& p, void* workspace, size_t ws_bytes, int16_t* disp_out, int raw_only, hipStream_t s, int* fail_flag = nullptr,
                  int form_cfg = 0, int concurrent = 1);
hipError_t k_sgbm_depth(const int16_t* disp, int w, int h, int nb, double baseline, double cu, double cv, double f, double roix, double roiy, double roiz, double scale,
                        int* min_scratch /* nb ints */, uint16_t* depth, hipStream_t s);
hipError_t k_vo_estimate(const ssm_pmatch* m, int n, const ssm_vo_params& P, const int32_t* samples, int iters,
                         double* tr_all, int32_t* count, double* tr_out, int32_t* inliers, int32_t* result, hipStream_t s);
hipError_t k_vo_estimate_batch(const ssm_pmatch* m_all, int stride, const int32_t* n_all, int nb, const ssm_vo_params& P, const uint32_t* rand_stream, int iters,
                               int32_t* consumed, int32_t* rand_off, double* tr_all, int32_t* count, double* tr_out, int32_t* inliers, int32_t* result, hipStream_t s);
hipError_t k_segnet_conv_argmax(const void* in, const void* wt, const float* scale, const float* shift, uint8_t* labels, int n, int H, int W,
                                int CinPad, int Cout, hipStream_t s);
hipError_t k_segnet_conv_pool(const void* in, const void* wt, const float* scale, const float* shift, void* out, uint8_t* code, int n, int H, int W,
                              int CinPad, int Cout, hipStream_t s);
int k_segnet_conv_unpool_available();
hipError_t k_segnet_conv_unpool(const void* pooled, const uint8_t* ucode, const void* wt, const float* scale, const float* shift, void* out, int n, int H, int W,
                                int CinPad, int Cout, hipStream_t s);
hipError_t k_segnet_pool(const void* in, int n, int H, int W, int C, void* out, uint8_t* code, hipStream_t s);
hipError_t k_segnet_unpool(const void* in, const uint8_t* code, int n, int PH, int PW, int C, void* out, int H, int W, hipStream_t s);
hipError_t k_segnet_argmax(const void* logits, int n, int npix, int Cstore, int ncls, uint8_t* labels, hipStream_t s);
hipError_t k_segnet_color(const uint8_t* ids, int n, int sw, int sh, int dw, int dh, const int32_t* xofs, const int16_t* xa,
                          const int32_t* yofs, const int16_t* ya, int pavement_to_road, int nearest, uint8_t* sem_bgr, uint8_t* ids_out, hipStream_t s);

// quad matcher (kernels_quad.hip).  Images of the stereo path live in SLOTS: 2 sides x B1 slots, a slot = the 4-level LK pyramid of one image
// (level l at element offset off[l], packed rows) in `pyr` and its Scharr derivatives (x, y int16 pairs) at the same element offsets in `der`.
// Slot 0 = the carried previous frame, slot 1 + f = frame f of the sub-batch.
struct QuadBatch { const uint8_t* pyr; const int16_t* der; size_t slot_elems; int B1; int w[4], h[4], off[4]; };
hipError_t k_quad_pyramids(const QuadBatch& q, int nb, hipStream_t s);
struct GfttWork { float* eig; int* cand_at; uint32_t* cand_bits; /* one bit per pixel: cand_at != 0; k_quad_gftt_bits_words(w, h) words per frame */ unsigned long long *keys, *kept; uint32_t* deps; uint8_t *depn, *state; int *maxord, *count, *nkept, *overflow; int cap; };
size_t k_quad_gftt_deps_per_candidate();
size_t k_quad_gftt_bits_words(int w, int h);
hipError_t k_quad_gftt(const QuadBatch& q, int nb, int max_corners, double quality, double min_distance, const GfttWork& g, float* pts, int stride, int* ncorner, hipStream_t s);
hipError_t k_quad_lk(const QuadBatch& q, const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err, int max_count, float eps2, float min_eig_thr, hipStream_t s);
hipError_t k_quad_track(const QuadBatch& q, int nb, float* pts, int stride, const int* ncorner, const int* has_prev, void* out, int* nout, hipStream_t s);
hipError_t k_quad_window_match(const float* kp1, const uint8_t* d1, int n1, const float* kp2, const uint8_t* d2, int n2, int sw, int sh, float thr,
                               ssm_dmatch* out, hipStream_t s);

// for the translation units that only use the public ABI (ssm_track.hip): the configuration a context was created with
void ssm_internal_get_config(const ssm_ctx* c, ssm_config* out);
int ssm_internal_get_device(const ssm_ctx* c);      // the HIP device the context lives on: raw HIP calls of another translation unit select it first
