/* ssm_oracle.h -- CPU ORACLE for the per-frame semantic-mapping front end.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product (libssm_hip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: /root/reference has no tests, fixtures or golden vectors
 * (SURVEY.md s.4), it cannot be compiled here (OpenCV 2.4 / PCL 1.7 / Caffe /
 * g2o and the un-vendored Thirdparty/orbslam_modified + Thirdparty/DBoW2 are
 * absent), so this restatement is pinned only by (i) the in-tree source it
 * follows line by line where the arithmetic IS in the tree (matcher ratio
 * test, unprojection, point gating, moving mask) and (ii) the published
 * algorithms of the third-party routines the reference calls
 * (ORB_SLAM2::ORBextractor, OpenCV 2.4 cvtColor / resize / FAST / GaussianBlur
 * / BFMatcher / dilate, PCL 1.7 transformPointCloud / VoxelGrid), each choice
 * documented as a CHOSEN CONTRACT in DESIGN.md.
 *
 * Single-threaded, deterministic, plain C99.  Build: oracle/Makefile with
 * -ffp-contract=off (the float contracts below forbid FMA contraction).
 */
#ifndef SSM_ORACLE_H
#define SSM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* layout-identical to cv::KeyPoint (OpenCV 2.4): 28 bytes */
typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } sso_keypoint;
/* layout-identical to cv::DMatch: 16 bytes */
typedef struct { int32_t queryIdx, trainIdx, imgIdx; float distance; } sso_dmatch;
/* layout-identical to pcl::PointXYZRGBL (superset of PointXYZRGBA): 32 bytes */
typedef struct { float x, y, z, w; uint8_t b, g, r, a; uint32_t label; uint32_t pad[2]; } sso_point;
/* rgbd_tutor::CAMERA_INTRINSIC_PARAMETERS (include/utils.h:8-16), distortion unused on the path */
typedef struct { double cx, cy, fx, fy, scale; } sso_camera;

/* ---------------- K1/K2/K5a image primitives ---------------- */
/* cv::cvtColor BGR2GRAY 8u (called at include/orb.h:39): Y=(B*1868+G*9617+R*4899+8192)>>14 */
void sso_bgr2gray(const uint8_t* bgr, int w, int h, int stride, uint8_t* gray);
/* cv::resize INTER_LINEAR 8u, fixed point 11 bits (ORBextractor::ComputePyramid) */
void sso_resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);
/* cv::GaussianBlur 7x7 sigma 2 BORDER_REFLECT_101 8u, fixed point taps {18,34,49,55,49,34,18}/256 per pass */
void sso_gaussian7(const uint8_t* src, int w, int h, uint8_t* dst);
/* resize coefficient tables (shared formula with the product's host code): ofs[d], coef[2*d] */
void sso_resize_tables(int ssize, int dsize, int32_t* ofs, int16_t* coef);

/* ---------------- K1..K5 ORB extractor (ORB_SLAM2::ORBextractor, call site include/orb.h:21-26,44) */
typedef struct sso_orb sso_orb;
sso_orb* sso_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th);
void     sso_orb_destroy(sso_orb*);
void     sso_orb_set_pattern(sso_orb*, const int8_t pattern[1024]);
int      sso_orb_capacity(const sso_orb*);               /* nfeatures + 3*nlevels */
int      sso_orb_level_size(const sso_orb*, int w, int h, int level, int* lw, int* lh);
int      sso_orb_features_per_level(const sso_orb*, int level);
/* full extraction on a gray image.  kps/desc sized sso_orb_capacity().  returns N */
int      sso_orb_extract(sso_orb*, const uint8_t* gray, int w, int h, int stride,
                         sso_keypoint* kps, uint8_t* desc);
/* stage taps for tests (valid after sso_orb_extract): level image, blurred level image,
 * FAST candidates of a level before the quad-tree (x,y level coords rel. to image, score) */
const uint8_t* sso_orb_level_image(const sso_orb*, int level, int blurred);
int      sso_orb_level_candidates(const sso_orb*, int level, int32_t* xys /*3 per cand*/, int cap);
/* the scalar pieces, exposed for KATs */
int      sso_fast_score(const uint8_t* p, int stride);   /* max(bright,dark arc-min) at p; corner iff > t; cv score = this-1 */
float    sso_fast_atan2(float y, float x);               /* cv::fastAtan2 (2.4), degrees */
void     sso_sincos(float angle_rad, float* s, float* c);/* contract sin/cos used to steer BRIEF */

/* ---------------- K6 matcher (src/orb.cpp:16-29) ---------------- */
/* cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, 2): idx/dist are nq x 2; needs nt >= 2 (returns -1 otherwise) */
int sso_hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist);
/* knn + Lowe ratio d0 < ratio*d1 (double compare of float distances); returns number kept, -1 if nt<2 */
int sso_match(const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio, sso_dmatch* out);

/* ---------------- K10/K11 mapper front half (src/mapper.cpp:12-94,189-216) ---------------- */
/* RGBDFrame::project2dTo3d (include/rgbdframe.h:63-75) */
void sso_project2dTo3d(const uint16_t* depth, int w, int h, const sso_camera* cam, int u, int v, float out[3]);
/* semantic_motion_fuse: 255 where pedestrian(0,64,64)|cyclist(192,128,0) BGR, then 3x3 ones dilate x2 */
void sso_moving_mask(const uint8_t* sem_bgr, int w, int h, uint8_t* mask);
/* class colour -> id 0..11, 255 if not in the 12-class palette */
int  sso_label_of_bgr(uint8_t b, uint8_t g, uint8_t r);
extern const uint8_t sso_palette_bgr[12][3];
/* generatePointCloud: gates + unprojection + rgb + label, row-major order, then T (column-major 4x4 double,
 * Eigen layout) applied as pcl::transformPointCloud.  T==NULL -> camera-frame cloud.  returns P */
int  sso_backproject(const uint16_t* depth, const uint8_t* rgb_bgr, const uint8_t* sem_bgr, const uint8_t* mask,
                     int w, int h, const sso_camera* cam, const double* T, double max_distance, sso_point* out);

/* ---------------- K12 voxel fusion (pcl::VoxelGrid, src/mapper.cpp:106-107,154-155) ---------------- */
/* table form: key -> exact integer sums.  24 fractional bits */
typedef struct {
    int64_t  key;            /* ((k+2^20)<<42)|((j+2^20)<<21)|(i+2^20), ijk=floor(p*inv_leaf) */
    int64_t  sx, sy, sz;     /* sum of llrint(coord * 2^24) */
    uint64_t sr, sg, sb;     /* colour sums */
    uint64_t n;              /* points */
    uint32_t hist[12];       /* label votes (ids 0..11 only) */
} sso_voxel;
int64_t sso_voxel_key(float x, float y, float z, float inv_leaf);
/* accumulate points into a key-sorted table (tab has cap entries, *m used); returns new m or -1 if cap exceeded */
int  sso_voxel_accumulate(const sso_point* pts, int n, float leaf, sso_voxel* tab, int m, int cap);
/* merge src table into dst (both key-sorted); returns new m or -1 */
int  sso_voxel_merge(sso_voxel* dst, int m, int cap, const sso_voxel* src, int ms);
/* centroids, sorted by key (== PCL's linear-index order) */
void sso_voxel_export(const sso_voxel* tab, int m, sso_point* out);
/* pcl::VoxelGrid::filter in one call: returns number of output points, -1 on cap overflow,
 * -2 if the PCL index-overflow guard would trip (dx*dy*dz > INT_MAX: PCL returns the input unfiltered) */
int  sso_voxel_filter(const sso_point* pts, int n, float leaf, sso_point* out, int cap);

/* ---------------- a5/a6 stereo quad-matcher (src/quadmatcher.cpp) ---------------- */
/* layout-identical to struct pmatch (include/quadmatcher.hpp:33-49): 52 bytes */
typedef struct { float u1p, v1p; int32_t i1p; float u2p, v2p; int32_t i2p; float u1c, v1c; int32_t i1c; float u2c, v2c; int32_t i2c; int16_t dis_c, dis_p; } sso_pmatch;
void sso_min_eigen_map(const uint8_t* img, int w, int h, float* eig);
int  sso_gftt(const uint8_t* img, int w, int h, int max_corners, double quality, double min_distance, float* pts);
void sso_pyrdown(const uint8_t* src, int w, int h, uint8_t* dst);
void sso_scharr(const uint8_t* src, int w, int h, int16_t* dxdy);
void sso_lk_track(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err,
                  int max_count, double epsilon, double min_eig_threshold);
int  sso_filter_tracks(const float* lc, const float* rc, const float* lp, const float* rp, const float* lp_direct, int n, sso_pmatch* out);
int  sso_quad_track(const uint8_t* lc, const uint8_t* rc, const uint8_t* lp, const uint8_t* rp, int w, int h, int max_corners, sso_pmatch* out);
int  sso_window_match(const float* kp1, const uint8_t* d1, int n1, const float* kp2, const uint8_t* d2, int n2,
                      int search_w, int search_h, float distance_threshold, sso_dmatch* out);
int  sso_quad_chain(const float* k_lc, const float* k_rc, const float* k_rp, const float* k_lp, int n_lc,
                    const sso_dmatch* m_lrc, const sso_dmatch* m_rcp, const sso_dmatch* m_rlp, sso_pmatch* out);

/* ---------------- stereo visual odometry on the quad matches (src/vo_stereo.cpp, src/vo.cpp): see vo.c ---------------- */
typedef struct { int32_t r[34]; int f, b; } sso_rand_state;             /* glibc rand(): TYPE_3 additive feedback */
void    sso_rand_seed(sso_rand_state* st, unsigned seed);               /* srand(seed) */
int32_t sso_rand_next(sso_rand_state* st);                              /* rand() */
void    sso_vo_random_sample(sso_rand_state* st, int N, int num, int32_t* out);   /* VisualOdometry::getRandomSample */
void    sso_sincos64(double x, double* s, double* c);                   /* the sin/cos contract shared with the HIP kernel */
int     sso_solve6_lu(double A[36], double b[6]);                       /* cv::solve(.., DECOMP_LU), 6x6; b <- x; 0 = singular */
typedef struct { double f, cu, cv, base, inlier_threshold; int32_t reweighting, pad; } sso_vo_params;
/* estimateMotion: samples = iters x 3 match indices (from sso_vo_random_sample).  tr = (rx, ry, rz, tx, ty, tz);
 * inliers (cap n) / n_inliers = the best RANSAC consensus set.  returns 1 on success (refinement converged, >= 6 inliers) */
int     sso_vo_estimate(const sso_pmatch* m, int n, const sso_vo_params* P, const int32_t* samples, int iters,
                        double tr[6], int32_t* inliers, int* n_inliers);
void    sso_vo_tr_to_matrix(const double tr[6], double T[16]);          /* transformationVectorToMatrix, row-major */

/* ---------------- depth from stereo (src/stereo.cpp:11-30 cv::StereoSGBM; src/rgbdframe.cpp:81-116): see sgbm.c ---------------- */
typedef struct { int32_t minDisparity, numberOfDisparities, SADWindowSize, P1, P2, disp12MaxDiff, preFilterCap, uniquenessRatio,
                 speckleWindowSize, speckleRange; } sso_sgbm_params;
int  sso_sgbm_raw(const uint8_t* left, const uint8_t* right, int w, int h, const sso_sgbm_params* p, int16_t* disp);   /* computeDisparitySGBM */
void sso_median3_s16(const int16_t* src, int w, int h, int16_t* dst);                                                  /* cv::medianBlur 3 */
void sso_filter_speckles(int16_t* img, int w, int h, int newVal, int maxSpeckleSize, int maxDiff);                    /* cv::filterSpeckles */
int  sso_sgbm(const uint8_t* left, const uint8_t* right, int w, int h, const sso_sgbm_params* p, int16_t* disp);       /* StereoSGBM::operator() */
void sso_disparity_to_depth(const int16_t* disp, int w, int h, double baseline, double cu, double cv, double f,
                            double roix, double roiy, double roiz, double scale, uint16_t* depth);

/* ---------------- synthetic stream, integer-only ---------------- */
void sso_synth_frame(uint64_t seed, int frame_id, int w, int h,
                     uint8_t* bgr, uint16_t* depth, uint8_t* sem_bgr, uint8_t* label_ids);
void sso_synth_pose(int frame_id, double T[16]);   /* translation (0.01*frame_id,0,0), column-major */

/* ---------------- whole per-frame path, for the CPU baseline ---------------- */
typedef struct {
    int w, h, nfeatures, nlevels, ini_th, min_th, ref_frames;
    float scale_factor, leaf;
    double ratio, max_distance;
    sso_camera cam;
    uint64_t seed;
} sso_pipeline_cfg;
typedef struct {
    double t_synth, t_orb, t_match, t_mask, t_backproject, t_voxel;   /* seconds, summed over frames */
    int64_t keypoints, matches, points, voxels;
    uint64_t checksum;                                                /* order-sensitive FNV over all outputs */
} sso_pipeline_stats;
/* runs frames [first, first+count): synth -> orb -> match vs <=ref_frames previous -> mask -> backproject -> voxel */
int sso_pipeline_run(const sso_pipeline_cfg* cfg, int first, int count, sso_pipeline_stats* st);
/* PnPSolver::solvePnP (/root/reference/src/pnp.cpp:5-118) with g2o's Levenberg restated: see pnp.c */
int sso_pnp_solve(const float* img, const float* obj, int n, const sso_camera* cam, int min_inliers, double T[16], int* inliers_out, int* n_inliers);

#ifdef __cplusplus
}
#endif
#endif
