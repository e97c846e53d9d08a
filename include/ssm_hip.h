/* ssm_hip.h -- C ABI of libssm_hip.so, the MI355X (gfx950) implementation of the per-frame semantic-mapping
 * front end of MuMuJun97/semantic_slam_mapping.  Plain pointers and sizes only; no C++/torch types.
 *
 * The reference has no FFI: its hot path sits behind plain C++ classes linked into librgbd_tutor_lib.so
 * (/root/reference/src/CMakeLists.txt:1-5).  Each entry point below names the reference interface it replaces;
 * the C++ classes of the same names (include/ssm/ *.h in this repo) are thin callers of these functions, and
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions: every function returns SSM_OK (0) or a negative ssm_status; nothing throws across the ABI; the caller
 * owns every host buffer and passes capacities; the context owns device memory and one HIP stream; calls on one
 * context are serialised by an internal mutex (the reference calls detectFeatures on the main thread and
 * generatePointCloud on the viewer thread, SURVEY.md s.8b "threading": give each thread its own context or share one).
 * Host-pointer functions are synchronous.  *_dev functions take DEVICE pointers, enqueue on the context stream and
 * return without waiting (ssm_sync waits).
 */
#ifndef SSM_HIP_H
#define SSM_HIP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SSM_OK = 0,
    SSM_E_INVAL = -1,        /* bad argument / unsupported configuration */
    SSM_E_NOMEM = -2,        /* device or host allocation failed */
    SSM_E_HIP = -3,          /* HIP runtime error (ssm_last_error has the text) */
    SSM_E_CAPACITY = -4,     /* caller buffer or internal table too small */
    SSM_E_TOO_FEW_TRAIN = -5,/* matcher needs >= 2 train descriptors (src/orb.cpp:25 indexes [1] unguarded) */
    SSM_E_VOXEL_RANGE = -6,  /* PCL VoxelGrid index-overflow guard would trip: output == input in the reference */
    SSM_E_NODEVICE = -7,
    SSM_E_COMM = -8          /* RCCL error (ssm_last_error has ncclGetErrorString) */
} ssm_status;

/* layout-identical to cv::KeyPoint (OpenCV 2.4), 28 bytes */
typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } ssm_keypoint;
/* layout-identical to cv::DMatch, 16 bytes */
typedef struct { int32_t queryIdx, trainIdx, imgIdx; float distance; } ssm_dmatch;
/* layout-identical to pcl::PointXYZRGBL (a superset of Mapper::PointT = pcl::PointXYZRGBA, include/mapper.h:18), 32 bytes */
typedef struct { float x, y, z, w; uint8_t b, g, r, a; uint32_t label; uint32_t pad[2]; } ssm_point;
/* rgbd_tutor::CAMERA_INTRINSIC_PARAMETERS (include/utils.h:8-16) */
typedef struct { double cx, cy, fx, fy, scale; } ssm_camera;
/* one voxel of the map table: exact integer sums (DESIGN.md "voxel contract"), 112 bytes */
typedef struct {
    int64_t  key;            /* ((k+2^20)<<42)|((j+2^20)<<21)|(i+2^20), ijk = floor(p * (1.0f/leaf)) */
    int64_t  sx, sy, sz;     /* sums of llrint(coord * 2^24) */
    uint64_t sr, sg, sb, n;  /* colour sums, point count */
    uint32_t hist[12];       /* label votes */
} ssm_voxel;

/* keys of /root/reference/parameters.txt that the path reads (include/orb.h:21-28, include/mapper.h:24-29,
 * include/track.h:69-70, src/parameter_reader.cpp:6-18) plus sizing knobs of this implementation */
typedef struct {
    int    width, height;            /* frame geometry the context is sized for */
    int    orb_features;             /* parameters.txt:66 */
    float  orb_scale;                /* :68 */
    int    orb_levels;               /* :69 */
    int    orb_iniThFAST;            /* :70 */
    int    orb_minThFAST;            /* :71 */
    double knn_match_ratio;          /* :72 */
    int    tracker_ref_frames;       /* :81 */
    double mapper_resolution;        /* :97 */
    double mapper_max_distance;      /* :98 */
    ssm_camera camera;               /* :37-40,63 */
    int    max_batch;                /* frames per batched launch (device workspace is sized for this) */
    int    voxel_capacity_log2;      /* slots the context's voxel map STARTS with = 2^this (8..28).  The map grows by itself like the reference's globalMap
                                        (src/mapper.cpp:121-158): it is re-hashed into a larger table whenever it is a quarter full */
    const int8_t* brief_pattern;     /* NULL = built-in 256x4 table; else 1024 int8 (x0,y0,x1,y1)*256 */
    int    voxel_max_capacity_log2;  /* the map never grows beyond 2^this slots (default and maximum 28: 30 GB); a map that needs more fails with SSM_E_CAPACITY */
    /* the stereo path (configs[3]).  0 = the default of each */
    int    sgbm_form;                /* SGBM formulation: 2 (default) = two volumes, top-down sweep with strip hand-offs; 1 = four path volumes, no cross-block waits
                                        (also what a sweep that times out is repeated with); 3 = the round-3 five-volume form.  All give the same disparities */
    int    sgbm_streams;             /* 1 | 2 | 3 streams (and workspaces) that alternate sub-batches of the batched stereo path; default 2 */
    int    stereo_batch;             /* frame pairs per launch of ssm_stereo_seq_process; default: max_batch, at most 128 */
} ssm_config;

typedef struct ssm_ctx ssm_ctx;

void        ssm_config_default(ssm_config* cfg);        /* parameters.txt values, 640x480 TUM fr1 camera at scale 1000 */
int         ssm_create(int device, const ssm_config* cfg, ssm_ctx** out);
void        ssm_destroy(ssm_ctx* ctx);
const char* ssm_last_error(const ssm_ctx* ctx);          /* valid until the next call on ctx; ctx==NULL: last create error */
const char* ssm_version(void);
int         ssm_orb_capacity(const ssm_ctx* ctx);        /* max keypoints per frame: orb_features + 3*orb_levels */
int         ssm_sync(ssm_ctx* ctx);
void*       ssm_stream(ssm_ctx* ctx);                    /* hipStream_t of the context */

/* ---- OrbFeature::detectFeatures (include/orb.h:32-53): gray conversion, ORB, 3-D position per keypoint.
 * img: 8-bit, channels 1 (gray) or 3 (BGR interleaved), row stride in bytes.  depth: u16 row-major w x h or NULL.
 * kps/desc/pos3d hold cap entries (desc 32 B each, pos3d 3 floats each, may be NULL). */
int ssm_orb_extract(ssm_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels, const uint16_t* depth,
                    ssm_keypoint* kps, uint8_t* desc, float* pos3d, int cap, int* n_out);

/* ---- cv::BFMatcher(NORM_HAMMING)::knnMatch(q,t,knn,2) as called at src/orb.cpp:21.  idx/dist: nq x 2 */
int ssm_hamming_knn2(ssm_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist);
/* ---- OrbFeature::match (src/orb.cpp:16-29): knn + ratio test, ascending queryIdx */
int ssm_match(ssm_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio,
              ssm_dmatch* out, int cap, int* n_out);
/* ---- asynchronous forms of the two calls a tracker frame makes (Tracker::trackRefFrame, src/track.cpp:140-163: detectFeatures, then one match per
 * reference frame).  The call stages its inputs in the context's pinned ring, enqueues the copies and kernels on the context stream and returns; the
 * results (and *n_out) are written into the caller's buffers by ssm_wait, which completes every pending call in the order it was made and returns the first
 * error.  Input buffers may be reused at once (they have been copied); OUTPUT buffers and n_out must stay valid until ssm_wait.  ssm_orb_extract / ssm_match
 * are these + ssm_wait; any other entry point of the context may be called in between (same stream: it runs behind the pending work). */
int ssm_orb_extract_async(ssm_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels, const uint16_t* depth,
                          ssm_keypoint* kps, uint8_t* desc, float* pos3d, int cap, int* n_out);
int ssm_match_async(ssm_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio, ssm_dmatch* out, int cap, int* n_out);
int ssm_wait(ssm_ctx* ctx);
/* the loop `for (pFrame : refFrames) orb.match(pFrame, currentFrame)` of Tracker::trackRefFrame (src/track.cpp:150-152) as one call: refs[i] (nrefs[i] x 32
 * bytes) are the query sets, cur (ncur x 32) the train set of every pair; outs[i] (caps[i] entries) / n_outs[i] receive what ssm_match(refs[i], cur) gives.
 * One upload, one matrix-core launch for all pairs, one download.  ncur < 2: SSM_E_TOO_FEW_TRAIN; nrefs[i] == 0: n_outs[i] = 0. */
int ssm_match_refs(ssm_ctx* ctx, const uint8_t* const* refs, const int* nrefs, int nref, const uint8_t* cur, int ncur, double ratio,
                   ssm_dmatch* const* outs, const int* caps, int* n_outs);
int ssm_match_refs_async(ssm_ctx* ctx, const uint8_t* const* refs, const int* nrefs, int nref, const uint8_t* cur, int ncur, double ratio,
                         ssm_dmatch* const* outs, const int* caps, int* n_outs);          /* completed by ssm_wait */

/* ---- Mapper::semantic_motion_fuse (src/mapper.cpp:189-216): sem = BGR class-colour image, mask = w*h bytes */
int ssm_moving_mask(ssm_ctx* ctx, const uint8_t* sem_bgr, int w, int h, int stride, uint8_t* mask);
/* ---- Mapper::generatePointCloud (src/mapper.cpp:12-94) incl. RGBDFrame::project2dTo3d (include/rgbdframe.h:63-75)
 * and pcl::transformPointCloud by T (16 doubles, column-major = Eigen::Isometry3d::matrix(); NULL = camera frame).
 * packed images (stride = w*channels).  out holds cap points (w*h always suffices). */
int ssm_backproject(ssm_ctx* ctx, const uint16_t* depth, const uint8_t* rgb_bgr, const uint8_t* sem_bgr, int w, int h,
                    const ssm_camera* cam, const double* T, double max_distance, ssm_point* out, int cap, int* n_out);

/* ---- pcl::VoxelGrid::filter as used in Mapper::viewer (src/mapper.cpp:106-107,154-155) */
int ssm_voxel_filter(ssm_ctx* ctx, const ssm_point* pts, int n, float leaf, ssm_point* out, int cap, int* n_out);
/* ---- the device-resident form of the two calls above, for Mapper::viewer (src/mapper.cpp:96-171): a key-frame's gated camera-frame cloud is made ONCE
 * (mapper.cpp:17-20 caches frame->pointcloud) and STAYS in device memory as an ssm_cloud; a map update transforms the chosen clouds by their current
 * poses, adds them to the previous filtered map and runs the VoxelGrid pass, all on the device -- the viewer's `*map += *generatePointCloud(kf)` loop and
 * its `voxel.filter(*tmp)` without the key-frame clouds or the whole global map crossing PCIe on every update; only the published map is downloaded.
 * Results are the bytes ssm_backproject + host transform + ssm_voxel_filter give (exact integer sums: independent of the order points are added in). */
typedef struct ssm_cloud ssm_cloud;
int  ssm_backproject_dev(ssm_ctx* ctx, const uint16_t* depth, const uint8_t* rgb_bgr, const uint8_t* sem_bgr, int w, int h,
                         const ssm_camera* cam, double max_distance, ssm_cloud** cloud_out);      /* host images in; camera-frame cloud left on the device */
int  ssm_cloud_size(const ssm_cloud* cloud);
int  ssm_cloud_fetch(ssm_ctx* ctx, const ssm_cloud* cloud, const double* T, ssm_point* out, int cap, int* n_out);   /* transformed copy to the host (T = NULL: as stored) */
void ssm_cloud_free(ssm_ctx* ctx, ssm_cloud* cloud);
/* gives the device memory of the viewer's map path back: every slab whose clouds have all been freed, the concatenation buffer and the map buffer of
 * ssm_viewer_map_update (the next update allocates again).  Mapper::viewer calls it when it falls back to its host path after a device error (out of memory on a long
 * sequence), so that the host path's own device work has room.  test_fail_next: != 0 makes the NEXT ssm_viewer_map_update of the context fail with SSM_E_NOMEM (tests). */
int  ssm_viewer_map_release(ssm_ctx* ctx, int test_fail_next);
/* map <- VoxelGrid(leaf)( (rebuild ? nothing : the previous map's centroids) + sum over i of poses[i] * clouds[i] ); poses: n x 16 doubles, column-major, HOST.
 * A cloud extent PCL's VoxelGrid refuses (dx dy dz > INT_MAX) leaves the unfiltered concatenation as the map, like mapper.h's `*out = *in`. */
int  ssm_viewer_map_update(ssm_ctx* ctx, int rebuild, ssm_cloud* const* clouds, const double* poses, int n, float leaf, int* n_map_out);
int  ssm_viewer_map_fetch(ssm_ctx* ctx, ssm_point* out, int cap, int* n_out);                       /* the map after the last update (sorted by voxel index) */
/* the persistent map of the context: table of exact sums keyed by voxel.  It grows like the reference's globalMap (src/mapper.cpp:121-158) and never drops a
 * contribution below 2^voxel_max_capacity_log2 slots, whatever the stream does: ssm_seq_process sizes the table from the stream's own voxel rate; when a launch brings
 * far more than that (the camera leaves a near wall at a fine leaf), what the table refuses waits in an overflow list, and blocks of the map kernel that start
 * while that list is nearly full add nothing, log themselves and are run again after the table has grown -- which is why the DEVICE INPUTS of an
 * ssm_seq_process call whose map stage ran must stay unchanged until the next ssm_sync / ssm_map_size / ssm_map_export* / ssm_voxel_allgather of the context.
 * Only a map that needs more than the configured maximum loses contributions: the entry point that notices returns SSM_E_CAPACITY once, and from then on
 * ssm_map_size / ssm_map_export* / ssm_voxel_allgather keep returning SSM_E_CAPACITY for this map -- it is incomplete -- until ssm_map_clear.  Skipped points
 * (non-finite, or outside the 21-bit voxel index range) are reported once as SSM_E_VOXEL_RANGE and leave the map usable (pcl::VoxelGrid skips such points too). */
int ssm_map_clear(ssm_ctx* ctx);
int ssm_map_insert(ssm_ctx* ctx, const ssm_point* pts, int n);                 /* globalMap += cloud */
int ssm_map_size(ssm_ctx* ctx, int* n_voxels);
/* how the map got where it is: stats[0] = log2 of the table's slots now, [1] = times it was re-hashed into a larger table, [2] = blocks of the map kernel that were
 * run again after skipping themselves, [3] = records its overflow list holds (a context that has run the map stage of ssm_seq_process keeps the large list) */
int ssm_map_stats(ssm_ctx* ctx, int64_t stats[4]);
int ssm_map_export(ssm_ctx* ctx, ssm_point* out, int cap, int* n_out);         /* centroids sorted by voxel index */
int ssm_map_export_table(ssm_ctx* ctx, ssm_voxel* out, int cap, int* n_out);   /* key-sorted table, for merging */
int ssm_map_merge_table(ssm_ctx* ctx, const ssm_voxel* tab, int n);            /* add another rank's table */
/* same two with DEVICE buffers (the RCCL all-gather of per-GPU tables works on device memory); export is synchronous
 * (it needs the voxel count on the host), merge is enqueued on the context stream.  ssm_map_size and ssm_map_export_table_dev wait for the MAP: after
 * an ssm_seq_process whose map stage ran on a side stream they wait for that stream only, and the call's ORB -> match chain may still be running when
 * they return (its outputs are ordered on the context stream as always; ssm_sync waits for everything).  A loop of ssm_map_clear / ssm_seq_process /
 * ssm_map_export_table_dev per sequence therefore keeps the GPU busy across its iterations. */
int ssm_map_export_table_dev(ssm_ctx* ctx, ssm_voxel* out_dev, int cap, int* n_out);
int ssm_map_merge_table_dev(ssm_ctx* ctx, const ssm_voxel* tab_dev, int n);

/* ---- multi-GPU (BASELINE.json configs[4]).  The reference is one process (experiment/exp_mapping.cpp:18-59), so these have no
 * counterpart there.  One process per GPU; frames shard by contiguous block (each rank first runs its tracker_ref_frames halo frames with
 * stages = SSM_STAGE_ORB, then its block with continue_sequence = 1, so that the match tables equal the single-GPU ones, src/track.cpp:150-152);
 * every rank fuses its frames into its own context map; ssm_voxel_allgather merges the maps with ONE RCCL all-gather over xGMI.
 * Bootstrap without any framework: rank 0 calls ssm_comm_get_unique_id and ships the SSM_COMM_ID_BYTES bytes to the other ranks (file, pipe,
 * MPI, torch.distributed ...); every rank calls ssm_comm_init_rank (= ncclCommInitRank on the context's device). */
#define SSM_COMM_ID_BYTES 128
int ssm_comm_get_unique_id(void* id /* SSM_COMM_ID_BYTES */);
int ssm_comm_init_rank(ssm_ctx* ctx, int nranks, int rank, const void* id);
int ssm_comm_finalize(ssm_ctx* ctx);
int ssm_comm_rank(const ssm_ctx* ctx);
int ssm_comm_size(const ssm_ctx* ctx);
/* COLLECTIVE: every rank of the communicator must call it, and every rank gets the same return code -- the voxel counts travel together with each
 * rank's table-full flag, and an allocation failure of the receive buffer is agreed on by a second 4-byte all-gather, so no rank can leave between two
 * collectives while its peers wait in the next one (on any failure no rank has merged anything).
 * rccl_comm: a ncclComm_t the caller created for this context's device, or NULL = the context's own communicator.  Afterwards the context
 * map of EVERY rank holds the union of all ranks' maps (exact integer sums: bit-identical on every rank and to the single-GPU map).
 * On the context stream: all-gather of the voxel counts, one in-place all-gather of the tables padded to the longest, re-insertion of the
 * nranks - 1 remote tables.  Waits on the host once (for the counts); the merge kernels are enqueued, not waited for. */
int ssm_voxel_allgather(ssm_ctx* ctx, void* rccl_comm);

/* ---- device-resident batched path (the benchmarked one): n frames of a sequence, packed, all DEVICE pointers.
 * Runs detectFeatures for every frame, match(ref, cur) against the <= tracker_ref_frames preceding frames (the
 * refFrames deque of Tracker::trackRefFrame, src/track.cpp:150-152, when every frame tracks), the moving mask,
 * generatePointCloud with pose[f], and fuses every cloud into the context map. */
typedef struct {
    const uint8_t*  bgr;      /* n x h x w x 3 */
    const uint16_t* depth;    /* n x h x w     */
    const uint8_t*  sem_bgr;  /* n x h x w x 3 */
    const double*   pose;     /* n x 16, column-major T_f_w */
    int n;
    int continue_sequence;    /* 1: the last tracker_ref_frames frames of the previous call are the first refs */
    int stages;               /* bit mask of SSM_STAGE_*, 0 = all */
} ssm_frames_dev;
enum { SSM_STAGE_ORB = 1, SSM_STAGE_MATCH = 2, SSM_STAGE_MAP = 4,
       SSM_STAGE_SEGNET = 8 /* BASELINE configs[2]: labels come from the on-GPU SegNet instead of sem_bgr (sem_bgr may be NULL) */ };
typedef struct {              /* DEVICE pointers owned by the context, valid until the next ssm_seq_process/destroy */
    const ssm_keypoint* kps;  /* n x cap */
    const uint8_t*  desc;     /* n x cap x 32 */
    const float*    pos3d;    /* n x cap x 3 */
    const int32_t*  nkp;      /* n */
    const ssm_dmatch* matches;/* n x R x cap ; slot r of frame f = match(frame f-R+r ... ), see nmatch */
    const int32_t*  nmatch;   /* n x R ; -1 where the ref frame does not exist */
    const int32_t*  npoints;  /* n : points emitted by generatePointCloud */
    int cap, R;
} ssm_seq_out_dev;
int ssm_seq_process(ssm_ctx* ctx, const ssm_frames_dev* in, ssm_seq_out_dev* out);

/* ---- Tracker::updateFrame in bulk (src/track.cpp:8-36,140-212): the consumer of ssm_seq_process's match tables that closes the pose loop.
 * ssm_seq_process computes features and match tables for whole sub-sequences ahead of the pose chain; ssm_tracker_run then walks the frames of that
 * call in order and does, per frame, what Tracker::updateFrame does in RGB-D mode: initFirstFrame / trackRefFrame / lostRecover -- the refFrames deque
 * (<= tracker_ref_frames successfully tracked frames), for every reference frame the matches (reference -> current), the 3-D points of the matched
 * reference features moved to the world by the inverse reference pose (track.cpp:150-163), PnPSolver::solvePnP (src/pnp.cpp:5-118) from
 * speed * lastPose, the < 15 correspondences / < 15 inliers tests, cntLost / max_lost_frame -> LOST -> lostRecover.  A reference frame that is one of
 * the tracker_ref_frames frames in front of the current one uses the precomputed table slot; after a tracking failure the deque holds older frames,
 * and those pairs are matched on demand (OrbFeature::match through the same matcher kernels), so the result is the per-frame Tracker's, bit for bit
 * (one numeric contract: include/ssm/pnp_core.h).  The solved poses go back into a stages = SSM_STAGE_MAP pass of ssm_seq_process.
 * The tracker object keeps the state between calls (speed, lastPose, cntLost, the deque with the features of its frames). */
typedef struct ssm_tracker ssm_tracker;
typedef struct {
    int32_t max_lost_frame;   /* tracker_max_lost_frame (include/track.h:69; 10) */
    int32_t ref_frames;       /* tracker_ref_frames (:70; 5) -- must equal the context's */
    int32_t pnp_min_inliers;  /* pnp_min_inliers (include/pnp.h; 10): only PnPSolver's unused return value depends on it */
    int32_t use_device;       /* 1: solve the pose chain of regular frames on the GPU (kernels_pnp.hip: a cluster of eight blocks per chain, one block for an own_stream tracker; SSM_PNP_BLOCKS overrides); 0: on the host.  Same bits. */
    double  first_pose[16];   /* T_f_w the first frame arrives with (initFirstFrame leaves it alone), column-major */
    int32_t own_stream;       /* 1: the device chain of this tracker runs on a stream of its own (behind what the context's stream holds at the time of the
                                 call), so that trackers of independent sequences, driven from different host threads, solve side by side -- a chain is one
                                 block = one CU of 256.  0: on the context's stream */
    int32_t blocks;           /* blocks per device chain: 1, 2, 4 or 8; 0 = default (8; 1 for an own_stream tracker).  Same bits in every form */
} ssm_tracker_params;
typedef struct { int32_t state; /* Tracker::getState() after the frame: 1 OK, 2 LOST */ int32_t tracked; /* 1: the frame joined refFrames */
                 int32_t n_matches; /* correspondences handed to solvePnP (-1: none gathered) */ int32_t n_inliers; } ssm_track_info;
void ssm_tracker_params_default(ssm_tracker_params* p);
int  ssm_tracker_create(ssm_ctx* ctx, const ssm_tracker_params* p, ssm_tracker** out);
void ssm_tracker_destroy(ssm_tracker* t);
int  ssm_tracker_reset(ssm_tracker* t);                       /* back to NOT_READY */
/* seq: the output of the most recent ssm_seq_process on the tracker's context (stages ORB | MATCH at least), n its frame count.  The frames are taken
 * as the continuation of the frames of the previous ssm_tracker_run.  pose_out: n x 16 doubles (HOST, column-major): RGBDFrame::T_f_w of every frame as
 * updateFrame leaves it (a frame that fails to track keeps the prediction speed * refFrames.back()); info_out: n entries (HOST, may be NULL). */
int  ssm_tracker_run(ssm_tracker* t, const ssm_seq_out_dev* seq, int n, double* pose_out, ssm_track_info* info_out);
const char* ssm_tracker_last_error(const ssm_tracker* t);
/* frames solved by the device chain / by the host path so far (use_device = 1: the host path takes the first frame, lostRecover, and the frames whose
 * refFrames deque reaches behind the match-table window after a tracking failure) */
int  ssm_tracker_stats(const ssm_tracker* t, int64_t* device_frames, int64_t* host_frames);
/* what the device chain has computed so far, for a roofline of PnPSolver::solvePnP (src/pnp.cpp:5-118: g2o's optimize(10) x 4 rounds): work[0] = Levenberg iterations
 * (one pass over the edges each: chi2 + normal equations), work[1] = chi2 passes (one per trial + one per optimize), work[2] / work[3] = level-0 edges those evaluated */
int  ssm_tracker_work(const ssm_tracker* t, int64_t work[4]);

/* ---- QuadFeatureMatch (include/quadmatcher.hpp:51-136, src/quadmatcher.cpp): the stereo quad matcher of the KITTI path
 * (Tracker::estimateVO, src/track.cpp:45-55).  Images are 8-bit gray, any size (buffers are re-sized on demand). */
/* layout-identical to struct pmatch (include/quadmatcher.hpp:33-49), 52 bytes */
typedef struct { float u1p, v1p; int32_t i1p; float u2p, v2p; int32_t i2p; float u1c, v1c; int32_t i1c; float u2c, v2c; int32_t i2c; int16_t dis_c, dis_p; } ssm_pmatch;
/* init(DET_GFTT, ..) + detectFeature() + circularMatching() in tracking mode (mode_track = true): GFTT (quality 0.04,
 * minDistance 8, max_corners: cv default 1000) on the current-left image, four pyramidal LK passes lc->rc, rc->rp,
 * rp->lp, lc->lp (11x11 window, 3 pyramid levels, <= 200 iterations, eps 0.01, min-eigen threshold 1e-6),
 * filteringTracks.  out: quadmatches in feature order. */
int ssm_quad_track(ssm_ctx* ctx, const uint8_t* lc, const uint8_t* rc, const uint8_t* lp, const uint8_t* rp, int w, int h, int stride,
                   int max_corners, ssm_pmatch* out, int cap, int* n_out);
/* the two OpenCV calls on their own: cv::goodFeaturesToTrack (blockSize 3, no Harris) and cv::calcOpticalFlowPyrLK
 * (win 11x11, maxLevel 3, OPTFLOW_LK_GET_MIN_EIGENVALS).  pts arrays hold (x, y) float pairs. */
int ssm_gftt(ssm_ctx* ctx, const uint8_t* img, int w, int h, int stride, int max_corners, double quality, double min_distance,
             float* pts, int cap, int* n_out);
int ssm_lk_track(ssm_ctx* ctx, const uint8_t* prev, const uint8_t* next, int w, int h, int stride, const float* prev_pts, int n,
                 float* next_pts, uint8_t* status, float* err, int max_count, double epsilon, double min_eig_threshold);
/* QuadFeatureMatch::matching on binary descriptors (:41-83, caldistance :525-544): windowed brute-force nearest
 * neighbour, one DMatch per query (trainIdx -1 when unmatched), kp = (x, y) float pairs, descriptors 32 B */
int ssm_window_match(ssm_ctx* ctx, const float* kp1, const uint8_t* d1, int n1, const float* kp2, const uint8_t* d2, int n2,
                     int search_width, int search_height, float distance_threshold, ssm_dmatch* out);

/* ---- depth from stereo: calDisparity_SGBM (src/stereo.cpp:11-30, cv::StereoSGBM) and FrameReader's disparity -> depth
 * conversion with the 3-D ROI gate (src/rgbdframe.cpp:81-116) -------------------------------------------------------------
 * params mirror the public fields of cv::StereoSGBM (fullDP = false).  ssm_sgbm_params_default fills what stereo.cpp sets:
 * 80 disparities, SAD window 11, P1 = 4*11*11, P2 = 32*11*11, uniqueness 10, speckle window 100 / range 32, disp12MaxDiff 1,
 * preFilterCap 63.  left / right: rectified 8-bit images.  disp: int16 per pixel, fixed point with 4 fractional bits,
 * (minDisparity - 1) * 16 where no disparity was accepted (what cv::StereoSGBM::operator() writes).  stage 1 stops after
 * computeDisparitySGBM (no medianBlur / filterSpeckles), for tests.  numberOfDisparities: a multiple of 16, <= 128.  minDisparity >= 2: OpenCV 2.4 reads its half-sample
 * interval buffers outside the range it filled there (calcPixelCostBT); this library uses the intervals of the pixels actually compared -- the
 * reference sets minDisparity = 0. */
typedef struct { int32_t minDisparity, numberOfDisparities, SADWindowSize, P1, P2, disp12MaxDiff, preFilterCap, uniquenessRatio,
                 speckleWindowSize, speckleRange; } ssm_sgbm_params;
void ssm_sgbm_params_default(ssm_sgbm_params* p);
int ssm_sgbm(ssm_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h, int stride, const ssm_sgbm_params* params, int stage, int16_t* disp);
/* the whole depth step of FrameReader::next() in KITTI mode: SGBM, then depth = ushort(f * baseline / d * 16 * scale) inside
 * the ROI (|x| < roix, |y| < roiy, 0 < z < roiz), 0 elsewhere and where d is 0 or the image's minimum disparity value.
 * disp (may be NULL) receives the disparity image too. */
int ssm_stereo_depth(ssm_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h, int stride, const ssm_sgbm_params* params,
                     double baseline, double cu, double cv, double f, double roix, double roiy, double roiz, double scale,
                     uint16_t* depth, int16_t* disp);

/* ---- stereo visual odometry on the quad matches: VisualOdometryStereo::estimateMotion (src/vo_stereo.cpp:47-152) --------
 * params mirror VisualOdometryStereo::parameters (include/vo_stereo.hpp: calib.f/cu/cv, base, inlier_threshold,
 * reweighting).  samples: iters x 3 match indices, what VisualOdometry::getRandomSample (src/vo.cpp:74-93) draws per RANSAC
 * iteration -- the host class owns the rand() stream (include/ssm/vo_stereo.hpp).  tr = (rx, ry, rz, tx, ty, tz) of
 * the best hypothesis after refinement; inliers (cap entries) / n_inliers = its consensus set in index order; *success = 0
 * where the reference returns an empty vector (fewer than 6 matches or inliers, refinement not converged).
 * Contracts (sin/cos, LU, summation order): oracle/vo.c. */
typedef struct { double f, cu, cv, base, inlier_threshold; int32_t reweighting, pad; } ssm_vo_params;
int ssm_vo_estimate(ssm_ctx* ctx, const ssm_pmatch* matches, int n, const ssm_vo_params* params, const int32_t* samples, int iters,
                    double tr[6], int32_t* inliers, int cap, int* n_inliers, int* success);

/* ---- PnPSolver::solvePnP (reference src/pnp.cpp:5-118) for ONE correspondence list, on the device -----------------------------------
 * The per-frame caller's entry point (Tracker::trackRefFrame, src/track.cpp:166-175; PnPSolver::solvePnPLazy, src/pnp.cpp:120-226): img = n x 2
 * pixels in frame 2, obj = n x 3 points in frame 1's camera frame ((0,0,0) = no depth: skipped as pnp.cpp:34 does), cam = (fx, fy, cx, cy),
 * T = column-major 4 x 4, initial value in, estimate out.  inliers (n bytes, may be NULL) = the flag vector as pnp.cpp keeps it (SURVEY.md
 * Appendix A quirk 14), *n_inliers = the number of set flags, *success (may be NULL) = the reference's return value (n > min_inliers: the
 * vector's LENGTH is what pnp.cpp:115 tests).  One 1024-thread block of kernels_pnp.hip; the result is the same bits as ssm_pnp::solve of
 * include/ssm/pnp_core.h on the host (lane-ordered sums, polynomial sin / cos) and as oracle/pnp.c.  n <= 65535. */
int ssm_pnp_solve(ssm_ctx* ctx, const float* img, const float* obj, int n, const double cam[4], int min_inliers, double T[16],
                  uint8_t* inliers, int* n_inliers, int* success);

/* ---- device-resident batched stereo path (BASELINE.json configs[3]): n frames of a rectified stereo sequence, all DEVICE pointers.
 * The reference walks the KITTI sequence one frame at a time: FrameReader::next() computes the depth of the current pair with SGBM
 * (src/rgbdframe.cpp:64-116, src/stereo.cpp:11-30), Tracker::estimateVO builds a QuadFeatureMatch on (current left, current right, previous left,
 * previous right), runs detectFeature + circularMatching (src/track.cpp:45-59, src/quadmatcher.cpp:388-417,548-588) and hands the quad matches to
 * VisualOdometryStereo::Process (src/track.cpp:62, src/vo_stereo.cpp:18-152).  Frame pairs are independent of each other (the pose chain only
 * multiplies the per-frame motions), so this entry point runs those three stages for EVERY frame of the sequence in bulk: frame f is "current",
 * frame f - 1 "previous" (frame 0: the last frame of the previous call when continue_sequence = 1, else it has no quad matches: nquad = -1).
 * The per-pair entry points above (ssm_quad_track, ssm_gftt, ssm_lk_track, ssm_sgbm, ssm_stereo_depth) run the same kernels with one frame. */
enum { SSM_STEREO_QUAD = 1, SSM_STEREO_DEPTH = 2, SSM_STEREO_VO = 4 /* needs SSM_STEREO_QUAD */ };
typedef struct {
    const uint8_t* left;      /* n x h x w, 8-bit gray, packed */
    const uint8_t* right;     /* n x h x w */
    int n, w, h;
    int continue_sequence;    /* 1: frame 0's previous pair is the last frame of the previous call on this context (same w, h; a per-pair stereo call in between ends the sequence) */
    int stages;               /* bit mask of SSM_STEREO_*, 0 = all */
    int max_corners;          /* cv::goodFeaturesToTrack maxCorners; QuadFeatureMatch uses the OpenCV default 1000 */
    ssm_sgbm_params sgbm;     /* ssm_sgbm_params_default = src/stereo.cpp:16-27 */
    double baseline, cu, cv, f, roix, roiy, roiz, scale;      /* the depth conversion, as in ssm_stereo_depth */
    ssm_vo_params vo;
    int ransac_iters;         /* VisualOdometryStereo::parameters::ransac_iters (200) */
    /* n * ransac_iters * 3 RAW rand() outputs, in the order VisualOdometry::getRandomSample (src/vo.cpp:74-93) would draw them: the host class owns
     * the stream (srand(0) in its constructor).  A frame with >= 6 quad matches consumes 3 * ransac_iters draws (r % N, r % (N-1), r % (N-2) per
     * hypothesis), a frame with fewer consumes none (src/vo_stereo.cpp:61-63); *rand_draws_used of the output says how far the stream advanced. */
    const uint32_t* rand_stream;
} ssm_stereo_frames_dev;
typedef struct {              /* DEVICE pointers owned by the context, valid until the next ssm_stereo_seq_process / per-pair stereo call / destroy */
    const ssm_pmatch* quad;   /* n x max_corners: QuadFeatureMatch::quadmatches of frame f */
    const int32_t* nquad;     /* n ; -1 = frame without a previous frame */
    const float*   corners;   /* n x max_corners x 2: the GFTT corners of the current-left image (strength order) */
    const int32_t* ncorners;  /* n */
    const int16_t* disp;      /* n x h x w: cv::StereoSGBM output (x16 fixed point) */
    const uint16_t* depth;    /* n x h x w: FrameReader's depth image */
    const double*  tr;        /* n x 6: (rx, ry, rz, tx, ty, tz) as ssm_vo_estimate */
    const int32_t* inliers;   /* n x max_corners: consensus set in index order */
    const int32_t* vo_result; /* n x 2: {n_inliers, success} */
    const int32_t* rand_draws_used;   /* 1 int: draws of rand_stream consumed by this call */
    int max_corners;
} ssm_stereo_out_dev;
int ssm_stereo_seq_process(ssm_ctx* ctx, const ssm_stereo_frames_dev* in, ssm_stereo_out_dev* out);
/* frame pairs per launch of the stereo path: ssm_config.stereo_batch, else min(config max_batch, 128) (0.23 GB of SGBM workspace per pair in the default
   formulation -- three cost-volume-sized buffers at 1241 x 376 x 80 --, two workspaces; 128 per launch measured 1 % above 64) */
int ssm_stereo_batch(const ssm_ctx* ctx);

/* ---- Classifier (include/segnet.h:22-46, src/segnet.cpp): SegNet driving_webdemo forward, fp16 MFMA, on the device.
 * Topology is fixed (VGG-16 encoder / mirrored decoder, 26 conv3x3 layers, 12 classes, 480x360 net input); weights
 * are DATA: the .caffemodel is not in the reference tree (README.md:25-32), so the caller supplies every layer.
 * layer l: weight[Cout][Cin][3][3] fp32 (Caffe blob order), scale[Cout], shift[Cout] = conv bias + BatchNorm folded:
 * y = scale * conv(x) + shift, then ReLU for every layer but the last. */
int ssm_segnet_num_layers(void);
int ssm_segnet_layer_shape(int layer, int* cin, int* cout, int* h, int* w);
int ssm_segnet_set_layer(ssm_ctx* ctx, int layer, const float* weight, const float* scale, const float* shift);
/* Classifier::Classify on one host frame (any size equal to the context geometry): Preprocess (cv::resize to 480x360,
 * float, mean 0; segnet.cpp:130-167) -> forward -> ArgMax.  labels_net: 360*480 class ids (may be NULL).
 * sem_bgr (may be NULL): the colour-label image at FRAME size produced like experiment/segnet.cpp:80-83,131-146
 * (Pavement->Road remap, resize of the ids, LUT through the 12-colour palette). */
int ssm_segnet_forward(ssm_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, uint8_t* labels_net, uint8_t* sem_bgr);
/* n device frames (packed BGR); outputs are device buffers or NULL.  flags: bit0 = nearest-neighbour resize of the ids
 * instead of the reference's bilinear-on-ids, bit1 = skip the Pavement->Road remap, bit2 = materialise the class logits
 * (ssm_segnet_logits can then read frame 0 of the last sub-batch; without it the ArgMax runs in the last layer's epilogue
 * and the logits never reach memory -- the labels are identical either way) */
int ssm_segnet_forward_dev(ssm_ctx* ctx, const uint8_t* bgr_dev, int n, uint8_t* labels_net_dev, uint8_t* sem_bgr_dev, int flags);
/* single layer ops on host NHWC fp16 tensors, for exact per-op tests (integer-valued data makes fp16/fp32 exact):
 * op 0 = conv layer `arg` on in[H][W][CinPad16] -> out[H][W][CoutPad16]; op 1 = max-pool 2x2 s2 ceil with C = arg:
 * in[H][W][C] -> out[PH][PW][C] + code[PH][PW][C]; op 2 = unpool: in[PH][PW][C] + code -> out[H][W][C];
 * op 3 = conv layer `arg` + max-pool through the fused kernel: out[PH][PW][CoutPad16] + code (same shape);
 * op 4 = un-pool + conv layer `arg` through the fused kernel: in[PH][PW][CinPad16] + code (same shape) -> out[H][W][CoutPad16] */
int ssm_segnet_debug_op(ssm_ctx* ctx, int op, int arg, const uint16_t* in, int H, int W, uint16_t* out, uint8_t* code);
/* class logits (12 floats per net pixel, 360*480 pixels) of frame 0 of the most recent forward: for tolerance tests */
int ssm_segnet_logits(ssm_ctx* ctx, float* out);

/* per-stage device time of the most recent ssm_seq_process, measured with hipEvents on the stream each stage runs on.
 * ssm_seq_process runs the (SegNet ->) map stage of a sub-batch on a second stream beside the ORB -> match chain, so with
 * on = 1 the stage times overlap (their sum exceeds the wall time); on = 2 additionally keeps everything on the context
 * stream, which gives each stage's undisturbed duration (and a lower throughput).  0 = off.
 * names/ms/launches hold cap entries; returns count in *n_out. */
int ssm_set_profiling(ssm_ctx* ctx, int on);
int ssm_get_stage_times(ssm_ctx* ctx, const char** names, float* ms, int* launches, int cap, int* n_out);

/* ---- utilities (device memory without torch; synthetic stream generator for bench/tests) */
int ssm_dev_alloc(ssm_ctx* ctx, size_t bytes, void** out);
int ssm_dev_free(ssm_ctx* ctx, void* p);
int ssm_dev_mem_info(ssm_ctx* ctx, size_t* free_bytes, size_t* total_bytes);      /* hipMemGetInfo on the context's device */
int ssm_memcpy_h2d(ssm_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int ssm_memcpy_d2h(ssm_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* the same upload, enqueued on the context stream without waiting: src_host must stay valid and unchanged until the next ssm_sync / synchronous call of the context
 * (BatchTracker uploads a frame when it is queued, while the caller reads the next one) */
int ssm_memcpy_h2d_async(ssm_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
/* the download, enqueued without waiting: dst_host holds the data after the next ssm_sync (page-locked dst_host: a plain DMA; BatchStereoTracker fetches a chunk's depth images so) */
int ssm_memcpy_d2h_async(ssm_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* page-locked host memory for frame buffers (the reference's cv::Mat data comes from cv::imread / a camera driver: a cv::Mat can wrap user memory, cv::Mat(rows, cols,
 * type, ptr)).  The host-pointer calls stage pageable inputs through a pinned ring (one extra pass over every image: ~0.06 ms of ssm_orb_extract's 0.25 for a 640 x 480
 * frame); inputs that already live in memory from ssm_host_alloc are read by the DMA engine / the kernels where they are.  Any device of the process may use the
 * memory.  ssm_host_free(NULL) is a no-op. */
int ssm_host_alloc(size_t bytes, void** out);
int ssm_host_free(void* p);
/* synthetic 640x480 RGB-D + 12-class stream of BASELINE.json configs[1]; device pointers;
 * label_ids may be NULL */
int ssm_synth_frames_dev(ssm_ctx* ctx, uint64_t seed, int first_frame, int n, int w, int h,
                         uint8_t* bgr, uint16_t* depth, uint8_t* sem_bgr, uint8_t* label_ids, double* pose);

#ifdef __cplusplus
}
#endif
#endif
