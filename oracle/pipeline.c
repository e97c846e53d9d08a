/* oracle/pipeline.c -- CPU oracle (TEST INFRASTRUCTURE, see ssm_oracle.h): the whole per-frame path in the order
 * the reference runs it, single-threaded like the reference (no -fopenmp, /root/reference/CMakeLists.txt:11).  This
 * is the "cpu_baseline" leg of bench.py (kind "port") and the end-to-end checksum the GPU pipeline is tested against.
 *
 *   per frame f:  Tracker::trackRefFrame (/root/reference/src/track.cpp:140-200):
 *                   orb->detectFeatures(cur)                      include/orb.h:32-53  (gray, ORB, 3-D positions)
 *                   for ref in refFrames (<= tracker_ref_frames):  orb->match(ref, cur)   src/orb.cpp:16-29
 *                 Mapper::generatePointCloud(cur) (src/mapper.cpp:12-94): moving mask, gated back-projection, T_f_w
 *                 map fusion: one pcl::VoxelGrid over everything added (src/mapper.cpp:121-131,154-155 "redraw" form)
 *   Poses are the stream's ground truth (PnP / pose graph are out of scope, SURVEY.md s.2 #8,#11), every frame is
 *   treated as a keyframe and every tracked frame becomes a reference frame.
 */
#include "ssm_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static uint64_t fnv(uint64_t h, const void* p, size_t n)
{
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001B3ULL; }
    return h;
}

int sso_pipeline_run(const sso_pipeline_cfg* cfg, int first, int count, sso_pipeline_stats* st)
{
    const int w = cfg->w, h = cfg->h, R = cfg->ref_frames > 0 ? cfg->ref_frames : 1;
    memset(st, 0, sizeof(*st));
    st->checksum = 0xCBF29CE484222325ULL;
    sso_orb* orb = sso_orb_create(cfg->nfeatures, cfg->scale_factor, cfg->nlevels, cfg->ini_th, cfg->min_th);
    if (!orb) return -1;
    const int cap = sso_orb_capacity(orb);
    uint8_t* bgr = (uint8_t*)malloc((size_t)w * h * 3); uint8_t* sem = (uint8_t*)malloc((size_t)w * h * 3);
    uint16_t* depth = (uint16_t*)malloc((size_t)w * h * 2); uint8_t* gray = (uint8_t*)malloc((size_t)w * h);
    uint8_t* mask = (uint8_t*)malloc((size_t)w * h);
    sso_point* pts = (sso_point*)malloc(sizeof(sso_point) * (size_t)w * h);
    sso_keypoint* kps = (sso_keypoint*)malloc(sizeof(sso_keypoint) * cap);
    float* pos = (float*)malloc(sizeof(float) * 3 * cap);
    uint8_t** rdesc = (uint8_t**)malloc(sizeof(uint8_t*) * (R + 1)); int* rn = (int*)calloc(R + 1, sizeof(int));
    for (int i = 0; i <= R; i++) rdesc[i] = (uint8_t*)malloc((size_t)cap * 32);
    sso_dmatch* dm = (sso_dmatch*)malloc(sizeof(sso_dmatch) * cap);
    int vcap = 1 << 20, vm = 0;
    sso_voxel* vox = (sso_voxel*)malloc(sizeof(sso_voxel) * vcap);
    int nref = 0, rc = 0;
    for (int f = first; f < first + count && rc == 0; f++) {
        double t0 = now_s();
        sso_synth_frame(cfg->seed, f, w, h, bgr, depth, sem, NULL);
        double T[16]; sso_synth_pose(f, T);
        double t1 = now_s();
        /* detectFeatures */
        sso_bgr2gray(bgr, w, h, w * 3, gray);
        uint8_t* cur = rdesc[R];
        int n = sso_orb_extract(orb, gray, w, h, w, kps, cur);
        if (n < 0) { rc = -2; break; }
        for (int i = 0; i < n; i++) sso_project2dTo3d(depth, w, h, &cfg->cam, (int)kps[i].x, (int)kps[i].y, pos + 3 * i);   /* orb.h:50 truncation */
        st->checksum = fnv(st->checksum, kps, sizeof(sso_keypoint) * n);
        st->checksum = fnv(st->checksum, cur, (size_t)n * 32);
        st->checksum = fnv(st->checksum, pos, sizeof(float) * 3 * n);
        st->keypoints += n;
        double t2 = now_s();
        /* match against the reference frames, oldest first (std::deque order, track.cpp:150) */
        for (int r = 0; r < nref; r++) {
            if (n < 2) continue;
            int nm = sso_match(rdesc[r], rn[r], cur, n, cfg->ratio, dm);
            st->checksum = fnv(st->checksum, dm, sizeof(sso_dmatch) * nm);
            st->matches += nm;
        }
        double t3 = now_s();
        sso_moving_mask(sem, w, h, mask);
        double t4 = now_s();
        int P = sso_backproject(depth, bgr, sem, mask, w, h, &cfg->cam, T, cfg->max_distance, pts);
        st->checksum = fnv(st->checksum, pts, sizeof(sso_point) * P);
        st->points += P;
        double t5 = now_s();
        vm = sso_voxel_accumulate(pts, P, cfg->leaf, vox, vm, vcap);
        if (vm < 0) { rc = -3; break; }
        double t6 = now_s();
        /* refFrames.push_back(cur); pop_front beyond tracker_ref_frames (track.cpp:192-196) */
        if (nref < R) { uint8_t* t = rdesc[nref]; rdesc[nref] = cur; rdesc[R] = t; rn[nref] = n; nref++; }
        else { uint8_t* t = rdesc[0]; for (int i = 0; i < R - 1; i++) { rdesc[i] = rdesc[i+1]; rn[i] = rn[i+1]; } rdesc[R-1] = cur; rn[R-1] = n; rdesc[R] = t; }
        st->t_synth += t1 - t0; st->t_orb += t2 - t1; st->t_match += t3 - t2; st->t_mask += t4 - t3; st->t_backproject += t5 - t4; st->t_voxel += t6 - t5;
    }
    if (rc == 0) {
        sso_point* ex = (sso_point*)malloc(sizeof(sso_point) * (vm > 0 ? vm : 1));
        double t0 = now_s();
        sso_voxel_export(vox, vm, ex);
        st->t_voxel += now_s() - t0;
        st->checksum = fnv(st->checksum, ex, sizeof(sso_point) * vm);
        st->voxels = vm;
        free(ex);
    }
    free(bgr); free(sem); free(depth); free(gray); free(mask); free(pts); free(kps); free(pos); free(dm); free(vox);
    for (int i = 0; i <= R; i++) free(rdesc[i]);
    free(rdesc); free(rn);
    sso_orb_destroy(orb);
    return rc;
}
