/* san_check.c -- drives every translation unit of the CPU oracle under a sanitizer (TEST INFRASTRUCTURE: `make -C oracle SAN=asan|ubsan|tsan san`).
 * The oracle is the checker of the HIP path, so memory errors / undefined behaviour / data races in it would silently weaken every parity claim.
 * One pass over the stages on small inputs; with SAN=tsan the whole-pipeline entry point also runs on four threads at once, the way bench.py's
 * all-cores CPU baseline calls it (the reference's own races at src/mapper.cpp:114-136 are what NOT to inherit). */
#include "ssm_oracle.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void* pipe_thread(void* arg)
{
    const int first = *(int*)arg;
    sso_pipeline_cfg c; memset(&c, 0, sizeof(c));
    c.w = 320; c.h = 240; c.nfeatures = 300; c.nlevels = 5; c.ini_th = 20; c.min_th = 7; c.ref_frames = 3; c.scale_factor = 1.2f; c.leaf = 0.1f;
    c.ratio = 0.8; c.max_distance = 40.0; c.cam.cx = 159.3; c.cam.cy = 127.6; c.cam.fx = 258.6; c.cam.fy = 258.2; c.cam.scale = 1000.0; c.seed = 0x5EED0000ull;
    sso_pipeline_stats st;
    if (sso_pipeline_run(&c, first, 3, &st) != 0 || st.keypoints <= 0) { fprintf(stderr, "pipeline failed\n"); exit(3); }
    return NULL;
}
int main(void)
{
    const int W = 320, H = 240;
    uint8_t *bgr = malloc((size_t)W * H * 3), *sem = malloc((size_t)W * H * 3), *lab = malloc((size_t)W * H), *gray = malloc((size_t)W * H), *mask = malloc((size_t)W * H);
    uint16_t* depth = malloc((size_t)W * H * 2);
    sso_synth_frame(0x5EED0000ull, 3, W, H, bgr, depth, sem, lab);
    sso_bgr2gray(bgr, W, H, W * 3, gray);
    /* ORB + matcher */
    sso_orb* o = sso_orb_create(300, 1.2f, 5, 20, 7);
    const int cap = sso_orb_capacity(o);
    sso_keypoint* kps = malloc(sizeof(sso_keypoint) * cap); uint8_t* desc = malloc((size_t)cap * 32);
    const int n = sso_orb_extract(o, gray, W, H, W, kps, desc);
    sso_dmatch* dm = malloc(sizeof(sso_dmatch) * (n + 1));
    const int nm = n >= 2 ? sso_match(desc, n, desc, n, 0.8, dm) : 0;
    /* mapper */
    sso_moving_mask(sem, W, H, mask);
    sso_camera cam = {159.3, 127.6, 258.6, 258.2, 1000.0};
    double T[16]; sso_synth_pose(3, T);
    sso_point* pts = malloc(sizeof(sso_point) * (size_t)W * H), *vox = malloc(sizeof(sso_point) * (size_t)W * H);
    const int np = sso_backproject(depth, bgr, sem, mask, W, H, &cam, T, 40.0, pts);
    const int nv = sso_voxel_filter(pts, np, 0.1f, vox, W * H);
    /* quad matcher + SGBM + VO on a shifted copy */
    uint8_t *rc = malloc((size_t)W * H), *lp = malloc((size_t)W * H), *rp = malloc((size_t)W * H);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        rc[y * W + x] = gray[y * W + (x + 9) % W]; lp[y * W + x] = gray[((y + H - 1) % H) * W + (x + W - 2) % W]; }
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) rp[y * W + x] = lp[y * W + (x + 9) % W];
    sso_pmatch* qm = malloc(sizeof(sso_pmatch) * 1000);
    const int nq = sso_quad_track(gray, rc, lp, rp, W, H, 1000, qm);
    sso_sgbm_params sp = {0, 32, 7, 4 * 49, 32 * 49, 1, 63, 10, 100, 32};
    int16_t* disp = malloc((size_t)W * H * 2); uint16_t* dep2 = malloc((size_t)W * H * 2);
    sso_sgbm(gray, rc, W, H, &sp, disp);
    {   /* a cost volume narrower than half the SAD window + 1 (34 columns, 32 disparities, window 9: 2 columns): OpenCV 2.4 reads past its cost row here; the contract replicates the border */
        sso_sgbm_params sn = {0, 32, 9, 4 * 81, 32 * 81, 1, 63, 10, 0, 32};
        uint8_t *nl = malloc(34 * 20), *nr = malloc(34 * 20); int16_t* nd = malloc(34 * 20 * 2);
        for (int i = 0; i < 34 * 20; i++) { nl[i] = gray[i]; nr[i] = rc[i]; }
        sso_sgbm(nl, nr, 34, 20, &sn, nd);
        free(nl); free(nr); free(nd);
    }
    sso_disparity_to_depth(disp, W, H, 0.5323, 159.3, 127.6, 258.6, 20, 5, 40, 1000.0, dep2);
    int vo_ok = 0;
    if (nq >= 6) {
        sso_rand_state rs; sso_rand_seed(&rs, 0);
        int32_t* smp = malloc(sizeof(int32_t) * 3 * 50);
        for (int k = 0; k < 50; k++) sso_vo_random_sample(&rs, nq, 3, smp + 3 * k);
        sso_vo_params vp = {258.6, 159.3, 127.6, 0.5323, 2.0, 1, 0};
        double tr[6]; int32_t* inl = malloc(sizeof(int32_t) * nq); int ninl = 0;
        vo_ok = sso_vo_estimate(qm, nq, &vp, smp, 50, tr, inl, &ninl);
        free(smp); free(inl);
    }
    /* PnP: the keypoints of the frame against their own unprojected positions (identity pose) */
    float* img = malloc(sizeof(float) * 2 * (n + 1)), *obj = malloc(sizeof(float) * 3 * (n + 1));
    for (int i = 0; i < n; i++) { img[2 * i] = kps[i].x; img[2 * i + 1] = kps[i].y; sso_project2dTo3d(depth, W, H, &cam, (int)kps[i].x, (int)kps[i].y, obj + 3 * i); }
    double Tp[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; int* pin = malloc(sizeof(int) * (n + 1)); int npin = 0;
    sso_pnp_solve(img, obj, n, &cam, 10, Tp, pin, &npin);
    /* the pipeline entry point, concurrently */
    pthread_t th[4]; int first[4] = {0, 3, 6, 9};
    for (int i = 0; i < 4; i++) pthread_create(&th[i], NULL, pipe_thread, &first[i]);
    for (int i = 0; i < 4; i++) pthread_join(th[i], NULL);
    printf("san_check OK: %d keypoints, %d self-matches, %d points, %d voxels, %d quad matches, vo %d, pnp inliers %d\n", n, nm, np, nv, nq, vo_ok, npin);
    sso_orb_destroy(o);
    free(bgr); free(sem); free(lab); free(gray); free(mask); free(depth); free(kps); free(desc); free(dm); free(pts); free(vox); free(rc); free(lp); free(rp); free(qm);
    free(disp); free(dep2); free(img); free(obj); free(pin);
    return 0;
}
