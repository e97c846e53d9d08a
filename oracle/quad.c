/* oracle/quad.c -- CPU oracle (TEST INFRASTRUCTURE, see ssm_oracle.h) for the stereo quad-matcher rows a5/a6:
 *   QuadFeatureMatch::detectFeature      /root/reference/src/quadmatcher.cpp:388-417  (GFTT: init() :301-308, q=0.04, minDistance 8)
 *   QuadFeatureMatch::circularMatching   /root/reference/src/quadmatcher.cpp:548-664  (tracking branch: 4x cv::calcOpticalFlowPyrLK :566-576)
 *   QuadFeatureMatch::filteringTracks    /root/reference/src/quadmatcher.cpp:420-503  (in-tree arithmetic, restated literally)
 *   QuadFeatureMatch::matching/caldistance  :41-83, 525-544 and the chain :591-661     (in-tree, restated literally)
 * cv::goodFeaturesToTrack and cv::calcOpticalFlowPyrLK are OpenCV 2.4 (absent): restated from the published algorithm.
 * PARITY UNPINNED.  CHOSEN CONTRACTS where OpenCV's float accumulation order is an implementation detail:
 *   - GFTT: Sobel/box sums are computed in EXACT integers (8-bit input), the min-eigenvalue in float from those integers
 *     with a fixed expression; corners sorted by (value desc, raster index asc) (OpenCV's std::sort leaves ties open);
 *   - LK: the per-window sums A11,A12,A22,b1,b2 are EXACT int64 sums of the integer products OpenCV forms
 *     (it adds them into floats one by one), converted to float once and scaled by 2^-20 like FLT_SCALE.
 */
#include "ssm_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

static inline int refl101(int i, int n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; if (i < 0) i = 0; if (i >= n) i = n - 1; return i; }
static inline int cv_round_f(float v) { return (int)lrint((double)v); }

/* ---------------- cv::cornerMinEigenVal(blockSize 3, ksize 3) restated on integers
 * Sobel: dx = (p[y-1][x+1]-p[y-1][x-1]) + 2(p[y][x+1]-p[y][x-1]) + (p[y+1][x+1]-p[y+1][x-1]), dy likewise, REFLECT_101;
 * cov sums over the 3x3 block (REFLECT_101 on the derivative maps, like boxFilter's border): Sxx, Sxy, Syy (exact ints);
 * scale s = 1/(255 * 4 * 3) per derivative -> eig = ((a + c) - sqrt((a - c)^2 + b^2)) with a = Sxx*s2*0.5, b = Sxy*s2, c = Syy*s2*0.5, s2 = s*s (float) */
void sso_min_eigen_map(const uint8_t* img, int w, int h, float* eig)
{
    int32_t* dx = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * h);
    int32_t* dy = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int ym = refl101(y - 1, h), yp = refl101(y + 1, h), xm = refl101(x - 1, w), xp = refl101(x + 1, w);
            const uint8_t *r0 = img + (size_t)ym * w, *r1 = img + (size_t)y * w, *r2 = img + (size_t)yp * w;
            dx[(size_t)y * w + x] = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
            dy[(size_t)y * w + x] = (r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]);
        }
    const float s = (float)(1.0 / (255.0 * 4.0 * 3.0)), s2 = s * s;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int64_t sxx = 0, sxy = 0, syy = 0;
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) {
                    size_t q = (size_t)refl101(y + j, h) * w + refl101(x + i, w);
                    sxx += (int64_t)dx[q] * dx[q]; sxy += (int64_t)dx[q] * dy[q]; syy += (int64_t)dy[q] * dy[q];
                }
            float a = (float)sxx * s2 * 0.5f, b = (float)sxy * s2, c = (float)syy * s2 * 0.5f;
            float d = a - c;
            eig[(size_t)y * w + x] = (a + c) - sqrtf(d * d + b * b);
        }
    free(dx); free(dy);
}

typedef struct { float v; int32_t idx; } gf_t;
static int gf_cmp(const void* a, const void* b)
{
    const gf_t* x = (const gf_t*)a; const gf_t* y = (const gf_t*)b;
    if (x->v != y->v) return x->v > y->v ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
/* cv::goodFeaturesToTrack(img, maxCorners, quality, minDistance, noArray, blockSize 3, useHarris false).  pts: x,y pairs. */
int sso_gftt(const uint8_t* img, int w, int h, int max_corners, double quality, double min_distance, float* pts)
{
    float* eig = (float*)malloc(sizeof(float) * (size_t)w * h);
    sso_min_eigen_map(img, w, h, eig);
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)w * h; i++) if (eig[i] > mx) mx = eig[i];
    const float thr = (float)((double)mx * quality);               /* threshold(eig, eig, maxVal*qualityLevel, 0, THRESH_TOZERO): keeps v > thr */
    gf_t* c = (gf_t*)malloc(sizeof(gf_t) * (size_t)w * h);
    int nc = 0;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            float v = eig[(size_t)y * w + x];
            if (!(v > thr)) continue;
            float m = 0.f;                                        /* dilate 3x3 of the thresholded map */
            for (int j = -1; j <= 1; j++) for (int i = -1; i <= 1; i++) { float q = eig[(size_t)(y + j) * w + x + i]; q = q > thr ? q : 0.f; if (q > m) m = q; }
            if (v == m) { c[nc].v = v; c[nc].idx = y * w + x; nc++; }
        }
    qsort(c, nc, sizeof(gf_t), gf_cmp);
    int n = 0;
    if (min_distance >= 1) {
        const int cell = cv_round_f((float)min_distance);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        int* head = (int*)malloc(sizeof(int) * (size_t)gw * gh); int* next = (int*)malloc(sizeof(int) * (nc > 0 ? nc : 1));
        for (int i = 0; i < gw * gh; i++) head[i] = -1;
        const float md2 = (float)(min_distance * min_distance);
        for (int i = 0; i < nc; i++) {
            const int y = c[i].idx / w, x = c[i].idx - y * w;
            const int xc = x / cell, yc = y / cell;
            int x1 = xc - 1, y1 = yc - 1, x2 = xc + 1, y2 = yc + 1;
            x1 = x1 < 0 ? 0 : x1; y1 = y1 < 0 ? 0 : y1; x2 = x2 > gw - 1 ? gw - 1 : x2; y2 = y2 > gh - 1 ? gh - 1 : y2;
            int good = 1;
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++)
                    for (int k = head[yy * gw + xx]; k >= 0; k = next[k]) {
                        float ddx = (float)(x - (int)pts[2*k]), ddy = (float)(y - (int)pts[2*k+1]);
                        if (ddx * ddx + ddy * ddy < md2) { good = 0; break; }
                    }
            if (good) {
                next[n] = head[yc * gw + xc]; head[yc * gw + xc] = n;
                pts[2*n] = (float)x; pts[2*n+1] = (float)y; n++;
                if (max_corners > 0 && n == max_corners) break;
            }
        }
        free(head); free(next);
    } else {
        for (int i = 0; i < nc && (max_corners <= 0 || n < max_corners); i++) { pts[2*n] = (float)(c[i].idx % w); pts[2*n+1] = (float)(c[i].idx / w); n++; }
    }
    free(c); free(eig);
    return n;
}

/* ---------------- cv::pyrDown 8u: 5x5 Gaussian [1 4 6 4 1]^2 / 256, (sum + 128) >> 8, REFLECT_101, dst = ((w+1)/2, (h+1)/2) */
void sso_pyrdown(const uint8_t* src, int w, int h, uint8_t* dst)
{
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int s = 0;
            for (int j = -2; j <= 2; j++) {
                const uint8_t* r = src + (size_t)refl101(2 * y + j, h) * w;
                int rs = 0;
                for (int i = -2; i <= 2; i++) rs += k[i + 2] * r[refl101(2 * x + i, w)];
                s += k[j + 2] * rs;
            }
            dst[(size_t)y * dw + x] = (uint8_t)((s + 128) >> 8);
        }
}
/* calcSharrDeriv: dx = 3(p[-1][+1]-p[-1][-1]) + 10(p[0][+1]-p[0][-1]) + 3(p[+1][+1]-p[+1][-1]); dy likewise; REFLECT_101; int16 interleaved (dx,dy) */
void sso_scharr(const uint8_t* src, int w, int h, int16_t* d)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *r0 = src + (size_t)refl101(y - 1, h) * w, *r1 = src + (size_t)y * w, *r2 = src + (size_t)refl101(y + 1, h) * w;
            const int xm = refl101(x - 1, w), xp = refl101(x + 1, w);
            d[2 * ((size_t)y * w + x)]     = (int16_t)(3 * (r0[xp] - r0[xm]) + 10 * (r1[xp] - r1[xm]) + 3 * (r2[xp] - r2[xm]));
            d[2 * ((size_t)y * w + x) + 1] = (int16_t)(3 * (r2[xm] - r0[xm]) + 10 * (r2[x] - r0[x]) + 3 * (r2[xp] - r0[xp]));
        }
}

/* ---------------- cv::calcOpticalFlowPyrLK (OpenCV 2.4 video/lkpyramid.cpp, LKTrackerInvoker) ----------------
 * winSize 11x11, maxLevel 3, criteria COUNT+EPS (200, 0.01), flags OPTFLOW_LK_GET_MIN_EIGENVALS, minEigThreshold 1e-6
 * (the call at quadmatcher.cpp:566-576).  Pixels outside an image read as REFLECT_101 for I/J and as 0 for the
 * derivative (OpenCV pads the pyramid with winSize of REFLECT_101 border and the derivative with constant 0). */
#define LK_WIN 11
#define LK_LEVELS 4
static inline int pix_at(const uint8_t* im, int w, int h, int x, int y) { return im[(size_t)refl101(y, h) * w + refl101(x, w)]; }
static inline int der_at(const int16_t* d, int w, int h, int x, int y, int c) { return (x < 0 || y < 0 || x >= w || y >= h) ? 0 : d[2 * ((size_t)y * w + x) + c]; }
#define DESCALE(v, n) (((v) + (1 << ((n) - 1))) >> (n))

void sso_lk_track(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err,
                  int max_count, double epsilon, double min_eig_threshold)
{
    uint8_t* P[LK_LEVELS]; uint8_t* N[LK_LEVELS]; int16_t* D[LK_LEVELS]; int lw[LK_LEVELS], lh[LK_LEVELS];
    for (int l = 0; l < LK_LEVELS; l++) {
        lw[l] = l ? (lw[l-1] + 1) / 2 : w; lh[l] = l ? (lh[l-1] + 1) / 2 : h;
        P[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l]); N[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l]);
        if (l == 0) { memcpy(P[0], prev, (size_t)w * h); memcpy(N[0], next, (size_t)w * h); }
        else { sso_pyrdown(P[l-1], lw[l-1], lh[l-1], P[l]); sso_pyrdown(N[l-1], lw[l-1], lh[l-1], N[l]); }
        D[l] = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)lw[l] * lh[l]);
        sso_scharr(P[l], lw[l], lh[l], D[l]);
    }
    const float eps2 = (float)(epsilon * epsilon);
    const float FLT_SCALE = 1.f / (1 << 20);
    for (int i = 0; i < n; i++) { status[i] = 1; if (err) err[i] = 0.f; }
    for (int level = LK_LEVELS - 1; level >= 0; level--) {
        const int W = lw[level], H = lh[level];
        for (int pi = 0; pi < n; pi++) {
            float ppx = prev_pts[2*pi] * (float)(1. / (1 << level)), ppy = prev_pts[2*pi+1] * (float)(1. / (1 << level));
            float npx, npy;
            if (level == LK_LEVELS - 1) { npx = ppx; npy = ppy; } else { npx = next_pts[2*pi] * 2.f; npy = next_pts[2*pi+1] * 2.f; }
            next_pts[2*pi] = npx; next_pts[2*pi+1] = npy;
            const float half = (LK_WIN - 1) * 0.5f;
            ppx -= half; ppy -= half;
            int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
            if (ipx < -LK_WIN || ipx >= W || ipy < -LK_WIN || ipy >= H) { if (level == 0) { status[pi] = 0; if (err) err[pi] = 0.f; } continue; }
            float a = ppx - ipx, b = ppy - ipy;
            int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << 14)), iw01 = cv_round_f(a * (1.f - b) * (1 << 14)), iw10 = cv_round_f((1.f - a) * b * (1 << 14));
            int iw11 = (1 << 14) - iw00 - iw01 - iw10;
            int I[LK_WIN * LK_WIN], Ix[LK_WIN * LK_WIN], Iy[LK_WIN * LK_WIN];
            int64_t sA11 = 0, sA12 = 0, sA22 = 0;
            for (int y = 0; y < LK_WIN; y++)
                for (int x = 0; x < LK_WIN; x++) {
                    const int gx = ipx + x, gy = ipy + y;
                    int iv = DESCALE(pix_at(P[level], W, H, gx, gy) * iw00 + pix_at(P[level], W, H, gx + 1, gy) * iw01 +
                                     pix_at(P[level], W, H, gx, gy + 1) * iw10 + pix_at(P[level], W, H, gx + 1, gy + 1) * iw11, 14 - 5);
                    int ix = DESCALE(der_at(D[level], W, H, gx, gy, 0) * iw00 + der_at(D[level], W, H, gx + 1, gy, 0) * iw01 +
                                     der_at(D[level], W, H, gx, gy + 1, 0) * iw10 + der_at(D[level], W, H, gx + 1, gy + 1, 0) * iw11, 14);
                    int iy = DESCALE(der_at(D[level], W, H, gx, gy, 1) * iw00 + der_at(D[level], W, H, gx + 1, gy, 1) * iw01 +
                                     der_at(D[level], W, H, gx, gy + 1, 1) * iw10 + der_at(D[level], W, H, gx + 1, gy + 1, 1) * iw11, 14);
                    I[y * LK_WIN + x] = (int16_t)iv; Ix[y * LK_WIN + x] = (int16_t)ix; Iy[y * LK_WIN + x] = (int16_t)iy;
                    sA11 += (int64_t)ix * ix; sA12 += (int64_t)ix * iy; sA22 += (int64_t)iy * iy;
                }
            float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
            float Dt = A11 * A22 - A12 * A12;
            float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * LK_WIN * LK_WIN);
            if (err) err[pi] = minEig;
            if (minEig < (float)min_eig_threshold || Dt < FLT_EPSILON) { if (level == 0) status[pi] = 0; continue; }
            Dt = 1.f / Dt;
            npx -= half; npy -= half;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < max_count; j++) {
                int inx = (int)floorf(npx), iny = (int)floorf(npy);
                if (inx < -LK_WIN || inx >= W || iny < -LK_WIN || iny >= H) { if (level == 0) status[pi] = 0; break; }
                a = npx - inx; b = npy - iny;
                iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << 14)); iw01 = cv_round_f(a * (1.f - b) * (1 << 14)); iw10 = cv_round_f((1.f - a) * b * (1 << 14));
                iw11 = (1 << 14) - iw00 - iw01 - iw10;
                int64_t sb1 = 0, sb2 = 0;
                for (int y = 0; y < LK_WIN; y++)
                    for (int x = 0; x < LK_WIN; x++) {
                        const int gx = inx + x, gy = iny + y;
                        int diff = DESCALE(pix_at(N[level], W, H, gx, gy) * iw00 + pix_at(N[level], W, H, gx + 1, gy) * iw01 +
                                           pix_at(N[level], W, H, gx, gy + 1) * iw10 + pix_at(N[level], W, H, gx + 1, gy + 1) * iw11, 14 - 5) - I[y * LK_WIN + x];
                        sb1 += (int64_t)diff * Ix[y * LK_WIN + x]; sb2 += (int64_t)diff * Iy[y * LK_WIN + x];
                    }
                float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
                float ddx = (A12 * b2 - A22 * b1) * Dt, ddy = (A12 * b1 - A11 * b2) * Dt;
                npx += ddx; npy += ddy;
                next_pts[2*pi] = npx + half; next_pts[2*pi+1] = npy + half;
                if (ddx * ddx + ddy * ddy <= eps2) break;
                if (j > 0 && fabsf(ddx + pdx) < 0.01f && fabsf(ddy + pdy) < 0.01f) { next_pts[2*pi] -= ddx * 0.5f; next_pts[2*pi+1] -= ddy * 0.5f; break; }
                pdx = ddx; pdy = ddy;
            }
        }
    }
    for (int l = 0; l < LK_LEVELS; l++) { free(P[l]); free(N[l]); free(D[l]); }
}

/* ---------------- QuadFeatureMatch::filteringTracks, quadmatcher.cpp:420-503, literal */
static inline int within_region(float x, float y) { return x < 1280 && x > 0.0f && y < 960 && y > 0.0f; }
int sso_filter_tracks(const float* lc, const float* rc, const float* lp, const float* rp, const float* lp_direct, int n, sso_pmatch* out)
{
    int m = 0;
    for (int i = 0; i < n; i++) {
        const float lcx = lc[2*i], lcy = lc[2*i+1], rcx = rc[2*i], rcy = rc[2*i+1], lpx = lp[2*i], lpy = lp[2*i+1], rpx = rp[2*i], rpy = rp[2*i+1];
        const float ldx = lp_direct[2*i], ldy = lp_direct[2*i+1];
        const int dh1 = cv_round_f(fabsf(lcy - rcy)), dh2 = cv_round_f(fabsf(lpy - rpy));
        const int dh11 = cv_round_f(fabsf(lcy - lpy)), dh22 = cv_round_f(fabsf(rcy - rpy));
        const int dw1 = cv_round_f(fabsf(lcx - lpx)), dw2 = cv_round_f(fabsf(rcx - rpx));
        const int disp1 = cv_round_f(fabsf(lcx - rcx)), disp2 = cv_round_f(fabsf(lpx - rpx));
        const int dfx = cv_round_f(fabsf(lpx - ldx)), dfy = cv_round_f(fabsf(lpy - ldy));
        if (within_region(lcx, lcy) && within_region(lpx, lpy) && within_region(rcx, rcy) && within_region(rpx, rpy) &&
            dh1 < 20 && dh2 < 20 && dh11 < 30 && dh22 < 30 && dw1 < 200 && dw2 < 200 && disp1 > 3 && disp2 > 3 && dfx < 1 && dfy < 1) {
            sso_pmatch* r = &out[m++];
            memset(r, 0, sizeof(*r));
            r->u1c = lcx; r->v1c = lcy; r->u1p = lpx; r->v1p = lpy; r->u2c = rcx; r->v2c = rcy; r->u2p = rpx; r->v2p = rpy;
            r->i1c = r->i1p = r->i2c = r->i2p = i;
        }
    }
    return m;
}
/* the whole tracking branch: GFTT on lc, lc->rc, rc->rp, rp->lp, lc->lp (direct), filter */
int sso_quad_track(const uint8_t* lc, const uint8_t* rc, const uint8_t* lp, const uint8_t* rp, int w, int h, int max_corners, sso_pmatch* out)
{
    float* p_lc = (float*)malloc(sizeof(float) * 2 * (size_t)(max_corners > 0 ? max_corners : w * h));
    const int n = sso_gftt(lc, w, h, max_corners, 0.04, 8.0, p_lc);
    float* p_rc = (float*)malloc(sizeof(float) * 2 * (n + 1)); float* p_rp = (float*)malloc(sizeof(float) * 2 * (n + 1));
    float* p_lp = (float*)malloc(sizeof(float) * 2 * (n + 1)); float* p_ld = (float*)malloc(sizeof(float) * 2 * (n + 1));
    uint8_t* st = (uint8_t*)malloc(n + 1);
    sso_lk_track(lc, rc, w, h, p_lc, n, p_rc, st, NULL, 200, 0.01, 1e-6);
    sso_lk_track(rc, rp, w, h, p_rc, n, p_rp, st, NULL, 200, 0.01, 1e-6);
    sso_lk_track(rp, lp, w, h, p_rp, n, p_lp, st, NULL, 200, 0.01, 1e-6);
    sso_lk_track(lc, lp, w, h, p_lc, n, p_ld, st, NULL, 200, 0.01, 1e-6);
    const int m = sso_filter_tracks(p_lc, p_rc, p_lp, p_rp, p_ld, n, out);
    free(p_lc); free(p_rc); free(p_rp); free(p_lp); free(p_ld); free(st);
    return m;
}

/* ---------------- QuadFeatureMatch::matching (:41-83) with caldistance on binary descriptors (:525-544): windowed brute-force NN.
 * one DMatch per query: trainIdx = first minimum inside the window, -1 when min > distance_threshold; id stays 0 when the
 * window is empty and min_distance 999999999.9f > threshold => -1 as well. */
int sso_window_match(const float* kp1, const uint8_t* d1, int n1, const float* kp2, const uint8_t* d2, int n2,
                     int search_w, int search_h, float distance_threshold, sso_dmatch* out)
{
    for (int i = 0; i < n1; i++) {
        int id = 0; float mind = 999999999.9f;
        for (int j = 0; j < n2; j++) {
            if (fabsf(kp2[2*j] - kp1[2*i]) < (float)search_w && fabsf(kp2[2*j+1] - kp1[2*i+1]) < (float)search_h) {
                int d = 0;
                for (int k = 0; k < 32; k++) d += __builtin_popcount((unsigned)(d1[(size_t)i * 32 + k] ^ d2[(size_t)j * 32 + k]));
                if ((float)d < mind) { mind = (float)d; id = j; }
            }
        }
        if (mind > distance_threshold) id = -1;
        out[i].queryIdx = i; out[i].trainIdx = id; out[i].imgIdx = -1; out[i].distance = mind;     /* DMatch(i, id, minDist): imgIdx defaults to -1 */
    }
    return n1;
}
/* the chain of circularMatching's matching branch (:591-661): lc -> rc -> rp -> lp; index 0 counts as "unmatched" (`> 0`, :622-630) */
int sso_quad_chain(const float* k_lc, const float* k_rc, const float* k_rp, const float* k_lp, int n_lc,
                   const sso_dmatch* m_lrc, const sso_dmatch* m_rcp, const sso_dmatch* m_rlp, sso_pmatch* out)
{
    int m = 0;
    for (int i = 0; i < n_lc; i++) {
        const int id_rc = m_lrc[i].trainIdx; if (!(id_rc > 0)) continue;
        const int id_rp = m_rcp[id_rc].trainIdx; if (!(id_rp > 0)) continue;
        const int id_lp = m_rlp[id_rp].trainIdx; if (!(id_lp > 0)) continue;
        sso_pmatch t; memset(&t, 0, sizeof(t));
        t.u1c = k_lc[2*i]; t.v1c = k_lc[2*i+1]; t.i1c = i; t.u2c = k_rc[2*id_rc]; t.v2c = k_rc[2*id_rc+1]; t.i2c = id_rc;
        t.u2p = k_rp[2*id_rp]; t.v2p = k_rp[2*id_rp+1]; t.i2p = id_rp; t.u1p = k_lp[2*id_lp]; t.v1p = k_lp[2*id_lp+1]; t.i1p = id_lp;
        const int delta_x = (int)fabsf(fabsf(t.u1c - t.u1p) - fabsf(t.u2c - t.u2p));
        const int disparity = (int)fabsf(t.u1c - t.u2c);
        if (delta_x < 2 && disparity > 3) out[m++] = t;
    }
    return m;
}
