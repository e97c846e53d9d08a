/* oracle/orb.c -- CPU oracle (TEST INFRASTRUCTURE, see ssm_oracle.h) for K1..K5:
 * the ORB extractor the reference calls at include/orb.h:21-26,44
 * ("(*extractor)(gray, cv::Mat(), kps, desps)") after cv::cvtColor at
 * include/orb.h:39.  ORB_SLAM2::ORBextractor itself is NOT in /root/reference
 * (un-vendored Thirdparty/orbslam_modified, include/orb.h:6); this file
 * restates the published algorithm of raulmur/ORB_SLAM2 src/ORBextractor.cc
 * and of the OpenCV 2.4 routines it calls.  PARITY UNPINNED (no golden data
 * exists upstream).  Every rounding rule below is a CHOSEN CONTRACT shared
 * with the HIP path; see DESIGN.md "ORB contract".
 */
#include "ssm_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

#define EDGE_THRESHOLD 19
#define HALF_PATCH 15
#define PATCH_SIZE 31

static const int8_t default_pattern[1024] = {
#include "orb_pattern.inc"
};

static inline int cv_round_d(double v) { return (int)lrint(v); }       /* cvRound: round-half-even */
static inline int cv_round_f(float v)  { return (int)lrint((double)v); }

/* ---------------- K1: cv::cvtColor(BGR2GRAY), 8u ---------------- */
void sso_bgr2gray(const uint8_t* bgr, int w, int h, int stride, uint8_t* gray)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* s = bgr + (size_t)y * stride;
        uint8_t* d = gray + (size_t)y * w;
        for (int x = 0; x < w; x++)
            d[x] = (uint8_t)((s[3*x] * 1868 + s[3*x+1] * 9617 + s[3*x+2] * 4899 + 8192) >> 14);
    }
}

/* ---------------- K2: cv::resize INTER_LINEAR, 8u (OpenCV 2.4 imgproc/imgwarp.cpp) ----------------
 * scale = 1/((double)dsize/ssize); f = (float)((d+0.5)*scale-0.5); s=floor(f); f-=s;
 * clamp (s<0 -> s=0,f=0 ; s>=ssize-1 -> s=ssize-1,f=0); coef = cvRound((1-f)*2048), cvRound(f*2048) as short.
 * rows:  H[x] = S[sx]*a0 + S[sx+1]*a1   (int)
 * cols:  dst = ( ((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2 ) >> 2           (the 8u VResizeLinear form) */
void sso_resize_tables(int ssize, int dsize, int32_t* ofs, int16_t* coef)
{
    double inv_scale = (double)dsize / ssize;
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
        ofs[d] = s;
        float c0 = 1.f - f, c1 = f;
        coef[2*d]   = (int16_t)cv_round_f(c0 * 2048.f);
        coef[2*d+1] = (int16_t)cv_round_f(c1 * 2048.f);
    }
}

void sso_resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
    int32_t* xofs = (int32_t*)malloc(sizeof(int32_t) * dw);
    int16_t* alpha = (int16_t*)malloc(sizeof(int16_t) * 2 * dw);
    int32_t* yofs = (int32_t*)malloc(sizeof(int32_t) * dh);
    int16_t* beta = (int16_t*)malloc(sizeof(int16_t) * 2 * dh);
    sso_resize_tables(sw, dw, xofs, alpha);
    sso_resize_tables(sh, dh, yofs, beta);
    for (int y = 0; y < dh; y++) {
        int sy0 = yofs[y], sy1 = sy0 + 1 < sh ? sy0 + 1 : sy0;
        const uint8_t* r0 = src + (size_t)sy0 * sw;
        const uint8_t* r1 = src + (size_t)sy1 * sw;
        int b0 = beta[2*y], b1 = beta[2*y+1];
        for (int x = 0; x < dw; x++) {
            int sx0 = xofs[x], sx1 = sx0 + 1 < sw ? sx0 + 1 : sx0;
            int a0 = alpha[2*x], a1 = alpha[2*x+1];
            int h0 = r0[sx0] * a0 + r0[sx1] * a1;
            int h1 = r1[sx0] * a0 + r1[sx1] * a1;
            dst[(size_t)y * dw + x] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
        }
    }
    free(xofs); free(alpha); free(yofs); free(beta);
}

/* ---------------- K5a: cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101), 8u ----------------
 * OpenCV 2.4 8u path: float kernel getGaussianKernel(7,2,CV_32F) -> x256 rounded -> {18,34,49,55,49,34,18};
 * row pass int, column pass int, (v + 2^15) >> 16, saturate. */
static inline int reflect101(int i, int n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; return i; }
static void gaussian_taps(int taps[7])
{
    float cf[7]; float sum = 0.f;
    double scale2x = -0.5 / (2.0 * 2.0);
    for (int i = 0; i < 7; i++) { double x = i - 3.0; float t = (float)exp(scale2x * x * x); cf[i] = t; sum += cf[i]; }
    sum = 1.f / sum;
    for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); taps[i] = cv_round_f(cf[i] * 256.f); }
}
void sso_gaussian7(const uint8_t* src, int w, int h, uint8_t* dst)
{
    int k[7]; gaussian_taps(k);
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int t = -3; t <= 3; t++) s += k[t+3] * src[(size_t)y * w + reflect101(x + t, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int t = -3; t <= 3; t++) s += k[t+3] * tmp[(size_t)reflect101(y + t, h) * w + x];
            s = (s + 32768) >> 16;
            dst[(size_t)y * w + x] = (uint8_t)(s > 255 ? 255 : s);
        }
    free(tmp);
}

/* ---------------- K3: FAST-9/16 score (OpenCV 2.4 features2d/fast_score.cpp, cornerScore<16>) ----------------
 * returns S = max over the 16 arcs of 9 contiguous ring pixels of min(ring - v) [bright] and min(v - ring) [dark].
 * pixel is a FAST corner at threshold t  iff  S > t ; cv::FAST's response is S - 1. */
static const int ring_dx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0,-1,-2,-3,-3,-3,-2,-1};
static const int ring_dy[16] = { 3, 3, 2, 1, 0,-1,-2,-3,-3,-3,-2,-1, 0, 1, 2, 3};
int sso_fast_score(const uint8_t* p, int stride)
{
    int d[25]; int v = p[0];
    for (int k = 0; k < 16; k++) d[k] = v - p[ring_dy[k] * stride + ring_dx[k]];
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    int best_dark = -256, best_bright = -256;      /* dark ring: v - ring > t ; bright ring: ring - v > t */
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; j++) { if (d[k+j] < mn) mn = d[k+j]; if (d[k+j] > mx) mx = d[k+j]; }
        if (mn > best_dark) best_dark = mn;
        if (-mx > best_bright) best_bright = -mx;
    }
    return best_dark > best_bright ? best_dark : best_bright;
}

/* cv::fastAtan2 (OpenCV 2.4 core/mathfuncs.cpp), degrees in [0,360) */
float sso_fast_atan2(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON); c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON); c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* contract sin/cos: reference does (float)cos(angle),(float)sin(angle) via libm on a float radian angle
 * (ORBextractor computeOrbDescriptor).  libm and the device math library are not bit-identical, so the
 * contract is this explicit double-precision routine (Cody-Waite reduction by pi/2, Taylor to x^15/x^16,
 * Horner, no FMA), rounded once to float.  Agrees with correctly-rounded sinf/cosf except on exact ties. */
void sso_sincos(float angle_rad, float* s, float* c)
{
    const double PIO2_HI = 1.57079632673412561417e+00, PIO2_LO = 6.07710050650619224932e-11;
    double x = (double)angle_rad;
    double kd = rint(x * 0.63661977236758134308);
    int k = (int)kd;
    double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    double r2 = r * r;
    double ps = -1.0 / 1307674368000.0;                  /* -1/15! */
    ps = ps * r2 + 1.0 / 6227020800.0;                   /* 1/13! */
    ps = ps * r2 - 1.0 / 39916800.0;                     /* 1/11! */
    ps = ps * r2 + 1.0 / 362880.0;                       /* 1/9! */
    ps = ps * r2 - 1.0 / 5040.0;
    ps = ps * r2 + 1.0 / 120.0;
    ps = ps * r2 - 1.0 / 6.0;
    double sn = r + r * (r2 * ps);
    double pc = 1.0 / 20922789888000.0;                  /* 1/16! */
    pc = pc * r2 - 1.0 / 87178291200.0;                  /* 1/14! */
    pc = pc * r2 + 1.0 / 479001600.0;                    /* 1/12! */
    pc = pc * r2 - 1.0 / 3628800.0;
    pc = pc * r2 + 1.0 / 40320.0;
    pc = pc * r2 - 1.0 / 720.0;
    pc = pc * r2 + 1.0 / 24.0;
    pc = pc * r2 - 0.5;
    double cs = 1.0 + r2 * pc;
    double S, C;
    switch (k & 3) {
        case 0: S = sn; C = cs; break;
        case 1: S = cs; C = -sn; break;
        case 2: S = -sn; C = -cs; break;
        default: S = -cs; C = sn; break;
    }
    *s = (float)S; *c = (float)C;
}

/* ---------------- extractor object ---------------- */
typedef struct { int16_t x, y; int32_t score; int32_t rank; } cand_t;   /* x,y relative to (minBorderX,minBorderY) */

struct sso_orb {
    int nfeatures, nlevels, ini_th, min_th;
    double scale_factor;                 /* ORBextractor::scaleFactor is double, constructed from a float */
    float sf[32], inv_sf[32];
    int feat_per_level[32];
    int umax[HALF_PATCH + 2];
    int8_t pattern[1024];
    /* per-extract scratch, kept for the stage taps */
    int lw[32], lh[32];
    uint8_t* img[32]; uint8_t* blur[32];
    cand_t* cand[32]; int ncand[32];
};

sso_orb* sso_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th)
{
    if (nlevels < 1 || nlevels > 32 || nfeatures < 1) return NULL;
    sso_orb* o = (sso_orb*)calloc(1, sizeof(sso_orb));
    o->nfeatures = nfeatures; o->nlevels = nlevels; o->ini_th = ini_th; o->min_th = min_th;
    o->scale_factor = (double)scale_factor;
    o->sf[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) o->sf[i] = (float)(o->sf[i-1] * o->scale_factor);
    for (int i = 0; i < nlevels; i++) o->inv_sf[i] = 1.0f / o->sf[i];
    float factor = (float)(1.0f / o->scale_factor);
    float nd = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) { o->feat_per_level[l] = cv_round_f(nd); sum += o->feat_per_level[l]; nd *= factor; }
    o->feat_per_level[nlevels-1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    /* umax: ORBextractor ctor */
    int vmax = (int)floor(HALF_PATCH * sqrt(2.0) / 2 + 1), vmin = (int)ceil(HALF_PATCH * sqrt(2.0) / 2);
    const double hp2 = HALF_PATCH * HALF_PATCH;
    for (int v = 0; v <= vmax; ++v) o->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (int v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
        o->umax[v] = v0; ++v0;
    }
    memcpy(o->pattern, default_pattern, 1024);
    return o;
}
static void free_scratch(sso_orb* o)
{
    for (int l = 0; l < 32; l++) { free(o->img[l]); free(o->blur[l]); free(o->cand[l]); o->img[l] = o->blur[l] = NULL; o->cand[l] = NULL; o->ncand[l] = 0; }
}
void sso_orb_destroy(sso_orb* o) { if (o) { free_scratch(o); free(o); } }
void sso_orb_set_pattern(sso_orb* o, const int8_t p[1024]) { memcpy(o->pattern, p, 1024); }
int  sso_orb_capacity(const sso_orb* o) { return o->nfeatures + 3 * o->nlevels; }
int  sso_orb_features_per_level(const sso_orb* o, int l) { return o->feat_per_level[l]; }
int  sso_orb_level_size(const sso_orb* o, int w, int h, int level, int* lw, int* lh)
{
    if (level < 0 || level >= o->nlevels) return -1;
    float s = o->inv_sf[level];
    *lw = cv_round_f((float)w * s); *lh = cv_round_f((float)h * s);
    return 0;
}
const uint8_t* sso_orb_level_image(const sso_orb* o, int l, int blurred) { return blurred ? o->blur[l] : o->img[l]; }
int sso_orb_level_candidates(const sso_orb* o, int l, int32_t* xys, int cap)
{
    int n = o->ncand[l] < cap ? o->ncand[l] : cap;
    for (int i = 0; i < n; i++) {
        xys[3*i] = o->cand[l][i].x + (EDGE_THRESHOLD - 3); xys[3*i+1] = o->cand[l][i].y + (EDGE_THRESHOLD - 3);
        xys[3*i+2] = o->cand[l][i].score;
    }
    return o->ncand[l];
}

/* ---------------- K3: per-cell FAST + NMS (ORBextractor::ComputeKeyPointsOctTree) ----------------
 * cell grid W=30 over [minBorder,maxBorder); each cell sub-image is (wCell+6)x(hCell+6), cv::FAST detects in its
 * inner region (3-px margin), NMS (score strictly greater than the 8 neighbours, neighbours outside the cell's
 * inner region count 0), iniThFAST first, minThFAST if the cell produced nothing.  Output order = cells row-major,
 * raster inside a cell; candidates carry that order as `rank`. */
static int level_candidates(sso_orb* o, int level)
{
    const uint8_t* im = o->img[level]; int w = o->lw[level], h = o->lh[level];
    const int minBX = EDGE_THRESHOLD - 3, minBY = minBX, maxBX = w - EDGE_THRESHOLD + 3, maxBY = h - EDGE_THRESHOLD + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
    if (nCols < 1 || nRows < 1) { o->ncand[level] = 0; return 0; }
    const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
    int cap = ((w + 1) / 2) * ((h + 1) / 2) + 16, n = 0;
    cand_t* out = (cand_t*)malloc(sizeof(cand_t) * cap);
    int* sc = (int*)malloc(sizeof(int) * (size_t)(wCell + 6) * (hCell + 6));
    int rank = 0;
    for (int i = 0; i < nRows; i++) {
        const int iniY = minBY + i * hCell; int maxY = iniY + hCell + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = maxBY;
        for (int j = 0; j < nCols; j++) {
            const int iniX = minBX + j * wCell; int maxX = iniX + wCell + 6;
            if (iniX >= maxBX - 6) continue;
            if (maxX > maxBX) maxX = maxBX;
            const int cw = maxX - iniX, ch = maxY - iniY;
            /* scores over the inner region; 0 elsewhere */
            for (int y = 0; y < ch; y++)
                for (int x = 0; x < cw; x++) {
                    int s = 0;
                    if (x >= 3 && x < cw - 3 && y >= 3 && y < ch - 3)
                        s = sso_fast_score(im + (size_t)(iniY + y) * w + iniX + x, w);
                    sc[y * cw + x] = s;
                }
            for (int pass = 0; pass < 2; pass++) {
                const int th = pass == 0 ? o->ini_th : o->min_th;
                int found = 0;
                for (int y = 3; y < ch - 3; y++)
                    for (int x = 3; x < cw - 3; x++) {
                        int s = sc[y * cw + x];
                        if (s <= th) continue;
                        /* neighbour score as cv::FAST sees it: (S-1) if corner at th else 0; s-1 > that  <=>  below */
                        int keep = 1;
                        for (int dy = -1; dy <= 1 && keep; dy++)
                            for (int dx = -1; dx <= 1; dx++) {
                                if (!dx && !dy) continue;
                                int q = sc[(y + dy) * cw + x + dx];
                                int qs = q > th ? q - 1 : 0;
                                if (!(s - 1 > qs)) { keep = 0; break; }
                            }
                        if (!keep) continue;
                        out[n].x = (int16_t)(x + j * wCell); out[n].y = (int16_t)(y + i * hCell);
                        out[n].score = s - 1; out[n].rank = rank++;
                        n++; found++;
                    }
                if (found) break;
            }
        }
    }
    free(sc);
    o->cand[level] = out; o->ncand[level] = n;
    return n;
}

/* ---------------- K4: ORBextractor::DistributeOctTree, literal (std::list semantics) ----------------
 * Deviation (forced): the reference sorts (size, ExtractorNode*) pairs, i.e. breaks size ties by HEAP ADDRESS,
 * which is not reproducible.  Contract: ties -> the more recently created node is split first. */
typedef struct node {
    struct node *prev, *next;
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int* keys; int nkeys;
    int no_more;
    int seq;                         /* creation order (stands in for the pointer in the tie-break) */
} node_t;
typedef struct { node_t* head; node_t* tail; int size; int seq; } nlist_t;

static node_t* node_new(nlist_t* L, int cap) { node_t* n = (node_t*)calloc(1, sizeof(node_t)); n->keys = (int*)malloc(sizeof(int) * (cap > 0 ? cap : 1)); n->seq = L->seq++; return n; }
static void list_push_front(nlist_t* L, node_t* n) { n->prev = NULL; n->next = L->head; if (L->head) L->head->prev = n; else L->tail = n; L->head = n; L->size++; }
static void list_push_back(nlist_t* L, node_t* n) { n->next = NULL; n->prev = L->tail; if (L->tail) L->tail->next = n; else L->head = n; L->tail = n; L->size++; }
static node_t* list_erase(nlist_t* L, node_t* n) { node_t* nx = n->next; if (n->prev) n->prev->next = n->next; else L->head = n->next; if (n->next) n->next->prev = n->prev; else L->tail = n->prev; L->size--; free(n->keys); free(n); return nx; }

static void divide_node(nlist_t* L, const node_t* p, const cand_t* c, node_t* ch[4])
{
    const int halfX = (int)ceilf((float)(p->URx - p->ULx) / 2), halfY = (int)ceilf((float)(p->BRy - p->ULy) / 2);
    for (int q = 0; q < 4; q++) ch[q] = node_new(L, p->nkeys);
    node_t *n1 = ch[0], *n2 = ch[1], *n3 = ch[2], *n4 = ch[3];
    n1->ULx = p->ULx; n1->ULy = p->ULy; n1->URx = p->ULx + halfX; n1->URy = p->ULy;
    n1->BLx = p->ULx; n1->BLy = p->ULy + halfY; n1->BRx = p->ULx + halfX; n1->BRy = p->ULy + halfY;
    n2->ULx = n1->URx; n2->ULy = n1->URy; n2->URx = p->URx; n2->URy = p->URy; n2->BLx = n1->BRx; n2->BLy = n1->BRy; n2->BRx = p->URx; n2->BRy = p->ULy + halfY;
    n3->ULx = n1->BLx; n3->ULy = n1->BLy; n3->URx = n1->BRx; n3->URy = n1->BRy; n3->BLx = p->BLx; n3->BLy = p->BLy; n3->BRx = n1->BRx; n3->BRy = p->BLy;
    n4->ULx = n3->URx; n4->ULy = n3->URy; n4->URx = n2->BRx; n4->URy = n2->BRy; n4->BLx = n3->BRx; n4->BLy = n3->BRy; n4->BRx = p->BRx; n4->BRy = p->BRy;
    for (int i = 0; i < p->nkeys; i++) {
        const cand_t* k = &c[p->keys[i]]; node_t* d;
        if (k->x < n1->URx) d = (k->y < n1->BRy) ? n1 : n3; else d = (k->y < n1->BRy) ? n2 : n4;
        d->keys[d->nkeys++] = p->keys[i];
    }
    for (int q = 0; q < 4; q++) if (ch[q]->nkeys == 1) ch[q]->no_more = 1;
}
typedef struct { int size; node_t* n; } szptr_t;
static int szptr_cmp(const void* a, const void* b)
{
    const szptr_t* x = (const szptr_t*)a; const szptr_t* y = (const szptr_t*)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->n->seq < y->n->seq ? -1 : (x->n->seq > y->n->seq ? 1 : 0);
}
/* returns number selected; sel[] = candidate indices in lNodes order */
static int distribute_octtree(const cand_t* c, int nc, int minX, int maxX, int minY, int maxY, int N, int* sel)
{
    if (nc == 0) return 0;
    nlist_t L = {0};
    int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (nIni < 1) nIni = 1;                                   /* guard (upstream divides by zero for tall images) */
    const float hX = (float)(maxX - minX) / nIni;
    node_t** ini = (node_t**)malloc(sizeof(node_t*) * nIni);
    for (int i = 0; i < nIni; i++) {
        node_t* n = node_new(&L, nc);
        n->ULx = (int)(hX * (float)i); n->ULy = 0; n->URx = (int)(hX * (float)(i + 1)); n->URy = 0;
        n->BLx = n->ULx; n->BLy = maxY - minY; n->BRx = n->URx; n->BRy = maxY - minY;
        list_push_back(&L, n); ini[i] = n;
    }
    for (int i = 0; i < nc; i++) {
        int b = (int)((float)c[i].x / hX);
        if (b >= nIni) b = nIni - 1;                          /* guard: upstream indexes out of range when x/hX rounds up */
        ini[b]->keys[ini[b]->nkeys++] = i;
    }
    free(ini);
    for (node_t* n = L.head; n; ) {
        if (n->nkeys == 1) { n->no_more = 1; n = n->next; }
        else if (n->nkeys == 0) n = list_erase(&L, n);
        else n = n->next;
    }
    int finish = 0;
    szptr_t* vs = (szptr_t*)malloc(sizeof(szptr_t) * (4 * (size_t)nc + 16)); int nvs = 0;
    szptr_t* vp = (szptr_t*)malloc(sizeof(szptr_t) * (4 * (size_t)nc + 16));
    while (!finish) {
        int prevSize = L.size, nToExpand = 0; nvs = 0;
        for (node_t* n = L.head; n; ) {
            if (n->no_more) { n = n->next; continue; }
            node_t* ch[4]; divide_node(&L, n, c, ch);
            for (int q = 0; q < 4; q++) {
                if (ch[q]->nkeys > 0) {
                    list_push_front(&L, ch[q]);
                    if (ch[q]->nkeys > 1) { nToExpand++; vs[nvs].size = ch[q]->nkeys; vs[nvs].n = ch[q]; nvs++; }
                } else { free(ch[q]->keys); free(ch[q]); }
            }
            n = list_erase(&L, n);
        }
        if (L.size >= N || L.size == prevSize) finish = 1;
        else if (L.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = L.size;
                int nvp = nvs; memcpy(vp, vs, sizeof(szptr_t) * nvs); nvs = 0;
                qsort(vp, nvp, sizeof(szptr_t), szptr_cmp);
                for (int j = nvp - 1; j >= 0; j--) {
                    node_t* ch[4]; divide_node(&L, vp[j].n, c, ch);
                    for (int q = 0; q < 4; q++) {
                        if (ch[q]->nkeys > 0) {
                            list_push_front(&L, ch[q]);
                            if (ch[q]->nkeys > 1) { vs[nvs].size = ch[q]->nkeys; vs[nvs].n = ch[q]; nvs++; }
                        } else { free(ch[q]->keys); free(ch[q]); }
                    }
                    list_erase(&L, vp[j].n);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) finish = 1;
            }
        }
    }
    free(vs); free(vp);
    int ns = 0;
    for (node_t* n = L.head; n; n = n->next) {
        int best = n->keys[0];
        for (int k = 1; k < n->nkeys; k++) {
            /* vKeys order is detection order; keys[] preserves it, so strict > keeps the first maximum */
            if (c[n->keys[k]].score > c[best].score) best = n->keys[k];
        }
        sel[ns++] = best;
    }
    while (L.head) list_erase(&L, L.head);
    return ns;
}

/* ---------------- K5: IC_Angle + steered BRIEF ---------------- */
static float ic_angle(const sso_orb* o, const uint8_t* im, int step, int x, int y)
{
    int m01 = 0, m10 = 0;
    const uint8_t* center = im + (size_t)y * step + x;
    for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH; ++v) {
        int v_sum = 0, d = o->umax[v];
        for (int u = -d; u <= d; ++u) {
            int vp = center[u + v * step], vm = center[u - v * step];
            v_sum += (vp - vm); m10 += u * (vp + vm);
        }
        m01 += v * v_sum;
    }
    return sso_fast_atan2((float)m01, (float)m10);
}
static void orb_descriptor(const sso_orb* o, const uint8_t* im, int step, int x, int y, float angle_deg, uint8_t* desc)
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float angle = angle_deg * factorPI, a, b;
    sso_sincos(angle, &b, &a);
    const uint8_t* center = im + (size_t)y * step + x;
    const int8_t* p = o->pattern;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            const int8_t* q = p + (i * 16 + j * 2) * 2;
            int t[2];
            for (int e = 0; e < 2; e++) {
                float px = (float)q[2*e], py = (float)q[2*e+1];
                int yy = cv_round_f(px * b + py * a), xx = cv_round_f(px * a - py * b);
                t[e] = center[yy * step + xx];
            }
            val |= (t[0] < t[1]) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

int sso_orb_extract(sso_orb* o, const uint8_t* gray, int w, int h, int stride, sso_keypoint* kps, uint8_t* desc)
{
    free_scratch(o);
    /* ComputePyramid: level k resized from level k-1 */
    for (int l = 0; l < o->nlevels; l++) {
        sso_orb_level_size(o, w, h, l, &o->lw[l], &o->lh[l]);
        if (o->lw[l] < 2 * EDGE_THRESHOLD + 8 || o->lh[l] < 2 * EDGE_THRESHOLD + 8) return -1;
        o->img[l] = (uint8_t*)malloc((size_t)o->lw[l] * o->lh[l]);
        if (l == 0) for (int y = 0; y < h; y++) memcpy(o->img[0] + (size_t)y * w, gray + (size_t)y * stride, w);
        else sso_resize_linear_u8(o->img[l-1], o->lw[l-1], o->lh[l-1], o->img[l], o->lw[l], o->lh[l]);
    }
    int total = 0;
    int* sel = (int*)malloc(sizeof(int) * (size_t)(o->nfeatures + 8) * 4);
    for (int l = 0; l < o->nlevels; l++) {
        int lw = o->lw[l], lh = o->lh[l];
        int nc = level_candidates(o, l);
        const int minBX = EDGE_THRESHOLD - 3, minBY = minBX, maxBX = lw - EDGE_THRESHOLD + 3, maxBY = lh - EDGE_THRESHOLD + 3;
        int* s = (int*)malloc(sizeof(int) * (nc + 4));
        int ns = distribute_octtree(o->cand[l], nc, minBX, maxBX, minBY, maxBY, o->feat_per_level[l], s);
        o->blur[l] = (uint8_t*)malloc((size_t)lw * lh);
        sso_gaussian7(o->img[l], lw, lh, o->blur[l]);
        const float scaledPatch = PATCH_SIZE * o->sf[l];
        for (int i = 0; i < ns; i++) {
            const cand_t* c = &o->cand[l][s[i]];
            int x = c->x + minBX, y = c->y + minBY;
            sso_keypoint* k = &kps[total];
            k->angle = ic_angle(o, o->img[l], lw, x, y);
            orb_descriptor(o, o->blur[l], lw, x, y, k->angle, desc + (size_t)total * 32);
            k->x = (float)x; k->y = (float)y;
            if (l != 0) { k->x *= o->sf[l]; k->y *= o->sf[l]; }
            k->size = scaledPatch; k->response = (float)c->score; k->octave = l; k->class_id = -1;
            total++;
        }
        free(s);
    }
    free(sel);
    return total;
}
