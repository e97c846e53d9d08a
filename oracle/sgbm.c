/* sgbm.c -- CPU ORACLE (test infrastructure) for the depth-from-stereo step of the KITTI path:
 * calDisparity_SGBM (/root/reference/src/stereo.cpp:11-30: cv::StereoSGBM, 80 disparities, SAD window 11, P1 = 4*121,
 * P2 = 32*121, uniqueness 10, speckle 100 / 32, disp12MaxDiff 1, preFilterCap 63, single-pass mode) and the
 * disparity -> depth conversion with the ROI gate of FrameReader (/root/reference/src/rgbdframe.cpp:81-116).
 * SURVEY.md s.8(f) rank 2.
 *
 * PARITY UNPINNED: OpenCV is absent, so cv::StereoSGBM::operator() of OpenCV 2.4 is restated from its published source
 * (modules/calib3d/src/stereosgbm.cpp: calcPixelCostBT, computeDisparitySGBM, then medianBlur 3x3 and filterSpeckles),
 * written here in the same row-by-row form with the same ring buffers.  CHOSEN CONTRACT, including two behaviours of the
 * 2.4 code that later OpenCV versions changed:
 *  - the aggregated cost C of column x = 0 (first valid column) is never updated after row 0;
 *  - rows whose window would need image rows >= height (y > height - 1 - SADWindowSize/2) keep the cost of the last
 *    row that could be updated.
 * Everything is integer arithmetic (costs are int16), so the GPU kernels are bit-exact against this file.
 */
#include "ssm_oracle.h"
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define DISP_SHIFT 4
#define DISP_SCALE (1 << DISP_SHIFT)
typedef short CostType;
typedef short DispType;
#define MAX_COST SHRT_MAX
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline CostType sat16(int v) { return (CostType)(v > SHRT_MAX ? SHRT_MAX : v < SHRT_MIN ? SHRT_MIN : v); }

/* Birchfield-Tomasi pixel cost of row y on (x-Sobel clipped to +-ftzero, raw intensity / 4): calcPixelCostBT, cn == 1 */
static void pixel_cost_bt(const uint8_t* img1, const uint8_t* img2, int width, int height, int y, int minD, int maxD,
                          CostType* cost, uint8_t* buffer, const uint8_t* tab /* already offset by TAB_OFS */)
{
    const int minX1 = imax(maxD, 0), maxX1 = width + imin(minD, 0);
    const int minX2 = imax(minX1 - maxD, 0), maxX2 = imin(maxX1 - minD, width);
    const int D = maxD - minD, width1 = maxX1 - minX1, width2 = maxX2 - minX2;
    const uint8_t *row1 = img1 + (size_t)y * width, *row2 = img2 + (size_t)y * width;
    uint8_t *prow1 = buffer + width2 * 2, *prow2 = prow1 + width * 2;
    for (int c = 0; c < 2; c++)
        prow1[width * c] = prow1[width * c + width - 1] = prow2[width * c] = prow2[width * c + width - 1] = tab[0];
    const int n1 = y > 0 ? -width : 0, s1 = y < height - 1 ? width : 0;
    for (int x = 1; x < width - 1; x++) {
        prow1[x] = tab[(row1[x + 1] - row1[x - 1]) * 2 + row1[x + n1 + 1] - row1[x + n1 - 1] + row1[x + s1 + 1] - row1[x + s1 - 1]];
        prow2[width - 1 - x] = tab[(row2[x + 1] - row2[x - 1]) * 2 + row2[x + n1 + 1] - row2[x + n1 - 1] + row2[x + s1 + 1] - row2[x + s1 - 1]];
        prow1[x + width] = row1[x];
        prow2[width - 1 - x + width] = row2[x];
    }
    memset(cost, 0, sizeof(CostType) * (size_t)width1 * D);
    buffer -= minX2;
    cost -= minX1 * D + minD;
    for (int c = 0; c < 2; c++, prow1 += width, prow2 += width) {
        const int diff_scale = c < 1 ? 0 : 2;
        for (int x = minX2; x < maxX2; x++) {
            const int v = prow2[x];
            const int vl = x > 0 ? (v + prow2[x - 1]) / 2 : v, vr = x < width - 1 ? (v + prow2[x + 1]) / 2 : v;
            buffer[x] = (uint8_t)imin(imin(vl, vr), v);
            buffer[x + width2] = (uint8_t)imax(imax(vl, vr), v);
        }
        for (int x = minX1; x < maxX1; x++) {
            const int u = prow1[x];
            const int ul = x > 0 ? (u + prow1[x - 1]) / 2 : u, ur = x < width - 1 ? (u + prow1[x + 1]) / 2 : u;
            const int u0 = imin(imin(ul, ur), u), u1 = imax(imax(ul, ur), u);
            for (int d = minD; d < maxD; d++) {
                const int v = prow2[width - x - 1 + d], v0 = buffer[width - x - 1 + d], v1 = buffer[width - x - 1 + d + width2];
                const int c0 = imax(imax(0, u - v1), v0 - u), c1 = imax(imax(0, v - u1), u0 - v);
                cost[x * D + d] = (CostType)(cost[x * D + d] + (imin(c0, c1) >> diff_scale));
            }
        }
    }
}

/* computeDisparitySGBM, fullDP == false.  disp: int16, fixed point with 4 fractional bits, (minD - 1) * 16 = invalid */
int sso_sgbm_raw(const uint8_t* img1, const uint8_t* img2, int width, int height, const sso_sgbm_params* p, int16_t* disp)
{
    const int minD = p->minDisparity, maxD = minD + p->numberOfDisparities;
    const int SW = p->SADWindowSize > 0 ? p->SADWindowSize : 5;
    const int ftzero = imax(p->preFilterCap, 15) | 1;
    const int uniquenessRatio = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    const int disp12MaxDiff = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;
    const int P1 = p->P1 > 0 ? p->P1 : 2, P2 = imax(p->P2 > 0 ? p->P2 : 5, P1 + 1);
    const int minX1 = imax(maxD, 0), maxX1 = width + imin(minD, 0);
    const int D = maxD - minD, width1 = maxX1 - minX1;
    const int INVALID_DISP = minD - 1, INVALID_DISP_SCALED = INVALID_DISP * DISP_SCALE;
    const int SW2 = SW / 2, SH2 = SW / 2;
    enum { TAB_OFS = 256 * 4, TAB_SIZE = 256 + TAB_OFS * 2, NR2 = 8, NLR = 2, LrBorder = NLR - 1 };
    if (D % 16 != 0 || D <= 0 || width <= 2 || height <= 0) return -1;
    uint8_t clipTab[TAB_SIZE];
    for (int k = 0; k < TAB_SIZE; k++) clipTab[k] = (uint8_t)(imin(imax(k - TAB_OFS, -ftzero), ftzero) + ftzero);
    if (minX1 >= maxX1) { for (size_t i = 0; i < (size_t)width * height; i++) disp[i] = (int16_t)INVALID_DISP_SCALED; return 0; }
    const int D2 = D + 16, NRD2 = NR2 * D2;
    const size_t costBufSize = (size_t)width1 * D;
    const size_t minLrSize = (size_t)(width1 + LrBorder * 2) * NR2, LrSize = minLrSize * D2;
    const int hsumBufNRows = SH2 * 2 + 2;
    CostType* Cbuf = (CostType*)calloc(costBufSize, sizeof(CostType));
    CostType* Sbuf = (CostType*)calloc(costBufSize, sizeof(CostType));
    CostType* hsumBuf = (CostType*)calloc(costBufSize * hsumBufNRows, sizeof(CostType));
    CostType* pixDiff = (CostType*)calloc(costBufSize, sizeof(CostType));
    CostType* LrMem = (CostType*)calloc((LrSize + minLrSize) * NLR + 64, sizeof(CostType));
    CostType* disp2cost = (CostType*)calloc((size_t)width, sizeof(CostType));
    DispType* disp2ptr = (DispType*)calloc((size_t)width, sizeof(DispType));
    uint8_t* tempBuf = (uint8_t*)calloc((size_t)width * 8 + 64, 1);
    for (size_t k = 0; k < costBufSize; k++) Cbuf[k] = (CostType)P2;        /* P2 is carried inside C: saves an add per L */
    CostType *Lr[NLR], *minLr[NLR];
    for (int k = 0; k < NLR; k++) {
        Lr[k] = LrMem + LrSize * k + NRD2 * LrBorder + 8;
        minLr[k] = LrMem + LrSize * NLR + 16 + minLrSize * k + NR2 * LrBorder;
    }
    for (int y = 0; y < height; y++) {
        DispType* disp1ptr = disp + (size_t)y * width;
        CostType *C = Cbuf, *S = Sbuf;
        const int dy1 = y == 0 ? 0 : y + SH2, dy2 = y == 0 ? SH2 : dy1;
        for (int k = dy1; k <= dy2; k++) {
            CostType* hsumAdd = hsumBuf + (size_t)(imin(k, height - 1) % hsumBufNRows) * costBufSize;
            if (k < height) {
                pixel_cost_bt(img1, img2, width, height, k, minD, maxD, pixDiff, tempBuf, clipTab + TAB_OFS);
                memset(hsumAdd, 0, sizeof(CostType) * D);
                /* OpenCV 2.4 reads pixDiff[x + d] here for columns 0 .. SW2 without a bound: when the cost volume is narrower than half the window + 1
                   (width1 <= SW2) that is a read past the row (undefined; found by tests/test_gpu_fuzz.py + ASan, round 4).  The contract replicates the last column,
                   as the running sums below do with imin(x + SW2 * D, (width1 - 1) * D). */
                for (int x = 0; x <= SW2 * D; x += D) {
                    const int scale = x == 0 ? SW2 + 1 : 1;
                    const CostType* pd = pixDiff + imin(x, (width1 - 1) * D);
                    for (int d = 0; d < D; d++) hsumAdd[d] = (CostType)(hsumAdd[d] + pd[d] * scale);
                }
                if (y > 0) {
                    const CostType* hsumSub = hsumBuf + (size_t)(imax(y - SH2 - 1, 0) % hsumBufNRows) * costBufSize;
                    for (int x = D; x < width1 * D; x += D) {
                        const CostType* pixAdd = pixDiff + imin(x + SW2 * D, (width1 - 1) * D);
                        const CostType* pixSub = pixDiff + imax(x - (SW2 + 1) * D, 0);
                        for (int d = 0; d < D; d++) {
                            const int hv = hsumAdd[x + d] = (CostType)(hsumAdd[x - D + d] + pixAdd[d] - pixSub[d]);
                            C[x + d] = (CostType)(C[x + d] + hv - hsumSub[x + d]);
                        }
                    }
                } else {
                    for (int x = D; x < width1 * D; x += D) {
                        const CostType* pixAdd = pixDiff + imin(x + SW2 * D, (width1 - 1) * D);
                        const CostType* pixSub = pixDiff + imax(x - (SW2 + 1) * D, 0);
                        for (int d = 0; d < D; d++) hsumAdd[x + d] = (CostType)(hsumAdd[x - D + d] + pixAdd[d] - pixSub[d]);
                    }
                }
            }
            if (y == 0) {
                const int scale = k == 0 ? SH2 + 1 : 1;
                for (int x = 0; x < width1 * D; x++) C[x] = (CostType)(C[x] + hsumAdd[x] * scale);
            }
        }
        memset(S, 0, sizeof(CostType) * costBufSize);
        /* clear the left and right borders of the current line */
        memset(Lr[0] - NRD2 * LrBorder - 8, 0, sizeof(CostType) * NRD2 * LrBorder);
        memset(Lr[0] + width1 * NRD2 - 8, 0, sizeof(CostType) * NRD2 * LrBorder);
        memset(minLr[0] - NR2 * LrBorder, 0, sizeof(CostType) * NR2 * LrBorder);
        memset(minLr[0] + width1 * NR2, 0, sizeof(CostType) * NR2 * LrBorder);
        /* L_r(p, d) = C(p, d) + min(L_r(p-r, d), L_r(p-r, d-1) + P1, L_r(p-r, d+1) + P1, min_k L_r(p-r, k) + P2) - min_k L_r(p-r, k)
         * for r = (-1, 0), (-1, -1), (0, -1), (1, -1), all at once, left to right */
        for (int x = 0; x < width1; x++) {
            const int xm = x * NR2, xd = xm * D2;
            const int delta0 = minLr[0][xm - NR2] + P2, delta1 = minLr[1][xm - NR2 + 1] + P2;
            const int delta2 = minLr[1][xm + 2] + P2, delta3 = minLr[1][xm + NR2 + 3] + P2;
            CostType* Lr_p0 = Lr[0] + xd - NRD2;
            CostType* Lr_p1 = Lr[1] + xd - NRD2 + D2;
            CostType* Lr_p2 = Lr[1] + xd + D2 * 2;
            CostType* Lr_p3 = Lr[1] + xd + NRD2 + D2 * 3;
            Lr_p0[-1] = Lr_p0[D] = Lr_p1[-1] = Lr_p1[D] = Lr_p2[-1] = Lr_p2[D] = Lr_p3[-1] = Lr_p3[D] = MAX_COST;
            CostType* Lr_p = Lr[0] + xd;
            const CostType* Cp = C + (size_t)x * D;
            CostType* Sp = S + (size_t)x * D;
            int minL0 = MAX_COST, minL1 = MAX_COST, minL2 = MAX_COST, minL3 = MAX_COST;
            for (int d = 0; d < D; d++) {
                const int Cpd = Cp[d];
                const int L0 = Cpd + imin((int)Lr_p0[d], imin(Lr_p0[d - 1] + P1, imin(Lr_p0[d + 1] + P1, delta0))) - delta0;
                const int L1 = Cpd + imin((int)Lr_p1[d], imin(Lr_p1[d - 1] + P1, imin(Lr_p1[d + 1] + P1, delta1))) - delta1;
                const int L2 = Cpd + imin((int)Lr_p2[d], imin(Lr_p2[d - 1] + P1, imin(Lr_p2[d + 1] + P1, delta2))) - delta2;
                const int L3 = Cpd + imin((int)Lr_p3[d], imin(Lr_p3[d - 1] + P1, imin(Lr_p3[d + 1] + P1, delta3))) - delta3;
                Lr_p[d] = (CostType)L0; minL0 = imin(minL0, L0);
                Lr_p[d + D2] = (CostType)L1; minL1 = imin(minL1, L1);
                Lr_p[d + D2 * 2] = (CostType)L2; minL2 = imin(minL2, L2);
                Lr_p[d + D2 * 3] = (CostType)L3; minL3 = imin(minL3, L3);
                Sp[d] = sat16(Sp[d] + L0 + L1 + L2 + L3);
            }
            minLr[0][xm] = (CostType)minL0; minLr[0][xm + 1] = (CostType)minL1; minLr[0][xm + 2] = (CostType)minL2; minLr[0][xm + 3] = (CostType)minL3;
        }
        /* disparity selection, with the fifth direction r = (+1, 0) computed on the way back */
        for (int x = 0; x < width; x++) { disp1ptr[x] = disp2ptr[x] = (DispType)INVALID_DISP_SCALED; disp2cost[x] = MAX_COST; }
        for (int x = width1 - 1; x >= 0; x--) {
            CostType* Sp = S + (size_t)x * D;
            int minS = MAX_COST, bestDisp = -1;
            {
                const int xm = x * NR2, xd = xm * D2;
                int minL0 = MAX_COST;
                const int delta0 = minLr[0][xm + NR2] + P2;
                CostType* Lr_p0 = Lr[0] + xd + NRD2;
                Lr_p0[-1] = Lr_p0[D] = MAX_COST;
                CostType* Lr_p = Lr[0] + xd;
                const CostType* Cp = C + (size_t)x * D;
                for (int d = 0; d < D; d++) {
                    const int L0 = Cp[d] + imin((int)Lr_p0[d], imin(Lr_p0[d - 1] + P1, imin(Lr_p0[d + 1] + P1, delta0))) - delta0;
                    Lr_p[d] = (CostType)L0;
                    minL0 = imin(minL0, L0);
                    const int Sval = Sp[d] = sat16(Sp[d] + L0);
                    if (Sval < minS) { minS = Sval; bestDisp = d; }
                }
                minLr[0][xm] = (CostType)minL0;
            }
            int d;
            for (d = 0; d < D; d++) if (Sp[d] * (100 - uniquenessRatio) < minS * 100 && abs(bestDisp - d) > 1) break;
            if (d < D) continue;
            d = bestDisp;
            const int x2 = x + minX1 - d - minD;
            if (disp2cost[x2] > minS) { disp2cost[x2] = (CostType)minS; disp2ptr[x2] = (DispType)(d + minD); }
            if (0 < d && d < D - 1) {
                const int denom2 = imax(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
                d = d * DISP_SCALE + ((Sp[d - 1] - Sp[d + 1]) * DISP_SCALE + denom2) / (denom2 * 2);
            } else d *= DISP_SCALE;
            disp1ptr[x + minX1] = (DispType)(d + minD * DISP_SCALE);
        }
        for (int x = minX1; x < maxX1; x++) {
            /* round the disparity towards -inf and +inf and check whether either is consistent with the right-image disparity */
            const int d1 = disp1ptr[x];
            if (d1 == INVALID_DISP_SCALED) continue;
            const int _d = d1 >> DISP_SHIFT, d_ = (d1 + DISP_SCALE - 1) >> DISP_SHIFT;
            const int _x = x - _d, x_ = x - d_;
            if (0 <= _x && _x < width && disp2ptr[_x] >= minD && abs(disp2ptr[_x] - _d) > disp12MaxDiff &&
                0 <= x_ && x_ < width && disp2ptr[x_] >= minD && abs(disp2ptr[x_] - d_) > disp12MaxDiff)
                disp1ptr[x] = (DispType)INVALID_DISP_SCALED;
        }
        { CostType* t = Lr[0]; Lr[0] = Lr[1]; Lr[1] = t; t = minLr[0]; minLr[0] = minLr[1]; minLr[1] = t; }
    }
    free(Cbuf); free(Sbuf); free(hsumBuf); free(pixDiff); free(LrMem); free(disp2cost); free(disp2ptr); free(tempBuf);
    return 0;
}

/* cv::medianBlur(ksize 3) on int16, BORDER_REPLICATE */
void sso_median3_s16(const int16_t* src, int w, int h, int16_t* dst)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int16_t v[9]; int n = 0;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    const int yy = imin(imax(y + dy, 0), h - 1), xx = imin(imax(x + dx, 0), w - 1);
                    v[n++] = src[(size_t)yy * w + xx];
                }
            for (int i = 1; i < 9; i++) { const int16_t t = v[i]; int j = i - 1; while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; j--; } v[j + 1] = t; }
            dst[(size_t)y * w + x] = v[4];
        }
}
/* cv::filterSpeckles: 4-connected components of pixels != newVal whose neighbours differ by <= maxDiff; components of at most
 * maxSpeckleSize pixels become newVal */
void sso_filter_speckles(int16_t* img, int w, int h, int newVal, int maxSpeckleSize, int maxDiff)
{
    const size_t n = (size_t)w * h;
    int32_t* label = (int32_t*)calloc(n, sizeof(int32_t));
    int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * n);
    uint8_t* small = (uint8_t*)calloc(n + 2, 1);
    int cur = 0;
    for (size_t i = 0; i < n; i++) {
        if (img[i] == newVal) continue;
        if (label[i]) { if (small[label[i]]) img[i] = (int16_t)newVal; continue; }
        cur++;
        int sp = 0, count = 0;
        stack[sp++] = (int32_t)i; label[i] = cur;
        while (sp > 0) {
            const int32_t p = stack[--sp];
            const int px = p % w, py = p / w, dp = img[p];
            count++;
            if (px < w - 1 && !label[p + 1] && img[p + 1] != newVal && abs(dp - img[p + 1]) <= maxDiff) { label[p + 1] = cur; stack[sp++] = p + 1; }
            if (px > 0 && !label[p - 1] && img[p - 1] != newVal && abs(dp - img[p - 1]) <= maxDiff) { label[p - 1] = cur; stack[sp++] = p - 1; }
            if (py < h - 1 && !label[p + w] && img[p + w] != newVal && abs(dp - img[p + w]) <= maxDiff) { label[p + w] = cur; stack[sp++] = p + w; }
            if (py > 0 && !label[p - w] && img[p - w] != newVal && abs(dp - img[p - w]) <= maxDiff) { label[p - w] = cur; stack[sp++] = p - w; }
        }
        if (count <= maxSpeckleSize) { small[cur] = 1; img[i] = (int16_t)newVal; }
    }
    free(label); free(stack); free(small);
}
/* cv::StereoSGBM::operator(): raw SGBM, medianBlur 3, filterSpeckles */
int sso_sgbm(const uint8_t* left, const uint8_t* right, int w, int h, const sso_sgbm_params* p, int16_t* disp)
{
    int16_t* raw = (int16_t*)malloc(sizeof(int16_t) * (size_t)w * h);
    const int rc = sso_sgbm_raw(left, right, w, h, p, raw);
    if (rc) { free(raw); return rc; }
    sso_median3_s16(raw, w, h, disp);
    free(raw);
    if (p->speckleWindowSize > 0) sso_filter_speckles(disp, w, h, (p->minDisparity - 1) * DISP_SCALE, p->speckleWindowSize, DISP_SCALE * p->speckleRange);
    return 0;
}
/* FrameReader (rgbdframe.cpp:81-116): disparity (x16) -> depth in `scale` units with the 3-D ROI gate; the minimum
 * disparity value of the image marks "no measurement" */
void sso_disparity_to_depth(const int16_t* disp, int w, int h, double baseline, double cu, double cv, double f,
                            double roix, double roiy, double roiz, double scale, uint16_t* depth)
{
    double minDisparity = 3.4028234663852886e+38;           /* FLT_MAX, then cv::minMaxIdx */
    for (size_t i = 0; i < (size_t)w * h; i++) if ((double)disp[i] < minDisparity) minDisparity = (double)disp[i];
    const double eps = 1.1920928955078125e-07;                /* FLT_EPSILON */
    for (int v = 0; v < h; v++)
        for (int u = 0; u < w; u++) {
            const int16_t d = disp[(size_t)v * w + u];
            uint16_t out = 0;
            if ((d < 0 ? -(double)d : (double)d) > eps) {
                const double pw = baseline / (1.0 * (double)d);
                const double px = (((double)u - cu) * pw) * 16.0, py = (((double)v - cv) * pw) * 16.0, pz = (f * pw) * 16.0;
                const double dm = (double)d - minDisparity;
                if (!((dm < 0 ? -dm : dm) <= eps)) {
                    if ((px < 0 ? -px : px) < roix && (py < 0 ? -py : py) < roiy && (pz < 0 ? -pz : pz) < roiz && pz > 0) out = (uint16_t)(pz * scale);
                }
            }
            depth[(size_t)v * w + u] = out;
        }
}
