/* oracle/synth.c -- CPU oracle side (TEST INFRASTRUCTURE, see ssm_oracle.h) of the synthetic 640x480 RGB-D +
 * 12-class stream of BASELINE.json configs[1] / SURVEY.md s.8(d) C2.  Integer-only so that the device generator in
 * semantic_slam_mapping_amd/csrc/synth.hip reproduces it bit for bit (tests/test_synth.py).
 *
 *   world coords      wx = u + 2*frame_id (camera pans 2 px/frame so consecutive frames match), wy = v
 *   texture           16-px value-noise base (64..191 gray) + 4 random axis-aligned rectangles per 32x32 world block
 *                     (origin in the block, 4..24 px sides, uniform BGR), later rectangle wins; +-1 per-frame noise
 *   depth (u16)       1000 + 600*sin(u/53)*cos(v/41) + 200*(frame_id mod 16)/16 counts at camera.scale 1000, with
 *                     sin/cos replaced by the parabolic integer sine below; 51/1024 (5 %) of pixels zeroed by hash (bits 8..17 of the per-pixel noise word)
 *   labels            one of 12 classes per 32x32 world block by hash; semantic image = palette BGR
 *   pose              T_f_w = translation (0.01*frame_id, 0, 0)
 */
#include "ssm_oracle.h"
#include <stdlib.h>

static inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t hash3(uint64_t seed, uint64_t tag, int64_t a, int64_t b)
{
    return mix64(seed ^ mix64(tag ^ mix64((uint64_t)a * 0x9E3779B1ULL ^ mix64((uint64_t)b))));
}
/* parabolic sine, Q15 in/out: phase 0..65535 = one period, result in [-32768, 32768] */
static inline int32_t isin_q15(uint32_t p)
{
    int32_t x = (int32_t)(p & 0x7FFF);
    int32_t y = (x * (32768 - x)) >> 13;
    return (p & 0x8000) ? -y : y;
}

typedef struct { int64_t x0, y0; int rw, rh, B, G, R; } srect_t;

void sso_synth_frame(uint64_t seed, int frame_id, int w, int h, uint8_t* bgr, uint16_t* depth, uint8_t* sem, uint8_t* lab)
{
    /* per-frame caches of the block / lattice hashes (pure speed-up; the per-pixel definition is in the header comment) */
    const int64_t wx0 = 2 * (int64_t)frame_id;
    const int64_t lx0 = wx0 >> 4, bx0 = (wx0 >> 5) - 1;
    const int nlx = (int)(((wx0 + w - 1) >> 4) - lx0) + 2, nly = ((h - 1) >> 4) + 2;
    const int nbx = (int)(((wx0 + w - 1) >> 5) - bx0) + 1, nby = ((h - 1) >> 5) + 2;   /* by from -1 */
    uint8_t* lat = (uint8_t*)malloc((size_t)nlx * nly);
    srect_t* rects = (srect_t*)malloc(sizeof(srect_t) * (size_t)nbx * nby * 4);
    uint8_t* cls = (uint8_t*)malloc((size_t)nbx * nby);
    for (int j = 0; j < nly; j++) for (int i = 0; i < nlx; i++) lat[j * nlx + i] = (uint8_t)(hash3(seed, 1, lx0 + i, j) & 255);
    for (int j = 0; j < nby; j++)
        for (int i = 0; i < nbx; i++) {
            int64_t bx = bx0 + i, by = (int64_t)j - 1;
            for (int k = 0; k < 4; k++) {
                uint64_t H = hash3(seed, 2 + (uint64_t)k, bx, by);
                srect_t* r = &rects[(j * nbx + i) * 4 + k];
                r->x0 = bx * 32 + (int64_t)(H & 31); r->y0 = by * 32 + (int64_t)((H >> 5) & 31);      /* (not << 5: by is -1 in the first block row) */
                r->rw = 4 + (int)((H >> 10) % 21); r->rh = 4 + (int)((H >> 20) % 21);
                r->B = (int)((H >> 32) & 255); r->G = (int)((H >> 40) & 255); r->R = (int)((H >> 48) & 255);
            }
            cls[j * nbx + i] = (uint8_t)(hash3(seed, 9, bx, by) % 12);
        }
    for (int v = 0; v < h; v++)
        for (int u = 0; u < w; u++) {
            const int64_t wx = (int64_t)u + wx0, wy = v;
            size_t pix = (size_t)v * w + u;
            /* value-noise base */
            int li = (int)((wx >> 4) - lx0), lj = (int)(wy >> 4); int fx = (int)(wx & 15), fy = (int)(wy & 15);
            int a = lat[lj * nlx + li], b = lat[lj * nlx + li + 1], c = lat[(lj + 1) * nlx + li], d = lat[(lj + 1) * nlx + li + 1];
            int vn = ((a * (16 - fx) + b * fx) * (16 - fy) + (c * (16 - fx) + d * fx) * fy + 128) >> 8;
            int B = 64 + (vn >> 1), G = B, R = B;
            /* rectangles of the 2x2 blocks that can reach this pixel: (by-1,bx-1),(by-1,bx),(by,bx-1),(by,bx), k ascending; last hit wins */
            int bi = (int)((wx >> 5) - bx0), bj = (int)(wy >> 5) + 1;
            for (int dy = -1; dy <= 0; dy++)
                for (int dx = -1; dx <= 0; dx++)
                    for (int k = 0; k < 4; k++) {
                        const srect_t* r = &rects[((bj + dy) * nbx + bi + dx) * 4 + k];
                        if (wx >= r->x0 && wx < r->x0 + r->rw && wy >= r->y0 && wy < r->y0 + r->rh) { B = r->B; G = r->G; R = r->R; }
                    }
            /* per-frame +-1 noise */
            uint64_t N = hash3(seed, 7, (int64_t)frame_id, (int64_t)pix);
            int nb = (int)(N & 3), ng = (int)((N >> 2) & 3), nr = (int)((N >> 4) & 3);
            B += (nb == 0) ? -1 : (nb == 1 ? 1 : 0); G += (ng == 0) ? -1 : (ng == 1 ? 1 : 0); R += (nr == 0) ? -1 : (nr == 1 ? 1 : 0);
            B = B < 0 ? 0 : (B > 255 ? 255 : B); G = G < 0 ? 0 : (G > 255 ? 255 : G); R = R < 0 ? 0 : (R > 255 ? 255 : R);
            bgr[3*pix] = (uint8_t)B; bgr[3*pix+1] = (uint8_t)G; bgr[3*pix+2] = (uint8_t)R;
            /* depth; holes share the noise hash word (bits 8..17) */
            int32_t s = isin_q15(((uint32_t)u * 197u) & 0xFFFF), cc = isin_q15(((uint32_t)v * 254u + 16384u) & 0xFFFF);
            int64_t prod = (int64_t)600 * s * cc;
            int dd = 1000 + (int)((prod + ((int64_t)1 << 29)) >> 30) + (200 * (frame_id & 15)) / 16;
            if (((N >> 8) & 1023) < 51) dd = 0;
            depth[pix] = (uint16_t)dd;
            /* labels */
            int cl = cls[bj * nbx + bi];
            if (lab) lab[pix] = (uint8_t)cl;
            sem[3*pix] = sso_palette_bgr[cl][0]; sem[3*pix+1] = sso_palette_bgr[cl][1]; sem[3*pix+2] = sso_palette_bgr[cl][2];
        }
    free(lat); free(rects); free(cls);
}

void sso_synth_pose(int frame_id, double T[16])
{
    for (int i = 0; i < 16; i++) T[i] = 0.0;
    T[0] = T[5] = T[10] = T[15] = 1.0;
    T[12] = 0.01 * (double)frame_id;
}
