/* oracle/mapper.c -- CPU oracle (TEST INFRASTRUCTURE, see ssm_oracle.h) for K10, K11, K12:
 *   RGBDFrame::project2dTo3d     /root/reference/include/rgbdframe.h:63-75
 *   Mapper::semantic_motion_fuse /root/reference/src/mapper.cpp:189-216
 *   Mapper::generatePointCloud   /root/reference/src/mapper.cpp:12-94
 *   pcl::VoxelGrid in Mapper::viewer, /root/reference/src/mapper.cpp:106-107,154-155 (PCL 1.7, absent: restated)
 */
#include "ssm_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* 12-class palette, BGR, SegNet driving_webdemo id order (src/mapper.cpp:42-54 comments; /root/reference/000000.png) */
const uint8_t sso_palette_bgr[12][3] = {
    {128,128,128}, /* 0 sky          */ {0,0,128},     /* 1 building */ {128,192,192}, /* 2 pole     */
    {0,69,255},    /* 3 road marking */ {128,64,128},  /* 4 road     */ {222,40,60},   /* 5 pavement */
    {0,128,128},   /* 6 tree         */ {128,128,192}, /* 7 sign     */ {128,64,64},   /* 8 fence    */
    {128,0,64},    /* 9 vehicle      */ {0,64,64},     /* 10 pedestrian */ {192,128,0} /* 11 cyclist */
};
int sso_label_of_bgr(uint8_t b, uint8_t g, uint8_t r)
{
    for (int i = 0; i < 12; i++)
        if (sso_palette_bgr[i][0] == b && sso_palette_bgr[i][1] == g && sso_palette_bgr[i][2] == r) return i;
    return 255;
}

/* rgbdframe.h:63-75.  (0,0,0) sentinel when d == 0.  z = float(double(d)/scale); x = float((u-cx)*double(z)/fx) */
void sso_project2dTo3d(const uint16_t* depth, int w, int h, const sso_camera* cam, int u, int v, float out[3])
{
    (void)h;
    out[0] = out[1] = out[2] = 0.f;
    if (!depth) return;
    uint16_t d = depth[(size_t)v * w + u];
    if (d == 0) return;
    float z = (float)((double)d / cam->scale);
    out[2] = z;
    out[0] = (float)(((double)u - cam->cx) * (double)z / cam->fx);
    out[1] = (float)(((double)v - cam->cy) * (double)z / cam->fy);
}

/* mapper.cpp:189-216.  The reference dilates with an UNINITIALISED cv::Mat(3,3,CV_8UC1) (mapper.cpp:214) -- not
 * reproducible; contract: full 3x3 ones, 2 iterations (== one 5x5 box dilate), border pixels do not contribute. */
void sso_moving_mask(const uint8_t* sem, int w, int h, uint8_t* mask)
{
    uint8_t* a = (uint8_t*)calloc((size_t)w * h, 1);
    uint8_t* b = (uint8_t*)malloc((size_t)w * h);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const uint8_t* p = sem + ((size_t)i * w + j) * 3;
            if ((p[0] == 0 && p[1] == 64 && p[2] == 64) || (p[0] == 192 && p[1] == 128 && p[2] == 0)) a[(size_t)i * w + j] = 255;
        }
    for (int it = 0; it < 2; it++) {
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                uint8_t m = 0;
                for (int di = -1; di <= 1; di++)
                    for (int dj = -1; dj <= 1; dj++) {
                        int y = i + di, x = j + dj;
                        if (y < 0 || y >= h || x < 0 || x >= w) continue;
                        if (a[(size_t)y * w + x] > m) m = a[(size_t)y * w + x];
                    }
                b[(size_t)i * w + j] = m;
            }
        uint8_t* t = a; a = b; b = t;
    }
    memcpy(mask, a, (size_t)w * h);
    free(a); free(b);
}

/* mapper.cpp:21-92.  Row-major (the omp pragma at :21 is inert: no -fopenmp, CMakeLists.txt:11).
 * pushed colour = camera rgb (mapper.cpp:72-84 pushes point_img); semantic colour only gates (:41-55).
 * Extension over PointXYZRGBA: label id in the PointXYZRGBL slot (bytes 20..23), w = 1.0f like PCL_ADD_POINT4D.
 * transformPointCloud (PCL 1.7, dense cloud): x' = float(t00*x + t01*y + t02*z + t03) in double, left to right. */
int sso_backproject(const uint16_t* depth, const uint8_t* rgb, const uint8_t* sem, const uint8_t* mask,
                    int w, int h, const sso_camera* cam, const double* T, double max_distance, sso_point* out)
{
    int n = 0;
    for (int m = 0; m < h; m++)
        for (int c = 0; c < w; c++) {
            uint16_t d = depth[(size_t)m * w + c];
            if (d == 0) continue;
            if ((double)d > max_distance * cam->scale) continue;
            if (mask[(size_t)m * w + c] == 255) continue;
            const uint8_t* s = sem + ((size_t)m * w + c) * 3;
            uint8_t pb = s[0], pg = s[1], pr = s[2];
            if ((pb == 128 && pg == 128 && pr == 128) || (pb == 128 && pg == 192 && pr == 192) || (pb == 192 && pg == 128 && pr == 0)) continue;
            float p[3]; sso_project2dTo3d(depth, w, h, cam, c, m, p);
            sso_point* o = &out[n++];
            memset(o, 0, sizeof(*o));
            if (T) {
                double x = p[0], y = p[1], z = p[2];
                o->x = (float)(T[0] * x + T[4] * y + T[8]  * z + T[12]);
                o->y = (float)(T[1] * x + T[5] * y + T[9]  * z + T[13]);
                o->z = (float)(T[2] * x + T[6] * y + T[10] * z + T[14]);
            } else { o->x = p[0]; o->y = p[1]; o->z = p[2]; }
            o->w = 1.0f;
            const uint8_t* q = rgb + ((size_t)m * w + c) * 3;
            o->b = q[0]; o->g = q[1]; o->r = q[2]; o->a = 0;
            o->label = (uint32_t)sso_label_of_bgr(pb, pg, pr);
        }
    return n;
}

/* ---------------- K12: pcl::VoxelGrid (PCL 1.7 filters/impl/voxel_grid.hpp) ----------------
 * ijk = floor(p * inv_leaf) (float product, inv_leaf = 1.0f/leaf); output one centroid per occupied voxel, sorted by
 * the linear index i + j*dx + k*dx*dy == sorted by (k,j,i); xyz mean; rgb mean truncated; alpha dropped.
 * PCL sums in float in std::sort's (unstable) order, i.e. its last bits are not defined by the source.  CONTRACT: sums
 * are EXACT -- each coordinate is rounded once to a 2^-24 m grid (llrint, half-even), summed in int64, and the mean is
 * float( double(sum)/double(n) * 2^-24 ); colour mean = floor(sum/n).  Exact sums are associative, so per-tile,
 * per-frame and per-GPU partial tables merge to the bit-identical map.  Label = argmax of votes over ids 0..11
 * (ties -> lowest id), 255 when no vote. */
#define VOX_BIAS (1 << 20)
/* CONTRACT (range): a point whose voxel index floor(c * inv_leaf) is not finite or not inside (-2^20, 2^20) on some axis cannot be keyed in 21 bits
 * per axis; it is SKIPPED, the way pcl::VoxelGrid skips non-finite points of a non-dense cloud (voxel_grid.hpp, `if (!input_->is_dense)`), and the
 * device raises SSM_E_VOXEL_RANGE once.  Returns -1 for such a point. */
int64_t sso_voxel_key(float x, float y, float z, float inv_leaf)
{
    const float fi = floorf(x * inv_leaf), fj = floorf(y * inv_leaf), fk = floorf(z * inv_leaf);
    if (!(fabsf(fi) < 1048576.0f) || !(fabsf(fj) < 1048576.0f) || !(fabsf(fk) < 1048576.0f)) return -1;
    int64_t i = (int64_t)floorf(x * inv_leaf) + VOX_BIAS;
    int64_t j = (int64_t)floorf(y * inv_leaf) + VOX_BIAS;
    int64_t k = (int64_t)floorf(z * inv_leaf) + VOX_BIAS;
    return (k << 42) | (j << 21) | i;
}
static void vox_add(sso_voxel* d, const sso_voxel* s)
{
    d->sx += s->sx; d->sy += s->sy; d->sz += s->sz; d->sr += s->sr; d->sg += s->sg; d->sb += s->sb; d->n += s->n;
    for (int i = 0; i < 12; i++) d->hist[i] += s->hist[i];
}
typedef struct { int64_t key; int32_t idx; } keyidx_t;
static int keyidx_cmp(const void* a, const void* b)
{
    int64_t x = ((const keyidx_t*)a)->key, y = ((const keyidx_t*)b)->key;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static int points_to_table(const sso_point* pts, int n, float leaf, sso_voxel** out)
{
    const float inv = 1.0f / leaf;
    keyidx_t* ki = (keyidx_t*)malloc(sizeof(keyidx_t) * (n > 0 ? n : 1));
    { int kept = 0;
      for (int i = 0; i < n; i++) { const int64_t key = sso_voxel_key(pts[i].x, pts[i].y, pts[i].z, inv); if (key < 0) continue; ki[kept].key = key; ki[kept].idx = i; kept++; }
      n = kept; }
    qsort(ki, n, sizeof(keyidx_t), keyidx_cmp);      /* like PCL: sort (index, point) pairs, then reduce runs */
    int m = 0;
    for (int i = 0; i < n; i++) if (i == 0 || ki[i].key != ki[i-1].key) m++;
    sso_voxel* v = (sso_voxel*)calloc(m > 0 ? m : 1, sizeof(sso_voxel));
    int o = -1;
    for (int i = 0; i < n; i++) {
        if (i == 0 || ki[i].key != ki[i-1].key) { o++; v[o].key = ki[i].key; }
        const sso_point* p = &pts[ki[i].idx];
        v[o].sx += llrint((double)p->x * 16777216.0);
        v[o].sy += llrint((double)p->y * 16777216.0);
        v[o].sz += llrint((double)p->z * 16777216.0);
        v[o].sr += p->r; v[o].sg += p->g; v[o].sb += p->b; v[o].n += 1;
        if (p->label < 12) v[o].hist[p->label] += 1;
    }
    free(ki);
    *out = v;
    return m;
}
int sso_voxel_merge(sso_voxel* dst, int m, int cap, const sso_voxel* src, int ms)
{
    /* count */
    int i = 0, j = 0, tot = 0;
    while (i < m || j < ms) {
        if (j >= ms || (i < m && dst[i].key < src[j].key)) i++;
        else if (i >= m || src[j].key < dst[i].key) j++;
        else { i++; j++; }
        tot++;
    }
    if (tot > cap) return -1;
    /* merge from the back, in place */
    i = m - 1; j = ms - 1; int o = tot - 1;
    while (j >= 0) {
        if (i >= 0 && dst[i].key > src[j].key) dst[o--] = dst[i--];
        else if (i >= 0 && dst[i].key == src[j].key) { sso_voxel t = dst[i--]; vox_add(&t, &src[j--]); dst[o--] = t; }
        else dst[o--] = src[j--];
    }
    return tot;
}
int sso_voxel_accumulate(const sso_point* pts, int n, float leaf, sso_voxel* tab, int m, int cap)
{
    sso_voxel* t; int mt = points_to_table(pts, n, leaf, &t);
    int r = sso_voxel_merge(tab, m, cap, t, mt);
    free(t);
    return r;
}
void sso_voxel_export(const sso_voxel* tab, int m, sso_point* out)
{
    for (int i = 0; i < m; i++) {
        const sso_voxel* v = &tab[i]; sso_point* o = &out[i];
        memset(o, 0, sizeof(*o));
        double n = (double)v->n;
        o->x = (float)(((double)v->sx / n) * (1.0 / 16777216.0));
        o->y = (float)(((double)v->sy / n) * (1.0 / 16777216.0));
        o->z = (float)(((double)v->sz / n) * (1.0 / 16777216.0));
        o->w = 1.0f;
        o->r = (uint8_t)(v->sr / v->n); o->g = (uint8_t)(v->sg / v->n); o->b = (uint8_t)(v->sb / v->n); o->a = 0;
        uint32_t best = 0; int lab = 255;
        for (int k = 0; k < 12; k++) if (v->hist[k] > best) { best = v->hist[k]; lab = k; }
        o->label = (uint32_t)lab;
    }
}
int sso_voxel_filter(const sso_point* pts, int n, float leaf, sso_point* out, int cap)
{
    if (n > 0) {   /* PCL's overflow guard on the bounding box (voxel_grid.hpp: dx*dy*dz > INT_MAX) */
        const float inv = 1.0f / leaf;
        float mn[3] = {pts[0].x, pts[0].y, pts[0].z}, mx[3] = {pts[0].x, pts[0].y, pts[0].z};
        for (int i = 1; i < n; i++) {
            const float p[3] = {pts[i].x, pts[i].y, pts[i].z};
            for (int a = 0; a < 3; a++) { if (p[a] < mn[a]) mn[a] = p[a]; if (p[a] > mx[a]) mx[a] = p[a]; }
        }
        int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
        if (dx * dy * dz > (int64_t)2147483647) return -2;
    }
    sso_voxel* t; int m = points_to_table(pts, n, leaf, &t);
    if (m > cap) { free(t); return -1; }
    sso_voxel_export(t, m, out);
    free(t);
    return m;
}
