// san_stub_device.cpp -- TEST INFRASTRUCTURE for the CPU sanitizer builds of the host layer (make SAN=asan|ubsan|tsan): a stand-in for the handful of
// libssm_hip.so entry points that Mapper / PoseGraph reach, so that the THREADING of the host classes (Mapper::viewer on its own thread against the main
// thread's tryInsertKeyFrame: the place of the reference's unlocked reads, /root/reference/src/mapper.cpp:114-136) can run under ThreadSanitizer in the
// build container, which has no GPU and where sanitizer runtimes and the HIP runtime do not mix.  It computes nothing of the product: the numbers it
// returns are placeholders.  It is linked into test_threads / test_pnp / test_png of a SAN build ONLY -- never into exp_mapping or the library.
#include "ssm_hip.h"
#include "ssm/pnp_core.h"
#include <cstring>
#include <string>
#include <vector>
struct ssm_ctx { ssm_config cfg; std::string err; };
extern "C" {
void ssm_config_default(ssm_config* c) { memset(c, 0, sizeof(*c)); c->width = 640; c->height = 480; c->orb_features = 1000; c->orb_scale = 1.2f; c->orb_levels = 8;
    c->orb_iniThFAST = 20; c->orb_minThFAST = 7; c->knn_match_ratio = 0.8; c->tracker_ref_frames = 5; c->mapper_resolution = 0.1; c->mapper_max_distance = 40;
    c->camera.cx = 318.6; c->camera.cy = 255.3; c->camera.fx = 517.3; c->camera.fy = 516.5; c->camera.scale = 1000.0; c->max_batch = 1; c->voxel_capacity_log2 = 16; }
int ssm_create(int, const ssm_config* cfg, ssm_ctx** out) { *out = new ssm_ctx(); (*out)->cfg = *cfg; return SSM_OK; }
void ssm_destroy(ssm_ctx* c) { delete c; }
const char* ssm_last_error(const ssm_ctx* c) { return c ? c->err.c_str() : "stub"; }
int ssm_orb_capacity(const ssm_ctx*) { return 1024; }
int ssm_backproject(ssm_ctx*, const uint16_t* depth, const uint8_t* rgb, const uint8_t*, int w, int h, const ssm_camera*, const double*, double, ssm_point* out, int cap, int* n_out)
{
    int n = 0;
    for (int v = 0; v < h; v += 4) for (int u = 0; u < w; u += 4) {
        const uint16_t d = depth[(size_t)v * w + u];
        if (!d || n >= cap) continue;
        ssm_point p; memset(&p, 0, sizeof(p)); p.x = u * 0.01f; p.y = v * 0.01f; p.z = d * 0.001f; p.w = 1.f; p.b = rgb[((size_t)v * w + u) * 3]; p.label = 255;
        out[n++] = p;
    }
    *n_out = n; return SSM_OK;
}
// PnPSolver::solvePnP goes to the device block whenever its thread has a context (include/ssm/pnp.h): under the sanitizers the same contract arithmetic
// (pnp_core.h) runs on the host in its place
int ssm_pnp_solve(ssm_ctx*, const float* img, const float* obj, int n, const double cam[4], int min_inliers, double T[16], uint8_t* inliers, int* n_inliers, int* success)
{
    if (n < 0 || n > 65535) return SSM_E_CAPACITY;
    ssm_pnp::Camera c; c.fx = cam[0]; c.fy = cam[1]; c.cx = cam[2]; c.cy = cam[3];
    std::vector<ssm_pnp::Edge> edges((size_t)n + 1); std::vector<unsigned char> inl((size_t)n + 1);
    int ok = 0;
    ssm_pnp::solve(img, obj, n, c, min_inliers, T, inl.data(), edges.data(), &ok);
    int m = 0;
    for (int i = 0; i < n; i++) { if (inliers) inliers[i] = inl[i]; m += inl[i] != 0; }
    if (n_inliers) *n_inliers = m;
    if (success) *success = ok;
    return SSM_OK;
}
// device-resident Mapper entry points: a "cloud" is a host vector here, the map update the same placeholder thinning as ssm_voxel_filter below
struct ssm_cloud { std::vector<ssm_point> p; };
static std::vector<ssm_point> g_vmap;
int ssm_backproject_dev(ssm_ctx* c, const uint16_t* depth, const uint8_t* rgb, const uint8_t* sem, int w, int h, const ssm_camera* cam, double md, ssm_cloud** out)
{
    ssm_cloud* cl = new ssm_cloud(); cl->p.resize((size_t)w * h / 16 + 1); int n = 0;
    ssm_backproject(c, depth, rgb, sem, w, h, cam, nullptr, md, cl->p.data(), (int)cl->p.size(), &n);
    cl->p.resize(n); *out = cl; return SSM_OK;
}
int ssm_cloud_size(const ssm_cloud* cl) { return cl ? (int)cl->p.size() : 0; }
void ssm_cloud_free(ssm_ctx*, ssm_cloud* cl) { delete cl; }
int ssm_cloud_fetch(ssm_ctx*, const ssm_cloud* cl, const double*, ssm_point* out, int cap, int* n_out)
{ *n_out = (int)cl->p.size(); if (*n_out > cap) return SSM_E_CAPACITY; memcpy(out, cl->p.data(), cl->p.size() * sizeof(ssm_point)); return SSM_OK; }
int ssm_viewer_map_update(ssm_ctx*, int rebuild, ssm_cloud* const* clouds, const double*, int n, float, int* n_out)
{
    std::vector<ssm_point> all; if (!rebuild) all = g_vmap;
    for (int i = 0; i < n; i++) all.insert(all.end(), clouds[i]->p.begin(), clouds[i]->p.end());
    g_vmap.clear();
    for (size_t i = 0; i < all.size(); i += 3) g_vmap.push_back(all[i]);
    if (n_out) *n_out = (int)g_vmap.size();
    return SSM_OK;
}
int ssm_viewer_map_release(ssm_ctx*, int) { return SSM_OK; }
int ssm_host_alloc(size_t bytes, void** out) { *out = malloc(bytes ? bytes : 1); return *out ? SSM_OK : SSM_E_NOMEM; }
int ssm_host_free(void* p) { free(p); return SSM_OK; }
int ssm_viewer_map_fetch(ssm_ctx*, ssm_point* out, int cap, int* n_out)
{ *n_out = (int)g_vmap.size(); if (*n_out > cap) return SSM_E_CAPACITY; if (*n_out) memcpy(out, g_vmap.data(), g_vmap.size() * sizeof(ssm_point)); return SSM_OK; }
int ssm_voxel_filter(ssm_ctx*, const ssm_point* pts, int n, float, ssm_point* out, int cap, int* n_out)
{
    int m = 0;
    for (int i = 0; i < n && m < cap; i += 3) out[m++] = pts[i];
    *n_out = m; return SSM_OK;
}
}
