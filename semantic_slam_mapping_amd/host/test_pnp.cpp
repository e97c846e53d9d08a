// test_pnp -- runs rgbd_tutor::PnPSolver::solvePnP (include/ssm/pnp.h) on a case file written by tests/test_pnp.py and dumps the result, so that the
// host class can be compared with oracle/pnp.c.  Host only: no device call (OrbFeature creates its context lazily and is never used here).
// case file: i32 n, i32 pad, f64 cam[5] (cx cy fx fy scale), f64 T[16] (column-major initial transform), f32 img[2n], f32 obj[3n]
// result   : i32 ok, i32 m, f64 T[16] (column-major), i32 inliers[m]
#include "ssm/pnp.h"
#include <cstdio>
using namespace std;
using namespace rgbd_tutor;
int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s parameters.txt case.bin result.bin\n", argv[0]); return 2; }
    ParameterReader para(argv[1]);
    OrbFeature orb(para);
    PnPSolver pnp(para, orb);
    FILE* f = fopen(argv[2], "rb"); if (!f) { perror(argv[2]); return 2; }
    int32_t hdr[2]; double cam[5], T[16];
    if (fread(hdr, 4, 2, f) != 2 || fread(cam, 8, 5, f) != 5 || fread(T, 8, 16, f) != 16) return 2;
    const int n = hdr[0];
    vector<float> img(2 * (size_t)n), obj(3 * (size_t)n);
    if (n && (fread(img.data(), 4, img.size(), f) != img.size() || fread(obj.data(), 4, obj.size(), f) != obj.size())) return 2;
    fclose(f);
    vector<cv::Point2f> im; vector<cv::Point3f> ob;
    for (int i = 0; i < n; i++) { im.push_back(cv::Point2f(img[2 * i], img[2 * i + 1])); ob.push_back(cv::Point3f(obj[3 * i], obj[3 * i + 1], obj[3 * i + 2])); }
    CAMERA_INTRINSIC_PARAMETERS k; k.cx = cam[0]; k.cy = cam[1]; k.fx = cam[2]; k.fy = cam[3]; k.scale = cam[4];
    Eigen::Isometry3d tr; for (int i = 0; i < 16; i++) tr.matrix().data()[i] = T[i];
    vector<int> inl;
    const int32_t ok = pnp.solvePnP(im, ob, k, inl, tr) ? 1 : 0, m = (int32_t)inl.size();
    FILE* o = fopen(argv[3], "wb"); if (!o) { perror(argv[3]); return 2; }
    fwrite(&ok, 4, 1, o); fwrite(&m, 4, 1, o); fwrite(tr.matrix().data(), 8, 16, o); if (m) fwrite(inl.data(), 4, (size_t)m, o);
    fclose(o);
    return 0;
}
