// ssm/pnp.h -- rgbd_tutor::PnPSolver (reference include/pnp.h, src/pnp.cpp): pose-only bundle adjustment of the reprojection error, the host-side
// CONSUMER of the match tables.  The reference drives g2o (un-vendored, absent here); this header implements the algorithm
// g2o runs for pnp.cpp's set-up -- EdgeSE3ProjectXYZOnlyPose (analytic Jacobian, update T <- exp(d) T), RobustKernelHuber with delta (float)sqrt(5.991),
// OptimizationAlgorithmLevenberg (lambda0 = 1e-5 max diag H, gain ratio with the +1e-3 guard, lambda scaled by clamp(1 - (2 gain - 1)^3, 1/3, 2/3) or
// multiplied by nu = 2, 4, ..., at most 10 trials per iteration), four rounds of optimize(10), chi2 > 5.991 -> outlier, kernels dropped in round 3 --
// and keeps pnp.cpp's inlier bookkeeping AS WRITTEN (stale chi2 of edges already out, `inliers[i]` indexed by edge
// position, success decided by the flag vector's length).  The arithmetic itself is include/ssm/pnp_core.h (shared with the bulk tracker inside libssm_hip.so,
// host path and device chain: lane-ordered sums, polynomial sin / cos); oracle/pnp.c is the C restatement it is tested against (tests/test_pnp.py) and
// tests/golden/pyref.py a second one; what is and is not pinned is said there.  The rotation is kept as a matrix (g2o keeps a re-normalised quaternion).
#pragma once
#include "common_headers.h"
#include "orb.h"
#include "pnp_core.h"
#include <limits>
namespace rgbd_tutor {
struct PNP_INFORMATION { int numFeatureMatches = 0, numInliers = 0; Eigen::Isometry3d T = Eigen::Isometry3d::Identity(); };
class PnPSolver {
public:
    PnPSolver(const ParameterReader& para, const OrbFeature& orbFeature) : parameterReader(para), orb(orbFeature) {
        min_inliers = para.getData<int>("pnp_min_inliers", 10); min_match = para.getData<int>("pnp_min_matches", 15);
        on_device = para.getData<int>("pnp_device", 1) != 0;                  // (not a reference parameter) 0: always solve on the host; same bits either way
    }
    // img: pixels in frame 2; obj: the same points in frame 1 (camera frame); transform: initial value in, estimate out (src/pnp.cpp:5-118)
    bool solvePnP(const vector<cv::Point2f>& img, const vector<cv::Point3f>& obj, const CAMERA_INTRINSIC_PARAMETERS& camera,
                  vector<int>& inliersIndex, Eigen::Isometry3d& transform) {
        // the arithmetic (and its CPU == GPU numeric contract) lives in pnp_core.h, shared with the bulk tracker of libssm_hip.so
        const int n = (int)img.size();
        vector<float> im((size_t)2 * n + 2), ob((size_t)3 * n + 3);
        for (int i = 0; i < n; i++) { im[2 * i] = img[i].x; im[2 * i + 1] = img[i].y; ob[3 * i] = obj[i].x; ob[3 * i + 1] = obj[i].y; ob[3 * i + 2] = obj[i].z; }
        vector<unsigned char> inl((size_t)n + 1);
        ssm_pnp::Camera cam; cam.fx = camera.fx; cam.fy = camera.fy; cam.cx = camera.cx; cam.cy = camera.cy;
        double T[16]; for (int k = 0; k < 16; k++) T[k] = transform.data()[k];
        int success = 0;
        ssm::Device* dev = on_device ? OrbFeature::lastDevice() : nullptr;
        if (dev && n > 65535) dev = nullptr;                                     // the device block numbers its edges with 16 bits: larger lists are solved on the host (same bits)
        if (dev) {                                                               // one 1024-thread block of libssm_hip.so (kernels_pnp.hip): ~3 x the host core
            const double kc[4] = {camera.fx, camera.fy, camera.cx, camera.cy}; int m = 0;
            dev->check(ssm_pnp_solve(dev->ctx(), im.data(), ob.data(), n, kc, min_inliers, T, inl.data(), &m, &success), "ssm_pnp_solve");
        } else {
            vector<ssm_pnp::Edge> edges((size_t)n + 1);
            ssm_pnp::solve(im.data(), ob.data(), n, cam, min_inliers, T, inl.data(), edges.data(), &success);
        }
        for (int i = 0; i < n; i++) if (inl[i]) inliersIndex.push_back(i);
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) transform(r, c) = T[c * 4 + r];
        return success != 0;
    }
    // match(frame1, frame2) + gather 3-D (frame1) / 2-D (frame2) + solve (reference src/pnp.cpp:120-226)
    bool solvePnPLazy(const RGBDFrame::Ptr& frame1, const RGBDFrame::Ptr frame2, PNP_INFORMATION& info, bool drawMatches = false) {
        (void)drawMatches;
        vector<cv::DMatch> matches = orb.match(frame1, frame2);
        if ((int)matches.size() <= min_match) return false;
        vector<cv::Point3f> obj; vector<cv::Point2f> img;
        for (auto& m : matches) {
            cv::Point3f p = frame1->features[m.queryIdx].position;
            if (p == cv::Point3f(0, 0, 0)) continue;
            obj.push_back(p); img.push_back(frame2->features[m.trainIdx].keypoint.pt);
        }
        if ((int)img.size() <= min_match) return false;
        vector<int> inl; Eigen::Isometry3d T = frame1->T_f_w.inverse() * frame2->T_f_w;      // init_transform, pnp.cpp:164
        bool ok = solvePnP(img, obj, frame1->camera, inl, T);
        info.numFeatureMatches = (int)img.size(); info.numInliers = (int)inl.size(); info.T = T;
        (void)ok;
        return info.numInliers >= min_inliers;                                  // pnp.cpp:221-225 ignores solvePnP's own return value
    }
protected:
    const ParameterReader& parameterReader;
    const OrbFeature& orb;
    int min_inliers = 10, min_match = 30; bool on_device = true;
};
}  // namespace rgbd_tutor
