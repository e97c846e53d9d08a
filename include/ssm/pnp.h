// ssm/pnp.h -- rgbd_tutor::PnPSolver (reference include/pnp.h, src/pnp.cpp:5-118): pose-only bundle adjustment of the
// reprojection error.  The reference drives g2o (EdgeSE3ProjectXYZOnlyPose, Levenberg, Huber delta sqrt(5.991), four
// rounds of ten iterations, chi2 > 5.991 -> outlier, robust kernel dropped after round 3).  g2o is absent: this is a
// from-scratch dense 6-DoF Levenberg-Marquardt with the same schedule.  It is a host-side CONSUMER of the match
// tables, not part of the GPU path, and is not bit-identical to g2o.  The inlier bookkeeping
// quirks of pnp.cpp:74-89,115 (mixed indices; success test on the vector length) are NOT reproduced.
#pragma once
#include "common_headers.h"
#include "orb.h"
namespace rgbd_tutor {
struct PNP_INFORMATION { int numFeatureMatches = 0, numInliers = 0; Eigen::Isometry3d T = Eigen::Isometry3d::Identity(); };
class PnPSolver {
public:
    PnPSolver(const ParameterReader& para, const OrbFeature& orbFeature) : parameterReader(para), orb(orbFeature) {
        min_inliers = para.getData<int>("pnp_min_inliers", 10); min_match = para.getData<int>("pnp_min_matches", 15);
    }
    // img: pixels in the current frame; obj: the same points in world coordinates; transform: world -> camera (initial value in, estimate out)
    bool solvePnP(const vector<cv::Point2f>& img, const vector<cv::Point3f>& obj, const CAMERA_INTRINSIC_PARAMETERS& camera,
                  vector<int>& inliersIndex, Eigen::Isometry3d& transform) {
        const size_t n = img.size();
        vector<char> inlier(n, 1);
        int good = 0;
        for (size_t i = 0; i < n; i++) { if (obj[i] == cv::Point3f(0, 0, 0)) inlier[i] = 0; else good++; }
        const double chi2_th = 5.991, delta = sqrt(5.991);
        double R[9], t[3];
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = transform(r, c); t[r] = transform(r, 3); }
        const Eigen::Isometry3d init = transform;
        for (int round = 0; round < 4; round++) {
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = init(r, c); t[r] = init(r, 3); }      // every round restarts from the initial value (pnp.cpp:66)
            double lambda = 1e-4;
            for (int it = 0; it < 10; it++) {
                double H[36] = {0}, b[6] = {0}, cost = 0;
                accumulate(img, obj, camera, inlier, R, t, round < 3 ? delta : 0.0, H, b, &cost);
                bool stepped = false;
                for (int tries = 0; tries < 6 && !stepped; tries++) {
                    double A[36], x[6];
                    for (int i = 0; i < 36; i++) A[i] = H[i];
                    for (int i = 0; i < 6; i++) A[i * 6 + i] += lambda * (H[i * 6 + i] + 1e-9);
                    if (!solve6(A, b, x)) { lambda *= 10; continue; }
                    double R2[9], t2[3]; applyUpdate(R, t, x, R2, t2);
                    double H2[36] = {0}, b2[6] = {0}, cost2 = 0;
                    accumulate(img, obj, camera, inlier, R2, t2, round < 3 ? delta : 0.0, H2, b2, &cost2);
                    if (cost2 <= cost) { memcpy(R, R2, sizeof(R)); memcpy(t, t2, sizeof(t)); lambda = max(lambda * 0.1, 1e-12); stepped = true; }
                    else lambda *= 10;
                }
                if (!stepped) break;
            }
            good = 0;
            for (size_t i = 0; i < n; i++) {
                if (obj[i] == cv::Point3f(0, 0, 0)) { inlier[i] = 0; continue; }
                double e[2]; if (!residual(img[i], obj[i], camera, R, t, e)) { inlier[i] = 0; continue; }
                inlier[i] = (e[0] * e[0] + e[1] * e[1] <= chi2_th); good += inlier[i];
            }
            if (good < 5) break;
        }
        for (size_t i = 0; i < n; i++) if (inlier[i]) inliersIndex.push_back((int)i);
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) transform(r, c) = R[r * 3 + c]; transform(r, 3) = t[r]; }
        return (int)inliersIndex.size() > min_inliers;
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
        vector<int> inl; Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
        bool ok = solvePnP(img, obj, frame1->camera, inl, T);
        info.numFeatureMatches = (int)img.size(); info.numInliers = (int)inl.size(); info.T = T;
        return ok && (int)inl.size() >= min_inliers;
    }
protected:
    static bool residual(const cv::Point2f& u, const cv::Point3f& X, const CAMERA_INTRINSIC_PARAMETERS& k, const double* R, const double* t, double e[2], double pc[3] = nullptr) {
        double p[3];
        for (int r = 0; r < 3; r++) p[r] = R[r * 3] * X.x + R[r * 3 + 1] * X.y + R[r * 3 + 2] * X.z + t[r];
        if (pc) { pc[0] = p[0]; pc[1] = p[1]; pc[2] = p[2]; }
        if (p[2] <= 1e-9) return false;
        e[0] = u.x - (k.fx * p[0] / p[2] + k.cx); e[1] = u.y - (k.fy * p[1] / p[2] + k.cy);
        return true;
    }
    static void accumulate(const vector<cv::Point2f>& img, const vector<cv::Point3f>& obj, const CAMERA_INTRINSIC_PARAMETERS& k, const vector<char>& inlier,
                           const double* R, const double* t, double huber, double* H, double* b, double* cost) {
        for (size_t i = 0; i < img.size(); i++) {
            if (!inlier[i]) continue;
            double e[2], p[3];
            if (!residual(img[i], obj[i], k, R, t, e, p)) continue;
            const double iz = 1.0 / p[2], iz2 = iz * iz;
            // d(proj)/d(xi), xi = (rho, phi) left-multiplied: p' = p + rho + phi x p ;  e = u - proj  =>  J = -dproj
            double J[2][6];
            const double a[2][3] = {{k.fx * iz, 0, -k.fx * p[0] * iz2}, {0, k.fy * iz, -k.fy * p[1] * iz2}};
            for (int r = 0; r < 2; r++) {
                J[r][0] = -a[r][0]; J[r][1] = -a[r][1]; J[r][2] = -a[r][2];
                J[r][3] = -(a[r][2] * p[1] - a[r][1] * p[2]);       // d/dphi_x :  (phi x p) = (phi_y p_z - phi_z p_y, phi_z p_x - phi_x p_z, phi_x p_y - phi_y p_x)
                J[r][4] = -(a[r][0] * p[2] - a[r][2] * p[0]);
                J[r][5] = -(a[r][1] * p[0] - a[r][0] * p[1]);
            }
            const double e2 = e[0] * e[0] + e[1] * e[1], en = sqrt(e2);
            double w = 1.0, rho = e2;
            if (huber > 0 && en > huber) { w = huber / en; rho = 2 * huber * en - huber * huber; }
            *cost += rho;
            for (int r = 0; r < 2; r++) for (int c = 0; c < 6; c++) { b[c] -= w * J[r][c] * e[r]; for (int d = 0; d < 6; d++) H[c * 6 + d] += w * J[r][c] * J[r][d]; }
        }
    }
    static bool solve6(double* A, const double* b, double* x) {                     // Cholesky, 6x6 SPD
        double L[36] = {0};
        for (int i = 0; i < 6; i++) for (int j = 0; j <= i; j++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) { if (s <= 1e-18) return false; L[i * 6 + i] = sqrt(s); } else L[i * 6 + j] = s / L[j * 6 + j];
        }
        double y[6];
        for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k]; y[i] = s / L[i * 6 + i]; }
        for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k]; x[i] = s / L[i * 6 + i]; }
        return true;
    }
    static void applyUpdate(const double* R, const double* t, const double* x, double* R2, double* t2) {   // T <- exp(xi) * T (first order in rho, Rodrigues in phi)
        const double th = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
        double dR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (th > 1e-12) {
            const double kx = x[3] / th, ky = x[4] / th, kz = x[5] / th, c = cos(th), s = sin(th), v = 1 - c;
            const double M[9] = {c + kx * kx * v, kx * ky * v - kz * s, kx * kz * v + ky * s, ky * kx * v + kz * s, c + ky * ky * v, ky * kz * v - kx * s,
                                 kz * kx * v - ky * s, kz * ky * v + kx * s, c + kz * kz * v};
            memcpy(dR, M, sizeof(M));
        }
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) R2[r * 3 + c] = dR[r * 3] * R[c] + dR[r * 3 + 1] * R[3 + c] + dR[r * 3 + 2] * R[6 + c];
            t2[r] = dR[r * 3] * t[0] + dR[r * 3 + 1] * t[1] + dR[r * 3 + 2] * t[2] + x[r];
        }
    }
    const ParameterReader& parameterReader;
    const OrbFeature& orb;
    int min_inliers = 10, min_match = 30;
};
}  // namespace rgbd_tutor
