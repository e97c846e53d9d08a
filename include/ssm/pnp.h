// ssm/pnp.h -- rgbd_tutor::PnPSolver (reference include/pnp.h, src/pnp.cpp): pose-only bundle adjustment of the reprojection error, the host-side
// CONSUMER of the match tables.  The reference drives g2o (un-vendored, absent here); this header implements the algorithm
// g2o runs for pnp.cpp's set-up -- EdgeSE3ProjectXYZOnlyPose (analytic Jacobian, update T <- exp(d) T), RobustKernelHuber with delta (float)sqrt(5.991),
// OptimizationAlgorithmLevenberg (lambda0 = 1e-5 max diag H, gain ratio with the +1e-3 guard, lambda scaled by clamp(1 - (2 gain - 1)^3, 1/3, 2/3) or
// multiplied by nu = 2, 4, ..., at most 10 trials per iteration), four rounds of optimize(10), chi2 > 5.991 -> outlier, kernels dropped in round 3 --
// and keeps pnp.cpp's inlier bookkeeping AS WRITTEN (stale chi2 of edges already out, `inliers[i]` indexed by edge
// position, success decided by the flag vector's length).  oracle/pnp.c is the C restatement it is tested against (tests/test_pnp.py); what is and is
// not pinned is said there.  The rotation is kept as a matrix (g2o keeps a re-normalised quaternion): rounding-level difference from g2o itself.
#pragma once
#include "common_headers.h"
#include "orb.h"
#include <limits>
namespace rgbd_tutor {
struct PNP_INFORMATION { int numFeatureMatches = 0, numInliers = 0; Eigen::Isometry3d T = Eigen::Isometry3d::Identity(); };
class PnPSolver {
public:
    PnPSolver(const ParameterReader& para, const OrbFeature& orbFeature) : parameterReader(para), orb(orbFeature) {
        min_inliers = para.getData<int>("pnp_min_inliers", 10); min_match = para.getData<int>("pnp_min_matches", 15);
    }
    // img: pixels in frame 2; obj: the same points in frame 1 (camera frame); transform: initial value in, estimate out (src/pnp.cpp:5-118)
    bool solvePnP(const vector<cv::Point2f>& img, const vector<cv::Point3f>& obj, const CAMERA_INTRINSIC_PARAMETERS& camera,
                  vector<int>& inliersIndex, Eigen::Isometry3d& transform) {
        const double delta = (double)(float)sqrt(5.991);
        struct Edge { int id, level; bool robust; double X[3], u, v, e0, e1; double chi2() const { return e0 * e0 + e1 * e1; } };
        vector<Edge> edges;
        vector<bool> inliers(img.size(), true);
        int good = 0;
        for (size_t i = 0; i < obj.size(); i++) {
            if (obj[i] == cv::Point3f(0, 0, 0)) { inliers[i] = false; continue; }
            good++;
            Edge e; e.id = (int)i; e.level = 0; e.robust = true; e.X[0] = obj[i].x; e.X[1] = obj[i].y; e.X[2] = obj[i].z; e.u = img[i].x; e.v = img[i].y; e.e0 = e.e1 = 0;
            edges.push_back(e);
        }
        Pose init; for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) init.R[3 * r + c] = transform(r, c); init.t[r] = transform(r, 3); }
        Pose est = init;
        auto project_error = [&](Edge& e, const Pose& P) { double p[3]; P.map(e.X, p); e.e0 = e.u - (p[0] / p[2] * camera.fx + camera.cx); e.e1 = e.v - (p[1] / p[2] * camera.fy + camera.cy); };
        auto huber = [&](double e2, double& rho0, double& rho1) { const double d2 = delta * delta; if (e2 <= d2) { rho0 = e2; rho1 = 1.0; } else { const double s = sqrt(e2); rho0 = 2 * s * delta - d2; rho1 = delta / s; } };
        auto active_chi2 = [&](const Pose& P) { double chi = 0; for (Edge& e : edges) { if (e.level != 0) continue; project_error(e, P); const double e2 = e.chi2();
                                                 if (e.robust) { double r0, r1; huber(e2, r0, r1); chi += r0; } else chi += e2; } return chi; };
        auto build = [&](const Pose& P, double* H, double* b) {
            for (int k = 0; k < 36; k++) H[k] = 0;
            for (int k = 0; k < 6; k++) b[k] = 0;
            for (const Edge& e : edges) {
                if (e.level != 0) continue;
                double p[3]; P.map(e.X, p);
                const double x = p[0], y = p[1], iz = 1.0 / p[2], iz2 = iz * iz;
                const double J[2][6] = {{x * y * iz2 * camera.fx, -(1 + (x * x * iz2)) * camera.fx, y * iz * camera.fx, -iz * camera.fx, 0, x * iz2 * camera.fx},
                                        {(1 + y * y * iz2) * camera.fy, -x * y * iz2 * camera.fy, -x * iz * camera.fy, 0, -iz * camera.fy, y * iz2 * camera.fy}};
                double w = 1.0;
                if (e.robust) { double r0; huber(e.chi2(), r0, w); }
                const double er[2] = {e.e0, e.e1};
                for (int r = 0; r < 2; r++) {
                    const double wr = -er[r] * w;
                    for (int a = 0; a < 6; a++) { b[a] += J[r][a] * wr; for (int c = 0; c < 6; c++) H[6 * a + c] += J[r][a] * w * J[r][c]; }
                }
            }
        };
        auto optimize = [&](Pose& P, int iterations) {                      // SparseOptimizer::optimize with OptimizationAlgorithmLevenberg
            bool any = false; for (const Edge& e : edges) any = any || e.level == 0;
            if (!any) return;
            double lambda = 0, nu = 2;
            for (int it = 0; it < iterations; it++) {
                double chi = active_chi2(P), chi_new = chi, H[36], b[6];
                build(P, H, b);
                if (it == 0) { double mx = 0; for (int j = 0; j < 6; j++) mx = max(mx, fabs(H[7 * j])); lambda = 1e-5 * mx; nu = 2; }
                double gain = 0; int trials = 0;
                do {
                    const Pose saved = P;
                    double x[6] = {0, 0, 0, 0, 0, 0};
                    const bool ok = solveLDLT(H, lambda, b, x);
                    P.oplus(x);
                    chi_new = active_chi2(P);
                    if (!ok) chi_new = numeric_limits<double>::max();
                    gain = chi - chi_new;
                    double scale = 0; for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                    scale += 1e-3; gain /= scale;
                    if (gain > 0 && std::isfinite(chi_new)) {
                        double alpha = 1. - pow((2 * gain - 1), 3); alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                        lambda *= alpha > 1. / 3. ? alpha : 1. / 3.; nu = 2; chi = chi_new;
                    } else { lambda *= nu; nu *= 2; P = saved; if (!std::isfinite(lambda)) break; }
                    trials++;
                } while (gain < 0 && trials < 10);
                if (trials == 10 || gain == 0) break;
            }
            active_chi2(P);
        };
        for (size_t it = 0; it < 4; it++) {
            est = init;                                                         // pnp.cpp:66: every round starts from the caller's transform
            optimize(est, 10);
            for (size_t i = 0; i < edges.size(); i++) {
                Edge& e = edges[i];
                if (inliers[e.id] == true) project_error(e, est);
                if (e.chi2() > 5.991) { inliers[e.id] = false; e.level = 1; good--; }
                else { inliers[i] = true; e.level = 0; }                        // [i], not [e.id]: as written at pnp.cpp:87
                if (it == 2) e.robust = false;
            }
            if (good < 5) break;
        }
        for (size_t i = 0; i < inliers.size(); i++) if (inliers[i]) inliersIndex.push_back((int)i);
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) transform(r, c) = est.R[3 * r + c]; transform(r, 3) = est.t[r]; }
        return (int)inliers.size() > min_inliers;                               // the vector's LENGTH, pnp.cpp:115
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
    struct Pose {                                                               // x_cam = R X + t, R row-major
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
        void map(const double* X, double* p) const { for (int r = 0; r < 3; r++) p[r] = R[3 * r] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2] + t[r]; }
        static void mul(const double* A, const double* B, double* C) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c]; }
        void oplus(const double* d) {                                           // T <- exp(d) T, d = (omega, upsilon): g2o::SE3Quat::exp
            const double th = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            const double W[9] = {0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0};
            double W2[9], dR[9], V[9]; mul(W, W, W2);
            if (th < 0.00001) { for (int k = 0; k < 9; k++) { dR[k] = (k % 4 == 0 ? 1.0 : 0.0) + W[k] + W2[k]; V[k] = dR[k]; } }
            else {
                const double a = sin(th) / th, b = (1 - cos(th)) / (th * th), c = (th - sin(th)) / (th * th * th);
                for (int k = 0; k < 9; k++) { dR[k] = (k % 4 == 0 ? 1.0 : 0.0) + a * W[k] + b * W2[k]; V[k] = (k % 4 == 0 ? 1.0 : 0.0) + b * W[k] + c * W2[k]; }
            }
            double nR[9], nt[3]; mul(dR, R, nR);
            for (int r = 0; r < 3; r++) { const double vt = V[3 * r] * d[3] + V[3 * r + 1] * d[4] + V[3 * r + 2] * d[5]; nt[r] = dR[3 * r] * t[0] + dR[3 * r + 1] * t[1] + dR[3 * r + 2] * t[2] + vt; }
            for (int k = 0; k < 9; k++) R[k] = nR[k];
            for (int k = 0; k < 3; k++) t[k] = nt[k];
        }
    };
    static bool solveLDLT(const double* Hin, double lambda, const double* b, double* x) {     // (H + lambda I) x = b, un-pivoted L D L^T
        double A[36], L[36] = {0}, D[6], y[6];
        for (int k = 0; k < 36; k++) A[k] = Hin[k];
        for (int i = 0; i < 6; i++) A[7 * i] += lambda;
        for (int j = 0; j < 6; j++) {
            double d = A[6 * j + j]; for (int k = 0; k < j; k++) d -= L[6 * j + k] * L[6 * j + k] * D[k];
            if (!(d > 0)) return false;
            D[j] = d; L[6 * j + j] = 1.0;
            for (int i = j + 1; i < 6; i++) { double s = A[6 * i + j]; for (int k = 0; k < j; k++) s -= L[6 * i + k] * L[6 * j + k] * D[k]; L[6 * i + j] = s / d; }
        }
        for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[6 * i + k] * y[k]; y[i] = s; }
        for (int i = 0; i < 6; i++) y[i] /= D[i];
        for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[6 * k + i] * x[k]; x[i] = s; }
        return true;
    }
    const ParameterReader& parameterReader;
    const OrbFeature& orb;
    int min_inliers = 10, min_match = 30;
};
}  // namespace rgbd_tutor
