// ssm/vo_stereo.hpp -- VisualOdometry / VisualOdometryStereo (reference include/vo.hpp, include/vo_stereo.hpp,
// src/vo.cpp, src/vo_stereo.cpp): ego-motion from the quad matches of QuadFeatureMatch, the consumer Tracker::estimateVO
// calls right after circularMatching (src/track.cpp:45-66).  Same public surface: the parameter blocks, Process(),
// getMotion() (4x4 CV_64F, previous -> current camera), getNumberOfMatches/Inliers, getInlierIndices, the public
// quadmatches / quadmatches_inlier / quadmatches_outlier vectors.
// The numerical work (200 three-point Gauss-Newton hypotheses, consensus, refinement) runs on the GPU through
// ssm_vo_estimate; this class keeps what is host state in the reference too: the rand() stream that
// VisualOdometry::getRandomSample draws from (srand(0) in the constructor, src/vo.cpp:17; restated here as glibc's
// generator so the stream is private to the object instead of process-global) and the result bookkeeping.
// Contracts and their deviations from a libm/OpenCV build: oracle/vo.c header.
#pragma once
#include "common_headers.h"
#include "quadmatcher.hpp"
class VisualOdometry {
public:
    struct calibration { double f = 1, cu = 0, cv = 0; };
    struct parameters { calibration calib; };
    explicit VisualOdometry(parameters p) : param_(p) { Tr_delta = eye4(); seed(0); }
    virtual ~VisualOdometry() {}
    cv::Mat getMotion() { return Tr_delta; }
    int getNumberOfMatches() { return (int)quadmatches.size(); }
    int getNumberOfInliers() { return (int)inliers.size(); }
    std::vector<int> getInlierIndices() { return inliers; }
    std::vector<pmatch> quadmatches, quadmatches_inlier, quadmatches_outlier;
    // the rand() stream as an object (the bulk path, ssm_stereo_seq_process, is handed the RAW draws this object would make and reports how many it used:
    // include/ssm/batch_stereo_tracker.h)
    struct RandState { int32_t r[31]; int f, b; };
    RandState saveRand() const { RandState s; for (int i = 0; i < 31; i++) s.r[i] = r_[i]; s.f = f_; s.b = b_; return s; }
    void restoreRand(const RandState& s) { for (int i = 0; i < 31; i++) r_[i] = s.r[i]; f_ = s.f; b_ = s.b; }
    uint32_t rawRand() { return next_rand(); }
protected:
    bool updateMotion() {                                               // vo.cpp:23-38
        std::vector<double> tr = estimateMotion(quadmatches);
        if (tr.size() != 6) return false;
        Tr_delta = transformationVectorToMatrix(tr);
        return true;
    }
    // vo.cpp:40-72, with the sin/cos contract of oracle/vo.c (sso_sincos64)
    static cv::Mat transformationVectorToMatrix(const std::vector<double>& tr) {
        double sx, cx, sy, cy, sz, cz;
        sincos64(tr[0], sx, cx); sincos64(tr[1], sy, cy); sincos64(tr[2], sz, cz);
        cv::Mat T = eye4();
        T.at<double>(0, 0) = +cy*cz;          T.at<double>(0, 1) = -cy*sz;          T.at<double>(0, 2) = +sy;    T.at<double>(0, 3) = tr[3];
        T.at<double>(1, 0) = +sx*sy*cz+cx*sz; T.at<double>(1, 1) = -sx*sy*sz+cx*cz; T.at<double>(1, 2) = -sx*cy; T.at<double>(1, 3) = tr[4];
        T.at<double>(2, 0) = -cx*sy*cz+sx*sz; T.at<double>(2, 1) = +cx*sy*sz+sx*cz; T.at<double>(2, 2) = +cx*cy; T.at<double>(2, 3) = tr[5];
        return T;
    }
    virtual std::vector<double> estimateMotion(std::vector<pmatch>& quadmatches) = 0;
    // vo.cpp:74-93: num distinct indices of 0..N-1, each rand() % (what is left), erased from the pool
    std::vector<int> getRandomSample(int N, int num) {
        std::vector<int> sample, totalset(N);
        for (int i = 0; i < N; i++) totalset[i] = i;
        for (int i = 0; i < num; i++) {
            const int j = (int)(next_rand() % (uint32_t)totalset.size());
            sample.push_back(totalset[j]);
            totalset.erase(totalset.begin() + j);
        }
        return sample;
    }
    cv::Mat Tr_delta;
    std::vector<int> inliers, outliers;
private:
    static cv::Mat eye4() { cv::Mat m(4, 4, CV_64F); for (int i = 0; i < 4; i++) m.at<double>(i, i) = 1.0; return m; }
    static void sincos64(double x, double& s, double& c) {
        const double fn = std::nearbyint(x * 6.36619772367581382433e-01);
        double r = x - fn * 1.57079632673412561417e+00; r = r - fn * 6.07710050650619224932e-11; r = r - fn * 2.02226624879595063154e-21;
        const double z = r * r;
        const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
        const double sr = r + (r * z) * (-1.66666666666666324348e-01 + z * ps);
        const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
        const double cr = (1.0 - 0.5 * z) + (z * z) * pc;
        switch ((int)((long long)fn & 3)) { case 0: s = sr; c = cr; break; case 1: s = cr; c = -sr; break; case 2: s = -sr; c = -cr; break; default: s = -cr; c = sr; }
    }
    // glibc srand()/rand(): TYPE_3 additive feedback generator (r[i] = r[i-3] + r[i-31], output >> 1)
    void seed(unsigned s) {
        if (s == 0) s = 1;
        r_[0] = (int32_t)s;
        for (int i = 1; i < 31; i++) {
            const long hi = r_[i - 1] / 127773, lo = r_[i - 1] % 127773;
            long w = 16807 * lo - 2836 * hi; if (w < 0) w += 2147483647;
            r_[i] = (int32_t)w;
        }
        f_ = 3; b_ = 0;
        for (int i = 0; i < 310; i++) next_rand();
    }
    uint32_t next_rand() {
        uint32_t* r = reinterpret_cast<uint32_t*>(r_);
        r[f_] += r[b_];
        const uint32_t out = r[f_] >> 1;
        if (++f_ >= 31) f_ = 0;
        if (++b_ >= 31) b_ = 0;
        return out;
    }
    parameters param_;
    int32_t r_[31]; int f_ = 3, b_ = 0;
};
class VisualOdometryStereo : public VisualOdometry {
public:
    struct parameters : public VisualOdometry::parameters {
        double base = 1.0; int ransac_iters = 200; double inlier_threshold = 1.1f; bool reweighting = true;      // vo_stereo.hpp:31-37
    };
    explicit VisualOdometryStereo(parameters p) : VisualOdometry(p), param(p) {}
    // copies the matches of the quad matcher and estimates the motion (vo_stereo.cpp:18-43); false: no estimate
    bool Process(QuadFeatureMatch& quadmatcher) {
        quadmatches.clear();
        for (const pmatch& q : quadmatcher.quadmatches) {
            pmatch t; memset(static_cast<void*>(&t), 0, sizeof(t));
            t.u1c = q.u1c; t.v1c = q.v1c; t.u2c = q.u2c; t.v2c = q.v2c; t.u1p = q.u1p; t.v1p = q.v1p; t.u2p = q.u2p; t.v2p = q.v2p;
            quadmatches.push_back(t);
        }
        ctx_ = quadmatcher.deviceContext();                             // the matcher's device context, when it has one
        return updateMotion();
    }
    // the same on a list of quad matches (u/v fields only), on the given device context
    bool ProcessMatches(const std::vector<pmatch>& qm, ssm_ctx* ctx) { quadmatches = qm; ctx_ = ctx; return updateMotion(); }
    const parameters& stereoParameters() const { return param; }
    static cv::Mat motionMatrix(const double tr[6]) { return transformationVectorToMatrix(std::vector<double>(tr, tr + 6)); }
private:
    std::vector<double> estimateMotion(std::vector<pmatch>& qm) override {
        const int N = (int)qm.size();
        inliers.clear(); outliers.clear(); quadmatches_inlier.clear(); quadmatches_outlier.clear();
        if (N < 6) return std::vector<double>();
        std::vector<int32_t> samples;                                   // the reference draws inside the loop: same stream, same order
        for (int k = 0; k < param.ransac_iters; k++) for (int v : getRandomSample(N, 3)) samples.push_back(v);
        if (!ctx_) {
            if (!own_) { ssm_config cfg; ssm_config_default(&cfg); cfg.width = 128; cfg.height = 128; cfg.orb_levels = 1; cfg.orb_features = 100; cfg.max_batch = 1; cfg.voxel_capacity_log2 = 10;   // smallest context (no images pass through it)
                         own_.reset(new ssm::Device(cfg)); }
            ctx_ = own_->ctx();
        }
        ssm_vo_params P; P.f = param.calib.f; P.cu = param.calib.cu; P.cv = param.calib.cv; P.base = param.base;
        P.inlier_threshold = param.inlier_threshold; P.reweighting = param.reweighting ? 1 : 0; P.pad = 0;
        double tr[6]; std::vector<int32_t> inl(N); int n_inl = 0, ok = 0;
        const int rc = ssm_vo_estimate(ctx_, reinterpret_cast<const ssm_pmatch*>(qm.data()), N, &P, samples.data(), param.ransac_iters, tr, inl.data(), N, &n_inl, &ok);
        if (rc != SSM_OK) throw ssm::DeviceError(rc, std::string("ssm_vo_estimate: ") + ssm_last_error(ctx_));
        inliers.assign(inl.begin(), inl.begin() + n_inl);
        // getInOutMatches (vo_stereo.cpp:182-201): inliers is in increasing index order
        size_t p = 0;
        for (int i = 0; i < N; i++) {
            if (p < inliers.size() && inliers[p] == i) { quadmatches_inlier.push_back(qm[i]); p++; }
            else { quadmatches_outlier.push_back(qm[i]); outliers.push_back(i); }
        }
        return ok ? std::vector<double>(tr, tr + 6) : std::vector<double>();
    }
    parameters param;
    ssm_ctx* ctx_ = nullptr;
    std::unique_ptr<ssm::Device> own_;
};
