// ssm/quadmatcher.hpp -- QuadFeatureMatch (reference include/quadmatcher.hpp:51-136, src/quadmatcher.cpp): the stereo
// quad matcher Tracker::estimateVO builds per frame (src/track.cpp:45-55): same constructor, init / detectFeature /
// extractDescriptor / circularMatching, public `quadmatches`.  The tracking branch (mode_track = true, the one the
// reference uses) runs GFTT + 4x pyramidal LK + filteringTracks on the GPU through ssm_quad_track.  The matching
// branch (mode_track = false) keeps the reference's structure -- detect x4, describe x4, windowed NN x3, chain -- with
// the GPU ORB extractor as detector+descriptor (DES_ORB) and ssm_window_match for QuadFeatureMatch::matching; the
// other OpenCV detectors/descriptors of init() (SIFT, SURF, STAR, BRISK ...) are not rebuilt.
#pragma once
#include "common_headers.h"
#include "device.h"
#include <map>
#include <tuple>
using cv::KeyPoint; using cv::DMatch; using cv::Point2f; using cv::Mat;
enum { DET_FAST, DET_STAR, DET_ORB, DET_SIFT, DET_SURF, DET_GFTT, DET_STAR_ADAPT, DET_FAST_ADAPT, DET_FAST_GRID, DET_STAR_GRID, DET_GFTT_GRID };
enum { DES_SIFT, DES_SURF, DES_BRISK, DES_FREAK, DES_ORB };
struct pmatch {                        // quadmatcher.hpp:33-49, layout == ssm_pmatch
    float u1p, v1p; int32_t i1p; float u2p, v2p; int32_t i2p; float u1c, v1c; int32_t i1c; float u2c, v2c; int32_t i2c; short dis_c, dis_p;
    pmatch() {}
};
static_assert(sizeof(pmatch) == sizeof(ssm_pmatch), "pmatch layout");
class QuadFeatureMatch {
public:
    QuadFeatureMatch() {}
    QuadFeatureMatch(cv::Mat& img_lc_, cv::Mat& img_rc_, cv::Mat& img_lp_, cv::Mat& img_rp_, cv::Mat& img_s_rc_, cv::Mat& img_s_rp_, bool mode_track_)
        : img_lc(img_lc_), img_lp(img_lp_), img_rc(img_rc_), img_rp(img_rp_), img_s_rc(img_s_rc_), img_s_rp(img_s_rp_), mode_track(mode_track_) {}
    void init(int detector_type, int descriptor_type) {
        if (mode_track && detector_type != DET_GFTT) throw std::invalid_argument("QuadFeatureMatch: tracking mode is built for DET_GFTT (the reference's choice, track.cpp:52)");
        if (!mode_track && (detector_type != DET_ORB || descriptor_type != DES_ORB)) throw std::invalid_argument("QuadFeatureMatch: matching mode is built for DET_ORB/DES_ORB only");
        descriptor_binary = true; distance_threshold = 80.0f;                // quadmatcher.cpp init(): ORB -> binary, threshold 80
        ssm_config cfg; ssm_config_default(&cfg); cfg.width = img_lc.cols < 64 ? 64 : img_lc.cols; cfg.height = img_lc.rows < 64 ? 64 : img_lc.rows;
        cfg.orb_features = 1000; cfg.max_batch = 1; cfg.voxel_capacity_log2 = 10;
        if (mode_track) cfg.orb_levels = 1;                                  // no ORB runs in tracking mode: the smallest pyramid passes ssm_create's geometry check
        else while (cfg.orb_levels > 1) { ssm_ctx* t = nullptr; if (ssm_create(0, &cfg, &t) == SSM_OK) { ssm_destroy(t); break; } cfg.orb_levels--; }
        dev = sharedDevice(cfg, mode_track);
    }
    void detectFeature() { /* tracking mode: GFTT runs inside circularMatching (one device call); matching mode: extractDescriptor() detects */ }
    void extractDescriptor() {
        const cv::Mat* ims[4] = {&img_lc, &img_rc, &img_lp, &img_rp};
        vector<KeyPoint>* kps[4] = {&keypoint_lc, &keypoint_rc, &keypoint_lp, &keypoint_rp};
        cv::Mat* des[4] = {&descriptor_lc, &descriptor_rc, &descriptor_lp, &descriptor_rp};
        const int cap = ssm_orb_capacity(dev->ctx());
        for (int i = 0; i < 4; i++) {
            vector<ssm_keypoint> k(cap); des[i]->create(cap, 32, CV_8UC1); int n = 0;
            dev->check(ssm_orb_extract(dev->ctx(), ims[i]->data, ims[i]->cols, ims[i]->rows, (int)ims[i]->step, ims[i]->channels(), nullptr, k.data(), des[i]->data, nullptr, cap, &n), "ssm_orb_extract");
            kps[i]->resize(n); if (n) memcpy((void*)kps[i]->data(), k.data(), sizeof(ssm_keypoint) * n);
            des[i]->rows = n;
        }
    }
    void circularMatching() {
        quadmatches.clear();
        if (mode_track) {                                                    // quadmatcher.cpp:550-588
            vector<pmatch> out(1000); int n = 0;
            dev->check(ssm_quad_track(dev->ctx(), img_lc.data, img_rc.data, img_lp.data, img_rp.data, img_lc.cols, img_lc.rows, (int)img_lc.step, 1000,
                                      reinterpret_cast<ssm_pmatch*>(out.data()), (int)out.size(), &n), "ssm_quad_track");
            out.resize(n); quadmatches.swap(out);
            return;
        }
        extractDescriptor();                                                 // quadmatcher.cpp:591-661
        vector<DMatch> m_lrc, m_rcp, m_rlp;
        matching(keypoint_lc, descriptor_lc, keypoint_rc, descriptor_rc, 20, 2, m_lrc);
        matching(keypoint_rc, descriptor_rc, keypoint_rp, descriptor_rp, 20, 20, m_rcp);
        matching(keypoint_rp, descriptor_rp, keypoint_lp, descriptor_lp, 20, 2, m_rlp);
        const int min_disparity = 3, max_delta_x = 2;
        for (int i = 0; i < (int)keypoint_lc.size(); i++) {
            const int id_rc = m_lrc[i].trainIdx; if (!(id_rc > 0)) continue;  // index 0 counts as "unmatched" in the reference (:622-630)
            const int id_rp = m_rcp[id_rc].trainIdx; if (!(id_rp > 0)) continue;
            const int id_lp = m_rlp[id_rp].trainIdx; if (!(id_lp > 0)) continue;
            pmatch t; memset((void*)&t, 0, sizeof(t));
            t.u1c = keypoint_lc[i].pt.x; t.v1c = keypoint_lc[i].pt.y; t.i1c = i;
            t.u2c = keypoint_rc[id_rc].pt.x; t.v2c = keypoint_rc[id_rc].pt.y; t.i2c = id_rc;
            t.u2p = keypoint_rp[id_rp].pt.x; t.v2p = keypoint_rp[id_rp].pt.y; t.i2p = id_rp;
            t.u1p = keypoint_lp[id_lp].pt.x; t.v1p = keypoint_lp[id_lp].pt.y; t.i1p = id_lp;
            const int delta_x = (int)std::abs(std::abs(t.u1c - t.u1p) - std::abs(t.u2c - t.u2p));
            const int disparity = (int)std::abs(t.u1c - t.u2c);
            if (delta_x < max_delta_x && disparity > min_disparity) quadmatches.push_back(t);
        }
    }
    vector<pmatch> quadmatches;
    ssm_ctx* deviceContext() const { return dev ? dev->ctx() : nullptr; }       // shared with VisualOdometryStereo::Process
private:
    void matching(vector<KeyPoint>& k1, cv::Mat& d1, vector<KeyPoint>& k2, cv::Mat& d2, int sw, int sh, vector<DMatch>& matches) {   // :41-83
        vector<float> p1(2 * k1.size()), p2(2 * k2.size());
        for (size_t i = 0; i < k1.size(); i++) { p1[2*i] = k1[i].pt.x; p1[2*i+1] = k1[i].pt.y; }
        for (size_t i = 0; i < k2.size(); i++) { p2[2*i] = k2[i].pt.x; p2[2*i+1] = k2[i].pt.y; }
        matches.resize(k1.size());
        static_assert(sizeof(DMatch) == sizeof(ssm_dmatch), "DMatch layout");
        dev->check(ssm_window_match(dev->ctx(), p1.data(), d1.data, (int)k1.size(), p2.data(), d2.data, (int)k2.size(), sw, sh, distance_threshold,
                                    reinterpret_cast<ssm_dmatch*>(matches.data())), "ssm_window_match");
    }
    cv::Mat img_lc, img_lp, img_rc, img_rp, img_s_rc, img_s_rp;
    vector<KeyPoint> keypoint_lc, keypoint_rc, keypoint_lp, keypoint_rp;
    cv::Mat descriptor_lc, descriptor_rc, descriptor_lp, descriptor_rp;
    bool mode_track = true, descriptor_binary = true;
    float distance_threshold = 80.0f;
    shared_ptr<ssm::Device> dev;
    // Tracker::estimateVO news a QuadFeatureMatch per frame (track.cpp:45): the device context behind it is kept per thread and
    // image size instead of being rebuilt every frame
    static shared_ptr<ssm::Device> sharedDevice(const ssm_config& cfg, bool track) {
        static thread_local std::map<std::tuple<int, int, int, bool>, shared_ptr<ssm::Device>> cache;
        auto key = std::make_tuple(cfg.width, cfg.height, cfg.orb_levels, track);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
        shared_ptr<ssm::Device> d(new ssm::Device(cfg));
        cache[key] = d; return d;
    }
};
