// ssm/orb.h -- rgbd_tutor::OrbFeature (reference include/orb.h:16-64, src/orb.cpp:16-29): same constructor, same two
// methods.  The ORB_SLAM2 extractor + cv::BFMatcher pair is replaced by libssm_hip.so (ssm_orb_extract / ssm_match).
#pragma once
#include "common_headers.h"
#include <deque>
#include "device.h"
#include "rgbdframe.h"
namespace rgbd_tutor {
class OrbFeature {
public:
    OrbFeature(const ParameterReader& para) : parameterReader(para) { knn_match_ratio = para.getData<double>("knn_match_ratio", 0.8); }
    // extract features into frame->features (gray conversion, ORB, 3-D position of every keypoint): orb.h:32-53
    void detectFeatures(RGBDFrame::Ptr& frame) const {
        ssm::Device& d = device(frame->rgb.cols, frame->rgb.rows);
        const int cap = ssm_orb_capacity(d.ctx());
        vector<ssm_keypoint> kps(cap); vector<uint8_t> desc((size_t)cap * 32); vector<float> pos((size_t)cap * 3);
        int n = 0;
        const bool has_depth = !frame->depth.empty() && frame->depth.isContinuous();
        d.check(ssm_orb_extract(d.ctx(), frame->rgb.data, frame->rgb.cols, frame->rgb.rows, (int)frame->rgb.step, frame->rgb.channels(),
                                has_depth ? frame->depth.ptr<uint16_t>() : nullptr, kps.data(), desc.data(), pos.data(), cap, &n), "ssm_orb_extract");
        // one block for the frame's descriptors; Feature::descriptor of feature i is row i of it (cv::Mat::row shares the storage, as the extractor's output Mat does
        // in the reference: orb.h:44-49 takes rows of `descriptors`)
        cv::Mat all; if (n > 0) { all.create(n, 32, CV_8UC1); memcpy(all.data, desc.data(), (size_t)n * 32); }
        frame->features.reserve(frame->features.size() + (size_t)n);
        for (int i = 0; i < n; i++) {
            Feature f;
            static_assert(sizeof(cv::KeyPoint) == sizeof(ssm_keypoint), "cv::KeyPoint layout");
            memcpy((void*)&f.keypoint, &kps[i], sizeof(ssm_keypoint));
            f.descriptor = all.row(i);
            f.position = cv::Point3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);     // == frame->project2dTo3d(int(pt.x), int(pt.y)), orb.h:50
            frame->features.push_back(std::move(f));
        }
        if ((int)frame->features.size() == n) frame->descriptors_all = all;
    }
    // descriptors of frame1 = query, frame2 = train; knn(2) + ratio test (src/orb.cpp:16-29)
    vector<cv::DMatch> match(const RGBDFrame::Ptr& frame1, const RGBDFrame::Ptr& frame2) const {
        vector<cv::DMatch> matches;
        cv::Mat d1 = frame1->getAllDescriptors(), d2 = frame2->getAllDescriptors();
        if (d1.rows == 0 || d2.rows < 2) return matches;      // knnMatch(k = 2) on fewer than two train descriptors: the reference indexes [1] unguarded (src/orb.cpp:25, UB); no matches here
        ssm::Device& d = device(frame2->rgb.cols, frame2->rgb.rows);
        matches.resize(d1.rows);
        int n = 0;
        static_assert(sizeof(cv::DMatch) == sizeof(ssm_dmatch), "cv::DMatch layout");
        d.check(ssm_match(d.ctx(), d1.data, d1.rows, d2.data, d2.rows, knn_match_ratio, reinterpret_cast<ssm_dmatch*>(matches.data()), d1.rows, &n), "ssm_match");
        matches.resize(n);
        return matches;
    }
    // match(ref, frame) for every reference frame, as Tracker::trackRefFrame's loop makes them (src/track.cpp:150-152), in ONE device call (ssm_match_refs:
    // one upload, one matrix-core launch for all pairs, one download).  Same lists as match().
    vector<vector<cv::DMatch>> matchMany(const std::deque<RGBDFrame::Ptr>& refs, const RGBDFrame::Ptr& frame) const {
        vector<vector<cv::DMatch>> all(refs.size());
        cv::Mat d2 = frame->getAllDescriptors();
        if (d2.rows < 2 || refs.empty()) return all;
        ssm::Device& d = device(frame->rgb.cols, frame->rgb.rows);
        const size_t k = refs.size();
        vector<cv::Mat> d1(k); vector<const uint8_t*> pr(k); vector<int> nr(k), caps(k), n(k, 0); vector<ssm_dmatch*> po(k);
        for (size_t i = 0; i < k; i++) {
            d1[i] = refs[i]->getAllDescriptors();
            all[i].resize(d1[i].rows);
            pr[i] = d1[i].data; nr[i] = d1[i].rows; caps[i] = d1[i].rows; po[i] = reinterpret_cast<ssm_dmatch*>(all[i].data());
        }
        d.check(ssm_match_refs(d.ctx(), pr.data(), nr.data(), (int)k, d2.data, d2.rows, knn_match_ratio, po.data(), caps.data(), n.data()), "ssm_match_refs");
        for (size_t i = 0; i < k; i++) all[i].resize(n[i]);
        return all;
    }
    ssm::Device& device(int w, int h) const {     // one context per calling thread and frame geometry
        thread_local map<pair<const void*, pair<int, int>>, unique_ptr<ssm::Device>> devs;
        auto key = make_pair((const void*)this, make_pair(w, h));
        auto it = devs.find(key);
        if (it == devs.end()) it = devs.emplace(key, unique_ptr<ssm::Device>(new ssm::Device(parameterReader.deviceConfig(w, h)))).first;
        lastDevice() = it->second.get();
        return *it->second;
    }
    // the context this thread used last (nullptr before its first detectFeatures / match): PnPSolver::solvePnP runs on it
    static ssm::Device*& lastDevice() { thread_local ssm::Device* d = nullptr; return d; }
protected:
    const ParameterReader& parameterReader;
    double knn_match_ratio = 0.8;
};
}  // namespace rgbd_tutor
