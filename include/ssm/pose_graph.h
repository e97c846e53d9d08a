// ssm/pose_graph.h -- the slice of rgbd_tutor::PoseGraph the mapping path touches (reference include/pose_graph.h,
// src/pose_graph.cpp:11-77): the keyframe gate and the shared `keyframes` list.  The g2o graph, the optimiser thread
// and loop closure are out of scope (SURVEY.md s.2 #11,#12); shutdown() only sets the flag Mapper::viewer reads.
#pragma once
#include "common_headers.h"
#include <atomic>
#include "track.h"
namespace rgbd_tutor {
class PoseGraph {
public:
    PoseGraph(const ParameterReader& para, shared_ptr<Tracker>& t) : parameterReader(para), tracker(t) {
        keyframe_min_translation = para.getData<double>("keyframe_min_translation", 5.5);
        keyframe_min_rotation = para.getData<double>("keyframe_min_rotation", 2.5);
    }
    // pose_graph.cpp:11-77: first frame always; afterwards when the motion w.r.t. the last keyframe is large enough
    bool tryInsertKeyFrame(RGBDFrame::Ptr& frame) {
        if (keyframes.size() == 0) { unique_lock<mutex> lck(keyframes_mutex); keyframes.push_back(frame); refFrame = frame; return true; }
        Eigen::Isometry3d delta = frame->getTransform().inverse() * refFrame->getTransform();
        if (norm_translate(delta) > keyframe_min_translation || norm_rotate(delta) > keyframe_min_rotation) {
            unique_lock<mutex> lck(keyframes_mutex);
            keyframes.push_back(frame); refFrame = frame;
            return true;
        }
        return false;
    }
    void shutdown() { shutDownFlag = true; }
    vector<RGBDFrame::Ptr> keyframes;
    mutex keyframes_mutex;
    std::atomic<bool> shutDownFlag{false};          // read by Mapper::viewer on its thread
protected:
    const ParameterReader& parameterReader;
    shared_ptr<Tracker> tracker;
    RGBDFrame::Ptr refFrame;
    double keyframe_min_translation = 0.3, keyframe_min_rotation = 0.3;
};
}  // namespace rgbd_tutor
