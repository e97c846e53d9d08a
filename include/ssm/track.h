// ssm/track.h -- rgbd_tutor::Tracker (reference include/track.h:50-191, src/track.cpp): the per-frame state machine.
// RGB-D mode = Tracker::trackRefFrame (src/track.cpp:140-200), restated line for line on the device-backed
// OrbFeature; the checked-in reference instead calls the stereo estimateVO() (track.cpp:19).  Its two device stages exist
// (QuadFeatureMatch, VisualOdometryStereo: quadmatcher.hpp, vo_stereo.hpp) but the SGBM depth / UVDisparity steps between
// them (track.cpp:68-80) are "next" rows, so tracker_mode=stereo is still rejected here.  PoseGraph is out of scope:
// setPoseGraph is kept as a no-op hook.
#pragma once
#include "common_headers.h"
#include "orb.h"
#include "pnp.h"
#include "vo_stereo.hpp"
namespace rgbd_tutor {
class PoseGraph;
class Tracker {
public:
    typedef shared_ptr<Tracker> Ptr;
    enum trackerState { NOT_READY = 0, OK, LOST };
    Tracker(const ParameterReader& para, VisualOdometryStereo::parameters param) : parameterReader(para), voparam(param) {
        orb = make_shared<OrbFeature>(para);
        pnp = make_shared<PnPSolver>(para, *orb);
        max_lost_frame = para.getData<int>("tracker_max_lost_frame", 10);
        refFramesSize = para.getData<int>("tracker_ref_frames", 5);
        if (para.getData<string>("tracker_mode", string("rgbd")) != "rgbd")
            throw invalid_argument("tracker_mode: only 'rgbd' (Tracker::trackRefFrame) is built; the stereo estimateVO path is a next row");
    }
    void setPoseGraph(shared_ptr<PoseGraph> pg) { poseGraph = pg; }
    // put in a new frame, returns its pose (src/track.cpp:8-28)
    Eigen::Isometry3d updateFrame(RGBDFrame::Ptr& newFrame) {
        unique_lock<mutex> lck(adjustMutex);
        currentFrame = newFrame;
        if (state == NOT_READY) { initFirstFrame(); return Eigen::Isometry3d::Identity(); }
        if (state == OK) { trackRefFrame(); return currentFrame->getTransform(); }
        lostRecover();
        return currentFrame->getTransform();
    }
    trackerState getState() const { return state; }
    // re-anchor the current frame on `ref` (called by the pose graph after optimisation), track.h:114-131
    bool adjust(const RGBDFrame::Ptr& ref) {
        unique_lock<mutex> lck(adjustMutex);
        PNP_INFORMATION info;
        if (pnp->solvePnPLazy(ref, currentFrame, info)) {
            currentFrame->setTransform(info.T * ref->getTransform());
            refFrames.clear(); refFrames.push_back(ref); cntLost = 0; state = OK;
            return true;
        }
        return false;
    }
    const deque<RGBDFrame::Ptr>& referenceFrames() const { return refFrames; }
    shared_ptr<OrbFeature> orbFeature() const { return orb; }
protected:
    void initFirstFrame() {                                             // track.cpp:30-36
        orb->detectFeatures(currentFrame);
        refFrames.push_back(currentFrame);
        speed = Eigen::Isometry3d::Identity();
        state = OK;
    }
    void trackRefFrame() {                                              // track.cpp:140-200
        currentFrame->setTransform(speed * refFrames.back()->getTransform());
        orb->detectFeatures(currentFrame);
        vector<cv::Point3f> obj; vector<cv::Point2f> img;
        for (auto pFrame : refFrames) {
            vector<cv::DMatch> matches = orb->match(pFrame, currentFrame);
            Eigen::Isometry3d invPose = pFrame->getTransform().inverse();
            for (auto m : matches) {
                cv::Point3f pObj = pFrame->features[m.queryIdx].position;
                if (pObj == cv::Point3f(0, 0, 0)) continue;
                Eigen::Vector4d vec = invPose * Eigen::Vector4d(pObj.x, pObj.y, pObj.z, 1);
                obj.push_back(cv::Point3f((float)vec(0), (float)vec(1), (float)vec(2)));
                img.push_back(currentFrame->features[m.trainIdx].keypoint.pt);
            }
        }
        lastMatches = (int)img.size();
        if (img.size() < 15) { cntLost++; if (cntLost > max_lost_frame) state = LOST; return; }
        vector<int> inlierIndex;
        Eigen::Isometry3d T = speed * lastPose;
        pnp->solvePnP(img, obj, currentFrame->camera, inlierIndex, T);
        lastInliers = (int)inlierIndex.size();
        if (inlierIndex.size() < 15) { cntLost++; if (cntLost > max_lost_frame) state = LOST; return; }
        currentFrame->setTransform(T);
        cntLost = 0;
        speed = T * lastPose.inverse();
        lastPose = currentFrame->getTransform();
        refFrames.push_back(currentFrame);
        while ((int)refFrames.size() > refFramesSize) refFrames.pop_front();
    }
    void lostRecover() {                                                // track.cpp:202-212
        cout << "trying to recover from lost" << endl;
        orb->detectFeatures(currentFrame);
        currentFrame->setTransform(refFrames.back()->getTransform());
        refFrames.clear(); refFrames.push_back(currentFrame);
        state = OK; cntLost = 0;
    }
public:
    int lastMatches = 0, lastInliers = 0;
protected:
    const ParameterReader& parameterReader;
    VisualOdometryStereo::parameters voparam;
    RGBDFrame::Ptr currentFrame = nullptr;
    deque<RGBDFrame::Ptr> refFrames;
    int refFramesSize = 5;
    Eigen::Isometry3d lastPose = Eigen::Isometry3d::Identity(), speed = Eigen::Isometry3d::Identity();
    trackerState state = NOT_READY;
    int cntLost = 0, max_lost_frame = 5;
    shared_ptr<OrbFeature> orb; shared_ptr<PnPSolver> pnp; shared_ptr<PoseGraph> poseGraph;
    mutex adjustMutex;
};
}  // namespace rgbd_tutor
