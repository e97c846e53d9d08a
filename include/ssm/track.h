// ssm/track.h -- rgbd_tutor::Tracker (reference include/track.h:50-191, src/track.cpp): the per-frame state machine.
// RGB-D mode = Tracker::trackRefFrame (src/track.cpp:140-200), restated line for line on the device-backed
// OrbFeature (tracker_mode = rgbd, the default here).  The checked-in reference instead calls the stereo estimateVO()
// (track.cpp:19, 38-138): tracker_mode = stereo runs that branch -- ORB features for the back end, QuadFeatureMatch in
// tracking mode on the frame's four stereo images, VisualOdometryStereo::Process, pose chained as pose * inv(motion).  The
// triangulate10D / UVDisparity block between them (track.cpp:68-80) only fills moving_mask, which Mapper overwrites
// (SURVEY.md s.2), and is not rebuilt; inv() of the rigid motion is the closed form [R^T | -R^T t] instead of Matrix_'s LU
// inverse.  PoseGraph is out of scope: setPoseGraph is kept as a no-op hook.
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
        const string mode = para.getData<string>("tracker_mode", string("rgbd"));
        if (mode != "rgbd" && mode != "stereo") throw invalid_argument("tracker_mode: 'rgbd' (Tracker::trackRefFrame) or 'stereo' (Tracker::estimateVO)");
        stereo = mode == "stereo";
        if (stereo) viso.reset(new VisualOdometryStereo(param));
    }
    void setPoseGraph(shared_ptr<PoseGraph> pg) { poseGraph = pg; }
    // put in a new frame, returns its pose (src/track.cpp:8-28)
    Eigen::Isometry3d updateFrame(RGBDFrame::Ptr& newFrame) {
        unique_lock<mutex> lck(adjustMutex);
        currentFrame = newFrame;
        if (state == NOT_READY) { initFirstFrame(); return Eigen::Isometry3d::Identity(); }
        if (state == OK) { if (stereo) estimateVO(); else trackRefFrame(); return currentFrame->getTransform(); }
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
    // a new sequence starts with the next frame: the state a freshly constructed Tracker has (not in the reference, which tracks one sequence per process;
    // exp_mapping's `sequence_length` uses it for streams that are a concatenation of independent sequences)
    void reset() {
        unique_lock<mutex> lck(adjustMutex);
        state = NOT_READY; refFrames.clear(); currentFrame = nullptr; cntLost = 0;
        lastPose = speed = pose = Eigen::Isometry3d::Identity();
    }
    // wall time spent inside updateFrame so far, split the way the reference's own drivers print it (experiment/match_orbfeature_tum.cpp:22-27: detect + match;
    // experiment/run_tracker.cpp:35-48: the whole updateFrame): milliseconds, summed over the calls
    struct Timing { double detect_ms = 0, match_ms = 0, pnp_ms = 0; long frames = 0; } timing;
    shared_ptr<OrbFeature> orbFeature() const { return orb; }
protected:
    void initFirstFrame() {                                             // track.cpp:30-36
        orb->detectFeatures(currentFrame);
        refFrames.push_back(currentFrame);
        speed = Eigen::Isometry3d::Identity();
        state = OK;
    }
    void estimateVO() {                                                 // track.cpp:38-138
        currentFrame->setTransform(speed * refFrames.back()->getTransform());
        orb->detectFeatures(currentFrame);
        bool success = false;
        if (!currentFrame->img_lc.empty() && !currentFrame->img_rc.empty() && !currentFrame->img_lp.empty() && !currentFrame->img_rp.empty()) {
            QuadFeatureMatch quadmatcher(currentFrame->img_lc, currentFrame->img_rc, currentFrame->img_lp, currentFrame->img_rp,
                                         currentFrame->semantic_cur_r, currentFrame->semantic_pre_r, true);
            quadmatcher.init(DET_GFTT, DES_SIFT);
            quadmatcher.detectFeature();
            quadmatcher.circularMatching();
            lastMatches = (int)quadmatcher.quadmatches.size();
            if (viso->Process(quadmatcher)) {
                cv::Mat motion = viso->getMotion();
                Eigen::Isometry3d M;
                for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) M(i, j) = motion.at<double>(i, j);
                pose = pose * M.inverse();
                lastInliers = viso->getNumberOfInliers();
                success = true;
            }
        }
        if (!success) { cntLost++; if (cntLost > max_lost_frame) state = LOST; return; }
        currentFrame->setTransform(pose);
        cntLost = 0;
        speed = pose * lastPose.inverse();
        lastPose = currentFrame->getTransform();
        refFrames.push_back(currentFrame);
        while ((int)refFrames.size() > refFramesSize) refFrames.pop_front();
    }
    void trackRefFrame() {                                              // track.cpp:140-200
        currentFrame->setTransform(speed * refFrames.back()->getTransform());
        const auto tm0 = chrono::steady_clock::now();
        orb->detectFeatures(currentFrame);
        const auto tm1 = chrono::steady_clock::now();
        vector<cv::Point3f> obj; vector<cv::Point2f> img;
        const vector<vector<cv::DMatch>> allMatches = orb->matchMany(refFrames, currentFrame);      // orb->match(pFrame, currentFrame) for every pFrame, one wait
        const auto tm2 = chrono::steady_clock::now();
        timing.detect_ms += chrono::duration<double, milli>(tm1 - tm0).count(); timing.match_ms += chrono::duration<double, milli>(tm2 - tm1).count(); timing.frames++;
        size_t ri = 0;
        for (auto pFrame : refFrames) {
            const vector<cv::DMatch>& matches = allMatches[ri++];
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
        const auto tm3 = chrono::steady_clock::now();
        pnp->solvePnP(img, obj, currentFrame->camera, inlierIndex, T);
        timing.pnp_ms += chrono::duration<double, milli>(chrono::steady_clock::now() - tm3).count();
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
    bool stereo = false; unique_ptr<VisualOdometryStereo> viso; Eigen::Isometry3d pose = Eigen::Isometry3d::Identity();      // estimateVO state (track.h:181)
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
