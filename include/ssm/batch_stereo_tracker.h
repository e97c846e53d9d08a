// ssm/batch_stereo_tracker.h -- the stereo tracker in bulk: what Tracker::updateFrame does per frame with tracker_mode = stereo (reference src/track.cpp:8-28,
// 30-36, 38-138, 202-212: QuadFeatureMatch on (current, previous) x (left, right), VisualOdometryStereo::Process, pose = pose * inv(motion), cntLost / LOST /
// lostRecover) and what FrameReader::next() does for a KITTI frame (SGBM depth of the current pair, src/rgbdframe.cpp:64-116), for a whole CHUNK of frames per
// launch: ssm_stereo_seq_process (include/ssm_hip.h) runs the quad matcher, SGBM + the depth conversion and the RANSAC ego-motion of every frame of the chunk in
// batched kernels; this class keeps what is host state in the reference too -- the tracker's state machine and the rand() stream of the VO object -- and hands
// every frame its pose and its depth image.  Poses, depth images and states are those of the per-frame classes, bit for bit (host/test_host.cpp).
//
// The rand() stream: a frame whose quad matcher finds >= 6 matches draws 3 x ransac_iters numbers (src/vo.cpp:74-93), others none.  The bulk call is handed the
// raw draws and reports how many it used.  One case breaks the bulk call's assumption that every frame with a previous frame runs the VO: a frame that arrives
// while the tracker is LOST goes through lostRecover and draws nothing -- from that frame to the end of the chunk the VO is redone per frame on the chunk's
// quad matches (VisualOdometryStereo::ProcessMatches), with the stream where the per-frame walk would have it.
// Not done here: estimateVO's orb->detectFeatures(currentFrame) (track.cpp:42) -- the ORB features of a stereo frame feed only the pose graph's loop closure
// (out of scope); a caller that wants them runs OrbFeature::detectFeatures on the frames this class returns (their depth is set).
#pragma once
#include "common_headers.h"
#include "device.h"
#include "rgbdframe.h"
#include "track.h"
#include "vo_stereo.hpp"
namespace rgbd_tutor {
class BatchStereoTracker {
public:
    struct Info { int state = Tracker::NOT_READY; bool tracked = false; int n_matches = -1, n_inliers = 0; };
    BatchStereoTracker(const ParameterReader& para, VisualOdometryStereo::parameters vp, int width, int height, int chunk = 0)
        : W(width), H(height), viso(vp), voparam(vp) {
        ssm_config cfg = para.deviceConfig(width, height);
        N = chunk > 0 ? chunk : para.getData<int>("tracker_chunk", 64);
        cfg.max_batch = para.getData<int>("ssm_max_batch", 32); if (cfg.max_batch > N) cfg.max_batch = N;
        dev.reset(new ssm::Device(cfg));
        max_lost_frame = para.getData<int>("tracker_max_lost_frame", 10);
        baseline = para.getData<double>("camera.baseline"); roix = para.getData<double>("camera.roix", 20.0); roiy = para.getData<double>("camera.roiy", 5.0);
        roiz = para.getData<double>("camera.roiz", 40.0);
        camera = para.getCamera();
        const size_t np = (size_t)W * H;
        dev->check(ssm_dev_alloc(dev->ctx(), (size_t)N * np, &d_left), "ssm_dev_alloc"); dev->check(ssm_dev_alloc(dev->ctx(), (size_t)N * np, &d_right), "ssm_dev_alloc");
        dev->check(ssm_dev_alloc(dev->ctx(), (size_t)N * vp.ransac_iters * 3 * 4 + 16, &d_rand), "ssm_dev_alloc");
    }
    ~BatchStereoTracker() { if (dev) for (void* p : {d_left, d_right, d_rand}) if (p) ssm_dev_free(dev->ctx(), p); }
    BatchStereoTracker(const BatchStereoTracker&) = delete; BatchStereoTracker& operator=(const BatchStereoTracker&) = delete;
    int chunk() const { return N; }
    Tracker::trackerState getState() const { return (Tracker::trackerState)state; }
    // queue a frame (img_lc / img_rc = its rectified gray pair); when the chunk is full it is processed.  Returns the frames whose poses are now known.
    // (the pair goes up when it is queued, without waiting: the frame is kept in `pending`, so its images stay valid)
    vector<RGBDFrame::Ptr> push(const RGBDFrame::Ptr& f) {
        const size_t np = (size_t)W * H; const size_t i = pending.size();
        if (f->img_lc.cols != W || f->img_lc.rows != H || f->img_lc.type() != CV_8UC1 || !f->img_lc.isContinuous() ||
            f->img_rc.cols != W || f->img_rc.rows != H || f->img_rc.type() != CV_8UC1 || !f->img_rc.isContinuous()) throw invalid_argument("BatchStereoTracker: frame geometry differs from the tracker's");
        pending.push_back(f);
        dev->check(ssm_memcpy_h2d_async(dev->ctx(), (uint8_t*)d_left + i * np, f->img_lc.data, np), "ssm_memcpy_h2d_async");
        dev->check(ssm_memcpy_h2d_async(dev->ctx(), (uint8_t*)d_right + i * np, f->img_rc.data, np), "ssm_memcpy_h2d_async");
        return (int)pending.size() >= N ? flush() : vector<RGBDFrame::Ptr>();
    }
    // process whatever is queued: sets T_f_w and the depth image of every queued frame, fills infos (one entry per frame, in order)
    vector<RGBDFrame::Ptr> flush() {
        const int n = (int)pending.size();
        if (n == 0) return vector<RGBDFrame::Ptr>();
        const size_t np = (size_t)W * H; const int iters = voparam.ransac_iters;
        // (the images were uploaded by push())
        // the raw draws the VO object would make if every frame of the chunk ran; the object is put where the stream really went afterwards
        const VisualOdometry::RandState rs0 = viso.saveRand();
        vector<uint32_t> draws((size_t)n * iters * 3);
        for (uint32_t& d : draws) d = viso.rawRand();
        if (!draws.empty()) dev->check(ssm_memcpy_h2d(dev->ctx(), d_rand, draws.data(), draws.size() * 4), "ssm_memcpy_h2d");
        ssm_stereo_frames_dev in; memset(&in, 0, sizeof(in));
        in.left = (const uint8_t*)d_left; in.right = (const uint8_t*)d_right; in.n = n; in.w = W; in.h = H; in.continue_sequence = fed > 0 ? 1 : 0; in.stages = 0;
        in.max_corners = 1000; ssm_sgbm_params_default(&in.sgbm);
        in.baseline = baseline; in.cu = camera.cx; in.cv = camera.cy; in.f = camera.fx; in.roix = roix; in.roiy = roiy; in.roiz = roiz; in.scale = camera.scale;
        in.vo.f = voparam.calib.f; in.vo.cu = voparam.calib.cu; in.vo.cv = voparam.calib.cv; in.vo.base = voparam.base; in.vo.inlier_threshold = voparam.inlier_threshold;
        in.vo.reweighting = voparam.reweighting ? 1 : 0; in.vo.pad = 0; in.ransac_iters = iters; in.rand_stream = (const uint32_t*)d_rand;
        ssm_stereo_out_dev out;
        dev->check(ssm_stereo_seq_process(dev->ctx(), &in, &out), "ssm_stereo_seq_process");
        // FrameReader's depth image of every frame (src/rgbdframe.cpp:81-116): into page-locked buffers, all downloads enqueued behind the kernels, one wait
        for (int i = 0; i < n; i++) {
            const RGBDFrame::Ptr& f = pending[i];
            std::shared_ptr<void> blk = ssm::PinnedPool::instance().take(np * 2);
            f->depth = cv::Mat(H, W, CV_16UC1, blk.get()); f->depth.hold(blk);
            dev->check(ssm_memcpy_d2h_async(dev->ctx(), f->depth.data, out.depth + (size_t)i * np, np * 2), "ssm_memcpy_d2h_async");
        }
        dev->check(ssm_sync(dev->ctx()), "ssm_sync");
        vector<int32_t> nquad(n), vres((size_t)n * 2); vector<double> tr((size_t)n * 6);
        dev->check(ssm_memcpy_d2h(dev->ctx(), nquad.data(), out.nquad, (size_t)n * 4), "ssm_memcpy_d2h");
        dev->check(ssm_memcpy_d2h(dev->ctx(), vres.data(), out.vo_result, (size_t)n * 8), "ssm_memcpy_d2h");
        dev->check(ssm_memcpy_d2h(dev->ctx(), tr.data(), out.tr, (size_t)n * 48), "ssm_memcpy_d2h");
        infos.assign(n, Info());
        bool redo = false;                                  // from here on the VO is redone per frame (a frame went through lostRecover)
        long used = 0;                                      // draws of the frames walked so far, while the bulk results are in use
        for (int i = 0; i < n; i++) {
            const RGBDFrame::Ptr& f = pending[i];
            Info& info = infos[i];
            if (state == Tracker::NOT_READY) {              // initFirstFrame (track.cpp:30-36): the frame keeps the transform it arrived with
                refBackT = f->getTransform(); speed = Eigen::Isometry3d::Identity(); state = Tracker::OK;
                info.state = state; info.tracked = true;
                continue;
            }
            if (state == Tracker::LOST) {                   // lostRecover (track.cpp:202-212): no quad matcher, no VO, no draws
                if (!redo && nquad[i] >= 6) {               // the bulk call drew for this frame: the stream goes back to where it was before it
                    redo = true;
                    viso.restoreRand(rs0); for (long k = 0; k < used; k++) (void)viso.rawRand();
                }
                f->setTransform(refBackT); refBackT = f->getTransform(); state = Tracker::OK; cntLost = 0;
                info.state = state;
                continue;
            }
            // estimateVO (track.cpp:38-138)
            f->setTransform(speed * refBackT);
            bool success = false; Eigen::Isometry3d M = Eigen::Isometry3d::Identity();
            info.n_matches = nquad[i] < 0 ? 0 : nquad[i];
            if (nquad[i] >= 0) {                            // (a frame without a previous pair has no quad matcher: img_lp / img_rp empty in the per-frame class)
                if (!redo) {
                    if (nquad[i] >= 6) used += (long)iters * 3;
                    success = vres[(size_t)i * 2 + 1] != 0; info.n_inliers = vres[(size_t)i * 2];
                    if (success) { cv::Mat mm = VisualOdometryStereo::motionMatrix(&tr[(size_t)i * 6]); for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) M(r, c) = mm.at<double>(r, c); }
                } else {
                    vector<pmatch> qm((size_t)nquad[i]);
                    static_assert(sizeof(pmatch) == sizeof(ssm_pmatch), "pmatch layout");
                    if (nquad[i] > 0) dev->check(ssm_memcpy_d2h(dev->ctx(), qm.data(), out.quad + (size_t)i * out.max_corners, (size_t)nquad[i] * sizeof(ssm_pmatch)), "ssm_memcpy_d2h");
                    success = viso.ProcessMatches(qm, dev->ctx()); info.n_inliers = viso.getNumberOfInliers();
                    if (success) { cv::Mat mm = viso.getMotion(); for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) M(r, c) = mm.at<double>(r, c); }
                }
            }
            if (!success) { cntLost++; if (cntLost > max_lost_frame) state = Tracker::LOST; info.state = state; continue; }
            pose = pose * M.inverse();
            f->setTransform(pose);
            cntLost = 0;
            speed = pose * lastPose.inverse();
            lastPose = f->getTransform();
            refBackT = f->getTransform();
            info.state = state; info.tracked = true;
        }
        if (!redo) { viso.restoreRand(rs0); for (long k = 0; k < used; k++) (void)viso.rawRand(); }
        fed += n;
        vector<RGBDFrame::Ptr> done; done.swap(pending);
        return done;
    }
    vector<Info> infos;                           // of the most recent flush
    ssm::Device& device() { return *dev; }
private:
    int W, H, N = 64; long fed = 0;
    unique_ptr<ssm::Device> dev; void *d_left = nullptr, *d_right = nullptr, *d_rand = nullptr;
    VisualOdometryStereo viso; VisualOdometryStereo::parameters voparam;
    CAMERA_INTRINSIC_PARAMETERS camera; double baseline = 0, roix = 20, roiy = 5, roiz = 40;
    int state = Tracker::NOT_READY, cntLost = 0, max_lost_frame = 10;
    Eigen::Isometry3d pose = Eigen::Isometry3d::Identity(), lastPose = Eigen::Isometry3d::Identity(), speed = Eigen::Isometry3d::Identity(), refBackT = Eigen::Isometry3d::Identity();
    vector<RGBDFrame::Ptr> pending;
};
}  // namespace rgbd_tutor
