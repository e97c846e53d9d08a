// ssm/batch_tracker.h -- rgbd_tutor::BatchTracker: Tracker::updateFrame (RGB-D mode, reference src/track.cpp:8-36,140-212) for a CHUNK of frames at a time.
// The reference tracks one frame per call; features and match tables of a frame do not depend on the pose chain, so this driver uploads a chunk of frames,
// runs ssm_seq_process (ORB for every frame, match against the tracker_ref_frames preceding frames, all in batched launches), then lets ssm_tracker_run
// walk the pose chain over the chunk (same state machine, same PnP bits as the per-frame Tracker: include/ssm/pnp_core.h) and writes every frame's T_f_w.
// exp_mapping --batched uses it in place of `tracker->updateFrame(frame)`; the frames then go through PoseGraph / Mapper like in the reference's loop.
#pragma once
#include "common_headers.h"
#include "device.h"
#include "rgbdframe.h"
namespace rgbd_tutor {
class BatchTracker {
public:
    BatchTracker(const ParameterReader& para, int width, int height, const Eigen::Isometry3d& first_pose = Eigen::Isometry3d::Identity(), int chunk = 0)
        : W(width), H(height) {
        ssm_config cfg = para.deviceConfig(width, height);
        N = chunk > 0 ? chunk : para.getData<int>("tracker_chunk", 64);
        cfg.max_batch = para.getData<int>("ssm_max_batch", 16); if (cfg.max_batch > N) cfg.max_batch = N;
        dev.reset(new ssm::Device(cfg));
        ssm_tracker_params p; ssm_tracker_params_default(&p);
        p.max_lost_frame = para.getData<int>("tracker_max_lost_frame", 10); p.ref_frames = cfg.tracker_ref_frames; p.pnp_min_inliers = para.getData<int>("pnp_min_inliers", 10);
        run_chain = para.getData<int>("tracker_batched_chain", 1) != 0;     // 0: the stream supplies the poses (use_stream_pose = 1) -- features + match tables only, T_f_w stays as the frame arrived
        p.use_device = para.getData<int>("tracker_pnp_on_device", 1);      // the chain of regular frames on the GPU (0.5 ms per frame; 0: one host core, 1.6 ms; same bits)
        for (int k = 0; k < 16; k++) p.first_pose[k] = first_pose.data()[k];
        dev->check(ssm_tracker_create(dev->ctx(), &p, &trk), "ssm_tracker_create");
        const size_t np = (size_t)W * H;
        dev->check(ssm_dev_alloc(dev->ctx(), (size_t)N * np * 3, &d_bgr), "ssm_dev_alloc"); dev->check(ssm_dev_alloc(dev->ctx(), (size_t)N * np * 2, &d_depth), "ssm_dev_alloc");
    }
    ~BatchTracker() { if (trk) ssm_tracker_destroy(trk); if (dev) { if (d_bgr) ssm_dev_free(dev->ctx(), d_bgr); if (d_depth) ssm_dev_free(dev->ctx(), d_depth); } }
    BatchTracker(const BatchTracker&) = delete; BatchTracker& operator=(const BatchTracker&) = delete;
    int chunk() const { return N; }
    // the next frame pushed starts a new sequence (what is queued must have been flushed): Tracker::reset() for the bulk tracker
    void reset() { if (!pending.empty()) throw logic_error("BatchTracker::reset with frames queued: flush() first"); dev->check(ssm_tracker_reset(trk), "ssm_tracker_reset"); fed = 0; }
    // queue a frame; when the chunk is full it is processed.  Returns the frames whose poses are now known (possibly none).
    // The frame's images go up when it is queued, without waiting (ssm_memcpy_h2d_async: the frame is kept in `pending`, so its buffers stay valid): the copy runs
    // while the caller reads the next frame.  Page-locked frame buffers (FrameReader's reader_pinned) make it a plain DMA.
    vector<RGBDFrame::Ptr> push(const RGBDFrame::Ptr& f) {
        const size_t np = (size_t)W * H; const size_t i = pending.size();
        if (f->rgb.cols != W || f->rgb.rows != H || f->rgb.channels() != 3 || !f->rgb.isContinuous()) throw invalid_argument("BatchTracker: frame geometry differs from the tracker's");
        if (f->depth.empty() || !f->depth.isContinuous()) throw invalid_argument("BatchTracker: RGB-D frames need a depth image");
        pending.push_back(f);
        dev->check(ssm_memcpy_h2d_async(dev->ctx(), (uint8_t*)d_bgr + i * np * 3, f->rgb.data, np * 3), "ssm_memcpy_h2d_async");
        dev->check(ssm_memcpy_h2d_async(dev->ctx(), (uint8_t*)d_depth + i * np * 2, f->depth.data, np * 2), "ssm_memcpy_h2d_async");
        return (int)pending.size() >= N ? flush() : vector<RGBDFrame::Ptr>();
    }
    // process whatever is queued: sets T_f_w of every queued frame, fills infos (one entry per frame, in order) and returns the frames
    vector<RGBDFrame::Ptr> flush() {
        const int n = (int)pending.size();
        if (n == 0) return vector<RGBDFrame::Ptr>();
        ssm_frames_dev in;                                 // (the images were uploaded by push(), on the context's stream: the launches below run behind the copies) memset(&in, 0, sizeof(in));
        in.bgr = (const uint8_t*)d_bgr; in.depth = (const uint16_t*)d_depth; in.n = n; in.continue_sequence = fed > 0 ? 1 : 0; in.stages = SSM_STAGE_ORB | SSM_STAGE_MATCH;
        ssm_seq_out_dev out;
        dev->check(ssm_seq_process(dev->ctx(), &in, &out), "ssm_seq_process");
        vector<double> poses((size_t)n * 16); infos.assign(n, ssm_track_info());
        if (!run_chain) {                                  // poses are the stream's: the chunk's features and match tables are on the device (ssm_seq_out_dev), nothing to solve
            dev->check(ssm_sync(dev->ctx()), "ssm_sync");
            for (ssm_track_info& i : infos) { i.state = 1; i.tracked = 1; i.n_matches = -1; i.n_inliers = 0; }
            last_out = out; fed += n;
            vector<RGBDFrame::Ptr> done0; done0.swap(pending);
            return done0;
        }
        const int rc = ssm_tracker_run(trk, &out, n, poses.data(), infos.data());
        if (rc != SSM_OK) throw ssm::DeviceError(rc, string("ssm_tracker_run: ") + ssm_tracker_last_error(trk));
        for (int i = 0; i < n; i++) {
            Eigen::Isometry3d T; for (int k = 0; k < 16; k++) T.matrix().data()[k] = poses[(size_t)i * 16 + k];
            pending[i]->setTransform(T);
        }
        last_out = out; fed += n;
        vector<RGBDFrame::Ptr> done; done.swap(pending);
        return done;
    }
    vector<ssm_track_info> infos;                 // of the most recent flush
    ssm_seq_out_dev last_out;                     // device tables of the most recent flush (features, descriptors, 3-D positions, match tables of its frames)
    ssm::Device& device() { return *dev; }
private:
    int W, H, N = 64; long fed = 0; bool run_chain = true;
    unique_ptr<ssm::Device> dev; ssm_tracker* trk = nullptr; void *d_bgr = nullptr, *d_depth = nullptr;
    vector<RGBDFrame::Ptr> pending;
};
}  // namespace rgbd_tutor
