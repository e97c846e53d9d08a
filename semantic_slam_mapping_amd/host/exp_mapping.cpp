// exp_mapping -- the reference's product driver (experiment/exp_mapping.cpp:18-59) on the MI355X front end: build the
// ParameterReader, Tracker, FrameReader, PoseGraph and Mapper, then loop next() -> updateFrame() -> tryInsertKeyFrame(),
// finally shut the graph and the mapper down.  Differences: no cv::imshow; the frame source is selected by `dataset`
// (synthetic | raw | tum | kitti; the reference hard-codes its FrameReader type); with tum / kitti the poses come from the
// tracker (use_stream_pose defaults to 0 there); prints frames/s at the end.
//
// `exp_mapping <parameters> --batched` (or tracker_batched=1): the same loop with the pose chain in bulk -- frames are queued in chunks of tracker_chunk,
// BatchTracker runs ORB + the match tables for a whole chunk in batched launches and then the Tracker state machine + PnP over it (ssm_tracker_run:
// the poses of the per-frame Tracker, bit for bit), after which the chunk's frames go through tryInsertKeyFrame like in the per-frame loop.
//
// `exp_mapping <parameters> --ranks N`: the multi-GPU form (BASELINE.json configs[4]; the reference is one process).  The parent starts
// N FRESH processes of itself (`--rank r`, fork + exec of /proc/self/exe) before anything touches HIP -- a forked copy of a process whose HIP / RCCL
// static constructors have already run is not a state either library is tested in; rank r drives GPU r, owns the contiguous frame block [lo, hi) of
// [start_index, end_index), feeds the tracker_ref_frames frames in front of its block to the tracker only (the matcher halo: Tracker::trackRefFrame
// matches against the refFrames deque, src/track.cpp:150-152), maps its key-frames as usual, then inserts their clouds into one context map and calls
// ssm_voxel_allgather (ONE RCCL all-gather behind the C ABI).  Every rank ends with the map of the whole sequence; rank 0 prints it.  The ncclUniqueId
// travels through a file in a private temporary directory (written under another name, then renamed).  The parent reaps with waitpid(-1): the first
// rank that fails (or `rank_timeout_s`, default 900) gets the others killed, so a rank blocked in ncclCommInitRank never outlives its failed peer.
#include "ssm/rgbdframe.h"
#include "ssm/track.h"
#include "ssm/pose_graph.h"
#include "ssm/common_headers.h"
#include "ssm/mapper.h"
#include "ssm/vo_stereo.hpp"
#include "ssm/batch_stereo_tracker.h"
#include "ssm/batch_tracker.h"
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
using namespace std;
using namespace rgbd_tutor;

static int run_rank(ParameterReader& parameterReader, int rank, int nranks, const string& id_dir);
static FrameReader::DATASET dataset_type(const ParameterReader& pr)
{
    const string ds = pr.getData<string>("dataset", string("synthetic"));
    return ds == "raw" ? FrameReader::RAW : ds == "tum" ? FrameReader::TUM : ds == "kitti" ? FrameReader::KITTI : FrameReader::SYNTHETIC;
}

// every frame's final T_f_w, in frame order: FNV-1a over the 128 bytes (printed as pose_fnv) and, with trajectory_output=<file>, one text line per frame
// (id, then the 16 column-major doubles as C99 hex floats: exact)
struct Trajectory {
    uint64_t h = 0xCBF29CE484222325ull; ofstream out;
    explicit Trajectory(const string& path) { if (!path.empty()) out.open(path); }
    void add(const RGBDFrame::Ptr& f) {
        const Eigen::Isometry3d T = f->getTransform();
        const unsigned char* b = (const unsigned char*)T.data();
        for (int k = 0; k < 128; k++) { h ^= b[k]; h *= 0x100000001B3ull; }
        if (out.is_open()) { char buf[64]; out << f->id; for (int k = 0; k < 16; k++) { snprintf(buf, sizeof(buf), " %a", T.data()[k]); out << buf; } out << "\n"; }
    }
};

int main(int argc, char** argv)
{
    ParameterReader parameterReader(argc > 1 ? argv[1] : "./parameters.txt");
    int nranks = 1, my_rank = -1; string id_dir;
    bool batched = parameterReader.getData<int>("tracker_batched", 0) != 0;
    for (int i = 2; i < argc; i++) if (string(argv[i]) == "--batched") batched = true;
    for (int i = 2; i + 1 < argc; i++) {
        if (string(argv[i]) == "--ranks") nranks = atoi(argv[i + 1]);
        if (string(argv[i]) == "--rank") my_rank = atoi(argv[i + 1]);
        if (string(argv[i]) == "--id-dir") id_dir = argv[i + 1];
    }
    if (my_rank >= 0) {                                   // a rank process started by the parent below
        if (nranks < 1 || my_rank >= nranks || id_dir.empty()) { cerr << "exp_mapping: --rank needs --ranks N and --id-dir" << endl; return 2; }
        try { return run_rank(parameterReader, my_rank, nranks, id_dir); }
        catch (const exception& e) { cerr << RED << "rank " << my_rank << ": " << e.what() << RESET << endl; return 2; }
    }
    if (nranks > 1 || parameterReader.getData<int>("force_rank_path", 0)) {
        char tmpl[] = "/tmp/ssm_ranks_XXXXXX";
        if (!mkdtemp(tmpl)) { perror("mkdtemp"); return 2; }
        const string dir = tmpl, ranks_s = to_string(nranks);
        vector<pid_t> kids;
        // INVARIANT: nothing above this point creates an ssm::Device (ParameterReader only parses the file): the parent forks and the children exec BEFORE any HIP
        // call.  Never move device work in front of this loop; the counter makes a violation fail here instead of on the node.
        if (ssm::devices_created().load() != 0) { cerr << RED << "exp_mapping: internal error: a device context exists before the ranks are started" << RESET << endl; return 2; }
        for (int r = 0; r < nranks; r++) {
            pid_t p = fork();                                 // no HIP call has happened in this process; the child execs at once
            if (p < 0) { perror("fork"); for (pid_t k : kids) kill(k, SIGKILL); return 2; }
            if (p == 0) {
                const string rs = to_string(r);
                const char* av[] = { argv[0], argv[1], "--ranks", ranks_s.c_str(), "--rank", rs.c_str(), "--id-dir", dir.c_str(), nullptr };
                execv("/proc/self/exe", const_cast<char* const*>(av));
                perror("execv"); _exit(127);
            }
            kids.push_back(p);
        }
        const int timeout_s = parameterReader.getData<int>("rank_timeout_s", 900);
        const auto t0 = chrono::steady_clock::now();
        int rc = 0; size_t alive = kids.size();
        while (alive) {
            int st = 0; const pid_t p = waitpid(-1, &st, WNOHANG);
            if (p > 0) {
                alive--;
                for (pid_t& k : kids) if (k == p) k = -1;
                if (!WIFEXITED(st) || WEXITSTATUS(st)) {
                    if (!rc) { cerr << RED << "exp_mapping: a rank failed (status " << st << "); stopping the others" << RESET << endl; for (pid_t k : kids) if (k > 0) kill(k, SIGKILL); }
                    rc = 1;
                }
                continue;
            }
            if (p < 0 && errno != EINTR) break;
            if (chrono::duration<double>(chrono::steady_clock::now() - t0).count() > timeout_s) {
                if (!rc) { cerr << RED << "exp_mapping: ranks still running after " << timeout_s << " s; killing them" << RESET << endl; for (pid_t k : kids) if (k > 0) kill(k, SIGKILL); }
                rc = 1;
            }
            this_thread::sleep_for(chrono::milliseconds(5));
        }
        unlink((dir + "/nccl_id").c_str()); rmdir(dir.c_str());
        return rc;
    }
    VisualOdometryStereo::parameters voparam;
    double f = parameterReader.getData<double>("camera.fx");
    double c_u = parameterReader.getData<double>("camera.cx");
    double c_v = parameterReader.getData<double>("camera.cy");
    double base = parameterReader.getData<double>("camera.baseline", 0.0);
    double inlier_threshold = parameterReader.getData<double>("inlier_threshold", 6.0);
    voparam.calib.f = f; voparam.calib.cu = c_u;
    voparam.calib.cv = c_v; voparam.base = base;
    voparam.inlier_threshold = inlier_threshold;
    try {
        Tracker::Ptr tracker(new Tracker(parameterReader, voparam));
        const FrameReader::DATASET type = dataset_type(parameterReader);
        const bool batched_stereo = batched && parameterReader.getData<string>("tracker_mode", string("rgbd")) == "stereo";
        if (batched_stereo) parameterReader.set("kitti_reader_depth", "0");     // the bulk stereo tracker computes the depth images, a chunk of frames per launch
        FrameReader frameReader(parameterReader, type);
        PoseGraph poseGraph(parameterReader, tracker);
        Mapper mapper(parameterReader, poseGraph);
        const bool use_gt_pose = parameterReader.getData<int>("use_stream_pose", (type == FrameReader::TUM || type == FrameReader::KITTI) ? 0 : 1) != 0;
        int nframes = 0;
        const int frame_period_ms = parameterReader.getData<int>("frame_period_ms", 0);
        Trajectory traj(parameterReader.getData<string>("trajectory_output", string("")));
        // `sequence_length` = L > 0: the stream is a concatenation of independent sequences of L frames (synthetic_rigid streams): the tracker starts over at every
        // multiple of L.  `timing_skip_frames`: the rates printed at the end leave out the first frames (context creation, code-object load, first-use allocations).
        const int seq_len = parameterReader.getData<int>("sequence_length", 0), skip = parameterReader.getData<int>("timing_skip_frames", 0);
        // where the loop's wall time goes, the way the reference's drivers time it: FrameReader::next (experiment/exp_mapping.cpp:36), Tracker::updateFrame
        // (experiment/run_tracker.cpp:35-48) / BatchTracker::push, PoseGraph::tryInsertKeyFrame (:47)
        double reader_s = 0, track_s = 0, kf_s = 0; int timed = 0;
        typedef chrono::steady_clock::time_point tp;
        auto now = [] { return chrono::steady_clock::now(); };
        auto sec = [](tp a, tp b) { return chrono::duration<double>(b - a).count(); };
        // reader_preload = 1: every frame is read before the loop starts (frames resident in host memory, like the inputs of bench.py's timed region are resident in HBM):
        // the loop then measures the tracker / mapper side alone.  preload_s is printed.
        vector<RGBDFrame::Ptr> preloaded; size_t pre_i = 0; double preload_s = 0;
        const bool preload = parameterReader.getData<int>("reader_preload", 0) != 0;
        if (preload) { const tp a = now(); while (RGBDFrame::Ptr f = frameReader.next()) preloaded.push_back(f); preload_s = sec(a, now()); }
        auto t0 = now(); tp t_timed0 = t0, t1 = t0;          // t1: the end of the loop (taken before a bulk tracker and its device context are torn down)
        auto read_next = [&]() {
            const tp a = now();
            RGBDFrame::Ptr f;
            if (preload) { if (pre_i < preloaded.size()) { f = preloaded[pre_i]; preloaded[pre_i++] = nullptr; } } else f = frameReader.next();
            if (nframes >= skip && f) reader_s += sec(a, now());
            return f; };
        if (batched && parameterReader.getData<string>("tracker_mode", string("rgbd")) == "rgbd") {
            unique_ptr<BatchTracker> bt; map<int, Eigen::Isometry3d> gt; int lost = 0; int pushed = 0;
            const bool chain = parameterReader.getData<int>("tracker_batched_chain", 1) != 0;
            if (!chain && !use_gt_pose) throw invalid_argument("tracker_batched_chain=0 needs use_stream_pose=1: without the chain nothing computes the poses");
            // (rates: a chunk counts as a whole -- the frames of a flush that started at or after timing_skip_frames, against the time of that flush)
            auto handle = [&](const vector<RGBDFrame::Ptr>& done, bool counted) {
                for (size_t i = 0; i < done.size(); i++) {
                    const RGBDFrame::Ptr& f = done[i];
                    if (use_gt_pose) f->setTransform(gt[f->id]);
                    gt.erase(f->id);
                    traj.add(f);
                    poseGraph.tryInsertKeyFrame(const_cast<RGBDFrame::Ptr&>(f));
                    if (bt->infos[i].state == Tracker::LOST) { cout << "tracker is lost" << endl; lost++; }
                    if (counted) timed++;
                    nframes++;
                }
            };
            while (RGBDFrame::Ptr frame = read_next()) {
                if (!bt) bt.reset(new BatchTracker(parameterReader, frame->rgb.cols, frame->rgb.rows, frame->T_f_w));
                const bool count = nframes >= skip; const tp a = now();
                if (count && timed == 0 && track_s == 0) t_timed0 = a;
                if (seq_len > 0 && chain && pushed > 0 && pushed % seq_len == 0) { handle(bt->flush(), count); bt->reset(); }     // (without the chain a sequence boundary changes nothing: features and tables only)
                gt[frame->id] = frame->T_f_w;
                pushed++;
                handle(bt->push(frame), count);
                if (count) track_s += sec(a, now());
            }
            const tp a = now(); const bool count = nframes >= skip;
            if (bt) handle(bt->flush(), count);
            if (count) track_s += sec(a, now());
            t1 = now();
            cout << "batched tracker: chunk " << (bt ? bt->chunk() : 0) << " lost " << lost << endl;
        } else if (batched_stereo) {
            // Tracker::estimateVO + FrameReader's SGBM depth in bulk (include/ssm/batch_stereo_tracker.h): quad matcher, depth and ego-motion of a chunk per launch
            unique_ptr<BatchStereoTracker> bs; int lost = 0;
            auto handle = [&](const vector<RGBDFrame::Ptr>& done, bool counted) {
                for (size_t i = 0; i < done.size(); i++) {
                    const RGBDFrame::Ptr& f = done[i];
                    traj.add(f);
                    poseGraph.tryInsertKeyFrame(const_cast<RGBDFrame::Ptr&>(f));
                    if (bs->infos[i].state == Tracker::LOST) { cout << "tracker is lost" << endl; lost++; }
                    if (counted) timed++;
                    nframes++;
                }
            };
            while (RGBDFrame::Ptr frame = read_next()) {
                if (!bs) bs.reset(new BatchStereoTracker(parameterReader, voparam, frame->img_lc.cols, frame->img_lc.rows));
                const bool count = nframes >= skip; const tp a = now();
                if (count && timed == 0 && track_s == 0) t_timed0 = a;
                handle(bs->push(frame), count);
                if (count) track_s += sec(a, now());
            }
            const tp a = now(); const bool count = nframes >= skip;
            if (bs) handle(bs->flush(), count);
            if (count) track_s += sec(a, now());
            t1 = now();
            cout << "batched stereo tracker: chunk " << (bs ? bs->chunk() : 0) << " lost " << lost << endl;
        } else {
        while (RGBDFrame::Ptr frame = read_next()) {
            if (nframes == skip) { t_timed0 = now(); tracker->timing = Tracker::Timing(); }
            if (seq_len > 0 && nframes > 0 && nframes % seq_len == 0) tracker->reset();
            Eigen::Isometry3d gt = frame->T_f_w;
            const tp a = now();
            tracker->updateFrame(frame);
            const tp b = now();
            if (use_gt_pose) frame->setTransform(gt);           // synthetic stream: poses are given, the tracker only produces features/matches
            traj.add(frame);
            poseGraph.tryInsertKeyFrame(frame);
            if (nframes >= skip) { track_s += sec(a, b); kf_s += sec(b, now()); timed++; }
            if (tracker->getState() == Tracker::LOST) cout << "tracker is lost" << endl;
            nframes++;
            if (frame_period_ms > 0) this_thread::sleep_for(chrono::milliseconds(frame_period_ms));      // a camera's frame period (measurements of the viewer thread under a paced stream)
        }
        t1 = now(); }
        const double s = sec(t0, t1), s_timed = sec(t_timed0, t1);
        mapper.SaveMap();
        poseGraph.shutdown();
        this_thread::sleep_for(chrono::milliseconds(parameterReader.getData<int>("mapper_drain_ms", 300)));
        mapper.shutdown();
        cout << "frames " << nframes << " keyframes " << poseGraph.keyframes.size() << " map_updates " << mapper.updates()
             << " map_points " << (mapper.getGlobalMap() ? mapper.getGlobalMap()->points.size() : 0) << " pose_fnv " << hex << traj.h << dec << " host_loop_fps " << nframes / s;
        // the rates of the frames after timing_skip_frames: loop_fps = the whole loop (reader included); tracker_fps = frames / time inside updateFrame (or BatchTracker::push /
        // flush) -- what experiment/run_tracker.cpp:35-48 times; the *_ms are per frame
        if (timed > 0) {
            const Tracker::Timing& tt = tracker->timing; const double fr = tt.frames > 0 ? (double)tt.frames : 1.0;
            cout << " timed_frames " << timed << " loop_fps " << timed / s_timed << " tracker_fps " << (track_s > 0 ? timed / track_s : 0.0) << " reader_ms " << reader_s * 1e3 / timed
                 << " preload_s " << preload_s << " tracker_ms " << track_s * 1e3 / timed << " keyframe_ms " << kf_s * 1e3 / timed << " detect_ms " << tt.detect_ms / fr << " match_ms " << tt.match_ms / fr << " pnp_ms " << tt.pnp_ms / fr;
        }
        // final_map_fnv = K > 0: the clouds of the first K key-frames fused into ONE context map (what `globalMap += cloud` over all key-frames + one VoxelGrid pass holds,
        // src/mapper.cpp:121-158), FNV-1a of its centroids -- unlike map_points above it does not depend on the viewer thread's update schedule, so two runs (per-frame
        // and --batched, 1 or N ranks) can be compared by it
        if (const int fm = parameterReader.getData<int>("final_map_fnv", 0); fm > 0 && !poseGraph.keyframes.empty()) {
            ssm::Device dev(parameterReader.deviceConfig(frameReader.width, frameReader.height));
            int used = 0;
            for (RGBDFrame::Ptr& kf : poseGraph.keyframes) {
                if (used++ >= fm) break;
                Mapper::PointCloud::Ptr c = mapper.generatePointCloud(kf);
                if (!c->points.empty()) dev.check(ssm_map_insert(dev.ctx(), reinterpret_cast<const ssm_point*>(c->points.data()), (int)c->points.size()), "ssm_map_insert");
            }
            int nvox = 0; dev.check(ssm_map_size(dev.ctx(), &nvox), "ssm_map_size");
            vector<ssm_point> m((size_t)max(nvox, 1));
            dev.check(ssm_map_export(dev.ctx(), m.data(), (int)m.size(), &nvox), "ssm_map_export");
            uint64_t h = 0xCBF29CE484222325ull;
            for (int i = 0; i < nvox; i++) { const unsigned char* b = (const unsigned char*)&m[i]; for (int k = 0; k < 24; k++) { h ^= b[k]; h *= 0x100000001B3ull; } }
            cout << " map_voxels " << nvox << " map_fnv " << hex << h << dec;
        }
        cout << endl;
    } catch (const exception& e) { cerr << RED << "exp_mapping: " << e.what() << RESET << endl; return 2; }
    return 0;
}

// one rank of `--ranks N` (its own process, its own GPU)
static int run_rank(ParameterReader& parameterReader, int rank, int nranks, const string& id_dir)
{
    ssm::default_device() = rank;
    VisualOdometryStereo::parameters voparam;
    voparam.calib.f = parameterReader.getData<double>("camera.fx"); voparam.calib.cu = parameterReader.getData<double>("camera.cx");
    voparam.calib.cv = parameterReader.getData<double>("camera.cy"); voparam.base = parameterReader.getData<double>("camera.baseline", 0.0);
    voparam.inlier_threshold = parameterReader.getData<double>("inlier_threshold", 6.0);
    const int first = parameterReader.getData<int>("start_index", 0), last = parameterReader.getData<int>("end_index", 100), total = last - first;
    const int R = parameterReader.getData<int>("tracker_ref_frames", 5);
    const int base = total / nranks, rem = total % nranks;                       // contiguous blocks, earlier ranks take the remainder (sharding.frame_block)
    const int lo = first + rank * base + min(rank, rem), hi = lo + base + (rank < rem ? 1 : 0);
    const int halo_lo = max(first, lo - R);
    Tracker::Ptr tracker(new Tracker(parameterReader, voparam));
    const FrameReader::DATASET type = dataset_type(parameterReader);
    FrameReader frameReader(parameterReader, type);
    const bool use_gt_pose = parameterReader.getData<int>("use_stream_pose", (type == FrameReader::TUM || type == FrameReader::KITTI) ? 0 : 1) != 0;
    PoseGraph poseGraph(parameterReader, tracker);
    Mapper mapper(parameterReader, poseGraph);
    auto t0 = chrono::steady_clock::now();
    int nframes = 0;
    for (int i = halo_lo; i < hi; i++) {
        RGBDFrame::Ptr frame = frameReader.get(i);
        if (!frame) break;
        Eigen::Isometry3d gt = frame->T_f_w;
        tracker->updateFrame(frame);
        if (use_gt_pose) frame->setTransform(gt);
        if (i < lo) continue;                                                   // halo frame: the previous rank maps it
        poseGraph.tryInsertKeyFrame(frame);
        nframes++;
    }
    const double loop_s = chrono::duration<double>(chrono::steady_clock::now() - t0).count();
    poseGraph.shutdown();
    this_thread::sleep_for(chrono::milliseconds(parameterReader.getData<int>("mapper_drain_ms", 300)));
    mapper.shutdown();
    // ---- the merge: this rank's key-frame clouds into one context map, then ONE all-gather of the voxel tables
    ssm_config cfg = parameterReader.deviceConfig(frameReader.width, frameReader.height);
    ssm::Device dev(cfg);
    unsigned char id[SSM_COMM_ID_BYTES];
    const string id_path = id_dir + "/nccl_id";
    if (rank == 0) {
        dev.check(ssm_comm_get_unique_id(id), "ssm_comm_get_unique_id");
        const string tmp = id_path + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id) || fclose(f) || rename(tmp.c_str(), id_path.c_str())) throw runtime_error("cannot publish the communicator id in " + id_dir);
    } else {
        for (;;) {                                                               // the parent kills this process if rank 0 dies first
            FILE* f = fopen(id_path.c_str(), "rb");
            if (f) { const size_t got = fread(id, 1, sizeof(id), f); fclose(f); if (got == sizeof(id)) break; }
            this_thread::sleep_for(chrono::milliseconds(2));
        }
    }
    dev.check(ssm_comm_init_rank(dev.ctx(), nranks, rank, id), "ssm_comm_init_rank");
    size_t local_points = 0;
    for (RGBDFrame::Ptr& kf : poseGraph.keyframes) {
        Mapper::PointCloud::Ptr c = mapper.generatePointCloud(kf);
        local_points += c->points.size();
        dev.check(ssm_map_insert(dev.ctx(), reinterpret_cast<const ssm_point*>(c->points.data()), (int)c->points.size()), "ssm_map_insert");
    }
    dev.check(ssm_voxel_allgather(dev.ctx(), nullptr), "ssm_voxel_allgather");
    int nvox = 0; dev.check(ssm_map_size(dev.ctx(), &nvox), "ssm_map_size");
    vector<ssm_point> merged((size_t)max(nvox, 1));
    dev.check(ssm_map_export(dev.ctx(), merged.data(), (int)merged.size(), &nvox), "ssm_map_export");
    uint64_t h = 0xCBF29CE484222325ull;                                          // FNV-1a of the merged map: identical on every rank and for every N
    for (int i = 0; i < nvox; i++) { const unsigned char* b = (const unsigned char*)&merged[i]; for (int k = 0; k < 24; k++) { h ^= b[k]; h *= 0x100000001B3ull; } }
    dev.check(ssm_comm_finalize(dev.ctx()), "ssm_comm_finalize");
    cout << "rank " << rank << "/" << nranks << " frames [" << lo << "," << hi << ") halo " << lo - halo_lo << " keyframes " << poseGraph.keyframes.size() << " local_points " << local_points
         << " merged_voxels " << nvox << " map_fnv " << hex << h << dec << " host_loop_fps " << nframes / loop_s << endl;
    const string out = parameterReader.getData<string>("map_output", string(""));
    if (rank == 0 && !out.empty()) {
        Mapper::PointCloud pc; pc.points.resize(nvox); memcpy((void*)pc.points.data(), merged.data(), (size_t)nvox * sizeof(ssm_point)); pc.width = nvox;
        Mapper::writePCD(out, pc);
    }
    return 0;
}
