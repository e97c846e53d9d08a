// exp_mapping -- the reference's product driver (experiment/exp_mapping.cpp:18-59) on the MI355X front end: build the
// ParameterReader, Tracker, FrameReader, PoseGraph and Mapper, then loop next() -> updateFrame() -> tryInsertKeyFrame(),
// finally shut the graph and the mapper down.  Differences: no cv::imshow; the frame source is selected by `dataset`
// (synthetic | raw | tum | kitti; the reference hard-codes its FrameReader type); with tum / kitti the poses come from the
// tracker (use_stream_pose defaults to 0 there); prints frames/s at the end.
#include "ssm/rgbdframe.h"
#include "ssm/track.h"
#include "ssm/pose_graph.h"
#include "ssm/common_headers.h"
#include "ssm/mapper.h"
#include "ssm/vo_stereo.hpp"
using namespace std;
using namespace rgbd_tutor;

int main(int argc, char** argv)
{
    ParameterReader parameterReader(argc > 1 ? argv[1] : "./parameters.txt");
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
        const string ds = parameterReader.getData<string>("dataset", string("synthetic"));
        const FrameReader::DATASET type = ds == "raw" ? FrameReader::RAW : ds == "tum" ? FrameReader::TUM : ds == "kitti" ? FrameReader::KITTI : FrameReader::SYNTHETIC;
        FrameReader frameReader(parameterReader, type);
        PoseGraph poseGraph(parameterReader, tracker);
        Mapper mapper(parameterReader, poseGraph);
        const bool use_gt_pose = parameterReader.getData<int>("use_stream_pose", (type == FrameReader::TUM || type == FrameReader::KITTI) ? 0 : 1) != 0;
        int nframes = 0;
        auto t0 = chrono::steady_clock::now();
        while (RGBDFrame::Ptr frame = frameReader.next()) {
            Eigen::Isometry3d gt = frame->T_f_w;
            tracker->updateFrame(frame);
            if (use_gt_pose) frame->setTransform(gt);           // synthetic stream: poses are given, the tracker only produces features/matches
            poseGraph.tryInsertKeyFrame(frame);
            if (tracker->getState() == Tracker::LOST) cout << "tracker is lost" << endl;
            nframes++;
        }
        const double s = chrono::duration<double>(chrono::steady_clock::now() - t0).count();
        mapper.SaveMap();
        poseGraph.shutdown();
        this_thread::sleep_for(chrono::milliseconds(parameterReader.getData<int>("mapper_drain_ms", 300)));
        mapper.shutdown();
        cout << "frames " << nframes << " keyframes " << poseGraph.keyframes.size() << " map_updates " << mapper.updates()
             << " map_points " << (mapper.getGlobalMap() ? mapper.getGlobalMap()->points.size() : 0) << " host_loop_fps " << nframes / s << endl;
    } catch (const exception& e) { cerr << RED << "exp_mapping: " << e.what() << RESET << endl; return 2; }
    return 0;
}
