// test_threads -- the host layer's three-thread structure under a sanitizer (make SAN=tsan test_threads): the main thread inserts key-frames
// (PoseGraph::tryInsertKeyFrame) while Mapper::viewer, on its own thread, snapshots the key-frame list, builds clouds and swaps the global map, and a
// third thread polls getGlobalMap() like a visualiser would.  The reference reads `keyframes` without its mutex (/root/reference/src/mapper.cpp:114-136)
// and shares globalMap unguarded; include/ssm/mapper.h takes keyframes_mutex / mapMutex.  Device calls go to san_stub_device.cpp (no GPU here).
#include "ssm/rgbdframe.h"
#include "ssm/pose_graph.h"
#include "ssm/mapper.h"
#include <atomic>
using namespace std;
using namespace rgbd_tutor;
int main(int argc, char** argv)
{
    ParameterReader para(argc > 1 ? argv[1] : "./parameters_test.txt");
    para.set("keyframe_min_translation", "0.005"); para.set("map_output", "/tmp/ssm_san_threads.pcd");
    shared_ptr<Tracker> none;
    PoseGraph pg(para, none);
    Mapper mapper(para, pg);
    atomic<bool> stop(false); atomic<long> seen(0);
    thread poller([&] { while (!stop) { Mapper::PointCloud::Ptr m = mapper.getGlobalMap(); if (m) seen += (long)m->points.size(); this_thread::sleep_for(chrono::microseconds(200)); } });
    const int W = 64, H = 48, NF = 120;
    for (int i = 0; i < NF; i++) {
        RGBDFrame::Ptr f(new RGBDFrame);
        f->id = i; f->camera = para.getCamera();
        f->rgb.create(H, W, CV_8UC3); f->depth.create(H, W, CV_16UC1); f->semantic.create(H, W, CV_8UC3);
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { f->rgb.ptr<uchar>(y)[3 * x] = (uchar)(x + i); f->depth.ptr<uint16_t>(y)[x] = (uint16_t)(1000 + ((x * 7 + y * 3 + i) & 255)); }
        Eigen::Isometry3d T = Eigen::Isometry3d::Identity(); T(0, 3) = 0.01 * i;
        f->setTransform(T);
        pg.tryInsertKeyFrame(f);
        if (i % 8 == 0) this_thread::sleep_for(chrono::milliseconds(2));
    }
    for (int k = 0; k < 500 && mapper.updates() < 2; k++) this_thread::sleep_for(chrono::milliseconds(2));
    pg.shutdown();
    this_thread::sleep_for(chrono::milliseconds(50));
    mapper.shutdown();
    stop = true; poller.join();
    const size_t npts = mapper.getGlobalMap() ? mapper.getGlobalMap()->points.size() : 0;
    cout << "test_threads: keyframes " << pg.keyframes.size() << " map_updates " << mapper.updates() << " map_points " << npts << " polled " << seen.load() << endl;
    const bool ok = pg.keyframes.size() == (size_t)NF && mapper.updates() >= 2 && npts > 0 && !mapper.viewerFailed;
    cout << (ok ? "ALL PASSED" : "FAILED") << endl;
    return ok ? 0 : 1;
}
