// ssm/mapper.h -- rgbd_tutor::Mapper (reference include/mapper.h:15-70, src/mapper.cpp): the viewer thread that turns
// key-frames into the global voxel map.  generatePointCloud (+ semantic_motion_fuse) and the pcl::VoxelGrid filter run
// on the GPU through ssm_backproject / ssm_voxel_filter; the schedule of Mapper::viewer (src/mapper.cpp:109-162) is
// kept: every 15th update rebuilds from every 2nd key-frame, otherwise the last <= 5 key-frames are added, then one
// VoxelGrid pass over the whole map.  Differences, all deliberate: no PCL visualiser window; key-frames are read under
// keyframes_mutex (the reference polls them unlocked, mapper.cpp:114-136); the PCD path comes from `map_output`
// instead of a hard-coded absolute path (mapper.cpp:167).  The unsigned wrap of `i > keyframes.size()-6`
// (mapper.cpp:134: with fewer than 6 key-frames the incremental branch adds nothing) IS reproduced unless
// mapper_fix_incremental=1.
// Round 4: the viewer's map lives on the DEVICE (mapper_device_map, default 1).  The reference (and this class before) makes a host copy of every chosen
// key-frame's cloud, transforms it in a host loop, concatenates on the host and pushes the WHOLE map through the filter on every update.  Now a key-frame's
// camera-frame cloud is made once and stays in HBM (ssm_backproject_dev: the device form of frame->pointcloud, mapper.cpp:17-20), an update transforms
// the chosen clouds by their current poses, adds the previous centroids and filters, all on the device (ssm_viewer_map_update), and only the map that is
// published (globalMap / the PCD) is downloaded.  Same bytes: the filter's sums are exact integers, independent of the order of the points.
#pragma once
#include "common_headers.h"
#include <atomic>
#include <unordered_map>
#include "device.h"
#include "pose_graph.h"
#include "rgbdframe.h"
namespace rgbd_tutor {
class Mapper {
public:
    typedef pcl::PointXYZRGBA PointT;
    typedef pcl::PointCloud<PointT> PointCloud;
    Mapper(const ParameterReader& para, PoseGraph& graph) : parameterReader(para), poseGraph(graph) {
        resolution = para.getData<double>("mapper_resolution", 0.1);
        max_distance = para.getData<double>("mapper_max_distance", 40.0);
        area_thres = para.getData<int>("motion_area_thres", 1000);
        overlay_portion_thres = para.getData<double>("motion_overlay_portion_thres", 0.143);
        fix_incremental = para.getData<int>("mapper_fix_incremental", 0) != 0;
        invert_pose = para.getData<int>("mapper_invert_pose", 0) != 0;          // the commented alternative at mapper.cpp:89
        map_output = para.getData<string>("map_output", string(""));
        device_map = para.getData<int>("mapper_device_map", 1) != 0;
        test_fail_update = para.getData<int>("mapper_test_fail_update", -1);        // tests: the device-resident update with this number fails (-> the host path from there on)
        viewerThread = make_shared<thread>(bind(&Mapper::viewer, this));
    }
    void shutdown() { shutdownFlag = true; if (viewerThread != nullptr && viewerThread->joinable()) viewerThread->join(); }
    void SaveMap() {}                                                          // empty in the reference too (mapper.cpp:179-187)
    PointCloud::Ptr getGlobalMap() { unique_lock<mutex> lck(mapMutex); return globalMap; }
    int updates() const { return cntGlobalUpdate.load(); }

    // viewer thread (src/mapper.cpp:96-171)
    void viewer() {
        PointCloud::Ptr map(new PointCloud);
        try {
        while (!shutdownFlag) {
            size_t nkf; { unique_lock<mutex> lck(poseGraph.keyframes_mutex); nkf = poseGraph.keyframes.size(); }
            if (nkf <= (size_t)keyframe_size) { this_thread::sleep_for(chrono::milliseconds(1)); continue; }
            vector<RGBDFrame::Ptr> kfs; { unique_lock<mutex> lck(poseGraph.keyframes_mutex); kfs = poseGraph.keyframes; }
            auto t0 = chrono::steady_clock::now();
            // the key-frames this update adds (mapper.cpp:121-141)
            vector<RGBDFrame::Ptr> sel; const bool rebuild = cntGlobalUpdate % 15 == 0;
            if (rebuild) { for (size_t i = 0; i < kfs.size(); i += 2) sel.push_back(kfs[i]); }
            else if (fix_incremental) { for (int i = (int)kfs.size() - 1; i >= 0 && i > (int)kfs.size() - 6; i--) sel.push_back(kfs[i]); }
            else { for (int i = (int)kfs.size() - 1; i >= 0 && (size_t)i > kfs.size() - 6; i--) sel.push_back(kfs[i]); }      // size_t wrap as in mapper.cpp:134
            bool on_device = false;
            if (device_map) {
                // clouds stay in HBM; transform + concatenation + VoxelGrid on the device, one download of the filtered map.  Every key-frame's cloud stays resident
                // until the viewer ends (the reference keeps frame->pointcloud in host RAM): when the device runs out of memory on a long sequence -- or any
                // other device error hits this path -- the update is done on the host instead, and so are all later ones: `map` holds the last published map,
                // which is byte for byte what the host schedule would hold at this point, so the switch changes no bits
                try {
                    ssm::Device& d = device(sel.empty() ? 0 : sel[0]->depth.cols, sel.empty() ? 0 : sel[0]->depth.rows);
                    vector<ssm_cloud*> cl; vector<double> poses;
                    for (const RGBDFrame::Ptr& f : sel) {
                        cl.push_back(deviceCloud(f));
                        const Eigen::Isometry3d T = invert_pose ? f->getTransform().inverse() : f->getTransform();
                        poses.insert(poses.end(), T.data(), T.data() + 16);
                    }
                    int nmap = 0;
                    if (test_fail_update >= 0 && cntGlobalUpdate == test_fail_update) d.check(ssm_viewer_map_release(d.ctx(), 1), "ssm_viewer_map_release");
                    d.check(ssm_viewer_map_update(d.ctx(), rebuild ? 1 : 0, cl.data(), poses.data(), (int)cl.size(), (float)resolution, &nmap), "ssm_viewer_map_update");
                    PointCloud::Ptr fetched(new PointCloud);
                    fetched->points.resize((size_t)nmap);
                    int got = 0;
                    d.check(ssm_viewer_map_fetch(d.ctx(), reinterpret_cast<ssm_point*>(fetched->points.data()), nmap, &got), "ssm_viewer_map_fetch");
                    fetched->points.resize((size_t)got); fetched->width = got;
                    map->swap(*fetched);
                    cntGlobalUpdate++;
                    keyframe_size = (int)kfs.size();
                    on_device = true;
                } catch (const std::exception& e) {
                    cerr << "Mapper: the device-resident map update failed (" << e.what() << "); this and the following updates run on the host path" << endl;
                    for (auto& kv : devClouds) ssm_cloud_free(dev ? dev->ctx() : nullptr, kv.second);
                    devClouds.clear(); device_map = false; deviceMapFellBack = true;
                    if (dev) ssm_viewer_map_release(dev->ctx(), 0);      // the slabs of the freed clouds and the update's buffers go back to the device: the host path's own device work (ssm_backproject, ssm_voxel_filter) needs the room
                }
            }
            if (on_device) {
            } else {
                if (rebuild) map->clear();
                for (const RGBDFrame::Ptr& f : sel) *map += *generatePointCloud(f);
                cntGlobalUpdate++;
                PointCloud::Ptr tmp = voxelFilter(map);
                keyframe_size = (int)kfs.size();
                map->swap(*tmp);
            }
            { unique_lock<mutex> lck(mapMutex); globalMap.reset(new PointCloud(*map)); }
            const double ms = chrono::duration<double, milli>(chrono::steady_clock::now() - t0).count();
            cout << "points in global map: " << map->points.size() << endl;
            cout << "Mapping cost time: " << ms << "ms" << endl;
        }
        } catch (const std::exception& e) {             // a device error on this thread must not std::terminate the process: report it and stop mapping
            cerr << "Mapper::viewer stopped: " << e.what() << endl; viewerFailed = true;
        }
        if (poseGraph.shutDownFlag && !map_output.empty()) { writePCD(map_output, *map); cout << "Map saved!" << endl; }
        for (auto& kv : devClouds) ssm_cloud_free(dev ? dev->ctx() : nullptr, kv.second);      // the device clouds belong to this thread's context
        devClouds.clear();
    }
    // the device form of frame->pointcloud: made once per key-frame, kept in HBM (viewer thread only)
    ssm_cloud* deviceCloud(const RGBDFrame::Ptr& frame) {
        auto it = devClouds.find(frame.get());
        if (it != devClouds.end()) return it->second;
        ssm::Device& d = device(frame->depth.cols, frame->depth.rows);
        const int w = frame->depth.cols, h = frame->depth.rows;
        ssm_camera cam; cam.cx = frame->camera.cx; cam.cy = frame->camera.cy; cam.fx = frame->camera.fx; cam.fy = frame->camera.fy; cam.scale = frame->camera.scale;
        ssm_cloud* cl = nullptr;
        d.check(ssm_backproject_dev(d.ctx(), frame->depth.ptr<uint16_t>(), frame->rgb.data, frame->semantic.data, w, h, &cam, max_distance, &cl), "ssm_backproject_dev");
        devClouds[frame.get()] = cl;
        cloudsComputed++;
        return cl;
    }
    // binary PCD, FIELDS x y z rgba (what pcl::PCDWriter::write emits for PointXYZRGBA)
    static bool writePCD(const string& path, const PointCloud& c) {
        ofstream out(path, ios::binary);
        if (!out) return false;
        out << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgba\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n"
            << "WIDTH " << c.points.size() << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << c.points.size() << "\nDATA binary\n";
        for (const PointT& p : c.points) { out.write((const char*)&p.x, 12); out.write((const char*)&p.b, 4); }
        return (bool)out;
    }
    // src/mapper.cpp:12-94.  Like the reference, the gated camera-frame cloud is computed ONCE per frame (moving-class mask, gated unprojection,
    // camera colour: ssm_backproject with T = NULL) and cached in frame->pointcloud (mapper.cpp:17-20); every call returns a copy transformed by the
    // frame's current pose (pcl::transformPointCloud, mapper.cpp:91: x' = float(t00 x + t01 y + t02 z + t03) in double, left to right -- the same
    // arithmetic ssm_backproject applies on the device when it is given T, so both routes give identical bits; -ffp-contract=off in host/Makefile)
    PointCloud::Ptr generatePointCloud(const RGBDFrame::Ptr& frame) {
        if (frame->pointcloud == nullptr) {
            ssm::Device& d = device(frame->depth.cols, frame->depth.rows);
            const int w = frame->depth.cols, h = frame->depth.rows;
            ssm_camera cam; cam.cx = frame->camera.cx; cam.cy = frame->camera.cy; cam.fx = frame->camera.fx; cam.fy = frame->camera.fy; cam.scale = frame->camera.scale;
            PointCloud::Ptr pc(new PointCloud());
            pc->points.resize((size_t)w * h);
            int n = 0;
            d.check(ssm_backproject(d.ctx(), frame->depth.ptr<uint16_t>(), frame->rgb.data, frame->semantic.data, w, h, &cam, nullptr, max_distance,
                                    reinterpret_cast<ssm_point*>(pc->points.data()), w * h, &n), "ssm_backproject");
            pc->points.resize(n); pc->points.shrink_to_fit(); pc->width = n; pc->is_dense = false;
            frame->pointcloud = pc;
            cloudsComputed++;
        }
        const Eigen::Isometry3d T = invert_pose ? frame->getTransform().inverse() : frame->getTransform();
        const double* t = T.data();                                            // column-major
        PointCloud::Ptr tmp(new PointCloud(*frame->pointcloud));
        for (PointT& p : tmp->points) {
            const double x = p.x, y = p.y, z = p.z;
            p.x = (float)(t[0] * x + t[4] * y + t[8] * z + t[12]);
            p.y = (float)(t[1] * x + t[5] * y + t[9] * z + t[13]);
            p.z = (float)(t[2] * x + t[6] * y + t[10] * z + t[14]);
        }
        tmp->is_dense = false;
        return tmp;
    }
    std::atomic<bool> viewerFailed{false};
    std::atomic<bool> deviceMapFellBack{false};                              // the device-resident map update failed once: host path since (same bits)
    int cloudsComputed = 0;                                                    // device back-projections so far (each frame costs one)
    PointCloud::Ptr voxelFilter(const PointCloud::Ptr& in) {                  // pcl::VoxelGrid::filter, mapper.cpp:154-155
        PointCloud::Ptr out(new PointCloud());
        if (in->points.empty()) return out;
        ssm::Device& d = device(0, 0);
        out->points.resize(in->points.size());
        int n = 0;
        int rc = ssm_voxel_filter(d.ctx(), reinterpret_cast<const ssm_point*>(in->points.data()), (int)in->points.size(), (float)resolution,
                                  reinterpret_cast<ssm_point*>(out->points.data()), (int)out->points.size(), &n);
        if (rc == SSM_E_VOXEL_RANGE) { *out = *in; return out; }              // PCL: "Leaf size is too small ..." -> output = input
        d.check(rc, "ssm_voxel_filter");
        out->points.resize(n); out->width = n;
        return out;
    }
protected:
    ssm::Device& device(int w, int h) {
        if (!dev) { if (w == 0) { w = 640; h = 480; } dev.reset(new ssm::Device(parameterReader.deviceConfig(w, h))); }
        return *dev;
    }
    shared_ptr<thread> viewerThread = nullptr;
    const ParameterReader& parameterReader;
    PoseGraph& poseGraph;
    unique_ptr<ssm::Device> dev;
    PointCloud::Ptr globalMap; mutex mapMutex;
    int keyframe_size = 0; std::atomic<int> cntGlobalUpdate{0};       // read by other threads (updates()): atomic -- the reference's plain int is a data race
    double resolution = 0.8, max_distance = 8.0;
    std::atomic<bool> shutdownFlag{false};                           // set by shutdown() on another thread (ThreadSanitizer finding, profiles/r03_sanitizers.log)
    bool fix_incremental = false, invert_pose = false, device_map = true; int test_fail_update = -1;
    std::unordered_map<const RGBDFrame*, ssm_cloud*> devClouds;       // key-frame -> its camera-frame cloud in device memory (held until the viewer ends)
    int area_thres = 1000; double overlay_portion_thres = 0.143;
    string map_output;
};
}  // namespace rgbd_tutor
