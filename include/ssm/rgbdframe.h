// ssm/rgbdframe.h -- rgbd_tutor::RGBDFrame and FrameReader (reference include/rgbdframe.h:26-186, src/rgbdframe.cpp).
// RGBDFrame keeps the reference's public field and method names.  Not reproduced: the per-frame imread of an
// absolute-path color.png in the constructor (rgbdframe.h:29-32) and the DBoW2 bag-of-words vector (loop closure, out of scope).
#pragma once
#include "common_headers.h"
#include "device.h"
#include "feature.h"
#include "parameter_reader.h"
#include "utils.h"
#include "png_io.h"
#include "stereo.h"
namespace rgbd_tutor {
class RGBDFrame {
public:
    typedef shared_ptr<RGBDFrame> Ptr;
    typedef pcl::PointCloud<pcl::PointXYZRGBA> PointCloud;
    RGBDFrame() {}
    int id = -1;
    cv::Mat rgb, depth, disparity, semantic, raw_semantic, color;
    cv::Mat result;
    cv::Mat img_lc, img_lp, img_rc, img_rp;
    cv::Mat rgb_pre_r, rgb_cur_r, semantic_pre_r, semantic_cur_r;
    cv::Mat xyz, roi_mask, ground_mask, moving_mask;
    Eigen::Isometry3d T_f_w = Eigen::Isometry3d::Identity();
    std::mutex mutexT;
    vector<Feature> features;
    CAMERA_INTRINSIC_PARAMETERS camera;
    PointCloud::Ptr pointcloud = nullptr;

    // pin-hole unprojection, reference include/rgbdframe.h:63-75 (host copy of the arithmetic the kernels use)
    cv::Point3f project2dTo3d(int u, int v) const {
        if (depth.data == nullptr) return cv::Point3f(0, 0, 0);
        ushort d = depth.ptr<ushort>(v)[u];
        if (d == 0) return cv::Point3f(0, 0, 0);
        cv::Point3f p;
        p.z = (float)(double(d) / camera.scale);
        p.x = (float)((u - camera.cx) * p.z / camera.fx);
        p.y = (float)((v - camera.cy) * p.z / camera.fy);
        return p;
    }
    // (detectFeatures leaves the frame's descriptors as ONE block, of which Feature::descriptor are the rows: the block is returned when it still matches the features)
    cv::Mat getAllDescriptors() const {
        if (!features.empty() && descriptors_all.rows == (int)features.size() && features.front().descriptor.data == descriptors_all.data) return descriptors_all;
        cv::Mat desp; for (size_t i = 0; i < features.size(); i++) desp.push_back(features[i].descriptor); return desp; }
    cv::Mat descriptors_all;                             // not in the reference's struct: see getAllDescriptors
    vector<cv::Mat> getAllDescriptorsVec() const { vector<cv::Mat> d; for (auto& f : features) d.push_back(f.descriptor); return d; }
    vector<cv::KeyPoint> getAllKeypoints() const { vector<cv::KeyPoint> k; for (auto& f : features) k.push_back(f.keypoint); return k; }
    void setTransform(const Eigen::Isometry3d& T) { std::unique_lock<std::mutex> lck(mutexT); T_f_w = T; }
    Eigen::Isometry3d getTransform() { std::unique_lock<std::mutex> lck(mutexT); return T_f_w; }
};

// Sequential frame source.  SYNTHETIC = the seeded 640x480 RGB-D + 12-class stream of BASELINE.json configs[1]
// (generated on the device, then copied to the frame); RAW = <data_source>/<id:06d>.{bgr,depth,sem} packed dumps.
// TUM / KITTI directory layouts need PNG decoding + SGBM (src/rgbdframe.cpp:34-191): SURVEY.md s.8(f) "next" rows.
class FrameReader {
public:
    enum DATASET { NYUD = 0, TUM = 1, KITTI = 2, SYNTHETIC = 3, RAW = 4 };
    FrameReader(const ParameterReader& para, const DATASET& dataset_type = SYNTHETIC) : parameterReader(para), dataset_type(dataset_type) {
        start_index = para.getData<int>("start_index", 0); end_index = para.getData<int>("end_index", 100);
        width = para.getData<int>("image_width", 640); height = para.getData<int>("image_height", 480);
        seed = (uint64_t)para.getData<unsigned long long>("synthetic_seed", 0x5EED0000ull);
        dataset_dir = para.getData<string>("data_source", string("./"));
        pinned = para.getData<int>("reader_pinned", 1) != 0;
        rigid = para.getData<int>("synthetic_rigid", 0) != 0; rigid_period = max(1, para.getData<int>("sequence_length", 20));
        camera = para.getCamera(); currentIndex = start_index;
        if (dataset_type == TUM) init_tum();
        else if (dataset_type == KITTI) init_kitti();
        else if (dataset_type == NYUD) cerr << RED << "FrameReader: the NYUD branch is empty in the reference too (rgbdframe.cpp:9-13)" << RESET << endl;
    }
    RGBDFrame::Ptr next() {
        if (currentIndex < start_index || currentIndex >= end_index) return nullptr;
        RGBDFrame::Ptr f = load(currentIndex);
        if (f) currentIndex++;
        return f;
    }
    void reset() { currentIndex = start_index; }
    RGBDFrame::Ptr get(const int& index) { return (index < 0) ? nullptr : load(index); }
    int width = 640, height = 480;
protected:
    RGBDFrame::Ptr load(int index) {
        RGBDFrame::Ptr f(new RGBDFrame);
        f->id = index; f->camera = camera;
        const size_t np = (size_t)width * height;
        if (pinned && (dataset_type == SYNTHETIC || dataset_type == RAW)) {
            // frame buffers in page-locked memory (reader_pinned = 1, the default): the device reads them where they are (ssm_host_alloc in include/ssm_hip.h) -- what a
            // capture driver that fills user buffers would hand over.  One block per frame: rgb | semantic | depth
            std::shared_ptr<void> blk = ssm::PinnedPool::instance().take(np * 8);
            uint8_t* b = (uint8_t*)blk.get();
            f->rgb = cv::Mat(height, width, CV_8UC3, b); f->semantic = cv::Mat(height, width, CV_8UC3, b + np * 3); f->depth = cv::Mat(height, width, CV_16UC1, b + np * 6);
            f->rgb.hold(blk); f->semantic.hold(blk); f->depth.hold(blk);
        } else { f->rgb.create(height, width, CV_8UC3); f->depth.create(height, width, CV_16UC1); f->semantic.create(height, width, CV_8UC3); }
        if (dataset_type == SYNTHETIC) {
            if (!dev) { ssm_config c = parameterReader.deviceConfig(width, height); dev.reset(new ssm::Device(c)); ssm_ctx* x = dev->ctx();
                        dev->check(ssm_dev_alloc(x, np * 3, &d_bgr), "alloc"); dev->check(ssm_dev_alloc(x, np * 2, &d_dep), "alloc");
                        dev->check(ssm_dev_alloc(x, np * 3, &d_sem), "alloc"); dev->check(ssm_dev_alloc(x, 128, &d_pose), "alloc"); }
            ssm_ctx* x = dev->ctx();
            if (rigid) {
                // synthetic_rigid = 1: a RIGID scene for the closed pose loop -- frame `synthetic_rigid_base` of the stream as the texture of a fronto-parallel plane at
                // synthetic_rigid_depth metres, seen by a camera that pans so that the image moves by (2, 1) px per frame (the stream's own depth pattern does not move
                // with its texture, so PnP loses track on it).  The odometry chain of the reference is not stable on a plane for long (depth error feeds back through
                // the reference poses), so the stream is a concatenation of independent sequences of `sequence_length` frames: frame k of a sequence is the base
                // image rolled by (k rows, 2 k columns); its stream pose is x_cam = x_world + k (2 Z / fx, Z / fy, 0).  bench.py's pose leg builds the same frames.
                if (base_bgr.empty()) {
                    base_bgr.create(height, width, CV_8UC3); base_sem.create(height, width, CV_8UC3);
                    dev->check(ssm_synth_frames_dev(x, seed, parameterReader.getData<int>("synthetic_rigid_base", 0), 1, width, height, (uint8_t*)d_bgr, (uint16_t*)d_dep, (uint8_t*)d_sem, nullptr, (double*)d_pose), "synth");
                    dev->check(ssm_memcpy_d2h(x, base_bgr.data, d_bgr, np * 3), "d2h"); dev->check(ssm_memcpy_d2h(x, base_sem.data, d_sem, np * 3), "d2h");
                }
                const int k = index % rigid_period, sy = k % height, sx = (2 * k) % width;
                const double Z = parameterReader.getData<double>("synthetic_rigid_depth", 2.0);
                const uint16_t dval = (uint16_t)(Z * camera.scale);
                for (int y = 0; y < height; y++) {
                    const int yo = (y + sy) % height;
                    for (int img = 0; img < 2; img++) {
                        const uint8_t* src = (img ? base_sem : base_bgr).ptr<uint8_t>(y); uint8_t* dst = (img ? f->semantic : f->rgb).ptr<uint8_t>(yo);
                        memcpy(dst + (size_t)sx * 3, src, (size_t)(width - sx) * 3); memcpy(dst, src + (size_t)(width - sx) * 3, (size_t)sx * 3);
                    }
                    uint16_t* d = f->depth.ptr<uint16_t>(y); for (int c = 0; c < width; c++) d[c] = dval;
                }
                f->T_f_w = Eigen::Isometry3d::Identity();
                f->T_f_w(0, 3) = k * 2 * Z / camera.fx; f->T_f_w(1, 3) = k * Z / camera.fy;
                f->raw_semantic = f->semantic; f->result = f->rgb;
                return f;
            }
            dev->check(ssm_synth_frames_dev(x, seed, index, 1, width, height, (uint8_t*)d_bgr, (uint16_t*)d_dep, (uint8_t*)d_sem, nullptr, (double*)d_pose), "synth");
            dev->check(ssm_memcpy_d2h(x, f->rgb.data, d_bgr, np * 3), "d2h"); dev->check(ssm_memcpy_d2h(x, f->depth.data, d_dep, np * 2), "d2h");
            dev->check(ssm_memcpy_d2h(x, f->semantic.data, d_sem, np * 3), "d2h");
            double T[16]; dev->check(ssm_memcpy_d2h(x, T, d_pose, 128), "d2h");
            for (int i = 0; i < 16; i++) f->T_f_w.matrix().data()[i] = T[i];      // ground-truth pose of the stream
        } else if (dataset_type == RAW) {
            char name[64]; snprintf(name, sizeof(name), "%06d", index);
            if (!readAll(dataset_dir + name + ".bgr", f->rgb.data, np * 3) || !readAll(dataset_dir + name + ".depth", f->depth.data, np * 2) ||
                !readAll(dataset_dir + name + ".sem", f->semantic.data, np * 3)) return nullptr;                      // missing file -> nullptr, like the reference
        } else if (dataset_type == TUM) {                    // rgbdframe.cpp:14-33: colour + unchanged (16-bit) depth PNGs named by associate.txt
            if (index < 0 || index >= (int)rgbFiles.size()) return nullptr;
            f->rgb = ssm::imreadPNG(dataset_dir + rgbFiles[index], 1);
            f->depth = ssm::imreadPNG(dataset_dir + depthFiles[index], -1);
            if (f->rgb.empty() || f->depth.empty() || f->depth.type() != CV_16UC1) return nullptr;
            f->semantic.create(f->rgb.rows, f->rgb.cols, CV_8UC3);             // TUM has no labels: an all-black image is outside the palette (label 255)
        } else if (dataset_type == KITTI) {                  // rgbdframe.cpp:34-195: frame i pairs image i+1 (current) with image i (previous)
            if (index < 0 || index + 1 >= (int)rgbFiles.size()) return nullptr;
            const string rgb_dir = parameterReader.getData<string>("rgb_dir", string("image_2/"));
            f->rgb = ssm::imreadPNG(dataset_dir + rgb_dir + rgbFiles[index + 1], 1);
            f->img_lc = ssm::imreadPNG(dataset_dir + "image_2/" + rgbFiles[index + 1], 0); f->img_lp = ssm::imreadPNG(dataset_dir + "image_2/" + rgbFiles[index], 0);
            f->img_rc = ssm::imreadPNG(dataset_dir + "image_3/" + rgbFiles[index + 1], 0); f->img_rp = ssm::imreadPNG(dataset_dir + "image_3/" + rgbFiles[index], 0);
            if (f->rgb.empty() || f->img_lc.empty() || f->img_rc.empty() || f->img_lp.empty() || f->img_rp.empty()) return nullptr;
            // depth from the current stereo pair: calDisparity_SGBM + the ROI-gated conversion (rgbdframe.cpp:81-116), on the GPU
            // (kitti_reader_depth = 0: left to the bulk stereo tracker, which computes it for a whole chunk of frames per launch)
            if (parameterReader.getData<int>("kitti_reader_depth", 1) != 0)
            stereoDepth(f->img_lc, f->img_rc, parameterReader.getData<double>("camera.baseline"), camera.cx, camera.cy, camera.fx,
                        parameterReader.getData<double>("camera.roix", 20.0), parameterReader.getData<double>("camera.roiy", 5.0),
                        parameterReader.getData<double>("camera.roiz", 40.0), camera.scale, f->depth, f->disparity);
            f->semantic = ssm::imreadPNG(dataset_dir + "segnet_0/" + rgbFiles[index + 1], 1);                         // precomputed label images, when present
            if (f->semantic.empty()) f->semantic.create(f->rgb.rows, f->rgb.cols, CV_8UC3);
        } else return nullptr;
        f->raw_semantic = f->semantic; f->result = f->rgb;
        return f;
    }
    void init_tum() {                                        // rgbdframe.cpp:198-226: lines of "rgb_time rgb_file depth_time depth_file"
        ifstream fin((dataset_dir + "/associate.txt").c_str());
        if (!fin) { cerr << RED << "FrameReader: " << dataset_dir << "/associate.txt not found (python associate.py rgb.txt depth.txt > associate.txt)" << RESET << endl; return; }
        string rgbTime, rgbFile, depthTime, depthFile;
        while (fin >> rgbTime >> rgbFile >> depthTime >> depthFile) { rgbFiles.push_back(rgbFile); depthFiles.push_back(depthFile); }
        if (!dataset_dir.empty() && dataset_dir.back() != '/') dataset_dir += "/";
        end_index = min(end_index, (int)rgbFiles.size());
    }
    void init_kitti() {                                      // rgbdframe.cpp:228-265: %06d.png for every entry of data_source/rgb_dir
        if (!dataset_dir.empty() && dataset_dir.back() != '/') dataset_dir += "/";
        const string rgb_dir = parameterReader.getData<string>("rgb_dir", string("image_2/"));
        for (int i = 0;; i++) {
            char name[32]; snprintf(name, sizeof(name), "%06d.png", i);
            ifstream probe((dataset_dir + rgb_dir + name).c_str(), ios::binary);
            if (!probe) break;
            rgbFiles.push_back(name); depthFiles.push_back(name);
        }
        end_index = min(end_index, (int)rgbFiles.size() - 1);
    }
    vector<string> rgbFiles, depthFiles;
    static bool readAll(const string& path, void* dst, size_t n) { ifstream in(path, ios::binary); if (!in) return false; in.read((char*)dst, (streamsize)n); return (size_t)in.gcount() == n; }
    const ParameterReader& parameterReader;
    DATASET dataset_type; int currentIndex = 0, start_index = 0, end_index = 0; uint64_t seed = 0; string dataset_dir;
    CAMERA_INTRINSIC_PARAMETERS camera;
    bool pinned = true, rigid = false; int rigid_period = 20; cv::Mat base_bgr, base_sem;
    unique_ptr<ssm::Device> dev; void *d_bgr = nullptr, *d_dep = nullptr, *d_sem = nullptr, *d_pose = nullptr;
};
}  // namespace rgbd_tutor
