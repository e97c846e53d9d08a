// ssm/rgbdframe.h -- rgbd_tutor::RGBDFrame and FrameReader (reference include/rgbdframe.h:26-186, src/rgbdframe.cpp).
// RGBDFrame keeps the reference's public field and method names.  Not reproduced: the per-frame imread of an
// absolute-path color.png in the constructor (rgbdframe.h:29-32) and the DBoW2 bag-of-words vector (loop closure, out of scope).
#pragma once
#include "common_headers.h"
#include "device.h"
#include "feature.h"
#include "parameter_reader.h"
#include "utils.h"
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
    cv::Mat getAllDescriptors() const { cv::Mat desp; for (size_t i = 0; i < features.size(); i++) desp.push_back(features[i].descriptor); return desp; }
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
        camera = para.getCamera(); currentIndex = start_index;
        if (dataset_type == TUM || dataset_type == KITTI || dataset_type == NYUD)
            cerr << RED << "FrameReader: TUM/KITTI/NYUD layouts need PNG decoding (not built, SURVEY.md s.8f); use SYNTHETIC or RAW" << RESET << endl;
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
        f->rgb.create(height, width, CV_8UC3); f->depth.create(height, width, CV_16UC1); f->semantic.create(height, width, CV_8UC3);
        const size_t np = (size_t)width * height;
        if (dataset_type == SYNTHETIC) {
            if (!dev) { ssm_config c = parameterReader.deviceConfig(width, height); dev.reset(new ssm::Device(c)); ssm_ctx* x = dev->ctx();
                        dev->check(ssm_dev_alloc(x, np * 3, &d_bgr), "alloc"); dev->check(ssm_dev_alloc(x, np * 2, &d_dep), "alloc");
                        dev->check(ssm_dev_alloc(x, np * 3, &d_sem), "alloc"); dev->check(ssm_dev_alloc(x, 128, &d_pose), "alloc"); }
            ssm_ctx* x = dev->ctx();
            dev->check(ssm_synth_frames_dev(x, seed, index, 1, width, height, (uint8_t*)d_bgr, (uint16_t*)d_dep, (uint8_t*)d_sem, nullptr, (double*)d_pose), "synth");
            dev->check(ssm_memcpy_d2h(x, f->rgb.data, d_bgr, np * 3), "d2h"); dev->check(ssm_memcpy_d2h(x, f->depth.data, d_dep, np * 2), "d2h");
            dev->check(ssm_memcpy_d2h(x, f->semantic.data, d_sem, np * 3), "d2h");
            double T[16]; dev->check(ssm_memcpy_d2h(x, T, d_pose, 128), "d2h");
            for (int i = 0; i < 16; i++) f->T_f_w.matrix().data()[i] = T[i];      // ground-truth pose of the stream
        } else if (dataset_type == RAW) {
            char name[64]; snprintf(name, sizeof(name), "%06d", index);
            if (!readAll(dataset_dir + name + ".bgr", f->rgb.data, np * 3) || !readAll(dataset_dir + name + ".depth", f->depth.data, np * 2) ||
                !readAll(dataset_dir + name + ".sem", f->semantic.data, np * 3)) return nullptr;                      // missing file -> nullptr, like the reference
        } else return nullptr;
        f->raw_semantic = f->semantic; f->result = f->rgb;
        return f;
    }
    static bool readAll(const string& path, void* dst, size_t n) { ifstream in(path, ios::binary); if (!in) return false; in.read((char*)dst, (streamsize)n); return (size_t)in.gcount() == n; }
    const ParameterReader& parameterReader;
    DATASET dataset_type; int currentIndex = 0, start_index = 0, end_index = 0; uint64_t seed = 0; string dataset_dir;
    CAMERA_INTRINSIC_PARAMETERS camera;
    unique_ptr<ssm::Device> dev; void *d_bgr = nullptr, *d_dep = nullptr, *d_sem = nullptr, *d_pose = nullptr;
};
}  // namespace rgbd_tutor
