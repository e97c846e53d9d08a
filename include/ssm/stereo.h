// ssm/stereo.h -- the depth-from-stereo step of the KITTI path: calDisparity_SGBM (reference include/stereo.h:15,
// src/stereo.cpp:11-30) and the disparity -> depth conversion FrameReader::next() applies right after it
// (src/rgbdframe.cpp:81-116), both on the GPU (ssm_sgbm, ssm_stereo_depth).  triangulate10D / correct3DPoints /
// setImageROI (the UV-disparity moving-object pipeline) are not rebuilt: Mapper discards their result.
#pragma once
#include "common_headers.h"
#include "device.h"
namespace ssm {
// one lazily created device context per thread (no frame passes through its ORB geometry)
inline Device& stereoDevice() {
    static thread_local std::unique_ptr<Device> dev;
    if (!dev) { ssm_config cfg; ssm_config_default(&cfg); cfg.width = 128; cfg.height = 128; cfg.orb_levels = 1; cfg.orb_features = 100; cfg.max_batch = 1; cfg.voxel_capacity_log2 = 10;
                dev.reset(new Device(cfg)); }
    return *dev;
}
}  // namespace ssm
// img_L, img_R: rectified 8-bit single-channel images; disp: CV_16SC1, disparity x 16, -16 where no disparity was accepted
inline void calDisparity_SGBM(const cv::Mat& img_L, const cv::Mat& img_R, cv::Mat& disp) {
    if (img_L.channels() != 1 || img_R.channels() != 1 || img_L.cols != img_R.cols || img_L.rows != img_R.rows) throw std::invalid_argument("calDisparity_SGBM: two 8-bit single-channel images of one size");
    ssm::Device& dev = ssm::stereoDevice();
    ssm_sgbm_params p; ssm_sgbm_params_default(&p);                          // exactly what src/stereo.cpp:16-27 sets
    disp.create(img_L.rows, img_L.cols, CV_16SC1);
    if (img_L.step != img_R.step) throw std::invalid_argument("calDisparity_SGBM: the two images must share a row stride");
    dev.check(ssm_sgbm(dev.ctx(), img_L.data, img_R.data, img_L.cols, img_L.rows, (int)img_L.step, &p, 0, disp.ptr<int16_t>()), "ssm_sgbm");
}
// the depth image FrameReader builds from the disparity (rgbdframe.cpp:81-116): CV_16UC1, depth * camera.scale inside the 3-D
// ROI (|x| < roix, |y| < roiy, 0 < z < roiz), 0 elsewhere; `disparity` receives calDisparity_SGBM's result
inline void stereoDepth(const cv::Mat& img_L, const cv::Mat& img_R, double baseline, double cu, double cv_, double f, double roix, double roiy, double roiz, double scale,
                        cv::Mat& depth, cv::Mat& disparity) {
    if (img_L.channels() != 1 || img_R.channels() != 1 || img_L.cols != img_R.cols || img_L.rows != img_R.rows || img_L.step != img_R.step) throw std::invalid_argument("stereoDepth: two 8-bit single-channel images of one size and stride");
    ssm::Device& dev = ssm::stereoDevice();
    ssm_sgbm_params p; ssm_sgbm_params_default(&p);
    depth.create(img_L.rows, img_L.cols, CV_16UC1); disparity.create(img_L.rows, img_L.cols, CV_16SC1);
    dev.check(ssm_stereo_depth(dev.ctx(), img_L.data, img_R.data, img_L.cols, img_L.rows, (int)img_L.step, &p, baseline, cu, cv_, f, roix, roiy, roiz, scale,
                               depth.ptr<uint16_t>(), disparity.ptr<int16_t>()), "ssm_stereo_depth");
}
