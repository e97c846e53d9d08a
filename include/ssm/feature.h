// ssm/feature.h -- reference include/feature.h
#pragma once
#include "common_headers.h"
namespace rgbd_tutor {
class Feature {
public:
    Feature() {}
    cv::KeyPoint keypoint;
    cv::Mat      descriptor;           // 1 x 32, CV_8UC1
    cv::Point3f  position;             // position in 3D space, camera frame; (0,0,0) = no depth
    float        observe_frequency = 0.0;
};
}  // namespace rgbd_tutor
