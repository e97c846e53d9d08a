// ssm/compat.h -- minimal, layout-compatible stand-ins for the third-party types that leak through the reference's
// interfaces (cv::Mat, cv::KeyPoint, cv::DMatch, cv::Point*, Eigen::Isometry3d, pcl::PointCloud; reference
// include/rgbdframe.h:38-58).  OpenCV 2.4 / Eigen / PCL are not in this image; a build that has them defines
// SSM_WITH_OPENCV / SSM_WITH_EIGEN / SSM_WITH_PCL and gets the real headers instead (same member names are used).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "../ssm_hip.h"

#ifdef SSM_WITH_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#else
namespace cv {
typedef unsigned char uchar;
typedef unsigned short ushort;
enum { CV_8U = 0, CV_16U = 2, CV_16S = 3, CV_32F = 5, CV_64F = 6 };
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(cv::CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(cv::CV_8U, 3)
#define CV_16UC1 CV_MAKETYPE(cv::CV_16U, 1)
#define CV_16SC1 CV_MAKETYPE(cv::CV_16S, 1)
#define CV_32FC1 CV_MAKETYPE(cv::CV_32F, 1)
#define CV_64F cv::CV_64F
struct Size { int width = 0, height = 0; Size() {} Size(int w, int h) : width(w), height(h) {} };
template <class T> struct Point_ { T x = 0, y = 0; Point_() {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<float> Point2f; typedef Point_<int> Point2i; typedef Point2i Point;
template <class T> struct Point3_ { T x = 0, y = 0, z = 0; Point3_() {} Point3_(T a, T b, T c) : x(a), y(b), z(c) {}
    bool operator==(const Point3_& o) const { return x == o.x && y == o.y && z == o.z; } };
typedef Point3_<float> Point3f;
struct KeyPoint { Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1; };     // 28 bytes, as OpenCV 2.4
struct DMatch { int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 0;
    bool operator<(const DMatch& m) const { return distance < m.distance; } };
// reference-counted dense matrix: only what the path touches
class Mat {
public:
    int rows = 0, cols = 0; size_t step = 0; uchar* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type) { create(s.height, s.width, type); }
    Mat(int r, int c, int type, void* user, size_t user_step = 0) {                 // wraps user memory (not owned), like cv::Mat(rows, cols, type, data, step)
        type_ = type; rows = r; cols = c; step = user_step ? user_step : (size_t)c * elemSize(); data = (uchar*)user;
    }
    // (stand-in only) ties the lifetime of wrapped user memory to the Mat and its copies; with OpenCV the owner of the buffer outlives the Mat instead
    void hold(std::shared_ptr<void> owner) { keep_ = std::move(owner); }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat zeros(Size s, int type) { return Mat(s, type); }
    void create(int r, int c, int type) {
        type_ = type; rows = r; cols = c; step = (size_t)c * elemSize();
        buf_ = std::make_shared<std::vector<uchar>>((size_t)r * step, 0); data = buf_->data();
    }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elemSize() const { static const int ds[8] = {1, 1, 2, 2, 4, 4, 8, 2}; return (size_t)ds[depth()] * channels(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Size size() const { return Size(cols, rows); }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
    template <class T> T& at(int r, int c) { return ptr<T>(r)[c]; }
    template <class T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
    Mat clone() const { Mat m; if (!empty()) { m.create(rows, cols, type_); for (int r = 0; r < rows; r++) memcpy(m.ptr<uchar>(r), ptr<uchar>(r), (size_t)cols * elemSize()); } return m; }
    Mat row(int r) const { Mat m = *this; m.rows = 1; m.data = data + (size_t)r * step; return m; }       // shares storage
    void push_back(const Mat& r) {                                                                      // append rows (getAllDescriptors); amortised like cv::Mat::push_back
        if (r.empty()) return;
        if (empty()) { *this = r.clone(); return; }
        const size_t rb = (size_t)cols * elemSize();
        if (buf_ && buf_.use_count() == 1 && data == buf_->data() && isContinuous() && r.cols == cols) {    // sole owner of a packed buffer: grow it in place (std::vector doubles its capacity)
            buf_->resize((size_t)(rows + r.rows) * step); data = buf_->data();
            for (int i = 0; i < r.rows; i++) memcpy(data + (size_t)(rows + i) * step, r.ptr<uchar>(i), rb);
            rows += r.rows;
            return;
        }
        Mat n(rows + r.rows, cols, type_);
        for (int i = 0; i < rows; i++) memcpy(n.ptr<uchar>(i), ptr<uchar>(i), rb);
        for (int i = 0; i < r.rows; i++) memcpy(n.ptr<uchar>(rows + i), r.ptr<uchar>(i), rb);
        *this = n;
    }
private:
    int type_ = 0; std::shared_ptr<std::vector<uchar>> buf_; std::shared_ptr<void> keep_;
};
template <class T> using Ptr = std::shared_ptr<T>;
}  // namespace cv
typedef unsigned char uchar;      // OpenCV exports these at global scope too (the reference uses them unqualified)
typedef unsigned short ushort;
#endif

#ifdef SSM_WITH_EIGEN
#include <Eigen/Core>
#include <Eigen/Geometry>
#else
namespace Eigen {
struct Vector3d { double v[3] = {0, 0, 0}; Vector3d() {} Vector3d(double a, double b, double c) { v[0] = a; v[1] = b; v[2] = c; }
    double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };
struct Vector4d { double v[4] = {0, 0, 0, 1}; Vector4d() {} Vector4d(double a, double b, double c, double d) { v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
    double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };
struct Matrix4d { double m[16]; double& operator()(int r, int c) { return m[c * 4 + r]; } double operator()(int r, int c) const { return m[c * 4 + r]; }
    const double* data() const { return m; } double* data() { return m; } };
// rigid transform, 4x4 column-major like Eigen::Isometry3d::matrix()
class Isometry3d {
public:
    Isometry3d() { setIdentity(); }
    static Isometry3d Identity() { return Isometry3d(); }
    void setIdentity() { for (int i = 0; i < 16; i++) M.m[i] = (i % 5 == 0) ? 1.0 : 0.0; }
    double& operator()(int r, int c) { return M(r, c); }
    double operator()(int r, int c) const { return M(r, c); }
    const Matrix4d& matrix() const { return M; }
    Matrix4d& matrix() { return M; }
    const double* data() const { return M.m; }
    Isometry3d operator*(const Isometry3d& o) const {
        Isometry3d r;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += M(i, k) * o.M(k, j); r.M(i, j) = s; }
        return r;
    }
    Vector4d operator*(const Vector4d& p) const { Vector4d r; for (int i = 0; i < 4; i++) { double s = 0; for (int k = 0; k < 4; k++) s += M(i, k) * p(k); r(i) = s; } return r; }
    Isometry3d inverse() const {                       // [R t]^-1 = [R^T  -R^T t]
        Isometry3d r;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.M(i, j) = M(j, i);
        for (int i = 0; i < 3; i++) r.M(i, 3) = -(r.M(i, 0) * M(0, 3) + r.M(i, 1) * M(1, 3) + r.M(i, 2) * M(2, 3));
        return r;
    }
private:
    Matrix4d M;
};
}  // namespace Eigen
#endif

#ifdef SSM_WITH_PCL
#include <pcl/point_types.h>
#include <pcl/point_cloud.h>
#else
namespace pcl {
// 32-byte point, pcl::PointXYZRGBL layout (a superset of PointXYZRGBA): x y z pad | b g r a | label | pad pad
struct PointXYZRGBL { float x = 0, y = 0, z = 0, data3 = 1.0f; uint8_t b = 0, g = 0, r = 0, a = 0; uint32_t label = 255; uint32_t pad_[2] = {0, 0}; };
typedef PointXYZRGBL PointXYZRGBA;
static_assert(sizeof(PointXYZRGBL) == sizeof(ssm_point), "point layout");
template <class PointT> class PointCloud {
public:
    typedef std::shared_ptr<PointCloud<PointT>> Ptr;
    std::vector<PointT> points; uint32_t width = 0, height = 1; bool is_dense = true;
    size_t size() const { return points.size(); }
    void clear() { points.clear(); width = 0; }
    void swap(PointCloud& o) { points.swap(o.points); std::swap(width, o.width); std::swap(height, o.height); std::swap(is_dense, o.is_dense); }
    PointCloud& operator+=(const PointCloud& o) { points.insert(points.end(), o.points.begin(), o.points.end()); width = (uint32_t)points.size(); height = 1; is_dense = is_dense && o.is_dense; return *this; }
};
}  // namespace pcl
#endif
