// DECLARATION-ONLY stand-in for <opencv2/core.hpp>, used by tests/test_abi_cpu.py for ONE purpose: to push the
// -DCTAG_WITH_OPENCV branch of cylindertag_amd/csrc/CylinderTag.{h,cpp} (the drop-in build next to the reference's
// pose_estimation.cpp, INTEGRATION.md option A) through a compiler's syntax and type checks in an image that has no
// OpenCV.  It declares the handful of cv:: names that branch uses with the signatures of OpenCV 4.x; nothing is defined,
// nothing links, nothing runs, and it pins nothing about OpenCV's behaviour.  It is NOT used to build the reference, the
// oracle or the product.
#pragma once
#include <cstddef>

#define CV_8UC1 0
#define CV_32F 5
#define CV_64F 6

namespace cv {

template <typename T>
struct Point_ {
    T x, y;
    Point_();
    Point_(T x_, T y_);
};
typedef Point_<float> Point2f;
template <typename T>
struct Point3_ {
    T x, y, z;
    Point3_();
    Point3_(T x_, T y_, T z_);
};
typedef Point3_<float> Point3f;

class Mat;
template <typename T>
class MatCommaInitializer_;

class Mat {
public:
    Mat();
    Mat(int rows, int cols, int type);
    Mat(int rows, int cols, int type, void* data, size_t step = 0);
    Mat(const Mat&);
    Mat& operator=(const Mat&);
    ~Mat();
    Mat clone() const;
    void convertTo(Mat& m, int rtype, double alpha = 1, double beta = 0) const;
    size_t total() const;
    bool empty() const;
    int channels() const;
    template <typename T> T* ptr(int i0 = 0);
    template <typename T> const T* ptr(int i0 = 0) const;
    template <typename T> T& at(int i0, int i1);
    template <typename T> const T& at(int i0, int i1) const;
    int rows, cols;
    unsigned char* data;
    struct MatStep {
        operator size_t() const;
    } step;
};

template <typename T>
class Mat_ : public Mat {
public:
    Mat_();
    Mat_(int rows, int cols);
    Mat_(const Mat&);
    T& operator()(int i0, int i1);
    const T& operator()(int i0, int i1) const;
};
typedef Mat_<int> Mat1i;

template <typename T>
class MatCommaInitializer_ {
public:
    template <typename T2> MatCommaInitializer_<T>& operator,(T2 v);
    operator Mat_<T>() const;
    operator Mat() const;
};
template <typename T, typename T2>
MatCommaInitializer_<T> operator<<(const Mat_<T>& m, T2 val);

}  // namespace cv
