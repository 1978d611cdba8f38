// CylinderTag.h -- C++ host layer with the reference's class interface on top of the C ABI (include/ctag.h).
//
// Mirrors /root/reference/header/CylinderTag.h:12-52 and the structs of header/corner_detector.h:10-22 for the
// detection path: same class name, constructor forms, detect() signature, MarkerInfo field names, the same
// `throw std::string` error behaviour of the loaders (CylinderTag.cpp:21,39,51,61) and the same two stdout
// messages with untouched output on the early returns (CylinderTag.cpp:87-96).  loadModel / loadCamera / estimatePose
// (header/CylinderTag.h:24-30, CylinderTag.cpp:161-209) are here too, on the GPU pose back end of include/ctag_pose.h
// (EPnP + LM per marker, k_pose.hip); drawAxis is the reference's GUI and stays there.
//
// Build with -DCTAG_WITH_OPENCV to use cv::Mat / cv::Point2f / cv::Mat1i (drop-in next to the reference's
// pose_estimation.cpp); without it a minimal stand-alone Mat / Point2f is used (this image has no OpenCV).
#pragma once
#ifndef CYLINDERTAG_AMD_H
#define CYLINDERTAG_AMD_H

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#ifdef CTAG_WITH_OPENCV
#include <opencv2/core.hpp>
namespace ctag_host {
using cv::Mat;
using cv::Mat1i;
using cv::Point2f;
using cv::Point3f;
}  // namespace ctag_host
#else
namespace ctag_host {
struct Point2f {
    float x = 0.f, y = 0.f;
    Point2f() = default;
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct Point3f {
    float x = 0.f, y = 0.f, z = 0.f;
    Point3f() = default;
    Point3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
// borrowed 8-bit image view (what detect() needs from cv::Mat): one channel (gray) or three (BGR, as a camera delivers it)
struct Mat {
    int rows = 0, cols = 0;
    size_t step = 0;  // bytes per row
    const unsigned char* data = nullptr;
    int nch = 1;
    Mat() = default;
    Mat(int r, int c, const unsigned char* d, size_t s = 0, int ch = 1) : rows(r), cols(c), step(s ? s : (size_t)c * ch), data(d), nch(ch) {}
    bool empty() const { return !data || rows <= 0 || cols <= 0; }
    int channels() const { return nch; }
};
// dictionary matrix (what the reference takes as cv::Mat1i)
struct Mat1i {
    int rows = 0, cols = 0;
    std::vector<int> v;
    Mat1i() = default;
    Mat1i(int r, int c) : rows(r), cols(c), v((size_t)r * c, 0) {}
    int& operator()(int i, int j) { return v[(size_t)i * cols + j]; }
    int operator()(int i, int j) const { return v[(size_t)i * cols + j]; }
};
}  // namespace ctag_host
#endif

struct ctag_handle;
struct ctag_params;  // include/ctag_types.h: the detector's tunables (the reference's member constants, header/corner_detector.h:90-144)

// reference: header/corner_detector.h:16-22
struct MarkerInfo {
    int markerID = -1;
    std::vector<int> featurePos, feature_ID, feature_ID_left, feature_ID_right;
    std::vector<std::vector<ctag_host::Point2f>> cornerLists;
    std::vector<ctag_host::Point2f> feature_center;
    std::vector<float> edge_length, cr_left, cr_right;
};

// reference: header/pose_estimation.h:12-25.  With OpenCV the matrices are cv::Mat exactly as in the reference
// (Intrinsic / distCoeffs CV_32F as cameraParams.yml stores them, rvec / tvec 3x1 CV_64F as solvePnP creates them);
// without it plain arrays of the same element types.
#ifdef CTAG_WITH_OPENCV
struct CamInfo {
    cv::Mat Intrinsic, distCoeffs;
};
struct PoseInfo {
    int markerID;
    cv::Mat rvec, tvec;
};
#else
struct CamInfo {
    float Intrinsic[9] = {0};       // row-major 3x3
    std::vector<float> distCoeffs;  // k1 k2 p1 p2 [k3 [k4 k5 k6 [s1 s2 s3 s4]]]
};
struct PoseInfo {
    int markerID = -1;
    double rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0};
};
#endif
struct ModelInfo {
    int MarkerID = -1;
    ctag_host::Point3f axis, base;
    std::vector<ctag_host::Point3f> corners;
};

class CylinderTag {
   public:
    // Load state matrix of CylinderTag from file (reference: CylinderTag.cpp:6-9, 16-41)
    // `params` (optional, new): tunables other than the reference's constants -- what a maintainer would otherwise edit in
    // header/corner_detector.h:90,110,122,135-137,144 -- see ctag_params_default / ctag_create_ex in include/ctag.h
    CylinderTag(const std::string& path, int device_id = 0, const ctag_params* params = nullptr);
    // Manual input of the state matrix (reference: CylinderTag.cpp:11-14, 43-54).  The reference leaves
    // featureSize unset on this path; it must be given here (default 2 as in CTag_2f12c).
    CylinderTag(const ctag_host::Mat1i& set_state, int feature_size = 2, int device_id = 0, const ctag_params* params = nullptr);
    ~CylinderTag();
    CylinderTag(const CylinderTag&) = delete;
    CylinderTag& operator=(const CylinderTag&) = delete;

    // Marker Detector (reference: header/CylinderTag.h:21, CylinderTag.cpp:67-159).  A three-channel image is taken as the BGR frame the
    // reference's caller would have passed through cvtColor(BGR2GRAY) first (main.cpp:36,54): that conversion then runs on the device.
    void detect(const ctag_host::Mat& img, std::vector<MarkerInfo>& cornerList, int adaptiveThresh = 5,
                const bool cornerSubPix = false, int cornerSubPixDist = 3);

    // Batch form (new): n frames of identical size, frame i at frames + i*frame_stride; one vector per frame.
    // status[i] is CTAG_OK / CTAG_NO_CORNER / CTAG_NO_FEATURE / error; lists[i] is assigned only on CTAG_OK.
    void detectBatch(const unsigned char* frames, int n, int rows, int cols, size_t row_stride, size_t frame_stride,
                     std::vector<std::vector<MarkerInfo>>& lists, std::vector<int>& status, int adaptiveThresh = 5,
                     const bool cornerSubPix = false, int cornerSubPixDist = 3);

    // Load the reconstructed marker models / the camera (reference: header/CylinderTag.h:24,27; CylinderTag.cpp:161-196;
    // both throw std::string when the file cannot be read)
    void loadModel(const std::string& path, std::vector<ModelInfo>& reconstruct_model);
    void loadCamera(const std::string& path, CamInfo& camera);

    // Estimate the pose of the markers (reference: header/CylinderTag.h:30, CylinderTag.cpp:198-209): one PoseInfo per
    // marker that has a model, PoseInfo::markerID = index into reconstruct_model (pose_estimation.cpp:59,69).
    // useDensePoseRefine is accepted and ignored: the reference's DenseSolver is empty (pose_estimation.cpp:145-148).
    void estimatePose(const ctag_host::Mat& img, std::vector<MarkerInfo> markers, std::vector<ModelInfo> reconstruct_model, CamInfo camera,
                      std::vector<PoseInfo>& pose, bool useDensePoseRefine = false);

    int featureSize() const { return featureSize_; }
    ctag_handle* handle() const { return h_; }

   private:
    void load_from_file(const std::string path);
    void load_from_set(const ctag_host::Mat1i& set_state);
    void check_dictionary(const std::vector<int>& state);
    void create(int device_id, const ctag_params* params);

    std::vector<int> state_;
    int state_rows_ = 0, state_cols_ = 0;
    int featureSize_ = 0;
    ctag_handle* h_ = nullptr;
};

#endif
