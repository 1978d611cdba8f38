// CylinderTag.h -- C++ host layer with the reference's class interface on top of the C ABI (include/ctag.h).
//
// Mirrors /root/reference/header/CylinderTag.h:12-52 and the structs of header/corner_detector.h:10-22 for the
// detection path: same class name, constructor forms, detect() signature, MarkerInfo field names, the same
// `throw std::string` error behaviour of the loaders (CylinderTag.cpp:21,39,51,61) and the same two stdout
// messages with untouched output on the early returns (CylinderTag.cpp:87-96).  loadModel / loadCamera /
// estimatePose / drawAxis are the reference's pose back end and GUI: out of scope, they stay in the reference.
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
}  // namespace ctag_host
#else
namespace ctag_host {
struct Point2f {
    float x = 0.f, y = 0.f;
    Point2f() = default;
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};
// borrowed 8-bit single-channel image view (what detect() needs from cv::Mat)
struct Mat {
    int rows = 0, cols = 0;
    size_t step = 0;  // bytes per row
    const unsigned char* data = nullptr;
    Mat() = default;
    Mat(int r, int c, const unsigned char* d, size_t s = 0) : rows(r), cols(c), step(s ? s : (size_t)c), data(d) {}
    bool empty() const { return !data || rows <= 0 || cols <= 0; }
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

// reference: header/corner_detector.h:16-22
struct MarkerInfo {
    int markerID = -1;
    std::vector<int> featurePos, feature_ID, feature_ID_left, feature_ID_right;
    std::vector<std::vector<ctag_host::Point2f>> cornerLists;
    std::vector<ctag_host::Point2f> feature_center;
    std::vector<float> edge_length, cr_left, cr_right;
};

class CylinderTag {
   public:
    // Load state matrix of CylinderTag from file (reference: CylinderTag.cpp:6-9, 16-41)
    CylinderTag(const std::string& path, int device_id = 0);
    // Manual input of the state matrix (reference: CylinderTag.cpp:11-14, 43-54).  The reference leaves
    // featureSize unset on this path; it must be given here (default 2 as in CTag_2f12c).
    CylinderTag(const ctag_host::Mat1i& set_state, int feature_size = 2, int device_id = 0);
    ~CylinderTag();
    CylinderTag(const CylinderTag&) = delete;
    CylinderTag& operator=(const CylinderTag&) = delete;

    // Marker Detector (reference: header/CylinderTag.h:21, CylinderTag.cpp:67-159)
    void detect(const ctag_host::Mat& img, std::vector<MarkerInfo>& cornerList, int adaptiveThresh = 5,
                const bool cornerSubPix = false, int cornerSubPixDist = 3);

    // Batch form (new): n frames of identical size, frame i at frames + i*frame_stride; one vector per frame.
    // status[i] is CTAG_OK / CTAG_NO_CORNER / CTAG_NO_FEATURE / error; lists[i] is assigned only on CTAG_OK.
    void detectBatch(const unsigned char* frames, int n, int rows, int cols, size_t row_stride, size_t frame_stride,
                     std::vector<std::vector<MarkerInfo>>& lists, std::vector<int>& status, int adaptiveThresh = 5,
                     const bool cornerSubPix = false, int cornerSubPixDist = 3);

    int featureSize() const { return featureSize_; }
    ctag_handle* handle() const { return h_; }

   private:
    void load_from_file(const std::string path);
    void load_from_set(const ctag_host::Mat1i& set_state);
    void check_dictionary(const std::vector<int>& state);
    void create(int device_id);

    std::vector<int> state_;
    int state_rows_ = 0, state_cols_ = 0;
    int featureSize_ = 0;
    ctag_handle* h_ = nullptr;
};

#endif
