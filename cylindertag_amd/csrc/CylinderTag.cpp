// CylinderTag.cpp -- see CylinderTag.h.  Host plumbing only: every number comes from the HIP kernels behind
// the C ABI; there is no CPU implementation of the detection path here.
#include "CylinderTag.h"

#include <cstring>
#include <fstream>
#include <iostream>

#include "ctag.h"

using ctag_host::Mat;
using ctag_host::Mat1i;
using ctag_host::Point2f;

CylinderTag::CylinderTag(const std::string& path, int device_id) {
    load_from_file(path);
    create(device_id);
}

CylinderTag::CylinderTag(const Mat1i& set_state, int feature_size, int device_id) {
    featureSize_ = feature_size;
    load_from_set(set_state);
    create(device_id);
}

CylinderTag::~CylinderTag() { ctag_destroy(h_); }

// reference: CylinderTag::load_from_file, CylinderTag.cpp:16-41
void CylinderTag::load_from_file(const std::string path) {
    std::ifstream input_file(path);
    if (!input_file.is_open()) {
        throw __FUNCTION__ + std::string(", ") + "could not open the file\n";
    }
    int marker_num = 0, marker_col = 0, feature_size = 0;
    input_file >> marker_num >> marker_col >> feature_size;
    if (marker_num < 1 || marker_col < 1) throw __FUNCTION__ + std::string(", ") + "illegal marker info\n";
    featureSize_ = feature_size;
    state_rows_ = marker_num;
    state_cols_ = marker_col;
    state_.assign((size_t)marker_num * marker_col, 0);
    for (int& i : state_) input_file >> i;
    try {
        check_dictionary(state_);
    } catch (const std::string s) {
        throw s + __FUNCTION__ + std::string(", ") + "illegal marker info\n";
    }
}

// reference: CylinderTag::load_from_set, CylinderTag.cpp:43-54
void CylinderTag::load_from_set(const Mat1i& set_state) {
    std::vector<int> v((size_t)set_state.rows * set_state.cols);
    for (int i = 0; i < set_state.rows; i++)
        for (int j = 0; j < set_state.cols; j++) v[(size_t)i * set_state.cols + j] = set_state(i, j);
    try {
        check_dictionary(v);
    } catch (const std::string s) {
        throw s + __FUNCTION__ + std::string(", ") + "illegal marker info\n";
    }
    state_ = v;
    state_rows_ = set_state.rows;
    state_cols_ = set_state.cols;
}

// reference: CylinderTag::check_dictionary, CylinderTag.cpp:56-65
void CylinderTag::check_dictionary(const std::vector<int>& input_state) {
    for (int i : input_state) {
        if (!(i >= 0 && i <= 63)) throw __FUNCTION__ + std::string(", ") + "the number in state matrix must between 0 to 63\n";
    }
}

void CylinderTag::create(int device_id) {
    std::vector<int32_t> s(state_.begin(), state_.end());
    const int st = ctag_create(s.data(), state_rows_, state_cols_, featureSize_, device_id, &h_);
    if (st != CTAG_OK) throw __FUNCTION__ + std::string(", ") + ctag_strerror(st) + "\n";
}

static void unflatten(const ctag_frame_result& r, std::vector<MarkerInfo>& out) {
    out.clear();
    for (int m = 0; m < r.n_markers; m++) {
        const ctag_marker_rec& M = r.markers[m];
        MarkerInfo mi;
        mi.markerID = M.marker_id;
        for (int j = 0; j < M.n_features; j++) {
            const ctag_feature_rec& F = r.features[M.first_feature + j];
            if (j < M.n_pos) mi.featurePos.push_back(F.pos);
            mi.feature_ID.push_back(F.id);
            mi.feature_ID_left.push_back(F.id_left);
            mi.feature_ID_right.push_back(F.id_right);
            std::vector<Point2f> c(8);
            for (int k = 0; k < 8; k++) c[k] = Point2f(F.corners[2 * k], F.corners[2 * k + 1]);
            mi.cornerLists.push_back(c);
            mi.feature_center.push_back(Point2f(F.center[0], F.center[1]));
            mi.edge_length.push_back(F.edge_length);
            mi.cr_left.push_back(F.cr_left);
            mi.cr_right.push_back(F.cr_right);
        }
        out.push_back(mi);
    }
}

// reference: CylinderTag::detect, CylinderTag.cpp:67-159
void CylinderTag::detect(const Mat& img, std::vector<MarkerInfo>& markers_info, int adaptiveThresh, const bool cornerSubPix,
                         int cornerSubPixDist) {
    ctag_frame_result res;
#ifdef CTAG_WITH_OPENCV
    const int st = ctag_detect_u8(h_, img.ptr<unsigned char>(0), img.rows, img.cols, (ptrdiff_t)img.step, adaptiveThresh, cornerSubPix ? 1 : 0,
                                  cornerSubPixDist, &res);
#else
    const int st = ctag_detect_u8(h_, img.data, img.rows, img.cols, (ptrdiff_t)img.step, adaptiveThresh, cornerSubPix ? 1 : 0, cornerSubPixDist, &res);
#endif
    if (st == CTAG_NO_CORNER) {
        std::cout << "No corner detected!" << std::endl;  // CylinderTag.cpp:88; output left untouched
        return;
    }
    if (st == CTAG_NO_FEATURE) {
        std::cout << "No feature detected!" << std::endl;  // CylinderTag.cpp:94
        return;
    }
    if (st != CTAG_OK) throw __FUNCTION__ + std::string(", ") + ctag_strerror(st) + "\n";
    unflatten(res, markers_info);  // markers_info = markers (CylinderTag.cpp:128)
}

void CylinderTag::detectBatch(const unsigned char* frames, int n, int rows, int cols, size_t row_stride, size_t frame_stride,
                              std::vector<std::vector<MarkerInfo>>& lists, std::vector<int>& status, int adaptiveThresh,
                              const bool cornerSubPix, int cornerSubPixDist) {
    std::vector<ctag_frame_result> res((size_t)n);
    const int st = ctag_detect_batch_u8(h_, frames, n, rows, cols, (ptrdiff_t)row_stride, (ptrdiff_t)frame_stride, adaptiveThresh,
                                        cornerSubPix ? 1 : 0, cornerSubPixDist, res.data());
    if (st != CTAG_OK) throw __FUNCTION__ + std::string(", ") + ctag_strerror(st) + "\n";
    lists.resize((size_t)n);
    status.assign((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        status[i] = res[i].status;
        if (res[i].status == CTAG_OK) unflatten(res[i], lists[i]);
    }
}
