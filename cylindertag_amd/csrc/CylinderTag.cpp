// CylinderTag.cpp -- see CylinderTag.h.  Host plumbing only: every number comes from the HIP kernels behind
// the C ABI; there is no CPU implementation of the detection path here.
#include "CylinderTag.h"

#include <cstring>
#include <fstream>
#include <iostream>

#include "ctag.h"
#include "ctag_pose.h"

using ctag_host::Mat;
using ctag_host::Mat1i;
using ctag_host::Point2f;
using ctag_host::Point3f;

CylinderTag::CylinderTag(const std::string& path, int device_id, const ctag_params* params) {
    load_from_file(path);
    create(device_id, params);
}

CylinderTag::CylinderTag(const Mat1i& set_state, int feature_size, int device_id, const ctag_params* params) {
    featureSize_ = feature_size;
    load_from_set(set_state);
    create(device_id, params);
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

void CylinderTag::create(int device_id, const ctag_params* params) {
    std::vector<int32_t> s(state_.begin(), state_.end());
    const int st = ctag_create_ex(s.data(), state_rows_, state_cols_, featureSize_, device_id, params, &h_);
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
    const unsigned char* px = img.ptr<unsigned char>(0);
#else
    const unsigned char* px = img.data;
#endif
    const int st = img.channels() == 3
                       ? ctag_detect_bgr8(h_, px, img.rows, img.cols, (ptrdiff_t)img.step, adaptiveThresh, cornerSubPix ? 1 : 0, cornerSubPixDist, &res)
                       : ctag_detect_u8(h_, px, img.rows, img.cols, (ptrdiff_t)img.step, adaptiveThresh, cornerSubPix ? 1 : 0, cornerSubPixDist, &res);
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

// ---------------------------------------------------------------------------------------------------------------
// pose back end (include/ctag_pose.h)
// ---------------------------------------------------------------------------------------------------------------

// reference: CylinderTag::loadModel, CylinderTag.cpp:161-190
void CylinderTag::loadModel(const std::string& path, std::vector<ModelInfo>& reconstruct_model) {
    ctag_model* m = nullptr;
    if (ctag_model_load(path.c_str(), &m) != CTAG_OK) throw __FUNCTION__ + std::string(", ") + "could not open the model file\n";
    ctag_model_view v;
    ctag_model_get_view(m, &v);
    reconstruct_model.resize((size_t)v.n_models);
    for (int i = 0; i < v.n_models; i++) {
        ModelInfo& mi = reconstruct_model[(size_t)i];
        mi.MarkerID = v.marker_id[i];
        mi.base = Point3f(v.base[3 * i], v.base[3 * i + 1], v.base[3 * i + 2]);
        mi.axis = Point3f(v.axis[3 * i], v.axis[3 * i + 1], v.axis[3 * i + 2]);
        mi.corners.resize((size_t)v.model_size * 8);
        const float* c = v.corners + (size_t)i * v.model_size * 24;
        for (int j = 0; j < v.model_size * 8; j++) mi.corners[(size_t)j] = Point3f(c[3 * j], c[3 * j + 1], c[3 * j + 2]);
    }
    ctag_model_free(m);
}

// reference: CylinderTag::loadCamera, CylinderTag.cpp:192-196 (cv::FileStorage there)
void CylinderTag::loadCamera(const std::string& path, CamInfo& camera) {
    ctag_camera c;
    if (ctag_camera_load(path.c_str(), &c) != CTAG_OK) throw __FUNCTION__ + std::string(", ") + "could not read the camera file\n";
#ifdef CTAG_WITH_OPENCV
    camera.Intrinsic = cv::Mat(3, 3, CV_32F, c.K).clone();
    camera.distCoeffs = cv::Mat(c.n_dist, 1, CV_32F, c.dist).clone();
#else
    std::memcpy(camera.Intrinsic, c.K, sizeof(c.K));
    camera.distCoeffs.assign(c.dist, c.dist + c.n_dist);
#endif
}

// markers[first...] -> one flat record; returns the index of the first marker that did not fit (records hold at most
// CTAG_MAX_MARKERS markers / CTAG_MAX_FEATURES features: estimatePose walks a longer list in several records, nothing
// is dropped).  A single marker with more features than a record holds cannot come from detect() (the reference's
// father[100], corner_detector.h:143); estimatePose rejects such a list before it calls this.
static size_t flatten(const std::vector<MarkerInfo>& markers, size_t first, ctag_frame_result& r) {
    std::memset(&r, 0, sizeof(r));
    r.status = CTAG_OK;
    int nf = 0;
    size_t m = first;
    for (; m < markers.size() && r.n_markers < CTAG_MAX_MARKERS; m++) {
        const MarkerInfo& mi = markers[m];
        const int n = (int)mi.cornerLists.size();
        if (nf + n > CTAG_MAX_FEATURES) break;  // n <= CTAG_MAX_FEATURES: estimatePose validated the list
        ctag_marker_rec& M = r.markers[r.n_markers++];
        M.marker_id = mi.markerID;
        M.first_feature = nf;
        M.n_features = n;
        M.n_pos = (int)mi.featurePos.size() < n ? (int)mi.featurePos.size() : n;
        for (int j = 0; j < n; j++) {
            ctag_feature_rec& F = r.features[nf + j];
            F.pos = j < M.n_pos ? mi.featurePos[(size_t)j] : -1;
            F.id = j < (int)mi.feature_ID.size() ? mi.feature_ID[(size_t)j] : -1;
            F.id_left = j < (int)mi.feature_ID_left.size() ? mi.feature_ID_left[(size_t)j] : -1;
            F.id_right = j < (int)mi.feature_ID_right.size() ? mi.feature_ID_right[(size_t)j] : -1;
            for (int k = 0; k < 8 && k < (int)mi.cornerLists[(size_t)j].size(); k++) {
                F.corners[2 * k] = mi.cornerLists[(size_t)j][(size_t)k].x;
                F.corners[2 * k + 1] = mi.cornerLists[(size_t)j][(size_t)k].y;
            }
        }
        nf += n;
    }
    r.n_features = nf;
    return m;
}

// reference: CylinderTag::estimatePose, CylinderTag.cpp:198-209 (+ PoseEstimator::PnPSolver / PoseBA)
void CylinderTag::estimatePose(const Mat& img, std::vector<MarkerInfo> markers, std::vector<ModelInfo> reconstruct_model, CamInfo camera,
                               std::vector<PoseInfo>& pose, bool useDensePoseRefine) {
    (void)img;
    (void)useDensePoseRefine;
    pose.clear();
    if (markers.empty()) return;
    for (const MarkerInfo& mi : markers)  // checked before anything is allocated: flatten() cannot fail afterwards
        if (mi.cornerLists.size() > (size_t)CTAG_MAX_FEATURES) throw std::string("estimatePose, a marker with more than 100 features\n");
    // vector<ModelInfo> -> ctag_model (every model must hold the same number of corners, as loadModel produces)
    const size_t nm = reconstruct_model.size();
    const size_t per = nm ? reconstruct_model[0].corners.size() : 8;
    std::vector<int32_t> ids(nm);
    std::vector<float> base(nm * 3), axis(nm * 3), corners(nm * per * 3);
    for (size_t i = 0; i < nm; i++) {
        const ModelInfo& mi = reconstruct_model[i];
        if (mi.corners.size() != per || per % 8 != 0) throw __FUNCTION__ + std::string(", ") + "illegal model\n";
        ids[i] = mi.MarkerID;
        base[3 * i] = mi.base.x, base[3 * i + 1] = mi.base.y, base[3 * i + 2] = mi.base.z;
        axis[3 * i] = mi.axis.x, axis[3 * i + 1] = mi.axis.y, axis[3 * i + 2] = mi.axis.z;
        for (size_t j = 0; j < per; j++) {
            corners[(i * per + j) * 3] = mi.corners[j].x;
            corners[(i * per + j) * 3 + 1] = mi.corners[j].y;
            corners[(i * per + j) * 3 + 2] = mi.corners[j].z;
        }
    }
    ctag_model_view v{(int32_t)nm, (int32_t)(per / 8 ? per / 8 : 1), ids.data(), base.data(), axis.data(), corners.data()};
    ctag_model* model = nullptr;
    if (ctag_model_create(&v, &model) != CTAG_OK) throw __FUNCTION__ + std::string(", ") + "illegal model\n";
    ctag_camera cam;
    std::memset(&cam, 0, sizeof(cam));
#ifdef CTAG_WITH_OPENCV
    cv::Mat Kf, Df;
    camera.Intrinsic.convertTo(Kf, CV_32F);
    camera.distCoeffs.convertTo(Df, CV_32F);
    for (int i = 0; i < 9; i++) cam.K[i] = Kf.at<float>(i / 3, i % 3);
    cam.n_dist = (int)Df.total() > 14 ? 14 : (int)Df.total();
    for (int i = 0; i < cam.n_dist; i++) cam.dist[i] = Df.ptr<float>(0)[i];
#else
    std::memcpy(cam.K, camera.Intrinsic, sizeof(cam.K));
    cam.n_dist = camera.distCoeffs.size() > 14 ? 14 : (int)camera.distCoeffs.size();
    for (int i = 0; i < cam.n_dist; i++) cam.dist[i] = camera.distCoeffs[(size_t)i];
#endif
    std::vector<ctag_pose_rec> rec;
    for (size_t first = 0; first < markers.size();) {  // any number of markers: one record (<= 100 markers / features) at a time
        ctag_frame_result res;
        const size_t next = flatten(markers, first, res);
        const size_t at = rec.size();
        rec.resize(at + (size_t)res.n_markers);
        const int st = ctag_estimate_pose(h_, &res, model, &cam, rec.data() + at);
        if (st != CTAG_OK) {
            ctag_model_free(model);
            throw __FUNCTION__ + std::string(", ") + ctag_strerror(st) + "\n";
        }
        first = next;
    }
    ctag_model_free(model);
    for (const ctag_pose_rec& p : rec) {
        if (p.status == CTAG_POSE_NO_MODEL) continue;  // pose.erase(remove_if(markerID == -1)), CylinderTag.cpp:206-208
        if (p.status != CTAG_POSE_OK)  // cv::solvePnP throws on < 4 points; an out-of-model position is UB in the reference
            throw __FUNCTION__ + std::string(", ") + "marker without a usable point set\n";
        PoseInfo pi;
        pi.markerID = p.model_index;
#ifdef CTAG_WITH_OPENCV
        pi.rvec = (cv::Mat_<double>(3, 1) << p.rvec[0], p.rvec[1], p.rvec[2]);
        pi.tvec = (cv::Mat_<double>(3, 1) << p.tvec[0], p.tvec[1], p.tvec[2]);
#else
        for (int i = 0; i < 3; i++) {
            pi.rvec[i] = p.rvec[i];
            pi.tvec[i] = p.tvec[i];
        }
#endif
        pose.push_back(pi);
    }
}
