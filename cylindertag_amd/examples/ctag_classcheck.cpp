// ctag_classcheck.cpp -- exercises the C++ class the way the reference's callers use it (header/CylinderTag.h:15-30), for
// tests/test_cpp_class_gpu.py: every constructor form, detect() on a one-channel and on a three-channel image, detectBatch,
// the loaders' error strings (CylinderTag.cpp:21,39,51,61,165), loadModel / loadCamera / estimatePose.  Prints every float with
// %.9g (a float32 survives that round trip), so the Python side compares bit for bit with the oracle's records.
//   ctag_classcheck errors <good.marker> <dir for scratch files>
//   ctag_classcheck dump <dictionary.marker> <image.bmp> [model.model cameraParams.yml]
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../csrc/CylinderTag.h"
#include "../csrc/ctag_io.h"

static void print_markers(const char* tag, const std::vector<MarkerInfo>& markers) {
    std::printf("%s markers %zu\n", tag, markers.size());
    for (const MarkerInfo& m : markers) {
        std::printf("marker id %d n %zu npos %zu\n", m.markerID, m.cornerLists.size(), m.featurePos.size());
        for (size_t j = 0; j < m.cornerLists.size(); j++) {
            std::printf("feature pos %d id %d %d %d", j < m.featurePos.size() ? m.featurePos[j] : -1, m.feature_ID[j], m.feature_ID_left[j], m.feature_ID_right[j]);
            for (int k = 0; k < 8; k++) std::printf(" %.9g %.9g", m.cornerLists[j][(size_t)k].x, m.cornerLists[j][(size_t)k].y);
            std::printf(" c %.9g %.9g len %.9g cr %.9g %.9g\n", m.feature_center[j].x, m.feature_center[j].y, m.edge_length[j], m.cr_left[j], m.cr_right[j]);
        }
    }
}

template <class F>
static void expect_throw(const char* what, F f) {
    try {
        f();
        std::printf("nothrow %s\n", what);
    } catch (const std::string& s) {  // the reference's loaders throw std::string, not std::exception
        std::string one = s;
        for (char& c : one)
            if (c == '\n') c = '|';
        std::printf("threw %s: %s\n", what, one.c_str());
    }
}

static ctag_host::Mat1i read_dictionary(const std::string& path, int& feature_size) {
    std::ifstream in(path);
    int n = 0, c = 0;
    in >> n >> c >> feature_size;
    ctag_host::Mat1i m(n, c);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < c; j++) in >> m(i, j);
    return m;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s errors <good.marker> <scratch dir> | dump <dictionary.marker> <image.bmp> [model cameraParams.yml]\n", argv[0]);
        return 2;
    }
    const std::string mode = argv[1];
    try {
        if (mode == "errors") {
            const std::string dir = argv[3];
            // CylinderTag.cpp:21 -- the file does not exist
            expect_throw("missing file", [&] { CylinderTag t(dir + "/does_not_exist.marker"); });
            // :61 through :39 -- a number outside 0..63 in a file
            {
                std::ofstream f(dir + "/bad.marker");
                f << "2 3 2\n1 2 3\n4 64 5\n";
            }
            expect_throw("bad file", [&] { CylinderTag t(dir + "/bad.marker"); });
            // :61 through :51 -- the same through the matrix constructor
            expect_throw("bad matrix", [&] {
                ctag_host::Mat1i m(2, 3);
                m(1, 1) = -1;
                CylinderTag t(m, 2);
            });
            // the loaders of the pose back end (:165; cv::FileStorage's failure in the reference)
            CylinderTag ok(argv[2]);
            std::vector<ModelInfo> model;
            CamInfo cam;
            expect_throw("missing model", [&] { ok.loadModel(dir + "/does_not_exist.model", model); });
            expect_throw("missing camera", [&] { ok.loadCamera(dir + "/does_not_exist.yml", cam); });
            return 0;
        }
        int fs = 0;
        const ctag_host::Mat1i dict = read_dictionary(argv[2], fs);
        CylinderTag marker(dict, fs);  // CylinderTag(const Mat1i&) (header/CylinderTag.h:18)
        CylinderTag from_file(argv[2]);
        const ctag_host::GrayImage g = ctag_host::read_bmp_gray(argv[3]);
        const ctag_host::Mat gray(g.rows, g.cols, g.px.data());
        std::vector<MarkerInfo> a, b, c;
        marker.detect(gray, a, 5, true, 5);
        print_markers("gray", a);
        from_file.detect(gray, b, 5, true, 5);
        print_markers("gray_from_file", b);
        // the colour frame a camera delivers: detect() takes the three-channel branch (CylinderTag.cpp:112-114 here; main.cpp:36,54 in the reference)
        std::vector<unsigned char> bgr((size_t)g.rows * g.cols * 3);
        for (size_t i = 0; i < g.px.size(); i++) bgr[3 * i] = bgr[3 * i + 1] = bgr[3 * i + 2] = g.px[i];
        const ctag_host::Mat colour(g.rows, g.cols, bgr.data(), 0, 3);
        marker.detect(colour, c, 5, true, 5);
        print_markers("bgr", c);
        // early returns leave the caller's vector untouched (CylinderTag.cpp:87-96)
        std::vector<unsigned char> blank((size_t)g.rows * g.cols, 200);
        std::vector<MarkerInfo> keep = a;
        marker.detect(ctag_host::Mat(g.rows, g.cols, blank.data()), keep, 5, true, 5);
        print_markers("after_blank", keep);
        // detectBatch: the image, a blank frame, the image again
        std::vector<unsigned char> three;
        three.insert(three.end(), g.px.begin(), g.px.end());
        three.insert(three.end(), blank.begin(), blank.end());
        three.insert(three.end(), g.px.begin(), g.px.end());
        std::vector<std::vector<MarkerInfo>> lists;
        std::vector<int> status;
        marker.detectBatch(three.data(), 3, g.rows, g.cols, (size_t)g.cols, (size_t)g.rows * g.cols, lists, status, 5, true, 5);
        for (int i = 0; i < 3; i++) {
            std::printf("batch %d status %d\n", i, status[(size_t)i]);
            print_markers("batch", lists[(size_t)i]);
        }
        if (argc > 5) {
            std::vector<ModelInfo> model;
            CamInfo cam;
            marker.loadModel(argv[4], model);
            marker.loadCamera(argv[5], cam);
            std::vector<PoseInfo> pose;
            marker.estimatePose(gray, a, model, cam, pose, false);
            std::printf("poses %zu\n", pose.size());
            for (const PoseInfo& p : pose) {
#ifdef CTAG_WITH_OPENCV
                const double *r = p.rvec.ptr<double>(0), *t = p.tvec.ptr<double>(0);
#else
                const double *r = p.rvec, *t = p.tvec;
#endif
                std::printf("pose %d rvec %.17g %.17g %.17g tvec %.17g %.17g %.17g\n", p.markerID, r[0], r[1], r[2], t[0], t[1], t[2]);
            }
        }
    } catch (const std::string& s) {
        std::cerr << "error: " << s;
        return 1;
    }
    return 0;
}
