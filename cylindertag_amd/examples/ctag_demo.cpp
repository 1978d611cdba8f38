// ctag_demo.cpp -- the detection half of the reference's demo driver (main.cpp:28-41 read_from_image) on the HIP path:
//   CylinderTag marker("CTag_2f12c.marker");  frame = imread(bmp) -> gray;  marker.detect(img_gray, markers, 5, true, 5);
// Prints one line per marker: id, then "pos:id_left:id_right" per feature, then the first corner of every feature.
// With a model and a camera file the pose half follows (main.cpp:33-34,40: loadModel, loadCamera, estimatePose) on the
// GPU pose back end: one line per pose "pose <model index> rvec tvec".  drawAxis is the reference's GUI.
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "../csrc/CylinderTag.h"
#include "../csrc/ctag_io.h"

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <dictionary.marker> <image.bmp> [adaptiveThresh=5] [cornerSubPix=1] [cornerSubPixDist=5] [model.model cameraParams.yml]\n", argv[0]);
        return 2;
    }
    try {
        CylinderTag marker(argv[1]);
        const ctag_host::GrayImage g = ctag_host::read_bmp_gray(argv[2]);
        const int at = argc > 3 ? std::atoi(argv[3]) : 5, sp = argc > 4 ? std::atoi(argv[4]) : 1, sd = argc > 5 ? std::atoi(argv[5]) : 5;
        std::vector<MarkerInfo> markers;
#ifdef CTAG_WITH_OPENCV  // the drop-in build of INTEGRATION.md option B: cv::Mat in, cv::Point2f out (oracle/ref_build.sh links and runs it)
        const cv::Mat img(g.rows, g.cols, CV_8UC1, const_cast<unsigned char*>(g.px.data()));
#else
        const ctag_host::Mat img(g.rows, g.cols, g.px.data());
#endif
        marker.detect(img, markers, at, sp != 0, sd);
        std::printf("markers %zu\n", markers.size());
        for (const MarkerInfo& m : markers) {
            std::printf("id %d n %zu :", m.markerID, m.cornerLists.size());
            for (size_t j = 0; j < m.cornerLists.size(); j++)
                std::printf(" %d:%d:%d", j < m.featurePos.size() ? m.featurePos[j] : -1, m.feature_ID_left[j], m.feature_ID_right[j]);
            std::printf(" |");
            for (size_t j = 0; j < m.cornerLists.size(); j++) std::printf(" %.9g,%.9g", m.cornerLists[j][0].x, m.cornerLists[j][0].y);
            std::printf("\n");
        }
        if (argc > 7) {
            std::vector<ModelInfo> model;
            CamInfo camera;
            marker.loadModel(argv[6], model);
            marker.loadCamera(argv[7], camera);
            std::vector<PoseInfo> pose;
            marker.estimatePose(img, markers, model, camera, pose, false);
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
