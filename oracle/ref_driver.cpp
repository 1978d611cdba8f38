// ref_driver.cpp -- dump tool linked against the REAL reference (oracle/ref_build.sh; test infrastructure).
// Runs the reference's own classes -- CylinderTag::detect (CylinderTag.cpp:67-159) and, stage by stage, the public
// methods of corner_detector (header/corner_detector.h:38-58) with the exact calls detect() makes -- on one image and
// writes every stage as a flat little-endian file that tests/test_oracle_ref_cpu.py compares with the CPU restatement:
//   <out>.half.bin        int32 rows, cols, then rows*cols u8        cv::resize(..., INTER_CUBIC)      CylinderTag.cpp:79
//   <out>.binary.bin      int32 rows, cols, then rows*cols u8        adaptiveThreshold                 CylinderTag.cpp:83
//   <out>.components.bin  int32 n, then per component int32 npix and npix (x,y) int32 pairs, reference order  :84
//   <out>.quads.bin       int32 n, then n*8 float                    edgeExtraction                    CylinderTag.cpp:86
//   <out>.result.bin      one ctag_frame_result (include/ctag_types.h), flattened like oracle/ctag_oracle.cpp:flatten
// Contains no reference code: only calls into it.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "CylinderTag.h"   // the reference's header (-I/root/reference/header)
#include "ctag_types.h"

static void flatten(const std::vector<MarkerInfo>& markers, int status, ctag_frame_result* out) {
    std::memset(out, 0, sizeof(*out));
    out->status = status;
    int nf = 0, nm = 0;
    for (const MarkerInfo& m : markers) {
        const int n = (int)m.cornerLists.size();
        if (nm >= CTAG_MAX_MARKERS || nf + n > CTAG_MAX_FEATURES) break;
        ctag_marker_rec& mr = out->markers[nm++];
        mr.marker_id = m.markerID;
        mr.first_feature = nf;
        mr.n_features = n;
        mr.n_pos = (int)m.featurePos.size();
        for (int j = 0; j < n; j++) {
            ctag_feature_rec& fr = out->features[nf + j];
            fr.pos = j < (int)m.featurePos.size() ? m.featurePos[j] : -1;
            fr.id = m.feature_ID[j];
            fr.id_left = m.feature_ID_left[j];
            fr.id_right = m.feature_ID_right[j];
            for (int k = 0; k < 8; k++) {
                fr.corners[2 * k] = m.cornerLists[j][k].x;
                fr.corners[2 * k + 1] = m.cornerLists[j][k].y;
            }
            fr.center[0] = m.feature_center[j].x;
            fr.center[1] = m.feature_center[j].y;
            fr.edge_length = m.edge_length[j];
            fr.cr_left = m.cr_left[j];
            fr.cr_right = m.cr_right[j];
        }
        nf += n;
    }
    out->n_markers = nm;
    out->n_features = nf;
}

static FILE* open_out(const std::string& base, const char* ext) {
    FILE* f = std::fopen((base + ext).c_str(), "wb");
    if (!f) {
        std::perror((base + ext).c_str());
        std::exit(2);
    }
    return f;
}
static void put_i32(FILE* f, int v) { std::fwrite(&v, 4, 1, f); }
static void put_mat_u8(const std::string& base, const char* ext, const cv::Mat& m) {
    FILE* f = open_out(base, ext);
    put_i32(f, m.rows);
    put_i32(f, m.cols);
    for (int y = 0; y < m.rows; y++) std::fwrite(m.ptr<uchar>(y), 1, (size_t)m.cols, f);
    std::fclose(f);
}

int main(int argc, char** argv) {
    if (argc != 4) {
        std::fprintf(stderr, "usage: ref_driver dictionary.marker image out_prefix\n");
        return 2;
    }
    const std::string out = argv[3];
    cv::Mat gray = cv::imread(argv[2], cv::IMREAD_GRAYSCALE);
    if (gray.empty()) {
        std::fprintf(stderr, "cannot read %s\n", argv[2]);
        return 2;
    }
    // ---- stage by stage, with the calls of CylinderTag::detect (canonical parameters of main.cpp:39: 5, true, 5)
    cv::Mat half, half_f;
    cv::resize(gray, half, cv::Size(gray.cols / 2, gray.rows / 2), 0.5, 0.5, cv::INTER_CUBIC);
    half.convertTo(half_f, CV_32FC1, 1.0 / 255);
    put_mat_u8(out, ".half.bin", half);
    corner_detector det;
    cv::Mat binary(half_f.rows, half_f.cols, CV_8UC1);  // preallocated by the caller, CylinderTag.cpp:82
    det.adaptiveThreshold(half_f, binary, 5);
    put_mat_u8(out, ".binary.bin", binary);
    std::vector<std::vector<cv::Point>> comps;
    det.connectedComponentLabeling(binary, comps);
    {
        FILE* f = open_out(out, ".components.bin");
        put_i32(f, (int)comps.size());
        for (const auto& c : comps) {
            put_i32(f, (int)c.size());
            for (const auto& p : c) {
                put_i32(f, p.x);
                put_i32(f, p.y);
            }
        }
        std::fclose(f);
    }
    std::vector<std::vector<cv::Point2f>> quads;
    det.edgeExtraction(half_f, comps, quads);
    {
        FILE* f = open_out(out, ".quads.bin");
        put_i32(f, (int)quads.size());
        for (const auto& q : quads)
            for (int k = 0; k < 4; k++) {
                std::fwrite(&q[k].x, 4, 1, f);
                std::fwrite(&q[k].y, 4, 1, f);
            }
        std::fclose(f);
    }
    // ---- the whole path through the reference's own façade
    CylinderTag marker(argv[1]);
    std::vector<MarkerInfo> markers;
    const MarkerInfo sentinel_probe;  // detect() leaves the vector untouched on its early returns (CylinderTag.cpp:87-96)
    markers.push_back(sentinel_probe);
    markers[0].markerID = -12345;
    marker.detect(gray, markers, 5, true, 5);
    int status = CTAG_OK;
    if (markers.size() == 1 && markers[0].markerID == -12345) {
        status = quads.empty() ? CTAG_NO_CORNER : CTAG_NO_FEATURE;
        markers.clear();
    }
    ctag_frame_result rec;
    flatten(markers, status, &rec);
    FILE* f = open_out(out, ".result.bin");
    std::fwrite(&rec, sizeof(rec), 1, f);
    std::fclose(f);
    std::printf("ref_driver: %dx%d, %zu components, %zu quads, status %d, %d markers\n", gray.cols, gray.rows, comps.size(), quads.size(), status,
                rec.n_markers);
    return 0;
}
