// ctag_io.h -- minimal frame ingest for the host layer: an uncompressed BMP reader and the
// BGR -> gray conversion the reference's demo applies before detect() (main.cpp:29,36: imread + cvtColor(BGR2GRAY)).
// OpenCV's 8-bit BGR2GRAY is the fixed-point  (B*1868 + G*9617 + R*4899 + 8192) >> 14  [OCV-recall of color_rgb.simd.hpp].
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ctag_host {

struct GrayImage {
    int rows = 0, cols = 0;
    std::vector<unsigned char> px;  // rows x cols, top-down
};

// 8-bit palettised, 24-bit or 32-bit uncompressed BMP -> gray.  Throws std::string like the reference's loaders.
GrayImage read_bmp_gray(const std::string& path);

inline unsigned char bgr_to_gray(unsigned b, unsigned g, unsigned r) { return (unsigned char)((b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14); }

}  // namespace ctag_host
