#include "ctag_io.h"

#include <cstring>
#include <fstream>

namespace ctag_host {

static uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

GrayImage read_bmp_gray(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw __FUNCTION__ + std::string(", ") + "could not open the file\n";
    std::vector<unsigned char> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') throw __FUNCTION__ + std::string(", ") + "not a BMP file\n";
    const uint32_t off = rd32(&d[10]);
    const int32_t w = (int32_t)rd32(&d[18]), hraw = (int32_t)rd32(&d[22]);
    const int bpp = rd16(&d[28]);
    const uint32_t comp = rd32(&d[30]);
    if (comp != 0 || w <= 0 || hraw == 0 || (bpp != 8 && bpp != 24 && bpp != 32)) throw __FUNCTION__ + std::string(", ") + "unsupported BMP format\n";
    const bool bottom_up = hraw > 0;
    const int h = hraw > 0 ? hraw : -hraw;
    const size_t rowbytes = (((size_t)w * bpp + 31) / 32) * 4;
    if (d.size() < off + rowbytes * (size_t)h) throw __FUNCTION__ + std::string(", ") + "truncated BMP file\n";
    unsigned char lut[256];
    if (bpp == 8) {
        const unsigned char* pal = &d[14 + rd32(&d[14])];
        uint32_t ncol = rd32(&d[46]);
        if (ncol == 0 || ncol > 256) ncol = 256;
        for (uint32_t i = 0; i < 256; i++) lut[i] = i < ncol ? bgr_to_gray(pal[4 * i], pal[4 * i + 1], pal[4 * i + 2]) : 0;
    }
    GrayImage img;
    img.rows = h;
    img.cols = w;
    img.px.resize((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const unsigned char* src = &d[off + rowbytes * (size_t)(bottom_up ? h - 1 - y : y)];
        unsigned char* dst = &img.px[(size_t)y * w];
        if (bpp == 8) {
            for (int x = 0; x < w; x++) dst[x] = lut[src[x]];
        } else {
            const int step = bpp / 8;
            for (int x = 0; x < w; x++) dst[x] = bgr_to_gray(src[x * step], src[x * step + 1], src[x * step + 2]);
        }
    }
    return img;
}

}  // namespace ctag_host

// C entry point (used by the tests and by non-C++ callers): returns 0 on success, -1 on error; copies min(cap, rows*cols)
extern "C" int ctag_host_read_bmp_gray(const char* path, int* rows, int* cols, unsigned char* dst, size_t cap) {
    try {
        const ctag_host::GrayImage g = ctag_host::read_bmp_gray(path);
        if (rows) *rows = g.rows;
        if (cols) *cols = g.cols;
        if (dst) std::memcpy(dst, g.px.data(), g.px.size() < cap ? g.px.size() : cap);
        return 0;
    } catch (const std::string&) {
        return -1;
    }
}
