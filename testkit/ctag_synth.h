// ctag_synth.h -- synthetic "random-stripe" frames: a pure function of
// (seed + frame index), rendered by identical code on the host and on the device.
//
// Strip geometry follows the reference's generator, /root/reference/CylinderTag_generator.m:221-245 (draw):
// a strip of height L holds N columns of width L/15 at pitch 1.5*L/15; every column is two black quads
// separated by a white gap of height 0.2*L whose centre sits at p_left on the column's left edge and p_right
// on its right edge, p being the root of  -p^2 + L*p + (0.11 - 0.2*cr)*L^2 = 0  for the cross ratio
// cr in {1.47, 1.54, 1.61, 1.68} of the code's left (code/8) and right (code%8) half; ids 4..7 take the
// larger root ("long" variant).  The strip is placed by a random homography (rotation + mild perspective),
// optionally wrapped on a cylinder (x -> R*sin(x/R)), and rasterised with 4x4 supersampling.
//
// 3-D scenes (BASELINE config 5, detect() + estimatePose with KNOWN poses): the same strips printed on real cylinders -- strip
// height 60 mm, a radius fixed per dictionary row -- seen by a pinhole camera; every marker has a planted rigid pose
// (R | t) and the image is ray-cast: pixel ray -> nearer intersection with the cylinder -> (arc length, height) = strip
// coordinates.  model_corners() gives the 3-D corner list of a dictionary row in the corner order detect() emits
// (cornerLists[j][0..7]: top-left, top-right, gap-top-right, gap-top-left, bottom-right, bottom-left, gap-bottom-left,
// gap-bottom-right of column featurePos[j]), i.e. the `.model` content of CylinderTag.cpp:168-188 for these objects.
#pragma once
#include <stdint.h>

#include "ctag_math.h"

namespace ctag_synth {

constexpr int kMaxMarkers = 8;
constexpr int kMaxCols = 32;

struct Marker {
    double Hinv[9];        // image (x,y,1) -> strip plane (u,v,w)
    double L, W;           // strip height and length (W = ncols * 1.5 * L/15 ... see layout)
    double cw, pitch;      // column width and pitch
    double margin;         // white paper margin around the strip
    double cylR;           // cylinder radius in strip units, 0 = flat
    float pl[kMaxCols], pr[kMaxCols];  // gap centre (v) at the left / right edge of each column
    int ncols;
    int black, paper;
    int bx0, by0, bx1, by1;  // conservative image-space bounding box of the paper
    // 3-D scenes only: camera -> object transform (x_obj = Rt * x_cam + ot) and the cylinder radius, all in strip units
    double Rt[9], ot[3];
    double radius;
};

struct Frame {
    int n;
    int bg_base;
    int ramp_x, ramp_y;   // 16.16 fixed-point gray levels per pixel
    uint64_t noise_seed;
    int mode3d;           // 0: homography scenes, 1: ray-cast cylinders seen by the pinhole camera below
    double fx, fy, cx, cy;
    Marker m[kMaxMarkers];
};

CTM_HD uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
CTM_HD uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// level (0..255) of the scene at the continuous image point (px, py); bg = background level there
CTM_HD int sample_level(const Frame& F, double px, double py, int bg) {
    for (int k = 0; k < F.n; k++) {
        const Marker& M = F.m[k];
        double u, v;
        if (F.mode3d) {
            // ray through the pixel in the object's frame; cylinder x^2 + (z - r)^2 = r^2 about the object's Y axis, strip
            // centre at the origin facing -Z
            const double dx = (px - F.cx) / F.fx, dy = (py - F.cy) / F.fy;
            const double ax = M.Rt[0] * dx + M.Rt[1] * dy + M.Rt[2], ay = M.Rt[3] * dx + M.Rt[4] * dy + M.Rt[5], az = M.Rt[6] * dx + M.Rt[7] * dy + M.Rt[8];
            const double ox = M.ot[0], oy = M.ot[1], oz = M.ot[2] - M.radius;
            const double qa = ax * ax + az * az, qb = ox * ax + oz * az, qc = ox * ox + oz * oz - M.radius * M.radius;
            const double disc = qb * qb - qa * qc;
            if (!(disc > 0) || !(qa > 1e-18)) continue;
            const double sh = (-qb - ctm::sqrt64(disc)) / qa;  // nearer intersection: the face towards the camera
            if (!(sh > 0)) continue;
            const double hx = ox + sh * ax, hz = oz + sh * az;
            u = 0.5 * M.W + M.radius * ctm::atan2_64(hx, -hz);
            v = oy + sh * ay + 0.5 * M.L;
        } else {
        const double w = M.Hinv[6] * px + M.Hinv[7] * py + M.Hinv[8];
        if (!(w > 1e-9)) continue;
        u = (M.Hinv[0] * px + M.Hinv[1] * py + M.Hinv[2]) / w;
        v = (M.Hinv[3] * px + M.Hinv[4] * py + M.Hinv[5]) / w;
        }
        if (!F.mode3d && M.cylR > 0) {
            const double t = (u - 0.5 * M.W) / M.cylR;
            if (!(t > -0.999 && t < 0.999)) continue;
            u = 0.5 * M.W + M.cylR * ctm::atan2_64(t, ctm::sqrt64(1.0 - t * t));
        }
        if (u < -M.margin || u > M.W + M.margin || v < -M.margin || v > M.L + M.margin) continue;
        int lvl = M.paper;
        if (u >= 0 && u <= M.W && v >= 0 && v <= M.L) {
            const int col = (int)(u / M.pitch);
            if (col < M.ncols) {
                const double du = u - col * M.pitch;
                if (du < M.cw) {
                    const double t = du / M.cw;
                    const double gc = M.pl[col] + (M.pr[col] - M.pl[col]) * t;
                    if (v < gc - 0.1 * M.L || v > gc + 0.1 * M.L) lvl = M.black;
                }
            }
        }
        return lvl;
    }
    return bg;
}

CTM_HD uint8_t pixel(const Frame& F, int x, int y, int rows, int cols) {
    int bg = F.bg_base + (int)(((long long)F.ramp_x * (x - cols / 2) + (long long)F.ramp_y * (y - rows / 2)) >> 16);
    bg = bg < 0 ? 0 : (bg > 255 ? 255 : bg);
    bool near = false;
    for (int k = 0; k < F.n; k++) {
        const Marker& M = F.m[k];
        near = near || (x >= M.bx0 && x <= M.bx1 && y >= M.by0 && y <= M.by1);
    }
    int val = bg;
    if (near) {
        int sum = 0;
        for (int sy = 0; sy < 4; sy++)
            for (int sx = 0; sx < 4; sx++) sum += sample_level(F, x + (sx + 0.5) * 0.25, y + (sy + 0.5) * 0.25, bg);
        val = (sum + 8) >> 4;
    }
    const uint64_t hsh = mix64(F.noise_seed ^ ((uint64_t)y * 0x100000001B3ULL + (uint64_t)x));
    val += (int)(hsh % 7) - 3;
    return (uint8_t)(val < 0 ? 0 : (val > 255 ? 255 : val));
}

// ---- host-side layout (deterministic; double arithmetic + ctm functions only) --------------------------
static inline double urand(uint64_t& s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }

static inline bool invert3(const double* a, double* o) {
    const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    if (det == 0) return false;
    const double id = 1.0 / det;
    o[0] = c00 * id;
    o[1] = (a[2] * a[7] - a[1] * a[8]) * id;
    o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c01 * id;
    o[4] = (a[0] * a[8] - a[2] * a[6]) * id;
    o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c02 * id;
    o[7] = (a[1] * a[6] - a[0] * a[7]) * id;
    o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return true;
}
static inline void mul3(const double* a, const double* b, double* o) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

struct Truth {
    int n;
    int dict_row[kMaxMarkers];
    float strip_len[kMaxMarkers];
    float corners[kMaxMarkers][8];
};

// gap centre for one half-code id 0..7 (CylinderTag_generator.m:223-243)
static inline double gap_centre(int id, double L) {
    static const double cr[8] = {1.47, 1.54, 1.61, 1.68, 1.68, 1.61, 1.54, 1.47};
    const bool use_max = id >= 4;
    const double k = 0.11 - 0.2 * cr[id];  // p^2 - L p - k L^2 = 0
    const double disc = ctm::sqrt64(1.0 + 4.0 * k);
    const double r1 = 0.5 * L * (1.0 - disc), r2 = 0.5 * L * (1.0 + disc);
    return use_max ? r2 : r1;
}

static inline void layout(const int32_t* state, int drows, int dcols, uint64_t seed, int frame_index, int rows, int cols,
                          int markers, Frame* F, Truth* T) {
    uint64_t s = seed + (uint64_t)frame_index;
    (void)splitmix64(s);
    F->bg_base = 150 + (int)(urand(s) * 71.0);
    const double rx = (urand(s) * 2 - 1) * 20.0 / (cols * 0.5), ry = (urand(s) * 2 - 1) * 20.0 / (rows * 0.5);
    F->ramp_x = (int)(rx * 0.5 * 65536.0);
    F->ramp_y = (int)(ry * 0.5 * 65536.0);
    F->noise_seed = splitmix64(s);
    markers = markers < 0 ? 0 : (markers > kMaxMarkers ? kMaxMarkers : markers);
    const int ncols = dcols > kMaxCols ? kMaxCols : dcols;
    const int gx = markers <= 1 ? 1 : (markers <= 4 ? 2 : 4), gy = markers <= 2 ? 1 : 2;
    const double cellw = (double)cols / gx, cellh = (double)rows / gy;
    const double sizek = cols / 1920.0;
    F->n = 0;
    F->mode3d = 0;
    F->fx = F->fy = 1.0;
    F->cx = F->cy = 0.0;
    if (T) T->n = 0;
    for (int k = 0; k < markers; k++) {
        Marker& M = F->m[F->n];
        const int cell = k % (gx * gy);
        const double ox = (cell % gx) * cellw, oy = (cell / gx) * cellh;
        const int row = (int)(urand(s) * drows) % drows;
        const double theta = urand(s) * 6.283185307179586;
        double L = (220.0 + urand(s) * 200.0) * sizek;
        const double p1 = (urand(s) * 2 - 1) * 0.12, p2 = (urand(s) * 2 - 1) * 0.12;
        const bool cyl = urand(s) < 0.5;
        const double cylk = 0.8 + urand(s) * 2.2;
        const double jx = urand(s), jy = urand(s);
        M.black = 15 + (int)(urand(s) * 21.0);
        M.paper = 225 + (int)(urand(s) * 21.0);
        const double ct = ctm::cos64(theta), st = ctm::sin64(theta);
        const double act = ct < 0 ? -ct : ct, ast = st < 0 ? -st : st;
        const double strip_w_rel = ncols * 0.1;  // W / L  (12 columns -> 1.2)
        const double mgk = 0.08;
        const double wrel = strip_w_rel + 2 * mgk, hrel = 1.0 + 2 * mgk;
        const double bw = (wrel * act + hrel * ast) * 1.12, bh = (wrel * ast + hrel * act) * 1.12;  // bbox per unit L (+ perspective slack)
        const double pad = 14.0;
        const double Lfit = ((cellw - 2 * pad) / bw) < ((cellh - 2 * pad) / bh) ? ((cellw - 2 * pad) / bw) : ((cellh - 2 * pad) / bh);
        if (L > Lfit) L = Lfit;
        if (L < 60.0) continue;
        M.L = L;
        M.W = strip_w_rel * L;
        M.cw = L / 15.0;
        M.pitch = 1.5 * L / 15.0;
        M.margin = mgk * L;
        M.cylR = cyl ? cylk * M.W : 0.0;
        M.ncols = ncols;
        for (int c = 0; c < ncols; c++) {
            const int code = state[row * dcols + c];
            M.pl[c] = (float)gap_centre(code / 8, L);
            M.pr[c] = (float)gap_centre(code % 8, L);
        }
        const double hw = 0.5 * bw * L, hh = 0.5 * bh * L;
        const double cx = ox + pad + hw + jx * (cellw - 2 * pad - 2 * hw), cy = oy + pad + hh + jy * (cellh - 2 * pad - 2 * hh);
        // H = T(cx,cy) * R(theta) * P(p1/L, p2/L) * T(-W/2, -L/2)
        const double Tm[9] = {1, 0, -0.5 * M.W, 0, 1, -0.5 * M.L, 0, 0, 1};
        const double Pm[9] = {1, 0, 0, 0, 1, 0, p1 / L, p2 / L, 1};
        const double Rm[9] = {ct, -st, cx, st, ct, cy, 0, 0, 1};
        double A[9], H[9];
        mul3(Pm, Tm, A);
        mul3(Rm, A, H);
        if (!invert3(H, M.Hinv)) continue;
        // bounding box of the paper rectangle in the image
        double minx = 1e30, miny = 1e30, maxx = -1e30, maxy = -1e30;
        const double us[2] = {-M.margin, M.W + M.margin}, vs[2] = {-M.margin, M.L + M.margin};
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) {
                const double w = H[6] * us[a] + H[7] * vs[b] + H[8];
                const double x = (H[0] * us[a] + H[1] * vs[b] + H[2]) / w, y = (H[3] * us[a] + H[4] * vs[b] + H[5]) / w;
                minx = x < minx ? x : minx;
                maxx = x > maxx ? x : maxx;
                miny = y < miny ? y : miny;
                maxy = y > maxy ? y : maxy;
            }
        M.bx0 = (int)minx - 2;
        M.by0 = (int)miny - 2;
        M.bx1 = (int)maxx + 2;
        M.by1 = (int)maxy + 2;
        if (T) {
            T->dict_row[T->n] = row;
            T->strip_len[T->n] = (float)L;
            const double cu[4] = {0, M.W, M.W, 0}, cv[4] = {0, 0, M.L, M.L};
            for (int q = 0; q < 4; q++) {
                const double w = H[6] * cu[q] + H[7] * cv[q] + H[8];
                T->corners[T->n][2 * q] = (float)((H[0] * cu[q] + H[1] * cv[q] + H[2]) / w);
                T->corners[T->n][2 * q + 1] = (float)((H[3] * cu[q] + H[4] * cv[q] + H[5]) / w);
            }
            T->n++;
        }
        F->n++;
    }
}

// =====================================================================================================
// 3-D scenes: printed strips on cylinders, pinhole camera, planted poses
// =====================================================================================================
constexpr double kStripHeightMM = 60.0;  // physical strip height of every synthetic object

// radius (mm) of the cylinder that carries dictionary row `row`: fixed per row, like a real set of objects
static inline double row_radius_mm(int row, int ncols) {
    const double W = ncols * 0.1 * kStripHeightMM;
    return W * (0.45 + 0.08 * (double)((row * 7 + 3) % 11));  // 0.45 W .. 1.25 W: the strip spans 0.8 .. 2.2 rad of arc
}

// 3-D corners of dictionary row `row` in the object frame (mm), [ncols * 8][3], in detect()'s corner order (see the header)
static inline void model_corners(const int32_t* state, int dcols, int row, float* out) {
    const int ncols = dcols > kMaxCols ? kMaxCols : dcols;
    const double L = kStripHeightMM, W = ncols * 0.1 * L, cw = L / 15.0, pitch = 1.5 * L / 15.0, r = row_radius_mm(row, ncols);
    for (int c = 0; c < ncols; c++) {
        const int code = state[row * dcols + c];
        const double gl = gap_centre(code / 8, L), gr = gap_centre(code % 8, L), u0 = c * pitch, u1 = u0 + cw;
        const double uv[8][2] = {{u0, 0}, {u1, 0}, {u1, gr - 0.1 * L}, {u0, gl - 0.1 * L}, {u1, L}, {u0, L}, {u0, gl + 0.1 * L}, {u1, gr + 0.1 * L}};
        for (int k = 0; k < 8; k++) {
            const double th = (uv[k][0] - 0.5 * W) / r;
            float* o = out + ((size_t)c * 8 + k) * 3;
            o[0] = (float)(r * ctm::sin64(th));
            o[1] = (float)(uv[k][1] - 0.5 * L);
            o[2] = (float)(r - r * ctm::cos64(th));
        }
    }
}

struct Truth3D {
    int n;
    int dict_row[kMaxMarkers];
    double R[kMaxMarkers][9];  // object -> camera
    double t[kMaxMarkers][3];
    double radius[kMaxMarkers];
};

// frame `frame_index` of a 3-D scene: `markers` cylinders in front of the camera (fx, fy, cx, cy), each in its own cell of the
// image so that they do not overlap, strip height on screen 440..840 px at 3840 columns (scaled with the frame width),
// tilted by at most ~30 degrees out of the image plane, rolled freely about the optical axis
static inline void layout3d(const int32_t* state, int drows, int dcols, uint64_t seed, int frame_index, int rows, int cols, int markers,
                            double fx, double fy, double cx, double cy, Frame* F, Truth3D* T) {
    uint64_t s = (seed ^ 0x3D3D3D3D3D3D3D3DULL) + (uint64_t)frame_index;
    (void)splitmix64(s);
    F->bg_base = 150 + (int)(urand(s) * 71.0);
    const double rx = (urand(s) * 2 - 1) * 20.0 / (cols * 0.5), ry = (urand(s) * 2 - 1) * 20.0 / (rows * 0.5);
    F->ramp_x = (int)(rx * 0.5 * 65536.0);
    F->ramp_y = (int)(ry * 0.5 * 65536.0);
    F->noise_seed = splitmix64(s);
    F->mode3d = 1;
    F->fx = fx;
    F->fy = fy;
    F->cx = cx;
    F->cy = cy;
    F->n = 0;
    if (T) T->n = 0;
    markers = markers < 0 ? 0 : (markers > kMaxMarkers ? kMaxMarkers : markers);
    const int ncols = dcols > kMaxCols ? kMaxCols : dcols;
    const int gx = markers <= 1 ? 1 : (markers <= 4 ? 2 : 4), gy = markers <= 2 ? 1 : 2;
    const double cellw = (double)cols / gx, cellh = (double)rows / gy;
    const double sizek = cols / 3840.0;
    for (int k = 0; k < markers; k++) {
        Marker& M = F->m[F->n];
        const int cell = k % (gx * gy);
        const double ox = (cell % gx) * cellw, oy = (cell / gx) * cellh;
        const int row = (int)(urand(s) * drows) % drows;
        const double roll = urand(s) * 6.283185307179586;
        const double tilt_x = (urand(s) * 2 - 1) * 0.5, tilt_y = (urand(s) * 2 - 1) * 0.5;  // radians about the image x / y axes
        double Lpx = (440.0 + urand(s) * 400.0) * sizek;
        const double jx = urand(s), jy = urand(s);
        M.black = 15 + (int)(urand(s) * 21.0);
        M.paper = 225 + (int)(urand(s) * 21.0);
        M.L = kStripHeightMM;
        M.W = ncols * 0.1 * M.L;
        M.cw = M.L / 15.0;
        M.pitch = 1.5 * M.L / 15.0;
        M.margin = 0.08 * M.L;
        M.cylR = 0.0;
        M.radius = row_radius_mm(row, ncols);
        M.ncols = ncols;
        for (int c = 0; c < ncols; c++) {
            const int code = state[row * dcols + c];
            M.pl[c] = (float)gap_centre(code / 8, M.L);
            M.pr[c] = (float)gap_centre(code % 8, M.L);
        }
        // the chord of the visible strip is shorter than its arc; bound the footprint by the flat strip
        const double wrel = M.W / M.L + 0.16, hrel = 1.16;
        const double diag = ctm::sqrt64(wrel * wrel + hrel * hrel) * 1.1;
        const double pad = 14.0;
        const double fit = ((cellw < cellh ? cellw : cellh) - 2 * pad) / diag;
        if (Lpx > fit) Lpx = fit;
        if (Lpx < 100.0 * sizek) continue;
        // R = Rz(roll) * Rx(tilt_x) * Ry(tilt_y)
        const double cz = ctm::cos64(roll), sz = ctm::sin64(roll), cxr = ctm::cos64(tilt_x), sxr = ctm::sin64(tilt_x), cyr = ctm::cos64(tilt_y), syr = ctm::sin64(tilt_y);
        const double Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1}, Rx[9] = {1, 0, 0, 0, cxr, -sxr, 0, sxr, cxr}, Ry[9] = {cyr, 0, syr, 0, 1, 0, -syr, 0, cyr};
        double A[9], R[9];
        mul3(Rx, Ry, A);
        mul3(Rz, A, R);
        const double depth = fy * M.L / Lpx;  // mm: the strip is Lpx pixels tall when it faces the camera
        const double half = 0.5 * diag * Lpx;
        const double pcx = ox + pad + half + jx * (cellw - 2 * pad - 2 * half), pcy = oy + pad + half + jy * (cellh - 2 * pad - 2 * half);
        const double t[3] = {(pcx - cx) / fx * depth, (pcy - cy) / fy * depth, depth};
        // camera -> object: x_obj = R^T (x_cam - t)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) M.Rt[i * 3 + j] = R[j * 3 + i];
        for (int i = 0; i < 3; i++) M.ot[i] = -(M.Rt[i * 3] * t[0] + M.Rt[i * 3 + 1] * t[1] + M.Rt[i * 3 + 2] * t[2]);
        for (int i = 0; i < 9; i++) M.Hinv[i] = 0.0;
        // image bounding box of the paper: project a grid of its surface points
        double minx = 1e30, miny = 1e30, maxx = -1e30, maxy = -1e30;
        for (int a = 0; a <= 16; a++)
            for (int b = 0; b <= 2; b++) {
                const double u = -M.margin + (M.W + 2 * M.margin) * a / 16.0, v = -M.margin + (M.L + 2 * M.margin) * b / 2.0;
                const double th = (u - 0.5 * M.W) / M.radius;
                const double X[3] = {M.radius * ctm::sin64(th), v - 0.5 * M.L, M.radius - M.radius * ctm::cos64(th)};
                const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1],
                             zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
                if (!(zc > 1.0)) continue;
                const double x = fx * xc / zc + cx, y = fy * yc / zc + cy;
                minx = x < minx ? x : minx;
                maxx = x > maxx ? x : maxx;
                miny = y < miny ? y : miny;
                maxy = y > maxy ? y : maxy;
            }
        M.bx0 = (int)minx - 3;
        M.by0 = (int)miny - 3;
        M.bx1 = (int)maxx + 3;
        M.by1 = (int)maxy + 3;
        if (T) {
            T->dict_row[T->n] = row;
            for (int i = 0; i < 9; i++) T->R[T->n][i] = R[i];
            for (int i = 0; i < 3; i++) T->t[T->n][i] = t[i];
            T->radius[T->n] = M.radius;
            T->n++;
        }
        F->n++;
    }
}

}  // namespace ctag_synth
