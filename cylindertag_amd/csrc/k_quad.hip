// k_quad.hip -- K6: per candidate component: silhouette boundary, ordered traversal, extended Ramer-Douglas-Peucker
// split into <= 4 edges (k_pack + k_quad_edges_packed: 8 components per wave, or a wave each for long boundaries), robust
// (Welsch) line fits (k_line_sort + k_welsch; k_welsch_lat in few-frame calls), quad selection (k_quad_final).
// GPU counterpart of corner_detector::edgeExtraction and helpers
//   /root/reference/corner_detector.cpp:125-169 (expand_line), :171-405 (edgeExtraction),
//   :407-418 (get_orientedEdgePoints), :420-452 (get_permutation), :454-463 (quadJudgment)
// and of the OpenCV fitLine(DIST_L2 / DIST_WELSCH) calls inside them.
//
// Design (not a translation): the reference builds a bbox mask + visited image and recurses; here the
// silhouette is kept as four sparse first-hit arrays (top/bottom per column, left/right per row), the
// recursion is an explicit stack that is never unwound past the last silhouette pixel, expand_line's per-point refits use
// exact integer moment sums and are speculated 8 (or 64) steps at a time, the 20 Welsch restarts of an edge run on lanes at
// once (the cv::RNG pick sequence depends only on the point count, so it is replayed up front), and every floating-point sum
// that the reference accumulates sequentially is accumulated in the same order so results are bit-identical to the oracle.
#include "ctag_internal.h"
#include "ctag_math.h"
#include <cstdio>
#include <type_traits>
#include <cstdlib>

namespace ctag {

struct QuadPtrs {
    const uint16_t* labels;
    const int32_t* tile_base;
    const int32_t* root_of;
    const int32_t* ncand;
    const Candidate* cand;
    QuadOut* quads;
    uint32_t* frame_flags;
    // edge clusters handed from k_quad_edges to k_welsch / k_quad_final
    int32_t* line_count;   // [F]
    int32_t* clp_used;     // [F]
    uint32_t* cl_pool;     // [F][cl_cap] packed (x | y << 16) points, in the order the reference pushes them
    LineDesc* line_desc;   // [F][line_cap]
    int32_t* line_sorted;  // [F][line_cap] line ids by descending point count
    int32_t* line_long;    // [F] edges of more than kWShort points = the first ranks of line_sorted
    float* line_fit;       // [F][line_cap][4]
    CandAux* cand_aux;     // [F][cand_cap]
    const uint8_t* pick_table;  // [kPickN][20][10] cv::RNG initial samples of fitLine2D for every point count < kPickN
    const uint16_t* pick_table16;  // [kPickN2 - kPickN][20][10] the same for point counts in [kPickN, kPickN2)
    const int32_t* pool_tile;    // [F][pool_cap]
    const int32_t* member_head;  // [F][pool_cap]
    const int32_t* member_next;  // [F][pool_cap]
    int32_t* npacks;       // [F][2] packs, oversize components
    uint32_t* packs;       // [F][cand_cap] first entry of pack_order | count << 24, longest first
    uint32_t* pack_order;  // [F][cand_cap] candidate indices by descending boundary capacity
    unsigned long long* stamps;  // developer aid (CTAG_QUAD_STAMPS=1): cycles per phase of k_quad_edges, else null
    // tunables (include/ctag.h: ctag_params; the reference's values in brackets)
    float thr_line, thr_expand, rac;  // threshold_line [1.8], threshold_expand [1.2], threshold_RAC [0.3]
    int c2_far, c2_near;              // collinearity cost [1.05] as bounds on the squared integer norm: [2], [1]
    // pool sizes of this workspace (ctag_internal.h: Workspace::cand_cap / line_cap / cl_cap)
    int cand_cap, line_cap;
    uint32_t cl_cap;
    float expand_eps;                 // half-width of expand_line's filter band in units of the frame extent [3e-6]; +inf: always the exact fits (builds that read P)
    const uint64_t* mask;             // the fused sweep's threshold mask (1 bit per half-size pixel, rows of mask_words 64-bit words), else null: k_silhouette_mask
    int mask_words;
};

__device__ __forceinline__ uint32_t pack_xy(int x, int y) { return (uint32_t)x | ((uint32_t)y << 16); }
__device__ __forceinline__ int ux(uint32_t p) { return (int)(p & 0xffffu); }
__device__ __forceinline__ int uy(uint32_t p) { return (int)(p >> 16); }

// fitLine2D_wods tail: moments -> (vx, vy, x0, y0)   [SURVEY App. A.5]
__device__ __forceinline__ void moments_to_line(double x, double y, double x2, double y2, double xy, double w, float* line) {
    const ctm::Recip64 W = ctm::recip64(w);  // five IEEE quotients by one denominator: its reciprocal refined once (ctag_math.h)
    x = ctm::div64(x, W);
    y = ctm::div64(y, W);
    x2 = ctm::div64(x2, W);
    y2 = ctm::div64(y2, W);
    xy = ctm::div64(xy, W);
    const double dx2 = x2 - x * x, dy2 = y2 - y * y, dxy = xy - x * y;
    const float t = (float)ctm::atan2_64(2 * dxy, dx2 - dy2) / 2;
    double sn, cs;
    ctm::sincos64(t, &sn, &cs);  // == sin64(t), cos64(t): one reduction, no quadrant divergence
    line[0] = (float)cs;
    line[1] = (float)sn;
    line[2] = (float)x;
    line[3] = (float)y;
}

// determinant + solve for 2x2 CV_32F [SURVEY App. A.7]
__device__ __forceinline__ bool solve2x2(float a00, float a01, float a10, float a11, float b0, float b1, float& x0, float& x1) {
    double d = (double)a00 * a11 - (double)a01 * a10;
    if (d == 0.) return false;
    d = 1. / d;
    const float t = (float)(((double)b0 * a11 - (double)b1 * a01) * d);
    x1 = (float)(((double)b1 * a00 - (double)b0 * a10) * d);
    x0 = t;
    return true;
}

// cv::RNG (multiply-with-carry) as fitLine2D seeds it: RNG rng((uint64)-1)
struct CvRng {
    uint64_t state;
    __device__ unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
};

// One Welsch restart (the body of fitLine2D's `for k` loop) on one lane.  pts: cluster points in the order
// the reference pushed them.  picks: the restart's initial sample (ascending).  Returns (_line, err) as they
// stand when the reference leaves the inner loop.
#ifndef CTAG_WCAP
#define CTAG_WCAP 16
#endif
#ifndef CTAG_WUNROLL
#define CTAG_WUNROLL 2
#endif
#define CTAG_PRAGMA_(x) _Pragma(#x)
#define CTAG_PRAGMA(x) CTAG_PRAGMA_(x)
constexpr int kWCap = CTAG_WCAP;  // Welsch weights of the first kWCap points are kept in LDS between the two passes
// Point sources of a restart: the cluster in global memory as packed (x | y << 16) words, or -- for a wave's three edges of at most
// kWPts points -- the same points converted to float pairs once per block in LDS (the 20 restarts of an edge read the same
// address: a broadcast, and the two integer unpacks, two converts and the scattered global load per point and pass are gone).
struct GlobalPts {
    const uint32_t* p;
    __device__ __forceinline__ float2 operator()(int j) const {
        const uint32_t v = p[j];
        return make_float2((float)ux(v), (float)uy(v));
    }
};
struct LdsPts {
    const float2* p;
    __device__ __forceinline__ float2 operator()(int j) const { return p[j]; }
};
struct LdsPackedPts {  // the packed words themselves in LDS: half the bytes of float pairs (longer edges fit), the unpacking of GlobalPts, no trip through the texture path
    const uint32_t* p;
    __device__ __forceinline__ float2 operator()(int j) const {
        const uint32_t v = p[j];
        return make_float2((float)ux(v), (float)uy(v));
    }
};
// Initial samples: the precomputed cv::RNG table (bytes, point counts below kPickN), a replayed list (16-bit), or -- short edges --
// every point of the edge.
struct TablePicks {
    const uint8_t* t;
    __device__ __forceinline__ int operator()(int i) const { return t[i]; }
};
struct ListPicks {
    const uint16_t* t;
    __device__ __forceinline__ int operator()(int i) const { return t[i]; }
};
struct AllPicks {
    __device__ __forceinline__ int operator()(int i) const { return i; }
};
// wc: this lane's column of a [kWCap][64] float array in LDS (element j at wc[j * 64]); nullptr = always recompute.
// The restart's results go to res[q * rstride]: the line in q = 0..3, the two halves of the error (a double) in q = 4, 5 -- with
// res = wc and rstride = 64 they land in the lane's own weight column, which is dead by then (needs kWCap >= 6).
struct RestartResult {
    float line[4];
    double err;
};
__device__ __forceinline__ RestartResult restart_result(const float* res, int rstride) {
    RestartResult r;
    for (int q = 0; q < 4; q++) r.line[q] = res[q * rstride];
    const uint64_t lo = ctm::f32_to_bits(res[4 * rstride]), hi = ctm::f32_to_bits(res[5 * rstride]);
    r.err = ctm::bits_to_f64(lo | (hi << 32));
    return r;
}
// the initial fit of a restart: fitLine2D's weights are 1 on the restart's sample and 0 elsewhere (zero-weight points add +0.0: skipping them is exact)
template <class Pts, class Picks>
__device__ __forceinline__ void welsch_first_fit(const Pts pts, const Picks picks, int npick, float* line) {
    double x = 0, y = 0, x2 = 0, y2 = 0, xy = 0, w = 0;
    for (int i = 0; i < npick; i++) {
        const float2 p = pts(picks(i));
        const float px = p.x, py = p.y;
        x += px;
        y += py;
        x2 += px * px;
        y2 += py * py;
        xy += px * py;
        w += 1.f;
    }
    moments_to_line(x, y, x2, y2, xy, w, line);
}
// IRLS iterations it0, it0 + 1, ... of one restart from `line` (the body of fitLine2D's inner loop).  Returns true when the restart has ENDED -- converged,
// err < EPS, or 30 iterations -- with (line, err) as the reference leaves them; returns false at iteration it_stop, AFTER that iteration's convergence test
// has failed: the caller goes on later (on any lane) with welsch_rounds(.., it0 = it_stop, resume = true), which skips the test it already knows the
// outcome of -- so neither the previous line nor err has to travel.  it_stop = 30: to the end.
// wc: this lane's column of a [rows][WS] float array in LDS (element j at wc[j * WS]) for the weights of the first `ncache` points between the two passes.
template <int WS, class Pts>
__device__ __forceinline__ bool welsch_rounds(const Pts pts, int n, double EPS, float* line, double& err, int it0, int it_stop, bool resume, float* wc, int ncache) {
    float prev[4] = {0.f, 0.f, 0.f, 0.f};
    const float c = 1 / 2.9846f;
#ifdef CTAG_WELSCH_MINERR_IN_LOOP
    // The other placement of fitLine2D's `if (err < min_err)` (oracle: OracleVariants::welsch_minerr_in_loop): the pair is taken right after calcDist2D, so a restart
    // hands on the FIRST SMALLEST (err, line) of its iterations instead of (last err, last refit).  The restart ends early on an error below EPS only when that error
    // improved on its own best; what earlier restarts had reached cannot matter unless two restarts reach different errors below EPS (exactly collinear points give 0).
    // A restart's best is part of its state here, so this build does not regroup (kWRegroupAt = 30).
    double berr = 1.7976931348623157e308;
    float bl[4] = {0.f, 0.f, 0.f, 0.f};
#define CTAG_WELSCH_HAND_ON()       \
    do {                            \
        for (int q = 0; q < 4; q++) line[q] = bl[q]; \
        err = berr;                 \
    } while (0)
#else
#define CTAG_WELSCH_HAND_ON() do { } while (0)
#endif
    for (int it = it0; it < 30; it++) {
        if (it > 0 && !(resume && it == it0)) {
            // reference: fabs(acos(clamp(t))) < 0.01f with t a float-valued dot product.  acos64 is decreasing, so the
            // test is a threshold on t: kAcosBelowTenMilli is the smallest float with acos64(t) < 0.01f (checked over every
            // float by tests/test_oracle_cpu.py), which keeps ~100 FP64 operations out of every iteration.
            const float t = line[0] * prev[0] + line[1] * prev[1];
            if (t >= ctm::kAcosBelowTenMilli) {
                const float dx = ctm::fabs32(line[2] - prev[2]);
                const float dy = ctm::fabs32(line[3] - prev[3]);
                const float d = dx > dy ? dx : dy;
                if (d < 0.01f) {
                    CTAG_WELSCH_HAND_ON();
                    return true;
                }
            }
        }
        if (it == it_stop) return false;
        const float lx = line[2], ly = line[3], nx = line[1], ny = -line[0];
        double sum_w = 0;
        err = 0;
        // (the point after the last is requested too and never used: the staged rows and the cluster pool are padded by one element, which
        // spares every trip the clamp of its index -- v_min_i32, sign extension and a 64-bit address per point and pass)
        float2 pn0 = pts(min(0, n - 1));  // next point, loaded one trip ahead
        CTAG_PRAGMA(unroll CTAG_WUNROLL)
        for (int j = 0; j < ncache; j++) {
            const float2 p = pn0;
            pn0 = pts(j + 1);
            const float x = p.x - lx, y = p.y - ly;
            const float r = ctm::fabs32(nx * x + ny * y);
            err += r;
            const float wj = ctm::exp32_nonpos(-r * r * c * c);
            wc[j * WS] = wj;
            sum_w += wj;
        }
        float2 pn1 = pts(min(ncache, n - 1));  // next point, loaded one trip ahead
        CTAG_PRAGMA(unroll CTAG_WUNROLL)
        for (int j = ncache; j < n; j++) {
            const float2 p = pn1;
            pn1 = pts(j + 1);
            const float x = p.x - lx, y = p.y - ly;
            const float r = ctm::fabs32(nx * x + ny * y);
            err += r;
            sum_w += ctm::exp32_nonpos(-r * r * c * c);
        }
#ifdef CTAG_WELSCH_MINERR_IN_LOOP
        if (err < berr) {
            berr = err;
            for (int q = 0; q < 4; q++) bl[q] = line[q];
            if (err < EPS) return true;  // (line, err) are the best already
        }
#else
        if (err < EPS) return true;
#endif
        double x = 0, y = 0, x2 = 0, y2 = 0, xy = 0, w = 0;
        if (ctm::fabs64(sum_w) > 1.1920928955078125e-07) {
            const double inv = 1. / sum_w;
            float2 pn2 = pts(min(0, n - 1));  // next point, loaded one trip ahead
            CTAG_PRAGMA(unroll CTAG_WUNROLL)
            for (int j = 0; j < ncache; j++) {
                const float2 p = pn2;
                pn2 = pts(j + 1);
                const float px = p.x, py = p.y;
                const float wj = (float)(wc[j * WS] * inv);
                x += wj * px;
                y += wj * py;
                x2 += wj * px * px;
                y2 += wj * py * py;
                xy += wj * px * py;
                w += wj;
            }
            float2 pn3 = pts(min(ncache, n - 1));  // next point, loaded one trip ahead
            CTAG_PRAGMA(unroll CTAG_WUNROLL)
            for (int j = ncache; j < n; j++) {
                const float2 p = pn3;
                pn3 = pts(j + 1);
                const float px = p.x, py = p.y;
                const float r = ctm::fabs32(nx * (px - lx) + ny * (py - ly));
                const float wj = (float)(ctm::exp32_nonpos(-r * r * c * c) * inv);
                x += wj * px;
                y += wj * py;
                x2 += wj * px * px;
                y2 += wj * py * py;
                xy += wj * px * py;
                w += wj;
            }
        } else {
            float2 pn4 = pts(min(0, n - 1));  // next point, loaded one trip ahead
            for (int j = 0; j < n; j++) {
                const float2 p = pn4;
                pn4 = pts(j + 1);
                const float px = p.x, py = p.y;
                x += px;
                y += py;
                x2 += px * px;
                y2 += py * py;
                xy += px * py;
                w += 1.f;
            }
        }
        prev[0] = line[0];
        prev[1] = line[1];
        prev[2] = line[2];
        prev[3] = line[3];
        moments_to_line(x, y, x2, y2, xy, w, line);
    }
    CTAG_WELSCH_HAND_ON();
    return true;
}
#undef CTAG_WELSCH_HAND_ON
// a restart's result where the selection reads it: the line in res[q * rstride], q = 0..3, the two halves of the error (a double) in q = 4, 5
__device__ __forceinline__ void restart_store(float* res, int rstride, const float* line, double err) {
    for (int q = 0; q < 4; q++) res[q * rstride] = line[q];
    const uint64_t eb = ctm::f64_to_bits(err);
    res[4 * rstride] = ctm::bits_to_f32((uint32_t)eb);
    res[5 * rstride] = ctm::bits_to_f32((uint32_t)(eb >> 32));
}
// One whole Welsch restart (the body of fitLine2D's `for k` loop) on one lane; with res = wc and rstride = WS the result lands in the lane's own
// weight column, which is dead by then (needs kWCap >= 6).
template <int WS, class Pts, class Picks>
__device__ __forceinline__ void welsch_restart(const Pts pts, int n, const Picks picks, int npick, double EPS, float* res, int rstride, float* wc) {
    float line[4];
    double err = 0;
    welsch_first_fit(pts, picks, npick, line);
    (void)welsch_rounds<WS>(pts, n, EPS, line, err, 0, 30, false, wc, min(n, kWCap));
    restart_store(res, rstride, line, err);
}

// ---- sub-wave packing: 8 components per wave, 8 lanes each (k_quad_edges_packed) -------------------------
#ifndef CTAG_PACK_WORDS
#define CTAG_PACK_WORDS 5120
#endif
#ifndef CTAG_PACK_WAVES
#define CTAG_PACK_WAVES 2
#endif
#ifndef CTAG_SCAN_ROWS4
#define CTAG_SCAN_ROWS4 0  // 1: the boundary-only build keeps four label rows in flight per lane (measured: 44 spilled registers, 4.03 vs 3.94 ms)
#endif
#ifndef CTAG_PACK_SPLIT_LARGE
#define CTAG_PACK_SPLIT_LARGE 0  // the same for the large configuration (4K frames)
#endif
#ifndef CTAG_PACK_SPLIT
#define CTAG_PACK_SPLIT 1  // the small-configuration packed build as two kernels (boundary, then edge clusters): see k_quad_edges_packed
#endif
constexpr int kPackWords = CTAG_PACK_WORDS;  // LDS words shared by the up to 8 components of a wave (large configuration)
// Two builds of the packed kernel.  The kernel is bound by dependent LDS round trips (24 % of the issue roof at 2 waves per
// SIMD), so for frames of 1080p class -- boundaries of ~150 points, a pack of 5 components in 10 KB -- the small configuration
// runs 4 waves per SIMD (128 VGPRs, 61 of them spilled to scratch: still 6.8 -> 6.2 ms per 4096 frames); 4K frames have
// components four times the size and keep the large one (same-box A/B: 8.4 vs 7.9 ms per 1024 frames with the small one).
#ifndef CTAG_PACK_WAVES_P2
#define CTAG_PACK_WAVES_P2 4
#endif
constexpr int kPackWordsSmall = 2560, kPackWavesSmall = 4, kPackWavesSmallP2 = CTAG_PACK_WAVES_P2;  // (the second kernel's own waves per SIMD: see launch_quads)
constexpr int kSG = 8;
constexpr int kUnwind = 8;  // stack frames the whole-wave build tests at once when it unwinds (<= 8: 8 lanes each)
constexpr int kWaveWords = 8192;      // LDS words of the common whole-wave build (32 KB: five components per CU)
constexpr int kWaveWordsMax = 36864;  // ... of the build for the longest boundaries (144 KB: one per CU)
constexpr int kLatencyBigPoints = 1;  // calls of <= kLatencyFrames frames: components with a boundary capacity above this get a wave of their own -- all of them (launch_quads)
__host__ __device__ __forceinline__ int pack_points(int w, int h) { return min(2 * (w + h), w * h) + 1; }
// LDS words one component needs in the packed kernel: silhouette arrays + boundary list + stack / ping-pong list
__host__ __device__ __forceinline__ int pack_need(int w, int h) {
    const int C = pack_points(w, h);
    return ((w + 1) & ~1) + 2 * h + 2 * C + 4;
}
// the packed kernel takes every component whose working set fits the wave's LDS budget (the silhouette scan walks a wide
// box in passes of 128 columns); the rest are "big" and get a wave of their own (k_quad_edges_packed<64, ...>)
// big_points: in latency mode (a few frames per call) components with a long boundary also go there -- a whole wave per
// component instead of 8 lanes shortens the critical path of the call; 0x7fffffff otherwise
// scan_words > 0: the silhouettes of the packed components come from the scan-only kernel (PHASE 3 below), which holds w + 3 h + 4
// words of a component in LDS; one that needs more is "big" too (the whole-wave builds scan for themselves)
constexpr int kScanWords = 2048;
__host__ __device__ __forceinline__ int scan_need(int w, int h) { return w + 3 * h + 4; }
// scan_words < 0: the silhouettes come from k_silhouette_mask (fused sweep), which keeps the component's own pixels as h rows of box-relative 32-bit words
constexpr int kMaskScanWords = 512, kMaskScanWordsLarge = 2048;  // 32-bit words per component: frames up to 1920x1200 (two components per wave) / larger ones (a wave each)
// (box-relative 32-bit words: (w + 31) / 32 per row, + a word per row where rows beyond the first two passes park their extents)
__host__ __device__ __forceinline__ int mask_scan_need(int w, int h) { return h * ((w + 31) / 32 + 1); }
__host__ __device__ __forceinline__ bool pack_big(int x_min, int w, int h, int big_points, int pack_words, int scan_words = 0) {
    (void)x_min;
    return pack_need(w, h) > pack_words || pack_points(w, h) > big_points || (scan_words > 0 && scan_need(w, h) > scan_words) ||
           (scan_words < 0 && mask_scan_need(w, h) > -scan_words);
}

// packed 16-bit minimum (v_pk_min_u16)
typedef unsigned short ctag_qus2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t q_pk_min_u16(uint32_t a, uint32_t b) {
    ctag_qus2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    const ctag_qus2 r = __builtin_elementwise_min(x, y);
    uint32_t o;
    __builtin_memcpy(&o, &r, 4);
    return o;
}

struct CornerPre {
    float x, y, dis, ang;
};

// =====================================================================================================
// K6p: packs of a frame's candidates for k_quad_edges_packed: <= 8 components whose LDS needs sum to <= pack_words per wave.
// Candidates are ranked by boundary capacity, LONGEST FIRST, and packed in that order: a pack's 8-lane sub-groups run in
// lockstep, so its time is its longest component's -- components of similar length share a wave -- and the packs come out
// longest first, which is the order the kernel dispatches them in across the whole batch (the long ones used to start
// anywhere and were the kernel's tail: a 150-point boundary is ~0.3 ms of dependent LDS round trips).
// One block per frame; rank sort in LDS like k_candidates.  Oversize components are skipped; the whole-wave builds take them.
// =====================================================================================================
// Block size: 64 for batches; 1024 for calls of a few frames (the rank sort is nc / threads trips of nc comparisons).  The pack builder -- a
// sequential scan of the ordered list -- runs on wave 0 with the list in registers, 64 entries at a time, read back with v_readlane.
__global__ __launch_bounds__(1024) void k_pack(QuadPtrs P, int nframes, int max_per_pack, int big_points, int pack_words, int scan_words) {
    __shared__ int s_key[kLdsCand + 2];
    __shared__ int s_need[kLdsCand];
    __shared__ uint16_t s_ord[kLdsCand];
    const int frame = blockIdx.x;
    if (frame >= nframes) return;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nc = min(P.ncand[frame], P.cand_cap);
    const Candidate* cand = P.cand + (size_t)frame * P.cand_cap;
    uint32_t* order = P.pack_order + (size_t)frame * P.cand_cap;
    uint32_t* packs = P.packs + (size_t)frame * P.cand_cap;
    auto key_of = [&](const Candidate& c) {
        const int w = c.x_max - c.x_min + 1, h = c.y_max - c.y_min + 1;
        return pack_big(c.x_min, w, h, big_points, pack_words, scan_words) ? -1 : pack_points(w, h);  // -1: not packed, sorts last
    };
    // the pack builder (wave 0): a pack = consecutive entries [0, npacked) of `order`; need_of(r) = LDS words of entry r
    auto build_packs = [&](int npacked, auto need_of) {
        int np = 0, first = -1, cnt = 0, words = 0;  // uniform across the wave
        for (int r0 = 0; r0 < npacked; r0 += 64) {
            const int r = r0 + tid;
            const int need = r < npacked ? need_of(r) : 0;
            const int m = min(64, npacked - r0);
            for (int k = 0; k < m; k++) {
                const int nk = __builtin_amdgcn_readlane(need, k);
                if (cnt > 0 && (cnt == max_per_pack || words + nk > pack_words)) {
                    if (tid == 0) packs[np] = (uint32_t)first | ((uint32_t)cnt << 24);
                    np++;
                    cnt = 0;
                    words = 0;
                }
                if (cnt == 0) first = r0 + k;
                cnt++;
                words += nk;
            }
        }
        if (cnt > 0) {
            if (tid == 0) packs[np] = (uint32_t)first | ((uint32_t)cnt << 24);
            np++;
        }
        return np;
    };
    // scan_words < 0 (the silhouettes come from k_silhouette_mask): the packed components' cluster-pool slots are handed out HERE, in pack order -- an exclusive prefix
    // sum of their C + 64 words (wave 0) -- instead of by one returning atomic per component in the scan kernel, whose chain of dependent loads it would lengthen
    auto slot_prefix = [&](int npacked, auto ci_of) {
        int base = 0;
        for (int r0 = 0; r0 < npacked; r0 += 64) {
            const int r = r0 + tid;
            int v = 0, ci = 0;
            if (r < npacked) {
                ci = ci_of(r);
                const Candidate c = cand[ci];
                v = pack_points(c.x_max - c.x_min + 1, c.y_max - c.y_min + 1) + 64;
            }
            int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(incl, d, 64);
                if (tid >= d) incl += t;
            }
            if (r < npacked) P.cand_aux[(size_t)frame * P.cand_cap + ci].line0 = base + incl - v;
            base += __shfl(incl, 63, 64);
        }
        if (tid == 0) P.clp_used[frame] = base;  // (the chain's first kernel that reserves cluster space: nothing has been added yet)
    };
    if (nc > kLdsCand) {
        // A frame of thousands of blobs (more candidates than the LDS arrays hold).  The order is scheduling only -- results do not
        // depend on it beyond "oversize components last" -- so a counting sort by boundary capacity (descending, capacities of 2047
        // and more in one bucket, ties in arrival order) replaces the rank sort, and the pack builder reads the ordered
        // candidates 64 at a time instead of from LDS.
        constexpr int kB = kLdsCand;  // buckets 0..kB-1: capacity kB-1-b (longest first); bucket kB: oversize
        auto bucket_of = [&](int key) { return key < 0 ? kB : kB - 1 - min(key, kB - 1); };
        for (int b = tid; b <= kB; b += nt) s_key[b] = 0;
        __syncthreads();
        for (int i = tid; i < nc; i += nt) atomicAdd(&s_key[bucket_of(key_of(cand[i]))], 1);
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int b = 0; b <= kB; b++) {
                const int v = s_key[b];
                s_key[b] = run;
                run += v;
            }
            s_key[kB + 1] = run;
        }
        __syncthreads();
        const int first_big = s_key[kB];  // oversize components: entries [first_big, nc) of `order`
        __syncthreads();
        for (int i = tid; i < nc; i += nt) order[atomicAdd(&s_key[bucket_of(key_of(cand[i]))], 1)] = (uint32_t)i;
        __syncthreads();
        if (tid >= 64) return;
        const int np = build_packs(first_big, [&](int r) {
            const Candidate c = cand[order[r]];
            return pack_need(c.x_max - c.x_min + 1, c.y_max - c.y_min + 1);
        });
        if (tid == 0) {
            P.npacks[2 * frame] = np;
            P.npacks[2 * frame + 1] = nc - first_big;
        }
        if (scan_words < 0) slot_prefix(first_big, [&](int r) { return (int)order[r]; });
        return;
    }
    for (int i = tid; i < nc; i += nt) {
        const Candidate c = cand[i];
        const int w = c.x_max - c.x_min + 1, h = c.y_max - c.y_min + 1;
        s_key[i] = key_of(c);
        s_need[i] = pack_need(w, h);  // for the pack builder below: it should not wait for global memory
    }
    if (tid == 0) s_key[kLdsCand] = 0;  // number of oversize components
    __syncthreads();
    for (int i = tid; i < nc; i += nt) {
        const int ki = s_key[i];
        int rank = 0;
        for (int j = 0; j < nc; j++) {
            const int kj = s_key[j];
            rank += (kj > ki || (kj == ki && j < i)) ? 1 : 0;
        }
        s_ord[rank] = (uint16_t)i;
        if (ki < 0) atomicAdd(&s_key[kLdsCand], 1);
    }
    __syncthreads();
    for (int i = tid; i < nc; i += nt) order[i] = s_ord[i];
    if (tid >= 64) return;
    const int nbig = s_key[kLdsCand];  // the oversize components sort last: entries [nc - nbig, nc) of `order`, the whole-wave builds walk only those
    const int np = build_packs(nc - nbig, [&](int r) { return s_need[s_ord[r]]; });
    if (tid == 0) {
        P.npacks[2 * frame] = np;
        P.npacks[2 * frame + 1] = nbig;
    }
    if (scan_words < 0) slot_prefix(nc - nbig, [&](int r) { return (int)s_ord[r]; });
}

// wave-level ordering point for the 8-lane sub-groups (no s_barrier: the sub-groups of a wave run in lockstep;
// this only stops the compiler from moving LDS accesses across it)
#define SG_SYNC()                                              \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)

#ifndef CTAG_EXPAND_INLINE
#define CTAG_EXPAND_INLINE __forceinline__
#endif
// ---- expand_line's distance test as a filtered exact predicate ---------------------------------------------------------------------
// The fitted line of a step is used for ONE thing: `dist_expand > threshold_expand` of the next candidate point
// (corner_detector.cpp:144,156).  The reference gets there through fitLine(DIST_L2) -- five divisions, atan2, cos, sin in double,
// rounded to float, then a float dot product (~200 FP64-class instructions per fit on this GPU).  The same decision follows from a
// cheap double-precision estimate D of the point's distance to the exact least-squares line whenever D is not within `eps` of the
// threshold; eps bounds |reference's float value - D|:
//   * the reference's float evaluation of x*vy - y*vx + vx*y0 - vy*x0 on coordinates <= K: 4 products and 3 sums, 9 K 2^-24 in all,
//     plus the float rounding of x0, y0 (2 K 2^-24);
//   * its direction: t = (float)atan2(..)/2 (6e-8), the arguments' cancellation error in double (<= 6e-8 once the anisotropy
//     h / n^2 >= K^2 2^-26, checked below), cos / sin rounded to float (3e-8): 1.5e-7 on each of four terms of size <= K;
//   * this estimate's own error: v_rcp_f64 / v_rsq_f64 without refinement are good to 2^-23 relative, and they enter the centroid (rn) and
//     the direction (rh, rp) of a CENTRED form whose lever is the distance to the centroid, <= K: about 4 x 2^-23 K ~ 0.5e-6 .. 1e-6 K.
//   Sum ~ 2e-6 K; eps = 3e-6 K, a margin of about 1.4.  A test inside the band (or a cluster too isotropic for the bound) sends its
//   sub-group through the exact fits for that round, so the outcome is the reference's in every case (~1 % of the rounds).
//   CTAG_OPT_EXPAND_EXACT (developer aid) widens the band to infinity -- every round takes the exact fits -- and
//   tests/test_gpu_parity.py::test_expand_line_filter... holds the two against each other and against the oracle on frames, fuzz
//   and non-default threshold_expand.
struct ALine {
    double c, s, x, y;  // unit direction (cos t, sin t), t in [-pi/2, pi/2], and the centroid
    bool ok;            // false: anisotropy too small for the error bound
};
__device__ __forceinline__ ALine approx_line(long long sx, long long sy, long long sxx, long long syy, long long sxy, int cnt, double k2lim) {
    const double n = (double)cnt, fx = (double)sx, fy = (double)sy;
    const double rn = __builtin_amdgcn_rcp(n);
    const double A = ((double)sxx * n - fx * fx) - ((double)syy * n - fy * fy);  // n^2 (dx2 - dy2)
    const double B = 2.0 * ((double)sxy * n - fx * fy);                           // n^2 2 dxy
    const double h2 = A * A + B * B;
    const double rh = __builtin_amdgcn_rsq(h2);  // 1 / h
    const double u = ctm::fabs64(A) * rh;        // |cos 2t|
    const double p = 0.5 + 0.5 * u;              // the larger of cos^2 t, sin^2 t: >= 1/2
    const double rp = __builtin_amdgcn_rsq(p);
    const double big = p * rp, small = (0.5 * ctm::fabs64(B) * rh) * rp;
    const bool bneg = B < 0.0;
    ALine L;
    L.c = A >= 0.0 ? big : small;
    const double sa = A >= 0.0 ? small : big;
    L.s = bneg ? -sa : sa;
    L.x = fx * rn;
    L.y = fy * rn;
    L.ok = h2 * (rn * rn) * (rn * rn) >= k2lim * k2lim;  // (h / n^2)^2 >= (K^2 2^-26)^2; false for h2 == 0 and for NaN
    return L;
}

// expand_line (corner_detector.cpp:125-169) for one sub-group (8 lanes, or the whole wave), speculatively: in every round lane t
// assumes the next t+1 candidate points are all accepted and builds the exact integer moment sums of that prefix; the point of step t
// is then tested against the line of step t-1 -- by the filtered predicate above, by the reference's own fit where that is not
// decisive -- exactly as the sequential loop does, and the prefix up to the first event (distance test fails, `left == right`,
// or every point used) is committed.  Returns nl / nr = points added on the left / right side.  All lanes return the same values.
// lane i <- lane i - D within its row of 16 lanes (v_mov_b32 with DPP row_shr: one vector instruction, no LDS crossbar round trip); lanes without a
// source get 0.  The 8-lane sub-groups are halves of such rows: the lanes that would read across a sub-group's edge are the ones that ignore the value.
template <int D>
__device__ __forceinline__ int dpp_shr(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + D, 0xf, 0xf, false);
}
template <int D>
__device__ __forceinline__ double dpp_shr(double v) {
    return __hiloint2double(dpp_shr<D>(__double2hiint(v)), dpp_shr<D>(__double2loint(v)));
}
template <int D>
__device__ __forceinline__ float dpp_shr(float v) {
    return __int_as_float(dpp_shr<D>(__float_as_int(v)));
}
// All-reduce over a sub-group of 8 lanes or over the wave, without the LDS crossbar: lane ^ 1 and lane ^ 2 by DPP quad permutes, the other quad of
// the 8 lanes by row_half_mirror (lane i <- 7 - i), the other half of a row of 16 by row_mirror, and the wave's four rows by four v_readlane.  The
// operations are exact (integer sums) or a total order (best distance, ties by index), so the order of combination does not matter.  (A
// __shfl_xor is a ds_bpermute_b32: address arithmetic, the instruction and ~100 cycles before the value can be used -- per step, and a wave's
// reduction has six; the split loop of the RDP runs one per round.)
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(dpp_mov<CTRL>(__float_as_int(v)));
}
template <int CTRL>
__device__ __forceinline__ long long dpp_mov(long long v) {
    const unsigned lo = (unsigned)dpp_mov<CTRL>((int)(unsigned)((unsigned long long)v & 0xffffffffull)), hi = (unsigned)dpp_mov<CTRL>((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)((unsigned long long)lo | ((unsigned long long)hi << 32));
}
__device__ __forceinline__ long long readlane_ll(long long v, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v & 0xffffffffull), l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), l);
    return (long long)((unsigned long long)lo | ((unsigned long long)hi << 32));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppRowMirror = 0x140;
template <int SG, class T>
__device__ __forceinline__ T sg_sum(T v) {  // T: int, long long
    v += dpp_mov<kDppXor1>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppHalfMirror>(v);
    if constexpr (SG == 64) {
        v += dpp_mov<kDppRowMirror>(v);
        if constexpr (sizeof(T) == 8) v = (T)(readlane_ll((long long)v, 0) + readlane_ll((long long)v, 16) + readlane_ll((long long)v, 32) + readlane_ll((long long)v, 48));
        else v = (T)(__builtin_amdgcn_readlane((int)v, 0) + __builtin_amdgcn_readlane((int)v, 16) + __builtin_amdgcn_readlane((int)v, 32) + __builtin_amdgcn_readlane((int)v, 48));
    }
    return v;
}
// the best (d, i) pair of the sub-group under `better(od, oi, d, i)` ("the other pair beats mine"): every lane ends with the same pair
template <int SG, class Better>
__device__ __forceinline__ void sg_best(float& bd, int& bi, Better better) {
    auto step = [&](float od, int oi) {
        if (better(od, oi, bd, bi)) {
            bd = od;
            bi = oi;
        }
    };
    step(dpp_mov<kDppXor1>(bd), dpp_mov<kDppXor1>(bi));
    step(dpp_mov<kDppXor2>(bd), dpp_mov<kDppXor2>(bi));
    step(dpp_mov<kDppHalfMirror>(bd), dpp_mov<kDppHalfMirror>(bi));
    if constexpr (SG == 64) {
        step(dpp_mov<kDppRowMirror>(bd), dpp_mov<kDppRowMirror>(bi));
        float rd[4];
        int ri[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            rd[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bd), 16 * r));
            ri[r] = __builtin_amdgcn_readlane(bi, 16 * r);
        }
        bd = rd[0];
        bi = ri[0];
#pragma unroll
        for (int r = 1; r < 4; r++) step(rd[r], ri[r]);
    }
}
// RED / rl: lanes (and this lane's index among them) that add up the moments of the initial span.  The whole-wave build speculates over its first
// EIGHT lanes only (SG = 8, RED = 64; the other lanes repeat them): an edge rarely grows by more than a few points, and a 64-step round paid six
// prefix-scan steps, integer divisions for the wrapped indices and a line estimate per lane for steps that were thrown away.
template <int SG, int RED = SG>
__device__ CTAG_EXPAND_INLINE void sg_expand_line(const uint32_t* W, int n, int init, int end, int sl, int lane0, int sgshift, float thr_expand, float kmax, float eps_scale,
                                               int& nl_out, int& nr_out, int rl = -1, unsigned long long* dbg = nullptr) {
    if (rl < 0) rl = sl;
    const unsigned long long dbg_t0 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    constexpr unsigned long long kSgMask = SG == 64 ? ~0ull : ((1ull << (SG & 63)) - 1ull);
    const double thr = (double)thr_expand, eps = (double)eps_scale * (double)kmax, k2lim = (double)kmax * (double)kmax * 1.4901161193847656e-08;  // K^2 2^-26
    long long Sx = 0, Sy = 0, Sxx = 0, Syy = 0, Sxy = 0;
    for (int k = init + rl; k <= end; k += RED) {
        const long long x = ux(W[k]), y = uy(W[k]);
        Sx += x;
        Sy += y;
        Sxx += x * x;
        Syy += y * y;
        Sxy += x * y;
    }
    Sx = sg_sum<RED>(Sx);
    Sy = sg_sum<RED>(Sy);
    Sxx = sg_sum<RED>(Sxx);
    Syy = sg_sum<RED>(Syy);
    Sxy = sg_sum<RED>(Sxy);
    int m = end - init + 1;
    ALine lineA = approx_line(Sx, Sy, Sxx, Syy, Sxy, m, k2lim);  // of the committed prefix
    bool fl = false, fr = false;
    int left = init - 1, right = end + 1, nl = 0, nr = 0;
    if (dbg) dbg[0] += __builtin_amdgcn_s_memtime() - dbg_t0;
    while ((!fl || !fr) && (left != right)) {
        if (dbg) dbg[1] += 1;
        const int mode = (!fl && !fr) ? 0 : (!fl ? 1 : 2);  // 0: L,R alternate; 1: left only; 2: right only
        // The index bookkeeping of steps 0..sl under the "all accepted" assumption, in closed form (a lane per step): before step u the
        // left cursor has moved kl times and the right one kr times; the k-th left step reads index (left - k) mod n and leaves the raw
        // cursor at that index - 1, the k-th right step reads (right + k) mod n and leaves index + 1 (`left` is in [-1, n-1], `right`
        // in [0, n]: -1 and n are the not-yet-wrapped values the reference compares).
        const bool stepL = mode == 0 ? ((sl & 1) == 0) : (mode == 1);
        const bool checkC = mode == 0 ? ((sl & 1) == 0) : true;  // `left != right` is tested at the top of an iteration
        const int kl = mode == 0 ? ((sl + 1) >> 1) : (mode == 1 ? sl : 0);  // left steps among steps 0..sl-1
        const int kr = mode == 0 ? (sl >> 1) : (mode == 2 ? sl : 0);
        auto idxL = [&](int k) {
            if constexpr (SG <= 8) {  // k <= 8 and n >= 3: at most three wraps, no division
                int x = left - k;
                x += x < 0 ? n : 0;
                x += x < 0 ? n : 0;
                x += x < 0 ? n : 0;
                return x;
            } else {
                return ((left - k) % n + n) % n;
            }
        };
        auto idxR = [&](int k) {
            if constexpr (SG <= 8) {
                int x = right + k;
                x -= x >= n ? n : 0;
                x -= x >= n ? n : 0;
                x -= x >= n ? n : 0;
                return x;
            } else {
                return (right + k) % n;
            }
        };
        const int l_before = kl == 0 ? left : idxL(kl - 1) - 1;
        const int r_before = kr == 0 ? right : idxR(kr - 1) + 1;
        const bool cstop = checkC && l_before == r_before;
        const unsigned long long cm = ((unsigned long long)__ballot(cstop) >> sgshift) & kSgMask;
        const int tc = cm ? (int)(__ffsll((unsigned long long)cm) - 1) : 99;  // every lane holds the sub-group's first stop
        const int q_idx = stepL ? idxL(kl) : idxR(kr);
        const int l = stepL ? q_idx - 1 : l_before;
        const int r = stepL ? r_before : q_idx + 1;
        const uint32_t qpt = W[q_idx];
        // inclusive prefix sums of the five moments over the lanes; 32 bits hold them: coordinates < 2^13 (an 8K frame at
        // half resolution), products < 2^26, 64 of them < 2^32
        uint32_t a0 = (uint32_t)ux(qpt), a1 = (uint32_t)uy(qpt), a2 = a0 * a0, a3 = a1 * a1, a4 = a0 * a1;
        static_assert(SG == 8, "the prefix sums below are three DPP steps inside half a row of 16 lanes");
#define CTAG_EXPAND_SCAN(D)                                                                                                          \
        {                                                                                                                            \
            const uint32_t b0 = (uint32_t)dpp_shr<D>((int)a0), b1 = (uint32_t)dpp_shr<D>((int)a1), b2 = (uint32_t)dpp_shr<D>((int)a2), \
                           b3 = (uint32_t)dpp_shr<D>((int)a3), b4 = (uint32_t)dpp_shr<D>((int)a4);                                    \
            if (sl >= D) {                                                                                                           \
                a0 += b0;                                                                                                            \
                a1 += b1;                                                                                                            \
                a2 += b2;                                                                                                            \
                a3 += b3;                                                                                                            \
                a4 += b4;                                                                                                            \
            }                                                                                                                        \
        }
        CTAG_EXPAND_SCAN(1)
        CTAG_EXPAND_SCAN(2)
        CTAG_EXPAND_SCAN(4)
#undef CTAG_EXPAND_SCAN
        const long long px = a0, py = a1, pxx = a2, pyy = a3, pxy = a4;
        // ---- the distance test of step sl against the line of step sl - 1: filtered, exact where the filter is not decisive
        const ALine mineA = approx_line(Sx + px, Sy + py, Sxx + pxx, Syy + pyy, Sxy + pxy, m + sl + 1, k2lim);
        ALine lpA;  // the line of step sl - 1: the lane below (step 0: the committed prefix's)
        lpA.c = dpp_shr<1>(mineA.c);
        lpA.s = dpp_shr<1>(mineA.s);
        lpA.x = dpp_shr<1>(mineA.x);
        lpA.y = dpp_shr<1>(mineA.y);
        lpA.ok = dpp_shr<1>((int)mineA.ok) != 0;
        if (sl == 0) lpA = lineA;
        const double D = ctm::fabs64(((double)ux(qpt) - lpA.x) * lpA.s - ((double)uy(qpt) - lpA.y) * lpA.c);
        bool fail = D > thr;
        const bool unsure = !lpA.ok || !(ctm::fabs64(D - thr) > eps);
        if (((unsigned long long)__ballot(unsure) >> sgshift) & kSgMask) {  // uniform within the sub-group
            if (dbg) dbg[2] += 1;
            float mine[4], base[4];
            moments_to_line((double)(Sx + px), (double)(Sy + py), (double)(Sxx + pxx), (double)(Syy + pyy), (double)(Sxy + pxy),
                            (double)(float)(m + sl + 1), mine);
            moments_to_line((double)Sx, (double)Sy, (double)Sxx, (double)Syy, (double)Sxy, (double)(float)m, base);  // the committed prefix's line
            float lp[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float up = dpp_shr<1>(mine[k]);
                lp[k] = sl == 0 ? base[k] : up;
            }
            const float de = ctm::fabs32(ux(qpt) * lp[1] - uy(qpt) * lp[0] + lp[0] * lp[3] - lp[1] * lp[2]);
            fail = de > thr_expand;
        }
        const unsigned long long failm = ((unsigned long long)__ballot(fail) >> sgshift) & kSgMask;
        const int tf = failm ? (int)(__ffsll(failm) - 1) : 99;
        const int te_raw = n - m - 1;  // the add of step te makes Slide.size() == edge_point.size()
        const int te = (te_raw >= 0 && te_raw < SG) ? te_raw : 99;
        int accepted;
        bool done = false, failed_step = false;
        if (tc < 99 && tc <= tf && tc <= te) {
            accepted = tc;
            done = true;
        } else if (tf < 99 && tf <= te) {
            accepted = tf;
            failed_step = true;
        } else if (te < 99) {
            accepted = te + 1;
            done = true;
        } else {
            accepted = SG;
        }
        if (accepted > 0) {
            const int src = lane0 + accepted - 1;
            // the whole-wave build's lanes agree on `src`: v_readlane; the packs' sub-groups each have their own: the crossbar (32-bit values)
            auto pick = [&](int v) { return RED == 64 ? __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)) : __shfl(v, src); };
            auto pickd = [&](double v) { return __hiloint2double(pick(__double2hiint(v)), pick(__double2loint(v))); };
            Sx += (long long)(uint32_t)pick((int)a0);
            Sy += (long long)(uint32_t)pick((int)a1);
            Sxx += (long long)(uint32_t)pick((int)a2);
            Syy += (long long)(uint32_t)pick((int)a3);
            Sxy += (long long)(uint32_t)pick((int)a4);
            lineA.c = pickd(mineA.c);
            lineA.s = pickd(mineA.s);
            lineA.x = pickd(mineA.x);
            lineA.y = pickd(mineA.y);
            lineA.ok = pick((int)mineA.ok) != 0;
            left = pick(l);
            right = pick(r);
            m += accepted;
            if (mode == 0) {
                nl += (accepted + 1) >> 1;
                nr += accepted >> 1;
            } else if (mode == 1) {
                nl += accepted;
            } else {
                nr += accepted;
            }
        }
        if (failed_step) {
            const bool stepLf = mode == 0 ? ((accepted & 1) == 0) : (mode == 1);
            const int fidx = RED == 64 ? __builtin_amdgcn_readlane(q_idx, __builtin_amdgcn_readfirstlane(lane0 + accepted)) : __shfl(q_idx, lane0 + accepted);
            if (stepLf) {
                fl = true;
                left = fidx;  // `left` keeps the tested (already wrapped) index
            } else {
                fr = true;
                right = fidx;
            }
        }
        if (done) break;
    }
    nl_out = nl;
    nr_out = nr;
}

// =====================================================================================================
// K6m (round 6; batches that took the fused sweep): the silhouette of every packed component from the threshold MASK k_decimate_mask left (1 bit per pixel,
// still in the workspace) instead of from every label of its bounding box.  corner_detector.cpp:184-232 asks, per row and per column of the box, for the
// first and last pixel OF THE COMPONENT; the label scan answers by testing every label of the box (~1700 for a bar of ~500 pixels, ~100 instructions per row
// and wave, 48 % of k_quad_edges_packed's cycles).  Here a lane owns a ROW of the box: it cuts the row's mask words into foreground runs with bit tricks --
// a run is a maximal horizontal stretch of foreground, hence of ONE component -- and asks for ONE label per run (its first pixel: tile-local label -> pool entry
// -> root): 1-3 probes per row.  The component's own pixels of the row (the OR of its runs) go to LDS as box-relative 32-bit words, the row's extents are their
// first / last set bits; then a lane owns a COLUMN and gathers the rows' bits of its column for the column's first / last set bit.  A run cut by the box's edge belongs to another component
// (the box of ours would otherwise reach further) and says so when probed.  Output: what PHASE 3 of k_quad_edges_packed leaves -- tb / lr in the component's
// cluster-pool slot, the slot in cand_aux -- so the packed builds run with PRESCAN = true behind it.
// =====================================================================================================
// Two components per wave, 32 lanes each (the packs' order puts components of similar size side by side).  A component is a chain of dependent loads -- order entry ->
// candidate -> mask words -> labels -> roots -- so (1) the probes of a row are issued TOGETHER, the first four runs of its first kMsRel words, then their root look-ups
// together (a row costs two round trips, not two per run); (2) the next pair's order entry, candidate and slot are requested before the current pair is worked on; (3) the
// slots come from k_pack's prefix sum, not from a returning atomic.  A row with more runs, or a box over more words, takes the one-probe-at-a-time loop for what is left.
// kMsRel: box-relative 32-bit words of a row held in registers and probed as a batch; kMsSub: lanes (rows at a time) per component; kWords: LDS words per component.
// <4, 32, 512> for frames up to 1920x1200 (boxes of ~75 x 25), <8, 64, 2048> for larger ones (~200 x 100: a wave per component).
// docs/history.md R6-A has the six forms and what each measured (1.08 -> 0.72 ms per 4096 1080p frames); the funnel shift that aligns a row with its box is ONE
// v_alignbit_b32 per word.
constexpr int kMsProbes = 4;
#ifndef CTAG_MS_WAVES
#define CTAG_MS_WAVES 6  // waves per SIMD of the small build (its 4 KB of LDS per wave would allow eight: measured 5 / 6 / 7 / 8 -> 3.49 / 3.54 / 3.63 / 3.78 ms quad_edges)
#endif
template <int kMsRel, int kMsSub, int kWords>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kMsRel <= 4 ? CTAG_MS_WAVES : 4, 8))) void k_silhouette_mask(QuadPtrs P, FrameGeom g, int nframes) {
    static_assert(32 * kMsRel <= 256, "run starts are packed as eight-bit column offsets");
    static_assert(kMsSub == 32 || kMsSub == 64, "one or two components per wave");
    constexpr int kPerWave = 64 / kMsSub;
    __shared__ uint32_t s_comp2[kPerWave][kWords];
    __shared__ int s_tb[kPerWave][8];
    const int frame = blockIdx.x;
    if (frame >= nframes) return;
    const int lane = threadIdx.x, sub = lane / kMsSub, sl = lane % kMsSub;
    uint32_t* const s_comp = s_comp2[sub];
    const int nc = min(P.ncand[frame], P.cand_cap);
    const int npk = nc - P.npacks[2 * frame + 1];  // every packed component: entries [0, nc - oversize) of k_pack's order
    if (kPerWave * (int)blockIdx.y >= npk) return;  // nothing for this wave (a frame without packed components has no order entry to read, either)
    const int mwords = 2 * P.mask_words;           // 32-bit words per mask row
    const uint16_t* __restrict__ limg = P.labels + ((size_t)frame * g.hrows) * g.lp;
    const int32_t* __restrict__ tbase = P.tile_base + (size_t)frame * g.tiles_x * g.tiles_y;
    const int32_t* __restrict__ rootof = P.root_of + (size_t)frame * g.pool_cap;
    const uint32_t* __restrict__ mimg = reinterpret_cast<const uint32_t*>(P.mask) + (size_t)frame * g.hrows * mwords;
    const uint32_t* __restrict__ order = P.pack_order + (size_t)frame * P.cand_cap;
    const Candidate* __restrict__ cands = P.cand + (size_t)frame * P.cand_cap;
    const CandAux* __restrict__ auxs = P.cand_aux + (size_t)frame * P.cand_cap;
    // this sub-group's components: entries kPerWave * blockIdx.y + sub, + kPerWave * gridDim.y, ...
    int pk = kPerWave * (int)blockIdx.y + sub;
    const int step = kPerWave * (int)gridDim.y;
    int ci_n = (int)order[min(pk, max(npk - 1, 0))];
    Candidate cd_n = cands[ci_n];
    int p0_n = auxs[ci_n].line0;  // the component's cluster-pool slot, handed out by k_pack
    for (; __ballot(pk < npk) != 0ull; pk += step) {  // wave-uniform trip count: the barrier below is the wave's
        __syncthreads();  // (one wave) the previous components' words are read
        const bool act = pk < npk;
        const int ci = ci_n, p0 = p0_n;
        const Candidate cd = cd_n;
        {   // the next pair's header, in flight under this pair's work
            const int pn = min(pk + step, max(npk - 1, 0));
            ci_n = (int)order[pn];
            cd_n = cands[ci_n];
            p0_n = auxs[ci_n].line0;
        }
        const int x_min = cd.x_min, y_min = cd.y_min;
        const int w = act ? cd.x_max - cd.x_min + 1 : 0, h = act ? cd.y_max - cd.y_min + 1 : 0;
        const int C = pack_points(w, h);
        CandAux* aux = P.cand_aux + (size_t)frame * P.cand_cap + ci;
        // p0: the component's cluster space (what the packed builds reserve after their traversal) and, in it for now, the silhouette: w + h + 4 <= C + 64 words.
        // The row's bits are brought into BOX-relative words (bit b of word j = column x_min + 32 j + b): ceil(w / 32) of them, only the last one partly valid.
        const int j0 = x_min >> 5, sh = x_min & 31, nwr = (w + 31) >> 5;
        // What the kernel is bound by is the number of cache lines its lanes touch (a lane per row: every lane's load is a line of its own; the first form: 367 line
        // accesses per component, the texture addressers 65 % busy): so the row's words come as ONE 16-byte load + one word, a run is probed only by the lanes that have
        // it, and the label tiles' pool bases -- a handful of values per component -- are loaded once, by the sub-group's first lanes, and handed round through LDS.
        const int ty0 = y_min / kTileH, tcx0 = x_min / kTileW;
        const int nty = act ? (y_min + h - 1) / kTileH - ty0 + 1 : 1, ntx = act ? (x_min + w - 1) / kTileW - tcx0 + 1 : 1;
        const bool tb_regs = nty * ntx <= 8;
        if (sl < 8) s_tb[sub][sl] = tbase[min((ty0 + sl / ntx) * g.tiles_x + tcx0 + sl % ntx, g.tiles_x * g.tiles_y - 1)];  // lane i < 8: tile (i / ntx, i % ntx) of the box
        __syncthreads();
        // ---- rows: a lane per row
        uint32_t lr_mine[2] = {0u, 0u};  // of rows sl and sl + kMsSub (more rows: parked in LDS as they are formed)
        for (int y = sl, yi = 0; y < h; y += kMsSub, yi++) {
            const uint32_t* __restrict__ mr = mimg + (size_t)(y_min + y) * mwords + j0;
            const uint16_t* __restrict__ lrow = limg + (size_t)(y_min + y) * g.lp + x_min;
            const int trow = ((y_min + y) / kTileH) * g.tiles_x;
            auto rel_word = [&](uint32_t lo, uint32_t hi, int j) -> uint32_t {  // box-relative word j from the two aligned words that hold it
                const uint32_t v = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh);  // ({hi, lo} >> sh)[31:0]
                const int nb = w - 32 * j;  // valid bits
                return nb >= 32 ? v : nb <= 0 ? 0u : (v & ((1u << nb) - 1u));
            };
            const int tyi = (y_min + y) / kTileH - ty0;
            auto base_of = [&](int xr) -> int {  // pool base of the label tile of column x_min + xr in this row
                const int tc = (x_min + xr) / kTileW;
                return tb_regs ? s_tb[sub][min(tyi * ntx + tc - tcx0, 7)] : tbase[trow + tc];
            };
            auto member_of = [&](int xr) -> bool {  // one probe by itself (rows with more runs than the batch below holds): label -> pool entry -> root
                const unsigned l = lrow[xr];
                return l != 0u && l < 0x8000u && rootof[tbase[trow + (x_min + xr) / kTileW] + (int)l - 1] == cd.root;  // (bit 15: an unpublished speck, never a candidate)
            };
            uint32_t mw[kMsRel];
            {
                // kMsRel + 1 consecutive words from the row's first one (beyond the box: masked below; beyond the row: the next row's bytes -- the mask lives in the
                // workspace's half-size image, eight times its size)
                uint32_t a[kMsRel + 1];
                struct __attribute__((packed, aligned(4))) W4 { uint32_t v[4]; };
#pragma unroll
                for (int q = 0; q < kMsRel / 4; q++) {
                    const W4 t = *reinterpret_cast<const W4*>(mr + 4 * q);
#pragma unroll
                    for (int u = 0; u < 4; u++) a[4 * q + u] = t.v[u];
                }
                a[kMsRel] = mr[kMsRel];
#pragma unroll
                for (int j = 0; j < kMsRel; j++) mw[j] = rel_word(a[j], a[j + 1], j);
            }
            // run starts of the row in order (a run that goes on from the word before is not a start): the first kMsProbes of them -- column offsets packed eight bits
            // each -- are probed at once, labels first, then their roots
            uint32_t xs = 0u;
            int nr = 0;
#pragma unroll
            for (int j = 0; j < kMsRel; j++) {
                const uint32_t carry = j > 0 ? (mw[j > 0 ? j - 1 : 0] >> 31) : 0u;
                uint32_t st = mw[j] & ~((mw[j] << 1) | carry);
                while (st && nr < kMsProbes) {
                    xs |= (uint32_t)(32 * j + (int)__builtin_ctz(st)) << (8 * nr);
                    nr++;
                    st &= st - 1u;
                }
            }
            // The probes go out together: the labels of the runs the row has (a lane without a k-th run sits that load out), then -- nothing in between uses a
            // loaded value -- their roots.  (The first build's ISA had a wait behind every probe: its loads sat in branches whose results were used at once.)
            unsigned lp[kMsProbes];
            int bp[kMsProbes];
#pragma unroll
            for (int k = 0; k < kMsProbes; k++) {
                const int xr = (int)((xs >> (8 * k)) & 255u);
                lp[k] = 0u;
                if (k < nr) lp[k] = lrow[xr];
                bp[k] = base_of(xr);
            }
            int rp[kMsProbes];
#pragma unroll
            for (int k = 0; k < kMsProbes; k++) {
                rp[k] = -1;
                if (lp[k] != 0u && lp[k] < 0x8000u) rp[k] = rootof[bp[k] + (int)lp[k] - 1];  // (bit 15: an unpublished speck, never a candidate)
            }
            unsigned memb = 0u;
#pragma unroll
            for (int k = 0; k < kMsProbes; k++) memb |= (rp[k] == cd.root ? 1u : 0u) << k;
            bool cin = false, cmem = false;  // the word before ends in a run / that run belongs to the component
            int left = -1, right = -1, idx = 0;
            auto do_word = [&](int j, uint32_t m) {
                uint32_t comp = 0u, starts = m & ~((m << 1) | (cin ? 1u : 0u));
                bool last31 = false;
                if (cin && (m & 1u)) {  // goes on from the word before
                    const uint32_t run = ((m + 1u) ^ m) & m;
                    if (cmem) comp |= run;
                    if (run >> 31) last31 = cmem;
                }
                while (starts) {
                    const uint32_t sb = starts & (0u - starts);
                    starts ^= sb;
                    bool member;
                    if (idx < nr) member = ((memb >> idx) & 1u) != 0u;  // (nr: the runs the batch probed -- the first kMsProbes of the first kMsRel words)
                    else member = member_of(32 * j + (int)__builtin_ctz(sb));
                    idx++;
                    const uint32_t run = ((m + sb) ^ m) & m;  // the carry of the addition runs through the run and stops behind it
                    if (member) comp |= run;
                    if (run >> 31) last31 = member;
                }
                cin = (m >> 31) != 0u;
                cmem = last31;
                s_comp[y * nwr + j] = comp;
                if (comp) {
                    if (left < 0) left = 32 * j + (int)__builtin_ctz(comp);
                    right = 32 * j + 31 - (int)__builtin_clz(comp);
                }
            };
#pragma unroll
            for (int j = 0; j < kMsRel; j++)
                if (j < nwr) do_word(j, mw[j]);
            for (int j = kMsRel; j < nwr; j++)  // wider boxes: the rest a word at a time (its runs a probe at a time)
                do_word(j, rel_word(mr[j], mr[j + 1], j));
            const uint32_t v = left < 0 ? 0u : ((uint32_t)(left + 2) | ((uint32_t)(right + 2) << 16));
            if (yi < 2) lr_mine[yi] = v;
            else s_comp[h * nwr + y] = v;  // (further rows: parked behind the words -- mask_scan_need counts a word more per row than the box has)
        }
        const bool fits = (uint32_t)(p0 + C + 64) <= P.cl_cap;
        if (act && !fits && sl == 0) {
            atomicOr(&P.frame_flags[frame], CTAG_FLAG_POOL_OVERFLOW);
            aux->line0 = -1;
            aux->acx = 0.f;
            aux->acy = 0.f;
            aux->n_boundary = 0;
        }
        __syncthreads();
        if (act && fits) {
            uint32_t* __restrict__ slot = P.cl_pool + (size_t)frame * P.cl_cap + p0;  // tb: [0, w + 2), lr: [w + 2, w + h + 4)
            if (sl < h) slot[w + 2 + sl + 1] = lr_mine[0];
            if (sl + kMsSub < h) slot[w + 2 + sl + kMsSub + 1] = lr_mine[1];
            for (int y = sl + 2 * kMsSub; y < h; y += kMsSub) slot[w + 2 + y + 1] = s_comp[h * nwr + y];
            // columns: a lane per column, kMsSub columns at a time -- the lanes of one 32-bit word read the SAME LDS address (a broadcast) and test their own bit; the
            // column's rows are gathered 32 at a time into a register (three instructions per row: read, bit extract, shift-or), first / last row from its ends
            for (int ps = 0; kMsSub * ps < w; ps++) {
                const int x = kMsSub * ps + sl, wd = min(x >> 5, nwr - 1), bit = x & 31;
                int top = -1, bot = -1;
                for (int y0 = 0; y0 < h; y0 += 32) {
                    uint32_t T = 0u;
                    const int ye = min(32, h - y0);
                    const uint32_t* cp = s_comp + y0 * nwr + wd;
#pragma unroll 8
                    for (int k = 0; k < ye; k++) T |= ((cp[k * nwr] >> bit) & 1u) << k;
                    if (T) {
                        if (top < 0) top = y0 + (int)__builtin_ctz(T);
                        bot = y0 + 31 - (int)__builtin_clz(T);
                    }
                }
                if (x < w) slot[x + 1] = top < 0 ? 0u : ((uint32_t)(top + 2) | ((uint32_t)(bot + 2) << 16));
            }
            if (sl == 0) {
                slot[0] = 0u;
                slot[w + 1] = 0u;
                slot[w + 2] = 0u;
                slot[w + 2 + h + 1] = 0u;
                aux->line0 = p0;
                aux->acx = 0.f;
                aux->acy = 0.f;
                aux->n_boundary = 0;
            }
        }
    }
}

// =====================================================================================================
// K6a (packed): the same boundary -> 4 edge clusters computation as k_quad_edges, 8 components per wave.
// Every lane of a sub-group executes the serial control flow redundantly (uniform within the sub-group), so no
// broadcasts are needed; loops over pixels / boundary points are strided over the 8 lanes.
// =====================================================================================================
// REFPRM: the boundary tunables are the reference's (threshold_line 1.8, threshold_expand 1.2, collinearity 1.05) and compiled in;
// a handle created with other values (ctag_create_ex) runs the builds that read them from P.  (As kernel arguments in the one
// build they cost the packed kernel 15 %: 4.9 -> 5.65 ms per 4096 frames -- its register allocation is that tight.)
// PHASE: 0 = the whole computation in one kernel; 1 = boundary only (silhouette, ordered traversal, centroid, rotation: the rotated
// boundary list goes to the component's slot of the cluster pool, its length / centroid / slot to cand_aux); 2 = the split into edge
// clusters only (reads them back).  The packed builds run as 1 then 2: the two halves want different registers -- dependent LDS
// round trips with many live indices in the first, FP64 line fits in the second -- and in one kernel at 128 VGPRs each change to
// one half re-spilled the other (the traversal tripled when the line fits got cheaper).
// PHASE 3 (batches; always the whole wave on one component): the silhouette scan ALONE, for every PACKED component of the frame -- tb / lr go to the
// component's cluster-pool slot (reserved here), and the packed builds run with PRESCAN = true: they load the two arrays instead of
// scanning.  The scan is the part of the boundary stage that waits for global memory (label rows); as a kernel of its own it keeps 64
// lanes on one component's rows (the packed build has 8), needs 11 KB of LDS and few registers -- three and a half waves per SIMD
// where the 4K packed build runs two -- and the builds behind it lose their largest phase (43-56 % of their cycles).
template <int SG, int WORDS, int WAVES, bool DYN, bool REFPRM, int PHASE = 0, bool PRESCAN = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_quad_edges_packed(QuadPtrs P, FrameGeom g, int nframes, int tier_lo) {
    static_assert(PHASE != 3 || (SG == 64 && !PRESCAN), "the scan-only build is a whole-wave build");
    static_assert(!PRESCAN || (SG == 8 && PHASE != 2), "PRESCAN: a packed build that would otherwise scan");
    const float k_thr_line = REFPRM ? 1.8f : P.thr_line, k_thr_expand = REFPRM ? 1.2f : P.thr_expand;
    const int k_c2_far = REFPRM ? 2 : P.c2_far, k_c2_near = REFPRM ? 1 : P.c2_near;
    static_assert(SG == 8 || SG == 64, "8 lanes per component (packs) or the whole wave (oversize components)");
    __shared__ uint32_t s_static[DYN ? 1 : WORDS];
    extern __shared__ uint32_t s_dynamic[];
    uint32_t* const s_mem = DYN ? s_dynamic : s_static;
    const int frame = blockIdx.x;
    if (frame >= nframes) return;
    const int lane = threadIdx.x, sub = lane / SG, sl = lane % SG, lane0 = lane - sl;
    // SG == 8: the frame's packs; SG == 64: the oversize tail of k_pack's order, one component per wave
    const int nc = min(P.ncand[frame], P.cand_cap);
    const int npk = PHASE == 3 ? nc - P.npacks[2 * frame + 1]  // every packed component: entries [0, nc - oversize) of k_pack's order
                               : P.npacks[2 * frame + (SG == 8 ? 0 : 1)];
    const uint16_t* __restrict__ limg = P.labels + ((size_t)frame * g.hrows) * g.lp;
    const int32_t* __restrict__ tbase = P.tile_base + (size_t)frame * g.tiles_x * g.tiles_y;
    const int32_t* __restrict__ rootof = P.root_of + (size_t)frame * g.pool_cap;

    unsigned long long t_prev = 0;
    auto stamp = [&](int phase) {  // developer aid: wave-level cycles per phase (the sub-groups reconverge between phases)
        if (P.stamps) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (phase >= 0 && lane == 0) atomicAdd(&P.stamps[phase + (SG == 64 ? 8 : 0)], t - t_prev);
            t_prev = t;
        }
    };
    // blockIdx.x (the fast dispatch index) is the frame, blockIdx.y the pack rank: the longest packs of ALL frames start first
    for (int pk = blockIdx.y; pk < npk; pk += gridDim.y) {
        __syncthreads();  // single-wave workgroup: the previous pack is done with s_mem
        stamp(-1);
        int first, cnt;
        if constexpr (SG == 8) {
            const uint32_t pw = P.packs[(size_t)frame * P.cand_cap + pk];
            first = (int)(pw & 0xffffffu);
            cnt = (int)(pw >> 24);
        } else if constexpr (PHASE == 3) {
            first = pk;
            cnt = 1;
        } else {
            first = nc - npk + pk;
            cnt = 1;
        }
        const bool act = sub < cnt;
        const int ci = (int)P.pack_order[(size_t)frame * P.cand_cap + first + (act ? sub : 0)];
        const Candidate cd = P.cand[(size_t)frame * P.cand_cap + ci];
        const int x_min = cd.x_min, y_min = cd.y_min;
        const int w = cd.x_max - cd.x_min + 1, h = cd.y_max - cd.y_min + 1;
        const int C = pack_points(w, h);
        const int need = act ? (PHASE == 3 ? scan_need(w, h) : pack_need(w, h)) : 0;
        int off = 0;
#pragma unroll
        for (int k = 0; k < 64 / SG; k++) {
            const int v = __shfl(need, k * SG);
            if (k < sub) off += v;
        }
        if (!act) continue;
        CandAux* aux = P.cand_aux + (size_t)frame * P.cand_cap + ci;
        if constexpr (SG == 64 && PHASE != 3) {
            // two builds share the oversize components by working-set size: (tier_lo, WORDS] is this one's
            if (need <= tier_lo) continue;
            if (need > WORDS) {
                if (tier_lo > 0 && sl == 0) {  // larger than the largest build: give up on this component, flagged
                    atomicOr(&P.frame_flags[frame], CTAG_FLAG_POOL_OVERFLOW);
                    aux->line0 = -1;
                    aux->acx = 0.f;
                    aux->acy = 0.f;
                    aux->n_boundary = 0;
                }
                continue;
            }
        }
        uint32_t* mem = s_mem + off;
        // silhouette as two sentinel-padded arrays: tb[x+1] = (top+2) | (bottom+2) << 16 per column, lr[y+1] = (left+2) |
        // (right+2) << 16 per row, 0 = none.  With the +2 bias a neighbour outside the box can never match, so the
        // traversal needs no bounds checks.
        uint32_t* tb = mem;               // w + 2 words
        uint32_t* lr = tb + w + 2;        // h + 2 words
        uint32_t* bufA = lr + h + 2;      // C words
        uint32_t* bufB = bufA + (PHASE == 3 ? h : C);  // C + 1 words (scan only: the row extents, h words each)
        uint32_t* lef = bufA;             // P1 only: row extents by atomics (h <= C words each)
        uint32_t* rig = bufB;
        const int sgshift = sub * SG;

        int n = 0, n_boundary = 0, p0 = 0;
        float acx = 0.f, acy = 0.f;
        if constexpr (PHASE != 2 && !PRESCAN) {
        stamp(6);
        // the component's pixels carry one tile-local label per CCL tile it touches: collect those (tile, label) keys
        // (root entry + its member list built by k_resolve) so the pixel scan needs no gathers
#ifndef CTAG_KEYS_LARGE
#define CTAG_KEYS_LARGE 16
#endif
        // (tile, label) keys a lane keeps: a component with more members falls back to the gather test for every pixel.  The packs of 4K
        // frames hold components that cross four or five 30-row label tiles: eight keys were not enough for most of them
#ifndef CTAG_KEYS_SMALL
#define CTAG_KEYS_SMALL 8
#endif
        constexpr int kMaxKeys = (SG == 8 && WORDS > kPackWordsSmall) ? CTAG_KEYS_LARGE : CTAG_KEYS_SMALL;  // (4K, measured: 8 keys 5.47 ms, 12: 4.46, 16: 4.08, 24: 4.16 per 1024 frames)
        uint32_t mykey[kMaxKeys];
        bool many = false;
        {
            const int32_t* ptile = P.pool_tile + (size_t)frame * g.pool_cap;
            const int32_t* mnext = P.member_next + (size_t)frame * g.pool_cap;
            int e = cd.root;
            bool first_e = true;
#pragma unroll
            for (int k = 0; k < kMaxKeys; k++) {
                mykey[k] = 0xffffffffu;
                if (e >= 0) {
                    const int t = ptile[e];
                    mykey[k] = ((uint32_t)t << 16) | (uint32_t)(e - tbase[t] + 1);
                    e = first_e ? P.member_head[(size_t)frame * g.pool_cap + e] : mnext[e];
                    first_e = false;
                }
            }
            many = e >= 0;  // more members than keys: fall back to the gather test
        }
        // Whole-wave components cross many CCL tiles and carry many labels (a long boundary: more members than keys, several labels
        // per tile), which would put every lane on the gather test.  They get a membership table instead: for every CCL tile the box
        // touches, a byte per label 1..128 "label l of this tile belongs to my component", filled by the whole wave at once (lane l
        // asks root_of of pool entry tile_base + l - 1; a 1 for a label the tile does not have is harmless, no pixel carries
        // it).  Entry 0 (background) is 0, entry 129 -- where larger labels are clamped to -- says "ask root_of": labels above 128
        // exist only in tiles that took the second CCL pass.  One more table, all "ask", serves boxes over more than 96 tiles.
        constexpr int kMemberTiles = PHASE == 3 ? 24 : 96, kLutPitch = 132;
        __shared__ uint8_t s_lut[SG == 64 ? (kMemberTiles + 1) * kLutPitch : 4];
        const int mt_x0 = x_min / kTileW, mt_y0 = y_min / kTileH;
        const int mt_nx = (x_min + w - 1) / kTileW - mt_x0 + 1, mt_ny = (y_min + h - 1) / kTileH - mt_y0 + 1;
        const bool bitmap = SG == 64 && mt_nx * mt_ny <= kMemberTiles;
        if constexpr (SG == 64) {
            if (bitmap) {
                const int nt = mt_nx * mt_ny;
                for (int t0 = 0; t0 < nt; t0 += 4) {  // four tiles per step: their loads are in flight together
                    int base[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int t = min(t0 + u, nt - 1);
                        base[u] = tbase[(mt_y0 + t / mt_nx) * g.tiles_x + mt_x0 + t % mt_nx];
                    }
                    int ra[4], rb[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int ea = min(base[u] + lane, g.pool_cap - 1), eb = min(base[u] + 64 + lane, g.pool_cap - 1);
                        ra[u] = rootof[ea];
                        rb[u] = rootof[eb];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (t0 + u < nt) {
                            uint8_t* T = s_lut + (t0 + u) * kLutPitch;
                            T[1 + lane] = ra[u] == cd.root ? 1 : 0;
                            T[65 + lane] = rb[u] == cd.root ? 1 : 0;
                            if (lane == 0) {
                                T[0] = 0;
                                T[129] = 2;
                            }
                        }
                    }
                }
            } else {
                uint8_t* T = s_lut + kMemberTiles * kLutPitch;
                for (int i = lane; i < 130; i += 64) T[i] = i ? 2 : 0;
            }
        }
        // ---- P1: silhouette first-hit arrays (corner_detector.cpp:184-232).  Each lane reads 8 labels with one
        // 16-byte load (64 columns per sub-group step, starting at a 16-byte aligned column), remembers the last
        // (tile-local label, tile) -> "is my component" decision so the two dependent gathers are rare, and has
        // the next row's load in flight while it digests the current one.
        for (int x = sl; x < w + 2; x += SG) tb[x] = 0u;
        for (int y = sl; y < h; y += SG) {
            lef[y] = 0xffffffffu;
            rig[y] = 0u;
        }
        SG_SYNC();
        constexpr int kChunk = 8 * SG;  // columns a sub-group covers with one 16-byte load per lane
        if constexpr (SG == 64) {
            // Whole-wave scan: 512 columns per pass, eight label rows requested together (unconditional loads from clamped
            // addresses: bits of lanes and rows outside the box are masked), membership from the bitmap, and the row extents from
            // one ballot per row -- the row's first and last foreground lanes each send ONE no-return LDS atomic (64 lanes'
            // atomics on one address are served one after the other).  Nothing in the loop waits for memory but the row batch.
            const int x_end = x_min + w;
            stamp(5);
            // kGather = false: membership from the table alone, so nothing but the row batches touches global memory, the compiler can
            // count the loads in flight and the next batch streams in under the current one; a pixel the table does not cover
            // flags the component, which is then rescanned by the kGather = true build (root_of asked per such pixel).
            bool redo = false;
            auto wave_scan = [&](auto gather_tag) {
                constexpr bool kGather = decltype(gather_tag)::value;
                for (int xa = x_min & ~7; xa < x_end; xa += kChunk) {
                    const int gxf = xa + 8 * lane;
                    unsigned valid = 0;
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        if (gxf + q >= x_min && gxf + q < x_end) valid |= 1u << q;
                    const int col = valid ? gxf : (x_min & ~7);
                    const int tcol = col / kTileW;  // eight 16-byte aligned columns share a CCL tile
                    const int xl0 = gxf - x_min;
                    uint32_t top[4], bot[4], seen[4];
                    const uint8_t* lut = s_lut + kMemberTiles * kLutPitch;  // membership table of this lane's tile
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        top[d] = bot[d] = 0xffffffffu;
                        seen[d] = 0u;
                    }
                    int cur_ty = -1;
                    auto load_batch = [&](int y0, uint4 (&rr)[8]) {
#pragma unroll
                        for (int u = 0; u < 8; u++) rr[u] = *reinterpret_cast<const uint4*>(limg + (size_t)(y_min + min(y0 + u, h - 1)) * g.lp + col);
                    };
                    auto digest_batch = [&](int y0, const uint4 (&rr)[8]) {
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const int y = y0 + u;
                            if (y >= h) break;  // uniform
                            const int ty = (y_min + y) / kTileH;
                            if (ty != cur_ty) {
                                cur_ty = ty;
                                if (bitmap) lut = s_lut + ((ty - mt_y0) * mt_nx + (tcol - mt_x0)) * kLutPitch;
                            }
                            const uint32_t wv[4] = {rr[u].x, rr[u].y, rr[u].z, rr[u].w};
                            unsigned bits = 0;
                            unsigned lab[8], mv[8];
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                lab[q] = (wv[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                                mv[q] = lut[min(lab[q], 129u)];  // the eight table bytes are requested together
                            }
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                bool in = mv[q] == 1u;
                                const bool ask = mv[q] == 2u && lab[q] < 0x8000u && ((valid >> q) & 1u);  // rare: a second-pass tile's label, or a box over > 96 tiles
                                if constexpr (kGather) {
                                    if (ask) in = rootof[tbase[ty * g.tiles_x + tcol] + (int)lab[q] - 1] == cd.root;
                                } else {
                                    redo = redo || ask;
                                }
                                bits |= (in ? 1u : 0u) << q;
                            }
                            bits &= valid;
                            const unsigned long long rowm = __ballot(bits != 0u);
                            if (rowm) {
                                if (lane == __builtin_ctzll(rowm)) atomicMin(&lef[y], (unsigned)(xl0 + __ffs(bits) - 1));
                                if (lane == 63 - __builtin_clzll(rowm)) atomicMax(&rig[y], (unsigned)(xl0 + 32 - __clz(bits)));
                            }
                            if (bits) {
                                const uint32_t ypk = (uint32_t)y * 0x10001u;
#pragma unroll
                                for (int d = 0; d < 4; d++) {
                                    const uint32_t m = ((bits >> (2 * d)) & 1u) * 0xffffu + ((bits >> (2 * d + 1)) & 1u) * 0xffff0000u;
                                    bot[d] = (bot[d] & ~m) | (ypk & m);
                                    const uint32_t nm = m & ~seen[d];
                                    top[d] = (top[d] & ~nm) | (ypk & nm);
                                    seen[d] |= m;
                                }
                            }
                        }
                    };
                    uint4 ra[8], rb[8];
                    load_batch(0, ra);
                    for (int y0 = 0; y0 < h; y0 += 16) {
                        load_batch(y0 + 8, rb);
                        digest_batch(y0, ra);
                        load_batch(y0 + 16, ra);
                        digest_batch(y0 + 8, rb);
                    }
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        if ((valid >> q) & 1u) {  // every column of a component's bounding box holds a pixel
                            const uint32_t t = (top[q >> 1] >> (16 * (q & 1))) & 0xffffu, bb = (bot[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                            tb[xl0 + q + 1] = t == 0xffffu ? 0u : ((t + 2) | ((bb + 2) << 16));
                        }
                    }
                }
            };
            // Boxes of up to 256 columns (every component of a 1080p frame's markers): the wave's lanes are (row, column group) pairs --
            // G column groups of 8 pixels across, 64 / G rows down -- instead of 64 column groups on ONE row, of which a 60-column
            // component used 8: a batch of 64 / G rows costs one digest (eight table bytes, the bit tricks, a few cross-lane
            // reductions) instead of 64 / G of them, and the row extents are plain stores (one pass covers the box: no atomics).
            auto wave_scan_narrow = [&](auto gather_tag, auto groups_tag) {
                constexpr bool kGather = decltype(gather_tag)::value;
                constexpr int G = decltype(groups_tag)::value, R = 64 / G;  // column groups across, rows per batch
                const int xa0 = x_min & ~7;
                const int cg = lane & (G - 1), rr = lane / G;
                const int gxf = xa0 + 8 * cg;
                unsigned valid = 0;
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (gxf + q >= x_min && gxf + q < x_end) valid |= 1u << q;
                const int col = valid ? gxf : xa0;
                const int tcol = col / kTileW;
                const int xl0 = gxf - x_min;
                uint32_t top[4], botp[4], seen[4];  // packed halfwords per column: first row (0xffff none), last row + 1 (0 none)
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    top[d] = 0xffffffffu;
                    botp[d] = 0u;
                    seen[d] = 0u;
                }
                auto load_row = [&](int y) { return *reinterpret_cast<const uint4*>(limg + (size_t)(y_min + min(y, h - 1)) * g.lp + col); };
                auto digest = [&](int y, const uint4& v) {  // this lane's row y of the batch (y >= h: nothing)
                    unsigned bits = 0;
                    if (y < h) {
                        const int ty = (y_min + y) / kTileH;
                        const uint8_t* lut = bitmap ? s_lut + ((ty - mt_y0) * mt_nx + (tcol - mt_x0)) * kLutPitch : s_lut + kMemberTiles * kLutPitch;
                        const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
                        unsigned lab[8], mv[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            lab[q] = (wv[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                            mv[q] = lut[min(lab[q], 129u)];
                        }
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            bool in = mv[q] == 1u;
                            const bool ask = mv[q] == 2u && lab[q] < 0x8000u && ((valid >> q) & 1u);
                            if constexpr (kGather) {
                                if (ask) in = rootof[tbase[ty * g.tiles_x + tcol] + (int)lab[q] - 1] == cd.root;
                            } else {
                                redo = redo || ask;
                            }
                            bits |= (in ? 1u : 0u) << q;
                        }
                        bits &= valid;
                    }
                    // row extents: first / last foreground column over the row's G lanes
                    unsigned lo = bits ? (unsigned)(xl0 + __ffs(bits) - 1) : 0xffffffffu, hi = bits ? (unsigned)(xl0 + 32 - __clz(bits)) : 0u;
                    {   // over the row's G consecutive lanes (G >= 8): DPP permutes inside a row of 16 lanes, the crossbar only across rows
                        auto mm = [&](int olo, int ohi) {
                            lo = min(lo, (unsigned)olo);
                            hi = max(hi, (unsigned)ohi);
                        };
                        mm(dpp_mov<kDppXor1>((int)lo), dpp_mov<kDppXor1>((int)hi));
                        mm(dpp_mov<kDppXor2>((int)lo), dpp_mov<kDppXor2>((int)hi));
                        mm(dpp_mov<kDppHalfMirror>((int)lo), dpp_mov<kDppHalfMirror>((int)hi));
                        if constexpr (G >= 16) mm(dpp_mov<kDppRowMirror>((int)lo), dpp_mov<kDppRowMirror>((int)hi));
                        if constexpr (G >= 32) mm(__shfl_xor((int)lo, 16), __shfl_xor((int)hi, 16));
                    }
                    if (cg == 0 && y < h) {
                        lef[y] = lo;
                        rig[y] = hi;
                    }
                    if (bits) {
                        const uint32_t ypk = (uint32_t)y * 0x10001u, ypk1 = (uint32_t)(y + 1) * 0x10001u;
#pragma unroll
                        for (int d = 0; d < 4; d++) {
                            const uint32_t m = ((bits >> (2 * d)) & 1u) * 0xffffu + ((bits >> (2 * d + 1)) & 1u) * 0xffff0000u;
                            botp[d] = (botp[d] & ~m) | (ypk1 & m);  // this lane's rows ascend: the last one wins
                            const uint32_t nm = m & ~seen[d];
                            top[d] = (top[d] & ~nm) | (ypk & nm);
                            seen[d] |= m;
                        }
                    }
                };
                uint4 va = load_row(rr), vb;
                for (int y0 = 0; y0 < h; y0 += 2 * R) {  // two batches in flight
                    vb = load_row(y0 + R + rr);
                    digest(y0 + rr, va);
                    va = load_row(y0 + 2 * R + rr);
                    digest(y0 + R + rr, vb);
                }
                // columns: first row = min, last row + 1 = max over the R lanes that share the column group (packed halfwords)
#pragma unroll
                for (int dd = G; dd < 64; dd <<= 1) {
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const uint32_t ot = (uint32_t)__shfl_xor((int)top[d], dd), ob = (uint32_t)__shfl_xor((int)botp[d], dd);
                        top[d] = min(top[d] & 0xffffu, ot & 0xffffu) | (min(top[d] >> 16, ot >> 16) << 16);
                        botp[d] = max(botp[d] & 0xffffu, ob & 0xffffu) | (max(botp[d] >> 16, ob >> 16) << 16);
                    }
                }
                if (rr == 0) {
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        if ((valid >> q) & 1u) {  // every column of a component's bounding box holds a pixel
                            const uint32_t t = (top[q >> 1] >> (16 * (q & 1))) & 0xffffu, bp = (botp[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                            tb[xl0 + q + 1] = t == 0xffffu ? 0u : ((t + 2) | ((bp - 1 + 2) << 16));
                        }
                    }
                }
            };
            const int span8 = (x_end - (x_min & ~7) + 7) >> 3;  // 8-column groups from the aligned start to the box's end
            auto scan = [&](auto gather_tag) {
                if (span8 <= 8) wave_scan_narrow(gather_tag, std::integral_constant<int, 8>{});
                else if (span8 <= 16) wave_scan_narrow(gather_tag, std::integral_constant<int, 16>{});
                else if (span8 <= 32) wave_scan_narrow(gather_tag, std::integral_constant<int, 32>{});
                else wave_scan(gather_tag);
            };
            if (bitmap) scan(std::false_type{});
            if (!bitmap || __ballot(redo)) {
                if (bitmap) {  // start over
                    SG_SYNC();
                    for (int x = sl; x < w + 2; x += SG) tb[x] = 0u;
                    for (int y = sl; y < h; y += SG) {
                        lef[y] = 0xffffffffu;
                        rig[y] = 0u;
                    }
                    SG_SYNC();
                }
                scan(std::true_type{});
            }
        } else
        for (int xa = x_min & ~7; xa < x_min + w; xa += 2 * kChunk) {  // one pass per two chunks of the box
            const int x_end = x_min + w;  // exclusive
            const int gxf = xa + 8 * sl;  // this lane's 8 columns of chunk 0; chunk 1 is kChunk columns further
            // Eight 16-byte aligned columns never straddle a 320-column tile boundary, so one lane's pixels of a chunk
            // share a CCL tile; the component's label(s) in that tile are looked up once per tile row (every 30 rows).
            const int tcol0 = gxf / kTileW, tcol1 = (gxf + kChunk) / kTileW;
            unsigned valid0 = 0, valid1 = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (gxf + q >= x_min && gxf + q < x_end) valid0 |= 1u << q;
                if (gxf + kChunk + q >= x_min && gxf + kChunk + q < x_end) valid1 |= 1u << q;
            }
            const bool ld_ok = valid0 != 0, ld_ok1 = valid1 != 0;
            bool over0 = false, over1 = false;  // more than two labels of this component in one tile: generic test
            int cur_trow = -1;
            uint32_t LA0 = 0, LB0 = 0, LA1 = 0, LB1 = 0;  // the two tiles' labels of the component as (label | label << 16)
            uint32_t top0[4], bot0[4], seen0[4], top1[4], bot1[4], seen1[4];  // packed halfwords: 8 columns per chunk
#pragma unroll
            for (int d = 0; d < 4; d++) {
                top0[d] = bot0[d] = top1[d] = bot1[d] = 0xffffffffu;
                seen0[d] = seen1[d] = 0u;
            }
            stamp(5);
            auto labels_of_tile = [&](int tile, unsigned& la, unsigned& lb, bool& over) {
                la = 0xffffffffu;
                lb = 0xffffffffu;
                over = many;
                int found = 0;
#pragma unroll
                for (int k = 0; k < kMaxKeys; k++) {
                    if ((int)(mykey[k] >> 16) == tile && mykey[k] != 0xffffffffu) {
                        if (found == 0) la = mykey[k] & 0xffffu;
                        else if (found == 1) lb = mykey[k] & 0xffffu;
                        else over = true;
                        found++;
                    }
                }
                if (found == 1) lb = la;
            };
            // generic (rare) membership test of one pixel
            auto slow_fg = [&](unsigned l, int tile) {
                if (!many) {
                    const uint32_t key = ((uint32_t)tile << 16) | l;
                    bool r = false;
                    for (int k = 0; k < kMaxKeys; k++) r = r || (key == mykey[k]);
                    return r;
                }
                return l < 0x8000u && rootof[tbase[tile] + (int)l - 1] == cd.root;  // bit 15: an unpublished speck, never a candidate
            };
            auto fg_of = [&](const uint4& v, unsigned la, unsigned lb, bool over, int tile, unsigned valid) {
                const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
                unsigned bits = 0;
                if (!over) {
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const unsigned l0 = wv[d] & 0xffffu, l1 = wv[d] >> 16;
                        bits |= ((l0 == la || l0 == lb) ? 1u : 0u) << (2 * d);
                        bits |= ((l1 == la || l1 == lb) ? 1u : 0u) << (2 * d + 1);
                    }
                } else {
                    for (int q = 0; q < 8; q++) {
                        const unsigned l = (wv[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                        if (l && slow_fg(l, tile)) bits |= 1u << q;
                    }
                }
                return bits & valid;
            };
            // masks of a row's foreground columns: M[d] = 0xffff per label half of word d that is foreground (what note() folds into the
            // column arrays) -- from the bit form, for the paths that produce bits
            auto masks_of_bits = [&](unsigned bits, uint32_t (&M)[4]) {
#pragma unroll
                for (int d = 0; d < 4; d++) M[d] = ((bits >> (2 * d)) & 1u) * 0xffffu + ((bits >> (2 * d + 1)) & 1u) * 0xffff0000u;
            };
            // the common case (at most two labels of the component in the tile) two labels at a time: w ^ (label | label << 16) has a zero
            // half where a label matches, v_pk_min_u16 against 1 turns "non-zero" into 1, and one 24-bit multiply widens the match flags to
            // half-word masks -- a third of the instructions of eight compare pairs + selects, then eight bit extracts to rebuild masks
            // (No test against the box's columns: the lane's eight columns lie in one tile, and a pixel of that tile carrying one of the
            // component's labels IS a pixel of the component, hence inside its box.  A tile without a label of the component has
            // LA2 = LB2 = 0xffffffff: label 0xffff does not occur -- specks are 0x8000 | id with id < 4864.)
            auto fg_masks = [&](const uint4& v, uint32_t LA2, uint32_t LB2, uint32_t (&M)[4]) -> unsigned {
                const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
                uint32_t f[4];
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint32_t na = q_pk_min_u16(wv[d] ^ LA2, 0x00010001u), nb = q_pk_min_u16(wv[d] ^ LB2, 0x00010001u);  // 1 per half that differs
                    f[d] = (na & nb) ^ 0x00010001u;  // 1 per half that matches a label
                    M[d] = __umul24(f[d], 0xffffu);
                }
                const uint32_t t = f[0] | (f[1] << 2) | (f[2] << 4) | (f[3] << 6);  // even columns in bits 0, 2, 4, 6; odd ones 16 bits up
                return (t | (t >> 15)) & 0xffu;
            };
            auto note = [&](unsigned bits, const uint32_t (&M)[4], int y, uint32_t* tp, uint32_t* bt, uint32_t* sn, int xl0) {
                if constexpr (SG == 64) {
                    // row extents of a whole-wave component: the lanes are in column order, so the row's first and last foreground
                    // lanes come from one ballot and lane 0 updates the row's words
                    const unsigned long long bm = __ballot(bits != 0u);
                    if (bm) {
                        const int lo = xl0 + __ffs(bits) - 1, hi = xl0 + 32 - __clz(bits);
                        const int first = __shfl(lo, __builtin_ctzll(bm)), last1 = __shfl(hi, 63 - __builtin_clzll(bm));
                        if (sl == 0) {
                            lef[y] = min(lef[y], (uint32_t)first);
                            rig[y] = max(rig[y], (uint32_t)last1);
                        }
                    }
                }
                if (!bits) return;
                const uint32_t ypk = (uint32_t)y * 0x10001u;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint32_t m = M[d];
                    bt[d] = (bt[d] & ~m) | (ypk & m);
                    const uint32_t nm = m & ~sn[d];
                    tp[d] = (tp[d] & ~nm) | (ypk & nm);
                    sn[d] |= m;
                }
                if constexpr (SG != 64) {
                    // row extents by LDS atomics (no result needed, so no latency is exposed)
                    atomicMin(&lef[y], (unsigned)(xl0 + __ffs(bits) - 1));
                    atomicMax(&rig[y], (unsigned)(xl0 + 32 - __clz(bits)));
                }
            };
#ifndef CTAG_SCAN_UNCOND
#define CTAG_SCAN_UNCOND 1
#endif
            // Unconditional loads from clamped addresses, the lanes / rows outside the box zeroed by a select: a load inside a branch makes the
            // compiler wait for EVERY load in flight where the branch rejoins, and the rows "in flight" arrived one at a time
            // (Round 5, from the ISA: `ld_ok ? v : 0` behind the load was enough for the compiler to put the load back INSIDE a branch on ld_ok -- with s_waitcnt
            // vmcnt(0) in front of it: a row was awaited before the next was requested, and the scan, 48 % of this kernel's phase clocks, ran at the latency of one
            // L2 round trip per row.  Now nothing selects on ld_ok behind the load, an empty asm at the row's consumption reads a word of it -- so the load stays
            // above it, outside every branch --, and a lane outside the box is neutralised where its tile's labels are looked up: it matches no label.)
            const int gx0 = ld_ok ? gxf : (x_min & ~7), gx1 = ld_ok1 ? gxf + kChunk : (x_min & ~7);
            auto touch = [](const uint4& v) {  // an empty asm that reads a word of the row: what it reads was loaded ABOVE it, unconditionally
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" ::"v"(v.x));
#endif
            };
            auto load_row = [&](int y) {
                if (CTAG_SCAN_UNCOND) {
                    return *reinterpret_cast<const uint4*>(limg + (size_t)(y_min + min(y, h - 1)) * g.lp + gx0);
                }
                uint4 r = make_uint4(0, 0, 0, 0);
                if (ld_ok && y < h) r = *reinterpret_cast<const uint4*>(limg + (size_t)(y_min + y) * g.lp + gxf);
                return r;
            };
            auto load_row1 = [&](int y) {
                if (CTAG_SCAN_UNCOND) {
                    return *reinterpret_cast<const uint4*>(limg + (size_t)(y_min + min(y, h - 1)) * g.lp + gx1);
                }
                uint4 r = make_uint4(0, 0, 0, 0);
                if (ld_ok1 && y < h) r = *reinterpret_cast<const uint4*>(limg + (size_t)(y_min + y) * g.lp + gxf + kChunk);
                return r;
            };
            auto process_row = [&](const uint4& v, const uint4& v1, int y) {
                if (CTAG_SCAN_UNCOND) {
                    touch(v);
                    touch(v1);
                }
                const int trow = ((y_min + y) / kTileH) * g.tiles_x;
                if (trow != cur_trow) {
                    cur_trow = trow;
                    // the tiles' labels in the packed form (label | label << 16) fg_masks takes; 0xffffffff (no label) stays 0xffffffff
                    unsigned la, lb;
                    labels_of_tile(trow + tcol0, la, lb, over0);
                    if (CTAG_SCAN_UNCOND && !ld_ok) la = lb = 0xffffffffu, over0 = false;  // a lane whose columns lie outside the box loads the box's first columns: it matches nothing
                    LA0 = (la & 0xffffu) * 0x10001u, LB0 = (lb & 0xffffu) * 0x10001u;
                    LA1 = LB1 = 0xffffffffu, over1 = false;
                    if (SG == 64 || !CTAG_SCAN_UNCOND || ld_ok1) {  // (most boxes are narrower than 64 columns: no second chunk, no second look-up)
                        labels_of_tile(trow + tcol1, la, lb, over1);
                        LA1 = (la & 0xffffu) * 0x10001u, LB1 = (lb & 0xffffu) * 0x10001u;
                    }
                }
                auto one = [&](const uint4& w, uint32_t LA2, uint32_t LB2, bool over, int tile, unsigned valid, uint32_t* tp, uint32_t* bt, uint32_t* sn, int xl0) {
                    uint32_t M[4];
                    unsigned bits;
                    if (!over) {
                        bits = fg_masks(w, LA2, LB2, M);
                    } else {
                        bits = fg_of(w, 0u, 0u, over, tile, valid);
                        masks_of_bits(bits, M);
                    }
                    note(bits, M, y, tp, bt, sn, xl0);
                };
                one(v, LA0, LB0, over0, trow + tcol0, valid0, top0, bot0, seen0, gxf - x_min);
                if (SG == 64 || ld_ok1) one(v1, LA1, LB1, over1, trow + tcol1, valid1, top1, bot1, seen1, gxf + kChunk - x_min);  // whole wave: note() holds a ballot
            };
            // label rows in flight per lane (the scan is bound by latency, not by bytes): 4 in the large configuration, 2 in the
            // small one, whose 128-register budget the eight row registers of the deeper pipeline would spill
            if constexpr (WAVES >= 4 && !(PHASE == 1 && CTAG_SCAN_ROWS4)) {
                uint4 r0 = load_row(0), r1 = load_row(1);
                uint4 s0 = load_row1(0), s1 = load_row1(1);
                for (int y = 0; y < h; y += 2) {
                    process_row(r0, s0, y);
                    r0 = load_row(y + 2);
                    s0 = load_row1(y + 2);
                    if (y + 1 < h) process_row(r1, s1, y + 1);
                    r1 = load_row(y + 3);
                    s1 = load_row1(y + 3);
                }
            } else {
            uint4 r0 = load_row(0), r1 = load_row(1), r2 = load_row(2), r3 = load_row(3);
            uint4 s0 = load_row1(0), s1 = load_row1(1), s2 = load_row1(2), s3 = load_row1(3);
            for (int y = 0; y < h; y += 4) {
                process_row(r0, s0, y);
                r0 = load_row(y + 4);
                s0 = load_row1(y + 4);
                if (y + 1 < h) process_row(r1, s1, y + 1);
                r1 = load_row(y + 5);
                s1 = load_row1(y + 5);
                if (y + 2 < h) process_row(r2, s2, y + 2);
                r2 = load_row(y + 6);
                s2 = load_row1(y + 6);
                if (y + 3 < h) process_row(r3, s3, y + 3);
                r3 = load_row(y + 7);
                s3 = load_row1(y + 7);
            }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int xl = gxf + q - x_min;
                if ((valid0 >> q) & 1u) {  // every column of a component's bounding box holds a pixel
                    const uint32_t t = (top0[q >> 1] >> (16 * (q & 1))) & 0xffffu, bb = (bot0[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                    tb[xl + 1] = t == 0xffffu ? 0u : ((t + 2) | ((bb + 2) << 16));
                }
                if ((valid1 >> q) & 1u) {
                    const uint32_t t = (top1[q >> 1] >> (16 * (q & 1))) & 0xffffu, bb = (bot1[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                    tb[xl + kChunk + 1] = t == 0xffffu ? 0u : ((t + 2) | ((bb + 2) << 16));
                }
            }
        }
        stamp(7);
        {
            SG_SYNC();  // row extents (LDS atomics) complete
            for (int y = sl; y < h + 2; y += SG) {
                uint32_t v = 0u;
                if (y >= 1 && y <= h) {
                    const uint32_t a = lef[y - 1], b = rig[y - 1];  // b = last + 1, 0 = empty row
                    if (b) v = (a + 2) | ((b + 1) << 16);
                }
                lr[y] = v;  // lr aliases nothing that is still live: lef/rig sit in bufA/bufB
            }
        }
        SG_SYNC();
        }  // the scan
        if constexpr (PHASE == 3) {
            // the component's cluster space (what the packed builds reserve after their traversal) and, in it for now, the silhouette:
            // w + h + 4 <= C + 64 words
            if (sl == 0) p0 = atomicAdd(&P.clp_used[frame], C + 64);
            p0 = __shfl(p0, lane0);
            if ((uint32_t)(p0 + C + 64) > P.cl_cap) {
                if (sl == 0) {
                    atomicOr(&P.frame_flags[frame], CTAG_FLAG_POOL_OVERFLOW);
                    aux->line0 = -1;
                    aux->acx = 0.f;
                    aux->acy = 0.f;
                    aux->n_boundary = 0;
                }
                continue;
            }
            uint32_t* slot = P.cl_pool + (size_t)frame * P.cl_cap + p0;
            for (int x = sl; x < w + 2; x += SG) slot[x] = tb[x];
            for (int y = sl; y < h + 2; y += SG) slot[w + 2 + y] = lr[y];
            if (sl == 0) {
                aux->line0 = p0;
                aux->acx = 0.f;
                aux->acy = 0.f;
                aux->n_boundary = 0;
            }
            continue;
        }
        if constexpr (PHASE != 2) {
        if constexpr (PRESCAN) {
            p0 = aux->line0;  // the scan-only kernel left the component's slot here (-1: the frame's cluster pool is exhausted, flags and aux are final)
            if (p0 < 0) continue;
            const uint32_t* slot = P.cl_pool + (size_t)frame * P.cl_cap + p0;
            for (int x = sl; x < w + 2; x += SG) tb[x] = slot[x];
            for (int y = sl; y < h + 2; y += SG) lr[y] = slot[w + 2 + y];
            SG_SYNC();
        }
        // distinct silhouette pixels: two per column (one where top == bottom) plus the row ends that head no column list.  The
        // traversal below visits every listed pixel at most once, so once it has appended this many points nothing is left to
        // find and the rest of its work -- unwinding a stack as deep as the boundary is long, a full neighbour test per frame --
        // changes nothing: it stops there.
        int ntotal = 0;
        for (int x = sl; x < w; x += SG) {
            const uint32_t t = tb[x + 1];
            if (t) ntotal += 1 + ((t & 0xffffu) != (t >> 16) ? 1 : 0);
        }
        for (int y = sl; y < h; y += SG) {
            const uint32_t v = lr[y + 1];
            if (v) {
                const uint32_t ky = (uint32_t)(y + 2), xl = (v & 0xffffu) - 2u, xr = (v >> 16) - 2u;
                const uint32_t tl = tb[xl + 1], tr = tb[xr + 1];
                if ((tl & 0xffffu) != ky && (tl >> 16) != ky) ntotal++;
                if (xr != xl && (tr & 0xffffu) != ky && (tr >> 16) != ky) ntotal++;
            }
        }
        ntotal = sg_sum<SG>(ntotal);
        stamp(0);
        // ---- P2: ordered traversal (corner_detector.cpp:235-247, :407-418): lanes 0..7 of the sub-group test the
        // 8 neighbours (N,NE,E,SE,S,SW,W,NW); the first hit at or after the frame's resume index wins (B7)
        n = 0;
        {
            const int jd = sl & 7;  // only lanes 0..7 of the sub-group take part (sl < 8 below)
            const int dxl = (int)((0x01A9u >> (2 * jd)) & 3u) - 1, dyl = (int)((0x1A90u >> (2 * jd)) & 3u) - 1;
            int sp = 0;
            int fx = 0, fy = 0, j0 = 0;  // top-of-stack frame in registers; bufB holds the frames below it
            {
                const uint32_t c0 = tb[1];
                if (c0 == 0u) {
                    sp = -1;  // inconsistent labels (only after a flagged pool overflow): give up on this component
                } else {
                    fy = (int)(c0 & 0xffffu) - 2;  // start: top-most pixel of the left-most column (:235-244)
                    n = 1;
                }
            }
            SG_SYNC();
            if (sp >= 0 && sl == 0) {
                bufA[0] = pack_xy(x_min, fy + y_min);
                // clear the start pixel from every list it heads
                uint32_t c = tb[1];
                if ((c >> 16) == (uint32_t)(fy + 2)) c &= 0xffffu;
                c &= 0xffff0000u;
                tb[1] = c;
                uint32_t r = lr[fy + 1];
                if ((r & 0xffffu) == 2u) r &= 0xffff0000u;
                if ((r >> 16) == 2u) r &= 0xffffu;
                lr[fy + 1] = r;
            }
            SG_SYNC();
            if constexpr (SG == 64) {
                // a hit: lane `hl` of the sub-group holds the list words (c, r) of the hit pixel (nx, ny), neighbour hl & 7 of frame (fx, fy)
                auto take_hit = [&](int hl, int nx, int ny, uint32_t c, uint32_t r) {
                    const uint32_t ky = (uint32_t)(ny + 2), kx = (uint32_t)(nx + 2);
                    const int j = hl & 7;
                    const int hx = fx + ((int)((0x01A9u >> (2 * j)) & 3u) - 1), hy = fy + ((int)((0x1A90u >> (2 * j)) & 3u) - 1);
                    if (sl == hl) {  // the hit lane appends the point and clears it from the lists it heads
                        bufA[n] = pack_xy(hx + x_min, hy + y_min);  // n < ntotal <= C - 1: the loop condition
                        uint32_t c2 = c, r2 = r;
                        if ((c2 & 0xffffu) == ky) c2 &= 0xffff0000u;
                        if ((c2 >> 16) == ky) c2 &= 0xffffu;
                        if ((r2 & 0xffffu) == kx) r2 &= 0xffff0000u;
                        if ((r2 >> 16) == kx) r2 &= 0xffffu;
                        tb[nx + 1] = c2;
                        lr[ny + 1] = r2;
                        // the current frame moves to the hit pixel and resumes at j+1 (B7); it becomes the frame below the top
                        bufB[sp] = (uint32_t)hx | ((uint32_t)hy << 14) | ((uint32_t)(j + 1) << 28);  // sp <= n: a frame per appended point at most
                    }
                    SG_SYNC();
                    n++;
                    fx = hx;
                    fy = hy;
                    sp++;
                    j0 = 0;
                };
                while (sp >= 0 && n < ntotal) {
                    const int nx = fx + dxl, ny = fy + dyl;
                    const uint32_t c = tb[nx + 1], r = lr[ny + 1];
                    const uint32_t ky = (uint32_t)(ny + 2), kx = (uint32_t)(nx + 2);
                    const bool hit = sl < 8 && sl >= j0 && ((c & 0xffffu) == ky || (c >> 16) == ky || (r & 0xffffu) == kx || (r >> 16) == kx);
                    const unsigned m = (unsigned)((__ballot(hit) >> sgshift) & 0xffull);
                    if (m) {
                        take_hit(__ffs(m) - 1, nx, ny, c, r);
                        continue;
                    }
                    // frame exhausted
                    if constexpr (SG == 64) {
                        // The whole wave unwinds kUnwind frames at a time: lane 8 q + d tests neighbour d of the q-th frame below the top;
                        // the first hit in lane order is the frame nearest to the top with its first direction, exactly what single
                        // pops would find.  (Kept as two separate calls of take_hit: with the hit lane merged into one variable across
                        // the branches hipcc 7.2 produced a wrong lane index for this path.)
                        const int q = sl >> 3, fidx = sp - 1 - q;
                        const bool fv = fidx >= 0 && q < kUnwind;
                        const uint32_t f = fv ? bufB[fidx] : 0u;
                        const int qx = (int)(f & 0x3fff), qy = (int)((f >> 14) & 0x3fff), qj0 = (int)(f >> 28);
                        const int ux2 = qx + dxl, uy2 = qy + dyl;
                        const uint32_t c2 = fv ? tb[ux2 + 1] : 0u, r2 = fv ? lr[uy2 + 1] : 0u;
                        const uint32_t ky2 = (uint32_t)(uy2 + 2), kx2 = (uint32_t)(ux2 + 2);
                        const bool hit2 = fv && jd >= qj0 && ((c2 & 0xffffu) == ky2 || (c2 >> 16) == ky2 || (r2 & 0xffffu) == kx2 || (r2 >> 16) == kx2);
                        const unsigned long long bal = __ballot(hit2);
                        if (!bal) {  // all of them exhausted
                            sp -= kUnwind;
                            j0 = 8;  // the frame now on top is one of them: nothing to test
                            continue;
                        }
                        const int hl = __builtin_ctzll(bal);
                        sp -= 1 + (hl >> 3);
                        fx = __shfl(qx, hl);
                        fy = __shfl(qy, hl);
                        take_hit(hl, ux2, uy2, c2, r2);
                    } else {  // pop
                        sp--;
                        if (sp >= 0) {
                            const uint32_t f = bufB[sp];
                            fx = (int)(f & 0x3fff);
                            fy = (int)((f >> 14) & 0x3fff);
                            j0 = (int)(f >> 28);
                        }
                    }
                }
            } else {
                while (sp >= 0 && n < ntotal) {
                    const int nx = fx + dxl, ny = fy + dyl;
                    const uint32_t c = tb[nx + 1], r = lr[ny + 1];
                    const uint32_t ky = (uint32_t)(ny + 2), kx = (uint32_t)(nx + 2);
                    const bool hit = sl < 8 && sl >= j0 && ((c & 0xffffu) == ky || (c >> 16) == ky || (r & 0xffffu) == kx || (r >> 16) == kx);
                    const unsigned m = (unsigned)((__ballot(hit) >> sgshift) & 0xffull);
                    if (!m) {  // frame exhausted: pop
                        sp--;
                        if (sp >= 0) {
                            const uint32_t f = bufB[sp];
                            fx = (int)(f & 0x3fff);
                            fy = (int)((f >> 14) & 0x3fff);
                            j0 = (int)(f >> 28);
                        }
                        continue;
                    }
                    const int j = __ffs(m) - 1;
                    const int hx = fx + ((int)((0x01A9u >> (2 * j)) & 3u) - 1), hy = fy + ((int)((0x1A90u >> (2 * j)) & 3u) - 1);
                    if (sl == j) {  // the hit lane holds the list words of (hx, hy): it appends the point and clears them
                        if (n < C) bufA[n] = pack_xy(hx + x_min, hy + y_min);
                        uint32_t c2 = c, r2 = r;
                        if ((c2 & 0xffffu) == ky) c2 &= 0xffff0000u;
                        if ((c2 >> 16) == ky) c2 &= 0xffffu;
                        if ((r2 & 0xffffu) == kx) r2 &= 0xffff0000u;
                        if ((r2 >> 16) == kx) r2 &= 0xffffu;
                        tb[nx + 1] = c2;
                        lr[ny + 1] = r2;
                        // the current frame moves to the hit pixel and resumes at j+1 (B7); it becomes the frame below the top
                        if (sp <= C) bufB[sp] = (uint32_t)hx | ((uint32_t)hy << 14) | ((uint32_t)(j + 1) << 28);
                    }
                    SG_SYNC();
                    n++;
                    fx = hx;
                    fy = hy;
                    if (sp + 1 <= C) {  // (always: kept because this build's register allocation is better with it -- 5.19 vs 5.55 ms)
                        sp++;
                        j0 = 0;
                    } else {
                        j0 = j + 1;
                    }
                }
            }
            n = min(n, C);
        }
        n_boundary = n;
        if (n == 0) {
            if (sl == 0) {
                aux->line0 = -1;
                aux->acx = 0.f;
                aux->acy = 0.f;
                aux->n_boundary = 0;
            }
            continue;
        }
        SG_SYNC();
        stamp(1);
        // ---- P3: boundary centroid (:250-256), nearest point (:259-263), rotation (:264-275)
        {
            unsigned long long sx = 0, sy = 0;
            for (int k = sl; k < n; k += SG) {
                sx += (unsigned)ux(bufA[k]);
                sy += (unsigned)uy(bufA[k]);
            }
            sx = (unsigned long long)sg_sum<SG>((long long)sx);
            sy = (unsigned long long)sg_sum<SG>((long long)sy);
            acx = (float)(1.0 * (long long)sx / (double)(unsigned long long)n);
            acy = (float)(1.0 * (long long)sy / (double)(unsigned long long)n);
            float bd = 3.0e38f;
            int bi = 0x7fffffff;
            for (int k = sl; k < n; k += SG) {
                const float dx = (float)ux(bufA[k]) - acx, dy = (float)uy(bufA[k]) - acy;
                const float d = ctm::sqrt32(dx * dx + dy * dy);
                if (d < bd || (d == bd && k < bi)) {
                    bd = d;
                    bi = k;
                }
            }
            sg_best<SG>(bd, bi, [](float od, int oi, float d, int i) { return od < d || (od == d && oi < i); });  // the nearest, ties: the first
            for (int k = sl; k < n; k += SG) {
                int src = k + bi;
                if (src >= n) src -= n;
                bufB[k] = bufA[src];
            }
        }
        SG_SYNC();
        stamp(2);
        if constexpr (!PRESCAN) {
        // cluster space in the frame's pool (upper bound; the clusters are written straight to global memory)
        p0 = 0;
        if (sl == 0) p0 = atomicAdd(&P.clp_used[frame], C + 64);
        p0 = __shfl(p0, lane0);
        if ((uint32_t)(p0 + C + 64) > P.cl_cap) {
            if (sl == 0) {
                atomicOr(&P.frame_flags[frame], CTAG_FLAG_POOL_OVERFLOW);
                aux->line0 = -1;
                aux->acx = 0.f;
                aux->acy = 0.f;
                aux->n_boundary = n_boundary;
            }
            continue;
        }
        }
        }  // PHASE != 2
        if constexpr (PHASE == 1) {  // hand the rotated boundary over to the second kernel through the component's cluster-pool slot
            uint32_t* slot = P.cl_pool + (size_t)frame * P.cl_cap + p0;
            for (int k = sl; k < n; k += SG) slot[k] = bufB[k];
            if (sl == 0) {
                aux->line0 = p0;
                aux->acx = acx;
                aux->acy = acy;
                aux->n_boundary = n_boundary;
            }
            continue;
        }
        if constexpr (PHASE == 2) {
            p0 = aux->line0;  // the first kernel left the slot here (-1: it gave up on the component, flags and aux are final)
            if (p0 < 0) continue;
            n = n_boundary = aux->n_boundary;
            acx = aux->acx;
            acy = aux->acy;
            const uint32_t* slot = P.cl_pool + (size_t)frame * P.cl_cap + p0;
            for (int k = sl; k < n; k += SG) bufB[k] = slot[k];
            SG_SYNC();
        }
        uint32_t* CLg = P.cl_pool + (size_t)frame * P.cl_cap + p0;
        // ---- P4: extended RDP (:278-349), uniform control flow inside the sub-group
        uint32_t* W = bufB;
        uint32_t* Wn = bufA;
        int cnt_b = 0, init = 0, cl_off[5] = {0, 0, 0, 0, 0};
        bool failed = false;
        unsigned long long rdp_t = 0, rdp_acc[4] = {0, 0, 0, 0}, exp_dbg[3] = {0, 0, 0};  // developer aid (CTAG_QUAD_STAMPS): where the whole-wave build's RDP spends its ticks
        auto rdp_mark = [&](int slot) {
            if (SG == 64 && P.stamps) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (slot >= 0) rdp_acc[slot] += t - rdp_t;
                rdp_t = t;
            }
        };
        rdp_mark(-1);
        while (n > 0 && !failed && cnt_b < 4) {
            auto tri2 = [&](int a) {
                const uint32_t q0 = W[a], q2 = W[(a + 2) % n], q1 = W[(a + 1) % n];
                const int vx = ux(q0) + ux(q2) - 2 * ux(q1), vy = uy(q0) + uy(q2) - 2 * uy(q1);
                return vx * vx + vy * vy;
            };
            if (n <= 2) {
                failed = true;
                break;
            }
            {
                int c2 = tri2(init);  // cost > collinear_cost [1.05]  <=>  squared norm >= c2_far [2]
                while (c2 >= k_c2_far && init < n - 3) {
                    init++;
                    c2 = tri2(init);
                }
            }
            rdp_mark(0);
            int end = init + n / 2;
            if (end > n - 1) end = n - 1;
            while (true) {  // :303-330
                if (end <= init + 1) {
                    failed = true;
                    break;
                }
                const uint32_t pi = W[init], pe = W[end];
                float nl0;
                if (ux(pi) == ux(pe)) {
                    nl0 = 100;
                } else {
                    nl0 = (float)(1.0 * (uy(pe) - uy(pi)) / (ux(pe) - ux(pi)));
                }
                const float nl1 = -1;
                const float d_line = -(nl0 * ux(pi) + nl1 * uy(pi));
                const float den = ctm::sqrt32(nl0 * nl0 + 1);
                float bd = -1.f;
                int bi = -1;
                for (int it = init + 1 + sl; it < end; it += SG) {
                    const uint32_t q = W[it];
                    const float d = ctm::fabs32(nl0 * ux(q) + nl1 * uy(q) + d_line) / den;
                    const int rel = it - init - 1;
                    if (d > bd || (d == bd && rel > bi)) {
                        bd = d;
                        bi = rel;
                    }
                }
                sg_best<SG>(bd, bi, [](float od, int oi, float d, int i) { return od > d || (od == d && oi > i); });  // the farthest, ties: the last
                const int count = end - init - 1;
                if (bd > k_thr_line && count > 1) {
                    end = bi;  // literal index into dist2line
                    continue;
                }
                rdp_mark(1);
                // ---- expand_line (:125-169), speculative over the sub-group's 8 lanes
                int nl, nr;
                if constexpr (SG == 64)
                    sg_expand_line<8, 64>(W, n, init, end, sl & 7, 0, 0, k_thr_expand, (float)(g.hcols > g.hrows ? g.hcols : g.hrows), REFPRM ? 3.0e-6f : P.expand_eps, nl, nr, sl,
                                          P.stamps ? exp_dbg : nullptr);
                else
                    sg_expand_line<SG>(W, n, init, end, sl, lane0, sgshift, k_thr_expand, (float)(g.hcols > g.hrows ? g.hcols : g.hrows), REFPRM ? 3.0e-6f : P.expand_eps, nl, nr);
                rdp_mark(2);
                const int m = end - init + 1 + nl + nr;
                // the span is a circular arc [a .. b] of m distinct indices
                const int a = ((init - nl) % n + n) % n;
                const int b = (end + nr) % n;
                const bool wrap = a > b;
                const int span0 = wrap ? n - 1 : b;
                const int keep = tri2(span0) <= k_c2_near ? 1 : 0;  // cost < collinear_cost [1.05] (:337-339)
                const int offc = cl_off[cnt_b];
                for (int k = sl; k < m; k += SG) {  // cluster points in descending index order (:332-334)
                    int idx;
                    if (!wrap) {
                        idx = b - k;
                    } else {
                        idx = (k < n - a) ? (n - 1 - k) : (b - (k - (n - a)));
                    }
                    if (offc + k < C + 64) CLg[offc + k] = W[idx];
                }
                const int new_n = n - m + keep;  // erase the span except (optionally) its largest index (:341-343)
                if (!wrap) {
                    for (int k = sl; k < new_n; k += SG) {
                        int src;
                        if (k < a) src = k;
                        else if (keep && k == a) src = b;
                        else src = k - keep + m;
                        Wn[k] = W[src];
                    }
                } else {
                    const int mid = a - b - 1;
                    for (int k = sl; k < new_n; k += SG) Wn[k] = (k < mid) ? W[b + 1 + k] : W[n - 1];
                }
                SG_SYNC();
                const int back = wrap ? 0 : a;
                cl_off[cnt_b + 1] = min(offc + m, C + 64);
                cnt_b++;
                init = (back >= new_n) ? 0 : back;
                n = new_n;
                uint32_t* t = W;
                W = Wn;
                Wn = t;
                rdp_mark(3);
                break;
            }
        }
        if (SG == 64 && P.stamps && lane == 0)
        {
            for (int q = 0; q < 4; q++) atomicAdd(&P.stamps[16 + q], rdp_acc[q]);
            for (int q = 0; q < 3; q++) atomicAdd(&P.stamps[20 + q], exp_dbg[q]);
        }
        stamp(3);
        // ---- export
        bool ok = true;
        for (int j = 0; j < 4; j++) {
            const int len = (j < cnt_b) ? (cl_off[j + 1] - cl_off[j]) : 0;
            if (len < 2) ok = false;  // flag_line_number (:353-357)
        }
        if (cl_off[min(cnt_b, 4)] >= C + 64) ok = false;
        int l0 = -1;
        if (ok) {
            if (sl == 0) l0 = atomicAdd(&P.line_count[frame], 4);
            l0 = __shfl(l0, lane0);
            if (l0 + 4 > P.line_cap) {
                ok = false;
                if (sl == 0) atomicOr(&P.frame_flags[frame], CTAG_FLAG_POOL_OVERFLOW);
            }
        }
        if (ok && sl < 4) {
            LineDesc d;
            d.off = (uint32_t)(p0 + cl_off[sl]);
            d.n = cl_off[sl + 1] - cl_off[sl];
            P.line_desc[(size_t)frame * P.line_cap + l0 + sl] = d;
        }
        if (sl == 0) {
            aux->line0 = ok ? l0 : -1;
            aux->acx = ok ? acx : 0.f;
            aux->acy = ok ? acy : 0.f;
            aux->n_boundary = n_boundary;
        }
        stamp(4);
    }
}

// =====================================================================================================
// K6s: per frame, order the edge clusters by descending point count so that the three edges a Welsch wave
// fits together cost about the same (scheduling only: results do not depend on this order).
// =====================================================================================================
// An edge of at most 10 points: fitLine2D's initial sample of min(n, 10) distinct indices is ALL of its points whatever cv::RNG
// draws, so its 20 restarts are the same computation twenty times and the selection keeps restart 0 (a later equal error
// is not smaller).  Such edges -- 46 % of the edges of the synthetic batch -- are fitted once, a lane each (welsch_short).
constexpr int kWShort = 10;
constexpr int kLineSortThreads = 1024;  // a rank sort: L / threads passes of L comparisons each; one frame has ~400 edges
__global__ __launch_bounds__(kLineSortThreads) void k_line_sort(QuadPtrs P, int nframes) {
    __shared__ int s_n[kLdsLines];
    __shared__ int s_long;
    const int frame = blockIdx.x;
    if (frame >= nframes) return;
    const int L = min(P.line_count[frame], P.line_cap);
    const LineDesc* d = P.line_desc + (size_t)frame * P.line_cap;
    int32_t* out = P.line_sorted + (size_t)frame * P.line_cap;
    if (threadIdx.x == 0) s_long = 0;
    __syncthreads();
    if (L > kLdsLines) {
        // More edges than the LDS array holds (a frame of thousands of blobs): counting sort by point count, descending, counts of
        // kB - 1 and more in one bucket.  What the consumers rely on still holds: ranks [0, line_long) are the edges of more
        // than kWShort points; below the shared bucket the order is exact, and everything in it is longer than any length a
        // consumer compares with (kWPts, kLatPoints).
        constexpr int kB = 2048;
        static_assert(kB - 1 > kLatPoints && kB + 1 <= kLdsLines, "shared bucket");
        auto bucket_of = [&](int n) { return kB - 1 - min(n, kB - 1); };
        for (int b = threadIdx.x; b <= kB; b += kLineSortThreads) s_n[b] = 0;
        __syncthreads();
        int mine = 0;
        for (int i = threadIdx.x; i < L; i += kLineSortThreads) {
            const int n = d[i].n;
            atomicAdd(&s_n[bucket_of(n)], 1);
            mine += n > kWShort ? 1 : 0;
        }
        if (mine) atomicAdd(&s_long, mine);
        __syncthreads();
        if (threadIdx.x == 0) {
            P.line_long[frame] = s_long;
            int run = 0;
            for (int b = 0; b < kB; b++) {
                const int v = s_n[b];
                s_n[b] = run;
                run += v;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < L; i += kLineSortThreads) out[atomicAdd(&s_n[bucket_of(d[i].n)], 1)] = i;
        return;
    }
    int mine = 0;
    for (int i = threadIdx.x; i < L; i += kLineSortThreads) {
        const int n = d[i].n;
        s_n[i] = n;
        mine += n > kWShort ? 1 : 0;
    }
    if (mine) atomicAdd(&s_long, mine);
    __syncthreads();
    if (threadIdx.x == 0) P.line_long[frame] = s_long;  // ranks [0, s_long) of the sorted list hold the edges of more than kWShort points
    for (int i = threadIdx.x; i < L; i += kLineSortThreads) {
        const int ni = s_n[i];
        int rank = 0;
        for (int j = 0; j < L; j++) {
            const int nj = s_n[j];
            rank += (nj > ni || (nj == ni && j < i)) ? 1 : 0;
        }
        out[rank] = i;
    }
}

// =====================================================================================================
// K6b: fitLine(DIST_WELSCH) of three edges per wave: lanes 0-19 / 20-39 / 40-59 are the 20 restarts of one
// edge each.  Restarts are independent once the cv::RNG pick sequence is known, and that
// sequence depends only on the point count, so it comes from a table (or is replayed for very long edges).
// The best-restart selection replays the reference's sequential `err < min_err` / `err < EPS` logic.
// =====================================================================================================
// fitLine2D's choice among `count` restarts whose results sit at res + kk * kstride (element q at [q * rstride]): the first restart
// below EPS ends the search, else the first minimum wins
__device__ __forceinline__ void welsch_select(const QuadPtrs& P, int frame, int lid, int n, const float* res, int kstride, int rstride, int count) {
    const double EPS = n * 1.1920928955078125e-07;
    double min_err = 1.7976931348623157e308;
    float best[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < count; kk++) {
        const RestartResult r = restart_result(res + kk * kstride, rstride);
        if (r.err < min_err) {
            min_err = r.err;
            for (int q = 0; q < 4; q++) best[q] = r.line[q];
#ifndef CTAG_WELSCH_MINERR_IN_LOOP  // (in-loop placement: an error below EPS ends the restart, not the search)
            if (r.err < EPS) break;
#endif
        }
    }
    float* o = P.line_fit + ((size_t)frame * P.line_cap + lid) * 4;
    for (int q = 0; q < 4; q++) o[q] = best[q];
}

// ---- k_welsch's block: kWB = 4 waves = kWE = 12 edges x 20 restarts, one restart per lane.  A restart runs two or three IRLS iterations (45.0 % of the
// synthetic batch's restarts go on after the second, 0.1 % after the third: CTAG_QUAD_STAMPS counts them); a wave used to run three while more than half of
// its lanes had ended after two.  Here the block REGROUPS after the second: the restarts whose convergence test failed are compacted onto the first lanes of
// the block (their state is the line: 16 bytes), waves without work wait at the next barrier and cost no instruction issue, and the rest goes on -- 2 x 4 + 2
// wave-rounds per 12 edges instead of 12.
// Measured (round 5, per 4096 frames, builds side by side on one box): the per-wave kernel of round 4 (3 edges per wave, no barriers) 4.87-4.93 ms; this
// block without the regroup 5.06; with it 4.81-4.86 -- the barriers of a block cost what the regroup saves, bar 1 %.  Five-wave blocks (16 edges fill
// 320 lanes exactly) ran 5.9-6.1 ms: a CU then holds three of them (15 waves), and the kernel needs its five waves per SIMD; idle waves leaving instead of
// waiting, at six blocks per CU: 5.05; a larger weight cache at four waves per SIMD (CTAG_WCAP 20 / 24): 5.18 / 5.06.
// LDS: the weight columns [kWCap][kWT] -- which at the regroup carry the states (rows 0-4 of column r: line, item) and from then on the results (rows
// kWRes .. kWRes + 5 of column 20 e + k; the second phase's weight cache keeps to rows below kWRes) --, the staged points of the block's edges, a table of
// the edges.  The replayed pick lists of edges of kPickN2 points and more (never staged) live in the point area.
#ifndef CTAG_WELSCH_BLOCK_WAVES
#define CTAG_WELSCH_BLOCK_WAVES 4
#endif
constexpr int kWB = CTAG_WELSCH_BLOCK_WAVES;
constexpr int kWT = kWB * 64;   // 256 threads
constexpr int kWE = kWT / 20;   // 12 edges (16 lanes of the block idle)
constexpr int kWRes = 10;       // first result row
#ifdef CTAG_WELSCH_MINERR_IN_LOOP
#undef CTAG_WELSCH_REGROUP_AT
#define CTAG_WELSCH_REGROUP_AT 30  // a restart's best (err, line) would have to travel: this build does not regroup
#endif
#ifndef CTAG_WELSCH_REGROUP_AT
#define CTAG_WELSCH_REGROUP_AT 2  // iterations before the regroup (30: never)
#endif
constexpr int kWRegroupAt = CTAG_WELSCH_REGROUP_AT;
#ifndef CTAG_WPTS
#define CTAG_WPTS 128
#endif
constexpr int kWPts = CTAG_WPTS;  // points of an edge staged in LDS as float pairs; a block with a longer edge stages packed words, or reads global memory
#ifndef CTAG_WPTSU
#define CTAG_WPTSU 240
#endif
constexpr int kWPtsU = CTAG_WPTSU;  // points of an edge staged in LDS as packed words
static_assert(kWPtsU % 2 == 0 && kWPtsU < kPickN, "packed rows: an odd number of words, picks from the byte table");
static_assert(kWE * 20 <= kWT && kWCap >= kWRes + 6 && kWRes >= 5 && kWRes >= kWShort && kWPts < kPickN && kWE * kWPts * 8 >= kWT * 10 * 2 && (kWPts + 1) % 32 == 1, "k_welsch LDS layout");
struct WelschLds {
    float wc[kWCap * kWT];
    union {
    uint32_t ptu[kWE][kWPtsU + 1];  // ... or, when the block's longest edge has more than kWPts points, as the packed words (rows of an odd number of words: one bank apart)
    float2 pt[kWE][kWPts + 1];  // rows one point longer than they need to be: welsch_rounds requests one point past an edge's last -- and the lanes of a wave read the
                                // same point index of up to five (after the regroup: sixteen) edges at once, which rows of a multiple of 256 bytes would put on ONE bank pair
    };
    const uint32_t* gpts[kWE];
    int n[kWE];
    int lid[kWE];
    int cnt[kWB];
};

template <int MODE>  // 0: float pairs in LDS, 1: packed words in LDS, 2: global memory
__device__ __forceinline__ void welsch_block_run(const QuadPtrs& P, WelschLds& S, int frame) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr bool STAGED = MODE != 2;
    using PtsT = typename std::conditional<MODE == 0, LdsPts, typename std::conditional<MODE == 1, LdsPackedPts, GlobalPts>::type>::type;
    auto pts_of = [&](int e) -> PtsT {
        if constexpr (MODE == 0) return LdsPts{S.pt[e]};
        else if constexpr (MODE == 1) return LdsPackedPts{S.ptu[e]};
        else return GlobalPts{S.gpts[e]};
    };
    float line[4] = {0.f, 0.f, 0.f, 0.f};
    double err = 0;
    bool fin = true;
    int item = tid;
    for (int phase = 0; phase < 2; phase++) {
        bool run;
        if (phase == 0) {
            run = tid < kWE * 20 && S.n[min(tid / 20, kWE - 1)] > 0;
            if (P.stamps) {
                const unsigned long long act = __ballot(run);
                if (lane == 0) atomicAdd(&P.stamps[24], (unsigned long long)__popcll(act));
            }
        } else {
            // ---- regroup: restarts that have not ended move to the block's first lanes
            const unsigned long long bal = __ballot(!fin);
            if (lane == 0) S.cnt[wave] = __popcll(bal);
            __syncthreads();  // every lane is done with its weight column
            int base = 0, total = 0;
#pragma unroll
            for (int w = 0; w < kWB; w++) {
                const int c = S.cnt[w];
                base += w < wave ? c : 0;
                total += c;
            }
            if (!fin) {
                const int r = base + __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
                for (int q = 0; q < 4; q++) S.wc[q * kWT + r] = line[q];
                S.wc[4 * kWT + r] = __int_as_float(tid);
            }
            __syncthreads();
            if (P.stamps && tid == 0) {  // developer aid (CTAG_QUAD_STAMPS): restarts that went past the regroup, blocks
                atomicAdd(&P.stamps[25], (unsigned long long)total);
                atomicAdd(&P.stamps[26], 1ull);
            }
            run = tid < total;
            if (run) {
#pragma unroll
                for (int q = 0; q < 4; q++) line[q] = S.wc[q * kWT + tid];
                item = __float_as_int(S.wc[4 * kWT + tid]);
            }
        }
        if (run) {
            const int e = item / 20, k = item - e * 20;
            const int n = S.n[e];
            const double EPS = n * 1.1920928955078125e-07;
            const PtsT pts = pts_of(e);
            if (phase == 0) {
                if (STAGED || n < kPickN) {
                    welsch_first_fit(pts, TablePicks{P.pick_table + ((size_t)n * 20 + k) * 10}, min(n, 10), line);
                } else if (n < kPickN2) {
                    welsch_first_fit(pts, ListPicks{P.pick_table16 + ((size_t)(n - kPickN) * 20 + k) * 10}, 10, line);
                } else {  // replay cv::RNG up to this restart
                    uint16_t* pk = reinterpret_cast<uint16_t*>(&S.pt[0][0]) + tid * 10;
                    CvRng rng;
                    rng.state = 0xffffffffffffffffULL;
                    for (int kk = 0; kk <= k; kk++) {
                        int got = 0;
                        while (got < 10) {
                            const int j = (int)(rng.next() % (unsigned)n);
                            bool dup = false;
                            for (int q = 0; q < got; q++) dup |= (pk[q] == j);
                            if (!dup) pk[got++] = (uint16_t)j;
                        }
                    }
                    for (int a = 1; a < 10; a++) {
                        const uint16_t v = pk[a];
                        int b = a - 1;
                        while (b >= 0 && pk[b] > v) {
                            pk[b + 1] = pk[b];
                            b--;
                        }
                        pk[b + 1] = v;
                    }
                    welsch_first_fit(pts, ListPicks{pk}, 10, line);
                }
            }
            fin = welsch_rounds<kWT>(pts, n, EPS, line, err, phase == 0 ? 0 : kWRegroupAt, phase == 0 ? kWRegroupAt : 30, phase != 0, S.wc + tid, min(n, phase == 0 ? kWCap : kWRes));
            if (fin) restart_store(S.wc + kWRes * kWT + item, kWT, line, err);
        } else {
            fin = true;
        }
    }
    __syncthreads();
    if (tid < kWE && S.n[tid] > 0) welsch_select(P, frame, S.lid[tid], S.n[tid], S.wc + kWRes * kWT + tid * 20, 1, kWT, 20);
}

// the edges of sorted ranks [first, first + kWE) of the frame (ranks below L: the edges of more than kWShort points)
__device__ __forceinline__ void welsch_block(const QuadPtrs& P, WelschLds& S, int frame, int first, int L) {
    const int tid = threadIdx.x;
    if (tid < kWE) {
        int n = 0, lid = 0;
        const uint32_t* p = nullptr;
        if (first + tid < L) {
            lid = P.line_sorted[(size_t)frame * P.line_cap + first + tid];
            const LineDesc d = P.line_desc[(size_t)frame * P.line_cap + lid];
            n = d.n;
            p = P.cl_pool + (size_t)frame * P.cl_cap + d.off;
        }
        S.n[tid] = n;
        S.lid[tid] = lid;
        S.gpts[tid] = p;
    }
    __syncthreads();
    // the block's edges are sorted by descending length: the first one decides which form all of them take
    const int n0 = S.n[0];
    if (n0 <= kWPts) {
        for (int idx = tid; idx < kWE * kWPts; idx += kWT) {
            const int e = idx / kWPts, j = idx - e * kWPts;
            if (j < S.n[e]) {
                const uint32_t v = S.gpts[e][j];
                S.pt[e][j] = make_float2((float)ux(v), (float)uy(v));
            }
        }
        __syncthreads();
        welsch_block_run<0>(P, S, frame);
    } else if (n0 <= kWPtsU) {
        for (int idx = tid; idx < kWE * kWPtsU; idx += kWT) {
            const int e = idx / kWPtsU, j = idx - e * kWPtsU;
            if (j < S.n[e]) S.ptu[e][j] = S.gpts[e][j];
        }
        __syncthreads();
        welsch_block_run<1>(P, S, frame);
    } else {
        welsch_block_run<2>(P, S, frame);
    }
}

// Edges of at most kWShort points, a lane each: ONE restart on all of the edge's points (see kWShort)
__device__ __forceinline__ void welsch_short(const QuadPtrs& P, WelschLds& S, int frame, int first, int L) {
    const int tid = threadIdx.x;
    if (first + tid >= L) return;
    const int lid = P.line_sorted[(size_t)frame * P.line_cap + first + tid];
    const LineDesc d = P.line_desc[(size_t)frame * P.line_cap + lid];
    const int n = d.n;  // 2 <= n <= kWShort <= kWCap: every weight stays in LDS
    welsch_restart<kWT>(GlobalPts{P.cl_pool + (size_t)frame * P.cl_cap + d.off}, n, AllPicks{}, n, n * 1.1920928955078125e-07, S.wc + tid, kWT, S.wc + tid);
    welsch_select(P, frame, lid, n, S.wc + tid, 0, kWT, 1);
}

// ---- few-frame calls: one wave per (edge, restart) ------------------------------------------------------------------
// A restart is one lane's sequential walk in k_welsch -- two passes over the edge's points per IRLS iteration, ~180 cycles per point:
// 0.15 ms for the 450-point edges of the reference's test frame, whatever else the GPU is doing.  Only the SUMS of an iteration are
// order-dependent; the per-point work (distance, weight, the five weighted products) is not.  Here the wave's lanes compute
// the terms of 64 points at a time into LDS and then one lane per sum adds its column in point order -- the same additions in the
// same order, so the same bits -- a load and an add per term instead of the whole point.  All 20 restarts of all edges run at once;
// k_welsch, launched behind, then applies fitLine2D's selection (first restart below EPS, else the first minimum) per edge.
__device__ __forceinline__ bool welsch_lat_takes(const QuadPtrs& P, int frame, int L) {
    if (L > kLatLines) return false;
    if (L == 0) return true;
    const int longest = P.line_sorted[(size_t)frame * P.line_cap];  // sorted by descending point count
    return P.line_desc[(size_t)frame * P.line_cap + longest].n <= kLatPoints;
}

// a += src[j] for j = 0 .. n-1 in that order; eight terms are loaded ahead of the additions that wait for one another
__device__ __forceinline__ double ordered_sum(double a, const float* src, int n) {
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[j + u];
#pragma unroll
        for (int u = 0; u < 8; u++) a += v[u];
    }
    for (; j < n; j++) a += src[j];
    return a;
}

// One build for every edge the few-frame path takes (up to kLatPoints points): the edge's points and their weights stay in the lanes' registers
// (point j in lane j & 63, register j >> 6), the terms go through LDS a chunk of kLatChunk points at a time, and the lanes that add them carry their
// sums from chunk to chunk -- the additions of one sum are the same, in the same order, whatever the chunking.  (Two builds -- 256 and 1024
// points of LDS -- ran side by side on two streams before; the fork and the join cost the call more than the second build saved.)
constexpr int kLatChunk = 128;  // (256: 9.3 KB of LDS per restart, 17 per CU -- a frame's ~5000-7700 restarts took two rounds of residency; 128: one)
static_assert(kLatPoints % kLatChunk == 0 && kLatChunk % 64 == 0, "k_welsch_lat: register / chunk layout");
#ifndef CTAG_WLAT_WAVES
#define CTAG_WLAT_WAVES 5
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CTAG_WLAT_WAVES))) void k_welsch_lat(QuadPtrs P, int nframes, float* rs) {
    constexpr int NQ = kLatPoints / 64, CQ = kLatChunk / 64, NC = kLatPoints / kLatChunk;
    __shared__ float s_r[kLatChunk], s_w[kLatChunk];
    __shared__ float s_t[6][kLatChunk];
    __shared__ double s_sum[8];
    __shared__ uint16_t s_pk[10];
    // blockIdx.x (the fast dispatch index) is the restart: the twenty blocks of the longest edge -- the list is sorted by descending point count -- are
    // dispatched first, and the grid's thousands of blocks without an edge (ranks beyond the frame's list) last
    const int frame = blockIdx.z, k = blockIdx.x, lane = threadIdx.x;
    if (frame >= nframes) return;
    const int L = min(P.line_count[frame], P.line_cap);
    if (!welsch_lat_takes(P, frame, L)) return;
    const float c = 1 / 2.9846f;
    for (int rank = blockIdx.y; rank < L; rank += gridDim.y) {
        const unsigned long long dbg_t0 = P.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
        __syncthreads();  // single wave: the previous edge is done with the arrays
        const int lid = P.line_sorted[(size_t)frame * P.line_cap + rank];
        const LineDesc d = P.line_desc[(size_t)frame * P.line_cap + lid];
        const int n = d.n;  // <= kLatPoints (welsch_lat_takes)
        const uint32_t* pts = P.cl_pool + (size_t)frame * P.cl_cap + d.off;
        uint32_t pp[NQ];  // packed (x | y << 16): a register per point, unpacked where it is used (two conversions), not two
        float ww[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            pp[q] = pts[min(lane + 64 * q, n - 1)];  // unconditional, from a clamped index: all NQ loads are in flight together
            ww[q] = 0.f;
        }
        const int npick = min(n, 10);
        if (lane == 0) {  // the restart's initial sample (ascending), as welsch_three builds it
            if (n < kPickN) {
                const uint8_t* t = P.pick_table + ((size_t)n * 20 + k) * 10;
                for (int q = 0; q < npick; q++) s_pk[q] = t[q];
            } else if (n < kPickN2) {  // (replayed on the spot the twentieth restart of a 450-point edge spent 27 us here: 200 draws, a division each)
                const uint16_t* t = P.pick_table16 + ((size_t)(n - kPickN) * 20 + k) * 10;
                for (int q = 0; q < npick; q++) s_pk[q] = t[q];
            } else {
                CvRng rng;
                rng.state = 0xffffffffffffffffULL;
                for (int kk = 0; kk <= k; kk++) {
                    int got = 0;
                    while (got < npick) {
                        const int j = (int)(rng.next() % (unsigned)n);
                        bool dup = false;
                        for (int q = 0; q < got; q++) dup |= (s_pk[q] == j);
                        if (!dup) s_pk[got++] = (uint16_t)j;
                    }
                }
                for (int a = 1; a < npick; a++) {
                    const uint16_t v = s_pk[a];
                    int b = a - 1;
                    while (b >= 0 && s_pk[b] > v) {
                        s_pk[b + 1] = s_pk[b];
                        b--;
                    }
                    s_pk[b + 1] = v;
                }
            }
        }
        __syncthreads();
        float line[4], prev[4] = {0.f, 0.f, 0.f, 0.f};
        {
            const uint32_t mine = pts[s_pk[lane < npick ? lane : 0]];  // the sample's points, one per lane
            double x = 0, y = 0, x2 = 0, y2 = 0, xy = 0, w = 0;
            for (int i = 0; i < npick; i++) {
                const uint32_t p = (uint32_t)__shfl((int)mine, i);
                const float fx = (float)ux(p), fy = (float)uy(p);
                x += fx;
                y += fy;
                x2 += fx * fx;
                y2 += fy * fy;
                xy += fx * fy;
                w += 1.f;
            }
            moments_to_line(x, y, x2, y2, xy, w, line);
        }
        const double EPS = n * 1.1920928955078125e-07;
        double err = 0;
#ifdef CTAG_WELSCH_MINERR_IN_LOOP
        double berr = 1.7976931348623157e308;
        float bl[4] = {0.f, 0.f, 0.f, 0.f};
#endif
        for (int it = 0; it < 30; it++) {
            if (it > 0) {
                const float t = line[0] * prev[0] + line[1] * prev[1];
                if (t >= ctm::kAcosBelowTenMilli) {
                    const float dx = ctm::fabs32(line[2] - prev[2]);
                    const float dy = ctm::fabs32(line[3] - prev[3]);
                    const float dd = dx > dy ? dx : dy;
                    if (dd < 0.01f) break;
                }
            }
            const float lx = line[2], ly = line[3], nx = line[1], ny = -line[0];
            double acc = 0;  // lane 0: err += r, lane 1: sum_w += w, in point order
#pragma unroll
            for (int ch = 0; ch < NC; ch++) {
                if (ch * kLatChunk >= n) break;  // uniform
                if (ch > 0) __syncthreads();     // the adding lanes are done with the previous chunk
#pragma unroll
                for (int u = 0; u < CQ; u++) {
                    const int q = ch * CQ + u;
                    const float x = (float)ux(pp[q]) - lx, y = (float)uy(pp[q]) - ly;
                    const float r = ctm::fabs32(nx * x + ny * y);
                    const float wj = ctm::exp32_nonpos(-r * r * c * c);
                    ww[q] = wj;
                    s_r[64 * u + lane] = r;
                    s_w[64 * u + lane] = wj;
                }
                __syncthreads();
                if (lane < 2) acc = ordered_sum(acc, lane == 0 ? s_r : s_w, min(kLatChunk, n - ch * kLatChunk));
            }
            if (lane < 2) s_sum[lane] = acc;
            __syncthreads();
            err = s_sum[0];
            const double sum_w = s_sum[1];
#ifdef CTAG_WELSCH_MINERR_IN_LOOP
            if (err < berr) {
                berr = err;
                for (int q = 0; q < 4; q++) bl[q] = line[q];
                if (err < EPS) break;
            }
#else
            if (err < EPS) break;
#endif
            const bool weighted = ctm::fabs64(sum_w) > 1.1920928955078125e-07;
            const double inv = weighted ? 1. / sum_w : 0.;
            acc = 0;  // lanes 0..5: x, y, x2, y2, xy, w -- each a column added in point order
#pragma unroll
            for (int ch = 0; ch < NC; ch++) {
                if (ch * kLatChunk >= n) break;  // uniform
                if (ch > 0) __syncthreads();
#pragma unroll
                for (int u = 0; u < CQ; u++) {
                    const int q = ch * CQ + u, j = 64 * u + lane;
                    const float fx = (float)ux(pp[q]), fy = (float)uy(pp[q]);
                    if (weighted) {
                        const float wj = (float)(ww[q] * inv);
                        s_t[0][j] = wj * fx;
                        s_t[1][j] = wj * fy;
                        s_t[2][j] = wj * fx * fx;
                        s_t[3][j] = wj * fy * fy;
                        s_t[4][j] = wj * fx * fy;
                        s_t[5][j] = wj;
                    } else {
                        s_t[0][j] = fx;
                        s_t[1][j] = fy;
                        s_t[2][j] = fx * fx;
                        s_t[3][j] = fy * fy;
                        s_t[4][j] = fx * fy;
                        s_t[5][j] = 1.f;
                    }
                }
                __syncthreads();
                if (lane < 6) acc = ordered_sum(acc, s_t[lane], min(kLatChunk, n - ch * kLatChunk));
            }
            if (lane < 6) s_sum[2 + lane] = acc;
            __syncthreads();
            prev[0] = line[0];
            prev[1] = line[1];
            prev[2] = line[2];
            prev[3] = line[3];
            moments_to_line(s_sum[2], s_sum[3], s_sum[4], s_sum[5], s_sum[6], s_sum[7], line);  // (the next write of s_sum is behind the next chunk's barrier)
        }
#ifdef CTAG_WELSCH_MINERR_IN_LOOP
        for (int q = 0; q < 4; q++) line[q] = bl[q];
        err = berr;
#endif
        if (lane == 0) {
            float* o = rs + (((size_t)frame * kLatLines + rank) * 20 + k) * 6;
            o[0] = line[0];
            o[1] = line[1];
            o[2] = line[2];
            o[3] = line[3];
            *reinterpret_cast<double*>(o + 4) = err;
            if (P.stamps && rank < 8 && (k == 0 || k == 19)) P.stamps[16 + 8 + (k ? 1 : 0) * 4 + (rank >> 1)] = __builtin_amdgcn_s_memtime() - dbg_t0;  // developer aid: restarts 0 / 19 of ranks 0, 2, 4, 6
        }
    }
}

// fitLine2D's choice among the restarts of the edge of sorted rank `rank`: the first restart below EPS ends the search, else the first minimum wins
__device__ __forceinline__ void welsch_pick(const QuadPtrs& P, int frame, int rank, const float* rs) {
    const int lid = P.line_sorted[(size_t)frame * P.line_cap + rank];
    const int n = P.line_desc[(size_t)frame * P.line_cap + lid].n;
    const double EPS = n * 1.1920928955078125e-07;
    double min_err = 1.7976931348623157e308;
    float best[4] = {0.f, 0.f, 0.f, 0.f};
    const float* r = rs + ((size_t)frame * kLatLines + rank) * 20 * 6;
    for (int kk = 0; kk < 20; kk++) {
        const double e = *reinterpret_cast<const double*>(r + kk * 6 + 4);
        if (e < min_err) {
            min_err = e;
            for (int q = 0; q < 4; q++) best[q] = r[kk * 6 + q];
#ifndef CTAG_WELSCH_MINERR_IN_LOOP
            if (e < EPS) break;
#endif
        }
    }
    float* o = P.line_fit + ((size_t)frame * P.line_cap + lid) * 4;
    for (int q = 0; q < 4; q++) o[q] = best[q];
}

#ifndef CTAG_WELSCH_PRIO_RANKS
#define CTAG_WELSCH_PRIO_RANKS 3  // blocks of kWE edges
#endif
#ifndef CTAG_WELSCH_WAVES
#define CTAG_WELSCH_WAVES 5
#endif
__global__ __launch_bounds__(kWT) __attribute__((amdgpu_waves_per_eu(CTAG_WELSCH_WAVES, CTAG_WELSCH_WAVES))) void k_welsch(QuadPtrs P, int nframes, const float* lat_rs, int gy_long) {
    // Longest first across the WHOLE batch: blockIdx.x (the fast dispatch index) is the frame, blockIdx.y the rank of the
    // group of kWE edges in the frame's list sorted by descending point count.  The longest groups of all frames are dispatched
    // first and the kernel drains on the short ones: a long edge runs ~0.3 ms, and with the rank on
    // the fast index some of them started last and WERE the kernel's tail.  Consecutive frames fall on consecutive XCDs, so
    // the load stays balanced without the column rotation the other layout needed.
    const int frame = blockIdx.x;
    if (frame >= nframes) return;
    const int L = min(P.line_count[frame], P.line_cap);
    if (lat_rs && welsch_lat_takes(P, frame, L)) {  // few-frame call: k_welsch_lat has run the restarts of this frame's edges; pick per edge
        for (int rank = (int)blockIdx.y * kWT + (int)threadIdx.x; rank < L; rank += (int)gridDim.y * kWT) welsch_pick(P, frame, rank, lat_rs);
        return;
    }
    // the first ranks are the long edges: their waves are the kernel's critical path, so they get issue priority over the short
    // ones they share a SIMD with (s_setprio; the bulk fills the slots they leave)
    if (blockIdx.y < (unsigned)CTAG_WELSCH_PRIO_RANKS) __builtin_amdgcn_s_setprio(3);
    __shared__ WelschLds S;
    const int nlong = min(P.line_long[frame], L);  // ranks [0, nlong): edges of more than kWShort points, kWE per block x 20 restarts
    if ((int)blockIdx.y < gy_long) {
        for (int first = (int)blockIdx.y * kWE; first < nlong; first += gy_long * kWE) {
            welsch_block(P, S, frame, first, nlong);
            __syncthreads();
        }
    } else {  // ranks [nlong, L): kWT short edges per block, one restart each
        for (int first = nlong + ((int)blockIdx.y - gy_long) * kWT; first < L; first += ((int)gridDim.y - gy_long) * kWT) welsch_short(P, S, frame, first, L);
    }
}

// =====================================================================================================
// K6c: per candidate, the six pairwise intersections of its four fitted edges, angular sort and the best
// 4-subset by RAC (corner_detector.cpp:362-403, :420-463).  One thread per candidate.
// =====================================================================================================
// LANES: 1 for batches (a thread per candidate); 8 for calls of a few frames: lane q < 6 of a candidate's group computes intersection q -- its atan2 is
// what a candidate costs -- and hands it to the others; the rest (a rank sort of six, fifteen subsets) every lane repeats, lane 0 stores
template <int LANES>
__device__ __forceinline__ void quad_final_one(const QuadPtrs& P, const FrameGeom& g, int frame, int ci, int sub);
template <int LANES>
__global__ __launch_bounds__(64) void k_quad_final(QuadPtrs P, FrameGeom g, int nframes) {
    const int frame = blockIdx.y;
    if (frame >= nframes) return;
    const int nc = min(P.ncand[frame], P.cand_cap);
    constexpr int PER = 64 / LANES;  // candidates per block
    // column rotated by frame: spreads the few busy blocks over the XCDs; a block loops when a frame has more candidates than the grid has threads
    // (uniform trip count per group of LANES lanes: the group's shuffles need all of them)
    for (int ci = (int)((blockIdx.x + frame) % gridDim.x) * PER + (int)threadIdx.x / LANES; ci < nc; ci += (int)gridDim.x * PER)
        quad_final_one<LANES>(P, g, frame, ci, (int)threadIdx.x % LANES);
}
template <int LANES>
__device__ __forceinline__ void quad_final_one(const QuadPtrs& P, const FrameGeom& g, int frame, int ci, int sub) {
    const CandAux aux = P.cand_aux[(size_t)frame * P.cand_cap + ci];
    QuadOut* out = P.quads + (size_t)frame * P.cand_cap + ci;
    if (sub == 0) out->n_boundary = aux.n_boundary;
    if (aux.line0 < 0) {  // (uniform within the candidate's group)
        if (sub == 0) out->valid = 0;
        return;
    }
    const int areaPx = P.cand[(size_t)frame * P.cand_cap + ci].area;
    const float acx = aux.acx, acy = aux.acy;
    float lf[4][4];
    {
        const float* src = P.line_fit + ((size_t)frame * P.line_cap + aux.line0) * 4;
        for (int j = 0; j < 4; j++)
            for (int q = 0; q < 4; q++) lf[j][q] = src[j * 4 + q];
    }
    // The reference collects the valid pairwise intersections in (j, k) order, sorts them by angle (std::sort on <= 6 elements: an
    // insertion sort, i.e. stable) and tries the 4-subsets in lexicographic order of the sorted list.  Same thing with every index
    // a compile-time constant (arrays indexed at run time live in scratch memory: a thread spent most of its 25 us there): the
    // six slots carry a valid flag, the stable sort is a rank computation, the sorted list is built by selects.
    CornerPre raw[6];
    bool ok6[6];
    {
        int q = 0;
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int k = j + 1; k < 4; k++) {
                const float a00 = lf[j][1], a01 = -lf[j][0], a10 = lf[k][1], a11 = -lf[k][0];
                const float b0 = lf[j][1] * lf[j][2] - lf[j][0] * lf[j][3];
                const float b1 = lf[k][1] * lf[k][2] - lf[k][0] * lf[k][3];
                CornerPre c{0.f, 0.f, 0.f, 0.f};
                bool ok = false;
                if (LANES == 1 || sub == q) {
                    ok = solve2x2(a00, a01, a10, a11, b0, b1, c.x, c.y);
                    if (ok) {
                        c.dis = ctm::sqrt32((c.x - acx) * (c.x - acx) + (c.y - acy) * (c.y - acy));
                        c.ang = (float)(ctm::atan2_32(c.y - acy, c.x - acx) * 180 / 3.1415926535897932384626433832795);
                        ok = c.dis < g.hcols && c.dis < g.hrows;
                    }
                }
                raw[q] = c;
                ok6[q] = ok;
                q++;
            }
        if constexpr (LANES > 1) {  // intersection q from lane q of the group
            const int l0 = (int)threadIdx.x & ~(LANES - 1);
#pragma unroll
            for (int u = 0; u < 6; u++) {
                raw[u].x = __shfl(raw[u].x, l0 + u);
                raw[u].y = __shfl(raw[u].y, l0 + u);
                raw[u].dis = __shfl(raw[u].dis, l0 + u);
                raw[u].ang = __shfl(raw[u].ang, l0 + u);
                ok6[u] = __shfl((int)ok6[u], l0 + u) != 0;
            }
        }
    }
    int ncp = 0;
    int rank6[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        ncp += ok6[i] ? 1 : 0;
        int r = 0;
#pragma unroll
        for (int j = 0; j < 6; j++)
            if (j != i) r += (ok6[j] && (raw[j].ang < raw[i].ang || (!(raw[i].ang < raw[j].ang) && j < i))) ? 1 : 0;  // stable: ties keep collection order
        rank6[i] = ok6[i] ? r : 6;
    }
    CornerPre cp[6];
#pragma unroll
    for (int p = 0; p < 6; p++) {
        CornerPre c{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 6; i++)
            if (rank6[i] == p) c = raw[i];
        cp[p] = c;
    }
    float rac_min = P.rac;
    int best_id = -1;
    {
        int id = 0;
#pragma unroll
        for (int i0 = 0; i0 < 6; i0++)
#pragma unroll
            for (int i1 = i0 + 1; i1 < 6; i1++)
#pragma unroll
                for (int i2 = i1 + 1; i2 < 6; i2++)
#pragma unroll
                    for (int i3 = i2 + 1; i3 < 6; i3++) {
                        const int my = id++;
                        if (i3 >= ncp) continue;
                        const CornerPre &p0 = cp[i0], &p1 = cp[i1], &p2 = cp[i2], &p3 = cp[i3];
                        const float s1 = p0.x * p1.y + p1.x * p2.y + p2.x * p0.y - p0.x * p2.y - p1.x * p0.y - p2.x * p1.y;
                        const float s2 = p1.x * p2.y + p2.x * p3.y + p3.x * p1.y - p1.x * p3.y - p2.x * p1.y - p3.x * p2.y;
                        const float s3 = p2.x * p3.y + p3.x * p0.y + p0.x * p2.y - p2.x * p0.y - p3.x * p2.y - p0.x * p3.y;
                        const float s4 = p0.x * p1.y + p1.x * p3.y + p3.x * p0.y - p0.x * p3.y - p1.x * p0.y - p3.x * p1.y;
                        if (ctm::fabs32(s1) < 1 || ctm::fabs32(s2) < 1 || ctm::fabs32(s3) < 1 || ctm::fabs32(s4) < 1) continue;
                        float qa = 0;
                        qa += p0.x * p1.y - p0.y * p1.x;
                        qa += p1.x * p2.y - p1.y * p2.x;
                        qa += p2.x * p3.y - p2.y * p3.x;
                        qa += p3.x * p0.y - p3.y * p0.x;
                        qa /= 2;
                        const float rac = ctm::fabs32(ctm::fabs32(qa) - areaPx) / areaPx;
                        if (rac < rac_min) {
                            rac_min = rac;
                            best_id = my;
                        }
                    }
    }
    // the winning subset's corners, again by selects over the 15 subsets
    CornerPre bc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    {
        int id = 0;
#pragma unroll
        for (int i0 = 0; i0 < 6; i0++)
#pragma unroll
            for (int i1 = i0 + 1; i1 < 6; i1++)
#pragma unroll
                for (int i2 = i1 + 1; i2 < 6; i2++)
#pragma unroll
                    for (int i3 = i2 + 1; i3 < 6; i3++) {
                        if (id++ == best_id) {
                            bc[0] = cp[i0];
                            bc[1] = cp[i1];
                            bc[2] = cp[i2];
                            bc[3] = cp[i3];
                        }
                    }
    }
    int valid = best_id >= 0 ? 1 : 0;
    if (valid) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const CornerPre& c = bc[j];
            if (c.x < 0 || c.y < 0 || c.x > g.hcols || c.y > g.hrows) valid = 0;
        }
    }
    if (sub != 0) return;
    out->valid = valid;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        out->c[2 * j] = valid ? bc[j].x : 0.f;
        out->c[2 * j + 1] = valid ? bc[j].y : 0.f;
    }
}

hipError_t launch_quads(int nframes, const Workspace& ws, hipStream_t s, hipEvent_t* ev5, const uint8_t* mask) {
    int evi = 0;
    auto mark = [&]() {
        if (ev5) (void)hipEventRecord(ev5[evi++], s);
    };
    QuadPtrs P{ws.labels, ws.tile_base, ws.root_of, ws.ncand, ws.cand, ws.quads, ws.frame_flags,
               ws.line_count, ws.clp_used, ws.cl_pool, ws.line_desc, ws.line_sorted, ws.line_long, ws.line_fit, ws.cand_aux, ws.pick_table, ws.pick_table16, ws.pool_tile, ws.member_head, ws.member_next, ws.npacks, ws.packs, ws.pack_order, nullptr, ws.kp.thr_line, ws.kp.thr_expand, ws.kp.rac, ws.kp.c2_far, ws.kp.c2_near,
               ws.cand_cap, ws.line_cap, ws.cl_cap, ws.kp.expand_eps};
    static unsigned long long* d_stamps = nullptr;
    const bool want_stamps = getenv("CTAG_QUAD_STAMPS") != nullptr;
    if (want_stamps) {
        if (!d_stamps) (void)hipMalloc(reinterpret_cast<void**>(&d_stamps), 32 * 8);
        (void)hipMemsetAsync(d_stamps, 0, 32 * 8, s);
        P.stamps = d_stamps;
    }
    static const int pack_max_env = getenv("CTAG_PACK_MAX") ? atoi(getenv("CTAG_PACK_MAX")) : 0;
    const int pack_max = pack_max_env > 0 ? std::min(pack_max_env, kSG) : kSG;
    static const int big_env = getenv("CTAG_BIG_POINTS") ? atoi(getenv("CTAG_BIG_POINTS")) : 0;
    const bool latency = nframes <= kLatencyFrames;
    const int big_points = ws.wave_points > 0 ? ws.wave_points : big_env > 0 ? big_env : (latency ? kLatencyBigPoints : 0x7fffffff);
    const bool small_cfg = (long long)ws.g.hrows * ws.g.hcols <= 960LL * 600;  // up to 1920x1200 frames
    const int pack_words = small_cfg ? kPackWordsSmall : kPackWords;
    // batches: the silhouettes of the packed components come from a kernel of their own (PHASE 3 of k_quad_edges_packed); CTAG_PRESCAN=0 (developer aid) keeps the scan in the packed builds
    static const int prescan_env = getenv("CTAG_PRESCAN") ? atoi(getenv("CTAG_PRESCAN")) : 1;
    // (measured: 4K frames, quad_edges 4.07 -> 2.78 ms per 1024 frames; 1080p frames 3.88 -> 4.01 per 4096 -- there a component's box is ~75 x 20, the
    // packed build's eight components per wave amortise the chain of dependent loads a component costs better than a wave per component does; a
    // scan-only kernel with the packs' own 8 x 8 lanes at 4 / 5 / 6 waves per SIMD: 4.09 / 4.11 / 4.14 -- the scan is not what the small build waits for)
    const bool small_frames = (long long)ws.g.hrows * ws.g.hcols <= 960LL * 600;
    // chunks that took the fused sweep: the silhouettes come from its threshold mask (k_silhouette_mask), whatever the frame size; CTAG_MASK_SCAN=0 (developer aid, A/B) turns it off
    static const int mask_scan_env = getenv("CTAG_MASK_SCAN") ? atoi(getenv("CTAG_MASK_SCAN")) : 1;
    const bool mask_scan = mask != nullptr && !latency && mask_scan_env != 0 && (ws.g.hcols & 63) == 0;
    const bool prescan = mask_scan || (!latency && (prescan_env == 2 || (prescan_env == 1 && !small_frames)));
    P.mask = reinterpret_cast<const uint64_t*>(mask);
    P.mask_words = ws.g.hcols >> 6;
    hipLaunchKernelGGL(k_pack, dim3(nframes), dim3(nframes <= kLatencyFrames ? 1024 : 64), 0, s, P, nframes, pack_max, big_points, pack_words, mask_scan ? -(small_cfg ? kMaskScanWords : kMaskScanWordsLarge) : prescan ? kScanWords : 0);
    mark();
    // A few frames per call (the reference's one detect() per camera frame): the call is as long as its slowest component,
    // so the packs and the whole-wave components run side by side (second stream, fork/join by events)
    // ... unless every component is a whole-wave one (the default of such calls: a frame has a few hundred components and the GPU a thousand SIMDs, so
    // the stage is as long as the longest boundary either way, and without packs there is nothing to fork or join: 0.476 -> 0.46 ms on test.bmp)
    const bool all_wave = latency && big_points <= 1;
    const bool fork = latency && !all_wave && ws.aux_stream != nullptr;
    hipStream_t sb = s;
    if (fork) {
        (void)hipEventRecord(ws.ev_fork, s);
        (void)hipStreamWaitEvent(ws.aux_stream, ws.ev_fork, 0);
        sb = ws.aux_stream;
    }
    static const int pack_gx_env = getenv("CTAG_PACK_GX") ? atoi(getenv("CTAG_PACK_GX")) : 0;
    const int pack_gx = pack_gx_env > 0 ? pack_gx_env : 32;
    const bool refprm = ws.kp.thr_line == 1.8f && ws.kp.thr_expand == 1.2f && ws.kp.c2_far == 2 && ws.kp.c2_near == 1 && ws.kp.expand_eps == 3.0e-6f;
#ifndef CTAG_SCAN_WAVES
#define CTAG_SCAN_WAVES 4
#endif
    static const int scan_gx = getenv("CTAG_SCAN_GX") ? std::max(1, atoi(getenv("CTAG_SCAN_GX"))) : 48;
    static const int mscan_gx = getenv("CTAG_MSCAN_GX") ? std::max(1, atoi(getenv("CTAG_MSCAN_GX"))) : 48;  // blocks (waves) per frame of k_silhouette_mask, two components each per trip
#define CTAG_LAUNCH_PACKED(REF)                                                                                                                              \
    do {                                                                                                                                                     \
        if (prescan) {                                                                                                                                       \
            if (mask_scan)                                                                                                                                   \
                if (small_cfg) hipLaunchKernelGGL((k_silhouette_mask<4, 32, kMaskScanWords>), dim3(nframes, mscan_gx), dim3(64), 0, s, P, ws.g, nframes);     \
                else hipLaunchKernelGGL((k_silhouette_mask<8, 64, kMaskScanWordsLarge>), dim3(nframes, 2 * mscan_gx), dim3(64), 0, s, P, ws.g, nframes);      \
            else                                                                                                                                             \
                hipLaunchKernelGGL((k_quad_edges_packed<64, kScanWords, CTAG_SCAN_WAVES, false, true, 3>), dim3(nframes, scan_gx), dim3(64), 0, s, P, ws.g, nframes, 0); \
            if (small_cfg) {                                                                                                                                 \
                hipLaunchKernelGGL((k_quad_edges_packed<8, kPackWordsSmall, kPackWavesSmall, false, REF, 1, true>), dim3(nframes, pack_gx), dim3(64), 0, s, P, ws.g, nframes, 0); \
                hipLaunchKernelGGL((k_quad_edges_packed<8, kPackWordsSmall, kPackWavesSmallP2, false, REF, 2>), dim3(nframes, pack_gx), dim3(64), 0, s, P, ws.g, nframes, 0); \
            } else {                                                                                                                                         \
                hipLaunchKernelGGL((k_quad_edges_packed<8, kPackWords, CTAG_PACK_WAVES, false, REF, CTAG_PACK_SPLIT_LARGE ? 1 : 0, true>), dim3(nframes, pack_gx), dim3(64), 0, s, P, ws.g, nframes, 0); \
                if (CTAG_PACK_SPLIT_LARGE)                                                                                                                   \
                    hipLaunchKernelGGL((k_quad_edges_packed<8, kPackWords, CTAG_PACK_WAVES, false, REF, 2>), dim3(nframes, pack_gx), dim3(64), 0, s, P, ws.g, nframes, 0); \
            }                                                                                                                                                \
        } else if (small_cfg)                                                                                                                                \
        {                                                                                                                                                    \
            hipLaunchKernelGGL((k_quad_edges_packed<8, kPackWordsSmall, kPackWavesSmall, false, REF, CTAG_PACK_SPLIT ? 1 : 0>), dim3(nframes, pack_gx), dim3(64), 0, s, P, ws.g, nframes, 0); \
            if (CTAG_PACK_SPLIT)                                                                                                                             \
                hipLaunchKernelGGL((k_quad_edges_packed<8, kPackWordsSmall, kPackWavesSmallP2, false, REF, 2>), dim3(nframes, pack_gx), dim3(64), 0, s, P, ws.g, nframes, 0); \
        } else {                                                                                                                                             \
            hipLaunchKernelGGL((k_quad_edges_packed<8, kPackWords, CTAG_PACK_WAVES, false, REF, CTAG_PACK_SPLIT_LARGE ? 1 : 0>), dim3(nframes, pack_gx), dim3(64), 0, s, P, ws.g, nframes, 0); \
            if (CTAG_PACK_SPLIT_LARGE)                                                                                                                       \
                hipLaunchKernelGGL((k_quad_edges_packed<8, kPackWords, CTAG_PACK_WAVES, false, REF, 2>), dim3(nframes, pack_gx), dim3(64), 0, s, P, ws.g, nframes, 0); \
        }                                                                                                                                                    \
    } while (0)
    if (all_wave) {
    } else if (refprm) CTAG_LAUNCH_PACKED(true);
    else CTAG_LAUNCH_PACKED(false);
#undef CTAG_LAUNCH_PACKED
    mark();
    // oversize components, a wave each: working sets up to kWaveWords in a 32 KB build (5 per CU), the rest (up to the 144 KB a
    // 4K frame's longest possible boundary needs three times over) in a build that owns a CU's LDS
    static const int bcols_env = getenv("CTAG_BIG_COLS") ? atoi(getenv("CTAG_BIG_COLS")) : 0;
    const int bcols = bcols_env > 0 ? bcols_env : (latency ? 512 : 4);  // (few frames: every component a block of its own up to 512 per frame; a 2666-blob frame 3.0 -> 2.5 ms against 128)
#define CTAG_LAUNCH_WAVE(REF)                                                                                                                               \
    do {                                                                                                                                                    \
        int dev = 0;                                                                                                                                        \
        (void)hipGetDevice(&dev);                                                                                                                           \
        dev = dev < 0 || dev >= 64 ? 0 : dev;                                                                                                               \
        static bool have[64] = {false};                                                                                                                     \
        if (!have[dev]) {                                                                                                                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_quad_edges_packed<64, kWaveWordsMax, 1, true, REF>),                                  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kWaveWordsMax * 4);                                                       \
            have[dev] = true;                                                                                                                               \
        }                                                                                                                                                   \
        hipLaunchKernelGGL((k_quad_edges_packed<64, kWaveWords, 1, false, REF>), dim3(nframes, bcols), dim3(64), 0, sb, P, ws.g, nframes, 0);                \
        /* few-frame calls: the build for the longest boundaries (rarely any) behind the packs on the main stream, not behind the 32 KB build */          \
        hipLaunchKernelGGL((k_quad_edges_packed<64, kWaveWordsMax, 1, true, REF>), dim3(nframes, latency ? 8 : 2), dim3(64), kWaveWordsMax * 4,              \
                           fork ? s : sb, P, ws.g, nframes, kWaveWords);                                                                                    \
    } while (0)
    if (refprm) CTAG_LAUNCH_WAVE(true);
    else CTAG_LAUNCH_WAVE(false);
#undef CTAG_LAUNCH_WAVE
    if (fork) {
        (void)hipEventRecord(ws.ev_join, sb);
        (void)hipStreamWaitEvent(s, ws.ev_join, 0);
    }
    mark();
    hipLaunchKernelGGL(k_line_sort, dim3(nframes), dim3(kLineSortThreads), 0, s, P, nframes);
    mark();
    static const int welsch_gs = getenv("CTAG_WELSCH_GS") ? std::max(1, atoi(getenv("CTAG_WELSCH_GS"))) : 1;   // blocks per frame for the edges of <= 10 points, 320 per block
    static const int welsch_gx = getenv("CTAG_WELSCH_GX") ? std::max(1, atoi(getenv("CTAG_WELSCH_GX"))) : 18;   // (both at least 1: the grid's two row ranges each own a class of edges)   // groups of kWE edges per frame with a block of their own; a block loops when a frame has more (the synthetic frames have ~210 edges of more than 10 points: 17.8 groups)
    if (latency && ws.welsch_rs) {  // one wave per (edge, restart); frames it declines (more edges / longer edges than it holds) fall through to k_welsch
        hipLaunchKernelGGL(k_welsch_lat, dim3(20, 512, nframes), dim3(64), 0, s, P, nframes, ws.welsch_rs);
    }
    hipLaunchKernelGGL(k_welsch, dim3(nframes, welsch_gx + welsch_gs), dim3(kWT), 0, s, P, nframes, latency && ws.welsch_rs ? ws.welsch_rs : (const float*)nullptr, welsch_gx);
    mark();
    if (latency) hipLaunchKernelGGL(k_quad_final<8>, dim3(std::min(ws.cand_cap, kLdsCand) / 8, nframes), dim3(64), 0, s, P, ws.g, nframes);
    else hipLaunchKernelGGL(k_quad_final<1>, dim3(std::min(ws.cand_cap, kLdsCand) / 64, nframes), dim3(64), 0, s, P, ws.g, nframes);
    if (want_stamps) {
        unsigned long long h[32];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost);
        if (h[16] | h[17] | h[18] | h[19])
            fprintf(stderr, "[whole-wave rdp ticks] corner scan %llu split rounds %llu expand_line %llu clusters + erase %llu\n", h[16], h[17], h[18], h[19]),
            fprintf(stderr, "[whole-wave expand_line] initial sums %llu ticks, %llu rounds, %llu of them through the exact fits\n", h[20], h[21], h[22]);
        if (h[26]) fprintf(stderr, "[k_welsch] %llu restarts in %llu blocks, %llu (%.1f %%) went on after the regroup at iteration %d\n", h[24], h[26], h[25], 100.0 * h[25] / (h[24] ? h[24] : 1), kWRegroupAt);
        else if (h[24]) fprintf(stderr, "[k_welsch_lat ticks] restart 0 of the edges of rank 0 / 2 / 4 / 6: %llu %llu %llu %llu; restart 19: %llu %llu %llu %llu\n", h[24], h[25], h[26], h[27], h[28], h[29], h[30], h[31]);
        for (int b = 0; b < 16; b += 8) {
            unsigned long long tot = 0;
            h[b + 0] += h[b + 7];  // the row scan is stamped separately; it belongs to the silhouette phase
            for (int i = 0; i < 5; i++) tot += h[b + i];
            if (!tot) continue;
            fprintf(stderr, "[%s] row scan %llu of the silhouette phase's %llu ticks\n", b ? "whole-wave" : "packed", h[b + 7], h[b + 0]);
            fprintf(stderr, "[%s cycles] header %.1f%% keys+init %.1f%% | silhouette %.1f%% traversal %.1f%% centroid/rotate %.1f%% rdp %.1f%% export %.1f%% (total %llu)\n",
                    b ? "whole-wave" : "packed", 100.0 * h[b + 6] / (tot + h[b + 5] + h[b + 6]), 100.0 * h[b + 5] / (tot + h[b + 5] + h[b + 6]), 100.0 * h[b + 0] / tot,
                    100.0 * h[b + 1] / tot, 100.0 * h[b + 2] / tot, 100.0 * h[b + 3] / tot, 100.0 * h[b + 4] / tot, tot);
        }
    }
    return hipGetLastError();
}

// cv::RNG replay on the host: initial samples of fitLine2D for every point count below kPickN2 (ascending per restart)
void build_pick_table(uint8_t* table, uint16_t* table16) {
    for (int n = 0; n < kPickN2; n++) {
        uint64_t state = 0xffffffffffffffffULL;
        for (int k = 0; k < 20; k++) {
            uint16_t pk[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (n >= 2) {
                const int npick = n < 10 ? n : 10;
                int got = 0;
                while (got < npick) {
                    state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
                    const int j = (int)((unsigned)state % (unsigned)n);
                    bool dup = false;
                    for (int q = 0; q < got; q++) dup |= (pk[q] == j);
                    if (!dup) pk[got++] = (uint16_t)j;
                }
                for (int a = 1; a < npick; a++) {
                    const uint16_t v = pk[a];
                    int b = a - 1;
                    while (b >= 0 && pk[b] > v) {
                        pk[b + 1] = pk[b];
                        b--;
                    }
                    pk[b + 1] = v;
                }
            }
            for (int q = 0; q < 10; q++) {
                if (n < kPickN) table[((size_t)n * 20 + k) * 10 + q] = (uint8_t)pk[q];
                else table16[((size_t)(n - kPickN) * 20 + k) * 10 + q] = pk[q];
            }
        }
    }
}

}  // namespace ctag
