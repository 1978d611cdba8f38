// ctag_refine.h -- the per-sample normal search of edgeRefine (/root/reference/corner_detector.cpp:623-657), in three
// forms that return THE SAME BITS (search_mid: at the end of the file):
//
//   search_exact   the reference's arithmetic, expression by expression: pixel = ((int)(x0 + m*nx), (int)(y0 + m*ny)) in
//                  double, running sums Mn += weight*n, Mcount += weight in step order.
//   search_fast    the same result from cheaper instructions (k_edge_refine is bound by vector-instruction ISSUE, an FP64
//                  add / mul / convert costing twice an integer one, profiles/r02_*_pmc_instmix.json):
//                  * pixel coordinates walk a 32.32 fixed-point progression X += D (two 32-bit integer adds per axis and
//                    step, the integer part IS the pixel index) instead of 2 FP64 multiplies, 3 FP64 adds and 2 converts.
//                    The walk may be off by < 2^-26 px; a coordinate that comes within 2^-22 px of an integer (where
//                    truncation could differ from the double expression) is detected with one v_min3_u32 per step and
//                    the sample is redone by search_exact (probability ~5e-5 per sample, and ALWAYS the same answer);
//                  * the running sums are exact in double (every weight is a float square, a multiple of 2^-39 <= 1; 65
//                    steps at most), hence independent of order and grouping:  with prefix sums P_k = w_0 + ... + w_k and
//                    Q = P_0 + ... + P_K,   Mcount = P_K   and   Mn = sum (k/4 - range) w_k = (range + 1/4) P_K - Q/4
//                    -- two FP64 adds per step instead of a multiply, two adds and the increment of n.
//                  Preconditions (else the caller uses search_exact): every fetched pixel is inside the image
//                  (`interior`), subpix <= kFastMaxSubpix.
// tests/test_refine_cpu.py checks the two against each other bit for bit on random and adversarial inputs (through the
// oracle library's probe), and the GPU parity tests check the kernel against the oracle, which keeps the literal loop.
#pragma once
#include <stdint.h>

#include "ctag_math.h"

#if defined(__HIPCC__)
#define CTR_UNROLL _Pragma("unroll")
#else
#define CTR_UNROLL
#endif
// On the device the coordinate walk is kept as ONE serial chain of 64-bit adds: left alone, the optimiser turns the eight
// unrolled positions of a group into eight induction variables of their own (2 x 8 extra 64-bit adds per group and ~90
// more registers, which halves the occupancy).  An empty asm that "modifies" the value is the barrier.
#if defined(__HIP_DEVICE_COMPILE__)
#define CTR_SERIAL(x) asm volatile("" : "+v"(x))
// The eight pixel requests of a group are issued before the first of them is used: left alone, the scheduler converts each byte
// as soon as it can, which puts a wait behind every second request (two LDS reads in flight, their latency exposed 25 times per
// search).  Nothing moves across this point.
#define CTR_ISSUE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define CTR_SERIAL(x) ((void)0)
#define CTR_ISSUE_FENCE() ((void)0)
#endif

namespace ctr {

constexpr int kFastMaxSubpix = 8;            // exactness bound of the prefix-sum form: 8*8+1 = 65 steps
constexpr uint32_t kGuard = 1u << 10;        // 2^-22 px on either side of an integer, in 2^-32 px units

CTM_HD float clamp01(float x) {  // x for x in [0, 1], 0 below (never above 1 here, never NaN)
    const float lo = x > 0.f ? x : 0.f;
    return lo < 1.f ? lo : 1.f;
}
CTM_HD float unit(unsigned u8) { return (float)u8 * (float)(1.0 / 255); }  // convertTo(CV_32F, 1/255), CylinderTag.cpp:101

// Reference arithmetic.  `px(x, y)` returns the pixel (0..255) at in-image integer coordinates.  Every pixel on the normal is
// fetched once: the point at n-1 is the point the step 8 earlier read at n+1 (same double expression, same pixel), kept in an
// 8-deep ring.  A step whose two pixels are not both inside the image, or whose gradient has the wrong sign, contributes
// weight +0.0, which leaves the running sums bit-identical to skipping it.
template <class Px>
CTM_HD void search_exact(double x0, double y0, double nx, double ny, int subpix, int rows, int cols, Px&& px, double& Mn_out, double& Mcount_out) {
    const double range = subpix;
    const int nsteps = 8 * subpix + 1;
    auto sample = [&](double m) -> float {  // pixel / 255 at (x0, y0) + m * normal, -1 outside the image
        const int x = (int)(x0 + m * nx);
        const int y = (int)(y0 + m * ny);
        const bool in = ((unsigned)x < (unsigned)cols) & ((unsigned)y < (unsigned)rows);
        const float g = unit(px(in ? x : 0, in ? y : 0));
        return in ? g : -1.f;
    };
    float ring[8];
    double m = -range - 1;  // all offsets are multiples of 0.25: exact
CTR_UNROLL
    for (int u = 0; u < 8; u++) {
        ring[u] = sample(m);
        m += 0.25;
    }
    double n = -range, Mn = 0, Mcount = 0;  // m == n + 1 from here on
    for (int st0 = 0; st0 < nsteps; st0 += 8) {
CTR_UNROLL
        for (int u = 0; u < 8; u++) {
            if (st0 + u < nsteps) {
                const float g1 = sample(m);
                const float g2 = ring[u];
                const bool use = (g1 >= 0.f) & (g2 >= 0.f) & !(g1 < g2);
                const double weight = use ? (double)((g2 - g1) * (g2 - g1)) : 0.0;
                Mn += weight * n;
                Mcount += weight;
                ring[u] = g1;
                m += 0.25;
                n += 0.25;
            }
        }
    }
    Mn_out = Mn;
    Mcount_out = Mcount;
}

// true when every pixel the search of this sample touches lies at least one pixel inside the image (so no bounds test is
// needed per pixel and a coordinate in (-1, 0), which truncates to 0, cannot occur)
CTM_HD bool interior(double x0, double y0, double nx, double ny, int subpix, int rows, int cols) {
    if (rows >= (1 << 19) || cols >= (1 << 19)) return false;  // to_fix32's range (a frame that large takes the reference arithmetic)
    const double r = (double)subpix + 1.0;
    const double xa = x0 - r * nx, xb = x0 + r * nx, ya = y0 - r * ny, yb = y0 + r * ny;
    const double xlo = xa < xb ? xa : xb, xhi = xa < xb ? xb : xa, ylo = ya < yb ? ya : yb, yhi = ya < yb ? yb : ya;
    return xlo >= 2.0 && ylo >= 2.0 && xhi <= (double)(cols - 3) && yhi <= (double)(rows - 3);
}

// Fast form; returns false when a coordinate came too close to an integer (the caller then runs search_exact).
// Requires interior(...) and subpix <= kFastMaxSubpix.
// `px` is called with coordinates relative to (org_x, org_y) (the origin of a staged box; exact: an integer offset of the
// integer part).
// SUBPIX > 0: the window is a compile-time constant (the group loop unrolls completely and the pixel ring becomes register
// renaming); SUBPIX == 0: `subpix` at run time.
// v in 32.32 fixed point (two's complement), rounded to nearest, for |v| < 2^19: adding 1.5 * 2^20 leaves a double whose unit in the
// last place is 2^-32, so its mantissa IS the fixed-point number (plus 2^19 * 2^32): one FP64 add, an AND and a subtraction -- a
// device has no double -> int64 conversion and builds one from ~8 double-rate instructions.
CTM_HD uint64_t to_fix32(double v) {
    const uint64_t bits = ctm::f64_to_bits(v + 1572864.0);
    return (bits & 0x000fffffffffffffULL) - 0x0008000000000000ULL;
}
// the step of a search (1/4 of the unit normal) in 32.32: the same for every sample of an edge, so a caller may compute it once
CTM_HD uint64_t fast_step(double n) { return to_fix32(n * 0.25); }

template <int SUBPIX = 0, class Px>
CTM_HD bool search_fast(double x0, double y0, double nx, double ny, int subpix_rt, Px&& px, double& Mn_out, double& Mcount_out, int org_x = 0, int org_y = 0,
                        const uint64_t* step_xy = nullptr /* {fast_step(nx), fast_step(ny)} when the caller has them */) {
    const int subpix = SUBPIX > 0 ? SUBPIX : subpix_rt;
    const double range = subpix;
    // start point (m = -range - 1) and step (1/4 of the normal) in 32.32 fixed point, biased by +kGuard so that the low word
    // of a coordinate within kGuard of an integer reads < 2*kGuard.  (Start and step are rounded to nearest: 2^-33 px each, inside
    // the 2^-26 px the walk may be off by.)
    uint64_t X = to_fix32(x0 - (range + 1) * nx) + kGuard - ((uint64_t)(uint32_t)org_x << 32);
    uint64_t Y = to_fix32(y0 - (range + 1) * ny) + kGuard - ((uint64_t)(uint32_t)org_y << 32);
    const uint64_t DX = step_xy ? step_xy[0] : fast_step(nx);
    const uint64_t DY = step_xy ? step_xy[1] : fast_step(ny);
    uint32_t gmin = 0xffffffffu;
    auto fetch = [&]() -> unsigned {  // the pixel, 0..255
        const uint32_t xl = (uint32_t)X, yl = (uint32_t)Y;
        gmin = gmin < xl ? gmin : xl;
        gmin = gmin < yl ? gmin : yl;
        const unsigned v = px((int)(uint32_t)(X >> 32), (int)(uint32_t)(Y >> 32));
        X += DX;
        Y += DY;
        CTR_SERIAL(X);
        CTR_SERIAL(Y);
        return v;
    };
    float ring[8];
    {
        unsigned raw[8];
CTR_UNROLL
        for (int u = 0; u < 8; u++) raw[u] = fetch();
        CTR_ISSUE_FENCE();
CTR_UNROLL
        for (int u = 0; u < 8; u++) ring[u] = unit(raw[u]);
    }
    double P = 0, Q = 0;
    auto step = [&](float g1, float g2) {
        // weight (g2 - g1)^2 when !(g1 < g2), else the step is skipped (:643-645): that is e^2 with e = max(g1 - g2, 0) (g1 - g2 is
        // -(g2 - g1) exactly; g1 < g2 -> 0 * 0 = +0).  e <= 1, so the clamp to [0, 1] below changes nothing -- and is free on the
        // device: an output modifier of the subtraction (v_sub_f32 ... clamp), where max alone is an instruction
        const float e = clamp01(g1 - g2);
        P += (double)(e * e);
        Q += P;
    };
    // nsteps = 8 * subpix + 1: subpix groups of eight steps whose eight pixels are requested together (their addresses do
    // not depend on the data), then the last step
#if defined(__HIPCC__)
#pragma unroll SUBPIX > 0 ? SUBPIX : 1
#endif
    for (int it = 0; it < subpix; it++) {
        unsigned raw[8];
CTR_UNROLL
        for (int u = 0; u < 8; u++) raw[u] = fetch();
        CTR_ISSUE_FENCE();
CTR_UNROLL
        for (int u = 0; u < 8; u++) {
            const float g = unit(raw[u]);
            step(g, ring[u]);
            ring[u] = g;
        }
    }
    step(unit(fetch()), ring[0]);
    Mcount_out = P;
    Mn_out = (range + 0.25) * P - 0.25 * Q;
    return gmin >= 2u * kGuard;
}

// Middle form: the reference's coordinate expressions in double (so nothing to guard and nothing to decline) with the fast form's
// prefix-sum moments and staged pixels.  For the samples the fast form cannot take: an axis-aligned edge between corners at x.5
// puts every fourth step of every sample exactly on a pixel border (half-resolution corners times two: 5 % of the edges of the
// synthetic frames), where only the reference's own rounding says which pixel it is.  Costs ~1.4 fast searches instead of the
// ~5 of search_exact (bounds tests, global pixels).  Requires interior(...) and subpix <= kFastMaxSubpix.
template <int SUBPIX = 0, class Px>
CTM_HD void search_mid(double x0, double y0, double nx, double ny, int subpix_rt, Px&& px, double& Mn_out, double& Mcount_out, int org_x = 0, int org_y = 0) {
    const int subpix = SUBPIX > 0 ? SUBPIX : subpix_rt;
    const double range = subpix;
    double m = -range - 1;  // multiples of 0.25: exact
    auto fetch = [&]() -> float {
        const int x = (int)(x0 + m * nx);
        const int y = (int)(y0 + m * ny);
        m += 0.25;
        return unit(px(x - org_x, y - org_y));
    };
    float ring[8];
CTR_UNROLL
    for (int u = 0; u < 8; u++) ring[u] = fetch();
    double P = 0, Q = 0;
    auto step = [&](float g1, float g2) {
        const float e = clamp01(g1 - g2);
        P += (double)(e * e);
        Q += P;
    };
    for (int it = 0; it < subpix; it++) {
        float g[8];
CTR_UNROLL
        for (int u = 0; u < 8; u++) g[u] = fetch();
CTR_UNROLL
        for (int u = 0; u < 8; u++) {
            step(g[u], ring[u]);
            ring[u] = g[u];
        }
    }
    step(fetch(), ring[0]);
    Mcount_out = P;
    Mn_out = (range + 0.25) * P - 0.25 * Q;
}

}  // namespace ctr
