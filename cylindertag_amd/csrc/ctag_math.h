// ctag_math.h -- deterministic elementary functions shared by the HIP kernels and the CPU oracle.
//
// Why this exists: the reference (corner_detector.cpp) calls atan2f/cosf/sinf/atan2/sqrt and, through
// OpenCV's fitLine, atan2/cos/sin/exp/acos.  Those libm results are not bit-portable between glibc, MSVC
// and ROCm's OCML, and the detection path is full of knife-edge branches.  Every
// function below is built only from IEEE-754 +,-,*,/ and sqrt in a fixed evaluation order, so the same
// source gives the same bits under gcc (x86-64, -ffp-contract=off) and hipcc (gfx950, -ffp-contract=off).
//
// Accuracy: the double kernels follow the classic fdlibm argument-reduction + minimax-polynomial schemes
// (error < 1 ulp in double).  The float entry points evaluate in double and round once, which is the
// correctly rounded float result except in ~2^-29 of cases.  tests/test_math.py pins all of this against
// glibc on the CPU and against the gfx950 build on the GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CTM_HD __host__ __device__ __forceinline__
#else
#define CTM_HD static inline
#endif

namespace ctm {

CTM_HD double bits_to_f64(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
CTM_HD uint64_t f64_to_bits(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }
CTM_HD float bits_to_f32(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
CTM_HD uint32_t f32_to_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }

CTM_HD double fabs64(double x) { return bits_to_f64(f64_to_bits(x) & 0x7fffffffffffffffULL); }
CTM_HD float fabs32(float x) { return bits_to_f32(f32_to_bits(x) & 0x7fffffffu); }
CTM_HD bool isnan64(double x) { return x != x; }

// ---------------------------------------------------------------- atan / atan2 (double)
CTM_HD double atan64(double x) {
    const double hi0 = 4.63647609000806093515e-01, hi1 = 7.85398163397448278999e-01,
                 hi2 = 9.82793723247329054082e-01, hi3 = 1.57079632679489655800e+00;
    const double lo0 = 2.26987774529616870924e-17, lo1 = 3.06161699786838301793e-17,
                 lo2 = 1.39033110312309984516e-17, lo3 = 6.12323399573676603587e-17;
    const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01,
                 a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01,
                 a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
                 a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02,
                 a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02,
                 a10 = 1.62858201153657823623e-02;
    if (isnan64(x)) return x;
    const bool neg = (f64_to_bits(x) >> 63) != 0;
    double ax = fabs64(x);
    if (ax >= 7.3786976294838206464e19) {  // 2^66: atan = +-pi/2
        double z = hi3 + lo3;
        return neg ? -z : z;
    }
    // One division whatever the range (selected numerator / denominator: a wave whose lanes fall into different ranges runs
    // one division, not five paths one after the other); below 0.4375 t = ax / 1.0, which is ax exactly.
    if (ax < 3.725290298461914e-09) return x;  // 2^-28
    const int id = ax < 0.4375 ? -1 : ax < 0.6875 ? 0 : ax < 1.1875 ? 1 : ax < 2.4375 ? 2 : 3;
    const double num = id < 0 ? ax : id == 0 ? 2.0 * ax - 1.0 : id == 1 ? ax - 1.0 : id == 2 ? ax - 1.5 : -1.0;
    const double den = id < 0 ? 1.0 : id == 0 ? 2.0 + ax : id == 1 ? ax + 1.0 : id == 2 ? 1.0 + 1.5 * ax : ax;
    const double t = num / den;
    const double z = t * t;
    const double w = z * z;
    const double s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const double s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    const double h = id == 0 ? hi0 : id == 1 ? hi1 : id == 2 ? hi2 : hi3;
    const double l = id == 0 ? lo0 : id == 1 ? lo1 : id == 2 ? lo2 : lo3;
    const double ts = t * (s1 + s2);
    const double r = id < 0 ? t - ts : h - ((ts - l) - t);
    return neg ? -r : r;
}

CTM_HD double atan2_64(double y, double x) {
    const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    const double pi_o_2 = 1.5707963267948965580E+00, pi_o_4 = 7.8539816339744827900E-01;
    if (isnan64(x) || isnan64(y)) return x + y;
    const bool sx = (f64_to_bits(x) >> 63) != 0, sy = (f64_to_bits(y) >> 63) != 0;
    const double ax = fabs64(x), ay = fabs64(y);
    const double inf = bits_to_f64(0x7ff0000000000000ULL);
    if (ay == 0.0) {
        if (!sx) return y;  // +-0
        return sy ? -pi : pi;
    }
    if (ax == 0.0) return sy ? -pi_o_2 : pi_o_2;
    if (ax == inf) {
        if (ay == inf) {
            const double v = sx ? 3.0 * pi_o_4 : pi_o_4;
            return sy ? -v : v;
        }
        const double v = sx ? pi : 0.0;
        return sy ? -v : v;
    }
    if (ay == inf) return sy ? -pi_o_2 : pi_o_2;
    // exponent difference shortcut as in fdlibm: |y/x| > 2^60 or < 2^-60
    const int ey = (int)((f64_to_bits(ay) >> 52) & 0x7ff), ex = (int)((f64_to_bits(ax) >> 52) & 0x7ff);
    const int k = ey - ex;
    double z;
    if (k > 60) {
        z = pi_o_2 + 0.5 * pi_lo;
    } else if (sx && k < -60) {
        z = 0.0;
    } else {
        z = atan64(fabs64(y / x));
    }
    if (!sx) return sy ? -z : z;
    return sy ? (z - pi_lo) - pi : pi - (z - pi_lo);
}

// ---------------------------------------------------------------- sin / cos (double), |x| modest
CTM_HD double ksin64(double x) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    const double v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x + v * (S1 + z * r);
}
CTM_HD double kcos64(double x) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double ax = fabs64(x);
    if (ax < 0.3) return 1.0 - (0.5 * z - z * r);
    const double qx = ax > 0.78125 ? 0.28125 : 0.25 * ax;
    const double hz = 0.5 * z - qx;
    const double a = 1.0 - qx;
    return a - (hz - z * r);
}
// reduce x to r in [-pi/4, pi/4], return quadrant (valid for |x| < ~1e5, far beyond what the path needs)
CTM_HD int rem_pio2_64(double x, double* r) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double p1 = 1.57079632673412561417e+00, p1t = 6.07710050650619224932e-11;
    const double p2 = 6.07710050630396597660e-11, p2t = 2.02226624879595063154e-21;
    const double fn = __builtin_floor(x * invpio2 + 0.5);
    const double a = x - fn * p1;
    const double w = fn * p2;
    const double b = a - w;
    const double w2 = fn * p2t - ((a - b) - w);
    *r = b - w2;
    (void)p1t;
    return (int)((long long)fn & 3);
}
CTM_HD double sin64(double x) {
    if (fabs64(x) <= 7.85398163397448278999e-01) return ksin64(x);
    double r;
    const int q = rem_pio2_64(x, &r);
    switch (q) {
        case 0: return ksin64(r);
        case 1: return kcos64(r);
        case 2: return -ksin64(r);
        default: return -kcos64(r);
    }
}
CTM_HD double cos64(double x) {
    if (fabs64(x) <= 7.85398163397448278999e-01) return kcos64(x);
    double r;
    const int q = rem_pio2_64(x, &r);
    switch (q) {
        case 0: return kcos64(r);
        case 1: return -ksin64(r);
        case 2: return -kcos64(r);
        default: return ksin64(r);
    }
}

// sin64(x) and cos64(x) at once, bit for bit what the two calls return: one argument reduction, both kernels evaluated once and
// assigned by quadrant with selects (lanes in different quadrants do not run the four cases one after the other)
CTM_HD void sincos64(double x, double* s, double* c) {
    double r = x;
    int q = 0;
    if (!(fabs64(x) <= 7.85398163397448278999e-01)) q = rem_pio2_64(x, &r);
    const double ks = ksin64(r), kc = kcos64(r);
    const double sv = (q & 1) ? kc : ks, cv = (q & 1) ? ks : kc;
    *s = (q & 2) ? -sv : sv;
    *c = ((q + 1) & 2) ? -cv : cv;
}

// ---------------------------------------------------------------- exp (double), acos (double)
// exp(x) = 2^k * 2^(j/32) * exp(r),  x = (32k + j) * ln2/32 + r,  |r| <= ln2/64: a 32-entry table of correctly rounded
// 2^(j/32) and the Taylor series of exp(r) to r^6 (truncation < 4e-17 on the reduced range); no division.
CTM_HD double exp64(double x) {
    const double T[32] = {
        1.00000000000000000e+00, 1.02189714865411663e+00, 1.04427378242741375e+00, 1.06714040067682370e+00,
        1.09050773266525769e+00, 1.11438674259589243e+00, 1.13878863475669156e+00, 1.16372485877757748e+00,
        1.18920711500272103e+00, 1.21524735998046896e+00, 1.24185781207348400e+00, 1.26905095719173322e+00,
        1.29683955465100964e+00, 1.32523664315974132e+00, 1.35425554693689265e+00, 1.38390988196383202e+00,
        1.41421356237309515e+00, 1.44518080697704665e+00, 1.47682614593949935e+00, 1.50916442759342284e+00,
        1.54221082540794074e+00, 1.57598084510788650e+00, 1.61049033194925428e+00, 1.64575547815396495e+00,
        1.68179283050742900e+00, 1.71861929812247793e+00, 1.75625216037329945e+00, 1.79470907500310717e+00,
        1.83400808640934243e+00, 1.87416763411029996e+00, 1.91520656139714740e+00, 1.95714412417540018e+00};
    const double inv = 4.61662413084468283841e+01;     // 32 / ln2
    const double c_hi = 2.16608493865351192653e-02;    // ln2/32, upper bits (exact product with |n| < 2^16)
    const double c_lo = 5.96317165397058656257e-12;    // ln2/32 - c_hi
    if (isnan64(x)) return x;
    if (x > 709.0) return bits_to_f64(0x7ff0000000000000ULL);
    if (x < -708.0) return 0.0;  // callers only need float range; no double denormals here
    const double fn = __builtin_floor(x * inv + 0.5);
    const int n = (int)fn;
    const int j = n & 31, k = (n - j) / 32;  // n = 32k + j, 0 <= j < 32 (also for negative n)
    const double r = (x - fn * c_hi) - fn * c_lo;
    double p = 1.3888888888888889e-03;   // 1/6!
    p = p * r + 8.3333333333333332e-03;  // 1/5!
    p = p * r + 4.1666666666666664e-02;  // 1/4!
    p = p * r + 1.6666666666666666e-01;  // 1/3!
    p = p * r + 0.5;
    p = p * r + 1.0;
    const double e = p * r + 1.0;
    const double scale = bits_to_f64((uint64_t)(k + 1023) << 52);  // 2^k, k in [-1022, 1023] given the clamps
    return (T[j] * e) * scale;
}

// acos via atan2; exact subtraction 1-t for the float-valued t the path feeds it
CTM_HD double acos64(double t) {
    if (t >= 1.0) return 0.0;
    if (t <= -1.0) return 3.1415926535897931160E+00;
    return 2.0 * atan2_64(__builtin_sqrt(1.0 - t), __builtin_sqrt(1.0 + t));
}

// smallest float t with fabs(acos64((double)t)) < 0.01f  (0x3f7ffcba = 0.99995005130767822266); see k_welsch
constexpr float kAcosBelowTenMilli = 0.99995005130767822266f;

// ---------------------------------------------------------------- float entry points
CTM_HD float atan2_32(float y, float x) { return (float)atan2_64((double)y, (double)x); }
CTM_HD float sin32(float x) { return (float)sin64((double)x); }
CTM_HD float cos32(float x) { return (float)cos64((double)x); }
CTM_HD void sincos32(float x, float* s, float* c) {  // sin32(x), cos32(x)
    double sd, cd;
    sincos64((double)x, &sd, &cd);
    *s = (float)sd;
    *c = (float)cd;
}
// expf as the Welsch weight uses it (fitLine's weightWelsch calls std::exp on a float).  Evaluated in FLOAT arithmetic
// only -- the Welsch loop calls it twice per point and iteration, and FP64 runs at half rate on the vector unit:
//   x = k ln2 + r (Cody-Waite, k ln2_hi exact for |k| < 2^8),  exp(r) = 1 + r + r^2 P(r),  P of degree 4 (near-minimax
//   on |r| <= ln2/2), result scaled by 2^k.
// Only IEEE +,-,* and rint, no contraction: the host and the device produce the same bits.  Against the correctly rounded
// value it is never more than 1 ulp off (90 % exact); glibc's and MSVC's expf have the same bound, which is all the
// reference can rely on.  Arguments below -87 flush to zero so no denormals are produced.
CTM_HD float exp32(float x) {
    if (x != x) return x;
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return bits_to_f32(0x7f800000u);
    const float k = __builtin_rintf(x * 1.44269502162933349609375f);
    const float r = (x - k * 0.693145751953125f) - k * 1.428606765330187045037746429443359375e-06f;
    float p = 1.3933733571320772171020508e-03f;
    p = p * r + 8.3632357418537139892578125e-03f;
    p = p * r + 4.1666463017463684082031250e-02f;
    p = p * r + 1.6666576266288757324218750e-01f;
    p = p * r + 0.5f;
    const float e = (p * (r * r) + r) + 1.0f;
    return e * bits_to_f32((uint32_t)((int)k + 127) << 23);  // k in [-126, 127] given the clamps: 2^k is a normal float
}
// exp32 for the Welsch weights: the argument is -(r c)^2 -- never NaN, never positive -- so of exp32's three range tests only the flush
// below -87 is left, and the scaling by 2^k is one v_ldexp_f32 on the device instead of add, shift and multiply (the product is a
// normal float, so e * 2^k == ldexp(e, k) bit for bit).  Same bits as exp32 on its whole domain x <= 0 (tests/test_oracle_cpu.py).
CTM_HD float exp32_nonpos(float x) {
    if (x < -87.0f) return 0.0f;
    const float k = __builtin_rintf(x * 1.44269502162933349609375f);
    const float r = (x - k * 0.693145751953125f) - k * 1.428606765330187045037746429443359375e-06f;
    float p = 1.3933733571320772171020508e-03f;
    p = p * r + 8.3632357418537139892578125e-03f;
    p = p * r + 4.1666463017463684082031250e-02f;
    p = p * r + 1.6666576266288757324218750e-01f;
    p = p * r + 0.5f;
    const float e = (p * (r * r) + r) + 1.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ldexpf(e, (int)k);
#else
    return e * bits_to_f32((uint32_t)((int)k + 127) << 23);
#endif
}
// Several quotients with ONE denominator (the five moments over their weight in fitLine and in edgeRefine): IEEE division, so a / d
// on the host; on the device the compiler's own expansion of a double division -- v_rcp_f64, two Newton steps on the reciprocal, then
// per numerator q = a r, e = a - d q, q + e r (correctly rounded: it is the sequence v_div_fmas / v_div_fixup finish) -- with the
// reciprocal's part done once instead of once per quotient.  The expansion's v_div_scale / v_div_fixup guard exponents near the ends of
// the range; left out, so: d and every a / d normal and far from overflow (|d| in [2^-500, 2^500], |a| <= 2^500 |d|, a / d >= 2^-500
// or zero), which sums of at most a few thousand pixel coordinates times weights in (0, 1] are.  tests: math probe 16 against `/`.
struct Recip64 {
    double d, r;
};
CTM_HD Recip64 recip64(double d) {
    Recip64 R;
    R.d = d;
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    R.r = r;
#else
    R.r = 0.0;
#endif
    return R;
}
CTM_HD double div64(double a, const Recip64& R) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double q = a * R.r;
    return __builtin_fma(__builtin_fma(-R.d, q, a), R.r, q);
#else
    return a / R.d;
#endif
}
CTM_HD float sqrt32(float x) { return __builtin_sqrtf(x); }
CTM_HD double sqrt64(double x) { return __builtin_sqrt(x); }
// roundf (half away from zero), as std::round(float)
CTM_HD float round32(float x) {
    const float ax = fabs32(x);
    if (!(ax < 8388608.0f)) return x;
    float r = __builtin_floorf(ax);
    if (ax - r >= 0.5f) r += 1.0f;
    return (f32_to_bits(x) >> 31) ? -r : r;
}

// cv::fastAtan2 (degrees, [0,360)) -- SURVEY.md App. A.8 [OCV-recall of mathfuncs_core atan_f32]
CTM_HD float fast_atan2_deg(float y, float x) {
    const float s = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s,
                p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabs32(x), ay = fabs32(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.0f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.0f - a;
    if (y < 0) a = 360.0f - a;
    return a;
}

}  // namespace ctm
