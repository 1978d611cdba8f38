// k_sweep.hip -- the "threshold + label sweep": K1 decimate, K2 threshold + tile CCL, K3 seam merge,
// K4 resolve, K5 candidate ordering.  Replaces, for the GPU path, the reference's
//   resize(INTER_CUBIC) + convertTo            /root/reference/CylinderTag.cpp:79-80
//   corner_detector::adaptiveThreshold          /root/reference/corner_detector.cpp:28-79
//   corner_detector::connectedComponentLabeling /root/reference/corner_detector.cpp:81-107
// HBM-bound integer/byte work: coalesced 16-byte row sweeps, LDS tiles, wave64 ballots for the row masks,
// union-find over row runs in LDS, one global atomic union-find only for the tile seams.
// Frames are mapped to XCDs (blockIdx % 8) so a frame's intermediates stay in one XCD's L2.
#include "ctag_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace ctag {

// blocks b and b+8 share an XCD (observed round-robin dispatch; speed only, never correctness)
__device__ __forceinline__ bool map_block(int b, int per_frame, int nframes, int& frame, int& idx) {
    const int xcd = b & 7;
    const int i = b >> 3;
    frame = (i / per_frame) * 8 + xcd;
    idx = i % per_frame;
    return frame < nframes;
}
static inline int grid_for(int nframes, int per_frame) { return ((nframes + 7) / 8) * 8 * per_frame; }

// the fused sweep (k_decimate_mask below)
constexpr int kFuseCols = 960;     // half-resolution columns per wave
constexpr int kFuseLanes = 60;     // ... = 60 lanes x 16 pixels
constexpr int kFuseTiles = kFuseCols / 5;
static hipError_t launch_decimate_mask(const uint8_t* frames, ptrdiff_t frame_stride, ptrdiff_t row_stride, int nframes, const Workspace& ws, hipStream_t s, int channels);

// =====================================================================================================
// K1: bicubic 2x decimation, u8 -> u8.  OpenCV resize(INTER_CUBIC) for an exact 2x scale has the fixed
// taps [-192, 1216, 1216, -192]/2048 at source positions 2x-1..2x+2 (index-clamped at the borders); the
// horizontal pass is integer, the vertical pass follows the float vector body (s0*b0 + (s1*b1 + (s2*b2 +
// s3*b3)), round-half-even, saturate) -- SURVEY.md App. A.1; the oracle restates the same arithmetic.
// Each lane owns 8 output pixels (one 16-byte source load per source row) and slides down a band of rows
// keeping the four horizontal-pass rows in registers.
// =====================================================================================================
// Band height: the host picks it so that a frame has a multiple of four bands (every 4-wave block full) of at most
// kDecBandMax rows -- 135 rows for 1080p and 4K.  Taller bands re-read fewer prologue rows (6 source rows per band);
// measured on 4096 1080p frames: 45 rows 2.35 ms, 90 (half-empty blocks) 2.58, 135 2.21, 180 2.49, 270 2.97.
#ifndef CTAG_DEC_BAND_MAX
#define CTAG_DEC_BAND_MAX 150
#endif
constexpr int kDecBandMax = CTAG_DEC_BAND_MAX;

struct Raw18 {  // source pixels x0-1 .. x0+16 of one row
    uint32_t w0, w1, w2, w3;
    uint32_t left;    // pixel x0-1
    uint32_t right2;  // pixels x0+16 (bits 0-7) and x0+17 (bits 8-15)
};

template <bool ALIGNED>
__device__ __forceinline__ Raw18 load_row(const uint8_t* __restrict__ rowp, int x0, int cols, bool active, int lane) {
    Raw18 r;
    r.w0 = r.w1 = r.w2 = r.w3 = 0;
    bool full = false;
    if (active) {
        if (ALIGNED && x0 + 16 <= cols) {
            const uint4 v = *reinterpret_cast<const uint4*>(rowp + x0);
            r.w0 = v.x;
            r.w1 = v.y;
            r.w2 = v.z;
            r.w3 = v.w;
            full = true;
        } else {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int xi = min(x0 + i, cols - 1);
                w[i >> 2] |= (uint32_t)rowp[xi] << (8 * (i & 3));
            }
            r.w0 = w[0];
            r.w1 = w[1];
            r.w2 = w[2];
            r.w3 = w[3];
        }
    }
    // the neighbour lanes' edge pixels by DPP wave shifts (lane 0 / lane 63 get 0 and are replaced below), not by the LDS crossbar
    uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(r.w3 >> 24), 0x138, 0xf, 0xf, false);       // wave_shr:1
    uint32_t right2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(r.w0 & 0xffffu), 0x130, 0xf, 0xf, false);  // wave_shl:1
    if (active) {
        if (lane == 0) left = rowp[max(x0 - 1, 0)];
        if (lane == 63 || !full || x0 + 32 > cols) {
            right2 = (uint32_t)rowp[min(x0 + 16, cols - 1)] | ((uint32_t)rowp[min(x0 + 17, cols - 1)] << 8);
        }
    }
    r.left = left;
    r.right2 = right2;
    return r;
}

// horizontal pass / 64 : q[i] = 19*(p[2i]+p[2i+1]) - 3*(p[2i-1]+p[2i+2])
__device__ __forceinline__ int px(const Raw18& r, int k) {  // k in [-1, 16]
    if (k < 0) return (int)r.left;
    if (k >= 16) return (int)(r.right2 & 0xff);
    const uint32_t w = k < 4 ? r.w0 : k < 8 ? r.w1 : k < 12 ? r.w2 : r.w3;
    return (int)((w >> (8 * (k & 3))) & 0xff);
}
// The horizontal-pass values lie in [-1530, 9690]: two per register as int16 halves the kernel's largest live set (six
// rows of eight values), which is what limits its occupancy; the sums qb+qc and qa+qd are formed on the packed pairs.
typedef short ctag_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_s16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ uint32_t pk_add_s16(uint32_t a, uint32_t b) {
    ctag_s2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    const ctag_s2 r = x + y;
    uint32_t o;
    __builtin_memcpy(&o, &r, 4);
    return o;
}
__device__ __forceinline__ void hpass(const Raw18& r, uint32_t q[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int q0 = 19 * (px(r, 4 * i) + px(r, 4 * i + 1)) - 3 * (px(r, 4 * i - 1) + px(r, 4 * i + 2));
        const int q1 = 19 * (px(r, 4 * i + 2) + px(r, 4 * i + 3)) - 3 * (px(r, 4 * i + 1) + px(r, 4 * i + 4));
        q[i] = pack_s16(q0, q1);
    }
}

// Width of OpenCV's vector body in the vertical pass (8: a 128-bit universal-intrinsics build, the assumption; 16: a 256-bit body -- EXTRA=-DCTAG_RESIZE_SIMD_LANES=16
// with the oracle's ctago_set_variants(.., 16)).  Frames whose half width is a multiple of 16 (1080p, 4K, 8K, test.bmp) give the same image either way.
#ifndef CTAG_RESIZE_SIMD_LANES
#define CTAG_RESIZE_SIMD_LANES 8
#endif
static_assert(CTAG_RESIZE_SIMD_LANES == 8 || CTAG_RESIZE_SIMD_LANES == 16, "vector body of 8 or 16 columns");
// vertical pass.  OpenCV's float vector body computes t = s0*b0 + (s1*b1 + (s2*b2 + s3*b3)) with s = 64*q and
// b = {-192,1216,1216,-192} * 2^-22 and rounds half-to-even.  Every product and partial sum is a multiple of 2^-10 of
// magnitude < 2^14, i.e. exactly representable in float, so t == V / 1024 exactly with V = 19*(qb+qc) - 3*(qa+qd) and
// the result is the integer round-half-even of V/1024 (tests/test_oracle_cpu.py checks this identity against the
// literal float evaluation the oracle uses).
// Output columns at or beyond (hcols & ~7) are OpenCV's scalar row tail: integer FixedPtCast, (v + 2^21) >> 22, i.e.
// round-half-UP of V/1024 -- `tail` switches the tie rule off for those (a lane's 8 columns are all body or all tail).
// Round-half-even of V/1024 is (V + 511 + bit10(V)) >> 10 (remainder < 512: no carry; == 512: carries iff the quotient is
// odd; > 512: carries), round-half-up is the same with the bit forced to 1 -- checked for every V in tests/test_oracle_cpu.py.
// The shift, the saturation to 0..255 and the packing of two pixels are ONE gfx950 instruction, v_ashr_pk_u8_i32.  It
// writes bits 15:0 of its destination only (measured: hipcc 7.2 forms it by itself from min(max(x >> 10, 0), 255) pairs and
// then ORs the stale upper half into the neighbouring pixels), so the halves are gathered with v_perm_b32, which reads
// nothing but the two valid bytes of each.
__device__ __forceinline__ int vsum(int V, int tail) { return V + 511 + (((V >> 10) & 1) | tail); }
// two output pixels from packed rows: S1 = qb + qc and S2 = qa + qd stay inside int16 (|.| <= 19380); result in bits 15:0
// (19 * s1 - 3 * s2 as ONE v_dot2_i32_i16 per pixel on the pair (s1, s2): the halves are gathered by v_perm_b32 instead of two sign
// extensions, and the multiply-adds go away -- 11 instead of 15 vector instructions per pixel pair; same integers)
__device__ __forceinline__ uint32_t vpass2(uint32_t qa, uint32_t qb, uint32_t qc, uint32_t qd, int tail) {
    const uint32_t s1 = pk_add_s16(qb, qc), s2 = pk_add_s16(qa, qd);
    const uint32_t lo = __builtin_amdgcn_perm(s2, s1, 0x05040100u), hi = __builtin_amdgcn_perm(s2, s1, 0x07060302u);  // (s1.lo, s2.lo), (s1.hi, s2.hi)
    ctag_s2 xl, xh;
    __builtin_memcpy(&xl, &lo, 4);
    __builtin_memcpy(&xh, &hi, 4);
    const ctag_s2 k = {(short)19, (short)-3};
    const int v0 = __builtin_amdgcn_sdot2(xl, k, 0, false), v1 = __builtin_amdgcn_sdot2(xh, k, 0, false);
    return (uint32_t)__builtin_amdgcn_ashr_pk_u8_i32(vsum(v0, tail), vsum(v1, tail), 10);
}
__device__ __forceinline__ uint32_t vpack4(uint32_t p01, uint32_t p23) { return __builtin_amdgcn_perm(p23, p01, 0x05040100u); }

// 5 waves per SIMD (94 VGPRs): measured 4 waves 2.40 ms, 5 waves 2.29, 6 waves (14 spilled registers) 2.34; nontemporal source
// loads 2.46
// the per-chunk counters a chain starts by zeroing (k_zero_counters); calls of a few frames pass them to their first kernel instead of launching one for them
struct ZeroList {
    int32_t* a;
    uint32_t* b;
    int32_t* c;
    int32_t* d;
    int32_t* one;
    int n;  // frames (<= 256); 0: nothing to zero
};
template <bool ALIGNED, int BAND, bool HAS_TAIL>  // BAND: compile-time band height (0 = run-time band_rows_rt); HAS_TAIL: hcols % 8 != 0
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void k_decimate(const uint8_t* __restrict__ frames, ptrdiff_t frame_stride, ptrdiff_t row_stride,
                                                  uint8_t* __restrict__ half, FrameGeom g, int nframes, int xblocks, int yblocks, int band_rows_rt, ZeroList Z) {
    if (BAND == 0 && blockIdx.x == 0) {  // (the kernels behind this one read the counters: nothing in this one does)
        if ((int)threadIdx.x < Z.n) {
            Z.a[threadIdx.x] = 0;
            Z.b[threadIdx.x] = 0u;
            Z.c[threadIdx.x] = 0;
            Z.d[threadIdx.x] = 0;
        }
        if (threadIdx.x == 0 && Z.n > 0) *Z.one = 0;
    }
    const int band_rows = BAND ? BAND : band_rows_rt;
    int frame, idx;
    if (!map_block(blockIdx.x, xblocks * yblocks, nframes, frame, idx)) return;
    const int bx = idx % xblocks, by = idx / xblocks;
    const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const int band = by * 4 + wy;
    const int y_begin = band * band_rows;
    if (y_begin >= g.hrows) return;  // wave-uniform
    const int y_end = min(y_begin + band_rows, g.hrows);
    const int hx0 = (bx * 64 + lane) * 8;
    const int x0 = hx0 * 2;
    const bool active = hx0 < g.hcols;
    const int tail = (HAS_TAIL && hx0 >= (g.hcols & ~(CTAG_RESIZE_SIMD_LANES - 1))) ? 1 : 0;
    const uint8_t* __restrict__ src = frames + (ptrdiff_t)frame * frame_stride;
    uint8_t* __restrict__ dst = half + ((size_t)frame * g.hrows) * g.hp;
    const int rmax = g.rows - 1;
    auto rowp = [&](int r) { return src + (ptrdiff_t)min(max(r, 0), rmax) * row_stride; };

    uint32_t qa[4], qb[4], qc[4], qd[4];
    {
        const Raw18 ra = load_row<ALIGNED>(rowp(2 * y_begin - 1), x0, g.cols, active, lane);
        const Raw18 rb = load_row<ALIGNED>(rowp(2 * y_begin), x0, g.cols, active, lane);
        const Raw18 rc = load_row<ALIGNED>(rowp(2 * y_begin + 1), x0, g.cols, active, lane);
        const Raw18 rd = load_row<ALIGNED>(rowp(2 * y_begin + 2), x0, g.cols, active, lane);
        hpass(ra, qa);
        hpass(rb, qb);
        hpass(rc, qc);
        hpass(rd, qd);
    }
    // two output rows per iteration: the four source rows of the NEXT iteration are requested before this iteration's
    // arithmetic, so every lane keeps 64 bytes in flight (the kernel is bound by memory latency x occupancy)
    auto emit = [&](int y, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d) {
        const uint32_t lo = vpack4(vpass2(a[0], b[0], c[0], d[0], tail), vpass2(a[1], b[1], c[1], d[1], tail));
        const uint32_t hi = vpack4(vpass2(a[2], b[2], c[2], d[2], tail), vpass2(a[3], b[3], c[3], d[3], tail));
        if (active) *reinterpret_cast<uint2*>(dst + (size_t)y * g.hp + hx0) = make_uint2(lo, hi);
    };
    Raw18 n0 = load_row<ALIGNED>(rowp(2 * y_begin + 3), x0, g.cols, active, lane);
    Raw18 n1 = load_row<ALIGNED>(rowp(2 * y_begin + 4), x0, g.cols, active, lane);
    for (int y = y_begin; y < y_end; y += 2) {
        // rows 2y+5, 2y+6 feed output rows y+2 (with 2y+3, 2y+4 already requested)
        const Raw18 m0 = load_row<ALIGNED>(rowp(2 * y + 5), x0, g.cols, active, lane);
        const Raw18 m1 = load_row<ALIGNED>(rowp(2 * y + 6), x0, g.cols, active, lane);
        emit(y, qa, qb, qc, qd);
        uint32_t qe[4], qf[4];
        hpass(n0, qe);
        hpass(n1, qf);
        if (y + 1 < y_end) emit(y + 1, qc, qd, qe, qf);  // wave-uniform
#pragma unroll
        for (int i = 0; i < 4; i++) {
            qa[i] = qe[i];
            qb[i] = qf[i];
        }
        hpass(m0, qc);
        hpass(m1, qd);
        n0 = load_row<ALIGNED>(rowp(2 * y + 7), x0, g.cols, active, lane);
        n1 = load_row<ALIGNED>(rowp(2 * y + 8), x0, g.cols, active, lane);
    }
}

// Wide variant for frames whose half width is a multiple of 16 and needs no row tail (1080p, 4K): a lane owns SIXTEEN output pixels
// (two 16-byte source loads per source row, one 16-byte store per output row), so one wave covers 2048 source columns -- a whole
// 1080p row, read contiguously and with no halo loads at all (the neighbours' pixels come from the lanes next door, the image border
// is clamped) -- and keeps 128 instead of 64 bytes in flight per lane: the kernel is bound by memory latency x bytes in flight.
struct Raw34 {  // source pixels x0-1 .. x0+33 of one row
    uint32_t w[8];
    uint32_t left, right2;
};
__device__ __forceinline__ Raw34 load_row_wide(const uint8_t* __restrict__ rowp, int x0, int cols, bool active, int lane) {
    Raw34 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.w[i] = 0;
    if (active) {  // aligned, x0 + 32 <= cols by construction
        const uint4 a = *reinterpret_cast<const uint4*>(rowp + x0), b = *reinterpret_cast<const uint4*>(rowp + x0 + 16);
        r.w[0] = a.x, r.w[1] = a.y, r.w[2] = a.z, r.w[3] = a.w;
        r.w[4] = b.x, r.w[5] = b.y, r.w[6] = b.z, r.w[7] = b.w;
    }
    uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(r.w[7] >> 24), 0x138, 0xf, 0xf, false);       // wave_shr:1
    uint32_t right2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(r.w[0] & 0xffffu), 0x130, 0xf, 0xf, false);  // wave_shl:1
    if (active) {
        if (lane == 0) left = x0 > 0 ? (uint32_t)rowp[x0 - 1] : (r.w[0] & 0xffu);  // index clamp at the image border
        if (lane == 63 || x0 + 32 >= cols)
            right2 = x0 + 32 >= cols ? ((r.w[7] >> 24) * 0x0101u) : ((uint32_t)rowp[x0 + 32] | ((uint32_t)rowp[min(x0 + 33, cols - 1)] << 8));
    }
    r.left = left;
    r.right2 = right2;
    return r;
}
__device__ __forceinline__ int pxw(const Raw34& r, int k) {  // k in [-1, 33]
    if (k < 0) return (int)r.left;
    if (k >= 32) return (int)((r.right2 >> (8 * (k - 32))) & 0xff);
    return (int)((r.w[k >> 2] >> (8 * (k & 3))) & 0xff);
}
// One v_dot4_i32_i8 per horizontal-pass value (the fused kernel is bound by its vector instructions: 0.80-0.83 of the SIMD cycles busy):
// q0 = 19 (p0 + p1) - 3 (p[-1] + p2) is the dot product of the four consecutive bytes (p[-1], p0, p1, p2) with (-3, 19, 19, -3), q1 the
// same of (p1, p2, p3, p4).  The instruction's bytes are SIGNED: the pixels go in as p - 128 (one xor per word) and the accumulator starts
// at 128 x (the weights' sum 32) = 4096.  The byte quadruples are funnel shifts over neighbouring words.  Six vector instructions per
// word (xor, two v_alignbit, two dots, one v_perm to pair the results as the int16 halves the vertical pass takes) instead of ~15 with byte
// extracts and 9 on packed 16-bit lanes.  Same integers: values lie in [-1530, 9690].
__device__ __forceinline__ void hpass_wide(const Raw34& r, uint32_t q[8]) {
    uint32_t x[10];  // x[j + 1]: word j with every pixel - 128; x[0]: p[-1] in the top byte; x[9]: p[32] in the low byte
    x[0] = (r.left ^ 0x80u) << 24;
    x[9] = (r.right2 ^ 0x80u) & 0xffu;
#pragma unroll
    for (int j = 0; j < 8; j++) x[j + 1] = r.w[j] ^ 0x80808080u;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t a = __builtin_amdgcn_alignbit(x[j + 1], x[j], 24);      // (p[4j - 1], p[4j], p[4j + 1], p[4j + 2])
        const uint32_t b = __builtin_amdgcn_alignbit(x[j + 2], x[j + 1], 8);   // (p[4j + 1], p[4j + 2], p[4j + 3], p[4j + 4])
        const int q0 = __builtin_amdgcn_sdot4((int)a, (int)0xfd1313fdu, 4096, false);
        const int q1 = __builtin_amdgcn_sdot4((int)b, (int)0xfd1313fdu, 4096, false);
        q[j] = __builtin_amdgcn_perm((uint32_t)q1, (uint32_t)q0, 0x05040100u);
    }
}
#ifndef CTAG_DEC_WIDE_WAVES
#define CTAG_DEC_WIDE_WAVES 4
#endif
template <int BAND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CTAG_DEC_WIDE_WAVES))) void k_decimate_wide(const uint8_t* __restrict__ frames, ptrdiff_t frame_stride,
                                                                                                           ptrdiff_t row_stride, uint8_t* __restrict__ half, FrameGeom g,
                                                                                                           int nframes, int xblocks, int yblocks, int band_rows_rt) {
    const int band_rows = BAND ? BAND : band_rows_rt;
    int frame, idx;
    if (!map_block(blockIdx.x, xblocks * yblocks, nframes, frame, idx)) return;
    const int bx = idx % xblocks, by = idx / xblocks;
    const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const int band = by * 4 + wy;
    const int y_begin = band * band_rows;
    if (y_begin >= g.hrows) return;  // wave-uniform
    const int y_end = min(y_begin + band_rows, g.hrows);
    const int hx0 = (bx * 64 + lane) * 16;
    const int x0 = hx0 * 2;
    const bool active = hx0 < g.hcols;
    const uint8_t* __restrict__ src = frames + (ptrdiff_t)frame * frame_stride;
    uint8_t* __restrict__ dst = half + ((size_t)frame * g.hrows) * g.hp;
    const int rmax = g.rows - 1;
    auto rowp = [&](int r) { return src + (ptrdiff_t)min(max(r, 0), rmax) * row_stride; };
    uint32_t qa[8], qb[8], qc[8], qd[8];
    {
        const Raw34 ra = load_row_wide(rowp(2 * y_begin - 1), x0, g.cols, active, lane);
        const Raw34 rb = load_row_wide(rowp(2 * y_begin), x0, g.cols, active, lane);
        const Raw34 rc = load_row_wide(rowp(2 * y_begin + 1), x0, g.cols, active, lane);
        const Raw34 rd = load_row_wide(rowp(2 * y_begin + 2), x0, g.cols, active, lane);
        hpass_wide(ra, qa);
        hpass_wide(rb, qb);
        hpass_wide(rc, qc);
        hpass_wide(rd, qd);
    }
    auto emit = [&](int y, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d) {
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = vpack4(vpass2(a[2 * k], b[2 * k], c[2 * k], d[2 * k], 0), vpass2(a[2 * k + 1], b[2 * k + 1], c[2 * k + 1], d[2 * k + 1], 0));
        if (active) *reinterpret_cast<uint4*>(dst + (size_t)y * g.hp + hx0) = make_uint4(o[0], o[1], o[2], o[3]);
    };
    Raw34 n0 = load_row_wide(rowp(2 * y_begin + 3), x0, g.cols, active, lane);
    Raw34 n1 = load_row_wide(rowp(2 * y_begin + 4), x0, g.cols, active, lane);
    for (int y = y_begin; y < y_end; y += 2) {
        const Raw34 m0 = load_row_wide(rowp(2 * y + 5), x0, g.cols, active, lane);
        const Raw34 m1 = load_row_wide(rowp(2 * y + 6), x0, g.cols, active, lane);
        emit(y, qa, qb, qc, qd);
        uint32_t qe[8], qf[8];
        hpass_wide(n0, qe);
        hpass_wide(n1, qf);
        if (y + 1 < y_end) emit(y + 1, qc, qd, qe, qf);  // wave-uniform
#pragma unroll
        for (int i = 0; i < 8; i++) {
            qa[i] = qe[i];
            qb[i] = qf[i];
        }
        hpass_wide(m0, qc);
        hpass_wide(m1, qd);
        n0 = load_row_wide(rowp(2 * y + 7), x0, g.cols, active, lane);
        n1 = load_row_wide(rowp(2 * y + 8), x0, g.cols, active, lane);
    }
}

// General sizes (odd rows or cols: the scale is not exactly 2, every output column / row has its own cubic taps).
// One thread per output pixel; the tap tables come from the host (build_resize_tables in ctag_api.hip).  Same
// arithmetic as OpenCV's resizeGeneric_ for 8UC1 INTER_CUBIC: integer horizontal pass with per-tap index clamping,
// float vertical pass with round-half-even for the SIMD body (x < hcols & ~7), integer FixedPtCast for the row tail.
__global__ __launch_bounds__(256) void k_decimate_general(const uint8_t* __restrict__ frames, ptrdiff_t frame_stride, ptrdiff_t row_stride,
                                                          uint8_t* __restrict__ half, FrameGeom g, int nframes, const int32_t* __restrict__ xofs,
                                                          const int16_t* __restrict__ alpha, const int32_t* __restrict__ yofs,
                                                          const int16_t* __restrict__ beta) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, frame = blockIdx.z;
    if (x >= g.hcols || frame >= nframes) return;
    const uint8_t* __restrict__ src = frames + (ptrdiff_t)frame * frame_stride;
    const int sx = xofs[x], sy = yofs[y];
    int a[4], b[4], R[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        a[j] = alpha[x * 4 + j];
        b[j] = beta[y * 4 + j];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint8_t* S = src + (ptrdiff_t)min(max(sy - 1 + k, 0), g.rows - 1) * row_stride;
        int v = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) v += (int)S[min(max(sx - 1 + j, 0), g.cols - 1)] * a[j];
        R[k] = v;
    }
    int out;
    if (x < (g.hcols & ~(CTAG_RESIZE_SIMD_LANES - 1))) {
        const float scale = 1.f / (2048.f * 2048.f);
        float t = (float)R[3] * ((float)b[3] * scale);
        t = (float)R[2] * ((float)b[2] * scale) + t;
        t = (float)R[1] * ((float)b[1] * scale) + t;
        t = (float)R[0] * ((float)b[0] * scale) + t;
        out = (int)__builtin_rintf(t);  // nearest-even, |t| < 2^16
    } else {
        const int v = R[0] * b[0] + R[1] * b[1] + R[2] * b[2] + R[3] * b[3];
        out = (v + (1 << 21)) >> 22;
    }
    half[((size_t)frame * g.hrows + y) * g.hp + x] = (uint8_t)min(max(out, 0), 255);
}

// The fused sweep (k_decimate_mask + the mask front end of K2) takes batches of frames whose half size is a multiple of 320 x 5 (round 6; 1080p, 4K, 8K,
// 1920x1200, 1280x720, 2560x1440, 640x480 ...) with the reference's 5x5 window and 16-byte aligned rows; everything else keeps the two-kernel form with `half`.
// CTAG_FUSED_SWEEP=0 (developer aid, A/B) turns it off.
static int fuse_env() {
    static const int env_mode = getenv("CTAG_FUSED_SWEEP") ? atoi(getenv("CTAG_FUSED_SWEEP")) : -1;
    return env_mode;
}
// the frame sizes the fused sweep takes (adaptiveThresh 5; half size a multiple of 320 x 5), whatever the batch
bool sweep_fused_size(int rows, int cols, int tw, int fuse_mode) {
    const int env = fuse_mode >= 0 ? fuse_mode : fuse_env() >= 0 ? fuse_env() : 1;
    if (!env || (rows & 1) || (cols & 1) || tw != 5) return false;
    const int hcols = cols / 2, hrows = rows / 2;
    // (round 6) half width a multiple of the labelling tile (320 columns: whole mask words, whole threshold tiles, whole 16-pixel lanes), half height of whole
    // threshold-tile rows; 1080p / 4K / 8K take the compile-time-band build, the others (1920x1200, 1280x720, 2560x1440, 640x480 ...) the run-time-band one
    return hcols >= kTileW && hcols % kTileW == 0 && hrows >= 10 && hrows % 5 == 0;
}
static int fuse_bands(int hrows);
// whether a call of nframes frames of this size is a batch for the fused sweep's tall bands (the rule sweep_fused applies to gray frames): BGR calls
// below it take k_bgr2gray + the latency-tuned short-band kernels instead of one 4-wave block per frame walking 135-row bands
bool sweep_fused_batch(int rows, int cols, int nframes, int fuse_mode) {
    const int env = fuse_mode >= 0 ? fuse_mode : fuse_env() >= 0 ? fuse_env() : 1;
    return env >= 2 || (long)nframes * ((cols / 2 + kFuseCols - 1) / kFuseCols) * fuse_bands(rows / 2) >= 2048;
}
// always: BGR frames (they take this form or none: the caller has checked sweep_fused_size and the alignment)
bool sweep_fused(const uint8_t* frames, ptrdiff_t frame_stride, ptrdiff_t row_stride, int nframes, const Workspace& ws, bool always) {
    const int env = ws.fuse_mode >= 0 ? ws.fuse_mode : fuse_env() >= 0 ? fuse_env() : 1;  // CTAG_OPT_FUSED_SWEEP: 0 never, 1 batches (default), 2 whenever the frame size allows
    const FrameGeom& g = ws.g;
    if (!sweep_fused_size(g.rows, g.cols, g.tw, ws.fuse_mode)) return false;
    if ((((uintptr_t)frames | (uintptr_t)frame_stride | (uintptr_t)row_stride) & 15) != 0) return false;
    const int bands = fuse_bands(g.hrows);
    if (!always && env < 2 && (long)nframes * ((g.hcols + kFuseCols - 1) / kFuseCols) * bands < 2048) return false;  // few frames: short bands and the latency-tuned kernels (launch_decimate)
    return true;
}

hipError_t launch_zero_counters(int nframes, const Workspace& ws, hipStream_t s);
// zero_too: the chain's counters have not been zeroed (a call of a few frames): the run-time-band kernel does it, any other form gets k_zero_counters first
hipError_t launch_decimate(const uint8_t* frames, ptrdiff_t frame_stride, ptrdiff_t row_stride, int nframes, const Workspace& ws, hipStream_t s, bool fused, bool zero_too, int channels) {
    if (channels != 1 && !(channels == 3 && fused)) return hipErrorInvalidValue;  // BGR frames take the fused sweep or none (bgr_fused, ctag_api.hip)
    const FrameGeom& g = ws.g;
    const bool general = (g.rows & 1) || (g.cols & 1) || getenv("CTAG_GENERAL_RESIZE");
    ZeroList Z{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (zero_too) {
        if (fused || general || nframes > 256) {
            const hipError_t e = launch_zero_counters(nframes, ws, s);
            if (e != hipSuccess) return e;
            zero_too = false;
        } else {
            Z = ZeroList{ws.frame_ncomp, ws.frame_flags, ws.line_count, ws.clp_used, ws.ovf_count, nframes};
        }
    }
    if (fused) return launch_decimate_mask(frames, frame_stride, row_stride, nframes, ws, s, channels);
    if (general) {  // odd sizes (env: developer aid, runs even sizes through the general kernel)
        hipLaunchKernelGGL(k_decimate_general, dim3((g.hcols + 255) / 256, g.hrows, nframes), dim3(256), 0, s, frames, frame_stride, row_stride, ws.half, g,
                           nframes, ws.rz_xofs, ws.rz_alpha, ws.rz_yofs, ws.rz_beta);
        return hipGetLastError();
    }
    const int lanes = (g.hcols + 7) / 8;
    const int xblocks = (lanes + 63) / 64;
    int bands = 4 * ((g.hrows + 4 * kDecBandMax - 1) / (4 * kDecBandMax));
    // small batches (a single frame through ctag_detect_u8): a band is a sequential walk of a wave, so shorter bands -- more
    // waves -- cut the latency (one 1080p frame: 135-row bands 0.19 ms, 15-row bands 0.034 ms, 4-row bands < 0.02 ms); a full batch keeps the tall ones
    static const int min_band = getenv("CTAG_DEC_MIN_BAND") ? atoi(getenv("CTAG_DEC_MIN_BAND")) : 4;  // one frame: 15-row bands 34 us, 8 rows 21, 4 rows < 20
    while ((long)nframes * xblocks * bands < 2048 && (g.hrows + 2 * bands - 1) / (2 * bands) >= min_band) bands *= 2;
    const int band_rows = (g.hrows + bands - 1) / bands;
    const int yblocks = bands / 4;
    const int grid = grid_for(nframes, xblocks * yblocks);
    const bool aligned = (((uintptr_t)frames | (uintptr_t)frame_stride | (uintptr_t)row_stride) & 15) == 0;
#define CTAG_DEC_LAUNCH(AL, B, TL)                                                                                                         \
    hipLaunchKernelGGL((k_decimate<AL, B, TL>), dim3(grid), dim3(256), 0, s, frames, frame_stride, row_stride, ws.half, g, nframes, xblocks, \
                       yblocks, band_rows, Z)
    const bool has_tail = (g.hcols & (CTAG_RESIZE_SIMD_LANES - 1)) != 0;
    static const int wide_env = getenv("CTAG_DEC_WIDE") ? atoi(getenv("CTAG_DEC_WIDE")) : 1;  // same-box A/B on 4096 1080p frames: 2.276 -> 2.209 ms (5 waves per SIMD: 2.28)
    if (zero_too && band_rows == 135) {  // (a few frames never get here: their bands are short) the forms below do not zero
        const hipError_t e = launch_zero_counters(nframes, ws, s);
        if (e != hipSuccess) return e;
    }
    if (wide_env && aligned && !has_tail && (g.hcols & 15) == 0 && band_rows == 135) {  // a lane owns 16 output pixels: a wave spans 1024 half-res columns
        const int xb = (g.hcols / 16 + 63) / 64;
        hipLaunchKernelGGL((k_decimate_wide<135>), dim3(grid_for(nframes, xb * yblocks)), dim3(256), 0, s, frames, frame_stride, row_stride, ws.half, g, nframes, xb, yblocks,
                           band_rows);
    } else if (aligned && band_rows == 135 && !has_tail)  // 1080p and 4K frames
        CTAG_DEC_LAUNCH(true, 135, false);
    else if (aligned)
        CTAG_DEC_LAUNCH(true, 0, true);
    else
        CTAG_DEC_LAUNCH(false, 0, true);
#undef CTAG_DEC_LAUNCH
    return hipGetLastError();
}

// =====================================================================================================
// K2: adaptive threshold + connected components inside one 320x30 tile.
// =====================================================================================================
struct SweepPtrs {
    const uint8_t* half;
    const uint8_t* mask;   // k_decimate_mask's 1 bit per half-resolution pixel, rows of hcols / 8 bytes (the fused sweep: `half` is not written then)
    uint16_t* labels;
    int32_t* tile_base;
    int32_t* tile_dirty;   // [F][tiles] bit b: label block b of the tile (8 rows x 64 columns) holds label pixels that are not all zero.  The label image
                           // starts out zeroed and zeros are not written over zeros: background is most of a frame, and its 2 bytes per pixel were most of K2's traffic
    int32_t* frame_ncomp;
    uint32_t* frame_flags;
    uint32_t* parent;
    int32_t* root_of;
    int32_t* area;
    int32_t* xmin;
    int32_t* ymin;
    int32_t* xmax;
    int32_t* ymax;
    int32_t* key;
    int32_t* pool_tile;    // tile of every pool entry (K2)
    int32_t* member_head;  // per root: head of the list of its non-root members (K4), -1 = none
    int32_t* member_next;
    int32_t* ncand;
    int32_t* nroots;
    Candidate* cand;
    int cand_cap;          // candidates per frame the workspace holds
    int2* cand_scratch;    // [F][cand_cap] (pool index, key) of a frame with more candidates than k_candidates sorts in LDS (cand_aux, unused until K6)
    int32_t* ovf_count;    // tiles the LDS-sized pass could not finish (too many runs / components, or the frame's pool filled up) ...
    int32_t* ovf_list;     // ... as frame * tiles_per_frame + tile: k_threshold_ccl_big takes them
    unsigned long long* stamps;  // developer aid (CTAG_CCL_STAMPS=1): cycles per phase of k_threshold_ccl, else null
    size_t pool_stride;    // bytes between consecutive pool arrays (parent, root_of, area, xmin, ymin, xmax, ymax, key, pool_tile, member_head, member_next)
};
static SweepPtrs sweep_ptrs(const Workspace& ws) {
    return SweepPtrs{ws.half, ws.half, ws.labels, ws.tile_base, ws.tile_dirty, ws.frame_ncomp, ws.frame_flags, ws.parent, ws.root_of,
                     ws.area, ws.xmin, ws.ymin, ws.xmax, ws.ymax, ws.key, ws.pool_tile, ws.member_head, ws.member_next, ws.ncand, ws.nroots, ws.cand, ws.cand_cap,
                     reinterpret_cast<int2*>(ws.cand_aux), ws.ovf_count, ws.ovf_list, nullptr,
                     (size_t)(reinterpret_cast<const char*>(ws.root_of) - reinterpret_cast<const char*>(ws.parent))};
}

struct CclLdsLayout {
    int rp, rh;       // half-res staging region pitch / rows
    int ec, er;       // extrema grid (with ring)
    int tc, tr;       // threshold grid (tiles overlapping the CCL tile)
    size_t off_region, off_ext, off_thr, off_mask, off_start, off_runbase, off_parent, off_lab, off_misc, off_stats, off_final, total;
};
__host__ __device__ inline CclLdsLayout ccl_layout(int tw, int run_cap = kRunCap, int slot_cap = kSlotCap, bool big = false) {
    CclLdsLayout L;
    // a tile starts at an arbitrary offset inside a threshold tile: span <= tile + 2*(tw-1) + 2*tw (ring)
    L.rp = ((kTileW + 4 * tw + 32) + 15) & ~15;
    L.rh = kTileH + 4 * tw;
    L.tc = (kTileW + tw - 1) / tw + 1;
    L.tr = (kTileH + tw - 1) / tw + 1;
    L.ec = L.tc + 2;
    L.er = L.tr + 2;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o = (o + bytes + 15) & ~(size_t)15;
        return at;
    };
    const size_t stats = (size_t)slot_cap * 5 * sizeof(int);  // area, xmin, xmax, row mask (-> ymin, ymax), key
    const size_t ext = (size_t)L.ec * L.er * 2;
    const size_t thr = (size_t)L.tr * (kTileW + 8);  // per threshold-tile row: one threshold byte per tile column
    if (tw == 5) {
        // Compact layout of the 5x5 front end: 19,968 bytes, 8 blocks per CU.  Lifetimes: column extrema + tile extrema
        // live until the thresholds exist (S3), the run parents from S5 to the end of S8, the per-slot statistics from
        // then on -- all three share one region; the threshold bytes are dead before the run labels are first written.
        const size_t parent = (size_t)run_cap * 4, vbuf = (size_t)2 * 8 * 352;
        size_t front = vbuf + ((ext + 15) & ~(size_t)15);
        front = front > parent ? front : parent;
        L.off_stats = take(front > stats ? front : stats);
        L.off_parent = L.off_stats;
        L.off_region = L.off_stats;
        L.off_ext = L.off_region + vbuf;
        L.off_mask = take((size_t)kTileH * kTileWords * 8);
        L.off_start = take((size_t)kTileH * kTileWords * 8);
        L.off_runbase = take(((size_t)kTileH * kTileWords + 1) * 4);
        const size_t lab = (size_t)run_cap * 2;
        L.off_lab = take(lab > thr ? lab : thr);
        L.off_thr = L.off_lab;
        L.off_misc = take(64);
        L.off_final = take(big ? (size_t)slot_cap * 2 : 0);
        L.total = o;
        return L;
    }
    const size_t region = (size_t)L.rp * L.rh;
    L.off_region = take(region > stats ? region : stats);
    L.off_stats = L.off_region;
    L.off_ext = take(ext);
    L.off_thr = take(thr);
    L.off_mask = take((size_t)kTileH * kTileWords * 8);
    L.off_start = take((size_t)kTileH * kTileWords * 8);
    L.off_runbase = take(((size_t)kTileH * kTileWords + 1) * 4);
    L.off_parent = take((size_t)run_cap * 4);
    L.off_lab = take((size_t)run_cap * 2);
    L.off_misc = take(64);
    L.off_final = take(big ? (size_t)slot_cap * 2 : 0);
    L.total = o;
    return L;
}
static_assert(kTileH <= 32, "the per-slot row mask is one 32-bit word");
size_t threshold_ccl_lds_bytes(int tw) { return ccl_layout(tw).total; }

__device__ __forceinline__ uint64_t mask_le(int b) { return b >= 63 ? ~0ull : ((1ull << (b + 1)) - 1ull); }
// set bits of x at positions <= b (0 <= b <= 63): the higher ones are shifted out -- a shift and the count instead of building the mask first
__device__ __forceinline__ int popc_le(uint64_t x, int b) { return __popcll(x << (63 - b)); }

// Barrier for k_threshold_ccl: its phases exchange data through LDS only, so the barrier waits for LDS traffic
// (lgkmcnt) and NOT for vector memory (vmcnt) -- __syncthreads() would drain the next tile's prefetch loads and the
// label stores at every phase boundary.
#define CCL_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
constexpr int kCclThreads = 256;  // threads per 320x30 tile: the phases are short dependent chains, so more waves per tile
                                  // shorten every barrier-to-barrier critical path and fill the CU at the same LDS footprint
__device__ __forceinline__ int block_excl_scan(int v, int* scratch, int& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // inclusive scan of the wave by DPP: shifts by 1, 2, 4, 8 inside the rows of 16 lanes (a lane without a source adds 0), then lane 15 of every even
    // row into the odd row behind it and lane 31 into rows 2 and 3 -- six vector instructions instead of six ds_bpermute round trips
    int inc = v;
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);   // row_shr:1
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);   // row_shr:2
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);   // row_shr:4
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);   // row_shr:8
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    if (lane == 63) scratch[w] = inc;
    CCL_SYNC();
    int base = 0;
    total = 0;
#pragma unroll
    for (int i = 0; i < kCclThreads / 64; i++) {
        if (i < w) base += scratch[i];
        total += scratch[i];
    }
    CCL_SYNC();
    return base + inc - v;
}

__device__ __forceinline__ unsigned lds_find(const unsigned* parent, unsigned x) {
    unsigned p;
    while ((p = __atomic_load_n(&parent[x], __ATOMIC_RELAXED)) != x) x = p;
    return x;
}
__device__ __forceinline__ void lds_union(unsigned* parent, unsigned a, unsigned b) {
    while (true) {
        a = lds_find(parent, a);
        b = lds_find(parent, b);
        if (a == b) return;
        if (a > b) {
            const unsigned t = a;
            a = b;
            b = t;
        }
        const unsigned old = atomicMin(&parent[b], a);
        if (old == b) return;
        b = old;
    }
}

// adaptive threshold of one tile as an integer bound: pixel u is foreground iff u < T, where
// fg <=> float(u)*(1/255) < min(cap, (maxF + minF)/2), cap = 0.3f in the reference    (corner_detector.cpp:71)
__host__ __device__ __forceinline__ int threshold_bound(int mn, int mx, float cap) {
    const float k = (float)(1.0 / 255);
    const float a = ((float)mx * k + (float)mn * k) / 2;
    const float thr = a < cap ? a : cap;
    int t = (int)(thr * 255.0f);
    t = t < 0 ? 0 : (t > 256 ? 256 : t);
    while (t < 256 && (float)t * k < thr) t++;
    while (t > 0 && !((float)(t - 1) * k < thr)) t--;
    return t;
}
// The bound is tcap (77 for the reference's 0.3 cap) once mn + mx >= dim (154 there) and otherwise depends on both operands (the two
// products round separately): a dim x dim byte table, built on the host with the function above when a handle is created, replaces
// the float search in the kernel (most threshold tiles are bright and never touch it).  K2's packed-byte compares need tcap < 128.
bool build_threshold_table(float dark_cap, uint8_t* table, int* dim_out, int* tcap_out) {
    if (!(dark_cap > 0.f) || !(dark_cap < 0.5f)) return false;
    const int tcap = threshold_bound(255, 255, dark_cap);  // the bound of a tile brighter than the cap
    if (tcap < 1 || tcap > 127) return false;
    int dim = 1;
    for (int mn = 0; mn < 256; mn++)
        for (int mx = mn; mx < 256; mx++)
            if (threshold_bound(mn, mx, dark_cap) != tcap) dim = mn + mx + 1 > dim ? mn + mx + 1 : dim;
    if (dim > 256) return false;
    for (int mn = 0; mn < dim; mn++)
        for (int mx = 0; mx < dim; mx++) table[mn * dim + mx] = (uint8_t)threshold_bound(mn, mx, dark_cap);
    *dim_out = dim;
    *tcap_out = tcap;
    return true;
}
__device__ __forceinline__ int threshold_lookup(int mn, int mx, const KParams& kp) { return mn + mx >= kp.thr_dim ? kp.tcap : (int)kp.thr_table[mn * kp.thr_dim + mx]; }

// geometry of one CCL tile and of the threshold tiles / pixels it needs
struct TileRegion {
    int frame, tile, tx0, ty0, tw_eff, th_eff;
    int tcs0, tcs1, trs0, trs1, tc0, tc1, tr0, tr1, px0, px1, py0, py1, lx0, chunks, nrows;
};
__device__ __forceinline__ TileRegion tile_region(int frame, int tile, const FrameGeom& g, int tw) {
    TileRegion t;
    t.frame = frame;
    t.tile = tile;
    const int tix = tile % g.tiles_x, tiy = tile / g.tiles_x;
    t.tx0 = tix * kTileW;
    t.ty0 = tiy * kTileH;
    t.tw_eff = min(kTileW, g.hcols - t.tx0);
    t.th_eff = min(kTileH, g.hrows - t.ty0);
    // threshold tiles this CCL tile needs: its own plus a one-tile ring
    t.tcs0 = t.tx0 / tw;
    t.tcs1 = (t.tx0 + t.tw_eff - 1) / tw;
    t.trs0 = t.ty0 / tw;
    t.trs1 = (t.ty0 + t.th_eff - 1) / tw;
    t.tc0 = max(t.tcs0 - 1, 0);
    t.tc1 = min(t.tcs1 + 1, g.tcols - 1);
    t.tr0 = max(t.trs0 - 1, 0);
    t.tr1 = min(t.trs1 + 1, g.trows - 1);
    t.px0 = t.tc0 * tw;
    t.px1 = min((t.tc1 + 1) * tw, g.hcols);
    t.py0 = t.tr0 * tw;
    t.py1 = min((t.tr1 + 1) * tw, g.hrows);
    t.lx0 = t.px0 & ~15;
    t.chunks = (t.px1 - t.lx0 + 15) >> 4;
    t.nrows = t.py1 - t.py0;
    return t;
}

// packed 16-bit min / max (v_pk_min_u16 / v_pk_max_u16): bytes are split into even / odd halves for them
typedef unsigned short ctag_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    ctag_us2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    const ctag_us2 r = __builtin_elementwise_min(x, y);
    uint32_t o;
    __builtin_memcpy(&o, &r, 4);
    return o;
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    ctag_us2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    const ctag_us2 r = __builtin_elementwise_max(x, y);
    uint32_t o;
    __builtin_memcpy(&o, &r, 4);
    return o;
}
// four pixel < threshold tests on packed bytes -> 4 bits.  T <= 77 < 128: with the pixel's top bit handled separately the
// per-byte subtract (0x80 | low7) - T never borrows across bytes and its bit 7 says low7 >= T; one multiply gathers the bits.
__device__ __forceinline__ uint32_t lt4_bytes(uint32_t u, uint32_t t) {
    const uint32_t ge = (((u & 0x7f7f7f7fu) | 0x80808080u) - t) & 0x80808080u;
    const uint32_t lt = ~(ge | u) & 0x80808080u;
    return (((lt >> 7) * 0x01020408u) >> 24) & 0xfu;
}
// exact n / d for n < 1024, d <= 66 from the 16-bit reciprocal m = ceil(65536 / d):  n * (m d - 65536) < 65536
__device__ __forceinline__ int div_small(int n, int m16) { return (int)(((unsigned)n * (unsigned)m16) >> 16); }
__device__ __forceinline__ int recip16(int d) { return (65536 + d - 1) / d; }
constexpr int kVPitch = 352;  // column extrema per threshold-tile row: 5 + 320 + 5 columns from an 8-aligned start

// =====================================================================================================
// K1 + the front half of K2 in one kernel (batches of 1080p-class frames, 5x5 window): decimate AND threshold, 1 bit per pixel out.
// k_decimate_wide holds every half-resolution pixel in registers with half of its vector slots idle, and K2 spent half of its
// cycles fetching those pixels back (its front end: stage, 5x5 extrema, 3x3 dilation, compare).  Here a wave slides down its band
// like k_decimate_wide, and per threshold-tile row (5 output rows) it
//   * keeps the rows' pixels in an LDS ring (10 rows) and their per-column min / max in registers,
//   * at the end of the tile row publishes the column extrema, reduces them to tile extrema (5 columns each; a ring of 3 tile rows),
//   * and can then finish the PREVIOUS tile row: 3x3 min-of-min / max-of-max (corner_detector.cpp:54-67, interior tiles only: B1),
//     the threshold bound (threshold_lookup, :71), pixel < bound on packed bytes, 16 mask bits per lane and row.
// The half-resolution image is never written (518 KB written + read per 1080p frame); the mask is 65 KB.  A band needs the
// extrema of the tile rows just above and below it: those 2 x 5 output rows are decimated again (extrema only: +7 % source reads,
// which the neighbouring band's wave reads at about the same time on the same XCD).  A wave covers 960 half-resolution columns --
// 60 lanes x 16 pixels, a multiple of the 5-pixel tile -- and lanes 60 / 61 decimate the 16 columns right / left of the span for
// the tile column just outside it (4K frames: two waves per row).  BAND = 135: frames with hcols % 960 == 0, hrows % 540 == 0 (1080p, 4K, 8K); BAND = 0: see below.
// =====================================================================================================
struct FuseLds {                   // per wave: 12 960 bytes, four waves per block, three blocks per CU
    uint4 ring[10][kFuseLanes];    // the pixels of the tile row being built and of the one waiting for its lower neighbour
    uint8_t vmin[16 + kFuseCols + 16], vmax[16 + kFuseCols + 16];  // column extrema of the finished tile row: [16 + column - X0]
    uint16_t ext[3][kFuseTiles + 4];  // tile extrema min | max << 8 of three tile rows: tile k = -1 .. 192 of the span at [k + 1]
    uint8_t tt[kFuseTiles];        // threshold bound of the tiles of the row being emitted
};
// A source row as the lane loaded it: 32 pixels, nothing else.  The neighbours' pixels the horizontal pass needs (one to the left, two
// to the right) are fetched from the lanes next door when the row is CONSUMED (wave_shr / wave_shl DPP moves, no LDS): a shuffle at
// load time makes the wave wait for every row it has just requested, which leaves one row in flight per wave (k_decimate_wide does
// that and leans on four waves per SIMD; this kernel has three and keeps four rows in flight instead).  No byte loads either: at
// the image border the edge pixel repeats (index clamp), at the seam between two waves of a 4K row the halo lanes hold the pixels.
struct Raw32 {
    uint32_t w[8];
};
__device__ __forceinline__ Raw32 load_row_fuse(const uint8_t* __restrict__ rowp, int x0) {  // every lane loads (inactive ones an in-row dummy): no branch around the loads
    Raw32 r;
    const uint4 a = *reinterpret_cast<const uint4*>(rowp + x0), b = *reinterpret_cast<const uint4*>(rowp + x0 + 16);
    r.w[0] = a.x, r.w[1] = a.y, r.w[2] = a.z, r.w[3] = a.w;
    r.w[4] = b.x, r.w[5] = b.y, r.w[6] = b.z, r.w[7] = b.w;
    return r;
}
// The same 32 pixels of a BGR row (CH == 3: frames handed over as the camera delivers them, main.cpp:52-54): 96 bytes per lane, converted to gray -- OpenCV's
// fixed-point cvtColor(BGR2GRAY), gray4_of -- when the row is consumed; the gray image is never written (ctag_detect_batch_bgr8_device, bgr_fused)
template <int CH>
struct RawSrc {
    uint32_t w[8 * CH];
};
// (measured, round 5: nontemporal loads here -- which stream 7.0 instead of 6.2 TB/s in tools/ubench/read_bw -- make K1f SLOWER, 1.71-1.86 against 1.68 ms per 4096
// 1080p frames and 2.1-2.3 against 1.79 ms per 1024 4K frames: the tile rows a band shares with its neighbour then come from HBM twice)
template <int CH>
__device__ __forceinline__ RawSrc<CH> load_row_src(const uint8_t* __restrict__ rowp, int x0) {  // every lane loads (inactive ones an in-row dummy): no branch around the loads
    RawSrc<CH> r;
#pragma unroll
    for (int i = 0; i < 2 * CH; i++) {
        const uint4 a = *reinterpret_cast<const uint4*>(rowp + (ptrdiff_t)CH * x0 + 16 * i);
        r.w[4 * i] = a.x, r.w[4 * i + 1] = a.y, r.w[4 * i + 2] = a.z, r.w[4 * i + 3] = a.w;
    }
    return r;
}
template <int CH>
__device__ __forceinline__ Raw32 gray_row(const RawSrc<CH>& r) {
    Raw32 g;
#pragma unroll
    for (int i = 0; i < 8; i++) g.w[i] = CH == 3 ? gray4_of(r.w[(3 * i) % (8 * CH)], r.w[(3 * i + 1) % (8 * CH)], r.w[(3 * i + 2) % (8 * CH)]) : r.w[i % (8 * CH)];
    return g;
}
// lane i <- lane i - 1 / lane i + 1 across the whole wave (DPP wave_shr:1 / wave_shl:1)
__device__ __forceinline__ uint32_t wave_from_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t wave_from_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); }
// left_edge: lane 0 is at the image's left border; right_edge: lane 59 at its right border (else lanes 61 / 60 hold the pixels beyond)
__device__ __forceinline__ void hpass_fuse(const Raw32& r, int lane, bool left_edge, bool right_edge, uint32_t q[8], int last_lane = kFuseLanes - 1) {
    Raw34 t;
#pragma unroll
    for (int i = 0; i < 8; i++) t.w[i] = r.w[i];
    uint32_t left = wave_from_prev(r.w[7] >> 24);
    uint32_t right2 = wave_from_next(r.w[0] & 0xffffu);
    const uint32_t last61 = (uint32_t)__builtin_amdgcn_readlane((int)(r.w[7] >> 24), 61), first0 = (uint32_t)__builtin_amdgcn_readlane((int)(r.w[0] & 0xffffu), 0);
    if (lane == 0) left = left_edge ? (r.w[0] & 0xffu) : last61;
    if (lane == last_lane && right_edge) right2 = (r.w[7] >> 24) * 0x0101u;
    if (lane == 61) right2 = first0;  // (the halo lanes' outer neighbours only reach pixels nobody reads)
    t.left = left;
    t.right2 = right2;
    hpass_wide(t, q);
}
// BAND = 0 (round 6): any frame whose half size is a multiple of 320 x 5 -- 1920x1200 (the reference's test.bmp), 1280x720, 2560x1440, 640x480 ... -- the bands are
// whole threshold-tile rows handed out evenly at run time (`nbands` of them; band b = tile rows [b trows / nbands, (b + 1) trows / nbands)), and the last wave of a row
// may cover fewer than 960 columns (`nact` lanes of 16 pixels).  BAND = 135 keeps 1080p / 4K / 8K on compile-time bands and full waves.
template <int BAND, int WAVES, int CH = 1>  // CH = 3: BGR frames (four source rows of 96 bytes per lane in flight: two waves per SIMD hold them)
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(CH == 3 ? 2 : 3, CH == 3 ? 2 : 3))) void k_decimate_mask(const uint8_t* __restrict__ frames, ptrdiff_t frame_stride, ptrdiff_t row_stride,
                                                                                                    uint8_t* __restrict__ mask, FrameGeom g, KParams kp, int nframes, int xblocks,
                                                                                                    int yblocks, int nbands) {
    static_assert(BAND % 5 == 0, "a band is whole threshold-tile rows");
    constexpr bool GEN = BAND == 0;
    __shared__ FuseLds S4[WAVES];
    int frame, idx;
    if (!map_block(blockIdx.x, xblocks * yblocks, nframes, frame, idx)) return;
    const int bx = idx % xblocks, by = idx / xblocks;
    const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    // Odd bands run UPWARD: such a wave works on the vertically mirrored frame (source row r <-> rows - 1 - r, output row y <-> hrows - 1 - y;
    // the 2x cubic taps, the 5-row tiles and the border rules are symmetric under that mirror) and so walks its band from the bottom.  Bands 2k and
    // 2k + 1 then end at their common border at the same time, bands 2k + 1 and 2k + 2 start at theirs together: the tile rows a band decimates a
    // second time for its neighbour's extrema (+7 % source rows) are read by both waves within microseconds of each other -- from the cache, not twice
    // from HBM (the four waves of a block are the four bands of a frame: same CU).
#ifndef CTAG_FUSE_MIRROR
#define CTAG_FUSE_MIRROR 1
#endif
    const int band_actual = by * WAVES + wy;
    if (GEN ? band_actual >= nbands : band_actual * BAND >= g.hrows) return;  // wave-uniform
    const bool mirror = CTAG_FUSE_MIRROR && (band_actual & 1);
    int y_begin, y_end;  // the band's rows in the (mirrored) frame it walks downward
    if constexpr (GEN) {
        const int r0 = 5 * (int)(((long)band_actual * g.trows) / nbands), r1 = 5 * (int)(((long)(band_actual + 1) * g.trows) / nbands);  // in the frame itself
        if (r0 == r1) return;  // fewer tile rows than bands (tiny frames)
        y_begin = mirror ? g.hrows - r1 : r0;
        y_end = mirror ? g.hrows - r0 : r1;
    } else {
        const int band = mirror ? g.hrows / BAND - 1 - band_actual : band_actual;
        y_begin = band * BAND;
        y_end = min(y_begin + BAND, g.hrows);
    }
    FuseLds& S = S4[wy];
    const int X0 = bx * kFuseCols;
    const int nact = GEN ? min(kFuseLanes, (g.hcols - X0) >> 4) : kFuseLanes;  // lanes of this wave that own 16 columns of the frame
    const int hx0 = lane < kFuseLanes ? X0 + 16 * lane : lane == 60 ? X0 + kFuseCols : X0 - 16;
    const bool active = lane < nact || (lane == 60 && X0 + kFuseCols < g.hcols) || (lane == 61 && X0 > 0);
    const int x0 = active ? hx0 * 2 : 0;  // inactive lanes load the row's first bytes and drop them
    const bool left_edge = X0 == 0, right_edge = X0 + kFuseCols >= g.hcols;
    const uint8_t* __restrict__ src = frames + (ptrdiff_t)frame * frame_stride;
    const int mpitch = g.hcols >> 3;
    uint8_t* __restrict__ mrow0 = mask + (size_t)frame * g.hrows * mpitch + (X0 >> 3) + 2 * lane;
    const int rmax = g.rows - 1;
    auto rowp = [&](int r) __attribute__((always_inline)) {
        const int rr = min(max(r, 0), rmax);
        return src + (ptrdiff_t)(mirror ? rmax - rr : rr) * row_stride;
    };
    // the lane's 16 pixels lie in the tiles j0 .. j0 + 3 of the span; sel[k] picks, for pixels 4k .. 4k + 3, their tile's byte of a packed word
    const int j0 = (16 * lane) / 5;
    uint32_t sel[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) v |= (uint32_t)((16 * lane + 4 * k + q) / 5 - j0) << (8 * q);
        sel[k] = v;
    }
    const int ys = max(y_begin - 5, 0), ye = min(y_end + 5, g.hrows);  // the band plus one tile row above and below (extrema only)
    const int e_lo = y_begin / 5, e_hi = y_end / 5;                    // tile rows this wave emits: [e_lo, e_hi)
    const int tc0 = X0 / 5;
    uint32_t mnE[4], mnO[4], mxE[4], mxO[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        mnE[k] = mnO[k] = 0x00ff00ffu;
        mxE[k] = mxO[k] = 0u;
    }
    int rit = 0, slot = ys % 10;
    // one wave, LDS only: its accesses are served in order, so a wait for LDS (NOT for the source rows in flight) orders them
    auto wave_sync = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    auto tile_row_done = [&](int gt) __attribute__((always_inline)) {  // the five rows of threshold-tile row gt are in the ring, their column extrema in registers
        if (active) {
            const uint4 mn = make_uint4(mnE[0] | (mnO[0] << 8), mnE[1] | (mnO[1] << 8), mnE[2] | (mnO[2] << 8), mnE[3] | (mnO[3] << 8));
            const uint4 mx = make_uint4(mxE[0] | (mxO[0] << 8), mxE[1] | (mxO[1] << 8), mxE[2] | (mxO[2] << 8), mxE[3] | (mxO[3] << 8));
            const int at = lane < kFuseLanes ? 16 + 16 * lane : lane == 60 ? 16 + kFuseCols : 0;
            *reinterpret_cast<uint4*>(S.vmin + at) = mn;
            *reinterpret_cast<uint4*>(S.vmax + at) = mx;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            mnE[k] = mnO[k] = 0x00ff00ffu;
            mxE[k] = mxO[k] = 0u;
        }
        wave_sync();
        const int r3 = gt % 3;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int j = lane + 64 * q;  // tile j - 1 of the span
            if (j < kFuseTiles + 2) {
                const uint8_t* a = S.vmin + 11 + 5 * j;
                const uint8_t* b = S.vmax + 11 + 5 * j;
                const int mn = min(min(min((int)a[0], (int)a[1]), min((int)a[2], (int)a[3])), (int)a[4]);
                const int mx = max(max(max((int)b[0], (int)b[1]), max((int)b[2], (int)b[3])), (int)b[4]);
                S.ext[r3][j] = (uint16_t)(mn | (mx << 8));
            }
        }
        wave_sync();
    };
    auto emit_tile_row = [&](int e) __attribute__((always_inline)) {  // tile row e's pixels are in the ring and the tile extrema of rows e - 1, e, e + 1 (as far as they exist) in `ext`
        if (e < e_lo || e >= e_hi) return;  // wave-uniform
        const int ra = (e + 2) % 3, rb = e % 3, rc = (e + 1) % 3;  // rows e - 1, e, e + 1 of the ring
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int j = lane + 64 * q;
            if (j < kFuseTiles) {
                int T = 0;
                const int tc = tc0 + j;
                // 3x3 min-of-min / max-of-max for interior tiles, zero elsewhere (corner_detector.cpp:54-67, B1); a tile whose own minimum
                // is >= tcap has no pixel below any bound (T <= tcap)
                if (e >= 1 && e <= g.trows - 2 && tc >= 1 && tc <= g.tcols - 2 && (S.ext[rb][j + 1] & 0xff) < kp.tcap) {
                    int mn = 255, mx = 0;
#pragma unroll
                    for (int dx = 0; dx < 3; dx++) {
                        const int ea = S.ext[ra][j + dx], eb = S.ext[rb][j + dx], ec = S.ext[rc][j + dx];
                        mn = min(mn, min(min(ea & 0xff, eb & 0xff), ec & 0xff));
                        mx = max(mx, max(max(ea >> 8, eb >> 8), ec >> 8));
                    }
                    // the bound from its definition, not from the table K2 reads: a global load inside this branch would make the wave
                    // wait for every source row it has in flight at each tile row (the table is built from the same function)
                    T = mn + mx >= kp.thr_dim ? kp.tcap : threshold_bound(mn, mx, kp.dark_cap);
                }
                S.tt[j] = (uint8_t)T;
            }
        }
        wave_sync();
        if (lane < nact) {
            const uint32_t T4 = (uint32_t)S.tt[j0] | ((uint32_t)S.tt[j0 + 1] << 8) | ((uint32_t)S.tt[j0 + 2] << 16) | ((uint32_t)S.tt[min(j0 + 3, kFuseTiles - 1)] << 24);
            // (mirrored wave: its row 5 e + r is row hrows - 1 - (5 e + r) of the frame)
            uint8_t* mp = mrow0 + (size_t)(mirror ? g.hrows - 1 - 5 * e : 5 * e) * mpitch;
            const ptrdiff_t mstep = mirror ? -(ptrdiff_t)mpitch : (ptrdiff_t)mpitch;
            if (T4 == 0u) {
#pragma unroll
                for (int r = 0; r < 5; r++) *reinterpret_cast<uint16_t*>(mp + r * mstep) = (uint16_t)0;
            } else {
                const uint32_t t0 = __builtin_amdgcn_perm(0u, T4, sel[0]), t1 = __builtin_amdgcn_perm(0u, T4, sel[1]), t2 = __builtin_amdgcn_perm(0u, T4, sel[2]),
                               t3 = __builtin_amdgcn_perm(0u, T4, sel[3]);
                const int s0 = (e & 1) * 5;  // 5 e mod 10
#pragma unroll
                for (int r = 0; r < 5; r++) {
                    const uint4 px = S.ring[s0 + r][lane];
                    const uint32_t bits = lt4_bytes(px.x, t0) | (lt4_bytes(px.y, t1) << 4) | (lt4_bytes(px.z, t2) << 8) | (lt4_bytes(px.w, t3) << 12);
                    *reinterpret_cast<uint16_t*>(mp + r * mstep) = (uint16_t)bits;
                }
            }
        }
        // (the ring slots just read are overwritten two tile rows from now; LDS serves a wave's accesses in order)
    };
    auto row_done = [&](int y, const uint32_t (&o)[4]) __attribute__((always_inline)) {
        if (lane < kFuseLanes) S.ring[slot][lane] = make_uint4(o[0], o[1], o[2], o[3]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t e = o[k] & 0x00ff00ffu, od = (o[k] >> 8) & 0x00ff00ffu;
            mnE[k] = pk_min_u16(mnE[k], e);
            mnO[k] = pk_min_u16(mnO[k], od);
            mxE[k] = pk_max_u16(mxE[k], e);
            mxO[k] = pk_max_u16(mxO[k], od);
        }
        slot = slot == 9 ? 0 : slot + 1;
        if (++rit == 5) {
            rit = 0;
            tile_row_done(y / 5);
            emit_tile_row(y / 5 - 1);  // its lower neighbour is known now
        }
    };
    uint32_t qa[8], qb[8], qc[8], qd[8];
    using Src = RawSrc<CH>;
    auto ld = [&](int r) __attribute__((always_inline)) { return load_row_src<CH>(rowp(r), x0); };
    auto hp = [&](const Src& r, uint32_t (&q)[8]) __attribute__((always_inline)) { hpass_fuse(gray_row<CH>(r), lane, left_edge, right_edge, q, nact - 1); };
    {
        const Src ra = ld(2 * ys - 1);
        const Src rb = ld(2 * ys);
        const Src rc = ld(2 * ys + 1);
        const Src rd = ld(2 * ys + 2);
        hp(ra, qa);
        hp(rb, qb);
        hp(rc, qc);
        hp(rd, qd);
    }
    auto emit = [&](int y, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d) __attribute__((always_inline)) {
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = vpack4(vpass2(a[2 * k], b[2 * k], c[2 * k], d[2 * k], 0), vpass2(a[2 * k + 1], b[2 * k + 1], c[2 * k + 1], d[2 * k + 1], 0));
        row_done(y, o);
    };
    Src n0 = ld(2 * ys + 3);
    Src n1 = ld(2 * ys + 4);
    for (int y = ys; y < ye; y += 2) {
        const Src m0 = ld(2 * y + 5);
        const Src m1 = ld(2 * y + 6);
        emit(y, qa, qb, qc, qd);
        uint32_t qe[8], qf[8];
        hp(n0, qe);
        hp(n1, qf);
        if (y + 1 < ye) emit(y + 1, qc, qd, qe, qf);  // wave-uniform
#pragma unroll
        for (int i = 0; i < 8; i++) {
            qa[i] = qe[i];
            qb[i] = qf[i];
        }
        n0 = ld(2 * y + 7);
        n1 = ld(2 * y + 8);
        hp(m0, qc);
        hp(m1, qd);
    }
    if (ye == g.hrows) emit_tile_row(g.trows - 1);  // the frame's last tile row has no lower neighbour: a border row (bound 0)
}

// bands of the general form: a multiple of four (the four waves of a block), about 90-150 rows each; reproduces the 135-row bands of 1080p / 4K / 8K
static int fuse_bands(int hrows) { return 4 * ((hrows + 599) / 600); }
static bool fuse_exact(const FrameGeom& g) { return g.hcols % kFuseCols == 0 && g.hrows % 135 == 0 && (g.hrows / 135) % 4 == 0; }
static hipError_t launch_decimate_mask(const uint8_t* frames, ptrdiff_t frame_stride, ptrdiff_t row_stride, int nframes, const Workspace& ws, hipStream_t s, int channels) {
    const FrameGeom& g = ws.g;
    const int xb = (g.hcols + kFuseCols - 1) / kFuseCols;
    static const int gen_env = getenv("CTAG_FUSE_GENERAL") ? atoi(getenv("CTAG_FUSE_GENERAL")) : 0;  // developer aid: 1 runs 1080p / 4K / 8K through the run-time-band build too
    if (!fuse_exact(g) || gen_env) {
        const int nb = fuse_bands(g.hrows), yblocks = nb / 4;
        if (channels == 3)
            hipLaunchKernelGGL((k_decimate_mask<0, 4, 3>), dim3(grid_for(nframes, xb * yblocks)), dim3(256), 0, s, frames, frame_stride, row_stride, ws.half, g, ws.kp, nframes, xb, yblocks, nb);
        else
            hipLaunchKernelGGL((k_decimate_mask<0, 4>), dim3(grid_for(nframes, xb * yblocks)), dim3(256), 0, s, frames, frame_stride, row_stride, ws.half, g, ws.kp, nframes, xb, yblocks, nb);
        return hipGetLastError();
    }
    if (channels == 3) {  // BGR frames: converted where they are loaded
        const int yblocks = g.hrows / 135 / 4;
        hipLaunchKernelGGL((k_decimate_mask<135, 4, 3>), dim3(grid_for(nframes, xb * yblocks)), dim3(256), 0, s, frames, frame_stride, row_stride, ws.half, g, ws.kp, nframes, xb, yblocks, 0);
        return hipGetLastError();
    }
    static const int band_env = getenv("CTAG_FUSE_BAND") ? atoi(getenv("CTAG_FUSE_BAND")) : 135;  // developer aid (A/B): 270-row bands in two-wave blocks
    if (band_env == 270 && g.hrows % 540 == 0) {
        const int yblocks = g.hrows / 270 / 2;
        hipLaunchKernelGGL((k_decimate_mask<270, 2>), dim3(grid_for(nframes, xb * yblocks)), dim3(128), 0, s, frames, frame_stride, row_stride, ws.half, g, ws.kp, nframes, xb, yblocks, 0);
    } else {
        const int yblocks = g.hrows / 135 / 4;
        hipLaunchKernelGGL((k_decimate_mask<135, 4>), dim3(grid_for(nframes, xb * yblocks)), dim3(256), 0, s, frames, frame_stride, row_stride, ws.half, g, ws.kp, nframes, xb, yblocks, 0);
    }
    return hipGetLastError();
}

// One 320x30 tile.  RUNCAP / SLOTCAP: row runs / components of the tile held in LDS.  BIG = false is the pass every tile
// takes first (2048 runs, 640 components, 8 tiles per CU); a tile that does not fit -- dense speckle, fine texture -- or
// whose components no longer fit the frame's pool is handed on through the overflow list instead of failing the frame, and
// k_threshold_ccl_big runs the same code on it with BIG = true: caps that hold ANY 320x30 tile (160 runs per row), one
// tile per CU, and only the components that can matter are published -- area >= 30 (corner_detector.cpp:88) or touching
// the tile border (they may grow by seam merging); the other specks keep a label of their own with bit 15 set, which no
// later stage ever looks up.
template <int TWC, int RUNCAP, int SLOTCAP, bool BIG, bool MASKIN = false>
__device__ __forceinline__ void ccl_tile(unsigned char* smem, const SweepPtrs& P, const FrameGeom& g, const KParams& kp, int frame0, int tile0) {
    const int tw = TWC ? TWC : g.tw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const CclLdsLayout L = ccl_layout(tw, RUNCAP, SLOTCAP, BIG);
    uint8_t* hr_s = smem + L.off_region;
    uint16_t* ext_s = reinterpret_cast<uint16_t*>(smem + L.off_ext);
    uint8_t* thr_s = smem + L.off_thr;  // [tile row][x - tx0], pitch kTileW + 8
    uint64_t* mask_s = reinterpret_cast<uint64_t*>(smem + L.off_mask);
    uint64_t* start_s = reinterpret_cast<uint64_t*>(smem + L.off_start);
    int* runbase_s = reinterpret_cast<int*>(smem + L.off_runbase);
    unsigned* parent_s = reinterpret_cast<unsigned*>(smem + L.off_parent);
    uint16_t* lab_s = reinterpret_cast<uint16_t*>(smem + L.off_lab);
    int* misc_s = reinterpret_cast<int*>(smem + L.off_misc);

    const TileRegion T = tile_region(frame0, tile0, g, tw);

    // developer aid: per-phase cycles are kept in registers and flushed once per tile (a global atomic per stamp would
    // cost as much as the phases it measures)
    unsigned long long t_prev = P.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long t_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int phase) {
        if (P.stamps && tid == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int q = 0; q < 10; q++)
                if (q == phase) t_acc[q] += t - t_prev;
            t_prev = t;
        }
    };
  {
    const int frame = T.frame, tile = T.tile, tx0 = T.tx0, ty0 = T.ty0, tw_eff = T.tw_eff, th_eff = T.th_eff;
    const int tcs0 = T.tcs0, tcs1 = T.tcs1, trs0 = T.trs0, trs1 = T.trs1, tc0 = T.tc0, tc1 = T.tc1, tr0 = T.tr0, tr1 = T.tr1;
    const int py0 = T.py0, lx0 = T.lx0;
    (void)tc1;
    (void)tr1;
    (void)py0;
    (void)lx0;
    // a tile without foreground: its labels are zero and it owns no component -- exactly what the phases below would produce
    int32_t* const dirty_p = P.tile_dirty + (size_t)frame * g.tiles_x * g.tiles_y + tile;
    const int was_dirty = *dirty_p;  // requested with the tile's first loads; block-uniform
    auto empty_tile = [&]() {
        if (was_dirty) {  // left over from the frame this slot held before: back to zeros
            uint16_t* __restrict__ limg = P.labels + ((size_t)frame * g.hrows) * g.lp;
            constexpr int groups = kTileW / 8;
            for (int i = tid; i < kTileH * groups; i += kCclThreads) {
                const int r = i / groups, gq = i - r * groups;
                if (r < th_eff && gq * 8 < tw_eff) *reinterpret_cast<uint4*>(limg + (size_t)(ty0 + r) * g.lp + tx0 + gq * 8) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        if (tid == 0) {
            P.tile_base[(size_t)frame * g.tiles_x * g.tiles_y + tile] = 0;
            if (was_dirty) *dirty_p = 0;
        }
    };
  if constexpr (MASKIN) {
    // ---- the fused sweep: k_decimate_mask thresholded the pixels where they were computed; the row masks arrive as 64-bit words
    // (hcols is a multiple of the tile width there, rows of hcols / 8 bytes: every word is aligned and inside the frame's columns)
    (void)tcs0, (void)tcs1, (void)trs0, (void)trs1, (void)tc0, (void)tr0, (void)hr_s, (void)ext_s, (void)thr_s;
    uint64_t m = 0ull;
    if (tid < kTileH * kTileWords) {
        const int r = tid / kTileWords, w = tid - r * kTileWords;
        const int mp = g.hcols >> 3;
        if (r < th_eff) m = *reinterpret_cast<const uint64_t*>(P.mask + ((size_t)frame * g.hrows + ty0 + r) * mp + (tx0 >> 3) + 8 * w);
        mask_s[tid] = m;
    }
    {
        const unsigned long long any = __ballot(m != 0ull);
        if (lane == 0) misc_s[12 + wave] = any != 0ull ? 1 : 0;
    }
    CCL_SYNC();
    stamp(0);
    if ((misc_s[12] | misc_s[13] | misc_s[14] | misc_s[15]) == 0) {
        empty_tile();
        return;
    }
    stamp(3);
  } else if constexpr (TWC == 5) {
    // ---- front end for the 5x5 window: no LDS staging.  An item is (threshold-tile row, 8-pixel column group); the thread
    // loads its 5 x 8 pixels straight into registers, reduces them vertically on packed bytes and leaves one min / max per
    // COLUMN in LDS; a thread per threshold tile then combines 5 columns.  The pixels stay in registers for the compare.
    const uint8_t* __restrict__ himg = P.half + ((size_t)frame * g.hrows) * g.hp;
    uint8_t* vmin_s = hr_s;                          // [<= 8][kVPitch]
    uint8_t* vmax_s = hr_s + 8 * kVPitch;
    const int lx8 = T.px0 & ~7;
    // groups run to the END of the last threshold tile, not to the frame edge: a partial last tile reads 5 column extrema,
    // and the columns past the frame must hold the neutral values rather than stale LDS
    const int ng = ((tc1 + 1) * 5 - lx8 + 7) >> 3;   // <= 43
    const int ncr = tr1 - tr0 + 1;                   // <= 8
    const int nitems = ng * ncr;
    const int m_ng = recip16(ng);  // wave-uniform: one division instead of one per item
    uint2 pix[2][5];
    int it_cr[2] = {-1, -1}, it_g[2] = {0, 0};
    bool dark = false;  // one of this thread's pixels INSIDE the tile is below 77
    bool item_dark[2] = {false, false};  // ... one of the item's 5 x 8 pixels is
    if (tid < kTileH * kTileWords) mask_s[tid] = 0ull;  // rows / groups outside the frame stay background
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int i = tid + q * kCclThreads;
        if (i < nitems) {
            const int cr = div_small(i, m_ng), gq = i - cr * ng;
            it_cr[q] = cr;
            it_g[q] = gq;
            const int gx = lx8 + gq * 8;
            const int y0 = (tr0 + cr) * 5;
            const uint8_t* __restrict__ p0 = himg + gx;
            const bool any_col = gx < g.hcols;  // a group wholly past the frame edge is not read at all
#pragma unroll
            for (int k = 0; k < 5; k++)  // rows past the frame repeat the last row: duplicates do not change a min / max
                pix[q][k] = any_col ? *reinterpret_cast<const uint2*>(p0 + (size_t)min(y0 + k, g.hrows - 1) * g.hp) : make_uint2(0u, 0u);
            uint32_t mnE0 = 0x00ff00ffu, mnO0 = 0x00ff00ffu, mnE1 = 0x00ff00ffu, mnO1 = 0x00ff00ffu, mxE0 = 0, mxO0 = 0, mxE1 = 0, mxO1 = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const uint32_t e0 = pix[q][k].x & 0x00ff00ffu, o0 = (pix[q][k].x >> 8) & 0x00ff00ffu;
                const uint32_t e1 = pix[q][k].y & 0x00ff00ffu, o1 = (pix[q][k].y >> 8) & 0x00ff00ffu;
                mnE0 = pk_min_u16(mnE0, e0);
                mnO0 = pk_min_u16(mnO0, o0);
                mnE1 = pk_min_u16(mnE1, e1);
                mnO1 = pk_min_u16(mnO1, o1);
                mxE0 = pk_max_u16(mxE0, e0);
                mxO0 = pk_max_u16(mxO0, o0);
                mxE1 = pk_max_u16(mxE1, e1);
                mxO1 = pk_max_u16(mxO1, o1);
            }
            // columns at or beyond the frame width are neutral (255 for the min, 0 for the max)
            const int nvalid = min(max(g.hcols - gx, 0), 8);
            const uint64_t vm = nvalid >= 8 ? ~0ull : ((1ull << (8 * nvalid)) - 1ull);
            const uint32_t vm0 = (uint32_t)vm, vm1 = (uint32_t)(vm >> 32);
            const uint2 mn = make_uint2((mnE0 | (mnO0 << 8)) | ~vm0, (mnE1 | (mnO1 << 8)) | ~vm1);
            const uint2 mx = make_uint2((mxE0 | (mxO0 << 8)) & vm0, (mxE1 | (mxO1 << 8)) & vm1);
            *reinterpret_cast<uint2*>(vmin_s + cr * kVPitch + gq * 8) = mn;
            *reinterpret_cast<uint2*>(vmax_s + cr * kVPitch + gq * 8) = mx;
            // tile rows / columns are multiples of 5 and 8, so an item lies wholly inside the tile or wholly in the ring
            item_dark[q] = (lt4_bytes(mn.x, kp.tcap4) | lt4_bytes(mn.y, kp.tcap4)) != 0u;
            if (tr0 + cr >= trs0 && tr0 + cr <= trs1 && gx >= tx0 && gx < tx0 + tw_eff) dark = dark || item_dark[q];
        }
    }
    {
        const unsigned long long any = __ballot(dark);
        if (lane == 0) misc_s[12 + wave] = any != 0ull ? 1 : 0;
    }
    CCL_SYNC();
    stamp(0);
    // ---- bright tile: the reference caps the threshold at 0.3 (T <= 77 for every threshold tile, 0 on the frame border),
    // so a tile whose pixels are all >= 77 has no foreground whatever its thresholds turn out to be: its labels are zero
    // and it owns no component.  Exactly what the phases below would produce, without running them.
    if ((misc_s[12] | misc_s[13] | misc_s[14] | misc_s[15]) == 0) {
        empty_tile();
        return;
    }
    // ---- per-threshold-tile min / max (corner_detector.cpp:42-53): 5 column extrema each
    {
        const int nc = tc1 - tc0 + 1;
        const int m_nc = recip16(nc);
        for (int i = tid; i < nc * ncr; i += kCclThreads) {
            const int r = div_small(i, m_nc), c = i - r * nc;
            const uint8_t* a = vmin_s + r * kVPitch + (tc0 + c) * 5 - lx8;
            const uint8_t* b = vmax_s + r * kVPitch + (tc0 + c) * 5 - lx8;
            const int mn = min(min(min((int)a[0], (int)a[1]), min((int)a[2], (int)a[3])), (int)a[4]);
            const int mx = max(max(max((int)b[0], (int)b[1]), max((int)b[2], (int)b[3])), (int)b[4]);
            ext_s[r * L.ec + c] = (uint16_t)(mn | (mx << 8));
        }
    }
    CCL_SYNC();
    stamp(1);
    // ---- S3: 3x3 min-of-min / max-of-max for interior tiles, zero elsewhere (corner_detector.cpp:54-67, B1)
    {
        const int nc = tcs1 - tcs0 + 1, nr = trs1 - trs0 + 1;
        const int m_nc = recip16(nc);
        for (int i = tid; i < nc * nr; i += kCclThreads) {
            const int r = div_small(i, m_nc), c = i - r * nc;
            const int tr = trs0 + r, tc = tcs0 + c;
            int T = 0;
            // a threshold tile whose own minimum is >= 77 has no pixel below any threshold (T <= 77): its T is never needed
            if (tr >= 1 && tr <= g.trows - 2 && tc >= 1 && tc <= g.tcols - 2 && (ext_s[(tr - tr0) * L.ec + (tc - tc0)] & 0xff) < kp.tcap) {
                int mn = 255, mx = 0;
#pragma unroll
                for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                    for (int dx = -1; dx <= 1; dx++) {
                        const int e = ext_s[(tr + dy - tr0) * L.ec + (tc + dx - tc0)];
                        mn = min(mn, e & 0xff);
                        mx = max(mx, e >> 8);
                    }
                T = threshold_lookup(mn, mx, kp);  // <= tcap: the reference caps the threshold (at 0.3: 77)
            }
            const int xa = max(tc * tw - tx0, 0), xb = min((tc + 1) * tw - tx0, tw_eff);
            for (int x = xa; x < xb; x++) thr_s[r * (kTileW + 8) + x] = (uint8_t)T;
        }
    }
    CCL_SYNC();
    stamp(2);
    // ---- binary row masks (corner_detector.cpp:69-78) from the pixels still in registers
    {
        uint8_t* mask_b = reinterpret_cast<uint8_t*>(mask_s);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int tr = tr0 + it_cr[q];
            const int cx = lx8 + it_g[q] * 8 - tx0;
            if (it_cr[q] >= 0 && tr >= trs0 && tr <= trs1 && cx >= 0 && cx < tw_eff) {
                const uint2 tt = *reinterpret_cast<const uint2*>(thr_s + (tr - trs0) * (kTileW + 8) + cx);
                const int left = tw_eff - cx;
                const unsigned keep = left < 8 ? (1u << left) - 1u : 0xffu;
                const int r0 = tr * 5 - ty0;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    if (r0 + k < th_eff) {
                        // an item without a pixel below 77 is background whatever its thresholds are
                        const unsigned bits = item_dark[q] ? (lt4_bytes(pix[q][k].x, tt.x) | (lt4_bytes(pix[q][k].y, tt.y) << 4)) & keep : 0u;
                        mask_b[(r0 + k) * (kTileW / 8) + (cx >> 3)] = (uint8_t)bits;
                    }
                }
            }
        }
    }
    CCL_SYNC();
    stamp(3);
  } else {
        // ---- S1: stage the half-res region in LDS (16-byte chunks, coalesced along rows)
        {
            const uint8_t* __restrict__ himg = P.half + ((size_t)frame * g.hrows) * g.hp;
            for (int i = tid; i < T.chunks * T.nrows; i += kCclThreads) {
                const int r = i / T.chunks, c = i - r * T.chunks;
                const int x = lx0 + c * 16;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (x + 16 <= g.hp) v = *reinterpret_cast<const uint4*>(himg + (size_t)(py0 + r) * g.hp + x);
                *reinterpret_cast<uint4*>(hr_s + (size_t)r * L.rp + c * 16) = v;
            }
        }
        CCL_SYNC();
        stamp(0);
        // ---- S2: per-threshold-tile min / max (corner_detector.cpp:42-53)
        {
            const int nc = tc1 - tc0 + 1, nr = tr1 - tr0 + 1;
            for (int i = tid; i < nc * nr; i += kCclThreads) {
                const int r = i / nc, c = i - r * nc;
                const int ya = (tr0 + r) * tw, yb = min(ya + tw, g.hrows);
                const int xa = (tc0 + c) * tw, xb = min(xa + tw, g.hcols);
                int mn = 255, mx = 0;
                for (int y = ya; y < yb; y++) {
                    const uint8_t* row = hr_s + (size_t)(y - py0) * L.rp - lx0;
                    for (int x = xa; x < xb; x++) {
                        const int u = row[x];
                        mn = min(mn, u);
                        mx = max(mx, u);
                    }
                }
                ext_s[r * L.ec + c] = (uint16_t)(mn | (mx << 8));
            }
        }
        CCL_SYNC();
        stamp(1);
        // ---- S3: 3x3 min-of-min / max-of-max for interior tiles, zero elsewhere (corner_detector.cpp:54-67, B1)
        {
            const int nc = tcs1 - tcs0 + 1, nr = trs1 - trs0 + 1;
            for (int i = tid; i < nc * nr; i += kCclThreads) {
                const int r = i / nc, c = i - r * nc;
                const int tr = trs0 + r, tc = tcs0 + c;
                int T = 0;
                if (tr >= 1 && tr <= g.trows - 2 && tc >= 1 && tc <= g.tcols - 2) {
                    int mn = 255, mx = 0;
    #pragma unroll
                    for (int dy = -1; dy <= 1; dy++)
    #pragma unroll
                        for (int dx = -1; dx <= 1; dx++) {
                            const int e = ext_s[(tr + dy - tr0) * L.ec + (tc + dx - tc0)];
                            mn = min(mn, e & 0xff);
                            mx = max(mx, e >> 8);
                        }
                    T = threshold_bound(mn, mx, kp.dark_cap);  // <= tcap: the reference caps the threshold (at 0.3: 77)
                }
                const int xa = max(tc * tw - tx0, 0), xb = min((tc + 1) * tw - tx0, tw_eff);
                for (int x = xa; x < xb; x++) thr_s[r * (kTileW + 8) + x] = (uint8_t)T;
            }
        }
        CCL_SYNC();
        stamp(2);
        // ---- S4: binary row masks (corner_detector.cpp:69-78), 8 pixels per thread.  pixel < T is evaluated on packed bytes:
        // T <= 77 < 128, so with the pixel's top bit handled separately the per-byte subtract (0x80 | low7) - T never borrows
        // across bytes and its bit 7 says low7 >= T.  The four result bits of a dword are gathered with one multiply.
        {
            constexpr int groups = kTileW / 8;
            uint8_t* mask_b = reinterpret_cast<uint8_t*>(mask_s);
            const int xoff = tx0 - lx0;
            const bool fast = (((uintptr_t)0 + xoff) & 7) == 0;  // 8-byte aligned pixel groups (always for tw = 5)
            for (int i = tid; i < kTileH * groups; i += kCclThreads) {
                const int r = i / groups, gq = i - r * groups;
                unsigned bits = 0;
                if (r < th_eff && gq * 8 < tw_eff) {
                    const int y = ty0 + r;
                    const uint8_t* prow = hr_s + (size_t)(y - py0) * L.rp + xoff + gq * 8;
                    const uint8_t* trow = thr_s + (y / tw - trs0) * (kTileW + 8) + gq * 8;
                    uint32_t u0, u1;
                    if (fast) {
                        const uint2 uu = *reinterpret_cast<const uint2*>(prow);
                        u0 = uu.x;
                        u1 = uu.y;
                    } else {
                        u0 = (uint32_t)prow[0] | ((uint32_t)prow[1] << 8) | ((uint32_t)prow[2] << 16) | ((uint32_t)prow[3] << 24);
                        u1 = (uint32_t)prow[4] | ((uint32_t)prow[5] << 8) | ((uint32_t)prow[6] << 16) | ((uint32_t)prow[7] << 24);
                    }
                    const uint2 tt = *reinterpret_cast<const uint2*>(trow);
                    auto lt4 = [](uint32_t u, uint32_t t) {
                        const uint32_t ge = (((u & 0x7f7f7f7fu) | 0x80808080u) - t) & 0x80808080u;
                        const uint32_t lt = ~(ge | u) & 0x80808080u;
                        return (((lt >> 7) * 0x01020408u) >> 24) & 0xfu;
                    };
                    bits = lt4(u0, tt.x) | (lt4(u1, tt.y) << 4);
                    const int left = tw_eff - gq * 8;  // columns of this group inside the frame
                    if (left < 8) bits &= (1u << left) - 1u;
                }
                mask_b[i] = (uint8_t)bits;
            }
        }
        CCL_SYNC();
        stamp(3);
  }
    // ---- S5: run starts, run numbering
    if (tid == 0) misc_s[10] = 0;  // the label blocks S11 finds foreground in (tile_dirty)
    int nruns_mine = 0;
    if (tid < kTileH * kTileWords) {
        const int w = tid % kTileWords;
        const uint64_t m = mask_s[tid];
        const uint64_t carry = w > 0 ? (mask_s[tid - 1] >> 63) : 0ull;
        const uint64_t st = m & ~((m << 1) | carry);
        start_s[tid] = st;
        nruns_mine = __popcll(st);
    }
    int nruns;
    const int rb = block_excl_scan(nruns_mine, misc_s, nruns);
    if (tid < kTileH * kTileWords) runbase_s[tid] = rb;
    // a tile this pass cannot hold goes to the overflow list (BIG: cannot happen, the caps cover every tile)
    auto hand_over = [&]() {
        if (tid == 0) {
            if (BIG) {
                atomicOr(&P.frame_flags[T.frame], CTAG_FLAG_POOL_OVERFLOW);
                P.tile_base[(size_t)T.frame * g.tiles_x * g.tiles_y + T.tile] = 0;
            } else {
                P.ovf_list[atomicAdd(P.ovf_count, 1)] = T.frame * (g.tiles_x * g.tiles_y) + T.tile;
            }
        }
    };
    bool overflow = nruns > RUNCAP;
    for (int i = tid; i < min(nruns, RUNCAP); i += kCclThreads) parent_s[i] = (unsigned)i;
    CCL_SYNC();
    auto runid = [&](int item, int b) -> int { return runbase_s[item] + popc_le(start_s[item], b) - 1; };

    stamp(4);
    // ---- S6: unions between vertically adjacent rows (8-connectivity)
    if (!overflow && tid < kTileH * kTileWords && tid >= kTileWords) {
        const int w = tid % kTileWords;
        const uint64_t cur = mask_s[tid];
        if (cur) {
            const int up_item = tid - kTileWords;
            const uint64_t up = mask_s[up_item];
            const uint64_t curL = w > 0 ? (mask_s[tid - 1] >> 63) : 0ull, curR = w < kTileWords - 1 ? (mask_s[tid + 1] & 1ull) : 0ull;
            const uint64_t upL = w > 0 ? (mask_s[up_item - 1] >> 63) : 0ull, upR = w < kTileWords - 1 ? (mask_s[up_item + 1] & 1ull) : 0ull;
            const uint64_t both = cur & up;
            const uint64_t A = both & ~((both << 1) | (curL & upL));
            const uint64_t B = cur & ~up & ((up << 1) | upL) & ~((cur << 1) | curL);
            const uint64_t C = cur & ~up & ((up >> 1) | (upR << 63)) & ~((cur >> 1) | (curR << 63));
            uint64_t todo = A | B | C;
            // (this word's and the upper word's starts and run bases once, in registers: the run of a bit is a population count away -- round 6)
            const uint64_t st_c = start_s[tid], st_u = start_s[up_item];
            const int rb_c = runbase_s[tid] - 1, rb_u = runbase_s[up_item] - 1;
            while (todo) {
                const int b = __ffsll((unsigned long long)todo) - 1;
                todo &= todo - 1;
                const unsigned c = (unsigned)(rb_c + popc_le(st_c, b));
                if ((A >> b) & 1) lds_union(parent_s, c, (unsigned)(rb_u + popc_le(st_u, b)));
                if ((B >> b) & 1) lds_union(parent_s, c, (unsigned)(b > 0 ? rb_u + popc_le(st_u, b - 1) : runid(up_item - 1, 63)));
                if ((C >> b) & 1) lds_union(parent_s, c, (unsigned)(b < 63 ? rb_u + popc_le(st_u, b + 1) : runid(up_item + 1, 0)));
            }
        }
    }
    CCL_SYNC();
    stamp(5);
    // ---- S7/S8: flatten, compact roots into slots
    int nslots = 0;
    {
        // thread t owns runs t, t + 256, t + 512, ...: consecutive lanes read consecutive words (as eight consecutive runs per
        // thread the lanes' addresses were 8 words apart -- a 16-way LDS bank conflict, the largest share of K2's 27 % conflict rate).
        // Slots are numbered in that (thread, k) order; any numbering is as good as another, later stages use the slot as a name only.
        // (loops bounded by the tile's run count, not by the capacity: a tile has ~100 runs, one trip instead of eight tested ones)
        int roots_mine = 0;
        if (!overflow)
            for (int i = tid; i < nruns; i += kCclThreads) roots_mine += parent_s[i] == (unsigned)i ? 1 : 0;
        int s = block_excl_scan(roots_mine, misc_s, nslots);
        if (!overflow)
            for (int i = tid; i < nruns; i += kCclThreads)
                if (parent_s[i] == (unsigned)i) lab_s[i] = (uint16_t)(s++);
    }
    if (nslots > SLOTCAP) overflow = true;
    CCL_SYNC();
    if (overflow) {  // block-uniform: the tile does not fit this pass
        hand_over();
        return;
    }
    int* st_area = reinterpret_cast<int*>(smem + L.off_stats);  // overlays buffers that are dead from here on (see ccl_layout)
    int* st_xmin = st_area + SLOTCAP;
    int* st_xmax = st_xmin + SLOTCAP;
    unsigned* st_rows = reinterpret_cast<unsigned*>(st_xmax + SLOTCAP);  // bit r: the component has a pixel in tile row r (kTileH <= 32)
    int* st_key = reinterpret_cast<int*>(st_rows + SLOTCAP);
    for (int i = tid; i < nruns; i += kCclThreads) {
        const unsigned r = lds_find(parent_s, (unsigned)i);
        if (r != (unsigned)i) lab_s[i] = lab_s[r];
    }
    CCL_SYNC();  // the statistics below overlay the run parents
    for (int i = tid; i < nslots; i += kCclThreads) {
        st_area[i] = 0;
        st_xmin[i] = 0x7fffffff;
        st_xmax[i] = -1;
        st_rows[i] = 0u;
        st_key[i] = 0x7fffffff;
    }
    CCL_SYNC();
    // pool entries for this tile.  The first pass cannot run out: it publishes at most SLOTCAP entries per tile and the pool
    // holds twice that for every tile (FrameGeom::pool_cap); the second pass takes what is left and fails the frame when
    // that is not enough (more than ~256 components per tile that are large or touch a tile border, frame-wide).  The
    // returning atomic is issued now and its result first used after the label stores.
    auto reserve = [&](int n) -> int {  // one lane
        if (n <= 0) return 0;
        const int at = atomicAdd(&P.frame_ncomp[frame], n);
        return at + n <= g.pool_cap ? at : -1;
    };
    int base_reg = 0;
    if (!BIG && tid == 0) base_reg = reserve(nslots);
    stamp(6);
    // ---- S9: per-slot stats from run segments (area, bbox, first 2x2 block in block-raster order)
    const int bcols = (g.hcols + 1) >> 1;
    if (tid < kTileH * kTileWords) {
        const int r = tid / kTileWords, w = tid - r * kTileWords;
        uint64_t cur = mask_s[tid];
        // the word's run segments are consecutive runs: the first one is the word's first start -- or, when it goes on from the word before (bit 0 set and not a
        // start), the run before that -- and every further segment is the next run: no population count per segment (round 6)
        int rid = runbase_s[tid] - (((cur & 1ull) != 0ull && (start_s[tid] & 1ull) == 0ull) ? 1 : 0);
        while (cur) {
            const int s = __ffsll((unsigned long long)cur) - 1;
            const uint64_t inv = ~(cur >> s);
            const int len = inv ? (__ffsll((unsigned long long)inv) - 1) : (64 - s);
            const int e = s + len - 1;
            cur = (e >= 63) ? 0ull : (cur & ~mask_le(e));
            const int slot = lab_s[rid++];
            const int gx0 = tx0 + w * 64 + s, gx1 = tx0 + w * 64 + e, gy = ty0 + r;
            atomicAdd(&st_area[slot], len);
            atomicMin(&st_xmin[slot], gx0);
            atomicMax(&st_xmax[slot], gx1);
            atomicOr(&st_rows[slot], 1u << r);
            atomicMin(&st_key[slot], (gy >> 1) * bcols + (gx0 >> 1));
        }
    }
    stamp(7);
    int npub = nslots;
    uint16_t* final_s = reinterpret_cast<uint16_t*>(smem + L.off_final);  // BIG: slot -> label (published: 1.., culled: 0x8000 | k)
    if (BIG) {
        // ---- cull: a component of fewer than 30 pixels that does not touch the tile border can neither pass the area filter
        // (corner_detector.cpp:88) nor merge with anything: it is not published (its pixels keep a private label)
        CCL_SYNC();  // S9's LDS atomics are done
        constexpr int per = (SLOTCAP + kCclThreads - 1) / kCclThreads;
        const int i0 = tid * per;
        auto keep = [&](int i) -> bool {
            const unsigned rows = st_rows[i];
            const bool border = st_xmin[i] == tx0 || st_xmax[i] == tx0 + tw_eff - 1 || (rows & 1u) != 0u || ((rows >> (th_eff - 1)) & 1u) != 0u;
            return border || st_area[i] >= kp.area_min;
        };
        int mine = 0;
        for (int k = 0; k < per; k++)
            if (i0 + k < nslots && keep(i0 + k)) mine++;
        int pub = block_excl_scan(mine, misc_s, npub);
        for (int k = 0; k < per; k++) {
            const int i = i0 + k;
            if (i < nslots) {
                const bool kp = keep(i);
                final_s[i] = kp ? (uint16_t)(pub + 1) : (uint16_t)(0x8000u | (unsigned)(i - pub));  // i - pub = culled slots before i: < 2^15
                pub += kp ? 1 : 0;
            }
        }
        CCL_SYNC();
        for (int i = tid; i < nruns; i += kCclThreads) lab_s[i] = final_s[lab_s[i]];
        if (tid == 0) base_reg = reserve(npub);
        CCL_SYNC();
    }
    // The reservation's result goes to LDS NOW, before the label stores are issued: the vector-memory counter is served in order, so the wait for a value
    // requested before the stores, taken after them, would be a wait for every label store of wave 0 as well.  (Round 5 moved it here on the suspicion that this
    // was the 21 % of a tile's lifetime the phase clocks put in "publish"; the kernel's time did not change -- that share is wave 0 waiting at the barrier for the
    // waves whose label blocks hold the foreground.  Also measured and not kept: path halving in lds_find -- a strip's column is a chain of 30 runs --, 0.985 ->
    // 0.975 ms; the two block scans on separate scratch words without their trailing barriers and the statistics initialised early at the end of the parents'
    // region, three barriers of twelve fewer: 1.00 -> 1.00 ms.  Neither chain depth nor barrier count is what a tile's 9 us are made of.)
    if (tid == 0) misc_s[9] = base_reg;
    // ---- S11: per-pixel tile-local labels, 8 pixels (16 bytes) per lane
    {
        uint16_t* __restrict__ limg = P.labels + ((size_t)frame * g.hrows) * g.lp;
        // A wave takes an 8-row x 8-group block (its stores are eight full 128-byte lines), not 64 consecutive groups of one
        // or two rows: foreground is clustered, so most waves of a tile see background only and skip the per-segment path,
        // whereas a row-major wave crosses a foreground stripe almost every time (8 % foreground groups, 99 % of the waves).
        constexpr int groups = kTileW / 8, gblocks = groups / 8, rblocks = (kTileH + 7) / 8;
        static_assert(groups % 8 == 0, "label blocks are 8 groups wide");
        // tile_dirty is a bit per such block (20 of them): a block without foreground whose label pixels are zero already -- never written, or
        // wiped since -- is skipped, stores and address arithmetic alike: within a tile that holds foreground most blocks still hold none
        static_assert(rblocks * gblocks <= 31, "a bit of tile_dirty per label block");
        int new_dirty = 0;
        for (int i = tid; i < rblocks * gblocks * 64; i += kCclThreads) {
            const int blk = i >> 6, l = i & 63;
            const int by = blk / gblocks, bx = blk - by * gblocks;
            const int r = by * 8 + (l >> 3), gq = bx * 8 + (l & 7);
            const bool inr = r < th_eff && gq * 8 < tw_eff;
            const int item = min(r, kTileH - 1) * kTileWords + (gq >> 3);
            const int b0 = (gq & 7) * 8;
            const unsigned byte = inr ? (unsigned)((mask_s[item] >> b0) & 0xff) : 0u;
            const bool any = __ballot(byte != 0u) != 0ull;  // wave-uniform: the trip counts are (1280 = 20 x 64 items)
            if (!any && !((was_dirty >> blk) & 1)) continue;
            new_dirty |= any ? (1 << blk) : 0;
            if (!inr) continue;
            uint32_t o[4] = {0, 0, 0, 0};
            if (byte) {
                // one run-id lookup per run segment of the group (a run's pixels share its label), then a select per pixel
                unsigned rest = byte;
                uint32_t lab8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int rid = runid(item, b0 + __ffs(rest) - 1);  // of the group's first segment; the further ones are the next runs
                while (rest) {
                    const int sb = __ffs(rest) - 1;
                    const unsigned inv = ~(rest >> sb);
                    const int len = __ffs(inv) - 1;  // rest >> sb has at most 8 significant bits, so inv != 0
                    const uint32_t lab = (uint32_t)lab_s[rid++] + (BIG ? 0u : 1u);
                    const unsigned seg = ((1u << len) - 1u) << sb;
#pragma unroll
                    for (int k = 0; k < 8; k++) lab8[k] = ((seg >> k) & 1u) ? lab : lab8[k];
                    rest &= ~seg;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) o[k] = lab8[2 * k] | (lab8[2 * k + 1] << 16);
            }
            *reinterpret_cast<uint4*>(limg + (size_t)(ty0 + r) * g.lp + tx0 + gq * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        if (lane == 0 && new_dirty) atomicOr(&misc_s[10], new_dirty);
    }
    stamp(8);
    // ---- S10: publish the tile's components in the frame pool
    CCL_SYNC();  // misc_s[9] (the pool entries' base, stored before S11); also orders the S9 LDS atomics before the reads below
    const int base = misc_s[9];
    if (base < 0) {  // second pass only: the frame's pool is exhausted
        // S11 has stored this tile's labels: the dirty bits must say so although the tile publishes nothing (the frame is rerun through the
        // any-frame workspace) -- or the next frame in this workspace slot skips label blocks that still hold THIS frame's labels (round-4 ADVICE)
        if (tid == 0 && misc_s[10] != was_dirty) *dirty_p = misc_s[10];
        hand_over();
        return;
    }
    if (tid == 0) {
        P.tile_base[(size_t)frame * g.tiles_x * g.tiles_y + tile] = base;
        if (misc_s[10] != was_dirty) *dirty_p = misc_s[10];  // the label blocks that hold foreground now (S11)
    }
    {
        const size_t pool0 = (size_t)frame * g.pool_cap;
        for (int i = tid; i < nslots; i += kCclThreads) {
            int at = i;
            if (BIG) {
                const unsigned f = final_s[i];
                if (f & 0x8000u) continue;
                at = (int)f - 1;
            }
            const size_t gidx = pool0 + base + at;
            // the pool arrays are carved back to back at a fixed stride: one base pointer instead of nine
            char* q = reinterpret_cast<char*>(P.parent + gidx);
            const size_t ps = P.pool_stride;
            *reinterpret_cast<unsigned*>(q) = (unsigned)(base + at);       // parent
            *reinterpret_cast<int*>(q + 2 * ps) = st_area[i];               // area
            *reinterpret_cast<int*>(q + 3 * ps) = st_xmin[i];
            const unsigned rows = st_rows[i];  // never 0: a slot owns at least one run
            *reinterpret_cast<int*>(q + 4 * ps) = ty0 + __ffs(rows) - 1;    // ymin
            *reinterpret_cast<int*>(q + 5 * ps) = st_xmax[i];
            *reinterpret_cast<int*>(q + 6 * ps) = ty0 + 31 - __clz(rows);   // ymax
            *reinterpret_cast<int*>(q + 7 * ps) = st_key[i];
            *reinterpret_cast<int*>(q + 8 * ps) = tile;                     // pool_tile
            *reinterpret_cast<int*>(q + 9 * ps) = -1;                       // member_head
        }
    }
    (void)npub;
    stamp(9);
    if (P.stamps && tid == 0) {
#pragma unroll
        for (int q = 0; q < 10; q++) {
            atomicAdd(&P.stamps[q], t_acc[q]);
            t_acc[q] = 0;
        }
        t_prev = __builtin_amdgcn_s_memtime();
    }
  }
}

template <int TWC, bool MASKIN = false>
#ifdef CTAG_CCL_WAVES
__global__ __launch_bounds__(kCclThreads) __attribute__((amdgpu_waves_per_eu(CTAG_CCL_WAVES, CTAG_CCL_WAVES)))
#else
__global__ __launch_bounds__(kCclThreads)
#endif
void k_threshold_ccl(SweepPtrs P, FrameGeom g, KParams kp, int nframes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // one tile per block; blocks b and b+8 share an XCD, so a frame's tiles stay on one XCD (map_block)
    int frame0, tile0;
    if (!map_block(blockIdx.x, g.tiles_x * g.tiles_y, nframes, frame0, tile0)) return;
    ccl_tile<TWC, kRunCap, kSlotCap, false, MASKIN>(smem, P, g, kp, frame0, tile0);
}

// second pass over the tiles of the overflow list (usually none: the blocks read the count and leave)
template <int TWC, bool MASKIN = false>
__global__ __launch_bounds__(kCclThreads) void k_threshold_ccl_big(SweepPtrs P, FrameGeom g, KParams kp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int count = *P.ovf_count, per_frame = g.tiles_x * g.tiles_y;
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        const int e = P.ovf_list[i];
        ccl_tile<TWC, kRunCapBig, kSlotCapBig, true, MASKIN>(smem, P, g, kp, e / per_frame, e % per_frame);
        __syncthreads();
    }
}
static_assert(kSlotCapBig < 0x7fff, "labels are 1..slots or 0x8000 | culled slot: 0xffff never occurs (k_quad's packed label match relies on it)");
static_assert(2 * kSlotCap <= 256, "FrameGeom::pool_cap = 256 entries per tile holds two first passes");
static_assert(kRunCapBig % kCclThreads == 0 && kRunCapBig >= (kTileW / 2) * kTileH && kSlotCapBig >= (kTileW / 2) * ((kTileH + 1) / 2), "second-pass caps hold any tile");

hipError_t launch_threshold_ccl(int nframes, const Workspace& ws, hipStream_t s, bool fused) {
    const FrameGeom& g = ws.g;
    const size_t lds = threshold_ccl_lds_bytes(g.tw);
    const int grid = grid_for(nframes, g.tiles_x * g.tiles_y);  // one 320x30 tile per block
    SweepPtrs P = sweep_ptrs(ws);
    static unsigned long long* d_stamps = nullptr;
    const bool want_stamps = getenv("CTAG_CCL_STAMPS") != nullptr;
    if (want_stamps) {
        if (!d_stamps) (void)hipMalloc(reinterpret_cast<void**>(&d_stamps), 16 * 8);
        (void)hipMemsetAsync(d_stamps, 0, 16 * 8, s);
        P.stamps = d_stamps;
    }
    const size_t lds_big = ccl_layout(g.tw, kRunCapBig, kSlotCapBig, true).total;
    const int grid_big = 1024;  // persistent: loops over the overflow list
    // kernels that ask for more than 64 KB of dynamic LDS need the attribute raised (per device; it only ever grows)
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 || dev >= 64 ? 0 : dev;
    auto want_lds = [dev](const void* fn, size_t bytes, size_t* have) {
        if (bytes > 64 * 1024 && bytes > have[dev]) {
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            have[dev] = bytes;
        }
    };
    static size_t have_big5[64] = {0}, have_big0[64] = {0}, have_0[64] = {0}, have_big5m[64] = {0};
    if (fused) {
        // (measured and not kept: K2 over work lists of the tiles with foreground, flagged by k_decimate_mask -- as looping blocks 1.33-1.55 ms, as a
        // block per list entry 1.36 ms per 4096 frames against 1.25 ms for a block per tile: the blocks of background tiles load, look and leave in the
        // shadow of their neighbours' label phases, while the lists cost two more dependent loads per tile and 0.1 ms of atomics in K1)
        hipLaunchKernelGGL((k_threshold_ccl<5, true>), dim3(grid), dim3(kCclThreads), lds, s, P, g, ws.kp, nframes);
        want_lds(reinterpret_cast<const void*>(k_threshold_ccl_big<5, true>), lds_big, have_big5m);
        hipLaunchKernelGGL((k_threshold_ccl_big<5, true>), dim3(grid_big), dim3(kCclThreads), lds_big, s, P, g, ws.kp);
    } else if (g.tw == 5) {
        hipLaunchKernelGGL(k_threshold_ccl<5>, dim3(grid), dim3(kCclThreads), lds, s, P, g, ws.kp, nframes);
        want_lds(reinterpret_cast<const void*>(k_threshold_ccl_big<5>), lds_big, have_big5);
        hipLaunchKernelGGL(k_threshold_ccl_big<5>, dim3(grid_big), dim3(kCclThreads), lds_big, s, P, g, ws.kp);
    } else {
        want_lds(reinterpret_cast<const void*>(k_threshold_ccl<0>), lds, have_0);
        hipLaunchKernelGGL(k_threshold_ccl<0>, dim3(grid), dim3(kCclThreads), lds, s, P, g, ws.kp, nframes);
        want_lds(reinterpret_cast<const void*>(k_threshold_ccl_big<0>), lds_big, have_big0);
        hipLaunchKernelGGL(k_threshold_ccl_big<0>, dim3(grid_big), dim3(kCclThreads), lds_big, s, P, g, ws.kp);
    }
    if (want_stamps) {
        unsigned long long h[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost);
        unsigned long long tot = 0;
        for (int i = 0; i < 10; i++) tot += h[i];
        static const char* nm[10] = {"S1 stage", "S2 minmax", "S3 dilate", "S4 masks", "S5 runs", "S6 unions", "S7-8 flatten", "S9 stats", "S11 labels", "S10 publish"};
        fprintf(stderr, "[k_threshold_ccl cycles]");
        for (int i = 0; i < 10; i++) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * h[i] / tot);
        fprintf(stderr, " (total %llu)\n", tot);
    }
    return hipGetLastError();
}

// =====================================================================================================
// K3: merge tile-local components across tile seams with a global (per-frame pool) union-find.
// =====================================================================================================
__device__ __forceinline__ unsigned g_find(const uint32_t* parent, unsigned x) {
    unsigned p;
    while ((p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != x) x = p;
    return x;
}
__device__ __forceinline__ void g_union(uint32_t* parent, unsigned a, unsigned b) {
    while (true) {
        a = g_find(parent, a);
        b = g_find(parent, b);
        if (a == b) return;
        if (a > b) {
            const unsigned t = a;
            a = b;
            b = t;
        }
        const unsigned old = atomicMin(&parent[b], a);
        if (old == b) return;
        b = old;
    }
}

// One seam pixel (the pixel on the lower side of a horizontal seam / right side of a vertical seam).  A union is
// issued only where a contact between two runs BEGINS along the seam: a neighbouring seam pixel that already sees the
// same pair of runs does the union, and adjacency across the other seam family makes the rest transitive.
__device__ __forceinline__ void seam_pixel(const uint16_t* __restrict__ limg, const int32_t* __restrict__ tbase, uint32_t* parent, const FrameGeom& g,
                                           int x, int y, bool horizontal, unsigned c0) {
    auto lab = [&](int xx, int yy) -> unsigned {
        if (xx < 0 || xx >= g.hcols || yy < 0 || yy >= g.hrows) return 0u;
        return limg[(size_t)yy * g.lp + xx];
    };
    auto gid = [&](int xx, int yy, unsigned l) { return (unsigned)tbase[(yy / kTileH) * g.tiles_x + (xx / kTileW)] + l - 1; };
    // along-seam step (dxs, dys); (ox, oy) leads to the neighbour across the seam
    const int dxs = horizontal ? 1 : 0, dys = horizontal ? 0 : 1, ox = horizontal ? 0 : -1, oy = horizontal ? -1 : 0;
    const unsigned o0 = lab(x + ox, y + oy);
    const unsigned cm = lab(x - dxs, y - dys), om = lab(x - dxs + ox, y - dys + oy);  // previous position along the seam
    const unsigned cp = lab(x + dxs, y + dys), op = lab(x + dxs + ox, y + dys + oy);  // next position
    const unsigned me = gid(x, y, c0);
    // the shortcuts rely on "adjacent pixels of one row/column segment inside a tile share a label"; across a tile
    // boundary (the 4-tile corners) they could defer to each other in a circle, so they are not applied there
    const int along = horizontal ? x : y, tile_len = horizontal ? kTileW : kTileH;
    const bool same_prev = (along % tile_len) != 0, same_next = ((along + 1) % tile_len) != 0;
    if (o0) {
        if (!(same_prev && cm && om)) g_union(parent, me, gid(x + ox, y + oy, o0));  // contact starts here
    } else {
        if (om && !(same_prev && cm)) g_union(parent, me, gid(x - dxs + ox, y - dys + oy, om));  // diagonal back
        if (op && !(same_next && cp)) g_union(parent, me, gid(x + dxs + ox, y + dys + oy, op));  // diagonal forward
    }
}

// Horizontal seams: an item is 8 consecutive seam pixels behind one 16-byte label load (nine out of ten are all
// background and need nothing else); vertical seams: an item is one pixel.  A thread takes kSeamItems items, 256 apart,
// and requests all their labels before it looks at the first.  The foreground seam pixels found are compacted into an
// LDS list and then handled one per lane: seam_pixel is a chain of dependent loads and atomics, and eight of them in a
// row on the one lane whose group is foreground kept the other 63 lanes of its wave waiting.
#ifndef CTAG_SEAM_ITEMS
#define CTAG_SEAM_ITEMS 2
#endif
constexpr int kSeamItems = CTAG_SEAM_ITEMS;
constexpr int kSeamList = 1024;  // foreground seam pixels per block held in LDS; the surplus is handled in place
__global__ __launch_bounds__(256) void k_seam_merge(SweepPtrs P, FrameGeom g, int nframes, int per_frame_blocks) {
    int frame, bidx;
    if (!map_block(blockIdx.x, per_frame_blocks, nframes, frame, bidx)) return;
    __shared__ uint32_t s_list[kSeamList];  // x (12 bits) | y (12 bits) << 12 | horizontal << 24
    __shared__ uint16_t s_lab[kSeamList];
    __shared__ int s_count;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const int hc8 = (g.hcols + 7) >> 3;
    const int nh = (g.tiles_y - 1) * hc8;      // 8-pixel groups on the lower side of horizontal seams
    const int nv = (g.tiles_x - 1) * g.hrows;  // pixels on the right side of vertical seams
    const uint16_t* __restrict__ limg = P.labels + ((size_t)frame * g.hrows) * g.lp;
    const int32_t* __restrict__ tbase = P.tile_base + (size_t)frame * g.tiles_x * g.tiles_y;
    uint32_t* parent = P.parent + (size_t)frame * g.pool_cap;
    uint4 v[kSeamItems];
    int px[kSeamItems], py[kSeamItems];
#pragma unroll
    for (int it = 0; it < kSeamItems; it++) {
        const int i = (bidx * kSeamItems + it) * 256 + threadIdx.x;
        v[it] = make_uint4(0u, 0u, 0u, 0u);
        px[it] = -1;
        py[it] = 0;
        if (i < nh) {
            const int sm = i / hc8;
            px[it] = (i - sm * hc8) * 8;
            py[it] = (sm + 1) * kTileH;
            v[it] = *reinterpret_cast<const uint4*>(limg + (size_t)py[it] * g.lp + px[it]);  // lp is a multiple of 64: in bounds, aligned
        } else if (i < nh + nv) {
            const int j = i - nh;
            const int sm = j / g.hrows;
            py[it] = j - sm * g.hrows;
            px[it] = -2 - (sm + 1) * kTileW;  // vertical item: x = -(px + 2)
            v[it].x = limg[(size_t)py[it] * g.lp + (sm + 1) * kTileW];
        }
    }
    const bool packable = g.hcols <= 4096 && g.hrows <= 4096;
#pragma unroll
    for (int it = 0; it < kSeamItems; it++) {
        if ((v[it].x | v[it].y | v[it].z | v[it].w) == 0u) continue;
        if (px[it] >= 0) {
            const uint32_t w[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) cnt += (((w[k >> 1] >> (16 * (k & 1))) & 0xffffu) != 0u && px[it] + k < g.hcols) ? 1 : 0;
            int slot = packable ? atomicAdd(&s_count, cnt) : kSeamList;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const unsigned c0 = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
                if (c0 && px[it] + k < g.hcols) {
                    if (slot < kSeamList) {
                        s_list[slot] = (uint32_t)(px[it] + k) | ((uint32_t)py[it] << 12) | (1u << 24);
                        s_lab[slot] = (uint16_t)c0;
                    } else {
                        seam_pixel(limg, tbase, parent, g, px[it] + k, py[it], true, c0);
                    }
                    slot++;
                }
            }
        } else {
            const int slot = packable ? atomicAdd(&s_count, 1) : kSeamList;
            if (slot < kSeamList) {
                s_list[slot] = (uint32_t)(-(px[it] + 2)) | ((uint32_t)py[it] << 12);
                s_lab[slot] = (uint16_t)v[it].x;
            } else {
                seam_pixel(limg, tbase, parent, g, -(px[it] + 2), py[it], false, v[it].x);
            }
        }
    }
    __syncthreads();
    const int n = min(s_count, kSeamList);
    for (int k = threadIdx.x; k < n; k += 256) {
        const uint32_t e = s_list[k];
        seam_pixel(limg, tbase, parent, g, (int)(e & 0xfffu), (int)((e >> 12) & 0xfffu), (e >> 24) != 0u, s_lab[k]);
    }
}

hipError_t launch_seam_merge(int nframes, const Workspace& ws, hipStream_t s) {
    const FrameGeom& g = ws.g;
    const int n = (g.tiles_y - 1) * ((g.hcols + 7) >> 3) + (g.tiles_x - 1) * g.hrows;
    if (n <= 0) return hipSuccess;
    const int per_frame = (n + 256 * kSeamItems - 1) / (256 * kSeamItems);
    hipLaunchKernelGGL(k_seam_merge, dim3(grid_for(nframes, per_frame)), dim3(256), 0, s, sweep_ptrs(ws), g, nframes, per_frame);
    return hipGetLastError();
}

// =====================================================================================================
// K4: flatten every pool entry to its root and fold its stats into the root.
// =====================================================================================================
__global__ __launch_bounds__(256) void k_resolve(SweepPtrs P, int nframes, int per_frame_blocks, int pool_cap) {
    int frame, bidx;
    if (!map_block(blockIdx.x, per_frame_blocks, nframes, frame, bidx)) return;
    const int n = min(P.frame_ncomp[frame], pool_cap);
    const size_t pool0 = (size_t)frame * pool_cap;
    for (int i = bidx * 256 + threadIdx.x; i < n; i += per_frame_blocks * 256) {
        const unsigned r = g_find(P.parent + pool0, (unsigned)i);
        P.root_of[pool0 + i] = (int)r;
        if (r != (unsigned)i) {
            P.member_next[pool0 + i] = atomicExch(&P.member_head[pool0 + r], i);  // set membership only; order is irrelevant
            atomicAdd(&P.area[pool0 + r], P.area[pool0 + i]);
            atomicMin(&P.xmin[pool0 + r], P.xmin[pool0 + i]);
            atomicMax(&P.xmax[pool0 + r], P.xmax[pool0 + i]);
            atomicMin(&P.ymin[pool0 + r], P.ymin[pool0 + i]);
            atomicMax(&P.ymax[pool0 + r], P.ymax[pool0 + i]);
            atomicMin(&P.key[pool0 + r], P.key[pool0 + i]);
        }
    }
}
// the per-chunk counters, zeroed by ONE kernel at the head of the chain (five hipMemsetAsync calls before; as memset nodes of a
// captured hipGraph they did not take effect on replay under ROCm 7.0, CTAG_OPT_GRAPH)
__global__ __launch_bounds__(256) void k_zero_counters(int32_t* a, uint32_t* b, int32_t* c, int32_t* d, int32_t* one, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        a[i] = 0;
        b[i] = 0u;
        c[i] = 0;
        d[i] = 0;
    }
    if (i == 0) *one = 0;
}
hipError_t launch_zero_counters(int nframes, const Workspace& ws, hipStream_t s) {
    hipLaunchKernelGGL(k_zero_counters, dim3((nframes + 255) / 256), dim3(256), 0, s, ws.frame_ncomp, ws.frame_flags, ws.line_count, ws.clp_used, ws.ovf_count, nframes);
    return hipGetLastError();
}

hipError_t launch_resolve(int nframes, const Workspace& ws, hipStream_t s) {
    const int per_frame = 2;
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(nframes, per_frame)), dim3(256), 0, s, sweep_ptrs(ws), nframes, per_frame, ws.g.pool_cap);
    return hipGetLastError();
}

// =====================================================================================================
// K5: area filter (corner_detector.cpp:87-91) and OpenCV label order: candidates sorted
// by the block-raster index of their first 2x2 block.  One block per frame; rank sort in LDS.
// =====================================================================================================
static_assert(sizeof(CandAux) >= sizeof(int2), "k_candidates borrows cand_aux as (pool index, key) scratch");
__global__ __launch_bounds__(256) void k_candidates(SweepPtrs P, FrameGeom g, int nframes, int area_min) {
    __shared__ int s_idx[kLdsCand];
    __shared__ int s_key[kLdsCand];
    __shared__ int s_count, s_roots;
    const int frame = blockIdx.x;
    if (frame >= nframes) return;
    if (threadIdx.x == 0) s_count = s_roots = 0;
    __syncthreads();
    const int n = min(P.frame_ncomp[frame], g.pool_cap);
    const size_t pool0 = (size_t)frame * g.pool_cap;
    int my_roots = 0;
    int2* scratch = P.cand_scratch + (size_t)frame * P.cand_cap;
    for (int i = threadIdx.x; i < n; i += 256) {
        if (P.root_of[pool0 + i] == i) {
            my_roots++;
            const int a = P.area[pool0 + i];
            if (!(a < area_min || a > g.max_area)) {
                const int at = atomicAdd(&s_count, 1);
                if (at < kLdsCand) {
                    s_idx[at] = i;
                    s_key[at] = P.key[pool0 + i];
                } else if (at < P.cand_cap) {
                    scratch[at] = make_int2(i, P.key[pool0 + i]);
                }
            }
        }
    }
    if (my_roots) atomicAdd(&s_roots, my_roots);
    __syncthreads();
    if (threadIdx.x == 0) P.nroots[frame] = s_roots;
    int c = s_count;
    if (c > P.cand_cap) {  // more candidates than this workspace holds: the frame goes through the any-frame workspace (ctag_api.hip)
        if (threadIdx.x == 0) {
            atomicOr(&P.frame_flags[frame], CTAG_FLAG_POOL_OVERFLOW);
            P.ncand[frame] = 0;
        }
        return;
    }
    if (P.frame_flags[frame] & CTAG_FLAG_POOL_OVERFLOW) c = 0;
    if (threadIdx.x == 0) P.ncand[frame] = c;
    Candidate* out = P.cand + (size_t)frame * P.cand_cap;
    auto emit = [&](int idx, int rank) {
        Candidate cd;
        cd.root = idx;
        cd.area = P.area[pool0 + idx];
        cd.x_min = (int16_t)P.xmin[pool0 + idx];
        cd.y_min = (int16_t)P.ymin[pool0 + idx];
        cd.x_max = (int16_t)P.xmax[pool0 + idx];
        cd.y_max = (int16_t)P.ymax[pool0 + idx];
        out[rank] = cd;
    };
    if (c <= kLdsCand) {
        for (int i = threadIdx.x; i < c; i += 256) {
            const int k = s_key[i];
            int rank = 0;
            for (int j = 0; j < c; j++) rank += (s_key[j] < k) ? 1 : 0;
            emit(s_idx[i], rank);
        }
        return;
    }
    // More candidates than the LDS arrays hold (a frame of thousands of blobs): the same rank sort with the (index, key) pairs in
    // global memory, the keys passing through LDS a tile at a time; a thread ranks four of its candidates per sweep over the keys.
    // A key is the block-raster index of a component's first 2x2 block: no two components share one.
    for (int i = threadIdx.x; i < kLdsCand; i += 256) scratch[i] = make_int2(s_idx[i], s_key[i]);
    __syncthreads();
    for (int i0 = 0; i0 < c; i0 += 4 * 256) {
        int2 mine[4];
        int rank[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * 256 + threadIdx.x;
            mine[u] = i < c ? scratch[i] : make_int2(-1, 0x7fffffff);
        }
        for (int t0 = 0; t0 < c; t0 += kLdsCand) {
            const int tn = min(kLdsCand, c - t0);
            __syncthreads();
            for (int j = threadIdx.x; j < tn; j += 256) s_key[j] = scratch[t0 + j].y;
            __syncthreads();
            for (int j = 0; j < tn; j++) {
                const int kj = s_key[j];
#pragma unroll
                for (int u = 0; u < 4; u++) rank[u] += (kj < mine[u].y) ? 1 : 0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (mine[u].x >= 0) emit(mine[u].x, rank[u]);
    }
}
// ctag_get_counters: sums and maxima of the per-frame counts over the frames of a chunk (one block; on demand, never in a timed chain)
__global__ __launch_bounds__(256) void k_counters(const int32_t* nroots, const int32_t* ncand, const int32_t* nquads, const int32_t* nfeat, const int32_t* status,
                                                  const ctag_frame_result* results, int nframes, long long* out10) {
    __shared__ long long s_sum[5];
    __shared__ int s_max[5];
    if (threadIdx.x < 5) {
        s_sum[threadIdx.x] = 0;
        s_max[threadIdx.x] = 0;
    }
    __syncthreads();
    long long sum[5] = {0, 0, 0, 0, 0};
    int mx[5] = {0, 0, 0, 0, 0};
    for (int f = threadIdx.x; f < nframes; f += 256) {
        const int v[5] = {nroots[f], ncand[f], nquads[f], status[f] == CTAG_OK || status[f] == CTAG_NO_FEATURE ? nfeat[f] : 0, results ? results[f].n_markers : 0};
#pragma unroll
        for (int k = 0; k < 5; k++) {
            sum[k] += v[k];
            mx[k] = max(mx[k], v[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) {
        atomicAdd(reinterpret_cast<unsigned long long*>(&s_sum[k]), (unsigned long long)sum[k]);
        atomicMax(&s_max[k], mx[k]);
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        out10[threadIdx.x] = s_sum[threadIdx.x];
        out10[5 + threadIdx.x] = s_max[threadIdx.x];
    }
}
hipError_t launch_counters(int nframes, const Workspace& ws, const ctag_frame_result* results, long long* out10, hipStream_t s) {
    hipLaunchKernelGGL(k_counters, dim3(1), dim3(256), 0, s, ws.nroots, ws.ncand, ws.nquads, ws.nfeat, ws.status, results, nframes, out10);
    return hipGetLastError();
}

hipError_t launch_candidates(int nframes, const Workspace& ws, hipStream_t s) {
    hipLaunchKernelGGL(k_candidates, dim3(nframes), dim3(256), 0, s, sweep_ptrs(ws), ws.g, nframes, ws.kp.area_min);
    return hipGetLastError();
}

}  // namespace ctag
