/* ctag_testkit.h -- C ABI of libctag_testkit.so: TEST AND BENCH SCAFFOLDING, not part of the product.
 *
 * Nothing here replaces an interface of the reference (/root/reference/header/CylinderTag.h:15-30 is covered by
 * include/ctag.h alone); a host that links libctag_hip.so never needs this library.  It holds what the parity tests,
 * bench.py and the developer tools need around the product:
 *   - parity probes: the intermediates of a frame of the last chunk (stages a1..a8 of SURVEY.md 8(a)),
 *   - the shared deterministic math (cylindertag_amd/csrc/ctag_math.h) evaluated on the device,
 *   - the synthetic frame generators,
 *   - the unpack half of ctag_gather_end on a caller-built gathered buffer (the multi-rank device path on one GPU).
 * libctag_testkit.so links against libctag_hip.so and reaches into a handle only through the private accessors of
 * cylindertag_amd/csrc/ctag_internal.h.
 */
#ifndef CTAG_TESTKIT_H
#define CTAG_TESTKIT_H
#include <stddef.h>
#include <stdint.h>

#include "ctag.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- parity probes: intermediates of frame `frame` of the last chunk -------------------------------------- */
#define CTAG_DBG_HALF 1       /* uint8  [hrows*hcols]   half-resolution image (a1) */
#define CTAG_DBG_LABELS 2     /* int32  [hrows*hcols]   0 = background, else 1 + frame-local root id (a2,a3) */
#define CTAG_DBG_CANDIDATES 3 /* int32  [ncand*8]       area, x_min, y_min, x_max, y_max, has_quad, n_boundary, root */
#define CTAG_DBG_CAND_QUADS 4 /* float  [ncand*8] */
#define CTAG_DBG_FEATURES0 5  /* float  [nfeat*19]      after featureRecovery (half-res) */
#define CTAG_DBG_FEATURES1 6  /* float  [nfeat*19]      after cornerObtain */
#define CTAG_DBG_FEATURES2 7  /* float  [nfeat*19]      after edgeRefine */
#define CTAG_DBG_PREMARKERS 8 /* ctag_frame_result      markers before decoding (needs CTAG_OPT_KEEP_PREMARKERS) */
#define CTAG_DBG_GRAY 9       /* uint8  [rows*cols]     gray image the BGR entry points computed (ctag_detect_batch_bgr8...) */
#define CTAG_DBG_MASK 11      /* uint8  [hrows*hcols]   0 / 1: the adaptive-threshold mask of the fused sweep (k_decimate_mask); -1 when the last chunk took the two-kernel form */
#define CTAG_DBG_LINES 10     /* int32  [nlines]        point count of every edge cluster handed to the Welsch fit (a4) */
/* returns the number of ELEMENTS available (copies min(available, capacity) elements), < 0 on error */
long ctag_debug_fetch(ctag_handle* h, int frame, int what, void* dst, size_t capacity_elems);

/* evaluates the shared deterministic math on the device; op codes as oracle/ctag_oracle.h:ctago_math_probe.
 * Host arrays in/out. */
int ctag_math_probe(ctag_handle* h, int op, int n, const double* a, const double* b, double* out);

/* ---- the multi-rank unpack on one GPU -----------------------------------------------------------------------
 * Runs exactly what ctag_gather_end runs after the payload all-gather (include/ctag_gather.h): the segment table of a
 * `world`-rank job over n_total frames (shard r = the ctag_shard_range of rank r, placed at r * width in the
 * gathered buffer) and the unpack kernels, on the handle's gather stream; waits for completion.  `gathered_dev` is what
 * the all-gather would have delivered: world packed shards, each padded to `width` bytes. */
int ctag_testkit_unpack_gathered(ctag_handle* h, const void* gathered_dev, int n_total, int world, uint64_t width,
                                 ctag_frame_result* out_dev);

/* Occupies the handle's stream (ctag_stream) with a kernel that spins for about `milliseconds` (<= 10 000): what a late peer looks
 * like to the gather's bounded waits (include/ctag_gather.h: ctag_gather_set_timeout) -- work enqueued behind it, the gather stream
 * included, does not complete until it ends.  Returns at once. */
int ctag_testkit_stall_stream(ctag_handle* h, int milliseconds);

/* ---- synthetic frames -------------------------------------------------------------
 * Frame f is a pure function of (seed + f): gray background with a ramp and noise plus `markers` planted
 * CylinderTag strips of the handle's dictionary.  The same code renders on the device and on the host. */
typedef struct ctag_synth_truth {
    int32_t n_markers;
    int32_t dict_row[8];
    float strip_len[8];      /* L, full-res pixels */
    float corners[8][8];     /* image positions of the strip's 4 outer corners */
} ctag_synth_truth;
int ctag_synth_frames_device(ctag_handle* h, uint8_t* frames_dev, int first_frame, int n, int rows, int cols,
                             ptrdiff_t row_stride, ptrdiff_t frame_stride, uint64_t seed, int markers_per_frame);
int ctag_synth_frame_host(const int32_t* state, int dict_rows, int dict_cols, uint8_t* frame, int frame_index, int rows,
                          int cols, ptrdiff_t row_stride, uint64_t seed, int markers_per_frame, ctag_synth_truth* truth);
/* planted markers of synthetic frame `frame_index` without rendering it */
int ctag_synth_layout_truth(const int32_t* state, int dict_rows, int dict_cols, int frame_index, int rows, int cols, uint64_t seed,
                            int markers_per_frame, ctag_synth_truth* truth);

/* ---- synthetic 3-D scenes (BASELINE config 5: detect() + estimatePose with known answers) ---------------------
 * The same strips printed on cylinders (strip height 60 mm, a radius fixed per dictionary row) in front of a pinhole
 * camera (fx, fy, cx, cy; no distortion), every marker with a planted rigid pose; the image is ray-cast.
 * ctag_synth3d_model gives the objects' 3-D corner lists -- the `.model` of CylinderTag.cpp:168-188 for them:
 * corners[row][feature*8 + k][3] in mm, corner order as detect() emits it -- ready for ctag_model_create with
 * marker ids 0..dict_rows-1. */
typedef struct ctag_synth3d_truth {
    int32_t n_markers;
    int32_t dict_row[8];
    double R[8][9];    /* object -> camera rotation, row-major */
    double t[8][3];    /* mm */
    double radius[8];  /* mm */
} ctag_synth3d_truth;
int ctag_synth3d_frames_device(ctag_handle* h, uint8_t* frames_dev, int first_frame, int n, int rows, int cols, ptrdiff_t row_stride,
                               ptrdiff_t frame_stride, uint64_t seed, int markers_per_frame, double fx, double fy, double cx, double cy);
int ctag_synth3d_frame_host(const int32_t* state, int dict_rows, int dict_cols, uint8_t* frame, int frame_index, int rows, int cols,
                            ptrdiff_t row_stride, uint64_t seed, int markers_per_frame, double fx, double fy, double cx, double cy,
                            ctag_synth3d_truth* truth);
int ctag_synth3d_model(const int32_t* state, int dict_rows, int dict_cols, float* corners);

#ifdef __cplusplus
}
#endif
#endif
