/* ctag_oracle.h -- C ABI of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * The oracle is a plain CPU restatement of the reference's CylinderTag::detect() path
 * (/root/reference/CylinderTag.cpp:67-159 + corner_detector.cpp:28-1324) with its own replicas of the
 * OpenCV 4.5.3 primitives that path calls.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (cylindertag_amd/) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" against the real reference binary -- the reference ships no tests or
 * golden vectors and cannot be built here (needs OpenCV/Ceres, absent).  What pins this oracle instead is
 * listed in DESIGN.md ("Oracle pins").
 */
#ifndef CTAG_ORACLE_H
#define CTAG_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ctago_run ctago_run; /* one traced detect() run */

/* The tunables of every following run (include/ctag_types.h: struct ctag_params -- the reference's member constants
 * header/corner_detector.h:90,110,122,135-137,144 and literals corner_detector.cpp:71,88,285); NULL = the reference's values.
 * Not to be called while detections run on other threads. */
struct ctag_params;
void ctago_set_params(const struct ctag_params* p);

/* status codes mirror include/ctag.h */
enum { CTAGO_OK = 0, CTAGO_NO_CORNER = 1, CTAGO_NO_FEATURE = 2, CTAGO_ERR_LIMIT = -3, CTAGO_ERR_ARG = -1 };

/* Run the full restated detect() on one 8-bit gray frame and keep every intermediate stage.
 * state: dictionary (dict_rows x dict_cols codes 0..63), feature_size as in the .marker header. */
ctago_run* ctago_detect(const uint8_t* gray, int rows, int cols, ptrdiff_t row_stride, const int32_t* state,
                        int dict_rows, int dict_cols, int feature_size, int adaptive_thresh, int corner_subpix,
                        int subpix_dist);
void ctago_free(ctago_run*);
int ctago_status(const ctago_run*);

/* stage accessors; every "count" call returns the element count, every "get" copies into caller memory */
int ctago_half_rows(const ctago_run*);
int ctago_half_cols(const ctago_run*);
void ctago_get_half(const ctago_run*, uint8_t* dst);     /* half-res u8 image (a1) */
void ctago_get_binary(const ctago_run*, uint8_t* dst);   /* 0/255 binary (a2) */
void ctago_get_labels(const ctago_run*, int32_t* dst);   /* OpenCV-order labels, 0 = background (a3) */
int ctago_num_labels(const ctago_run*);                  /* incl. background */
void ctago_get_label_areas(const ctago_run*, int32_t* dst);
int ctago_num_candidates(const ctago_run*);              /* components passing the area filter, in order */
/* per candidate: label, area, x_min, y_min, x_max, y_max, has_quad(0/1), n_boundary */
void ctago_get_candidates(const ctago_run*, int32_t* dst8);
/* per candidate 8 floats (4 corners x,y; zeros when has_quad == 0), half-res coordinates (a4) */
void ctago_get_candidate_quads(const ctago_run*, float* dst8);
int ctago_num_quads(const ctago_run*);
void ctago_get_quads(const ctago_run*, float* dst8);      /* accepted quads in order, 8 floats each */
int ctago_num_features(const ctago_run*);
/* per feature 19 floats: 16 corner coords, centre x,y, angle; stage 0 = after featureRecovery (half-res),
 * 1 = after cornerObtain, 2 = after edgeRefine (or == stage 1 when subpix off) */
void ctago_get_features(const ctago_run*, int stage, float* dst19);
/* final result in the flat per-frame layout of include/ctag.h (ctag_frame_result) */
size_t ctago_result_bytes(void);
void ctago_get_result(const ctago_run*, void* dst);
/* markers before decoding (after markerOrganization), same flat layout, marker_id = -1, pos = -1 */
void ctago_get_premarkers(const ctago_run*, void* dst);

/* untraced fast path used as the timed CPU baseline: returns status, writes the flat result */
int ctago_detect_fast(const uint8_t* gray, int rows, int cols, ptrdiff_t row_stride, const int32_t* state,
                      int dict_rows, int dict_cols, int feature_size, int adaptive_thresh, int corner_subpix,
                      int subpix_dist, void* result);

/* frame-parallel ctago_detect_fast over n frames (frame i at frames + i*frame_stride) on `threads` std::threads
 * (<= 0: hardware_concurrency); results = n flat records.  Returns the number of threads used. */
int ctago_detect_many(const uint8_t* frames, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                      const int32_t* state, int dict_rows, int dict_cols, int feature_size, int adaptive_thresh,
                      int corner_subpix, int subpix_dist, int threads, void* results);
int ctago_hardware_concurrency(void);

/* probe of the PRODUCT header cylindertag_amd/csrc/ctag_refine.h on the host (see ctag_oracle.cpp) */
void ctago_refine_probe(const uint8_t* img, int rows, int cols, ptrdiff_t stride, int subpix, int n, const double* xy_nxny,
                        double* exact, double* fast, double* lit, int32_t* flag, double* mid);

/* primitive probes for unit tests */
/* cvtColor(BGR2GRAY) on 8-bit BGR (main.cpp:36,54): OpenCV's fixed-point (B*1868 + G*9617 + R*4899 + 8192) >> 14 */
void ctago_bgr2gray(const uint8_t* bgr, int rows, int cols, ptrdiff_t row_stride, uint8_t* gray);
void ctago_resize_half(const uint8_t* gray, int rows, int cols, ptrdiff_t row_stride, uint8_t* dst);
void ctago_threshold(const uint8_t* half, int rows, int cols, int tw, uint8_t* dst);
int ctago_ccl(const uint8_t* bin, int rows, int cols, int32_t* labels, int32_t* areas, int areas_cap);
void ctago_fitline_l2(const int32_t* xy, int n, float* line4);
void ctago_fitline_welsch(const int32_t* xy, int n, float* line4);
/* The two assumptions about OpenCV 4.5.3 nothing here can check, as switches (see OracleVariants in ctag_oracle.cpp): process-wide setting for every later
 * call (defaults 0, 8), and the Welsch fit under either placement of the min_err update for tests/cv2_pins.py to compare a real cv2.fitLine with. */
void ctago_set_variants(int welsch_minerr_in_loop, int resize_simd_lanes);
void ctago_fitline_welsch_variant(const int32_t* xy, int n, int variant, float* line4);
/* op: 0 atan2_64(a,b) 1 sin64(a) 2 cos64(a) 3 exp64(a) 4 acos64(a) 5 atan2_32 6 sin32 7 cos32 8 exp32
 *     9 fast_atan2_deg(a,b) 10 a/b (f64) 11 sqrt(a) (f64) 12 a/b (f32) 13 sqrtf(a) 14 round32(a) */
void ctago_math_probe(int op, int n, const double* a, const double* b, double* out);
/* 1 when built with -DCTAG_ORACLE_LIBM (glibc libm instead of ctag_math.h) */
int ctago_uses_libm(void);

#ifdef __cplusplus
}
#endif
#endif
