/* ctag.h -- C ABI of the MI355X-native CylinderTag detection front end (libctag_hip.so).
 *
 * Drop-in boundary: everything the reference does inside
 *     void CylinderTag::detect(const Mat& img, vector<MarkerInfo>& cornerList,
 *                              int adaptiveThresh = 5, const bool cornerSubPix = false, int cornerSubPixDist = 3)
 *     (/root/reference/header/CylinderTag.h:21, definition /root/reference/CylinderTag.cpp:67-159)
 * runs behind these entry points on one gfx950 device.  The reference has no FFI of its own (it is a single
 * C++ program); the C++ class in cylindertag_amd/csrc/CylinderTag.h keeps the reference's class interface
 * and calls this ABI, and INTEGRATION.md shows the few lines a maintainer of the reference changes.
 *
 * Plain pointers and sizes only; no C++ or torch types.  Every function returns a CTAG_* status
 * (include/ctag_types.h) and never throws.  The library has NO CPU fallback: without a usable HIP device
 * ctag_create fails with CTAG_ERR_HIP.
 */
#ifndef CTAG_H
#define CTAG_H
#include <stddef.h>
#include <stdint.h>

#include "ctag_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ctag_handle ctag_handle;

/* ---- lifetime ------------------------------------------------------------------------------------
 * Replaces the reference constructors CylinderTag(const string&) / CylinderTag(const Mat1i&)
 * (header/CylinderTag.h:15,18; CylinderTag.cpp:6-65): `state` is the dictionary, dict_rows x dict_cols
 * codes in 0..63 (check_dictionary, CylinderTag.cpp:56-65), `feature_size` the third number of the
 * .marker header (CylinderTag.cpp:24-26).  Returns CTAG_ERR_ARG for an illegal dictionary. */
int ctag_create(const int32_t* state, int dict_rows, int dict_cols, int feature_size, int device_id, ctag_handle** out);
void ctag_destroy(ctag_handle* h);

/* The detector's tunables.  In the reference they are member constants of corner_detector
 * (/root/reference/header/corner_detector.h:90,110,122,135-137,144) and literals of corner_detector.cpp (:71, :88, :285-288,337);
 * ctag_params_default fills in exactly those values and ctag_create uses them.  ctag_create_ex takes other values -- a
 * maintainer who edits the reference's constants passes the same numbers here and the results stay identical to the edited
 * reference (the CPU oracle takes the same struct).  Always start from ctag_params_default: it fills in struct_size, which
 * ctag_create_ex checks against the library's own sizeof(ctag_params).  Limits: 0 < dark_cap < 0.5, area_min >= 1, 0 < area_max_fraction <= 1,
 * finite positive thresholds; else CTAG_ERR_ARG. */
/* (struct ctag_params: include/ctag_types.h) */
void ctag_params_default(ctag_params* p);
int ctag_create_ex(const int32_t* state, int dict_rows, int dict_cols, int feature_size, int device_id, const ctag_params* params,
                   ctag_handle** out);

/* Parses a .marker text file exactly as CylinderTag::load_from_file does (CylinderTag.cpp:16-41).
 * On success *state is malloc()ed (free with ctag_free). */
int ctag_load_marker_file(const char* path, int32_t** state, int* dict_rows, int* dict_cols, int* feature_size);
void ctag_free(void* p);

/* ---- detection -----------------------------------------------------------------------------------
 * One frame, host memory in, host result out.  Replaces the body of CylinderTag::detect
 * (CylinderTag.cpp:67-159).  `gray` is an 8-bit single-channel image with `row_stride` bytes per row.
 * The return value is the frame status: CTAG_OK, CTAG_NO_CORNER / CTAG_NO_FEATURE (the reference's two
 * early returns, CylinderTag.cpp:87-96, which leave the caller's vector untouched), or an error. */
int ctag_detect_u8(ctag_handle* h, const uint8_t* gray, int rows, int cols, ptrdiff_t row_stride, int adaptive_thresh,
                   int corner_subpix, int subpix_dist, ctag_frame_result* out);

/* A batch of independent frames in HOST memory (frame i starts at frames + i*frame_stride).  `out` holds n
 * results in host memory.  Returns CTAG_OK when the batch ran (per-frame status is in out[i].status). */
int ctag_detect_batch_u8(ctag_handle* h, const uint8_t* frames, int n, int rows, int cols, ptrdiff_t row_stride,
                         ptrdiff_t frame_stride, int adaptive_thresh, int corner_subpix, int subpix_dist,
                         ctag_frame_result* out);
/* Page-locked host memory for frame buffers handed to ctag_detect_batch_u8: with it the PCIe upload of sub-chunk k+1
 * overlaps the detection of sub-chunk k (frame ingest of main.cpp:29,36,52-54).  NULL on failure. */
void* ctag_host_alloc(size_t bytes);
void ctag_host_free(void* p);

/* A batch of frames already resident in DEVICE memory; results are written to DEVICE memory `out_dev`
 * (n records).  Work is enqueued on the handle's stream and this call returns without waiting; use
 * ctag_sync() (or stream ordering, see below).  This is the entry point the throughput bench times.
 *
 * Frames that need more than the batch workspace's pools.  The reference keeps EVERY component of 30 px .. 1 % of the frame
 * and walks them all (corner_detector.cpp:81-107,171-405); the batch workspace holds what a frame ordinarily needs (at
 * 1080p: 2048 such components, 262 144 reserved boundary points; scaled with the frame's area).  A frame beyond that --
 * thousands of blobs, fine texture -- is not failed: it is run again, alone, through a workspace whose pools no frame of
 * its size can exhaust, and its record is the reference's like any other.  Host-memory calls (ctag_detect_u8 / _batch_u8 /
 * _bgr8) do that before they return.  Device-memory calls do it at the handle's next synchronisation point: ctag_sync(), a call
 * with CTAG_OPT_TIMING on, ctag_pose_batch_device, ctag_pack_results / ctag_gather_end; a caller that relies on stream ordering
 * alone sees such a frame's record with status CTAG_PENDING until then (the source frames must stay valid that long).  The list of
 * waiting frames holds 65 536 entries between two synchronisation points; a frame beyond that gets the terminal status CTAG_ERR_LIMIT
 * with CTAG_FLAG_POOL_OVERFLOW set (and the synchronisation point returns CTAG_ERR_LIMIT): it is never left pending. */
int ctag_detect_batch_device(ctag_handle* h, const uint8_t* frames_dev, int n, int rows, int cols, ptrdiff_t row_stride,
                             ptrdiff_t frame_stride, int adaptive_thresh, int corner_subpix, int subpix_dist,
                             ctag_frame_result* out_dev);
/* ---- BGR frames -------------------------------------------------------------------------------------
 * The reference's stream loop converts every camera frame before detect(): cvtColor(frame, gray, COLOR_BGR2GRAY)
 * (main.cpp:36,52-54).  These entry points take the 8-bit BGR frames themselves (3 bytes per pixel, B G R order, `row_stride`
 * >= 3 * cols bytes) and do that conversion on the device with OpenCV's fixed-point weights -- (B*1868 + G*9617 + R*4899 + 8192)
 * >> 14 -- in front of the same detection chain; a host feed then uploads the camera's bytes as they are instead of spending
 * a host core per ~1 G pixels/s on the conversion.  Results equal ctag_detect_*_u8 on the converted frames. */
int ctag_detect_bgr8(ctag_handle* h, const uint8_t* bgr, int rows, int cols, ptrdiff_t row_stride, int adaptive_thresh, int corner_subpix,
                     int subpix_dist, ctag_frame_result* out);
int ctag_detect_batch_bgr8(ctag_handle* h, const uint8_t* bgr, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                           int adaptive_thresh, int corner_subpix, int subpix_dist, ctag_frame_result* out);
int ctag_detect_batch_bgr8_device(ctag_handle* h, const uint8_t* bgr_dev, int n, int rows, int cols, ptrdiff_t row_stride,
                                  ptrdiff_t frame_stride, int adaptive_thresh, int corner_subpix, int subpix_dist,
                                  ctag_frame_result* out_dev);
/* ---- one frame per call, not waited for ------------------------------------------------------------------------------------------
 * The reference's camera loop (main.cpp:44-61) calls detect() on every frame and waits.  With this pair the upload of frame k + 1
 * overlaps the detection of frame k: ctag_submit_u8 starts the upload (its own stream) and enqueues the detection behind it,
 * ctag_collect waits for the OLDEST submitted frame and returns its record and status (as ctag_detect_u8 would).  Up to two frames may
 * be in flight (a third submit returns CTAG_ERR_ARG until one is collected); a frame buffer must stay valid until its collect
 * returns (page-locked memory, ctag_host_alloc, makes the upload asynchronous). */
int ctag_submit_u8(ctag_handle* h, const uint8_t* gray, int rows, int cols, ptrdiff_t row_stride, int adaptive_thresh, int corner_subpix,
                   int subpix_dist);
int ctag_collect(ctag_handle* h, ctag_frame_result* out);
int ctag_sync(ctag_handle* h);
/* HIP stream (hipStream_t) all work of this handle is enqueued on */
void* ctag_stream(ctag_handle* h);

/* ---- options / introspection --------------------------------------------------------------------- */
#define CTAG_OPT_MAX_CHUNK 1      /* frames processed per pass (workspace is sized for it); default 1024 */
#define CTAG_OPT_TIMING 2         /* 1: bracket every kernel with HIP events (ctag_get_timings) */
#define CTAG_OPT_KEEP_PREMARKERS 3 /* 1: also keep every frame's markers before decoding (read by the test kit, include/ctag_testkit.h) */
#define CTAG_OPT_HOST_SUBCHUNK 4   /* frames per upload/detect pipeline step of ctag_detect_batch_u8; default 128 */
#define CTAG_OPT_GRAPH 5           /* replay a chunk whose pointers / sizes / parameters repeat as one hipGraph launch: 0 never, 1 every chunk, 2 (default)
                                      calls of up to 4 frames only -- one frame per call in a loop gains 0.03-0.04 ms of 0.7; batches gain
                                      nothing, their chain is not launch-bound */
#define CTAG_OPT_WAVE_POINTS 6     /* components whose boundary can hold more than this many points get a wave of their own instead of 8 lanes of a
                                      shared one (the longest boundary decides how long a one-frame call takes); 0 = automatic: every component of a call
                                      of up to 4 frames (no shared waves there, hence no second stream to fork and join), never for batches.  Results do not depend on it. */
#define CTAG_OPT_STREAMS 8         /* 2 (default): a chunk of >= 256 frames of a device-memory call runs as two halves on two internal streams and workspaces -- the
                                      tail of one half's kernels overlaps the other half's next kernel; the second stream forks from and joins the handle's stream
                                      inside the call, so callers order against ctag_stream() as before.  1: one stream; 3 / 4: thirds / quarters on as many streams (measured on
                                      4096-frame chunks: 2 and 3 the same within the spread, 4 slower).  Chunks of 448..1023 frames stay whole (their halves run slower than the chunk,
                                      docs/history.md).  A call with CTAG_OPT_TIMING on
                                      always uses one stream (the per-kernel events would time the neighbour's kernels too).  Results do not depend on it. */
#define CTAG_OPT_EXPAND_EXACT 9    /* developer aid: 1 makes expand_line (corner_detector.cpp:125-169) refit the line with the reference's own arithmetic at EVERY step
                                      instead of deciding most distance tests from a filtered estimate (k_quad.hip).  Results do not depend on it -- that is what
                                      the option exists to check. */
#define CTAG_OPT_FUSED_SWEEP 7     /* the threshold + label sweep as k_decimate_mask -> k_threshold_ccl (thresholds where the half-size pixels are computed, hands
                                      1 bit per pixel on; frames whose half size is a multiple of 320 x 5 -- 1080p, 4K, 8K, 1920x1200, 1280x720 ... -- with adaptiveThresh 5):
                                      0 never, 1 (default) batches of 512 frames' worth of bands and more, 2 whenever the frame size allows.  Results do not depend on it. */
#define CTAG_OPT_BGR_DIRECT 10     /* 1 (default): ctag_detect_batch_bgr8_device hands BGR frames of a size the fused sweep takes (see CTAG_OPT_FUSED_SWEEP), with
                                      16-byte aligned rows and frames, to the chain as they are -- the decimation kernel and edgeRefine convert (cvtColor(BGR2GRAY),
                                      main.cpp:36,52-54) as they load, no gray image is written; 0: always convert into a gray image first.  Results do not depend on it. */
int ctag_set_option(ctag_handle* h, int option, int64_t value);

/* Per-stage device time of the LAST ctag_detect_batch_* call, milliseconds measured with HIP events on
 * the handle's stream (needs CTAG_OPT_TIMING).  Order: decimate, threshold_ccl, seam_merge, resolve,
 * candidates, quad_pack, quad_edges, quad_edges_big, line_sort, welsch, quad_final, features, edge_refine,
 * markers (one entry per kernel; names from ctag_stage_name; quad_edges_big = the whole-wave builds of the boundary kernel: in calls of
 * up to 4 frames they run beside quad_edges on a second stream and the entry is the time the chain waited for them after quad_edges).
 * Returns the number of stages written. */
#define CTAG_NUM_STAGES 14
int ctag_get_timings(ctag_handle* h, float* ms, int capacity);
const char* ctag_stage_name(int stage);
/* Per-frame counts of the LAST chunk the handle processed (at most CTAG_OPT_MAX_CHUNK frames -- with CTAG_OPT_STREAMS >= 2, the default, the
 * last PIECE of that chunk: the part its last stream ran; frames completed through the any-frame workspace do not count as a chunk;
 * one small kernel, waits for the stream): sums and maxima of components / candidates / quads / features / markers -- the numbers behind the reference's two
 * log lines (CylinderTag.cpp:88,94) -- and how many frames have needed the any-frame workspace so far. */
int ctag_get_counters(ctag_handle* h, ctag_counters* out);
const char* ctag_strerror(int status);
int ctag_version(void);

#ifdef __cplusplus
}
#endif
#endif
