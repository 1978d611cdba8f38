// ctag_internal.h -- device workspace layout and kernel launch prototypes (not part of the public ABI).
//
// Pipeline per chunk of frames (all on one HIP stream; see DESIGN.md for the data-flow picture):
//   K1 k_decimate          full-res u8 -> half-res u8 (bicubic 2x)              CylinderTag.cpp:79-80
//   K2 k_threshold_ccl     half-res u8 -> per-tile u16 labels + component pool  corner_detector.cpp:28-97
//   K3 k_seam_merge        union components across tile seams (global UF)        (part of :82)
//   K4 k_resolve           flatten roots, fold tile-local stats into roots       (part of :82, :87-91)
//   K5 k_candidates        area filter + OpenCV label order -> candidate list    corner_detector.cpp:87-106
//   K6 k_quad              per candidate: boundary -> 4 lines -> quad            corner_detector.cpp:171-463
//   K7 k_features          per frame: quad pairing, cornerObtain                 corner_detector.cpp:465-598
//   K8 k_edge_refine       per (feature, quad): sub-pixel edge refinement        corner_detector.cpp:600-951
//   K9 k_markers           per frame: grouping, cross ratios, dictionary decode  corner_detector.cpp:976-1324
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/ctag_types.h"

struct ctag_handle;

namespace ctag {

// ---- CCL tile geometry (half-resolution pixels) ---------------------------------------------------------
constexpr int kTileW = 320;             // 5 x 64-bit mask words per tile row
constexpr int kTileH = 30;
constexpr int kTileWords = kTileW / 64;
#ifndef CTAG_RUN_CAP
#define CTAG_RUN_CAP 2048
#endif
constexpr int kRunCap = CTAG_RUN_CAP;           // row-runs per tile the first CCL pass holds in LDS; a tile with more takes the second pass
constexpr int kSlotCap = 128;           // tile-local components a tile publishes in the first CCL pass; a tile with more takes the second pass
constexpr int kRunCapBig = 4864;        // second CCL pass (k_threshold_ccl_big): >= 160 runs per row x 30 rows, a multiple of the block size
constexpr int kSlotCapBig = 2560;       // ... and >= 160 x 15 isolated pixels: its caps hold any tile
constexpr int kPoolCapMin = 8192;       // tile-local components per frame (global pool): max(this, 256 per CCL tile), FrameGeom::pool_cap
// Pools of the stages behind the label sweep are sized per workspace (Workspace::cand_cap / line_cap / cl_cap): the batch
// workspace holds what a frame of its size ordinarily needs (make_caps, ctag_api.hip); a frame that needs more is flagged
// CTAG_FLAG_POOL_OVERFLOW, reported CTAG_PENDING and run again, alone, through a workspace whose pools hold ANY frame of
// that size (Workspace::big) -- the reference has no such limits (corner_detector.cpp:81-107,171-405).
constexpr int kCandCapMin = 2048;       // area-filtered candidates per frame the batch workspace holds at least
constexpr int kLdsCand = 2048;          // candidates k_candidates / k_pack sort in LDS; more take their global-memory paths
constexpr int kLdsLines = 8192;         // fitted edges k_line_sort ranks in LDS; more take a counting sort
constexpr int kQuadStride = 1024;       // accepted quads per frame K7 keeps (CTAG_MAX_QUADS = 1000: more is the reference's UB)
constexpr int kMaxThreshWin = 32;       // supported adaptiveThresh window range [1, 32]

// ---- K6 limits ----------------------------------------------------------------------------------------------
constexpr int kLatencyFrames = 4;           // calls with at most this many frames are tuned for the latency of the call (launch_quads, hipGraph replay)
constexpr int kLatLines = 2048;             // edges per frame / points per edge the one-wave-per-restart Welsch kernel of such calls holds;
constexpr int kLatPoints = 512;             // a frame beyond either takes the batch kernel
constexpr int kClPoolMin = 262144;        // edge-cluster points per frame the batch workspace holds at least (a candidate reserves its boundary capacity + 64)
constexpr int kPickN = 256;               // point counts covered by the precomputed cv::RNG pick table (bytes)
constexpr int kPickN2 = 4096;             // ... and by its 16-bit continuation [kPickN, kPickN2); longer edges replay cv::RNG on the device
constexpr int kMaxDictCells = 2048;       // dictionary rows*cols supported by K9 (reference dictionary: 41*12)

struct Candidate {  // one area-filtered connected component, in OpenCV label order
    int32_t root;   // pool index of the root component (frame-local)
    int32_t area;
    int16_t x_min, y_min, x_max, y_max;
};

struct LineDesc {   // one edge cluster in the frame's cluster pool
    uint32_t off;
    int32_t n;
};
struct CandAux {    // k_quad_edges -> k_quad_final
    int32_t line0;  // first of 4 consecutive line ids, -1 when the component produced no 4 edges
    float acx, acy; // boundary centroid (area_center)
    int32_t n_boundary;
};

struct QuadOut {    // K6 output per candidate
    int32_t valid;
    int32_t n_boundary;
    float c[8];     // 4 corners x,y (half-res coordinates), sorted as the reference leaves them
};

struct FeatureDev { // K7/K8 output per feature (reference struct featureInfo)
    float c[16];
    float center[2];
    float angle;
    int32_t pad;
};

// The detector's tunables as the kernels consume them (include/ctag.h: ctag_params; built once per handle)
struct KParams {
    float thr_line, thr_expand, rac, angle, vertical;  // threshold_line / _expand / _RAC / _angle / _vertical
    float cr_id[4], cr_lo[4], cr_hi[4];                // ID_cr_correspond, cr_covariance_left, cr_covariance_right
    float dark_cap;
    int tcap;                 // pixel u is below the cap iff u < tcap (77 for 0.3)
    uint32_t tcap4;           // tcap in every byte
    int thr_dim;              // threshold table is thr_dim x thr_dim; mn + mx >= thr_dim means T = tcap
    const uint8_t* thr_table; // device memory, owned by the handle
    int area_min;
    double area_max_fraction;
    int c2_far;               // squared triplet norm c2 >= c2_far  <=>  (float)sqrt(c2) >  collinear_cost   (2 for 1.05)
    int c2_near;              // c2 <= c2_near                      <=>  (float)sqrt(c2) <  collinear_cost   (1 for 1.05)
    float expand_eps;         // expand_line's filter band (k_quad.hip: sg_expand_line): 3e-6, or +inf with CTAG_OPT_EXPAND_EXACT
};

// A frame of a DEVICE-memory call that exceeded the batch workspace's pools: everything the library needs to run it again
// through the any-frame workspace at the handle's next synchronisation point (ctag_sync and every call that waits).
struct PendingRec {
    const uint8_t* src;        // the frame in the caller's device memory (gray, or BGR when ch == 3)
    ctag_frame_result* out;    // its record
    int64_t row_stride;
    int32_t rows, cols, ch;
    int32_t tw, subpix, dist;
};
struct PendingCtx {            // per chunk, by value to k_markers; list == nullptr: no list (host-memory calls find CTAG_PENDING in the records)
    PendingRec* list;
    int32_t* count;
    int32_t cap;
    const uint8_t* src;        // first frame of the chunk in the caller's memory
    int64_t frame_stride, row_stride;
    int32_t rows, cols, ch, tw, subpix, dist;
};

struct FrameGeom {
    int rows, cols;            // full-res
    int hrows, hcols;          // half-res
    int hp;                    // half-res pitch (bytes)
    int lp;                    // label pitch (elements)
    int tw;                    // adaptiveThresh window
    int trows, tcols;          // threshold tile grid
    int tiles_x, tiles_y;      // CCL tile grid
    int max_area;              // round(0.01*hcols*hrows)
    int pool_cap;              // component pool entries per frame
};

// Per-chunk device workspace (structure of arrays; one allocation, carved by Workspace::layout()).
struct Workspace {
    int chunk_frames = 0;
    FrameGeom g{};
    uint8_t* half = nullptr;        // [F][hrows][hp]
    uint16_t* labels = nullptr;     // [F][hrows][lp]   tile-local label (0 = background)
    int32_t* tile_base = nullptr;   // [F][tiles]       pool offset of each tile's local components
    int32_t* tile_dirty = nullptr;  // [F][tiles]       bit b: label block b of the tile (8 rows x 64 columns) is not all zero (K2 skips rewriting zeros over zeros)
    int32_t* frame_ncomp = nullptr; // [F]              pool fill
    int32_t* ovf_count = nullptr;   // [1]              tiles handed to the second CCL pass (k_threshold_ccl_big) ...
    int32_t* ovf_list = nullptr;    // [F * tiles]      ... as frame * tiles + tile
    uint32_t* frame_flags = nullptr;// [F]
    // component pool, [F][g.pool_cap] each
    uint32_t* parent = nullptr;
    int32_t* root_of = nullptr;
    int32_t* area = nullptr;
    int32_t* xmin = nullptr;
    int32_t* ymin = nullptr;
    int32_t* xmax = nullptr;
    int32_t* ymax = nullptr;
    int32_t* key = nullptr;
    int32_t* pool_tile = nullptr;
    int32_t* member_head = nullptr;
    int32_t* member_next = nullptr;
    // candidates
    int32_t* ncand = nullptr;       // [F]
    int32_t* nroots = nullptr;      // [F]              connected components the label sweep published (ctag_get_counters)
    int cand_cap = 0;               // candidates per frame this workspace holds
    int line_cap = 0;               // fitted edges per frame (4 per candidate)
    uint32_t cl_cap = 0;            // edge-cluster points per frame
    bool big = false;               // the any-frame workspace (one frame, worst-case pools): an overflow here is final (CTAG_ERR_LIMIT)
    Candidate* cand = nullptr;      // [F][cand_cap]
    QuadOut* quads = nullptr;       // [F][cand_cap]
    int32_t* line_count = nullptr;  // [F]
    int32_t* clp_used = nullptr;    // [F]
    uint32_t* cl_pool = nullptr;    // [F][cl_cap]
    LineDesc* line_desc = nullptr;  // [F][line_cap]
    int32_t* line_sorted = nullptr; // [F][line_cap]
    int32_t* line_long = nullptr;   // [F] edges of more than 10 points (the first ranks of line_sorted)
    float* line_fit = nullptr;      // [F][line_cap][4]
    CandAux* cand_aux = nullptr;    // [F][cand_cap]   (k_candidates' sort scratch before K6 writes it)
    int32_t* npacks = nullptr;      // [F][2]          packs, oversize components
    uint32_t* packs = nullptr;      // [F][cand_cap]
    uint32_t* pack_order = nullptr; // [F][cand_cap]
    float* welsch_rs = nullptr;     // [min(F, kLatencyFrames)][kLatLines][20][6]: line + err (as a double) of every restart, few-frame calls only
    const uint8_t* pick_table = nullptr;  // [kPickN][20][10], owned by the handle
    const uint16_t* pick_table16 = nullptr;  // [kPickN2 - kPickN][20][10]
    hipStream_t aux_stream = nullptr;     // owned by the handle: side branch for few-frame calls (launch_quads)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int wave_points = 0;                  // CTAG_OPT_WAVE_POINTS (0 = automatic)
    int fuse_mode = -1;                   // CTAG_OPT_FUSED_SWEEP (-1 = not set: CTAG_FUSED_SWEEP from the environment, else 1)
    KParams kp{};                         // the handle's tunables
    // features
    void* quad_derived = nullptr;   // [F][kQuadStride] x 48 B (K7 scratch)
    int32_t* quad_index = nullptr;  // [F][kQuadStride]
    int32_t* nquads = nullptr;      // [F]
    int32_t* nfeat = nullptr;       // [F]
    int32_t* status = nullptr;      // [F]
    FeatureDev* feat0 = nullptr;    // [F][CTAG_MAX_FEATURES] after featureRecovery (half-res)
    FeatureDev* feat1 = nullptr;    // after cornerObtain
    FeatureDev* feat2 = nullptr;    // after edgeRefine
    ctag_frame_result* premarkers = nullptr;  // [F] (debug: before decode)
    double* refine_n0 = nullptr;    // [F][CTAG_MAX_FEATURES * 2][4][128] K8: search kernel -> sums kernel
    int32_t* frame_long = nullptr;  // [F] K8: the frame has a quad that needs the one-kernel form
    // cubic tap tables of the general (odd-size) decimation: [hcols] / [hcols][4] / [hrows] / [hrows][4]
    int32_t* rz_xofs = nullptr;
    int16_t* rz_alpha = nullptr;
    int32_t* rz_yofs = nullptr;
    int16_t* rz_beta = nullptr;
    void* base = nullptr;
    size_t bytes = 0;
};

struct DetectParams {
    int adaptive_thresh;
    int corner_subpix;
    int subpix_dist;
    int feature_size;
    int dict_rows, dict_cols;
    const int32_t* dict;  // device pointer
    const uint32_t* dict_pos;  // device pointer: [dict_rows][64] column sets per symbol (k_markers), null for > 32 columns
    int channels = 1;  // 3: the chunk's frames are 8-bit BGR (3 bytes per pixel) and the kernels that read pixels -- the fused decimation, edgeRefine -- convert as they load (bgr_fused)
};
#if defined(__HIPCC__)
// cvtColor(BGR2GRAY) on 8-bit pixels is fixed point in OpenCV: (B*1868 + G*9617 + R*4899 + 8192) >> 14 (RGB2Gray<uchar>: B2Y, G2Y, R2Y at yuv_shift 14,
// [OCV-recall of color_rgb.simd.hpp]; the same formula as the host BMP reader, csrc/ctag_io.h)
__device__ __forceinline__ uint32_t gray_of(uint32_t b, uint32_t g, uint32_t r) { return (b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14; }
// ... of four pixels at once: twelve bytes B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3 -> four gray bytes.  The weights split into bytes (1868 = 7 * 256 + 76,
// 9617 = 37 * 256 + 145, 4899 = 19 * 256 + 35), so a pixel is two v_dot4_u32_u8 of its three bytes against (76, 145, 35) and (7, 37, 19) -- no byte extraction
__device__ __forceinline__ uint32_t gray4_of(uint32_t w0, uint32_t w1, uint32_t w2) {
    constexpr uint32_t kLo = 76u | (145u << 8) | (35u << 16), kHi = 7u | (37u << 8) | (19u << 16);
    constexpr uint32_t kLo1 = kLo << 8, kHi1 = kHi << 8;  // the weights against bytes 1..3 of a word (pixel 3 sits there)
    const uint32_t p1 = __builtin_amdgcn_alignbyte(w1, w0, 3), p2 = __builtin_amdgcn_alignbyte(w2, w1, 2);
    const uint32_t g0 = ((__builtin_amdgcn_udot4(w0, kHi, 0u, false) << 8) + __builtin_amdgcn_udot4(w0, kLo, 8192u, false)) >> 14;
    const uint32_t g1 = ((__builtin_amdgcn_udot4(p1, kHi, 0u, false) << 8) + __builtin_amdgcn_udot4(p1, kLo, 8192u, false)) >> 14;
    const uint32_t g2 = ((__builtin_amdgcn_udot4(p2, kHi, 0u, false) << 8) + __builtin_amdgcn_udot4(p2, kLo, 8192u, false)) >> 14;
    const uint32_t g3 = ((__builtin_amdgcn_udot4(w2, kHi1, 0u, false) << 8) + __builtin_amdgcn_udot4(w2, kLo1, 8192u, false)) >> 14;
    return g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
}
#endif

// kernel launchers (each enqueues on `s`, returns hipGetLastError())
hipError_t launch_zero_counters(int nframes, const Workspace& ws, hipStream_t s);  // frame_ncomp, frame_flags, line_count, clp_used, ovf_count
// fused: k_decimate_mask + the mask front end of K2 (1 bit per pixel between them, no `half`) -- sweep_fused says when that form applies
bool sweep_fused(const uint8_t* frames, ptrdiff_t frame_stride, ptrdiff_t row_stride, int nframes, const Workspace& ws, bool always = false);
bool sweep_fused_size(int rows, int cols, int tw, int fuse_mode);
bool sweep_fused_batch(int rows, int cols, int nframes, int fuse_mode);  // a call of that many frames is a batch for the fused sweep (else: short-band kernels)
hipError_t launch_decimate(const uint8_t* frames, ptrdiff_t frame_stride, ptrdiff_t row_stride, int nframes, const Workspace& ws, hipStream_t s, bool fused = false, bool zero_too = false,
                           int channels = 1);  // channels = 3: BGR frames, fused form only
hipError_t launch_threshold_ccl(int nframes, const Workspace& ws, hipStream_t s, bool fused = false);
hipError_t launch_seam_merge(int nframes, const Workspace& ws, hipStream_t s);
hipError_t launch_resolve(int nframes, const Workspace& ws, hipStream_t s);
hipError_t launch_candidates(int nframes, const Workspace& ws, hipStream_t s);
hipError_t launch_quads(int nframes, const Workspace& ws, hipStream_t s, hipEvent_t* ev5 = nullptr, const uint8_t* mask = nullptr);  // mask: the chunk took the fused sweep (ws.half holds its threshold mask)  // ev5: 5 events, one after each kernel but the last
hipError_t launch_features(int nframes, const Workspace& ws, const DetectParams& p, hipStream_t s);
hipError_t launch_edge_refine(const uint8_t* frames, ptrdiff_t frame_stride, ptrdiff_t row_stride, int nframes, const Workspace& ws, const DetectParams& p, hipStream_t s);
// sums and maxima over the frames of the last chunk: components, candidates, quads, features, markers -> out10 (device, 5 x int64 sums then 5 x int64 maxima)
hipError_t launch_counters(int nframes, const Workspace& ws, const ctag_frame_result* results, long long* out10, hipStream_t s);
hipError_t launch_markers(int nframes, const Workspace& ws, const DetectParams& p, ctag_frame_result* out, const PendingCtx& pend, hipStream_t s);
size_t threshold_ccl_lds_bytes(int tw);
// adaptive-threshold bound table for a dark cap (host): returns false when the cap is outside what K2's packed compares hold
bool build_threshold_table(float dark_cap, uint8_t* table /* 256*256 */, int* dim, int* tcap);
void build_pick_table(uint8_t* table, uint16_t* table16);  // kPickN*20*10 bytes, (kPickN2 - kPickN)*20*10 halfwords


// accessors of the opaque handle for the pose back end (k_pose.hip)
void** handle_pose_slot(struct ::ctag_handle* h, void (*free_fn)(void*));
// ... and for the multi-GPU gather layer (ctag_gather.hip)
void** handle_gather_slot(struct ::ctag_handle* h, void (*free_fn)(void*));
bool handle_timing(const struct ::ctag_handle* h);
// completes the frames of earlier device-memory calls that wait for the any-frame workspace (CTAG_PENDING records); waits for the
// handle's stream when there may be any.  Every entry point that reads result records on the device calls it first.
int handle_finish_pending(struct ::ctag_handle* h);
// the gather layer defers that wait (ctag_gather_begin must not stall the host behind the detection it follows): `may_have` = a device-memory call was
// enqueued since the list was last read; *count_dev = the list's length on the device (written by k_markers: read it in stream order behind the
// detection); *gen counts such calls.  handle_pending_clean: the length read behind call number `gen` was 0 -- nothing is pending unless a later call ran.
bool handle_pending_state(struct ::ctag_handle* h, const int32_t** count_dev, uint64_t* gen);
void handle_pending_clean(struct ::ctag_handle* h, uint64_t gen);
int handle_device(const struct ::ctag_handle* h);

// Private window for libctag_testkit.so (include/ctag_testkit.h: parity probes, synthetic frames).  Not declared in any
// public header; the product itself never calls these two.
struct HandleView {
    int device;
    const Workspace* ws;            // workspace of the last chunk (null before the first call)
    int last_chunk_frames;
    bool keep_pre;
    const int32_t* dict;            // host copy of the dictionary
    int dict_rows, dict_cols;
    bool fused;                     // the last chunk took the fused sweep: ws->half holds the threshold mask (1 bit per pixel), not the half-size image
    const uint8_t* frames;          // the frames the last chunk read (device memory) and their strides
    ptrdiff_t row_stride, frame_stride;
    hipStream_t stream;
    const uint8_t* gray;            // device gray frames of the last BGR call (null otherwise)
    ptrdiff_t gray_row_stride, gray_frame_stride;
};
void handle_view(const struct ::ctag_handle* h, HandleView* out);
// the part of ctag_gather_end behind the payload all-gather: segment table of a `world`-rank job + unpack kernels on the
// handle's gather stream, on a caller-supplied gathered buffer (world shards of `width` bytes); waits for completion
int gather_unpack_gathered(struct ::ctag_handle* h, const void* gathered_dev, int n_total, int world, uint64_t width, ctag_frame_result* out_dev);

}  // namespace ctag
