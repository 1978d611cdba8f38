// ctag_api.hip -- implementation of the C ABI in include/ctag.h: handle, workspace, batching, probes.
// Host orchestration only; the arithmetic lives in k_sweep.hip / k_quad.hip / k_feature.hip.
// There is deliberately no CPU fallback: if HIP is unavailable every entry point returns CTAG_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <string>
#include <vector>

#include "../../include/ctag.h"
#include "ctag_internal.h"

using namespace ctag;

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) {                                                                        \
            std::snprintf(h->last_error, sizeof(h->last_error), "%s: %s", #expr, hipGetErrorString(e__)); \
            return CTAG_ERR_HIP;                                                                        \
        }                                                                                               \
    } while (0)

struct ctag_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;          // side branch of the chain in few-frame calls (launch_quads), joined by events
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<int32_t> dict;
    ctag_params params{};            // the tunables this handle was created with (ctag_params_default unless ctag_create_ex said otherwise)
    KParams kp{};                    // ... as the kernels consume them
    uint8_t* d_thr_table = nullptr;  // adaptive-threshold bound table for params.dark_cap
    int32_t* d_dict = nullptr;
    uint32_t* d_dict_pos = nullptr;  // [dict_rows][64]: columns holding each symbol (k_markers' bit-parallel coverage); null for > 32 columns
    uint8_t* d_pick_table = nullptr;
    int dict_rows = 0, dict_cols = 0, feature_size = 0;
    // Two workspaces: `batch` holds chunks of frames with pools sized for what a frame of its size ordinarily needs (make_caps);
    // `big` holds ONE frame with pools no frame of that size can exhaust -- a frame that overflowed a batch pool (thousands of
    // blobs, fine texture: CTAG_FLAG_POOL_OVERFLOW -> CTAG_PENDING) is run again there (rerun_frame).  `big` is allocated on first use.
    struct WsSlot {
        Workspace ws;
        int rows = 0, cols = 0, tw = 0, cap = 0;
        double* n0_buf = nullptr;   // refine_n0, allocated when a chunk of more than kLatencyFrames frames first runs with corner_subpix
        size_t n0_frames = 0;
    };
    WsSlot batch, big;
    // CTAG_OPT_STREAMS = 2 (default): a chunk of a device-memory batch runs as two halves on two streams (`stream` + `stream2`, workspaces
    // `batch` + `batch2`), so that the tail of one half's kernels -- a few long boundary / Welsch blocks -- overlaps the other half's next
    // kernel.  stream2 forks from `stream` at the head of a call and joins it at the end: callers still order against `stream` alone.
    static constexpr int kMaxStreams = 4;
    WsSlot batchx[kMaxStreams - 1];                 // workspaces of the extra streams (stream k > 0 uses batchx[k - 1])
    hipStream_t streamx[kMaxStreams - 1] = {};      // (stream2 = streamx[0])
    hipEvent_t ev_joinx[kMaxStreams - 1] = {};
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork2 = nullptr;
    int streams = 2;
    const Workspace* last_ws = nullptr;  // whichever ran last (handle_view)
    bool last_fused = false;  // the last chunk took the fused sweep: its workspace holds the threshold mask, not the half-size image
    const uint8_t* last_frames = nullptr;  // ... and the frames it read (the test kit decimates them again for CTAG_DBG_HALF)
    ptrdiff_t last_row_stride = 0, last_frame_stride = 0;
    const ctag_frame_result* last_out = nullptr;  // ... and where its records went (ctag_get_counters)
    // frames of device-memory calls that wait for the any-frame pass (k_markers appends, finish_pending drains)
    PendingRec* d_pending = nullptr;
    int32_t* d_pending_count = nullptr;
    int pending_cap = 65536;
    bool pending_dirty = false;      // a device-memory call was enqueued since the list was last read
    uint64_t pending_gen = 0;        // ... and how many such calls there were
    uint8_t* d_big_frame = nullptr;  // private copy of a pending frame (host-memory calls upload it again; BGR frames are converted into d_big_gray)
    size_t d_big_frame_bytes = 0;
    uint8_t* d_big_gray = nullptr;
    size_t d_big_gray_bytes = 0;
    ctag_frame_result* d_big_result = nullptr;
    int reruns = 0;                  // frames completed through the any-frame workspace so far (ctag_get_counters)
    // ctag_submit_u8 / ctag_collect: frames in flight, one per call (a camera loop, main.cpp:44-61).  Slot k: a device slab for the frame (filled on
    // copy_stream), a device record and its pinned host copy; `up` = upload done, `done` = record downloaded
    struct AsyncSlot {
        uint8_t* d_frame = nullptr;
        size_t d_bytes = 0;
        ctag_frame_result* d_res = nullptr;
        ctag_frame_result* h_res = nullptr;
        hipEvent_t up = nullptr, done = nullptr;
        const uint8_t* host = nullptr;  // the caller's frame (valid until its ctag_collect: a pending frame is uploaded again from it)
        int rows = 0, cols = 0, ch = 1, tw = 0, subpix = 0, dist = 0;
        ptrdiff_t row_stride = 0;
    };
    static constexpr int kAsyncDepth = 2;
    AsyncSlot aslot[kAsyncDepth];
    int a_head = 0, a_count = 0;     // oldest slot in flight, slots in flight
    int max_chunk = 1024;
    int wave_points = 0;  // CTAG_OPT_WAVE_POINTS
    int fuse_mode = -1;   // CTAG_OPT_FUSED_SWEEP
    int bgr_direct = 1;   // CTAG_OPT_BGR_DIRECT
    bool timing = false;
    bool keep_pre = false;
    // timing (CTAG_OPT_TIMING): one set of CTAG_NUM_STAGES + 1 events per chunk of a public call, read back ONCE after the
    // call's last chunk has been enqueued (a read-back per chunk would serialise the upload / detect overlap being measured)
    std::vector<hipEvent_t> ev;
    int ev_sets_used = 0;
    float stage_ms[CTAG_NUM_STAGES] = {};
    // staging for host-memory entry points: two slabs of `host_sub` frames, filled on copy_stream while the
    // other slab is being processed on `stream`
    uint8_t* d_frames = nullptr;
    size_t d_frames_bytes = 0;
    ctag_frame_result* d_results = nullptr;
    ctag_frame_result* h_res1 = nullptr;       // pinned: the record of a one-frame host call, written by k_markers itself (no download)
    ctag_frame_result* h_res1_dev = nullptr;   // ... as the device addresses it
    size_t d_results_count = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copied[2] = {}, ev_done[2] = {};
    // BGR ingest (ctag_detect_*bgr8*): the gray frames of the chunk in flight, written by k_bgr2gray and read by K1 / K8
    uint8_t* d_gray = nullptr;
    size_t d_gray_bytes = 0;
    ptrdiff_t gray_row_stride = 0, gray_frame_stride = 0;
    bool last_was_bgr = false;
    int host_sub = 128;
    int last_chunk_frames = 0;
    char last_error[256] = {0};
    // hipGraph cache of the per-chunk kernel chain (CTAG_OPT_GRAPH): a chunk with the same pointers, sizes and parameters is
    // replayed with one hipGraphLaunch instead of ~25 launches / memsets -- what a stream of single frames (main.cpp:52-59)
    // and the small per-GPU shards of a multi-GPU job repeat thousands of times
    struct GraphEntry {
        const void* frames = nullptr;
        void* out = nullptr;
        const void* ws_base = nullptr;
        int n = 0, rows = 0, cols = 0, tw = 0, subpix = 0, dist = 0, keep_pre = 0, channels = 1;
        ptrdiff_t row_stride = 0, frame_stride = 0;
        const void* pend_src = nullptr;  // PendingCtx::src baked into k_markers' arguments (null: no list)
        hipGraphExec_t exec = nullptr;
        uint64_t last_use = 0;
    };
    std::vector<GraphEntry> graphs;
    uint64_t graph_clock = 0;
    int use_graph = 2;   // CTAG_OPT_GRAPH: 0 off, 1 every chunk, 2 (default) few-frame calls whose pointers / sizes / parameters repeat;
                         // back to 0 after a capture failed once (the direct path takes over for good)
    GraphEntry last_key;  // automatic mode: the previous few-frame chunk (a graph is only worth capturing for a repeat)
    // state of the pose back end (k_pose.hip), created on first use
    void* pose_state = nullptr;
    void (*pose_state_free)(void*) = nullptr;
    // state of the multi-GPU gather layer (ctag_gather.hip), created on first use
    void* gather_state = nullptr;
    void (*gather_state_free)(void*) = nullptr;
};

namespace ctag {
void** handle_pose_slot(ctag_handle* h, void (*free_fn)(void*)) {
    h->pose_state_free = free_fn;
    return &h->pose_state;
}
void** handle_gather_slot(ctag_handle* h, void (*free_fn)(void*)) {
    h->gather_state_free = free_fn;
    return &h->gather_state;
}
bool handle_timing(const ctag_handle* h) { return h->timing; }
int handle_device(const ctag_handle* h) { return h->device; }
void handle_view(const ctag_handle* h, HandleView* out) {
    out->device = h->device;
    out->ws = h->last_ws && h->last_ws->base ? h->last_ws : nullptr;
    out->last_chunk_frames = h->last_chunk_frames;
    out->keep_pre = h->keep_pre;
    out->dict = h->dict.data();
    out->dict_rows = h->dict_rows;
    out->dict_cols = h->dict_cols;
    out->fused = h->last_fused;
    out->frames = h->last_frames;
    out->row_stride = h->last_row_stride;
    out->frame_stride = h->last_frame_stride;
    out->stream = h->stream;
    out->gray = h->last_was_bgr ? h->d_gray : nullptr;
    out->gray_row_stride = h->gray_row_stride;
    out->gray_frame_stride = h->gray_frame_stride;
}
}  // namespace ctag

static const char* kStageNames[CTAG_NUM_STAGES] = {"decimate",  "threshold_ccl", "seam_merge", "resolve", "candidates", "quad_pack", "quad_edges", "quad_edges_big",
                                                    "line_sort", "welsch",        "quad_final", "features", "edge_refine", "markers"};

// ---------------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------------
static FrameGeom make_geom(int rows, int cols, int tw, double area_max_fraction = 0.01) {
    FrameGeom g{};
    g.rows = rows;
    g.cols = cols;
    g.hrows = rows / 2;
    g.hcols = cols / 2;
    g.hp = (g.hcols + 63) & ~63;
    g.lp = (g.hcols + 63) & ~63;
    g.tw = tw;
    g.trows = g.hrows / tw + (g.hrows % tw != 0 ? 1 : 0);
    g.tcols = g.hcols / tw + (g.hcols % tw != 0 ? 1 : 0);
    g.tiles_x = (g.hcols + kTileW - 1) / kTileW;
    g.tiles_y = (g.hrows + kTileH - 1) / kTileH;
    g.max_area = (int)std::round(area_max_fraction * g.hcols * g.hrows);  // corner_detector.cpp:88 (0.01 there)
    g.pool_cap = std::max(kPoolCapMin, 256 * g.tiles_x * g.tiles_y);
    return g;
}

// cv::resize INTER_CUBIC tap tables for one axis (resize.cpp: fx = (dx + 0.5) * scale - 0.5 in float, cubic weights with
// A = -0.75 in float, x2048 rounded to nearest-even into short) -- SURVEY App. A.1; CylinderTag.cpp:79.
static void build_resize_tables(int src, int dst, std::vector<int32_t>& ofs, std::vector<int16_t>& coef) {
    ofs.assign((size_t)std::max(dst, 0), 0);
    coef.assign((size_t)std::max(dst, 0) * 4, 0);
    if (dst <= 0) return;
    const double scale = (double)src / dst;
    for (int d = 0; d < dst; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        const int s0 = (int)std::floor(f);
        f -= s0;
        ofs[d] = s0;
        const float A = -0.75f;
        float c[4];
        c[0] = ((A * (f + 1) - 5 * A) * (f + 1) + 8 * A) * (f + 1) - 4 * A;
        c[1] = ((A + 2) * f - (A + 3)) * f * f + 1;
        c[2] = ((A + 2) * (1 - f) - (A + 3)) * (1 - f) * (1 - f) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
        for (int k = 0; k < 4; k++) {
            long r = std::lrint(c[k] * 2048.f);
            r = std::min(std::max(r, -32768L), 32767L);
            coef[(size_t)d * 4 + k] = (int16_t)r;
        }
    }
}

static void drop_graphs(ctag_handle* h);
static int finish_pending(ctag_handle* h);

// Pool sizes of a workspace.  Batch: what a frame of this size ordinarily needs, scaled with its area so that a 4K frame is not held
// to a 1080p frame's numbers (at 1080p: 2048 candidates, 262 144 cluster points -- a camera frame uses a few hundred / ~50 K).
// Big: bounds no frame can exceed --
//   candidates: components of >= area_min pixels: hw / area_min;
//   cluster points: a component of a pixels has w + h <= 2a, so its reservation min(2(w + h), wh) + 65 <= 4a + 65: 4 hw + 65 candidates;
//   component pool: a 320x30 label tile publishes every component only in the first pass (<= 128); the second publishes the ones
//   of >= area_min pixels (<= 9600 / area_min) or touching the tile border (<= 348 non-adjacent border pixels), at most kSlotCapBig.
struct Caps {
    int pool_cap, cand_cap;
    uint32_t cl_cap;
};
static Caps make_caps(const FrameGeom& g, const ctag_params& prm, bool big) {
    const long long hw = (long long)g.hrows * g.hcols;
    Caps c{};
    if (!big) {
        c.pool_cap = g.pool_cap;
        c.cand_cap = (int)std::max<long long>(kCandCapMin, ((hw / 256 + 63) / 64) * 64);
        c.cl_cap = (uint32_t)std::max<long long>(kClPoolMin, ((hw / 2 + 1023) / 1024) * 1024);
    } else {
        const int amin = std::max(prm.area_min, 1);
        const long long per_tile = std::min<long long>(kSlotCapBig, 352 + (kTileW * kTileH) / amin);
        c.pool_cap = (int)std::max<long long>(g.pool_cap, per_tile * g.tiles_x * g.tiles_y);
        const long long cand = ((hw / amin + 1 + 63) / 64) * 64;
        c.cand_cap = (int)std::max<long long>(kCandCapMin, cand);
        c.cl_cap = (uint32_t)std::max<long long>(kClPoolMin, 4 * hw + 65 * cand + 1024);
    }
    return c;
}

static int ensure_workspace(ctag_handle* h, ctag_handle::WsSlot& S, int rows, int cols, int tw, int frames, bool big, bool need_n0) {
    auto refresh = [&](Workspace& W) {
        const Caps c = make_caps(make_geom(rows, cols, tw, h->params.area_max_fraction), h->params, big);
        W.g = make_geom(rows, cols, tw, h->params.area_max_fraction);
        W.g.pool_cap = c.pool_cap;
        W.kp = h->kp;
        W.pick_table = h->d_pick_table;
        W.pick_table16 = reinterpret_cast<const uint16_t*>(h->d_pick_table + (size_t)kPickN * 200);
        W.aux_stream = h->aux_stream;
        W.wave_points = h->wave_points;
        W.fuse_mode = h->fuse_mode;
        W.ev_fork = h->ev_fork;
        W.ev_join = h->ev_join;
        W.big = big;
    };
    auto ensure_n0 = [&]() -> int {  // the searches -> sums hand-over of k_edge_refine<1> / <2>: 0.8 MB per frame, batches with corner_subpix only
        if (!need_n0 || S.n0_frames >= (size_t)S.cap) {
            S.ws.refine_n0 = S.n0_buf;
            return CTAG_OK;
        }
        for (hipStream_t x : h->streamx)
            if (x) HIP_TRY(hipStreamSynchronize(x));
        HIP_TRY(hipStreamSynchronize(h->stream));
        drop_graphs(h);
        if (S.n0_buf) HIP_TRY(hipFree(S.n0_buf));
        S.n0_buf = nullptr;
        S.n0_frames = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&S.n0_buf), (size_t)S.cap * CTAG_MAX_FEATURES * 2 * 4 * 128 * 8));
        S.n0_frames = (size_t)S.cap;
        S.ws.refine_n0 = S.n0_buf;
        return CTAG_OK;
    };
    if (S.ws.base && S.rows == rows && S.cols == cols && S.cap >= frames) {
        refresh(S.ws);
        S.tw = tw;
        return ensure_n0();
    }
    if (S.ws.base) {
        for (hipStream_t x : h->streamx)
            if (x) HIP_TRY(hipStreamSynchronize(x));
        HIP_TRY(hipStreamSynchronize(h->stream));
        drop_graphs(h);  // they hold pointers into the old workspace
        HIP_TRY(hipFree(S.ws.base));
        if (S.n0_buf) HIP_TRY(hipFree(S.n0_buf));
        S.n0_buf = nullptr;
        S.n0_frames = 0;
        if (h->last_ws == &S.ws) h->last_ws = nullptr;
        S.ws = Workspace{};
        S.cap = 0;
    }
    Workspace& W = S.ws;
    refresh(W);
    const Caps caps = make_caps(W.g, h->params, big);
    W.cand_cap = caps.cand_cap;
    W.line_cap = 4 * caps.cand_cap;
    W.cl_cap = caps.cl_cap;
    const FrameGeom& g = W.g;
    const size_t F = (size_t)frames;
    const size_t tiles = (size_t)g.tiles_x * g.tiles_y;
    const size_t CC = (size_t)W.cand_cap, LC = (size_t)W.line_cap;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off = (off + bytes + 255) & ~(size_t)255;
        return at;
    };
    const size_t o_half = take(F * g.hrows * g.hp + 256);
    const size_t o_labels = take(F * g.hrows * g.lp * 2 + 256);
    const size_t o_tbase = take(F * tiles * 4), o_tdirty = take(F * tiles * 4);
    const size_t o_ncomp = take(F * 4);
    const size_t o_flags = take(F * 4);
    const size_t o_ovfc = take(256);
    const size_t o_ovfl = take(F * tiles * 4);
    const size_t pool = F * (size_t)g.pool_cap * 4;
    const size_t o_parent = take(pool), o_root = take(pool), o_area = take(pool), o_xmin = take(pool), o_ymin = take(pool),
                 o_xmax = take(pool), o_ymax = take(pool), o_key = take(pool), o_ptile = take(pool), o_mhead = take(pool), o_mnext = take(pool);
    const size_t o_ncand = take(F * 4), o_nroots = take(F * 4);
    const size_t o_cand = take(F * CC * sizeof(Candidate));
    const size_t o_quads = take(F * CC * sizeof(QuadOut));
    const size_t o_lcount = take(F * 4), o_clused = take(F * 4), o_llong = take(F * 4);
    const size_t o_clpool = take(F * (size_t)W.cl_cap * 4 + 16);  // + one element: welsch_restart requests one point past a cluster's last
    const size_t o_ldesc = take(F * LC * sizeof(LineDesc));
    const size_t o_lsort = take(F * LC * 4);
    const size_t o_lfit = take(F * LC * 16);
    const size_t o_aux = take(F * CC * sizeof(CandAux));
    const size_t o_npk = take(F * 8), o_pk = take(F * CC * 4), o_pord = take(F * CC * 4);
    const size_t o_wrs = take(std::min<size_t>(F, kLatencyFrames) * kLatLines * 20 * 6 * 4);
    const size_t o_der = take(F * kQuadStride * 48);
    const size_t o_qidx = take(F * kQuadStride * 4);
    const size_t o_nq = take(F * 4), o_nf = take(F * 4), o_st = take(F * 4);
    const size_t o_f0 = take(F * CTAG_MAX_FEATURES * sizeof(FeatureDev));
    const size_t o_f1 = take(F * CTAG_MAX_FEATURES * sizeof(FeatureDev));
    const size_t o_f2 = take(F * CTAG_MAX_FEATURES * sizeof(FeatureDev));
    const size_t o_pre = take(F * sizeof(ctag_frame_result));
    const size_t o_flong = take(F * 4);
    const size_t o_rzx = take((size_t)g.hcols * 4), o_rza = take((size_t)g.hcols * 8), o_rzy = take((size_t)g.hrows * 4), o_rzb = take((size_t)g.hrows * 8);
    void* base = nullptr;
    HIP_TRY(hipMalloc(&base, off));
    char* b = static_cast<char*>(base);
    W.base = base;
    W.bytes = off;
    W.chunk_frames = frames;
    W.half = reinterpret_cast<uint8_t*>(b + o_half);
    W.labels = reinterpret_cast<uint16_t*>(b + o_labels);
    W.tile_base = reinterpret_cast<int32_t*>(b + o_tbase);
    W.tile_dirty = reinterpret_cast<int32_t*>(b + o_tdirty);
    // the label image's invariant: a tile whose tile_dirty is 0 holds zeros
    HIP_TRY(hipMemsetAsync(W.labels, 0, F * g.hrows * g.lp * 2 + 256, h->stream));
    HIP_TRY(hipMemsetAsync(W.tile_dirty, 0, F * tiles * 4, h->stream));
    HIP_TRY(hipMemsetAsync(W.tile_base, 0, F * tiles * 4, h->stream));  // the fused sweep leaves the entries of tiles without foreground alone: they must stay valid offsets
    W.frame_ncomp = reinterpret_cast<int32_t*>(b + o_ncomp);
    W.frame_flags = reinterpret_cast<uint32_t*>(b + o_flags);
    W.ovf_count = reinterpret_cast<int32_t*>(b + o_ovfc);
    W.ovf_list = reinterpret_cast<int32_t*>(b + o_ovfl);
    W.parent = reinterpret_cast<uint32_t*>(b + o_parent);
    W.root_of = reinterpret_cast<int32_t*>(b + o_root);
    W.area = reinterpret_cast<int32_t*>(b + o_area);
    W.xmin = reinterpret_cast<int32_t*>(b + o_xmin);
    W.ymin = reinterpret_cast<int32_t*>(b + o_ymin);
    W.xmax = reinterpret_cast<int32_t*>(b + o_xmax);
    W.ymax = reinterpret_cast<int32_t*>(b + o_ymax);
    W.key = reinterpret_cast<int32_t*>(b + o_key);
    W.pool_tile = reinterpret_cast<int32_t*>(b + o_ptile);
    W.member_head = reinterpret_cast<int32_t*>(b + o_mhead);
    W.member_next = reinterpret_cast<int32_t*>(b + o_mnext);
    W.ncand = reinterpret_cast<int32_t*>(b + o_ncand);
    W.nroots = reinterpret_cast<int32_t*>(b + o_nroots);
    W.cand = reinterpret_cast<Candidate*>(b + o_cand);
    W.quads = reinterpret_cast<QuadOut*>(b + o_quads);
    W.line_count = reinterpret_cast<int32_t*>(b + o_lcount);
    W.clp_used = reinterpret_cast<int32_t*>(b + o_clused);
    W.cl_pool = reinterpret_cast<uint32_t*>(b + o_clpool);
    W.line_desc = reinterpret_cast<LineDesc*>(b + o_ldesc);
    W.line_sorted = reinterpret_cast<int32_t*>(b + o_lsort);
    W.line_long = reinterpret_cast<int32_t*>(b + o_llong);
    W.line_fit = reinterpret_cast<float*>(b + o_lfit);
    W.cand_aux = reinterpret_cast<CandAux*>(b + o_aux);
    W.npacks = reinterpret_cast<int32_t*>(b + o_npk);
    W.packs = reinterpret_cast<uint32_t*>(b + o_pk);
    W.pack_order = reinterpret_cast<uint32_t*>(b + o_pord);
    W.welsch_rs = reinterpret_cast<float*>(b + o_wrs);
    W.quad_derived = b + o_der;
    W.quad_index = reinterpret_cast<int32_t*>(b + o_qidx);
    W.nquads = reinterpret_cast<int32_t*>(b + o_nq);
    W.nfeat = reinterpret_cast<int32_t*>(b + o_nf);
    W.status = reinterpret_cast<int32_t*>(b + o_st);
    W.feat0 = reinterpret_cast<FeatureDev*>(b + o_f0);
    W.feat1 = reinterpret_cast<FeatureDev*>(b + o_f1);
    W.feat2 = reinterpret_cast<FeatureDev*>(b + o_f2);
    W.premarkers = reinterpret_cast<ctag_frame_result*>(b + o_pre);
    W.refine_n0 = nullptr;
    W.frame_long = reinterpret_cast<int32_t*>(b + o_flong);
    W.rz_xofs = reinterpret_cast<int32_t*>(b + o_rzx);
    W.rz_alpha = reinterpret_cast<int16_t*>(b + o_rza);
    W.rz_yofs = reinterpret_cast<int32_t*>(b + o_rzy);
    W.rz_beta = reinterpret_cast<int16_t*>(b + o_rzb);
    {   // tap tables of the general decimation (used for odd sizes only; a few KB)
        std::vector<int32_t> xo, yo;
        std::vector<int16_t> al, be;
        build_resize_tables(cols, g.hcols, xo, al);
        build_resize_tables(rows, g.hrows, yo, be);
        HIP_TRY(hipMemcpy(W.rz_xofs, xo.data(), xo.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(W.rz_alpha, al.data(), al.size() * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(W.rz_yofs, yo.data(), yo.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(W.rz_beta, be.data(), be.size() * 2, hipMemcpyHostToDevice));
    }
    S.rows = rows;
    S.cols = cols;
    S.tw = tw;
    S.cap = frames;
    return ensure_n0();
}

static int check_args(ctag_handle* h, const void* frames, int n, int rows, int cols, ptrdiff_t row_stride, int adaptive_thresh, int subpix_dist) {
    if (!h || !frames || n < 0 || rows < 4 || cols < 4 || row_stride < cols || adaptive_thresh < 1 || subpix_dist < 0) return CTAG_ERR_ARG;
    if (adaptive_thresh > kMaxThreshWin) return CTAG_ERR_UNSUPPORTED;
    if (rows / 2 > 16000 || cols / 2 > 16000) return CTAG_ERR_UNSUPPORTED;
    if (row_stride >= (1 << 24) || (long long)rows * row_stride > 0xffffffffLL) return CTAG_ERR_UNSUPPORTED;  // 32-bit pixel offsets (k_edge_refine)
    return CTAG_OK;
}

// enqueue the whole pipeline for `n` device-resident frames (n <= workspace capacity); evs: CTAG_NUM_STAGES + 1 timing events or null
static int enqueue_chunk(ctag_handle* h, const Workspace& ws, const uint8_t* frames_dev, int n, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                         const DetectParams& p, ctag_frame_result* out_dev, hipEvent_t* evs, const PendingCtx& pend, hipStream_t s = nullptr) {
    if (!s) s = h->stream;
    h->last_ws = &ws;
    h->last_out = out_dev;
    h->last_frames = p.channels == 3 ? nullptr : frames_dev;  // (BGR frames handed to the chain as they are: no gray rows for the half-size probe to decimate)
    h->last_row_stride = row_stride;
    h->last_frame_stride = frame_stride;
    const bool zero_in_k1 = n <= kLatencyFrames;  // a few frames: one launch less at the head of the chain
    if (!zero_in_k1) HIP_TRY(launch_zero_counters(n, ws, s));
    int st = 0;
    auto mark = [&](int i) -> hipError_t { return evs ? hipEventRecord(evs[i], s) : hipSuccess; };
    HIP_TRY(mark(0));
    const bool fused = sweep_fused(frames_dev, frame_stride, row_stride, n, ws, p.channels == 3);  // threshold where the pixels are computed: 1 bit per pixel to K2, no `half`
    h->last_fused = fused;
    if (p.channels == 3 && !fused) return CTAG_ERR_UNSUPPORTED;  // (bgr_direct_ok has checked: BGR frames go through k_bgr2gray otherwise)
    HIP_TRY(launch_decimate(frames_dev, frame_stride, row_stride, n, ws, s, fused, zero_in_k1, p.channels));
    HIP_TRY(mark(++st));
    HIP_TRY(launch_threshold_ccl(n, ws, s, fused));
    HIP_TRY(mark(++st));
    HIP_TRY(launch_seam_merge(n, ws, s));
    HIP_TRY(mark(++st));
    HIP_TRY(launch_resolve(n, ws, s));
    HIP_TRY(mark(++st));
    HIP_TRY(launch_candidates(n, ws, s));
    HIP_TRY(mark(++st));
    HIP_TRY(launch_quads(n, ws, s, evs ? evs + st + 1 : nullptr, fused ? ws.half : nullptr));  // records one event after each of its first five kernels
    st += 5;
    HIP_TRY(mark(++st));
    HIP_TRY(launch_features(n, ws, p, s));
    HIP_TRY(mark(++st));
    HIP_TRY(launch_edge_refine(frames_dev, frame_stride, row_stride, n, ws, p, s));
    HIP_TRY(mark(++st));
    Workspace wtmp = ws;
    if (!h->keep_pre) wtmp.premarkers = nullptr;
    HIP_TRY(launch_markers(n, wtmp, p, out_dev, pend, s));
    HIP_TRY(mark(++st));
    return CTAG_OK;
}

// an exec owns the kernarg storage of its launches: nothing it enqueued may still be running when it is destroyed
static void graphs_idle(ctag_handle* h) {
    if (h->graphs.empty()) return;
    if (h->aux_stream) (void)hipStreamSynchronize(h->aux_stream);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
}
static void drop_graphs(ctag_handle* h) {
    graphs_idle(h);
    for (auto& g : h->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    h->graphs.clear();
}

// replay (or first capture) the chunk as a hipGraph; any failure disables the graph path for this handle and the caller
// falls back to direct launches -- results are the same either way
static bool run_chunk_graph(ctag_handle* h, const uint8_t* frames_dev, int n, ptrdiff_t row_stride, ptrdiff_t frame_stride, const DetectParams& p,
                            ctag_frame_result* out_dev, const PendingCtx& pend) {
    const Workspace& W = h->batch.ws;
    constexpr size_t kMaxGraphs = 8;
    ctag_handle::GraphEntry* hit = nullptr;
    for (auto& g : h->graphs)
        if (g.frames == frames_dev && g.out == out_dev && g.ws_base == W.base && g.n == n && g.rows == W.g.rows && g.cols == W.g.cols &&
            g.tw == p.adaptive_thresh && g.subpix == p.corner_subpix && g.dist == p.subpix_dist && g.keep_pre == (h->keep_pre ? 1 : 0) &&
            g.row_stride == row_stride && g.frame_stride == frame_stride && g.pend_src == (pend.list ? (const void*)pend.src : nullptr) && g.channels == p.channels)
            hit = &g;
    if (!hit) {
        if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed) != hipSuccess) return false;
        const int r = enqueue_chunk(h, W, frames_dev, n, row_stride, frame_stride, p, out_dev, nullptr, pend);
        hipGraph_t graph = nullptr;
        const hipError_t e = hipStreamEndCapture(h->stream, &graph);
        if (r != CTAG_OK || e != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            return false;
        }
        hipGraphExec_t exec = nullptr;
        const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess || !exec) {
            (void)hipGetLastError();
            return false;
        }
        if (h->graphs.size() >= kMaxGraphs) {  // evict the least recently used
            size_t lru = 0;
            for (size_t i = 1; i < h->graphs.size(); i++)
                if (h->graphs[i].last_use < h->graphs[lru].last_use) lru = i;
            graphs_idle(h);  // the victim may still be executing
            (void)hipGraphExecDestroy(h->graphs[lru].exec);
            h->graphs.erase(h->graphs.begin() + (ptrdiff_t)lru);
        }
        ctag_handle::GraphEntry g;
        g.frames = frames_dev;
        g.out = out_dev;
        g.ws_base = W.base;
        g.n = n;
        g.rows = W.g.rows;
        g.cols = W.g.cols;
        g.pend_src = pend.list ? (const void*)pend.src : nullptr;
        g.tw = p.adaptive_thresh;
        g.subpix = p.corner_subpix;
        g.dist = p.subpix_dist;
        g.keep_pre = h->keep_pre ? 1 : 0;
        g.row_stride = row_stride;
        g.frame_stride = frame_stride;
        g.channels = p.channels;
        g.exec = exec;
        h->graphs.push_back(g);
        hit = &h->graphs.back();
    }
    hit->last_use = ++h->graph_clock;
    h->last_ws = &W;
    h->last_out = out_dev;
    h->last_frames = p.channels == 3 ? nullptr : frames_dev;  // (BGR frames handed to the chain as they are: no gray rows for the half-size probe to decimate)
    h->last_row_stride = row_stride;
    h->last_frame_stride = frame_stride;
    h->last_fused = sweep_fused(frames_dev, frame_stride, row_stride, n, W, p.channels == 3);
    return hipGraphLaunch(hit->exec, h->stream) == hipSuccess;
}

static int run_chunk(ctag_handle* h, const uint8_t* frames_dev, int n, ptrdiff_t row_stride, ptrdiff_t frame_stride, const DetectParams& p,
                     ctag_frame_result* out_dev, const PendingCtx& pend) {
    const Workspace& W = h->batch.ws;
    h->last_chunk_frames = n;
    if (pend.list) {
        h->pending_dirty = true;
        h->pending_gen++;
    }
    hipEvent_t* evs = nullptr;
    if (h->timing) {
        const size_t need = (size_t)(h->ev_sets_used + 1) * (CTAG_NUM_STAGES + 1);
        while (h->ev.size() < need) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            h->ev.push_back(e);
        }
        evs = h->ev.data() + (size_t)h->ev_sets_used * (CTAG_NUM_STAGES + 1);
        h->ev_sets_used++;
    }
    static const bool stamps = getenv("CTAG_CCL_STAMPS") != nullptr || getenv("CTAG_QUAD_STAMPS") != nullptr || getenv("CTAG_FEAT_STAMPS") != nullptr;  // developer aids that synchronise inside the chain (no graph capture around them)
    bool graph = h->use_graph == 1;
    if (h->use_graph == 2 && n <= kLatencyFrames) {  // one frame per call in a loop (main.cpp:52-59): same staging buffers, same sizes every time
        ctag_handle::GraphEntry& k = h->last_key;
        const void* psrc = pend.list ? (const void*)pend.src : nullptr;
        graph = k.frames == frames_dev && k.out == out_dev && k.ws_base == W.base && k.n == n && k.rows == W.g.rows && k.cols == W.g.cols &&
                k.tw == p.adaptive_thresh && k.subpix == p.corner_subpix && k.dist == p.subpix_dist && k.keep_pre == (h->keep_pre ? 1 : 0) &&
                k.row_stride == row_stride && k.frame_stride == frame_stride && k.pend_src == psrc;
        k.frames = frames_dev, k.out = out_dev, k.ws_base = W.base, k.n = n, k.rows = W.g.rows, k.cols = W.g.cols, k.tw = p.adaptive_thresh,
        k.subpix = p.corner_subpix, k.dist = p.subpix_dist, k.keep_pre = h->keep_pre ? 1 : 0, k.row_stride = row_stride, k.frame_stride = frame_stride,
        k.pend_src = psrc;
    }
    if (!evs && graph && !stamps) {
        if (run_chunk_graph(h, frames_dev, n, row_stride, frame_stride, p, out_dev, pend)) return CTAG_OK;
        h->use_graph = 0;  // capture / instantiate / launch failed: direct launches from now on
        drop_graphs(h);
    }
    return enqueue_chunk(h, W, frames_dev, n, row_stride, frame_stride, p, out_dev, evs, pend);
}

// public entry points call begin_timings() before their first chunk and collect_timings() after their last
static void begin_timings(ctag_handle* h) {
    h->ev_sets_used = 0;
    for (int i = 0; i < CTAG_NUM_STAGES; i++) h->stage_ms[i] = 0.f;
}
static int collect_timings(ctag_handle* h) {
    const int sets = h->ev_sets_used;
    h->ev_sets_used = 0;
    if (!h->timing || sets == 0) return CTAG_OK;
    HIP_TRY(hipEventSynchronize(h->ev[(size_t)sets * (CTAG_NUM_STAGES + 1) - 1]));
    for (int k = 0; k < sets; k++) {
        const hipEvent_t* evs = h->ev.data() + (size_t)k * (CTAG_NUM_STAGES + 1);
        for (int i = 0; i < CTAG_NUM_STAGES; i++) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, evs[i], evs[i + 1]));
            h->stage_ms[i] += ms;
        }
    }
    return finish_pending(h);  // the call has waited anyway: frames that need the any-frame pass are completed now (outside the stage times)
}

// pend: where the frames of this call live in the CALLER's device memory (what a frame that needs the any-frame pass is read from
// again: `frames_dev` itself for gray calls, the BGR source for colour calls), or null for host-memory calls -- their frames
// pass through staging slabs that are reused, and they find CTAG_PENDING records in the results they download
static int detect_device_impl(ctag_handle* h, const uint8_t* frames_dev, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                              int adaptive_thresh, int corner_subpix, int subpix_dist, ctag_frame_result* out_dev, const PendingCtx* pend, int channels = 1) {
    const int rc = check_args(h, frames_dev, n, rows, cols, row_stride, adaptive_thresh, subpix_dist);
    if (rc != CTAG_OK) return rc;
    if (n == 0) return CTAG_OK;
    HIP_TRY(hipSetDevice(h->device));
    const int chunk = std::min(n, h->max_chunk);
    DetectParams p{adaptive_thresh, corner_subpix, subpix_dist, h->feature_size, h->dict_rows, h->dict_cols, h->d_dict, h->d_dict_pos};
    p.channels = channels;
    static const bool two_env = getenv("CTAG_STREAMS_MIN") != nullptr;
    static const int two_min = two_env ? std::max(2 * kLatencyFrames + 2, atoi(getenv("CTAG_STREAMS_MIN"))) : 256;
    // chunks of [448, 1024) frames stay on one stream: their halves (224-511 frames) leave the stage kernels' grids a partial last round of
    // blocks on each stream.  Measured (round 5, 1080p, two handles alternating, K frames/s split / whole): 256: 189 / 178, 384: 213 / 214,
    // 512: 222 / 237, 640: 229 / 234, 768: 232 / 234, 1024: 247 / 241, 2048: 255 / 250 (tools/step_overlap.py).
    // Scope of the rule: it looks at the call's chunk = min(n, max_chunk), measured at 1080p only -- frames of other sizes take it by frame COUNT all
    // the same (a 4K frame is four 1080p frames of blocks: its chunks of 112-255 frames, the equivalent window, are below two_min and stay whole anyway); the ragged tail of
    // a longer call (n = 1100 at max_chunk 1024: 512 + 512 + 76) is a piece of its own on the next stream; CTAG_STREAMS_MIN (developer aid) replaces the rule altogether.
    const bool whole = !two_env && chunk >= 448 && chunk < 1024;
    if (h->streams >= 2 && !h->timing && chunk >= two_min && !whole && h->stream2) {
        const int ns = std::min(h->streams, (int)ctag_handle::kMaxStreams);
        // two halves of every chunk side by side (see WsSlot batch2).  With CTAG_OPT_TIMING the chunk stays on one stream: the HIP events
        // around a kernel would otherwise time the other stream's kernels as well.
        // (measured, round 4: pieces of 1024 / 512 frames instead of half a chunk 226 / 203 K frames/s against 231 K; the second stream half a
        // piece out of phase -- one stream in its memory-bound sweep while the other computes -- 216-227 K: the kernels' tails want large pieces)
        const int piece = (chunk + ns - 1) / ns;
        ctag_handle::WsSlot* slot[ctag_handle::kMaxStreams] = {&h->batch, &h->batchx[0], &h->batchx[1], &h->batchx[2]};
        for (int k = 0; k < ns; k++) {
            const int wr = ensure_workspace(h, *slot[k], rows, cols, adaptive_thresh, std::max(piece, slot[k]->rows == rows && slot[k]->cols == cols ? slot[k]->cap : 0),
                                            false, corner_subpix != 0);
            if (wr != CTAG_OK) return wr;
        }
        HIP_TRY(hipEventRecord(h->ev_fork2, h->stream));
        for (int k = 1; k < ns; k++) HIP_TRY(hipStreamWaitEvent(h->streamx[k - 1], h->ev_fork2, 0));
        int k = 0;
        for (int f0 = 0; f0 < n; f0 += piece, k = (k + 1) % ns) {
            const int m = std::min(piece, n - f0);
            PendingCtx pc{};
            if (pend) {
                pc = *pend;
                pc.src = pend->src + (ptrdiff_t)f0 * pend->frame_stride;
                h->pending_dirty = true;
                h->pending_gen++;
            }
            h->last_chunk_frames = m;
            const int r = enqueue_chunk(h, slot[k]->ws, frames_dev + (ptrdiff_t)f0 * frame_stride, m, row_stride, frame_stride, p, out_dev + f0, nullptr, pc,
                                        k ? h->streamx[k - 1] : h->stream);
            if (r != CTAG_OK) return r;
        }
        for (int j = 1; j < ns; j++) {
            HIP_TRY(hipEventRecord(h->ev_joinx[j - 1], h->streamx[j - 1]));
            HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_joinx[j - 1], 0));
        }
        return CTAG_OK;
    }
    const int wr = ensure_workspace(h, h->batch, rows, cols, adaptive_thresh, std::max(chunk, h->batch.rows == rows && h->batch.cols == cols ? h->batch.cap : 0),
                                    false, corner_subpix != 0 && chunk > kLatencyFrames);
    if (wr != CTAG_OK) return wr;
    for (int f0 = 0; f0 < n; f0 += chunk) {
        const int m = std::min(chunk, n - f0);
        PendingCtx pc{};
        if (pend) {
            pc = *pend;
            pc.src = pend->src + (ptrdiff_t)f0 * pend->frame_stride;
        }
        const int r = run_chunk(h, frames_dev + (ptrdiff_t)f0 * frame_stride, m, row_stride, frame_stride, p, out_dev + f0, pc);
        if (r != CTAG_OK) return r;
    }
    return CTAG_OK;
}

// ---------------------------------------------------------------------------------------------------
// BGR ingest: cvtColor(BGR2GRAY) of the reference's stream loop (main.cpp:36,52-54) on the device
// ---------------------------------------------------------------------------------------------------
// OpenCV's 8-bit BGR2GRAY is fixed point: (B*1868 + G*9617 + R*4899 + 8192) >> 14 (RGB2Gray<uchar>: B2Y, G2Y, R2Y at yuv_shift 14,
// [OCV-recall of color_rgb.simd.hpp]; the same formula as the host BMP reader, csrc/ctag_io.h).  A lane converts four pixels:
// three aligned words in, one word out -- consecutive lanes read consecutive 12-byte groups, so a wave's loads are contiguous.
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t* __restrict__ bgr, ptrdiff_t frame_stride, ptrdiff_t row_stride, uint8_t* __restrict__ gray,
                                                  ptrdiff_t gframe_stride, ptrdiff_t grow_stride, int rows, int cols, int aligned) {
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y, f = blockIdx.z;
    if (x4 >= cols) return;
    const uint8_t* src = bgr + (ptrdiff_t)f * frame_stride + (ptrdiff_t)y * row_stride + (ptrdiff_t)x4 * 3;
    uint8_t* dst = gray + (ptrdiff_t)f * gframe_stride + (ptrdiff_t)y * grow_stride + x4;
    if (aligned && x4 + 3 < cols) {
        const uint32_t w0 = reinterpret_cast<const uint32_t*>(src)[0], w1 = reinterpret_cast<const uint32_t*>(src)[1], w2 = reinterpret_cast<const uint32_t*>(src)[2];
        // bytes: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
        const uint32_t g0 = gray_of(w0 & 0xffu, (w0 >> 8) & 0xffu, (w0 >> 16) & 0xffu);
        const uint32_t g1 = gray_of(w0 >> 24, w1 & 0xffu, (w1 >> 8) & 0xffu);
        const uint32_t g2 = gray_of((w1 >> 16) & 0xffu, w1 >> 24, w2 & 0xffu);
        const uint32_t g3 = gray_of((w2 >> 8) & 0xffu, (w2 >> 16) & 0xffu, w2 >> 24);
        *reinterpret_cast<uint32_t*>(dst) = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    } else {
        for (int k = 0; k < 4 && x4 + k < cols; k++) dst[k] = (uint8_t)gray_of(src[3 * k], src[3 * k + 1], src[3 * k + 2]);
    }
}

// gray slab for `frames` frames of rows x cols (rows padded to 16 bytes); grows only
static int ensure_gray(ctag_handle* h, int frames, int rows, int cols) {
    const ptrdiff_t pitch = ((ptrdiff_t)cols + 15) & ~(ptrdiff_t)15;
    const size_t need = (size_t)pitch * rows * (size_t)frames;
    if (h->d_gray_bytes < need || h->gray_row_stride != pitch || h->gray_frame_stride != pitch * rows) {
        if (h->d_gray_bytes < need) {
            if (h->copy_stream) HIP_TRY(hipStreamSynchronize(h->copy_stream));
            if (h->aux_stream) HIP_TRY(hipStreamSynchronize(h->aux_stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            drop_graphs(h);  // they hold the old slab's address
            if (h->d_gray) HIP_TRY(hipFree(h->d_gray));
            h->d_gray = nullptr;
            h->d_gray_bytes = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_gray), need));
            h->d_gray_bytes = need;
        }
        h->gray_row_stride = pitch;
        h->gray_frame_stride = pitch * rows;
    }
    return CTAG_OK;
}

static int enqueue_bgr2gray(ctag_handle* h, const uint8_t* bgr_dev, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride, uint8_t* gray = nullptr) {
    const int aligned = ((reinterpret_cast<uintptr_t>(bgr_dev) | (uintptr_t)row_stride | (uintptr_t)frame_stride) & 3u) == 0 ? 1 : 0;
    const ptrdiff_t pitch = ((ptrdiff_t)cols + 15) & ~(ptrdiff_t)15;  // == gray_row_stride of the chunk slab
    hipLaunchKernelGGL(k_bgr2gray, dim3((cols + 1023) / 1024, rows, n), dim3(256), 0, h->stream, bgr_dev, frame_stride, row_stride, gray ? gray : h->d_gray,
                       gray ? pitch * rows : h->gray_frame_stride, gray ? pitch : h->gray_row_stride, rows, cols, aligned);
    HIP_TRY(hipGetLastError());
    return CTAG_OK;
}

// ---------------------------------------------------------------------------------------------------
// the any-frame pass: frames that exceeded a pool of the batch workspace
// ---------------------------------------------------------------------------------------------------
static int grow(ctag_handle* h, uint8_t** buf, size_t* have, size_t need) {
    if (*have >= need) return CTAG_OK;
    if (*buf) HIP_TRY(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(buf), need));
    *have = need;
    return CTAG_OK;
}

// One frame in device memory (gray, or BGR for ch == 3) through the workspace whose pools hold any frame of its size; the record
// goes to out_dev.  Enqueued on the handle's stream; the caller waits.  The reference has none of the batch workspace's limits
// (corner_detector.cpp:81-107 keeps every component of [area_min, 1 %], :171-405 walks them all), so neither does a result.
static int rerun_frame(ctag_handle* h, const uint8_t* src_dev, int ch, int rows, int cols, ptrdiff_t row_stride, int tw, int subpix, int dist,
                       ctag_frame_result* out_dev) {
    const int wr = ensure_workspace(h, h->big, rows, cols, tw, 1, true, false);
    if (wr != CTAG_OK) return wr;
    const uint8_t* gray = src_dev;
    ptrdiff_t gstride = row_stride;
    if (ch == 3) {
        const ptrdiff_t pitch = ((ptrdiff_t)cols + 15) & ~(ptrdiff_t)15;
        const int gr = grow(h, &h->d_big_gray, &h->d_big_gray_bytes, (size_t)pitch * rows);
        if (gr != CTAG_OK) return gr;
        const int r = enqueue_bgr2gray(h, src_dev, 1, rows, cols, row_stride, 0, h->d_big_gray);
        if (r != CTAG_OK) return r;
        gray = h->d_big_gray;
        gstride = pitch;
    }
    DetectParams p{tw, subpix, dist, h->feature_size, h->dict_rows, h->dict_cols, h->d_dict, h->d_dict_pos};
    h->last_chunk_frames = 1;
    h->reruns++;
    return enqueue_chunk(h, h->big.ws, gray, 1, gstride, (ptrdiff_t)gstride * rows, p, out_dev, nullptr, PendingCtx{});
}

// drain the list device-memory calls left (see handle_finish_pending, ctag_internal.h)
static int finish_pending(ctag_handle* h) {
    if (!h->pending_dirty) return CTAG_OK;
    HIP_TRY(hipSetDevice(h->device));
    if (h->aux_stream) HIP_TRY(hipStreamSynchronize(h->aux_stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    int count = 0;
    HIP_TRY(hipMemcpy(&count, h->d_pending_count, 4, hipMemcpyDeviceToHost));
    h->pending_dirty = false;
    if (count <= 0) return CTAG_OK;
    const int take = std::min(count, h->pending_cap);
    std::vector<PendingRec> recs((size_t)take);
    HIP_TRY(hipMemcpy(recs.data(), h->d_pending, sizeof(PendingRec) * (size_t)take, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(h->d_pending_count, 0, 4));
    // "the last chunk" (ctag_get_counters, the parity probes) stays the caller's chunk, not the one-frame passes below (round-4 ADVICE)
    const Workspace* const keep_ws = h->last_ws;
    const ctag_frame_result* const keep_out = h->last_out;
    const int keep_frames = h->last_chunk_frames;
    int rc = CTAG_OK;
    for (const PendingRec& r : recs) {
        rc = rerun_frame(h, r.src, r.ch, r.rows, r.cols, (ptrdiff_t)r.row_stride, r.tw, r.subpix, r.dist, r.out);
        if (rc != CTAG_OK) break;
    }
    h->last_ws = keep_ws;
    h->last_out = keep_out;
    h->last_chunk_frames = keep_frames;
    if (rc != CTAG_OK) return rc;
    if (h->aux_stream) HIP_TRY(hipStreamSynchronize(h->aux_stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (count > h->pending_cap) {  // more than the list holds between two synchronisation points: k_markers gave the surplus the terminal status CTAG_ERR_LIMIT
        std::snprintf(h->last_error, sizeof(h->last_error), "%d frames waited for the any-frame pass, the list holds %d: the surplus carries CTAG_ERR_LIMIT; synchronise more often",
                      count, h->pending_cap);
        return CTAG_ERR_LIMIT;
    }
    return CTAG_OK;
}
namespace ctag {
int handle_finish_pending(ctag_handle* h) { return h ? finish_pending(h) : CTAG_ERR_ARG; }
bool handle_pending_state(ctag_handle* h, const int32_t** count_dev, uint64_t* gen) {
    *count_dev = h->d_pending_count;
    *gen = h->pending_gen;
    return h->pending_dirty && h->d_pending_count != nullptr;
}
void handle_pending_clean(ctag_handle* h, uint64_t gen) {
    if (h->pending_gen == gen) h->pending_dirty = false;
}
}  // namespace ctag

// host-memory calls: the records are on the host already; a CTAG_PENDING one is completed from the caller's own frame
static int rerun_host_frames(ctag_handle* h, const uint8_t* frames, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride, int ch, int tw,
                             int subpix, int dist, ctag_frame_result* out) {
    for (int i = 0; i < n; i++) {
        if (out[i].status != CTAG_PENDING) continue;
        const ptrdiff_t rowbytes = (ptrdiff_t)cols * ch, dstride = (rowbytes + 15) & ~(ptrdiff_t)15;
        int r = grow(h, &h->d_big_frame, &h->d_big_frame_bytes, (size_t)dstride * rows);
        if (r != CTAG_OK) return r;
        if (!h->d_big_result) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_big_result), sizeof(ctag_frame_result)));
        HIP_TRY(hipMemcpy2DAsync(h->d_big_frame, dstride, frames + (ptrdiff_t)i * frame_stride, row_stride, (size_t)rowbytes, rows, hipMemcpyHostToDevice, h->stream));
        r = rerun_frame(h, h->d_big_frame, ch, rows, cols, dstride, tw, subpix, dist, h->d_big_result);
        if (r != CTAG_OK) return r;
        HIP_TRY(hipMemcpyAsync(out + i, h->d_big_result, sizeof(ctag_frame_result), hipMemcpyDeviceToHost, h->stream));
        if (h->aux_stream) HIP_TRY(hipStreamSynchronize(h->aux_stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return CTAG_OK;
}

// n BGR frames in device memory -> gray (chunk by chunk, on the handle's stream) -> the detection chain
static int detect_bgr_device_impl(ctag_handle* h, const uint8_t* bgr_dev, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                                  int adaptive_thresh, int corner_subpix, int subpix_dist, ctag_frame_result* out_dev) {
    if (row_stride < (ptrdiff_t)cols * 3) return CTAG_ERR_ARG;
    const int rc = check_args(h, bgr_dev, n, rows, cols, ((ptrdiff_t)cols + 15) & ~(ptrdiff_t)15, adaptive_thresh, subpix_dist);
    if (rc != CTAG_OK) return rc;
    if (n == 0) return CTAG_OK;
    HIP_TRY(hipSetDevice(h->device));
    // The direct form (round 5): frames of a size the fused sweep takes (half size a multiple of 320 x 5; adaptiveThresh 5), rows and frames 16-byte aligned -- the
    // decimation kernel loads the BGR bytes themselves and converts as it consumes them, edgeRefine converts the boxes it stages: no gray image is
    // written or read back (8.3 of the 24 MB a 1080p frame moved through the gray slab).  CTAG_OPT_BGR_DIRECT 0 turns it off.
    // Calls of a few frames (round 6): the fused sweep's K1 is one block per frame band group -- 0.19 ms for a single 1080p frame against < 0.02 ms
    // of the short-band kernels -- so they convert first and take the gray chain, as gray calls of that size do (sweep_fused).
    if (h->bgr_direct && sweep_fused_size(rows, cols, adaptive_thresh, h->fuse_mode) && sweep_fused_batch(rows, cols, n, h->fuse_mode) && row_stride < (1 << 24) && (long long)rows * row_stride <= 0xffffffffLL &&
        (((uintptr_t)bgr_dev | (uintptr_t)frame_stride | (uintptr_t)row_stride) & 15) == 0) {
        const PendingCtx pc{h->d_pending, h->d_pending_count, h->pending_cap, bgr_dev, (int64_t)frame_stride, (int64_t)row_stride,
                            rows, cols, 3, adaptive_thresh, corner_subpix, subpix_dist};
        const int r = detect_device_impl(h, bgr_dev, n, rows, cols, row_stride, frame_stride, adaptive_thresh, corner_subpix, subpix_dist, out_dev, &pc, 3);
        h->last_was_bgr = false;  // no gray image to probe
        return r;
    }
    const int chunk = std::min(n, h->max_chunk);
    const int gr = ensure_gray(h, chunk, rows, cols);
    if (gr != CTAG_OK) return gr;
    for (int f0 = 0; f0 < n; f0 += chunk) {
        const int m = std::min(chunk, n - f0);
        int r = enqueue_bgr2gray(h, bgr_dev + (ptrdiff_t)f0 * frame_stride, m, rows, cols, row_stride, frame_stride);
        if (r != CTAG_OK) return r;
        const PendingCtx pc{h->d_pending, h->d_pending_count, h->pending_cap, bgr_dev + (ptrdiff_t)f0 * frame_stride, (int64_t)frame_stride, (int64_t)row_stride,
                            rows, cols, 3, adaptive_thresh, corner_subpix, subpix_dist};
        r = detect_device_impl(h, h->d_gray, m, rows, cols, h->gray_row_stride, h->gray_frame_stride, adaptive_thresh, corner_subpix, subpix_dist, out_dev + f0, &pc);
        if (r != CTAG_OK) return r;
    }
    h->last_was_bgr = true;
    return CTAG_OK;
}

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
extern "C" {

int ctag_version(void) { return 110; }  // 110: ctag_params carries struct_size; CTAG_PENDING; test scaffolding left the product ABI (was 100)

const char* ctag_strerror(int status) {
    switch (status) {
        case CTAG_OK: return "ok";
        case CTAG_NO_CORNER: return "No corner detected!";
        case CTAG_NO_FEATURE: return "No feature detected!";
        case CTAG_ERR_ARG: return "invalid argument";
        case CTAG_ERR_HIP: return "HIP runtime error (no usable gfx950 device?)";
        case CTAG_ERR_LIMIT: return "frame exceeds a fixed-array limit of the reference";
        case CTAG_PENDING: return "frame waits for the any-frame pass (ctag_sync completes it)";
        case CTAG_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown status";
    }
}
const char* ctag_stage_name(int stage) { return (stage >= 0 && stage < CTAG_NUM_STAGES) ? kStageNames[stage] : ""; }

void ctag_params_default(ctag_params* p) {  // header/corner_detector.h:90,110,122,135-137,144; corner_detector.cpp:71,88,285
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->struct_size = (uint32_t)sizeof(*p);
    p->threshold_line = 1.8f;
    p->threshold_expand = 1.2f;
    p->threshold_RAC = 0.3f;
    p->threshold_angle = 5.f;
    p->threshold_vertical = 0.5f;
    const float id[4] = {1.47f, 1.54f, 1.61f, 1.68f}, lo[4] = {0.1f, 0.035f, 0.035f, 0.035f}, hi[4] = {0.035f, 0.035f, 0.035f, 0.1f};
    for (int j = 0; j < 4; j++) {
        p->ID_cr_correspond[j] = id[j];
        p->cr_covariance_left[j] = lo[j];
        p->cr_covariance_right[j] = hi[j];
    }
    p->dark_cap = 0.3f;
    p->area_min = 30;
    p->area_max_fraction = 0.01;
    p->collinear_cost = 1.05;
}

// ctag_params -> KParams (everything but the device table).  The collinearity test compares cost = (float)sqrt((double)c2), c2 the
// squared integer norm of P0 + P2 - 2 P1, with a double threshold (corner_detector.cpp:285-288 `cost > 1.05`, :337 `cost < 1.05`):
// as bounds on c2, found by stepping from the threshold's square.
static bool derive_kparams(const ctag_params& p, KParams* k) {
    auto ok = [](double v) { return v > 0 && v < 1e30; };
    if (!ok(p.threshold_line) || !ok(p.threshold_expand) || !ok(p.threshold_RAC) || !ok(p.threshold_angle) || !ok(p.threshold_vertical) || p.area_min < 1 ||
        !(p.area_max_fraction > 0 && p.area_max_fraction <= 1) || !(p.collinear_cost > 0 && p.collinear_cost < 1e4))
        return false;
    for (int j = 0; j < 4; j++)
        if (!ok(p.ID_cr_correspond[j]) || !(p.cr_covariance_left[j] >= 0) || !(p.cr_covariance_right[j] >= 0)) return false;
    k->thr_line = p.threshold_line;
    k->thr_expand = p.threshold_expand;
    k->expand_eps = 3.0e-6f;
    k->rac = p.threshold_RAC;
    k->angle = p.threshold_angle;
    k->vertical = p.threshold_vertical;
    for (int j = 0; j < 4; j++) {
        k->cr_id[j] = p.ID_cr_correspond[j];
        k->cr_lo[j] = p.cr_covariance_left[j];
        k->cr_hi[j] = p.cr_covariance_right[j];
    }
    k->dark_cap = p.dark_cap;
    k->area_min = p.area_min;
    k->area_max_fraction = p.area_max_fraction;
    auto cost = [](long long c2) { return (double)(float)std::sqrt((double)c2); };
    long long c = (long long)(p.collinear_cost * p.collinear_cost);
    while (c > 0 && cost(c) > p.collinear_cost) c--;       // largest c2 with cost <= threshold ...
    while (!(cost(c + 1) > p.collinear_cost)) c++;
    k->c2_far = (int)(c + 1);                                // ... so cost > threshold from c + 1 on
    long long d = c;
    while (d >= 0 && !(cost(d) < p.collinear_cost)) d--;     // largest c2 with cost < threshold (-1: none)
    k->c2_near = (int)d;
    return true;
}

int ctag_create(const int32_t* state, int dict_rows, int dict_cols, int feature_size, int device_id, ctag_handle** out) {
    return ctag_create_ex(state, dict_rows, dict_cols, feature_size, device_id, nullptr, out);
}

int ctag_create_ex(const int32_t* state, int dict_rows, int dict_cols, int feature_size, int device_id, const ctag_params* params, ctag_handle** out) {
    if (!out) return CTAG_ERR_ARG;
    *out = nullptr;
    ctag_params prm;
    ctag_params_default(&prm);
    if (params) {
        if (params->struct_size != sizeof(ctag_params)) return CTAG_ERR_ARG;  // built against another version of include/ctag_types.h
        prm = *params;
    }
    KParams kp{};
    if (!derive_kparams(prm, &kp)) return CTAG_ERR_ARG;
    std::vector<uint8_t> thr(256 * 256);
    if (!build_threshold_table(prm.dark_cap, thr.data(), &kp.thr_dim, &kp.tcap)) return CTAG_ERR_ARG;
    kp.tcap4 = (uint32_t)kp.tcap * 0x01010101u;
    if (!state || dict_rows < 1 || dict_cols < 1 || (long)dict_rows * dict_cols > kMaxDictCells) return CTAG_ERR_ARG;
    for (long i = 0; i < (long)dict_rows * dict_cols; i++)
        if (!(state[i] >= 0 && state[i] <= 63)) return CTAG_ERR_ARG;  // check_dictionary, CylinderTag.cpp:56-65
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return CTAG_ERR_HIP;
    ctag_handle* h = new (std::nothrow) ctag_handle();
    if (!h) return CTAG_ERR_HIP;
    h->device = device_id;
    h->params = prm;
    h->dict.assign(state, state + (size_t)dict_rows * dict_cols);
    h->dict_rows = dict_rows;
    h->dict_cols = dict_cols;
    h->feature_size = feature_size;
    bool ok = hipSetDevice(device_id) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < ctag_handle::kMaxStreams - 1; k++) {
        ok = ok && hipStreamCreateWithFlags(&h->streamx[k], hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&h->ev_joinx[k], hipEventDisableTiming) == hipSuccess;
    }
    h->stream2 = h->streamx[0];
    ok = ok && hipEventCreateWithFlags(&h->ev_fork2, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&h->d_dict), h->dict.size() * 4) == hipSuccess;
    ok = ok && hipMemcpy(h->d_dict, h->dict.data(), h->dict.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && dict_cols <= 32) {
        std::vector<uint32_t> pos((size_t)dict_rows * 64, 0u);
        for (int i = 0; i < dict_rows; i++)
            for (int c = 0; c < dict_cols; c++) pos[(size_t)i * 64 + (state[(size_t)i * dict_cols + c] & 63)] |= 1u << c;
        ok = hipMalloc(reinterpret_cast<void**>(&h->d_dict_pos), pos.size() * 4) == hipSuccess &&
             hipMemcpy(h->d_dict_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok) {
        // fitLine2D's initial samples: cv::RNG is seeded the same way for every fit, so the ten points a restart starts from depend on the edge's
        // point count alone -- tabulated for every count below kPickN2 (bytes below kPickN, halfwords above, one allocation)
        const size_t nb = (size_t)kPickN * 200, nh = (size_t)(kPickN2 - kPickN) * 200;
        std::vector<uint8_t> tab(nb);
        std::vector<uint16_t> tab16(nh);
        build_pick_table(tab.data(), tab16.data());
        ok = hipMalloc(reinterpret_cast<void**>(&h->d_pick_table), nb + nh * 2) == hipSuccess &&
             hipMemcpy(h->d_pick_table, tab.data(), nb, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(h->d_pick_table + nb, tab16.data(), nh * 2, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok) {
        const size_t tb = (size_t)kp.thr_dim * kp.thr_dim;
        ok = hipMalloc(reinterpret_cast<void**>(&h->d_thr_table), tb) == hipSuccess && hipMemcpy(h->d_thr_table, thr.data(), tb, hipMemcpyHostToDevice) == hipSuccess;
        kp.thr_table = h->d_thr_table;
        h->kp = kp;
    }
    ok = ok && hipMalloc(reinterpret_cast<void**>(&h->d_pending), sizeof(PendingRec) * (size_t)h->pending_cap) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&h->d_pending_count), 256) == hipSuccess && hipMemset(h->d_pending_count, 0, 256) == hipSuccess;
    if (!ok) {
        ctag_destroy(h);
        return CTAG_ERR_HIP;
    }
    *out = h;
    return CTAG_OK;
}

void ctag_destroy(ctag_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    if (h->aux_stream) (void)hipStreamSynchronize(h->aux_stream);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    drop_graphs(h);
    for (int k = 0; k < ctag_handle::kMaxStreams - 1; k++) {
        if (h->streamx[k]) (void)hipStreamSynchronize(h->streamx[k]);
        if (h->ev_joinx[k]) (void)hipEventDestroy(h->ev_joinx[k]);
    }
    if (h->ev_fork2) (void)hipEventDestroy(h->ev_fork2);
    for (ctag_handle::WsSlot* S : {&h->batch, &h->batchx[0], &h->batchx[1], &h->batchx[2], &h->big}) {
        if (S->ws.base) (void)hipFree(S->ws.base);
        if (S->n0_buf) (void)hipFree(S->n0_buf);
    }
    if (h->d_pending) (void)hipFree(h->d_pending);
    if (h->d_pending_count) (void)hipFree(h->d_pending_count);
    if (h->d_big_frame) (void)hipFree(h->d_big_frame);
    if (h->d_big_gray) (void)hipFree(h->d_big_gray);
    if (h->d_big_result) (void)hipFree(h->d_big_result);
    for (auto& A : h->aslot) {
        if (A.d_frame) (void)hipFree(A.d_frame);
        if (A.h_res) (void)hipHostFree(A.h_res);
        if (A.up) (void)hipEventDestroy(A.up);
        if (A.done) (void)hipEventDestroy(A.done);
    }
    if (h->d_dict) (void)hipFree(h->d_dict);
    if (h->d_thr_table) (void)hipFree(h->d_thr_table);
    if (h->d_dict_pos) (void)hipFree(h->d_dict_pos);
    if (h->d_pick_table) (void)hipFree(h->d_pick_table);
    if (h->d_frames) (void)hipFree(h->d_frames);
    if (h->d_results) (void)hipFree(h->d_results);
    if (h->h_res1) (void)hipHostFree(h->h_res1);
    if (h->d_gray) (void)hipFree(h->d_gray);
    if (h->pose_state && h->pose_state_free) h->pose_state_free(h->pose_state);
    if (h->gather_state && h->gather_state_free) h->gather_state_free(h->gather_state);
    for (auto& e : h->ev)
        if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; i++) {
        if (h->ev_copied[i]) (void)hipEventDestroy(h->ev_copied[i]);
        if (h->ev_done[i]) (void)hipEventDestroy(h->ev_done[i]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
    for (int k = 0; k < ctag_handle::kMaxStreams - 1; k++)
        if (h->streamx[k]) (void)hipStreamDestroy(h->streamx[k]);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int ctag_load_marker_file(const char* path, int32_t** state, int* dict_rows, int* dict_cols, int* feature_size) {
    if (!path || !state || !dict_rows || !dict_cols || !feature_size) return CTAG_ERR_ARG;
    std::ifstream in(path);
    if (!in.is_open()) return CTAG_ERR_ARG;
    int n = 0, c = 0, fs = 0;
    in >> n >> c >> fs;
    if (!in || n < 1 || c < 1 || (long)n * c > (1L << 24)) return CTAG_ERR_ARG;
    int32_t* s = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (size_t)n * c));
    if (!s) return CTAG_ERR_ARG;
    for (long i = 0; i < (long)n * c; i++) {
        int v = 0;
        in >> v;  // like the reference, a short file leaves the remaining entries at their last parsed value (0)
        s[i] = v;
        if (!(v >= 0 && v <= 63)) {
            std::free(s);
            return CTAG_ERR_ARG;
        }
    }
    *state = s;
    *dict_rows = n;
    *dict_cols = c;
    *feature_size = fs;
    return CTAG_OK;
}
void ctag_free(void* p) { std::free(p); }

int ctag_set_option(ctag_handle* h, int option, int64_t value) {
    if (!h) return CTAG_ERR_ARG;
    switch (option) {
        case CTAG_OPT_MAX_CHUNK:
            if (value < 1 || value > (1 << 20)) return CTAG_ERR_ARG;
            h->max_chunk = (int)value;
            return CTAG_OK;
        case CTAG_OPT_TIMING: h->timing = value != 0; return CTAG_OK;
        case CTAG_OPT_KEEP_PREMARKERS: h->keep_pre = value != 0; return CTAG_OK;
        case CTAG_OPT_GRAPH:
            if (value < 0 || value > 2) return CTAG_ERR_ARG;
            h->use_graph = (int)value;
            if (!h->use_graph) drop_graphs(h);
            return CTAG_OK;
        case CTAG_OPT_WAVE_POINTS:
            if (value < 0 || value > 0x7fffffff) return CTAG_ERR_ARG;
            h->wave_points = (int)value;
            drop_graphs(h);  // a captured chain holds the old value
            return CTAG_OK;
        case CTAG_OPT_EXPAND_EXACT:
            h->kp.expand_eps = value ? INFINITY : 3.0e-6f;
            drop_graphs(h);
            return CTAG_OK;
        case CTAG_OPT_STREAMS:
            if (value < 1 || value > ctag_handle::kMaxStreams) return CTAG_ERR_ARG;
            h->streams = (int)value;
            return CTAG_OK;
        case CTAG_OPT_FUSED_SWEEP:
            if (value < 0 || value > 2) return CTAG_ERR_ARG;
            h->fuse_mode = (int)value;
            drop_graphs(h);
            return CTAG_OK;
        case CTAG_OPT_BGR_DIRECT:
            h->bgr_direct = value != 0;
            drop_graphs(h);
            return CTAG_OK;
        case CTAG_OPT_HOST_SUBCHUNK:
            if (value < 1 || value > (1 << 20)) return CTAG_ERR_ARG;
            h->host_sub = (int)value;
            return CTAG_OK;
        default: return CTAG_ERR_ARG;
    }
}

int ctag_get_timings(ctag_handle* h, float* ms, int capacity) {
    if (!h || !ms) return 0;
    const int n = std::min(capacity, (int)CTAG_NUM_STAGES);
    for (int i = 0; i < n; i++) ms[i] = h->stage_ms[i];
    return n;
}

int ctag_get_counters(ctag_handle* h, ctag_counters* out) {
    if (!h || !out) return CTAG_ERR_ARG;
    std::memset(out, 0, sizeof(*out));
    out->reruns = h->reruns;
    if (!h->last_ws || !h->last_ws->base || h->last_chunk_frames <= 0) return CTAG_OK;
    HIP_TRY(hipSetDevice(h->device));
    const int fr = finish_pending(h);
    if (fr != CTAG_OK) return fr;
    out->reruns = h->reruns;
    long long* d = reinterpret_cast<long long*>(h->d_pending_count + 16);  // 80 bytes of the 256-byte counter block
    HIP_TRY(launch_counters(h->last_chunk_frames, *h->last_ws, h->last_out, d, h->stream));
    long long v[10];
    HIP_TRY(hipMemcpyAsync(v, d, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    out->frames = h->last_chunk_frames;
    for (int k = 0; k < 5; k++) {
        out->sum[k] = v[k];
        out->max[k] = (int32_t)v[5 + k];
    }
    return CTAG_OK;
}

void* ctag_stream(ctag_handle* h) { return h ? (void*)h->stream : nullptr; }

int ctag_sync(ctag_handle* h) {
    if (!h) return CTAG_ERR_ARG;
    h->last_error[0] = 0;
    const int r = finish_pending(h);  // frames that wait for the any-frame pass (CTAG_PENDING) are completed here
    if (r != CTAG_OK) return r;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CTAG_OK;
}

int ctag_detect_batch_device(ctag_handle* h, const uint8_t* frames_dev, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                             int adaptive_thresh, int corner_subpix, int subpix_dist, ctag_frame_result* out_dev) {
    if (!h || !out_dev) return CTAG_ERR_ARG;
    begin_timings(h);
    h->last_was_bgr = false;
    const PendingCtx pc{h->d_pending, h->d_pending_count, h->pending_cap, frames_dev, (int64_t)frame_stride, (int64_t)row_stride,
                        rows, cols, 1, adaptive_thresh, corner_subpix, subpix_dist};
    const int r = detect_device_impl(h, frames_dev, n, rows, cols, row_stride, frame_stride, adaptive_thresh, corner_subpix, subpix_dist, out_dev, &pc);
    if (r != CTAG_OK) {
        h->ev_sets_used = 0;
        return r;
    }
    return collect_timings(h);  // with CTAG_OPT_TIMING the call waits for its last chunk (and completes CTAG_PENDING frames); otherwise it returns at once
}

// both streams idle: every exit of the host-batch path that leaves copies or kernels in flight goes through here, and so does
// every reallocation of the staging slabs (an upload still running on copy_stream must not lose its destination)
static int quiesce(ctag_handle* h) {
    if (h->copy_stream) HIP_TRY(hipStreamSynchronize(h->copy_stream));
    for (hipStream_t x : h->streamx)
        if (x) HIP_TRY(hipStreamSynchronize(x));
    if (h->aux_stream) HIP_TRY(hipStreamSynchronize(h->aux_stream));  // side branch of few-frame calls
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CTAG_OK;
}

static int detect_batch_u8_impl(ctag_handle* h, const uint8_t* frames, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                                int adaptive_thresh, int corner_subpix, int subpix_dist, ctag_frame_result* out, int ch = 1) {
    // Frames stream through two device slabs: while slab k is processed on the compute stream, slab k+1 is filled over
    // PCIe on the copy stream (main.cpp:29,36,52-54 feed one frame at a time; this is the batched equivalent).  The
    // copies only overlap when `frames` is pinned (ctag_host_alloc / hipHostRegister); pageable memory still works.
    const int sub = std::min(n, std::min(h->max_chunk, h->host_sub));
    // ch = 3: BGR frames (3 bytes per pixel) are uploaded as they are and converted on the device (k_bgr2gray) into the gray slab
    const ptrdiff_t rowbytes = (ptrdiff_t)cols * ch;
    const ptrdiff_t dstride = (rowbytes + 15) & ~(ptrdiff_t)15;  // packed, 16-byte aligned rows on the device
    if (ch == 3) {
        const int gr = ensure_gray(h, sub, rows, cols);
        if (gr != CTAG_OK) return gr;
    }
    const size_t dframe = (size_t)dstride * rows;
    if (h->d_frames_bytes < dframe * sub * 2) {
        if (quiesce(h) != CTAG_OK) return CTAG_ERR_HIP;
        if (h->d_frames) HIP_TRY(hipFree(h->d_frames));
        h->d_frames = nullptr;
        h->d_frames_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_frames), dframe * sub * 2));
        h->d_frames_bytes = dframe * sub * 2;
    }
    if (h->d_results_count < (size_t)sub * 2) {
        if (quiesce(h) != CTAG_OK) return CTAG_ERR_HIP;
        if (h->d_results) HIP_TRY(hipFree(h->d_results));
        h->d_results = nullptr;
        h->d_results_count = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_results), sizeof(ctag_frame_result) * sub * 2));
        h->d_results_count = (size_t)sub * 2;
    }
    if (!h->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            HIP_TRY(hipEventCreateWithFlags(&h->ev_copied[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_done[i], hipEventDisableTiming));
        }
    }
    const bool packed = row_stride == rowbytes && dstride == rowbytes && frame_stride == (ptrdiff_t)rows * rowbytes;
    const size_t slab_frames = h->d_frames_bytes / dframe / 2;
    const int nsub = (n + sub - 1) / sub;
    auto upload = [&](int k) -> int {  // enqueue the upload of sub-chunk k on the copy stream
        const int f0 = k * sub, m = std::min(sub, n - f0), slot = k & 1;
        uint8_t* slab = h->d_frames + dframe * slab_frames * slot;
        if (k >= 2) HIP_TRY(hipStreamWaitEvent(h->copy_stream, h->ev_done[slot], 0));  // the slab's previous sub-chunk is finished
        if (packed) {
            HIP_TRY(hipMemcpyAsync(slab, frames + (ptrdiff_t)f0 * frame_stride, dframe * m, hipMemcpyHostToDevice, h->copy_stream));
        } else {
            for (int i = 0; i < m; i++)
                HIP_TRY(hipMemcpy2DAsync(slab + dframe * i, dstride, frames + (ptrdiff_t)(f0 + i) * frame_stride, row_stride, (size_t)rowbytes, rows,
                                         hipMemcpyHostToDevice, h->copy_stream));
        }
        HIP_TRY(hipEventRecord(h->ev_copied[slot], h->copy_stream));
        return CTAG_OK;
    };
    if (n == 1 && ch == 1) {
        // One frame (main.cpp:52-59: the reference's loop): everything on the compute stream -- a second stream and its event cost the call more than
        // they could hide -- and the record written by the last kernel straight into pinned host memory instead of fetched by a copy behind it
        if (!h->h_res1) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_res1), sizeof(ctag_frame_result), hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->h_res1_dev), h->h_res1, 0));
        }
        uint8_t* slab = h->d_frames;
        if (packed) HIP_TRY(hipMemcpyAsync(slab, frames, dframe, hipMemcpyHostToDevice, h->stream));
        else HIP_TRY(hipMemcpy2DAsync(slab, dstride, frames, row_stride, (size_t)rowbytes, rows, hipMemcpyHostToDevice, h->stream));
        const int r1 = detect_device_impl(h, slab, 1, rows, cols, dstride, (ptrdiff_t)dframe, adaptive_thresh, corner_subpix, subpix_dist, h->h_res1_dev, nullptr);
        if (r1 != CTAG_OK) return r1;
        HIP_TRY(hipStreamSynchronize(h->stream));
        std::memcpy(out, h->h_res1, sizeof(ctag_frame_result));
        return rerun_host_frames(h, frames, n, rows, cols, row_stride, frame_stride, ch, adaptive_thresh, corner_subpix, subpix_dist, out);
    }
    int r = upload(0);
    if (r != CTAG_OK) return r;
    for (int k = 0; k < nsub; k++) {
        const int f0 = k * sub, m = std::min(sub, n - f0), slot = k & 1;
        uint8_t* slab = h->d_frames + dframe * slab_frames * slot;
        ctag_frame_result* res = h->d_results + (h->d_results_count / 2) * slot;
        HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_copied[slot], 0));
        if (ch == 3) {
            r = enqueue_bgr2gray(h, slab, m, rows, cols, dstride, (ptrdiff_t)dframe);
            if (r != CTAG_OK) return r;
            r = detect_device_impl(h, h->d_gray, m, rows, cols, h->gray_row_stride, h->gray_frame_stride, adaptive_thresh, corner_subpix, subpix_dist, res, nullptr);
        } else {
            r = detect_device_impl(h, slab, m, rows, cols, dstride, (ptrdiff_t)dframe, adaptive_thresh, corner_subpix, subpix_dist, res, nullptr);
        }
        if (r != CTAG_OK) return r;
        HIP_TRY(hipEventRecord(h->ev_done[slot], h->stream));
        // the next upload is enqueued before the result download: a download into pageable memory blocks the host
        if (k + 1 < nsub && (r = upload(k + 1)) != CTAG_OK) return r;
        HIP_TRY(hipMemcpyAsync(out + f0, res, sizeof(ctag_frame_result) * m, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    // a frame that exceeded a pool of the batch workspace came back CTAG_PENDING: complete it through the any-frame workspace
    return rerun_host_frames(h, frames, n, rows, cols, row_stride, frame_stride, ch, adaptive_thresh, corner_subpix, subpix_dist, out);
}

int ctag_detect_batch_u8(ctag_handle* h, const uint8_t* frames, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                         int adaptive_thresh, int corner_subpix, int subpix_dist, ctag_frame_result* out) {
    if (!h || !out) return CTAG_ERR_ARG;
    const int rc = check_args(h, frames, n, rows, cols, row_stride, adaptive_thresh, subpix_dist);
    if (rc != CTAG_OK) return rc;
    if (n == 0) return CTAG_OK;
    HIP_TRY(hipSetDevice(h->device));
    begin_timings(h);
    h->last_was_bgr = false;
    const int r = detect_batch_u8_impl(h, frames, n, rows, cols, row_stride, frame_stride, adaptive_thresh, corner_subpix, subpix_dist, out);
    if (r != CTAG_OK) {  // uploads / kernels may still be in flight: nothing may outlive this call (the caller frees `frames`)
        h->ev_sets_used = 0;
        char keep[sizeof(h->last_error)];
        std::memcpy(keep, h->last_error, sizeof(keep));
        (void)quiesce(h);
        std::memcpy(h->last_error, keep, sizeof(keep));
        return r;
    }
    return collect_timings(h);
}

// ---- one frame per call, not waited for (include/ctag.h) ----------------------------------------------------------------------------
int ctag_submit_u8(ctag_handle* h, const uint8_t* gray, int rows, int cols, ptrdiff_t row_stride, int adaptive_thresh, int corner_subpix, int subpix_dist) {
    if (!h) return CTAG_ERR_ARG;
    h->last_error[0] = 0;
    const int rc = check_args(h, gray, 1, rows, cols, row_stride, adaptive_thresh, subpix_dist);
    if (rc != CTAG_OK) return rc;
    if (h->a_count >= ctag_handle::kAsyncDepth) return CTAG_ERR_ARG;  // collect first
    HIP_TRY(hipSetDevice(h->device));
    ctag_handle::AsyncSlot& A = h->aslot[(h->a_head + h->a_count) % ctag_handle::kAsyncDepth];
    const ptrdiff_t dstride = ((ptrdiff_t)cols + 15) & ~(ptrdiff_t)15;
    const size_t need = (size_t)dstride * rows;
    if (!h->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            HIP_TRY(hipEventCreateWithFlags(&h->ev_copied[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_done[i], hipEventDisableTiming));
        }
    }
    if (!A.up) {
        HIP_TRY(hipEventCreateWithFlags(&A.up, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&A.done, hipEventDisableTiming));
        // the record is written by k_markers itself into pinned host memory (d_res = that memory as the device addresses it): no download behind the chain
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&A.h_res), sizeof(ctag_frame_result), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&A.d_res), A.h_res, 0));
    }
    if (A.d_bytes < need) {  // the slot is free: nothing reads its slab
        drop_graphs(h);      // ... but a captured chain may hold its address
        if (A.d_frame) HIP_TRY(hipFree(A.d_frame));
        A.d_frame = nullptr;
        A.d_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&A.d_frame), need));
        A.d_bytes = need;
    }
    A.host = gray, A.rows = rows, A.cols = cols, A.ch = 1, A.row_stride = row_stride, A.tw = adaptive_thresh, A.subpix = corner_subpix, A.dist = subpix_dist;
    // the upload runs on the copy stream while the previous frame's kernels run on the compute stream (the slab's previous reader finished before its
    // ctag_collect returned); the detection waits for it; the record comes back into pinned memory behind the detection
    HIP_TRY(hipMemcpy2DAsync(A.d_frame, dstride, gray, row_stride, (size_t)cols, rows, hipMemcpyHostToDevice, h->copy_stream));
    HIP_TRY(hipEventRecord(A.up, h->copy_stream));
    HIP_TRY(hipStreamWaitEvent(h->stream, A.up, 0));
    h->last_was_bgr = false;
    begin_timings(h);
    const int r = detect_device_impl(h, A.d_frame, 1, rows, cols, dstride, (ptrdiff_t)need, adaptive_thresh, corner_subpix, subpix_dist, A.d_res, nullptr);
    h->ev_sets_used = 0;
    if (r != CTAG_OK) {
        (void)quiesce(h);
        return r;
    }
    HIP_TRY(hipEventRecord(A.done, h->stream));
    h->a_count++;
    return CTAG_OK;
}

int ctag_collect(ctag_handle* h, ctag_frame_result* out) {
    if (!h || !out) return CTAG_ERR_ARG;
    h->last_error[0] = 0;
    if (h->a_count <= 0) return CTAG_ERR_ARG;  // nothing in flight
    HIP_TRY(hipSetDevice(h->device));
    ctag_handle::AsyncSlot& A = h->aslot[h->a_head];
    HIP_TRY(hipEventSynchronize(A.done));
    h->a_head = (h->a_head + 1) % ctag_handle::kAsyncDepth;
    h->a_count--;
    *out = *A.h_res;
    if (out->status == CTAG_PENDING) {  // needs the any-frame workspace: from the caller's frame, like every host-memory call
        const int r = rerun_host_frames(h, A.host, 1, A.rows, A.cols, A.row_stride, 0, A.ch, A.tw, A.subpix, A.dist, out);
        if (r != CTAG_OK) return r;
    }
    return out->status;
}

void* ctag_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void ctag_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int ctag_detect_u8(ctag_handle* h, const uint8_t* gray, int rows, int cols, ptrdiff_t row_stride, int adaptive_thresh, int corner_subpix,
                   int subpix_dist, ctag_frame_result* out) {
    if (!h || !out) return CTAG_ERR_ARG;
    const int r = ctag_detect_batch_u8(h, gray, 1, rows, cols, row_stride, (ptrdiff_t)row_stride * rows, adaptive_thresh, corner_subpix, subpix_dist, out);
    if (r != CTAG_OK) return r;
    return out->status;
}

int ctag_detect_batch_bgr8_device(ctag_handle* h, const uint8_t* bgr_dev, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                                  int adaptive_thresh, int corner_subpix, int subpix_dist, ctag_frame_result* out_dev) {
    if (!h || !out_dev) return CTAG_ERR_ARG;
    begin_timings(h);
    const int r = detect_bgr_device_impl(h, bgr_dev, n, rows, cols, row_stride, frame_stride, adaptive_thresh, corner_subpix, subpix_dist, out_dev);
    if (r != CTAG_OK) {
        h->ev_sets_used = 0;
        return r;
    }
    return collect_timings(h);
}

int ctag_detect_batch_bgr8(ctag_handle* h, const uint8_t* bgr, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride, int adaptive_thresh,
                           int corner_subpix, int subpix_dist, ctag_frame_result* out) {
    if (!h || !out || row_stride < (ptrdiff_t)cols * 3) return CTAG_ERR_ARG;
    const int rc = check_args(h, bgr, n, rows, cols, ((ptrdiff_t)cols + 15) & ~(ptrdiff_t)15, adaptive_thresh, subpix_dist);
    if (rc != CTAG_OK) return rc;
    if (n == 0) return CTAG_OK;
    HIP_TRY(hipSetDevice(h->device));
    begin_timings(h);
    const int r = detect_batch_u8_impl(h, bgr, n, rows, cols, row_stride, frame_stride, adaptive_thresh, corner_subpix, subpix_dist, out, 3);
    if (r != CTAG_OK) {
        h->ev_sets_used = 0;
        char keep[sizeof(h->last_error)];
        std::memcpy(keep, h->last_error, sizeof(keep));
        (void)quiesce(h);
        std::memcpy(h->last_error, keep, sizeof(keep));
        return r;
    }
    h->last_was_bgr = true;
    return collect_timings(h);
}

int ctag_detect_bgr8(ctag_handle* h, const uint8_t* bgr, int rows, int cols, ptrdiff_t row_stride, int adaptive_thresh, int corner_subpix, int subpix_dist,
                     ctag_frame_result* out) {
    if (!h || !out) return CTAG_ERR_ARG;
    const int r = ctag_detect_batch_bgr8(h, bgr, 1, rows, cols, row_stride, (ptrdiff_t)row_stride * rows, adaptive_thresh, corner_subpix, subpix_dist, out);
    if (r != CTAG_OK) return r;
    return out->status;
}

}  // extern "C"
