/* ctag_types.h -- flat POD result records shared by the C ABI (include/ctag.h) and the test oracle.
 *
 * They flatten the reference's `struct MarkerInfo` (/root/reference/header/corner_detector.h:16-22):
 *   markerID, featurePos[], feature_ID[], feature_ID_left[], feature_ID_right[], cornerLists[n][8],
 *   feature_center[n], edge_length[n], cr_left[n], cr_right[n]
 * into fixed-size per-frame storage so that a frame's result is one memcpy / one RCCL gather element.
 * Limits come from the reference's own fixed arrays: father[100] (corner_detector.h:143) bounds the
 * features of a frame, code[20] (corner_detector.h:152) bounds the code positions of a marker.
 */
#ifndef CTAG_TYPES_H
#define CTAG_TYPES_H
#include <stdint.h>

#define CTAG_MAX_FEATURES 100 /* per frame; reference father[100] */
#define CTAG_MAX_MARKERS 100  /* per frame; every marker owns >= 1 feature */
#define CTAG_MAX_QUADS 1000   /* per frame; reference isVisited[1000], corner_detector.h:124 */
#define CTAG_MAX_CODE_POS 20  /* reference code[20] */

/* status of one frame */
#define CTAG_OK 0          /* markers assigned (possibly zero markers) */
#define CTAG_NO_CORNER 1   /* reference prints "No corner detected!" and leaves the output untouched */
#define CTAG_NO_FEATURE 2  /* reference prints "No feature detected!" and leaves the output untouched */
#define CTAG_ERR_ARG (-1)
#define CTAG_ERR_HIP (-2)
#define CTAG_ERR_LIMIT (-3) /* frame exceeded a fixed-array limit the reference would overflow (UB there) */
#define CTAG_ERR_UNSUPPORTED (-4)
#define CTAG_PENDING (-5) /* DEVICE-memory calls only, and only until the handle's next synchronisation point (ctag_sync, or any call
                             that waits): the frame needs larger pools than the batch workspace holds -- thousands of blobs, fine
                             texture -- and is completed there through a workspace that holds any frame.  Host-memory calls
                             never return it. */

/* flags (bit set) */
#define CTAG_FLAG_QUAD_OVERFLOW 1u     /* > CTAG_MAX_QUADS quads */
#define CTAG_FLAG_FEATURE_OVERFLOW 2u  /* > CTAG_MAX_FEATURES features */
#define CTAG_FLAG_CODE_OVERFLOW 4u     /* a marker's code position reached CTAG_MAX_CODE_POS (marker dropped) */
#define CTAG_FLAG_POOL_OVERFLOW 8u     /* a pool of the batch workspace was exhausted: the frame is CTAG_PENDING, then completed (flag cleared);
                                          with CTAG_ERR_LIMIT: a component larger than any 8K frame can hold (frames beyond 7680x4320 only) */
#define CTAG_FLAG_ERASE_CLAMPED 16u    /* reference UB: vector::erase past end, defined as no-op */

typedef struct ctag_feature_rec {
    int32_t pos;          /* k-th entry of MarkerInfo::featurePos for the first n_pos records, else -1 */
    int32_t id;           /* feature_ID */
    int32_t id_left;      /* feature_ID_left */
    int32_t id_right;     /* feature_ID_right */
    float corners[16];    /* cornerLists[j][0..7] as x,y pairs, full-resolution pixel coordinates */
    float center[2];      /* feature_center[j] */
    float edge_length;    /* edge_length[j] */
    float cr_left;        /* cr_left[j] */
    float cr_right;       /* cr_right[j] */
} ctag_feature_rec;       /* 100 bytes */

typedef struct ctag_marker_rec {
    int32_t marker_id;     /* MarkerInfo::markerID (dictionary row), -1 before decoding */
    int32_t first_feature; /* index into ctag_frame_result::features */
    int32_t n_features;    /* cornerLists.size() */
    int32_t n_pos;         /* featurePos.size() (<= n_features) */
} ctag_marker_rec;

typedef struct ctag_frame_result {
    int32_t status;
    int32_t n_markers;
    int32_t n_features;
    uint32_t flags;
    ctag_marker_rec markers[CTAG_MAX_MARKERS];
    ctag_feature_rec features[CTAG_MAX_FEATURES];
} ctag_frame_result; /* 16 + 1600 + 10000 = 11616 bytes */

/* What the frames of the last chunk held, stage by stage (ctag_get_counters, include/ctag.h).  Index: 0 connected components the label
 * sweep published (every component of >= area_min pixels is among them), 1 candidates (area within [area_min, 1 %],
 * corner_detector.cpp:88), 2 quads (edgeExtraction's output, :171-405), 3 features (:465-559), 4 markers in the record. */
#define CTAG_NUM_COUNTERS 5
typedef struct ctag_counters {
    int64_t frames;                  /* frames of the chunk */
    int64_t sum[CTAG_NUM_COUNTERS];  /* over those frames */
    int32_t max[CTAG_NUM_COUNTERS];
    int32_t reruns;                  /* frames this handle has completed through the any-frame workspace so far (CTAG_PENDING) */
} ctag_counters;

/* The detector's tunables (ctag_create_ex, include/ctag.h); the reference's values in the comments */
typedef struct ctag_params {
    uint32_t struct_size;          /* sizeof(ctag_params) of the header the caller was compiled with: ctag_params_default fills it in,
                                      ctag_create_ex refuses a struct of another size (a later library version may append fields) */
    float threshold_line;          /* 1.8   split a boundary span while a point lies farther than this from its chord   h:90,  cpp:320-329 */
    float threshold_expand;        /* 1.2   expand_line accepts a point within this distance of the refitted line       h:90,  cpp:144,156 */
    float threshold_RAC;           /* 0.3   quadJudgment: |shoelace area - pixel count| / pixel count below this         h:110, cpp:454-463 */
    float threshold_angle;         /* 5     degrees: featureRecovery (x1, x10) and markerOrganization (x2, x1)           h:122, cpp:488-548,985 */
    float threshold_vertical;      /* 0.5   markerOrganization: |cos(centre vector, long edge)| below this               h:144, cpp:985 */
    float ID_cr_correspond[4];     /* 1.47 1.54 1.61 1.68   cross ratio of code 0..3                                     h:135, cpp:1165-1189 */
    float cr_covariance_left[4];   /* 0.1 0.035 0.035 0.035 interval below ID_cr_correspond[j]                           h:136 */
    float cr_covariance_right[4];  /* 0.035 0.035 0.035 0.1 interval above                                               h:137 */
    float dark_cap;                /* 0.3   a pixel is foreground iff p < min(dark_cap, (max + min) / 2), p in [0, 1]    cpp:71 */
    int32_t area_min;              /* 30    components of fewer pixels are dropped                                       cpp:88 */
    double area_max_fraction;      /* 0.01  ... and those above round(fraction * cols * rows) of the half-size image     cpp:88 */
    double collinear_cost;         /* 1.05  |P0 + P2 - 2 P1| of a boundary triplet counts as collinear below this        cpp:285-288,337 */
} ctag_params;

#endif
