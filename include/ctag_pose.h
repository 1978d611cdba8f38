/* ctag_pose.h -- C ABI of the pose back end on the MI355X: what the reference does in
 *     CylinderTag::loadModel    /root/reference/CylinderTag.cpp:161-190   (.model text file)
 *     CylinderTag::loadCamera   /root/reference/CylinderTag.cpp:192-196   (OpenCV FileStorage YAML: cameraMatrix, distCoeffs)
 *     CylinderTag::estimatePose /root/reference/CylinderTag.cpp:198-209
 *     PoseEstimator::PnPSolver  /root/reference/pose_estimation.cpp:50-98  (correspondences, solvePnP EPNP)
 *     PoseEstimator::PoseBA     /root/reference/pose_estimation.cpp:100-127 (undistortPoints + Ceres LM on the
 *                                                                            reprojection error of :5-48)
 * for every decoded marker of a batch of frames, on the device, straight from the detection result records
 * (ctag_frame_result in HBM) -- one wavefront per marker.
 *
 * Third-party arithmetic restated here (un-vendored in the reference, absent from this image):
 * OpenCV 4.5.3 solvePnP(SOLVEPNP_EPNP) / undistortPoints / Rodrigues and Ceres 2.0 trust-region
 * Levenberg-Marquardt (Release.props:6,11).  Floating point: the parity bar against the CPU oracle is stated in
 * tests/test_pose_gpu.py.
 *
 * Plain pointers and sizes only.  Every function returns a CTAG_* status and never throws.
 */
#ifndef CTAG_POSE_H
#define CTAG_POSE_H
#include <stddef.h>
#include <stdint.h>

#include "ctag.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CTAG_POSE_MAX_POINTS 160 /* CTAG_MAX_CODE_POS features x 8 corners */

/* status of one marker's pose */
#define CTAG_POSE_OK 0
#define CTAG_POSE_NO_MODEL 1     /* reference: pose.markerID = -1 (pose_estimation.cpp:63-66), erased by estimatePose */
#define CTAG_POSE_TOO_FEW 2      /* < 4 correspondences: cv::solvePnP throws in the reference */
#define CTAG_POSE_BAD_POS 3      /* featurePos outside the model (out-of-bounds read in the reference) */
#define CTAG_POSE_DEGENERATE 4   /* non-finite EPnP result */

/* CamInfo (header/pose_estimation.h:12-14): cameraMatrix 3x3 and distCoeffs, both 'dt: f' in cameraParams.yml */
typedef struct ctag_camera {
    float K[9];      /* row-major cameraMatrix */
    float dist[14];  /* k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 (tau_x tau_y must be 0) */
    int32_t n_dist;  /* 0, 4, 5, 8, 12 or 14 */
} ctag_camera;

/* vector<ModelInfo> (header/pose_estimation.h:16-20) flattened */
typedef struct ctag_model_view {
    int32_t n_models;         /* model_num   (CylinderTag.cpp:169) */
    int32_t model_size;       /* model_size: features per marker; corners has model_size*8 points per model */
    const int32_t* marker_id; /* [n_models]  ModelInfo::MarkerID */
    const float* base;        /* [n_models*3] */
    const float* axis;        /* [n_models*3] */
    const float* corners;     /* [n_models*model_size*8*3] */
} ctag_model_view;

typedef struct ctag_pose_rec {
    int32_t status;      /* CTAG_POSE_* */
    int32_t model_index; /* PoseInfo::markerID: the INDEX into the model list (pose_estimation.cpp:59,69), -1 if none */
    int32_t frame;       /* frame index in the batch */
    int32_t marker;      /* marker index inside the frame's ctag_frame_result */
    int32_t n_points;    /* correspondences used */
    int32_t iterations;  /* LM iterations taken (successful + unsuccessful) */
    double rvec[3];      /* PoseInfo::rvec after PoseBA */
    double tvec[3];      /* PoseInfo::tvec after PoseBA */
    double rvec0[3];     /* solvePnP(EPNP) result the refinement started from */
    double tvec0[3];
    double cost0;        /* 0.5 * sum of squared reprojection residuals at the EPnP pose */
    double cost;         /* ... at the final pose */
} ctag_pose_rec;         /* 136 bytes */

typedef struct ctag_model ctag_model; /* host + device copy of a model list */

/* ---- loaders (no OpenCV FileStorage) ------------------------------------------------------------- */
/* Parses a .model text file exactly as CylinderTag::loadModel does (CylinderTag.cpp:161-190). */
int ctag_model_load(const char* path, ctag_model** out);
/* Same from arrays (copied). */
int ctag_model_create(const ctag_model_view* view, ctag_model** out);
void ctag_model_free(ctag_model* m);
int ctag_model_get_view(const ctag_model* m, ctag_model_view* view); /* host pointers, owned by the model */
/* Parses the `cameraMatrix` and `distCoeffs` !!opencv-matrix nodes of an OpenCV YAML 1.0 file
 * (CylinderTag.cpp:192-196 reads them with cv::FileStorage). */
int ctag_camera_load(const char* path, ctag_camera* out);

/* ---- pose ---------------------------------------------------------------------------------------- */
/* Poses of all markers of n_frames detection results resident in DEVICE memory (as ctag_detect_batch_device
 * leaves them).  offsets_dev[f] .. offsets_dev[f+1] index the pose records of frame f in poses_dev (one record
 * per marker of a CTAG_OK frame, in marker order, CTAG_POSE_NO_MODEL records included so that record k of a frame
 * is marker k).  offsets_dev holds n_frames+1 int32, poses_dev `capacity` records (n_frames*CTAG_MAX_MARKERS is
 * always enough).  Enqueued on the handle's stream; returns without waiting.  If the batch has more markers than
 * `capacity` the surplus is not computed and offsets_dev[n_frames] still holds the needed count. */
int ctag_pose_batch_device(ctag_handle* h, const ctag_frame_result* results_dev, int n_frames, const ctag_model* model,
                           const ctag_camera* camera, int32_t* offsets_dev, ctag_pose_rec* poses_dev, int capacity);

/* One frame, host result in, host poses out: what CylinderTag::estimatePose does before its erase
 * (CylinderTag.cpp:198-204).  out holds result->n_markers records (0 for a frame whose status is not CTAG_OK). */
int ctag_estimate_pose(ctag_handle* h, const ctag_frame_result* result, const ctag_model* model, const ctag_camera* camera,
                       ctag_pose_rec* out);

/* device time of the last ctag_pose_batch_device call (needs CTAG_OPT_TIMING), milliseconds */
float ctag_pose_last_ms(ctag_handle* h);

#ifdef __cplusplus
}
#endif
#endif
