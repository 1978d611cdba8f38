/* ctag_pose_oracle.h -- C ABI of the CPU oracle of the pose back end.  TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's estimatePose path (/root/reference/CylinderTag.cpp:198-209,
 * pose_estimation.cpp:50-143) and of the OpenCV 4.5.3 / Ceres 2.0 arithmetic it calls (un-vendored third-party
 * dependencies, Release.props:6,11).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it; the product (cylindertag_amd/) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" against the real reference binary (no pose fixtures in the reference, and it
 * cannot be built here).  Pins: tests/test_pose_cpu.py (exact synthetic poses, scipy least_squares minimum, the
 * reference's own test.bmp + CTag_2f12c.model + cameraParams.yml).
 */
#ifndef CTAG_POSE_ORACLE_H
#define CTAG_POSE_ORACLE_H
#include "../include/ctag_pose.h"

#ifdef __cplusplus
extern "C" {
#endif

/* cv::undistortPoints (5 iterations); with_P != 0 maps back through the camera matrix (P = K) */
void ctago_undistort_points(const ctag_camera* cam, int n, const float* uv, int with_P, double* out);
/* cv::solvePnP(..., SOLVEPNP_EPNP): returns a CTAG_POSE_* status */
int ctago_solve_pnp_epnp(const ctag_camera* cam, int n, const float* obj, const float* img, double* rvec, double* tvec);
/* PoseEstimator::PoseBA: refines rvec/tvec in place, returns the number of LM iterations */
int ctago_pose_ba(const ctag_camera* cam, int n, const float* obj, const float* img, double* rvec, double* tvec, double* cost0,
                  double* cost);
/* the correspondence builder of PoseEstimator::PnPSolver (pose_estimation.cpp:72-95) */
int ctago_build_correspondences(const ctag_frame_result* r, int marker, const ctag_model_view* model, int model_index, float* obj,
                                float* img, int* n_out);
/* probes of the shared dense linear algebra (cylindertag_amd/csrc/ctag_linalg.h); op codes in ctag_pose_oracle.cpp */
void ctago_linalg_probe(int op, const double* in, double* out);
/* all markers of one frame result: out[r->n_markers]; returns the number of records written */
int ctago_pose_frame(const ctag_frame_result* r, const ctag_model_view* model, const ctag_camera* cam, int frame_index,
                     ctag_pose_rec* out);

#ifdef __cplusplus
}
#endif
#endif
