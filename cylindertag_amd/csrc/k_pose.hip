// k_pose.hip -- pose of every decoded marker of a batch, on the device, straight from the detection records.
// Replaces, for the GPU path,
//   CylinderTag::estimatePose        /root/reference/CylinderTag.cpp:198-209
//   PoseEstimator::PnPSolver         /root/reference/pose_estimation.cpp:50-98   correspondences + cv::solvePnP(SOLVEPNP_EPNP)
//   PoseEstimator::PoseBA            /root/reference/pose_estimation.cpp:100-143 cv::undistortPoints + Ceres LM on the
//                                                                                reprojection residual of :5-48
// Third-party arithmetic restated from the published algorithms (OpenCV 4.5.3 calib3d epnp.cpp / undistort, Ceres 2.0
// trust_region_minimizer.cc + levenberg_marquardt_strategy.cc); none of it is GEMM-shaped at these sizes (<= 160 points,
// 6 unknowns), so no MFMA: FP64 VALU, one wavefront per marker.
//
// Mapping: block = one wave = one marker.  Lanes are points wherever the work is per point (undistortion, barycentric
// coordinates, camera-frame points, residuals and Jacobian rows); every SUM the CPU path accumulates sequentially over
// the points is accumulated by ONE lane in the same order (lane = matrix entry: 144 entries of M^T M, 21+6+1 entries of
// the normal equations), so the result does not depend on the wave width and equals the sequential evaluation bit for
// bit.  The 12x12 symmetric eigenproblem runs as cyclic Jacobi in round-robin order, six disjoint rotations at a time.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <string>
#include <vector>

#include "../../include/ctag_pose.h"
#include "ctag_internal.h"
#include "ctag_linalg.h"

namespace ctag {

struct PoseCam {
    double fx, fy, cx, cy;
    double k[12];
};

struct PoseModelDev {
    int n_models, model_size;
    const int32_t* marker_id;
    const float* corners;
};

constexpr int kPoseMaxPts = CTAG_POSE_MAX_POINTS;
constexpr int kPoseSmallPts = 96;  // dictionaries of <= 12 columns (the reference's CTag_2f12c): 16 KB of LDS, two waves per SIMD
constexpr int kJStride = 15;  // 12 Jacobian entries + 2 residuals per point, odd stride: conflict-free lane-per-point writes

// LDS image of one marker's problem (doubles)
template <int kPoseMaxPts>
struct PoseLds {
    double X[kPoseMaxPts * 3];    // world points
    double OBS[kPoseMaxPts * 2];  // BA observations (undistorted, through K, rounded to float)
    union {
        struct {
            double US[kPoseMaxPts * 2];  // EPnP pixel coordinates
            double AL[kPoseMaxPts * 4];  // barycentric coordinates
            double PC[kPoseMaxPts * 3];  // camera-frame points
            double A[144], V[144];       // M^T M and its eigenvectors
            double E[kPoseMaxPts];       // per-point reprojection error
        } e;
        struct {
            double JR[kPoseMaxPts * kJStride];  // per point: 2 x 6 column-scaled Jacobian entries, 2 residuals
        } b;
    } u;
    double cws[12], ccs[12], ci[9];
    double vv[48];  // the four null-space vectors
    double L[60], rho[6];
    double betas[16];
    double Rs[36], ts[12], rep[4];
    double s9[9], s3a[3], s3b[3];
    double red[34];
    double x[6];
    double rot[18];
    int rflag[6];
    int jac_flag;
};

__device__ __forceinline__ void wave_sync() { __syncthreads(); }

#ifdef CTAG_POSE_PROF
__device__ unsigned long long g_pose_prof[16];
#define PROF_MARK(i)                                                                    \
    do {                                                                                \
        const unsigned long long now__ = __builtin_readcyclecounter();                  \
        if (lane == 0) atomicAdd(&g_pose_prof[i], now__ - prof_t);                      \
        prof_t = now__;                                                                 \
    } while (0)
#else
#define PROF_MARK(i)
#endif

// cvUndistortPointsInternal, 5 iterations (same statement as the oracle's)
__device__ __forceinline__ void undistort_normalised(const PoseCam& c, double u, double v, double& xo, double& yo) {
    double x = (u - c.cx) / c.fx, y = (v - c.cy) / c.fy;
    const double x0 = x, y0 = y;
    const double* k = c.k;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) {
            x = x0;
            y = y0;
            break;
        }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    xo = x;
    yo = y;
}

// epnp::gauss_newton on one lane
__device__ void gauss_newton(const double* L, const double* rho, double* b) {
    for (int it = 0; it < 5; it++) {
        double A[24], B[6], X[4];
        for (int i = 0; i < 6; i++) {
            const double* l = L + 10 * i;
            A[4 * i] = 2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3];
            A[4 * i + 1] = l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3];
            A[4 * i + 2] = l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3];
            A[4 * i + 3] = l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3];
            B[i] = rho[i] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] + l[4] * b[1] * b[2] +
                             l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] + l[8] * b[2] * b[3] + l[9] * b[3] * b[3]);
        }
        ctl::qr_solve<6, 4>(A, B, X);
        for (int i = 0; i < 4; i++) b[i] += X[i];
    }
}

// residual and Jacobian rows of point p under pose (R, dR, t): the arithmetic of the oracle's BA::eval
__device__ __forceinline__ void point_residual(const double* R, const double* dR, const double* x, double fx, double fy, double cx,
                                               double cy, const double* p, const double* ob, double& r0, double& r1, double* j0,
                                               double* j1, bool with_j) {
    const double P0 = (R[0] * p[0] + R[1] * p[1] + R[2] * p[2]) + x[3];
    const double P1 = (R[3] * p[0] + R[4] * p[1] + R[5] * p[2]) + x[4];
    const double P2 = (R[6] * p[0] + R[7] * p[1] + R[8] * p[2]) + x[5];
    const double iz = 1.0 / P2;
    r0 = (fx * (P0 * iz) + cx) - ob[0];
    r1 = (fy * (P1 * iz) + cy) - ob[1];
    if (with_j) {
        const double a0 = fx * iz, a1 = fy * iz;
        const double b0 = fx * P0 * iz * iz, b1 = fy * P1 * iz * iz;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double* D = dR + 9 * k;
            const double d0 = D[0] * p[0] + D[1] * p[1] + D[2] * p[2];
            const double d1 = D[3] * p[0] + D[4] * p[1] + D[5] * p[2];
            const double d2 = D[6] * p[0] + D[7] * p[1] + D[8] * p[2];
            j0[k] = a0 * d0 - b0 * d2;
            j1[k] = a1 * d1 - b1 * d2;
        }
        j0[3] = a0;
        j0[4] = 0.0;
        j0[5] = -b0;
        j1[3] = 0.0;
        j1[4] = a1;
        j1[5] = -b1;
    }
}

__global__ __launch_bounds__(256) void k_pose_offsets(const ctag_frame_result* __restrict__ res, int n_frames, int32_t* __restrict__ offsets) {
    // exclusive scan of the per-frame marker counts, one block
    __shared__ int32_t part[256];
    const int tid = threadIdx.x;
    const int per = (n_frames + 255) / 256;
    const int f0 = tid * per, f1 = min(f0 + per, n_frames);
    int32_t s = 0;
    // a corrupted record cannot make the work list (or a reader of FR.markers[]) run past the record's arrays
    auto count = [&](int f) -> int32_t { return res[f].status == CTAG_OK ? min(max(res[f].n_markers, 0), CTAG_MAX_MARKERS) : 0; };
    for (int f = f0; f < f1; f++) s += count(f);
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int32_t run = 0;
        for (int i = 0; i < 256; i++) {
            const int32_t v = part[i];
            part[i] = run;
            run += v;
        }
        offsets[n_frames] = run;
    }
    __syncthreads();
    int32_t run = part[tid];
    for (int f = f0; f < f1; f++) {
        offsets[f] = run;
        run += count(f);
    }
}

template <int PTS, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_pose(const ctag_frame_result* __restrict__ res, int n_frames, const int32_t* __restrict__ offsets,
                                             PoseModelDev model, PoseCam cam, ctag_pose_rec* __restrict__ out, int capacity) {
    __shared__ PoseLds<PTS> S;
    const int lane = threadIdx.x;
    const int total = min(offsets[n_frames], capacity);
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        // work item -> (frame, marker): last frame with offsets[f] <= w
        int lo = 0, hi = n_frames - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (offsets[mid] <= w) lo = mid; else hi = mid - 1;
        }
        const int frame = lo, mk = w - offsets[lo];
        const ctag_frame_result& FR = res[frame];
        const ctag_marker_rec M = FR.markers[mk];
#ifdef CTAG_POSE_PROF
        unsigned long long prof_t = __builtin_readcyclecounter();
#endif
        ctag_pose_rec* P = out + w;
        wave_sync();  // previous item's LDS reads are done

        // ---- model lookup (pose_estimation.cpp:57-70) and correspondences (:72-95)
        int mi = -1;
        for (int j = 0; j < model.n_models; j++)
            if (model.marker_id[j] == M.marker_id) {
                mi = j;
                break;
            }
        int status = mi < 0 ? CTAG_POSE_NO_MODEL : CTAG_POSE_OK;
        int n = 0;
        const int nf = M.n_features;
        // a marker that points outside the frame's feature array (hand-built or corrupted record) is rejected, never read
        if (status == CTAG_POSE_OK && (M.first_feature < 0 || nf < 0 || M.first_feature > CTAG_MAX_FEATURES - nf)) status = CTAG_POSE_BAD_POS;
        if (status == CTAG_POSE_OK) {
            const float* __restrict__ corners = model.corners + (size_t)mi * model.model_size * 24;
            for (int j = 0; j < nf; j++) {
                const ctag_feature_rec& F = FR.features[M.first_feature + j];
                const int idl = F.id_left, idr = F.id_right, pos = F.pos;
                const int d = idl - idr;
                const int ad = d < 0 ? -d : d;
                if (nf > 3 && (j == 0 || j == nf - 1) && (ad > 1 || idr == -1)) continue;
                if (j >= M.n_pos || pos < 0 || pos >= model.model_size) {
                    status = CTAG_POSE_BAD_POS;
                    break;
                }
                const int cnt = (ad < 3 && idr != -1) ? 8 : 4;
                if (n + cnt > model.model_size * 8 || n + cnt > PTS) {  // repeated positions: more points than the model has
                    status = CTAG_POSE_BAD_POS;
                    break;
                }
                if (lane < cnt) {
                    const int k = lane < 2 ? lane : lane < 4 ? lane + 2 : lane < 6 ? lane - 2 : lane;  // 0 1 4 5 2 3 6 7
                    const int i = n + lane;
                    const double u = (double)F.corners[2 * k], v = (double)F.corners[2 * k + 1];
                    double xn, yn;
                    undistort_normalised(cam, u, v, xn, yn);
                    S.u.e.US[2 * i] = (double)(float)xn * cam.fx + cam.cx;
                    S.u.e.US[2 * i + 1] = (double)(float)yn * cam.fy + cam.cy;
                    S.OBS[2 * i] = (double)(float)(cam.fx * xn + cam.cx);
                    S.OBS[2 * i + 1] = (double)(float)(cam.fy * yn + cam.cy);
                    const float* cp = corners + (pos * 8 + k) * 3;
                    S.X[3 * i] = (double)cp[0];
                    S.X[3 * i + 1] = (double)cp[1];
                    S.X[3 * i + 2] = (double)cp[2];
                }
                n += cnt;
            }
        }
        if (status == CTAG_POSE_OK && n < 4) status = CTAG_POSE_TOO_FEW;
        if (lane == 0) {
            P->status = status;
            P->model_index = mi;
            P->frame = frame;
            P->marker = mk;
            P->n_points = status == CTAG_POSE_BAD_POS ? 0 : n;
            P->iterations = 0;
            for (int i = 0; i < 3; i++) P->rvec[i] = P->tvec[i] = P->rvec0[i] = P->tvec0[i] = 0.0;
            P->cost0 = P->cost = 0.0;
        }
        if (status != CTAG_POSE_OK) continue;  // wave-uniform
        wave_sync();

        PROF_MARK(0);
        // =========================================== EPnP ===========================================
        const double dn = (double)n;
        // choose_control_points: centroid (lane j sums coordinate j in point order)
        if (lane < 3) {
            double s = 0.0;
            for (int i = 0; i < n; i++) s += S.X[3 * i + lane];
            S.cws[lane] = s / dn;
        }
        wave_sync();
        if (lane < 9) {  // PW0^T PW0, lane = entry (a,b)
            const int a = lane / 3, b = lane % 3;
            const double ca = S.cws[a], cb = S.cws[b];
            double s = 0.0;
            for (int i = 0; i < n; i++) s += (S.X[3 * i + a] - ca) * (S.X[3 * i + b] - cb);
            S.s9[lane] = s;
        }
        wave_sync();
        if (lane == 0) {
            double C[9], V[9], w3[3];
            for (int i = 0; i < 9; i++) C[i] = S.s9[i];
            ctl::jacobi_eig<3>(C, V, w3);
            int ord[3];
            ctl::sort_desc<3>(w3, ord);
            for (int i = 1; i < 4; i++) {
                const double dc = w3[ord[i - 1]];
                const double k = ctm::sqrt64((dc > 0 ? dc : 0.0) / dn);
                for (int j = 0; j < 3; j++) S.cws[3 * i + j] = S.cws[j] + k * V[j * 3 + ord[i - 1]];
            }
            // compute_barycentric_coordinates: CC^-1
            double cc[9], ci[9];
            for (int i = 0; i < 3; i++)
                for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = S.cws[3 * j + i] - S.cws[i];
            const bool ok = ctl::inv3(cc, ci);
            for (int i = 0; i < 9; i++) S.ci[i] = ok ? ci[i] : 0.0;
            S.jac_flag = ok ? 1 : 0;
            // compute_rho
            const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
            for (int i = 0; i < 6; i++) {
                double d2 = 0.0;
                for (int k = 0; k < 3; k++) d2 += (S.cws[3 * pa[i] + k] - S.cws[3 * pb[i] + k]) * (S.cws[3 * pa[i] + k] - S.cws[3 * pb[i] + k]);
                S.rho[i] = d2;
            }
        }
        wave_sync();
        if (!S.jac_flag) {
            if (lane == 0) P->status = CTAG_POSE_DEGENERATE;
            continue;
        }
        for (int i = lane; i < n; i += 64) {  // alphas
            const double p0 = S.X[3 * i] - S.cws[0], p1 = S.X[3 * i + 1] - S.cws[1], p2 = S.X[3 * i + 2] - S.cws[2];
            double a[4];
#pragma unroll
            for (int j = 0; j < 3; j++) a[1 + j] = S.ci[3 * j] * p0 + S.ci[3 * j + 1] * p1 + S.ci[3 * j + 2] * p2;
            a[0] = 1.0 - a[1] - a[2] - a[3];
#pragma unroll
            for (int j = 0; j < 4; j++) S.u.e.AL[4 * i + j] = a[j];
        }
        wave_sync();
        PROF_MARK(1);
        // M^T M: lane = entry (r,c), rows of M in point order (fill_M + cvMulTransposed)
        for (int e = lane; e < 144; e += 64) {
            const int r = e / 12, c = e % 12;
            const int rj = r / 3, rk = r % 3, cj = c / 3, ck = c % 3;
            double acc = 0.0;
            for (int i = 0; i < n; i++) {
                const double ar = S.u.e.AL[4 * i + rj], ac = S.u.e.AL[4 * i + cj];
                const double u = S.u.e.US[2 * i], v = S.u.e.US[2 * i + 1];
                const double m1r = rk == 0 ? ar * cam.fx : (rk == 1 ? 0.0 : ar * (cam.cx - u));
                const double m2r = rk == 0 ? 0.0 : (rk == 1 ? ar * cam.fy : ar * (cam.cy - v));
                const double m1c = ck == 0 ? ac * cam.fx : (ck == 1 ? 0.0 : ac * (cam.cx - u));
                const double m2c = ck == 0 ? 0.0 : (ck == 1 ? ac * cam.fy : ac * (cam.cy - v));
                acc += m1r * m1c;
                acc += m2r * m2c;
            }
            S.u.e.A[e] = acc;
            S.u.e.V[e] = (r == c) ? 1.0 : 0.0;
        }
        wave_sync();
        PROF_MARK(2);
        // cyclic Jacobi with the round-robin pair order of ctl::jacobi_eig_rr12: the six rotations of a round are computed
        // by six lanes and applied side by side (15 lanes own the 2x2 blocks between two pairs, 6 the pairs' own entries,
        // 72 lane-tasks the eigenvector columns); bit-identical to the sequential sweep, see ctl::rr12_pair
        for (int sweep = 0; sweep < 60; sweep++) {
            double sm = 0.0;
            for (int p = 0; p < 11; p++)
                for (int q = p + 1; q < 12; q++) sm += ctm::fabs64(S.u.e.A[p * 12 + q]);
            if (sm == 0.0) break;  // uniform
            for (int round = 0; round < 11; round++) {
                if (lane < 6) {
                    int p, q;
                    ctl::rr12_pair(round, lane, p, q);
                    const ctl::JacobiRot r = ctl::jacobi_rot(S.u.e.A[p * 12 + p], S.u.e.A[q * 12 + q], S.u.e.A[p * 12 + q], sweep);
                    S.rot[3 * lane] = r.s;
                    S.rot[3 * lane + 1] = r.tau;
                    S.rot[3 * lane + 2] = r.h;
                    S.rflag[lane] = r.zero ? 2 : (r.rotate ? 1 : 0);
                }
                wave_sync();
                if (lane < 15) {  // block between pair i and pair j, i < j (processing order)
                    int i = 0, e = lane;
                    while (e >= 5 - i) {
                        e -= 5 - i;
                        i++;
                    }
                    const int j = i + 1 + e;
                    const int fi = S.rflag[i], fj = S.rflag[j];
                    if (fi == 1 || fj == 1) {
                        int pi, qi, pj, qj;
                        ctl::rr12_pair(round, i, pi, qi);
                        ctl::rr12_pair(round, j, pj, qj);
                        double b00 = S.u.e.A[pi * 12 + pj], b01 = S.u.e.A[pi * 12 + qj], b10 = S.u.e.A[qi * 12 + pj], b11 = S.u.e.A[qi * 12 + qj];
                        if (fi == 1) {
                            const double s_ = S.rot[3 * i], t_ = S.rot[3 * i + 1];
                            ctl::jacobi_apply(b00, b10, s_, t_);
                            ctl::jacobi_apply(b01, b11, s_, t_);
                        }
                        if (fj == 1) {
                            const double s_ = S.rot[3 * j], t_ = S.rot[3 * j + 1];
                            ctl::jacobi_apply(b00, b01, s_, t_);
                            ctl::jacobi_apply(b10, b11, s_, t_);
                        }
                        S.u.e.A[pi * 12 + pj] = b00;
                        S.u.e.A[pj * 12 + pi] = b00;
                        S.u.e.A[pi * 12 + qj] = b01;
                        S.u.e.A[qj * 12 + pi] = b01;
                        S.u.e.A[qi * 12 + pj] = b10;
                        S.u.e.A[pj * 12 + qi] = b10;
                        S.u.e.A[qi * 12 + qj] = b11;
                        S.u.e.A[qj * 12 + qi] = b11;
                    }
                } else if (lane < 21) {  // the pair's own entries
                    const int i = lane - 15;
                    const int f = S.rflag[i];
                    if (f) {
                        int p, q;
                        ctl::rr12_pair(round, i, p, q);
                        if (f == 1) {
                            const double h = S.rot[3 * i + 2];
                            S.u.e.A[p * 12 + p] -= h;
                            S.u.e.A[q * 12 + q] += h;
                        }
                        S.u.e.A[p * 12 + q] = 0.0;
                        S.u.e.A[q * 12 + p] = 0.0;
                    }
                }
                for (int t = lane - 21; t < 72; t += 64) {  // eigenvector columns: task = (pair, row k)
                    if (t < 0) continue;
                    const int i = t / 12, k = t % 12;
                    if (S.rflag[i] == 1) {
                        int p, q;
                        ctl::rr12_pair(round, i, p, q);
                        double vx = S.u.e.V[k * 12 + p], vy = S.u.e.V[k * 12 + q];
                        ctl::jacobi_apply(vx, vy, S.rot[3 * i], S.rot[3 * i + 1]);
                        S.u.e.V[k * 12 + p] = vx;
                        S.u.e.V[k * 12 + q] = vy;
                    }
                }
                wave_sync();
            }
        }
        wave_sync();
        PROF_MARK(3);
        if (lane == 0) {  // the four smallest eigenvalues' vectors = rows 11, 10, 9, 8 of cvSVD's U^T
            double w12[12];
            for (int i = 0; i < 12; i++) w12[i] = S.u.e.A[i * 12 + i];
            int ord[12];
            ctl::sort_desc<12>(w12, ord);
            for (int j = 0; j < 4; j++)
                for (int k = 0; k < 12; k++) S.vv[12 * j + k] = S.u.e.V[k * 12 + ord[11 - j]];
        }
        wave_sync();
        if (lane < 6) {  // compute_L_6x10, lane = control-point pair
            const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
            const int a = pa[lane], b = pb[lane];
            double dv[4][3];
            for (int i = 0; i < 4; i++)
                for (int k = 0; k < 3; k++) dv[i][k] = S.vv[12 * i + 3 * a + k] - S.vv[12 * i + 3 * b + k];
            auto dot = [](const double* x, const double* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
            double* row = S.L + 10 * lane;
            row[0] = dot(dv[0], dv[0]);
            row[1] = 2.0 * dot(dv[0], dv[1]);
            row[2] = dot(dv[1], dv[1]);
            row[3] = 2.0 * dot(dv[0], dv[2]);
            row[4] = 2.0 * dot(dv[1], dv[2]);
            row[5] = dot(dv[2], dv[2]);
            row[6] = 2.0 * dot(dv[0], dv[3]);
            row[7] = 2.0 * dot(dv[1], dv[3]);
            row[8] = 2.0 * dot(dv[2], dv[3]);
            row[9] = dot(dv[3], dv[3]);
        }
        wave_sync();
        PROF_MARK(4);
        if (lane >= 1 && lane <= 3) {  // find_betas_approx_{1,2,3} + gauss_newton, lane = N
            double L[60], rho[6], be[4];
            for (int i = 0; i < 60; i++) L[i] = S.L[i];
            for (int i = 0; i < 6; i++) rho[i] = S.rho[i];
            if (lane == 1) {
                double A[24], B[6], b4[4];
                for (int i = 0; i < 6; i++) {
                    A[4 * i] = L[10 * i];
                    A[4 * i + 1] = L[10 * i + 1];
                    A[4 * i + 2] = L[10 * i + 3];
                    A[4 * i + 3] = L[10 * i + 6];
                    B[i] = rho[i];
                }
                ctl::qr_solve<6, 4>(A, B, b4);
                if (b4[0] < 0) {
                    be[0] = ctm::sqrt64(-b4[0]);
                    be[1] = -b4[1] / be[0];
                    be[2] = -b4[2] / be[0];
                    be[3] = -b4[3] / be[0];
                } else {
                    be[0] = ctm::sqrt64(b4[0]);
                    be[1] = b4[1] / be[0];
                    be[2] = b4[2] / be[0];
                    be[3] = b4[3] / be[0];
                }
            } else if (lane == 2) {
                double A[18], B[6], b3[3];
                for (int i = 0; i < 6; i++) {
                    A[3 * i] = L[10 * i];
                    A[3 * i + 1] = L[10 * i + 1];
                    A[3 * i + 2] = L[10 * i + 2];
                    B[i] = rho[i];
                }
                ctl::qr_solve<6, 3>(A, B, b3);
                if (b3[0] < 0) {
                    be[0] = ctm::sqrt64(-b3[0]);
                    be[1] = (b3[2] < 0) ? ctm::sqrt64(-b3[2]) : 0.0;
                } else {
                    be[0] = ctm::sqrt64(b3[0]);
                    be[1] = (b3[2] > 0) ? ctm::sqrt64(b3[2]) : 0.0;
                }
                if (b3[1] < 0) be[0] = -be[0];
                be[2] = 0.0;
                be[3] = 0.0;
            } else {
                double A[30], B[6], b5[5];
                for (int i = 0; i < 6; i++) {
                    for (int j = 0; j < 5; j++) A[5 * i + j] = L[10 * i + j];
                    B[i] = rho[i];
                }
                ctl::qr_solve<6, 5>(A, B, b5);
                if (b5[0] < 0) {
                    be[0] = ctm::sqrt64(-b5[0]);
                    be[1] = (b5[2] < 0) ? ctm::sqrt64(-b5[2]) : 0.0;
                } else {
                    be[0] = ctm::sqrt64(b5[0]);
                    be[1] = (b5[2] > 0) ? ctm::sqrt64(b5[2]) : 0.0;
                }
                if (b5[1] < 0) be[0] = -be[0];
                be[2] = b5[3] / be[0];
                be[3] = 0.0;
            }
            gauss_newton(L, rho, be);
            for (int i = 0; i < 4; i++) S.betas[4 * lane + i] = be[i];
        }
        wave_sync();
        PROF_MARK(5);
        for (int N = 1; N <= 3; N++) {  // compute_R_and_t + reprojection_error
            if (lane < 12) {  // compute_ccs: ccs[j][k] = sum_i betas[i] * v[i][3j+k], i ascending from 0
                double s = 0.0;
                for (int i = 0; i < 4; i++) s += S.betas[4 * N + i] * S.vv[12 * i + lane];
                S.ccs[lane] = s;
            }
            wave_sync();
            {
                // solve_for_sign looks at pcs[2] of point 0
                const double* a0 = S.u.e.AL;
                const double z0 = a0[0] * S.ccs[2] + a0[1] * S.ccs[5] + a0[2] * S.ccs[8] + a0[3] * S.ccs[11];
                const bool neg = z0 < 0.0;
                for (int i = lane; i < n; i += 64) {
                    const double* a = S.u.e.AL + 4 * i;
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const double v = a[0] * S.ccs[j] + a[1] * S.ccs[3 + j] + a[2] * S.ccs[6 + j] + a[3] * S.ccs[9 + j];
                        S.u.e.PC[3 * i + j] = neg ? -v : v;
                    }
                }
            }
            wave_sync();
            if (lane < 6) {  // centroids pc0 (lanes 0-2) and pw0 (lanes 3-5)
                const double* src = lane < 3 ? S.u.e.PC : S.X;
                const int j = lane % 3;
                double s = 0.0;
                for (int i = 0; i < n; i++) s += src[3 * i + j];
                (lane < 3 ? S.s3a : S.s3b)[j] = s / dn;
            }
            wave_sync();
            if (lane < 9) {
                const int j = lane / 3, k = lane % 3;
                const double cj = S.s3a[j], wk = S.s3b[k];
                double s = 0.0;
                for (int i = 0; i < n; i++) s += (S.u.e.PC[3 * i + j] - cj) * (S.X[3 * i + k] - wk);
                S.s9[lane] = s;
            }
            wave_sync();
            if (lane == 0) {
                double abt[9], U[9], sv[3], V[9], R[9];
                for (int i = 0; i < 9; i++) abt[i] = S.s9[i];
                ctl::svd3(abt, U, sv, V);
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
                const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] -
                                   R[0] * R[5] * R[7];
                if (det < 0) {
                    R[6] = -R[6];
                    R[7] = -R[7];
                    R[8] = -R[8];
                }
                for (int i = 0; i < 9; i++) S.Rs[9 * N + i] = R[i];
                for (int j = 0; j < 3; j++)
                    S.ts[3 * N + j] = S.s3a[j] - (R[3 * j] * S.s3b[0] + R[3 * j + 1] * S.s3b[1] + R[3 * j + 2] * S.s3b[2]);
            }
            wave_sync();
            for (int i = lane; i < n; i += 64) {
                const double* R = S.Rs + 9 * N;
                const double* t = S.ts + 3 * N;
                const double* pw = S.X + 3 * i;
                const double Xc = R[0] * pw[0] + R[1] * pw[1] + R[2] * pw[2] + t[0];
                const double Yc = R[3] * pw[0] + R[4] * pw[1] + R[5] * pw[2] + t[1];
                const double inv_Zc = 1.0 / (R[6] * pw[0] + R[7] * pw[1] + R[8] * pw[2] + t[2]);
                const double ue = cam.cx + cam.fx * Xc * inv_Zc, ve = cam.cy + cam.fy * Yc * inv_Zc;
                const double u = S.u.e.US[2 * i], v = S.u.e.US[2 * i + 1];
                S.u.e.E[i] = ctm::sqrt64((u - ue) * (u - ue) + (v - ve) * (v - ve));
            }
            wave_sync();
            if (lane == 0) {
                double s = 0.0;
                for (int i = 0; i < n; i++) s += S.u.e.E[i];
                S.rep[N] = s / dn;
            }
            wave_sync();
        }
        PROF_MARK(6);
        if (lane == 0) {
            int N = 1;
            if (S.rep[2] < S.rep[1]) N = 2;
            if (S.rep[3] < S.rep[N]) N = 3;
            bool fin = true;
            for (int i = 0; i < 9; i++) fin = fin && ctl::finite64(S.Rs[9 * N + i]);
            for (int i = 0; i < 3; i++) fin = fin && ctl::finite64(S.ts[3 * N + i]);
            double rv[3] = {0, 0, 0};
            if (fin) {
                double R[9];
                for (int i = 0; i < 9; i++) R[i] = S.Rs[9 * N + i];
                ctl::rodrigues_from_matrix(R, rv);
                for (int i = 0; i < 3; i++) fin = fin && ctl::finite64(rv[i]);
            }
            S.jac_flag = fin ? 1 : 0;
            for (int i = 0; i < 3; i++) {
                S.x[i] = rv[i];
                S.x[3 + i] = S.ts[3 * N + i];
            }
        }
        wave_sync();
        if (!S.jac_flag) {
            if (lane == 0) P->status = CTAG_POSE_DEGENERATE;
            continue;
        }

        PROF_MARK(7);
        // =========================================== PoseBA ===========================================
        // Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy (see the oracle for the statement of the loop).
        // All lanes carry the same x / radius / cost.  Lane = point writes its two (column-scaled) Jacobian rows and
        // residuals to LDS; lane e < 34 owns one sequentially accumulated sum over those rows: e < 21 an entry of
        // J^T J, 21..26 of J^T r, 27 the cost, 28..33 a squared column norm.  Every owner runs the same loop
        // (acc += o[i0]*o[i1]; acc += o[i2]*o[i3]) with its own four offsets, so the 34 sums advance together.  The
        // candidate point is evaluated WITH its Jacobian and normal equations, which an accepted step then keeps.
        double x[6], xc[6];
#pragma unroll
        for (int i = 0; i < 6; i++) x[i] = S.x[i];
        if (lane == 0)
            for (int i = 0; i < 3; i++) {
                P->rvec0[i] = x[i];
                P->tvec0[i] = x[3 + i];
            }
        int i0 = 12, i1 = 12, i2 = 13, i3 = 13;  // lane 27 (and the idle lanes): the cost
        {
            int e = lane, a = 0;
            if (e < 21) {
                while (e >= 6 - a) {
                    e -= 6 - a;
                    a++;
                }
                i0 = a;
                i1 = a + e;
                i2 = 6 + a;
                i3 = 6 + a + e;
            } else if (e < 27) {
                i0 = e - 21;
                i1 = 12;
                i2 = 6 + e - 21;
                i3 = 13;
            } else if (e >= 28 && e < 34) {
                i0 = i1 = e - 28;
                i2 = i3 = 6 + e - 28;
            }
        }
        double scale[6] = {1, 1, 1, 1, 1, 1};
        auto eval_points = [&](const double* y) {  // rows of every point at pose y -> S.u.b.JR
            double R[9], dR[27];
            ctl::angle_axis_rot(y, R, dR);
            wave_sync();
            for (int i = lane; i < n; i += 64) {
                double r0, r1, j0[6], j1[6];
                point_residual(R, dR, y, cam.fx, cam.fy, cam.cx, cam.cy, S.X + 3 * i, S.OBS + 2 * i, r0, r1, j0, j1, true);
                double* o = S.u.b.JR + i * kJStride;
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    o[a] = j0[a] * scale[a];
                    o[6 + a] = j1[a] * scale[a];
                }
                o[12] = r0;
                o[13] = r1;
            }
            wave_sync();
        };
        auto accumulate = [&]() -> double {
            double acc = 0.0;
            if (lane < 34) {
                for (int q = 0; q < n; q++) {
                    const double* o = S.u.b.JR + q * kJStride;
                    acc += o[i0] * o[i1];
                    acc += o[i2] * o[i3];
                }
            }
            return acc;
        };
        double H[36], g[6];
        auto publish = [&](double v) {  // the owners' sums to LDS
            wave_sync();
            if (lane < 34) S.red[lane] = v;
            wave_sync();
        };
        auto take = [&](double* Ho, double* go) {  // ... and from there to every lane
            int e = 0;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = a; b < 6; b++) {
                    Ho[a * 6 + b] = S.red[e];
                    Ho[b * 6 + a] = S.red[e];
                    e++;
                }
#pragma unroll
            for (int a = 0; a < 6; a++) go[a] = S.red[21 + a];
        };
        auto gradient_max = [&]() {
            double m = 0.0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const double v = ctm::fabs64(g[a] / scale[a]);
                if (v > m) m = v;
            }
            return m;
        };
        // first evaluation with unit scaling: cost and the Jacobi scaling (column norms); then the rows are scaled in place
        eval_points(x);
        publish(accumulate());
        double cost = 0.5 * S.red[27];
#pragma unroll
        for (int a = 0; a < 6; a++) scale[a] = 1.0 / (1.0 + ctm::sqrt64(S.red[28 + a]));
        const double cost0 = cost;
        int iter = 0;
        bool live = ctl::finite64(cost);
        if (live) {
            for (int i = lane; i < n; i += 64) {
                double* o = S.u.b.JR + i * kJStride;
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    o[a] = o[a] * scale[a];
                    o[6 + a] = o[6 + a] * scale[a];
                }
            }
            wave_sync();
            publish(accumulate());
            take(H, g);
            if (gradient_max() <= 1e-15) live = false;
        }
        double radius = 1e4, decrease_factor = 2.0;
        while (live && iter < 50) {
            iter++;
            if (radius < 1e-32) break;
            double A[36], rhs[6], delta[6];
#pragma unroll
            for (int i = 0; i < 36; i++) A[i] = H[i];
#pragma unroll
            for (int a = 0; a < 6; a++) {
                double d = H[a * 6 + a];
                d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
                A[a * 6 + a] += d / radius;
                rhs[a] = -g[a];
            }
            bool ok = ctl::chol6_solve(A, rhs, delta);
            double model_cost_change = 0.0;
            if (ok) {
                double dg = 0.0, dHd = 0.0;
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    dg += delta[a] * g[a];
                    double hd = 0.0;
#pragma unroll
                    for (int b = 0; b < 6; b++) hd += H[a * 6 + b] * delta[b];
                    dHd += delta[a] * hd;
                }
                model_cost_change = -(dg + 0.5 * dHd);
#pragma unroll
                for (int a = 0; a < 6; a++) ok = ok && ctl::finite64(delta[a]);
            }
            if (!ok || !(model_cost_change > 0.0)) {
                radius = radius / decrease_factor;
                decrease_factor *= 2.0;
                continue;
            }
            double step2 = 0.0, x2 = 0.0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const double du = delta[a] * scale[a];
                xc[a] = x[a] + du;
                step2 += du * du;
                x2 += x[a] * x[a];
            }
            eval_points(xc);
            publish(accumulate());
            const double cost_c = 0.5 * S.red[27];
            const bool finite = ctl::finite64(cost_c);
            if (ctm::sqrt64(step2) <= 1e-10 * (ctm::sqrt64(x2) + 1e-10)) break;
            const double cost_change = cost - cost_c;
            if (finite && ctm::fabs64(cost_change) <= 1e-15 * cost) break;
            const double rho = cost_change / model_cost_change;
            if (finite && rho > 1e-3) {
#pragma unroll
                for (int a = 0; a < 6; a++) x[a] = xc[a];
                take(H, g);
                cost = cost_c;
                const double t = 2.0 * rho - 1.0;
                double f = 1.0 - t * t * t;
                if (f < 1.0 / 3.0) f = 1.0 / 3.0;
                radius = radius / f;
                if (radius > 1e16) radius = 1e16;
                decrease_factor = 2.0;
                if (gradient_max() <= 1e-15) break;
            } else {
                radius = radius / decrease_factor;
                decrease_factor *= 2.0;
            }
        }
        PROF_MARK(8);
        if (lane == 0) {
            for (int i = 0; i < 3; i++) {
                P->rvec[i] = x[i];
                P->tvec[i] = x[3 + i];
            }
            P->iterations = iter;
            P->cost0 = cost0;
            P->cost = cost;
        }
    }
}

}  // namespace ctag

// =====================================================================================================
// host side: model / camera objects, loaders, launch
// =====================================================================================================
struct ctag_model {
    int n_models = 0, model_size = 0;
    std::vector<int32_t> ids;
    std::vector<float> base, axis, corners;
    // device copies, created on first use on a device
    int device = -1;
    int32_t* d_ids = nullptr;
    float* d_corners = nullptr;
};

namespace {

struct PoseState {
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool pending = false;
    float last_ms = 0.f;
    int32_t* d_offsets = nullptr;  // scratch of ctag_estimate_pose
    ctag_pose_rec* d_poses = nullptr;
    ctag_frame_result* d_result = nullptr;
};

void pose_state_free(void* p) {
    PoseState* s = static_cast<PoseState*>(p);
    for (auto& e : s->ev)
        if (e) (void)hipEventDestroy(e);
    if (s->d_offsets) (void)hipFree(s->d_offsets);
    if (s->d_poses) (void)hipFree(s->d_poses);
    if (s->d_result) (void)hipFree(s->d_result);
    delete s;
}

PoseState* pose_state(ctag_handle* h) {
    void** slot = ctag::handle_pose_slot(h, pose_state_free);
    if (!*slot) {
        PoseState* s = new (std::nothrow) PoseState();
        if (!s) return nullptr;
        if (hipEventCreate(&s->ev[0]) != hipSuccess || hipEventCreate(&s->ev[1]) != hipSuccess) {
            pose_state_free(s);
            return nullptr;
        }
        *slot = s;
    }
    return static_cast<PoseState*>(*slot);
}

int model_to_device(ctag_model* m, int device) {
    if (m->device == device && m->d_ids) return CTAG_OK;
    if (m->d_ids) (void)hipFree(m->d_ids);
    if (m->d_corners) (void)hipFree(m->d_corners);
    m->d_ids = nullptr;
    m->d_corners = nullptr;
    if (hipMalloc(&m->d_ids, sizeof(int32_t) * std::max<size_t>(1, m->ids.size())) != hipSuccess) return CTAG_ERR_HIP;
    if (hipMalloc(&m->d_corners, sizeof(float) * std::max<size_t>(1, m->corners.size())) != hipSuccess) return CTAG_ERR_HIP;
    if (hipMemcpy(m->d_ids, m->ids.data(), sizeof(int32_t) * m->ids.size(), hipMemcpyHostToDevice) != hipSuccess) return CTAG_ERR_HIP;
    if (hipMemcpy(m->d_corners, m->corners.data(), sizeof(float) * m->corners.size(), hipMemcpyHostToDevice) != hipSuccess)
        return CTAG_ERR_HIP;
    m->device = device;
    return CTAG_OK;
}

bool camera_ok(const ctag_camera* c) {
    if (!c) return false;
    if (!(c->n_dist == 0 || c->n_dist == 4 || c->n_dist == 5 || c->n_dist == 8 || c->n_dist == 12 || c->n_dist == 14)) return false;
    if (c->n_dist == 14 && (c->dist[12] != 0.f || c->dist[13] != 0.f)) return false;  // tilted sensor model not supported
    return c->K[0] != 0.f && c->K[4] != 0.f;
}

}  // namespace

extern "C" {

int ctag_model_create(const ctag_model_view* v, ctag_model** out) {
    if (!v || !out || v->n_models < 0 || v->model_size < 1 || v->model_size > (1 << 16) || (v->n_models > 0 && (!v->marker_id || !v->corners)))
        return CTAG_ERR_ARG;
    ctag_model* m = new (std::nothrow) ctag_model();
    if (!m) return CTAG_ERR_ARG;
    m->n_models = v->n_models;
    m->model_size = v->model_size;
    const size_t n = (size_t)v->n_models;
    m->ids.assign(v->marker_id, v->marker_id + n);
    m->base.assign(n * 3, 0.f);
    m->axis.assign(n * 3, 0.f);
    if (v->base) m->base.assign(v->base, v->base + n * 3);
    if (v->axis) m->axis.assign(v->axis, v->axis + n * 3);
    m->corners.assign(v->corners, v->corners + n * v->model_size * 24);
    *out = m;
    return CTAG_OK;
}

// CylinderTag::loadModel (CylinderTag.cpp:161-190): "model_num model_size", then per model: id, base xyz, axis xyz and
// model_size*8 lines "corner_id x y z" (values parsed as float, stored at corner_id).
int ctag_model_load(const char* path, ctag_model** out) {
    if (!path || !out) return CTAG_ERR_ARG;
    std::ifstream in(path);
    if (!in.is_open()) return CTAG_ERR_ARG;
    int n = 0, size = 0;
    in >> n >> size;
    if (!in || n < 0 || n > (1 << 20) || size < 1 || size > (1 << 16)) return CTAG_ERR_ARG;
    ctag_model* m = new (std::nothrow) ctag_model();
    if (!m) return CTAG_ERR_ARG;
    m->n_models = n;
    m->model_size = size;
    m->ids.assign(n, 0);
    m->base.assign((size_t)n * 3, 0.f);
    m->axis.assign((size_t)n * 3, 0.f);
    m->corners.assign((size_t)n * size * 24, 0.f);
    for (int i = 0; i < n; i++) {
        in >> m->ids[i];
        for (int k = 0; k < 3; k++) in >> m->base[3 * i + k];
        for (int k = 0; k < 3; k++) in >> m->axis[3 * i + k];
        for (int j = 0; j < 8 * size; j++) {
            int cid = -1;
            float x = 0, y = 0, z = 0;
            in >> cid >> x >> y >> z;
            if (!in || cid < 0 || cid >= 8 * size) {  // the reference would write out of bounds
                delete m;
                return CTAG_ERR_ARG;
            }
            float* c = &m->corners[((size_t)i * size * 8 + cid) * 3];
            c[0] = x;
            c[1] = y;
            c[2] = z;
        }
    }
    *out = m;
    return CTAG_OK;
}

void ctag_model_free(ctag_model* m) {
    if (!m) return;
    if (m->d_ids) (void)hipFree(m->d_ids);
    if (m->d_corners) (void)hipFree(m->d_corners);
    delete m;
}

int ctag_model_get_view(const ctag_model* m, ctag_model_view* v) {
    if (!m || !v) return CTAG_ERR_ARG;
    v->n_models = m->n_models;
    v->model_size = m->model_size;
    v->marker_id = m->ids.data();
    v->base = m->base.data();
    v->axis = m->axis.data();
    v->corners = m->corners.data();
    return CTAG_OK;
}

// The two nodes cv::FileStorage reads at CylinderTag.cpp:192-196, from an OpenCV "%YAML:1.0" file:
//   name: !!opencv-matrix \n rows: R \n cols: C \n dt: f|d \n data: [ v, v, ... ]
int ctag_camera_load(const char* path, ctag_camera* out) {
    if (!path || !out) return CTAG_ERR_ARG;
    std::ifstream in(path);
    if (!in.is_open()) return CTAG_ERR_ARG;
    std::string txt((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    auto read_node = [&](const char* name, int& rows, int& cols, std::vector<double>& data) -> bool {
        size_t p = 0;
        const std::string key = std::string(name) + ":";
        for (;;) {  // a key at the start of a line
            p = txt.find(key, p);
            if (p == std::string::npos) return false;
            if (p == 0 || txt[p - 1] == '\n') break;
            p += key.size();
        }
        auto field = [&](const char* f, size_t from) -> size_t {
            const size_t q = txt.find(f, from);
            return q == std::string::npos ? q : q + std::strlen(f);
        };
        size_t q = field("rows:", p);
        if (q == std::string::npos) return false;
        rows = std::atoi(txt.c_str() + q);
        q = field("cols:", q);
        if (q == std::string::npos) return false;
        cols = std::atoi(txt.c_str() + q);
        q = field("data:", q);
        if (q == std::string::npos) return false;
        q = txt.find('[', q);
        const size_t e = txt.find(']', q);
        if (q == std::string::npos || e == std::string::npos) return false;
        data.clear();
        const char* s = txt.c_str() + q + 1;
        const char* end = txt.c_str() + e;
        while (s < end) {
            char* nx = nullptr;
            const double v = std::strtod(s, &nx);
            if (nx == s) {
                s++;
                continue;
            }
            data.push_back(v);
            s = nx;
        }
        return rows > 0 && cols > 0 && (size_t)rows * cols == data.size();
    };
    int r = 0, c = 0;
    std::vector<double> d;
    std::memset(out, 0, sizeof(*out));
    if (!read_node("cameraMatrix", r, c, d) || r != 3 || c != 3) return CTAG_ERR_ARG;
    for (int i = 0; i < 9; i++) out->K[i] = (float)d[i];
    if (!read_node("distCoeffs", r, c, d) || d.size() > 14) return CTAG_ERR_ARG;
    for (size_t i = 0; i < d.size(); i++) out->dist[i] = (float)d[i];
    out->n_dist = (int)d.size();
    return camera_ok(out) ? CTAG_OK : CTAG_ERR_UNSUPPORTED;
}

int ctag_pose_batch_device(ctag_handle* h, const ctag_frame_result* results_dev, int n_frames, const ctag_model* model_c,
                           const ctag_camera* camera, int32_t* offsets_dev, ctag_pose_rec* poses_dev, int capacity) {
    if (!h || !results_dev || n_frames < 0 || !model_c || !offsets_dev || !poses_dev || capacity < 0) return CTAG_ERR_ARG;
    if (!camera_ok(camera)) return CTAG_ERR_UNSUPPORTED;
    ctag_model* model = const_cast<ctag_model*>(model_c);
    const int dev = ctag::handle_device(h);
    if (hipSetDevice(dev) != hipSuccess) return CTAG_ERR_HIP;
    if (model_to_device(model, dev) != CTAG_OK) return CTAG_ERR_HIP;
    PoseState* st = pose_state(h);
    if (!st) return CTAG_ERR_HIP;
    {   // records of frames that wait for the any-frame pass (CTAG_PENDING) are completed before they are read
        const int fr = ctag::handle_finish_pending(h);
        if (fr != CTAG_OK) return fr;
    }
    hipStream_t s = static_cast<hipStream_t>(ctag_stream(h));
    ctag::PoseCam cam;
    cam.fx = (double)camera->K[0];
    cam.fy = (double)camera->K[4];
    cam.cx = (double)camera->K[2];
    cam.cy = (double)camera->K[5];
    for (int i = 0; i < 12; i++) cam.k[i] = i < camera->n_dist ? (double)camera->dist[i] : 0.0;
    ctag::PoseModelDev md{model->n_models, model->model_size, model->d_ids, model->d_corners};
    const bool timing = ctag::handle_timing(h);
    if (timing && hipEventRecord(st->ev[0], s) != hipSuccess) return CTAG_ERR_HIP;
    hipLaunchKernelGGL(ctag::k_pose_offsets, dim3(1), dim3(256), 0, s, results_dev, n_frames, offsets_dev);
    if (n_frames > 0 && capacity > 0) {
        const int grid = std::min(capacity, 256 * 16);
        if (model->model_size * 8 <= ctag::kPoseSmallPts)
            hipLaunchKernelGGL((ctag::k_pose<ctag::kPoseSmallPts, 2>), dim3(grid), dim3(64), 0, s, results_dev, n_frames, offsets_dev, md, cam,
                               poses_dev, capacity);
        else
            hipLaunchKernelGGL((ctag::k_pose<ctag::kPoseMaxPts, 1>), dim3(grid), dim3(64), 0, s, results_dev, n_frames, offsets_dev, md, cam,
                               poses_dev, capacity);
    }
    if (hipGetLastError() != hipSuccess) return CTAG_ERR_HIP;
    if (timing) {
        if (hipEventRecord(st->ev[1], s) != hipSuccess) return CTAG_ERR_HIP;
        st->pending = true;
    }
    return CTAG_OK;
}

#ifdef CTAG_POSE_PROF
int ctag_pose_debug_prof(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ctag::g_pose_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return CTAG_ERR_HIP;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ctag::g_pose_prof), z, sizeof(z)) != hipSuccess) return CTAG_ERR_HIP;
    }
    return CTAG_OK;
}
#endif

float ctag_pose_last_ms(ctag_handle* h) {
    if (!h) return 0.f;
    PoseState* st = pose_state(h);
    if (!st) return 0.f;
    if (st->pending) {
        if (hipEventSynchronize(st->ev[1]) == hipSuccess) (void)hipEventElapsedTime(&st->last_ms, st->ev[0], st->ev[1]);
        st->pending = false;
    }
    return st->last_ms;
}

int ctag_estimate_pose(ctag_handle* h, const ctag_frame_result* result, const ctag_model* model, const ctag_camera* camera,
                       ctag_pose_rec* out) {
    if (!h || !result || !model || !camera) return CTAG_ERR_ARG;
    if (result->status != CTAG_OK || result->n_markers <= 0) return CTAG_OK;
    if (!out || result->n_markers > CTAG_MAX_MARKERS) return CTAG_ERR_ARG;
    const int dev = ctag::handle_device(h);
    if (hipSetDevice(dev) != hipSuccess) return CTAG_ERR_HIP;
    PoseState* st = pose_state(h);
    if (!st) return CTAG_ERR_HIP;
    if (!st->d_result) {
        if (hipMalloc(&st->d_result, sizeof(ctag_frame_result)) != hipSuccess) return CTAG_ERR_HIP;
        if (hipMalloc(&st->d_offsets, 2 * sizeof(int32_t)) != hipSuccess) return CTAG_ERR_HIP;
        if (hipMalloc(&st->d_poses, CTAG_MAX_MARKERS * sizeof(ctag_pose_rec)) != hipSuccess) return CTAG_ERR_HIP;
    }
    hipStream_t s = static_cast<hipStream_t>(ctag_stream(h));
    if (hipMemcpyAsync(st->d_result, result, sizeof(ctag_frame_result), hipMemcpyHostToDevice, s) != hipSuccess) return CTAG_ERR_HIP;
    const int rc = ctag_pose_batch_device(h, st->d_result, 1, model, camera, st->d_offsets, st->d_poses, CTAG_MAX_MARKERS);
    if (rc != CTAG_OK) return rc;
    if (hipMemcpyAsync(out, st->d_poses, sizeof(ctag_pose_rec) * result->n_markers, hipMemcpyDeviceToHost, s) != hipSuccess)
        return CTAG_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess) return CTAG_ERR_HIP;
    return CTAG_OK;
}

}  // extern "C"
