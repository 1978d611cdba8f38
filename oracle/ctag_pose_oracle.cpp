// ctag_pose_oracle.cpp -- CPU oracle of the pose back end.  TEST INFRASTRUCTURE ONLY (see ctag_pose_oracle.h).
//
// Single-threaded restatement of
//   PoseEstimator::PnPSolver   /root/reference/pose_estimation.cpp:50-98
//   PoseEstimator::PoseBA      /root/reference/pose_estimation.cpp:100-127   (+ buildProblem :129-143, the residual :5-48)
//   CylinderTag::estimatePose  /root/reference/CylinderTag.cpp:198-209
// and of the third-party arithmetic they call, which the reference does not vendor and this image lacks:
//   OpenCV 4.5.3 (Release.props:11)  undistortPoints, solvePnP(SOLVEPNP_EPNP) = calib3d/src/epnp.cpp, Rodrigues
//   Ceres 2.0    (Release.props:6)   TrustRegionMinimizer + LevenbergMarquardtStrategy, AutoDiff of AngleAxisRotatePoint
// restated from the published algorithms (Lepetit/Moreno-Noguer/Fua EPnP; Ceres' documented LM loop).
//
// PARITY STATUS: "parity unpinned" -- the reference holds no pose fixtures and cannot be built here.  What pins this
// file instead (tests/test_pose_cpu.py): exact synthetic poses are recovered; on noisy data the result equals
// scipy.optimize.least_squares' minimum of the same residual; on the reference's own test.bmp + CTag_2f12c.model +
// cameraParams.yml every decoded marker gets a sub-pixel reprojection RMS.
#include "ctag_pose_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../cylindertag_amd/csrc/ctag_linalg.h"

namespace {

struct Cam {
    double fx, fy, cx, cy;
    double k[12];
};

Cam make_cam(const ctag_camera* c) {
    Cam m;
    // solvePnP / undistortPoints convert the CV_32F matrices to double first
    m.fx = (double)c->K[0];
    m.fy = (double)c->K[4];
    m.cx = (double)c->K[2];
    m.cy = (double)c->K[5];
    for (int i = 0; i < 12; i++) m.k[i] = (i < c->n_dist) ? (double)c->dist[i] : 0.0;
    return m;
}

// cvUndistortPointsInternal (OpenCV 4.5.3 imgproc/src/undistort.dispatch.cpp): 5 fixed-point iterations
// (the public undistortPoints passes TermCriteria(MAX_ITER, 5, 0.01)); result in normalised coordinates.
void undistort_normalised(const Cam& c, double u, double v, double& xo, double& yo) {
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

// ---------------------------------------------------------------------------------------------------------------
// EPnP, OpenCV calib3d/src/epnp.cpp (class epnp).  pws: world points, us: pixel coordinates of the undistorted points.
// ---------------------------------------------------------------------------------------------------------------
struct Epnp {
    int n;
    Cam cam;
    std::vector<double> pws, us, alphas, pcs;
    double cws[4][3], ccs[4][3];

    void choose_control_points() {  // epnp::choose_control_points
        cws[0][0] = cws[0][1] = cws[0][2] = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
        for (int j = 0; j < 3; j++) cws[0][j] /= n;
        double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // PW0^T PW0
        for (int i = 0; i < n; i++) {
            double d[3];
            for (int j = 0; j < 3; j++) d[j] = pws[3 * i + j] - cws[0][j];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) C[a * 3 + b] += d[a] * d[b];
        }
        double V[9], w[3];
        ctl::jacobi_eig<3>(C, V, w);
        int ord[3];
        ctl::sort_desc<3>(w, ord);
        for (int i = 1; i < 4; i++) {
            const double dc = w[ord[i - 1]];
            const double k = ctm::sqrt64((dc > 0 ? dc : 0.0) / n);
            for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * V[j * 3 + ord[i - 1]];
        }
    }
    bool compute_barycentric_coordinates() {  // epnp::compute_barycentric_coordinates
        double cc[9], ci[9];
        for (int i = 0; i < 3; i++)
            for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
        if (!ctl::inv3(cc, ci)) return false;
        alphas.resize(4 * n);
        for (int i = 0; i < n; i++) {
            const double* pi = &pws[3 * i];
            double* a = &alphas[4 * i];
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
            a[0] = 1.0 - a[1] - a[2] - a[3];
        }
        return true;
    }
    // rows 2i and 2i+1 of M (epnp::fill_M)
    void m_rows(int i, double* m1, double* m2) const {
        const double* a = &alphas[4 * i];
        const double u = us[2 * i], v = us[2 * i + 1];
        for (int j = 0; j < 4; j++) {
            m1[3 * j] = a[j] * cam.fx;
            m1[3 * j + 1] = 0.0;
            m1[3 * j + 2] = a[j] * (cam.cx - u);
            m2[3 * j] = 0.0;
            m2[3 * j + 1] = a[j] * cam.fy;
            m2[3 * j + 2] = a[j] * (cam.cy - v);
        }
    }
    void compute_ccs(const double* betas, const double* const v[4]) {
        for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
                for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[i][3 * j + k];
    }
    void compute_pcs() {
        pcs.resize(3 * n);
        for (int i = 0; i < n; i++) {
            const double* a = &alphas[4 * i];
            for (int j = 0; j < 3; j++) pcs[3 * i + j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
        }
    }
    void solve_for_sign() {
        if (pcs[2] < 0.0) {
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
            for (int i = 0; i < n; i++)
                for (int j = 0; j < 3; j++) pcs[3 * i + j] = -pcs[3 * i + j];
        }
    }
    void estimate_R_and_t(double R[9], double t[3]) {
        double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) {
                pc0[j] += pcs[3 * i + j];
                pw0[j] += pws[3 * i + j];
            }
        for (int j = 0; j < 3; j++) {
            pc0[j] /= n;
            pw0[j] /= n;
        }
        double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) abt[3 * j + k] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i + k] - pw0[k]);
        double U[9], s[3], V[9];
        ctl::svd3(abt, U, s, V);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
        const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] -
                           R[0] * R[5] * R[7];
        if (det < 0) {
            R[6] = -R[6];
            R[7] = -R[7];
            R[8] = -R[8];
        }
        for (int j = 0; j < 3; j++) t[j] = pc0[j] - (R[3 * j] * pw0[0] + R[3 * j + 1] * pw0[1] + R[3 * j + 2] * pw0[2]);
    }
    double reprojection_error(const double R[9], const double t[3]) const {
        double sum2 = 0.0;
        for (int i = 0; i < n; i++) {
            const double* pw = &pws[3 * i];
            const double Xc = R[0] * pw[0] + R[1] * pw[1] + R[2] * pw[2] + t[0];
            const double Yc = R[3] * pw[0] + R[4] * pw[1] + R[5] * pw[2] + t[1];
            const double inv_Zc = 1.0 / (R[6] * pw[0] + R[7] * pw[1] + R[8] * pw[2] + t[2]);
            const double ue = cam.cx + cam.fx * Xc * inv_Zc, ve = cam.cy + cam.fy * Yc * inv_Zc;
            const double u = us[2 * i], v = us[2 * i + 1];
            sum2 += ctm::sqrt64((u - ue) * (u - ue) + (v - ve) * (v - ve));
        }
        return sum2 / n;
    }
    double compute_R_and_t(const double* const v[4], const double* betas, double R[9], double t[3]) {
        compute_ccs(betas, v);
        compute_pcs();
        solve_for_sign();
        estimate_R_and_t(R, t);
        return reprojection_error(R, t);
    }
    static void gauss_newton(const double L[60], const double rho[6], double b[4]) {  // epnp::gauss_newton, 5 iterations
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

    // epnp::compute_pose; false when the configuration is degenerate
    bool compute_pose(double R[9], double t[3]) {
        choose_control_points();
        if (!compute_barycentric_coordinates()) return false;
        double MtM[144];
        for (int i = 0; i < 144; i++) MtM[i] = 0.0;
        for (int i = 0; i < n; i++) {
            double m1[12], m2[12];
            m_rows(i, m1, m2);
            for (int r = 0; r < 12; r++)
                for (int c = 0; c < 12; c++) {
                    MtM[r * 12 + c] += m1[r] * m1[c];
                    MtM[r * 12 + c] += m2[r] * m2[c];
                }
        }
        double V[144], w[12];
        ctl::jacobi_eig_rr12(MtM, V, w);  // cvSVD of the symmetric M^T M: cyclic Jacobi, round-robin pair order
        int ord[12];
        ctl::sort_desc<12>(w, ord);
        // ut rows 11, 10, 9, 8 of cvSVD(MtM) = eigenvectors of the four smallest eigenvalues
        double vv[4][12];
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 12; k++) vv[j][k] = V[k * 12 + ord[11 - j]];
        const double* const v[4] = {vv[0], vv[1], vv[2], vv[3]};
        // epnp::compute_L_6x10
        double dv[4][6][3];
        for (int i = 0; i < 4; i++) {
            int a = 0, b = 1;
            for (int j = 0; j < 6; j++) {
                for (int k = 0; k < 3; k++) dv[i][j][k] = v[i][3 * a + k] - v[i][3 * b + k];
                b++;
                if (b > 3) {
                    a++;
                    b = a + 1;
                }
            }
        }
        auto dot = [](const double* x, const double* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
        double L[60], rho[6];
        for (int i = 0; i < 6; i++) {
            double* row = L + 10 * i;
            row[0] = dot(dv[0][i], dv[0][i]);
            row[1] = 2.0 * dot(dv[0][i], dv[1][i]);
            row[2] = dot(dv[1][i], dv[1][i]);
            row[3] = 2.0 * dot(dv[0][i], dv[2][i]);
            row[4] = 2.0 * dot(dv[1][i], dv[2][i]);
            row[5] = dot(dv[2][i], dv[2][i]);
            row[6] = 2.0 * dot(dv[0][i], dv[3][i]);
            row[7] = 2.0 * dot(dv[1][i], dv[3][i]);
            row[8] = 2.0 * dot(dv[2][i], dv[3][i]);
            row[9] = dot(dv[3][i], dv[3][i]);
        }
        {  // epnp::compute_rho
            const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
            for (int i = 0; i < 6; i++) {
                double d2 = 0.0;
                for (int k = 0; k < 3; k++) d2 += (cws[pa[i]][k] - cws[pb[i]][k]) * (cws[pa[i]][k] - cws[pb[i]][k]);
                rho[i] = d2;
            }
        }
        double Betas[4][4], rep[4], Rs[4][9], ts[4][3];
        {  // epnp::find_betas_approx_1: betas10 = [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44], approx = [B11 B12 B13 B14]
            double A[24], B[6], b4[4];
            for (int i = 0; i < 6; i++) {
                A[4 * i] = L[10 * i];
                A[4 * i + 1] = L[10 * i + 1];
                A[4 * i + 2] = L[10 * i + 3];
                A[4 * i + 3] = L[10 * i + 6];
                B[i] = rho[i];
            }
            ctl::qr_solve<6, 4>(A, B, b4);
            double* be = Betas[1];
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
        }
        {  // epnp::find_betas_approx_2: approx = [B11 B12 B22]
            double A[18], B[6], b3[3];
            for (int i = 0; i < 6; i++) {
                A[3 * i] = L[10 * i];
                A[3 * i + 1] = L[10 * i + 1];
                A[3 * i + 2] = L[10 * i + 2];
                B[i] = rho[i];
            }
            ctl::qr_solve<6, 3>(A, B, b3);
            double* be = Betas[2];
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
        }
        {  // epnp::find_betas_approx_3: approx = [B11 B12 B22 B13 B23]
            double A[30], B[6], b5[5];
            for (int i = 0; i < 6; i++) {
                for (int j = 0; j < 5; j++) A[5 * i + j] = L[10 * i + j];
                B[i] = rho[i];
            }
            ctl::qr_solve<6, 5>(A, B, b5);
            double* be = Betas[3];
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
        for (int N = 1; N <= 3; N++) {
            gauss_newton(L, rho, Betas[N]);
            rep[N] = compute_R_and_t(v, Betas[N], Rs[N], ts[N]);
        }
        int N = 1;
        if (rep[2] < rep[1]) N = 2;
        if (rep[3] < rep[N]) N = 3;
        for (int i = 0; i < 9; i++) R[i] = Rs[N][i];
        for (int i = 0; i < 3; i++) t[i] = ts[N][i];
        for (int i = 0; i < 9; i++)
            if (!ctl::finite64(R[i])) return false;
        for (int i = 0; i < 3; i++)
            if (!ctl::finite64(t[i])) return false;
        return true;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// PoseBA: Ceres 2.0 TrustRegionMinimizer with the LevenbergMarquardtStrategy, options of pose_estimation.cpp:112-116
// (gradient_tolerance 1e-15, function_tolerance 1e-15, parameter_tolerance 1e-10; defaults otherwise: 50 iterations,
// initial radius 1e4, max radius 1e16, min radius 1e-32, min_relative_decrease 1e-3, LM diagonal clamp [1e-6, 1e32],
// Jacobi scaling).  DENSE_SCHUR on two dense 3-blocks solves the same 6x6 normal equations that Cholesky solves here.
// ---------------------------------------------------------------------------------------------------------------
struct BA {
    int n;
    double fx, fy, cx, cy;
    const double* X;    // world points
    const double* obs;  // observed (undistorted, re-projected with K) pixel positions

    // residuals (2n), jacobian rows (2n x 6, unscaled), returns cost = 0.5 |r|^2; false if not finite
    bool eval(const double x[6], std::vector<double>& r, std::vector<double>* J, double& cost) const {
        double R[9], dR[27];
        ctl::angle_axis_rot(x, R, J ? dR : nullptr);
        double c2 = 0.0;
        for (int i = 0; i < n; i++) {
            const double* p = X + 3 * i;
            const double P0 = (R[0] * p[0] + R[1] * p[1] + R[2] * p[2]) + x[3];
            const double P1 = (R[3] * p[0] + R[4] * p[1] + R[5] * p[2]) + x[4];
            const double P2 = (R[6] * p[0] + R[7] * p[1] + R[8] * p[2]) + x[5];
            const double iz = 1.0 / P2;
            const double r0 = (fx * (P0 * iz) + cx) - obs[2 * i];
            const double r1 = (fy * (P1 * iz) + cy) - obs[2 * i + 1];
            r[2 * i] = r0;
            r[2 * i + 1] = r1;
            c2 += r0 * r0;
            c2 += r1 * r1;
            if (J) {
                double* j0 = &(*J)[12 * i];
                double* j1 = j0 + 6;
                const double a0 = fx * iz, a1 = fy * iz;
                const double b0 = fx * P0 * iz * iz, b1 = fy * P1 * iz * iz;
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
        cost = 0.5 * c2;
        return ctl::finite64(cost);
    }

    int solve(double x[6], double& cost0, double& cost_final) const {
        std::vector<double> r(2 * n), J(12 * n), rc(2 * n);
        double cost;
        if (!eval(x, r, &J, cost)) {
            cost0 = cost_final = cost;
            return 0;
        }
        cost0 = cost;
        double scale[6];
        {
            double cn[6] = {0, 0, 0, 0, 0, 0};
            for (int i = 0; i < 2 * n; i++)
                for (int a = 0; a < 6; a++) cn[a] += J[6 * i + a] * J[6 * i + a];
            for (int a = 0; a < 6; a++) scale[a] = 1.0 / (1.0 + ctm::sqrt64(cn[a]));
        }
        double radius = 1e4, decrease_factor = 2.0;
        double H[36], g[6];  // J_s^T J_s and J_s^T r with J_s = J diag(scale)
        auto normal_equations = [&]() {
            for (int i = 0; i < 36; i++) H[i] = 0.0;
            for (int a = 0; a < 6; a++) g[a] = 0.0;
            for (int i = 0; i < 2 * n; i++) {
                double js[6];
                for (int a = 0; a < 6; a++) js[a] = J[6 * i + a] * scale[a];
                for (int a = 0; a < 6; a++) {
                    for (int b = a; b < 6; b++) H[a * 6 + b] += js[a] * js[b];
                    g[a] += js[a] * r[i];
                }
            }
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < a; b++) H[a * 6 + b] = H[b * 6 + a];
        };
        normal_equations();
        auto gradient_max = [&]() {  // max-norm of the UNSCALED gradient J^T r
            double m = 0.0;
            for (int a = 0; a < 6; a++) {
                const double v = ctm::fabs64(g[a] / scale[a]);
                if (v > m) m = v;
            }
            return m;
        };
        int iter = 0;
        if (gradient_max() <= 1e-15) {
            cost_final = cost;
            return 0;
        }
        while (iter < 50) {
            iter++;
            if (radius < 1e-32) break;
            // LevenbergMarquardtStrategy::ComputeStep
            double A[36], rhs[6], delta[6];
            for (int i = 0; i < 36; i++) A[i] = H[i];
            for (int a = 0; a < 6; a++) {
                double d = H[a * 6 + a];
                d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
                A[a * 6 + a] += d / radius;
                rhs[a] = -g[a];
            }
            bool ok = ctl::chol6_solve(A, rhs, delta);
            double model_cost_change = 0.0;
            if (ok) {
                // model_cost_change = -(J_s d)^T (r + J_s d / 2) = -(d^T g + d^T H d / 2)
                double dg = 0.0, dHd = 0.0;
                for (int a = 0; a < 6; a++) {
                    dg += delta[a] * g[a];
                    double hd = 0.0;
                    for (int b = 0; b < 6; b++) hd += H[a * 6 + b] * delta[b];
                    dHd += delta[a] * hd;
                }
                model_cost_change = -(dg + 0.5 * dHd);
                for (int a = 0; a < 6; a++) ok = ok && ctl::finite64(delta[a]);
            }
            if (!ok || !(model_cost_change > 0.0)) {  // invalid step
                radius = radius / decrease_factor;
                decrease_factor *= 2.0;
                continue;
            }
            double xc[6], step2 = 0.0, x2 = 0.0;
            for (int a = 0; a < 6; a++) {
                const double du = delta[a] * scale[a];
                xc[a] = x[a] + du;
                step2 += du * du;
                x2 += x[a] * x[a];
            }
            double cost_c;
            const bool finite = eval(xc, rc, nullptr, cost_c);
            // ParameterToleranceReached
            if (ctm::sqrt64(step2) <= 1e-10 * (ctm::sqrt64(x2) + 1e-10)) break;
            // FunctionToleranceReached
            const double cost_change = cost - cost_c;
            if (finite && ctm::fabs64(cost_change) <= 1e-15 * cost) break;
            const double rho = cost_change / model_cost_change;
            if (finite && rho > 1e-3) {  // HandleSuccessfulStep
                for (int a = 0; a < 6; a++) x[a] = xc[a];
                eval(x, r, &J, cost);
                normal_equations();
                const double t = 2.0 * rho - 1.0;
                double f = 1.0 - t * t * t;
                if (f < 1.0 / 3.0) f = 1.0 / 3.0;
                radius = radius / f;
                if (radius > 1e16) radius = 1e16;
                decrease_factor = 2.0;
                if (gradient_max() <= 1e-15) break;
            } else {  // HandleUnsuccessfulStep
                radius = radius / decrease_factor;
                decrease_factor *= 2.0;
            }
        }
        cost_final = cost;
        return iter;
    }
};

int find_model(const ctag_model_view* m, int marker_id) {  // pose_estimation.cpp:57-63
    for (int j = 0; j < m->n_models; j++)
        if (m->marker_id[j] == marker_id) return j;
    return -1;
}

}  // namespace

extern "C" {

void ctago_undistort_points(const ctag_camera* cam, int n, const float* uv, int with_P, double* out) {
    const Cam c = make_cam(cam);
    for (int i = 0; i < n; i++) {
        double x, y;
        undistort_normalised(c, (double)uv[2 * i], (double)uv[2 * i + 1], x, y);
        if (with_P) {
            x = c.fx * x + c.cx;
            y = c.fy * y + c.cy;
        }
        out[2 * i] = x;
        out[2 * i + 1] = y;
    }
}

int ctago_solve_pnp_epnp(const ctag_camera* cam, int n, const float* obj, const float* img, double* rvec, double* tvec) {
    if (n < 4) return CTAG_POSE_TOO_FEW;
    const Cam c = make_cam(cam);
    Epnp e;
    e.n = n;
    e.cam = c;
    e.pws.resize(3 * n);
    e.us.resize(2 * n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < 3; j++) e.pws[3 * i + j] = (double)obj[3 * i + j];
        double x, y;
        undistort_normalised(c, (double)img[2 * i], (double)img[2 * i + 1], x, y);
        // undistortPoints writes CV_32FC2; epnp::init_points maps back with fu, uc (calib3d/src/epnp.h)
        e.us[2 * i] = (double)(float)x * c.fx + c.cx;
        e.us[2 * i + 1] = (double)(float)y * c.fy + c.cy;
    }
    double R[9], t[3];
    if (!e.compute_pose(R, t)) return CTAG_POSE_DEGENERATE;
    ctl::rodrigues_from_matrix(R, rvec);
    for (int i = 0; i < 3; i++) tvec[i] = t[i];
    for (int i = 0; i < 3; i++)
        if (!ctl::finite64(rvec[i])) return CTAG_POSE_DEGENERATE;
    return CTAG_POSE_OK;
}

int ctago_pose_ba(const ctag_camera* cam, int n, const float* obj, const float* img, double* rvec, double* tvec, double* cost0,
                  double* cost) {
    const Cam c = make_cam(cam);
    std::vector<double> X(3 * n), obs(2 * n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < 3; j++) X[3 * i + j] = (double)obj[3 * i + j];
        double x, y;
        undistort_normalised(c, (double)img[2 * i], (double)img[2 * i + 1], x, y);
        // undistortPoints(imagePoints, imagePoints, K, dist, noArray(), K) into vector<Point2f> (pose_estimation.cpp:109)
        obs[2 * i] = (double)(float)(c.fx * x + c.cx);
        obs[2 * i + 1] = (double)(float)(c.fy * y + c.cy);
    }
    BA ba{n, c.fx, c.fy, c.cx, c.cy, X.data(), obs.data()};
    double x[6] = {rvec[0], rvec[1], rvec[2], tvec[0], tvec[1], tvec[2]};
    const int it = ba.solve(x, *cost0, *cost);
    for (int i = 0; i < 3; i++) {
        rvec[i] = x[i];
        tvec[i] = x[3 + i];
    }
    return it;
}

int ctago_build_correspondences(const ctag_frame_result* r, int marker, const ctag_model_view* model, int model_index, float* obj,
                                float* img, int* n_out) {
    const ctag_marker_rec& M = r->markers[marker];
    const int nf = M.n_features;
    const float* corners = model->corners + (size_t)model_index * model->model_size * 8 * 3;
    int n = 0;
    *n_out = 0;
    // a record whose marker points outside the frame's feature array (hand-built or corrupted input) is rejected, never read
    if (M.first_feature < 0 || nf < 0 || M.first_feature > CTAG_MAX_FEATURES - nf) return CTAG_POSE_BAD_POS;
    for (int j = 0; j < nf; j++) {
        const ctag_feature_rec& F = r->features[M.first_feature + j];
        const int d = F.id_left - F.id_right;
        const int ad = d < 0 ? -d : d;
        if (nf > 3) {  // pose_estimation.cpp:73-76
            if (j == 0 && (ad > 1 || F.id_right == -1)) continue;
            if (j == nf - 1 && (ad > 1 || F.id_right == -1)) continue;
        }
        if (j >= M.n_pos || F.pos < 0 || F.pos >= model->model_size) return CTAG_POSE_BAD_POS;
        const bool inner = ad < 3 && F.id_right != -1;  // :85
        if (n + (inner ? 8 : 4) > model->model_size * 8 || n + (inner ? 8 : 4) > CTAG_POSE_MAX_POINTS) return CTAG_POSE_BAD_POS;
        const int ks[8] = {0, 1, 4, 5, 2, 3, 6, 7};
        for (int q = 0; q < (inner ? 8 : 4); q++) {
            const int k = ks[q];
            img[2 * n] = F.corners[2 * k];
            img[2 * n + 1] = F.corners[2 * k + 1];
            for (int c = 0; c < 3; c++) obj[3 * n + c] = corners[(F.pos * 8 + k) * 3 + c];
            n++;
        }
    }
    *n_out = n;
    return CTAG_POSE_OK;
}

// primitive probes for unit tests (tests/test_pose_cpu.py checks them against numpy.linalg)
// op 0: jacobi_eig_rr12  in a[144]            out w[12] (unsorted), v[144] (eigenvectors as columns)
// op 1: svd3             in a[9]              out U[9], s[3], V[9]
// op 2: qr_solve<6,4>    in A[24], b[6]       out x[4]
// op 3: chol6_solve      in H[36], g[6]       out x[6], ok
// op 4: angle_axis_rot   in r[3]              out R[9], dR[27]
// op 5: rodrigues        in R[9]              out r[3]
// op 6: jacobi_eig<3>    in a[9]              out w[3], v[9]
void ctago_linalg_probe(int op, const double* in, double* out) {
    if (op == 0) {
        double a[144];
        for (int i = 0; i < 144; i++) a[i] = in[i];
        ctl::jacobi_eig_rr12(a, out + 12, out);
    } else if (op == 1) {
        ctl::svd3(in, out, out + 9, out + 12);
    } else if (op == 2) {
        double A[24], b[6];
        for (int i = 0; i < 24; i++) A[i] = in[i];
        for (int i = 0; i < 6; i++) b[i] = in[24 + i];
        ctl::qr_solve<6, 4>(A, b, out);
    } else if (op == 3) {
        double H[36];
        for (int i = 0; i < 36; i++) H[i] = in[i];
        out[6] = ctl::chol6_solve(H, in + 36, out) ? 1.0 : 0.0;
    } else if (op == 4) {
        ctl::angle_axis_rot(in, out, out + 9);
    } else if (op == 5) {
        ctl::rodrigues_from_matrix(in, out);
    } else if (op == 6) {
        double a[9];
        for (int i = 0; i < 9; i++) a[i] = in[i];
        ctl::jacobi_eig<3>(a, out + 3, out);
    }
}

int ctago_pose_frame(const ctag_frame_result* r, const ctag_model_view* model, const ctag_camera* cam, int frame_index,
                     ctag_pose_rec* out) {
    if (r->status != CTAG_OK) return 0;
    const int n_markers = std::min(std::max(r->n_markers, 0), CTAG_MAX_MARKERS);  // as k_pose_offsets clamps it
    for (int m = 0; m < n_markers; m++) {
        ctag_pose_rec& P = out[m];
        std::memset(&P, 0, sizeof(P));
        P.frame = frame_index;
        P.marker = m;
        P.model_index = find_model(model, r->markers[m].marker_id);
        if (P.model_index < 0) {
            P.status = CTAG_POSE_NO_MODEL;
            continue;
        }
        float obj[3 * CTAG_POSE_MAX_POINTS], img[2 * CTAG_POSE_MAX_POINTS];
        int n = 0;
        P.status = ctago_build_correspondences(r, m, model, P.model_index, obj, img, &n);
        if (P.status != CTAG_POSE_OK) continue;
        P.n_points = n;
        if (n < 4) {
            P.status = CTAG_POSE_TOO_FEW;
            continue;
        }
        P.status = ctago_solve_pnp_epnp(cam, n, obj, img, P.rvec0, P.tvec0);
        if (P.status != CTAG_POSE_OK) continue;
        for (int i = 0; i < 3; i++) {
            P.rvec[i] = P.rvec0[i];
            P.tvec[i] = P.tvec0[i];
        }
        P.iterations = ctago_pose_ba(cam, n, obj, img, P.rvec, P.tvec, &P.cost0, &P.cost);
    }
    return n_markers;
}

}  // extern "C"
