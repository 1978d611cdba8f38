// ctag_linalg.h -- small deterministic dense linear algebra shared by the pose kernel and the CPU oracle, in the same
// spirit as ctag_math.h: every routine is built from IEEE-754 +,-,*,/ and sqrt in a fixed evaluation order, so that
// gcc (x86-64, -ffp-contract=off) and hipcc (gfx950, -ffp-contract=off) produce the same bits.  No algorithm of the
// reference lives here -- only the primitives OpenCV/Ceres call underneath (cvSVD of symmetric matrices -> cyclic
// Jacobi, cv::SVD of a 3x3 -> one-sided Jacobi, least squares -> Householder QR, Ceres' dense normal equations ->
// Cholesky).
#pragma once
#include "ctag_math.h"

namespace ctl {

// Cyclic Jacobi eigen-decomposition of a symmetric N x N matrix.  a (row-major, full storage) is destroyed, v receives
// the eigenvectors as COLUMNS, w the eigenvalues (unsorted).  Rotation formulas after Rutishauser / Numerical Recipes.
struct JacobiRot {
    double s, tau, h;
    bool rotate, zero;
};
// decision and rotation parameters for the pair (p,q) in sweep `sweep` -- used by the serial routine below and by the
// lane-parallel 12x12 version in k_pose.hip, so both take identical decisions
CTM_HD JacobiRot jacobi_rot(double app, double aqq, double apq, int sweep) {
    JacobiRot r;
    r.s = 0.0;
    r.tau = 0.0;
    r.h = 0.0;
    r.rotate = false;
    r.zero = false;
    const double g = 100.0 * ctm::fabs64(apq);
    if (sweep > 3 && ctm::fabs64(app) + g == ctm::fabs64(app) && ctm::fabs64(aqq) + g == ctm::fabs64(aqq)) {
        r.zero = true;
        return r;
    }
    if (apq == 0.0) return r;
    const double h = aqq - app;
    double t;
    if (ctm::fabs64(h) + g == ctm::fabs64(h)) {
        t = apq / h;
    } else {
        // t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)) with theta = h / (2 apq), written without the division that
        // forms theta: t = 2 apq / (h + sgn(h) sqrt(h^2 + 4 apq^2)) -- one square root and one division in sequence
        // instead of division, square root, division (the chain is what a rotation costs on the GPU)
        const double rad = ctm::sqrt64(h * h + 4.0 * (apq * apq));
        t = (2.0 * apq) / (h < 0.0 ? h - rad : h + rad);
    }
    // c = 1 / sqrt(1 + t^2), s = t c, tau = s / (1 + c) = t / (sqrt(1 + t^2) + 1): the two divisions are independent
    const double rt = ctm::sqrt64(1.0 + t * t);
    r.s = t / rt;
    r.tau = t / (rt + 1.0);
    r.h = t * apq;
    r.rotate = true;
    return r;
}
CTM_HD void jacobi_apply(double& x, double& y, double s, double tau) {
    const double g = x, h = y;
    x = g - s * (h + g * tau);
    y = h + s * (g - h * tau);
}

template <int N>
CTM_HD void jacobi_eig(double* a, double* v, double* w) {
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) v[i * N + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double sm = 0.0;
        for (int p = 0; p < N - 1; p++)
            for (int q = p + 1; q < N; q++) sm += ctm::fabs64(a[p * N + q]);
        if (sm == 0.0) break;
        for (int p = 0; p < N - 1; p++) {
            for (int q = p + 1; q < N; q++) {
                const JacobiRot r = jacobi_rot(a[p * N + p], a[q * N + q], a[p * N + q], sweep);
                if (r.zero) {
                    a[p * N + q] = 0.0;
                    a[q * N + p] = 0.0;
                    continue;
                }
                if (!r.rotate) continue;
                a[p * N + p] -= r.h;
                a[q * N + q] += r.h;
                a[p * N + q] = 0.0;
                a[q * N + p] = 0.0;
                for (int k = 0; k < N; k++) {
                    if (k != p && k != q) {
                        double x = a[k * N + p], y = a[k * N + q];
                        jacobi_apply(x, y, r.s, r.tau);
                        a[k * N + p] = x;
                        a[p * N + k] = x;
                        a[k * N + q] = y;
                        a[q * N + k] = y;
                    }
                    double vx = v[k * N + p], vy = v[k * N + q];
                    jacobi_apply(vx, vy, r.s, r.tau);
                    v[k * N + p] = vx;
                    v[k * N + q] = vy;
                }
            }
        }
    }
    for (int i = 0; i < N; i++) w[i] = a[i * N + i];
}

// Round-robin ("circle method") pair ordering for N = 12: 11 rounds of 6 disjoint pairs cover every pair once.  Pairs of
// one round touch disjoint rows/columns, so a round's six rotations can be applied side by side (k_pose.hip) and still
// equal this sequential sweep bit for bit: rotation j's parameters read only (a_pp, a_qq, a_pq) of its own pair, which
// no other rotation of the round modifies.
CTM_HD void rr12_pair(int round, int slot, int& p, int& q) {
    int a, b;
    if (slot == 0) {
        a = 11;
        b = round;
    } else {
        a = (round + slot) % 11;
        b = (round + 11 - slot) % 11;
    }
    p = a < b ? a : b;
    q = a < b ? b : a;
}

// jacobi_eig<12> with the round-robin ordering (same rotation formulas, same stopping rule)
CTM_HD void jacobi_eig_rr12(double* a, double* v, double* w) {
    const int N = 12;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) v[i * N + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double sm = 0.0;
        for (int p = 0; p < N - 1; p++)
            for (int q = p + 1; q < N; q++) sm += ctm::fabs64(a[p * N + q]);
        if (sm == 0.0) break;
        for (int round = 0; round < 11; round++) {
            for (int slot = 0; slot < 6; slot++) {
                int p, q;
                rr12_pair(round, slot, p, q);
                const JacobiRot r = jacobi_rot(a[p * N + p], a[q * N + q], a[p * N + q], sweep);
                if (r.zero) {
                    a[p * N + q] = 0.0;
                    a[q * N + p] = 0.0;
                    continue;
                }
                if (!r.rotate) continue;
                a[p * N + p] -= r.h;
                a[q * N + q] += r.h;
                a[p * N + q] = 0.0;
                a[q * N + p] = 0.0;
                for (int k = 0; k < N; k++) {
                    if (k != p && k != q) {
                        double x = a[k * N + p], y = a[k * N + q];
                        jacobi_apply(x, y, r.s, r.tau);
                        a[k * N + p] = x;
                        a[p * N + k] = x;
                        a[k * N + q] = y;
                        a[q * N + k] = y;
                    }
                    double vx = v[k * N + p], vy = v[k * N + q];
                    jacobi_apply(vx, vy, r.s, r.tau);
                    v[k * N + p] = vx;
                    v[k * N + q] = vy;
                }
            }
        }
    }
    for (int i = 0; i < N; i++) w[i] = a[i * N + i];
}

// order[] = indices of w sorted by DESCENDING value (stable: ties keep index order) -- the order cvSVD reports
template <int N>
CTM_HD void sort_desc(const double* w, int* order) {
    for (int i = 0; i < N; i++) order[i] = i;
    for (int i = 1; i < N; i++) {
        const int k = order[i];
        int j = i - 1;
        while (j >= 0 && w[order[j]] < w[k]) {
            order[j + 1] = order[j];
            j--;
        }
        order[j + 1] = k;
    }
}

// One-sided (Hestenes) Jacobi SVD of a general 3x3 matrix A (row-major): A = U diag(s) V^T, s descending, U and V
// row-major with singular vectors as COLUMNS.  Columns of U belonging to a vanishing singular value are completed
// by cross products (right-handed with the others).
CTM_HD void svd3(const double* A, double* U, double* s, double* V) {
    double W[9];
    for (int i = 0; i < 9; i++) {
        W[i] = A[i];
        V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    const double eps = 2.220446049250313e-15;  // 10 * DBL_EPSILON
    for (int iter = 0; iter < 40; iter++) {
        bool changed = false;
        for (int i = 0; i < 2; i++) {
            for (int j = i + 1; j < 3; j++) {
                double a = 0.0, b = 0.0, p = 0.0;
                for (int k = 0; k < 3; k++) {
                    a += W[k * 3 + i] * W[k * 3 + i];
                    b += W[k * 3 + j] * W[k * 3 + j];
                    p += W[k * 3 + i] * W[k * 3 + j];
                }
                if (ctm::fabs64(p) <= eps * ctm::sqrt64(a * b)) continue;
                p *= 2.0;
                const double beta = a - b, gamma = ctm::sqrt64(p * p + beta * beta);
                double c, sn;
                if (beta < 0.0) {
                    const double delta = (gamma - beta) * 0.5;
                    sn = ctm::sqrt64(delta / gamma);
                    c = p / (gamma * sn * 2.0);
                } else {
                    c = ctm::sqrt64((gamma + beta) / (gamma * 2.0));
                    sn = p / (gamma * c * 2.0);
                }
                for (int k = 0; k < 3; k++) {
                    const double x = W[k * 3 + i], y = W[k * 3 + j];
                    W[k * 3 + i] = c * x + sn * y;
                    W[k * 3 + j] = c * y - sn * x;
                    const double vx = V[k * 3 + i], vy = V[k * 3 + j];
                    V[k * 3 + i] = c * vx + sn * vy;
                    V[k * 3 + j] = c * vy - sn * vx;
                }
                changed = true;
            }
        }
        if (!changed) break;
    }
    double sv[3];
    for (int i = 0; i < 3; i++) sv[i] = ctm::sqrt64(W[i] * W[i] + W[3 + i] * W[3 + i] + W[6 + i] * W[6 + i]);
    int ord[3];
    sort_desc<3>(sv, ord);
    double Wc[9], Vc[9];
    for (int c = 0; c < 3; c++) {
        s[c] = sv[ord[c]];
        for (int k = 0; k < 3; k++) {
            Wc[k * 3 + c] = W[k * 3 + ord[c]];
            Vc[k * 3 + c] = V[k * 3 + ord[c]];
        }
    }
    for (int i = 0; i < 9; i++) V[i] = Vc[i];
    const double tiny = s[0] * 1e-14;
    int ngood = 0;
    for (int c = 0; c < 3; c++) {
        if (s[c] > tiny && s[c] > 0.0) {
            const double inv = 1.0 / s[c];
            for (int k = 0; k < 3; k++) U[k * 3 + c] = Wc[k * 3 + c] * inv;
            ngood++;
        } else {
            break;
        }
    }
    if (ngood == 0) {
        for (int i = 0; i < 9; i++) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else if (ngood == 1) {
        // any unit vector orthogonal to u0, then the cross product
        const double ux = U[0], uy = U[3], uz = U[6];
        double ex = 1.0, ey = 0.0, ez = 0.0;
        if (ctm::fabs64(ux) > ctm::fabs64(uy) && ctm::fabs64(ux) > ctm::fabs64(uz)) {
            ex = 0.0;
            ey = 1.0;
        }
        const double d = ex * ux + ey * uy + ez * uz;
        double bx = ex - d * ux, by = ey - d * uy, bz = ez - d * uz;
        const double bn = ctm::sqrt64(bx * bx + by * by + bz * bz);
        bx /= bn;
        by /= bn;
        bz /= bn;
        U[1] = bx;
        U[4] = by;
        U[7] = bz;
        ngood = 2;
    }
    if (ngood == 2) {
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

// Least squares min |A x - b| for an M x N matrix (row-major, M >= N, destroyed) by Householder QR without pivoting.
// A (numerically) rank-deficient column leaves non-finite values in x; callers compare results with '<' so that such
// a candidate is never selected.
template <int M, int N>
CTM_HD void qr_solve(double* A, double* b, double* x) {
    for (int k = 0; k < N; k++) {
        double nrm = 0.0;
        for (int i = k; i < M; i++) nrm += A[i * N + k] * A[i * N + k];
        nrm = ctm::sqrt64(nrm);
        const double akk = A[k * N + k];
        const double alpha = akk > 0.0 ? -nrm : nrm;
        // v = column - alpha e_k
        const double v0 = akk - alpha;
        double vnorm2 = v0 * v0;
        for (int i = k + 1; i < M; i++) vnorm2 += A[i * N + k] * A[i * N + k];
        if (vnorm2 > 0.0) {
            for (int j = k + 1; j < N; j++) {
                double d = v0 * A[k * N + j];
                for (int i = k + 1; i < M; i++) d += A[i * N + k] * A[i * N + j];
                const double f = 2.0 * d / vnorm2;
                A[k * N + j] -= f * v0;
                for (int i = k + 1; i < M; i++) A[i * N + j] -= f * A[i * N + k];
            }
            double d = v0 * b[k];
            for (int i = k + 1; i < M; i++) d += A[i * N + k] * b[i];
            const double f = 2.0 * d / vnorm2;
            b[k] -= f * v0;
            for (int i = k + 1; i < M; i++) b[i] -= f * A[i * N + k];
        }
        A[k * N + k] = alpha;
    }
    for (int k = N - 1; k >= 0; k--) {
        double sacc = b[k];
        for (int j = k + 1; j < N; j++) sacc -= A[k * N + j] * x[j];
        x[k] = sacc / A[k * N + k];
    }
}

// Solves H x = g for a symmetric positive definite 6x6 H (row-major, destroyed) by Cholesky; false if not SPD.
CTM_HD bool chol6_solve(double* H, const double* g, double* x) {
    const int N = 6;
    for (int j = 0; j < N; j++) {
        double d = H[j * N + j];
        for (int k = 0; k < j; k++) d -= H[j * N + k] * H[j * N + k];
        if (!(d > 0.0)) return false;
        d = ctm::sqrt64(d);
        H[j * N + j] = d;
        for (int i = j + 1; i < N; i++) {
            double sacc = H[i * N + j];
            for (int k = 0; k < j; k++) sacc -= H[i * N + k] * H[j * N + k];
            H[i * N + j] = sacc / d;
        }
    }
    double y[6];
    for (int i = 0; i < N; i++) {
        double sacc = g[i];
        for (int k = 0; k < i; k++) sacc -= H[i * N + k] * y[k];
        y[i] = sacc / H[i * N + i];
    }
    for (int i = N - 1; i >= 0; i--) {
        double sacc = y[i];
        for (int k = i + 1; k < N; k++) sacc -= H[k * N + i] * x[k];
        x[i] = sacc / H[i * N + i];
    }
    return true;
}

CTM_HD bool finite64(double x) { return (ctm::f64_to_bits(x) & 0x7ff0000000000000ULL) != 0x7ff0000000000000ULL; }

// 3x3 inverse by cofactors; false when the determinant is 0 or non-finite
CTM_HD bool inv3(const double* m, double* o) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (det == 0.0 || !finite64(det)) return false;
    const double id = 1.0 / det;
    o[0] = c00 * id;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id;
    o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id;
    o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

// Rotation matrix of an angle-axis vector and its three partial derivatives, row-major.  Same function Ceres'
// AngleAxisRotatePoint evaluates (rotation.h: Rodrigues' formula for theta^2 > DBL_EPSILON, the first-order form
// p + r x p below that); the derivatives are the exact ones its automatic differentiation produces (up to rounding).
CTM_HD void angle_axis_rot(const double* r, double* R, double* dR /* [3][9] or nullptr */) {
    const double theta2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    if (theta2 > 2.220446049250313e-16) {
        const double theta = ctm::sqrt64(theta2);
        const double c = ctm::cos64(theta), s = ctm::sin64(theta);
        const double it = 1.0 / theta;
        const double w[3] = {r[0] * it, r[1] * it, r[2] * it};
        const double oc = 1.0 - c;
        R[0] = c + oc * w[0] * w[0];
        R[1] = oc * w[0] * w[1] - s * w[2];
        R[2] = oc * w[0] * w[2] + s * w[1];
        R[3] = oc * w[1] * w[0] + s * w[2];
        R[4] = c + oc * w[1] * w[1];
        R[5] = oc * w[1] * w[2] - s * w[0];
        R[6] = oc * w[2] * w[0] - s * w[1];
        R[7] = oc * w[2] * w[1] + s * w[0];
        R[8] = c + oc * w[2] * w[2];
        if (dR) {
            for (int k = 0; k < 3; k++) {
                double dw[3];
                for (int i = 0; i < 3; i++) dw[i] = ((i == k ? 1.0 : 0.0) - w[i] * w[k]) * it;
                const double wk = w[k];
                double* D = dR + 9 * k;
                // d/dr_k [ c I + s [w]x + (1-c) w w^T ]
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++)
                        D[i * 3 + j] = (i == j ? -s * wk : 0.0) + s * wk * w[i] * w[j] + oc * (dw[i] * w[j] + w[i] * dw[j]);
                // + c wk [w]x + s [dw]x
                const double a0 = c * wk * w[0] + s * dw[0], a1 = c * wk * w[1] + s * dw[1], a2 = c * wk * w[2] + s * dw[2];
                D[1] -= a2;
                D[2] += a1;
                D[3] += a2;
                D[5] -= a0;
                D[6] -= a1;
                D[7] += a0;
            }
        }
    } else {
        R[0] = 1.0;
        R[1] = -r[2];
        R[2] = r[1];
        R[3] = r[2];
        R[4] = 1.0;
        R[5] = -r[0];
        R[6] = -r[1];
        R[7] = r[0];
        R[8] = 1.0;
        if (dR) {
            for (int i = 0; i < 27; i++) dR[i] = 0.0;
            dR[0 * 9 + 5] = -1.0;
            dR[0 * 9 + 7] = 1.0;
            dR[1 * 9 + 2] = 1.0;
            dR[1 * 9 + 6] = -1.0;
            dR[2 * 9 + 1] = -1.0;
            dR[2 * 9 + 3] = 1.0;
        }
    }
}

// cv::Rodrigues, matrix -> vector branch (OpenCV 4.5.3 calib3d; the preceding SVD re-orthonormalisation is left out:
// the callers pass U V^T products that are orthonormal to rounding).
CTM_HD void rodrigues_from_matrix(const double* R, double* r) {
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = ctm::sqrt64((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    double theta = ctm::acos64(c);
    if (s < 1e-5) {
        if (c > 0.0) {
            rx = ry = rz = 0.0;
        } else {
            double t = (R[0] + 1.0) * 0.5;
            rx = ctm::sqrt64(t > 0.0 ? t : 0.0);
            t = (R[4] + 1.0) * 0.5;
            ry = ctm::sqrt64(t > 0.0 ? t : 0.0) * (R[1] < 0.0 ? -1.0 : 1.0);
            t = (R[8] + 1.0) * 0.5;
            rz = ctm::sqrt64(t > 0.0 ? t : 0.0) * (R[2] < 0.0 ? -1.0 : 1.0);
            if (ctm::fabs64(rx) < ctm::fabs64(ry) && ctm::fabs64(rx) < ctm::fabs64(rz) && (R[5] > 0.0) != (ry * rz > 0.0)) rz = -rz;
            theta /= ctm::sqrt64(rx * rx + ry * ry + rz * rz);
            rx *= theta;
            ry *= theta;
            rz *= theta;
        }
    } else {
        const double vth = theta / (2.0 * s);
        rx *= vth;
        ry *= vth;
        rz *= vth;
    }
    r[0] = rx;
    r[1] = ry;
    r[2] = rz;
}

}  // namespace ctl
