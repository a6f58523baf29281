// Batched cuboid PnP on device, float64, one lane per detection.
//
// Replaces the per-detection host loop `pnp_shell` -> `CuboidPNPSolver.solve_pnp` ->
// `cv2.solvePnPGeneric(flags=SOLVEPNP_ITERATIVE)` + `cv2.projectPoints`
// (/root/reference/src/lib/utils/pnp/cuboid_pnp_shell.py:11-24, cuboid_pnp_solver.py:141-239,
// base_detector.py:547-654).  OpenCV's arithmetic is not in the reference tree (un-vendored
// opencv-python>=4.5.3.56); this kernel restates calib3d's published SOLVEPNP_ITERATIVE for >= 6
// non-planar points exactly as oracle/pnp.py does:
//   normalise by K -> DLT: smallest eigenvector of L^T L (12x12; shifted inverse iteration in registers) -> det sign fix ->
//   polar factor R = U V^T, t *= |R| / |R_raw| -> Rodrigues -> Levenberg-Marquardt (<= 20 iterations,
//   eps = FLT_EPSILON, lambda = 10^k from k = -3, diag(JtJ) *= 1 + lambda) on pixel reprojection error.
// The cuboid model, point filtering (x or y < -5000 dropped; point i belongs to vertex i / (n/8)),
// z < 0 rejection, OpenGL conversion and axis-angle quaternion follow the reference's Python.
//
// The work is ~10^5 float64 operations per detection and a few hundred detections per batch: it is
// latency-bound, not bandwidth- or MFMA-bound; one lane per detection keeps every solve independent
// and deterministic, and the whole batch is one launch instead of a Python loop.
#include "cp_common.h"

namespace {

constexpr double DBL_EPS = 2.220446049250313e-16;
constexpr double FLT_EPS = 1.1920928955078125e-07;

struct Cam { double fx, fy, cx, cy; };

__device__ void rodrigues(const double r[3], double R[9], double* J /*27 or null*/) {
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPS) {
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        if (J) {
            const double j0[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
            for (int i = 0; i < 27; ++i) J[i] = j0[i];
        }
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1.0 - c, it = 1.0 / theta;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    const double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int k = 0; k < 9; ++k) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * rx[k];
    if (!J) return;
    const double drrt[27] = {x + x, y, z, y, 0, 0, z, 0, 0, 0, x, 0, x, y + y, z, 0, z, 0, 0, 0, x, 0, 0, y, x, y, z + z};
    const double drx[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
    const double a[3] = {x, y, z};
    for (int i = 0; i < 3; ++i) {
        const double ri = a[i];
        const double a0 = -s * ri, a1 = (s - 2 * c1 * it) * ri, a2 = c1 * it, a3 = (c - s * it) * ri, a4 = s * it;
        for (int k = 0; k < 9; ++k)
            J[i * 9 + k] = a0 * ((k % 4 == 0) ? 1.0 : 0.0) + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * rx[k] +
                           a4 * drx[i * 9 + k];
    }
}

// polar factor U V^T of a 3x3 matrix with positive determinant (Newton iteration X <- (X + X^-T)/2)
__device__ void polar3(const double A[9], double R[9]) {
    double X[9];
    for (int i = 0; i < 9; ++i) X[i] = A[i];
    for (int it = 0; it < 60; ++it) {
        const double c00 = X[4] * X[8] - X[5] * X[7], c01 = X[5] * X[6] - X[3] * X[8], c02 = X[3] * X[7] - X[4] * X[6];
        const double c10 = X[2] * X[7] - X[1] * X[8], c11 = X[0] * X[8] - X[2] * X[6], c12 = X[1] * X[6] - X[0] * X[7];
        const double c20 = X[1] * X[5] - X[2] * X[4], c21 = X[2] * X[3] - X[0] * X[5], c22 = X[0] * X[4] - X[1] * X[3];
        const double det = X[0] * c00 + X[1] * c01 + X[2] * c02;
        const double id = 1.0 / det;
        // inverse-transpose = cofactor matrix / det
        const double T[9] = {c00 * id, c01 * id, c02 * id, c10 * id, c11 * id, c12 * id, c20 * id, c21 * id, c22 * id};
        double diff = 0;
        for (int i = 0; i < 9; ++i) {
            const double n = 0.5 * (X[i] + T[i]);
            diff += fabs(n - X[i]);
            X[i] = n;
        }
        if (diff < 1e-15) break;
    }
    for (int i = 0; i < 9; ++i) R[i] = X[i];
}

// cv::Rodrigues matrix -> vector for an orthonormal R
__device__ void rot_to_rvec(const double R[9], double r[3]) {
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1 ? 1 : (c < -1 ? -1 : c);
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t = (R[0] + 1) * 0.5;
        rx = sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5;
        ry = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1.0 : 1.0);
        t = (R[8] + 1) * 0.5;
        rz = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1.0 : 1.0);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && ((R[5] > 0) != (ry * rz > 0))) rz = -rz;
        theta /= sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * theta; r[1] = ry * theta; r[2] = rz * theta;
        return;
    }
    const double v = theta / (2 * s);
    r[0] = rx * v; r[1] = ry * v; r[2] = rz * v;
}

// Smallest eigenvector of a symmetric positive semi-definite 12x12 matrix (lower triangle, packed: element (i, j),
// j <= i, at i*(i+1)/2 + j), by shifted inverse iteration on a Cholesky factor held entirely in registers:
//   A + mu*I = L L^T (mu = 1e-13 * trace keeps the factorisation positive when the smallest eigenvalue is ~0, as it is for
//   exact correspondences, without moving the eigenvectors), then x <- normalise(L^-T L^-1 x) until the direction stops
//   changing.  ~300 FMAs for the factor + 160 per iteration, against ~10^5 strided global loads / stores for the cyclic
//   Jacobi sweep this replaces (which was 90 % of the kernel's time: 4.1 ms per 6400 detections).  The convergence
//   ratio is lambda_1 / lambda_2; the iteration cap only bites on (near-)degenerate point sets, whose pose is
//   ill-defined anyway and is refined by the Levenberg-Marquardt stage regardless.
#define TRI(i, j) ((i) * ((i) + 1) / 2 + (j))
__device__ void smallest_eigvec12(double* A /*78, destroyed*/, double out[12]) {
    constexpr int n = 12;
    double tr = 0;
#pragma unroll
    for (int i = 0; i < n; ++i) tr += A[TRI(i, i)];
    const double mu = 1e-13 * tr + 1e-300;
#pragma unroll
    for (int i = 0; i < n; ++i) A[TRI(i, i)] += mu;
    // in-place Cholesky, column by column (all indices are compile-time constants after unrolling -> registers)
    double dinv[n];
#pragma unroll
    for (int j = 0; j < n; ++j) {
        double d = A[TRI(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= A[TRI(j, k)] * A[TRI(j, k)];
        d = d > mu * 1e-3 ? d : mu * 1e-3;  // rounding can eat a ~0 pivot; keep the factor real
        const double l = sqrt(d);
        A[TRI(j, j)] = l;
        dinv[j] = 1.0 / l;
#pragma unroll
        for (int i = j + 1; i < n; ++i) {
            double v = A[TRI(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= A[TRI(i, k)] * A[TRI(j, k)];
            A[TRI(i, j)] = v * dinv[j];
        }
    }
    double x[n];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = 0.28867513459481287 * ((i & 1) ? 1.0 : 0.9) * ((i % 3 == 2) ? -1.0 : 1.0);  // generic start
    for (int it = 0; it < 400; ++it) {
        double y[n];
        // forward: L z = x
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double v = x[i];
#pragma unroll
            for (int k = 0; k < i; ++k) v -= A[TRI(i, k)] * y[k];
            y[i] = v * dinv[i];
        }
        // backward: L^T w = z
#pragma unroll
        for (int i = n - 1; i >= 0; --i) {
            double v = y[i];
#pragma unroll
            for (int k = i + 1; k < n; ++k) v -= A[TRI(k, i)] * y[k];
            y[i] = v * dinv[i];
        }
        double nn = 0, dot = 0;
#pragma unroll
        for (int i = 0; i < n; ++i) nn += y[i] * y[i];
        const double inv = 1.0 / sqrt(nn);
#pragma unroll
        for (int i = 0; i < n; ++i) { y[i] *= inv; dot += y[i] * x[i]; }
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = y[i];
        if (it >= 2 && 1.0 - fabs(dot) < 1e-16) break;
    }
#pragma unroll
    for (int i = 0; i < n; ++i) out[i] = x[i];
}

// solve 6x6 A x = b (Gaussian elimination, partial pivoting); A, b destroyed
__device__ void solve6(double A[36], double b[6], double x[6]) {
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        for (int r = c + 1; r < 6; ++r)
            if (fabs(A[r * 6 + c]) > fabs(A[piv * 6 + c])) piv = r;
        if (piv != c) {
            for (int k = 0; k < 6; ++k) { const double t = A[c * 6 + k]; A[c * 6 + k] = A[piv * 6 + k]; A[piv * 6 + k] = t; }
            const double t = b[c]; b[c] = b[piv]; b[piv] = t;
        }
        const double d = A[c * 6 + c];
        if (d == 0.0) continue;
        for (int r = c + 1; r < 6; ++r) {
            const double f = A[r * 6 + c] / d;
            if (f == 0.0) continue;
            for (int k = c; k < 6; ++k) A[r * 6 + k] -= f * A[c * 6 + k];
            b[r] -= f * b[c];
        }
    }
    for (int r = 5; r >= 0; --r) {
        double s = b[r];
        for (int k = r + 1; k < 6; ++k) s -= A[r * 6 + k] * x[k];
        x[r] = (A[r * 6 + r] != 0.0) ? s / A[r * 6 + r] : 0.0;
    }
}

__device__ __forceinline__ void project1(const double R[9], const double t[3], const Cam& cam, const double M[3],
                                         double& u, double& v, double& x, double& y, double& z) {
    const double X = R[0] * M[0] + R[1] * M[1] + R[2] * M[2] + t[0];
    const double Y = R[3] * M[0] + R[4] * M[1] + R[5] * M[2] + t[1];
    const double Z = R[6] * M[0] + R[7] * M[1] + R[8] * M[2] + t[2];
    z = (Z != 0.0) ? 1.0 / Z : 1.0;
    x = X * z;
    y = Y * z;
    u = cam.fx * x + cam.cx;
    v = cam.fy * y + cam.cy;
}

__device__ void axis_angle_quat(const double r[3], double q[4]) {
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    double ax = r[0] / theta, ay = r[1] / theta, az = r[2] / theta;
    const double n = sqrt(ax * ax + ay * ay + az * az);
    ax /= n; ay /= n; az /= n;
    const double h = theta * 0.5, sh = sin(h);
    q[0] = sh * ax; q[1] = sh * ay; q[2] = sh * az; q[3] = cos(h);
}

// pts [N][npts][2] float (npts = 8 or 16), scale [N][3] float, cam [N][4] double (fx, fy, cx, cy)
// out [N][CP_PNP_STRIDE] double;  scratch: unused since the eigen-solver moved into registers (kept in the ABI).
__global__ __launch_bounds__(64) void pnp_kernel(const float* __restrict__ pts, const float* __restrict__ scale,
                                                 const double* __restrict__ camp, int N, int npts,
                                                 double* __restrict__ out, double* __restrict__ scratch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double* o = out + (size_t)i * CP_PNP_STRIDE;
    for (int k = 0; k < CP_PNP_STRIDE; ++k) o[k] = 0.0;
    const Cam cam = {camp[i * 4 + 0], camp[i * 4 + 1], camp[i * 4 + 2], camp[i * 4 + 3]};
    // cuboid: size = scale / scale[1]  (cuboid_pnp_shell.py:12), vertices cuboid_objectron.py:97-109
    const double s1 = (double)scale[i * 3 + 1];
    const double hw = 0.5 * ((double)scale[i * 3 + 0] / s1), hh = 0.5 * ((double)scale[i * 3 + 1] / s1),
                 hd = 0.5 * ((double)scale[i * 3 + 2] / s1);
    double V3[8][3];
    for (int v = 0; v < 8; ++v) {
        V3[v][0] = (v & 4) ? hw : -hw;
        V3[v][1] = (v & 2) ? hh : -hh;
        V3[v][2] = (v & 1) ? hd : -hd;
    }
    const int per = npts / 8;
    const float* P = pts + (size_t)i * npts * 2;
    // valid points
    int nv = 0;
    double Mc[3] = {0, 0, 0};
    for (int k = 0; k < npts; ++k) {
        if (P[2 * k] < -5000.f || P[2 * k + 1] < -5000.f) continue;
        ++nv;
        for (int d = 0; d < 3; ++d) Mc[d] += V3[k / per][d];
    }
    o[35] = nv;
    if (nv < 4) { o[0] = -1; return; }
    if (nv < 6) { o[0] = -2; return; }  // EPnP branch of the reference (cuboid_pnp_solver.py:162-163): not restated
    for (int d = 0; d < 3; ++d) Mc[d] /= nv;
    // planarity test of cvFindExtrinsicCameraParams2: second/third singular value of the 3x3 scatter
    {
        double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < npts; ++k) {
            if (P[2 * k] < -5000.f || P[2 * k + 1] < -5000.f) continue;
            double dd[3];
            for (int d = 0; d < 3; ++d) dd[d] = V3[k / per][d] - Mc[d];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) S[a * 3 + b] += dd[a] * dd[b];
        }
        // eigenvalues of symmetric 3x3 by Jacobi
        for (int sw = 0; sw < 20; ++sw)
            for (int p = 0; p < 2; ++p)
                for (int q = p + 1; q < 3; ++q) {
                    const double apq = S[p * 3 + q];
                    if (fabs(apq) < 1e-300) continue;
                    const double tau = (S[q * 3 + q] - S[p * 3 + p]) / (2 * apq);
                    const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
                    const double c = 1 / sqrt(1 + t * t), s = t * c;
                    for (int k = 0; k < 3; ++k) { const double a = S[k * 3 + p], b = S[k * 3 + q]; S[k * 3 + p] = c * a - s * b; S[k * 3 + q] = s * a + c * b; }
                    for (int k = 0; k < 3; ++k) { const double a = S[p * 3 + k], b = S[q * 3 + k]; S[p * 3 + k] = c * a - s * b; S[q * 3 + k] = s * a + c * b; }
                }
        double e0 = S[0], e1 = S[4], e2 = S[8], tmp;
        if (e0 < e1) { tmp = e0; e0 = e1; e1 = tmp; }
        if (e1 < e2) { tmp = e1; e1 = e2; e2 = tmp; }
        if (e0 < e1) { tmp = e0; e0 = e1; e1 = tmp; }
        if (e2 / e1 < 1e-3) { o[0] = -3; return; }  // planar: homography branch not restated
    }
    // ---- DLT ----  L^T L accumulated as a packed lower triangle in registers
    double As[78];
#pragma unroll
    for (int k = 0; k < 78; ++k) As[k] = 0.0;
    for (int k = 0; k < npts; ++k) {
        if (P[2 * k] < -5000.f || P[2 * k + 1] < -5000.f) continue;
        const double* M = V3[k / per];
        const double x = -((double)P[2 * k] - cam.cx) / cam.fx, y = -((double)P[2 * k + 1] - cam.cy) / cam.fy;
        const double r0[12] = {M[0], M[1], M[2], 1, 0, 0, 0, 0, x * M[0], x * M[1], x * M[2], x};
        const double r1[12] = {0, 0, 0, 0, M[0], M[1], M[2], 1, y * M[0], y * M[1], y * M[2], y};
#pragma unroll
        for (int a = 0; a < 12; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) As[TRI(a, b)] += r0[a] * r0[b] + r1[a] * r1[b];
    }
    double ev[12];
    smallest_eigvec12(As, ev);
    double RR[9] = {ev[0], ev[1], ev[2], ev[4], ev[5], ev[6], ev[8], ev[9], ev[10]};
    double tt[3] = {ev[3], ev[7], ev[11]};
    const double det = RR[0] * (RR[4] * RR[8] - RR[5] * RR[7]) - RR[1] * (RR[3] * RR[8] - RR[5] * RR[6]) +
                       RR[2] * (RR[3] * RR[7] - RR[4] * RR[6]);
    if (det < 0) {
        for (int k = 0; k < 9; ++k) RR[k] = -RR[k];
        for (int k = 0; k < 3; ++k) tt[k] = -tt[k];
    }
    double sc = 0;
    for (int k = 0; k < 9; ++k) sc += RR[k] * RR[k];
    sc = sqrt(sc);
    if (!(sc > DBL_EPS)) { o[0] = 0; return; }
    double R0[9];
    polar3(RR, R0);
    const double f = sqrt(3.0) / sc;  // |R|_F of an orthonormal matrix is sqrt(3)
    double param[6], prev[6];
    rot_to_rvec(R0, param);
    param[3] = tt[0] * f; param[4] = tt[1] * f; param[5] = tt[2] * f;

    // ---- Levenberg-Marquardt (CvLevMarq) ----
    int lambda_lg10 = -3, iters = 0;
    double prev_err = 0, JtJ[36], JtE[6];
    bool calc_j = true;
    for (int guard = 0; guard < 2000; ++guard) {
        double R[9], dR[27];
        if (calc_j) {
            rodrigues(param, R, dR);
            for (int k = 0; k < 36; ++k) JtJ[k] = 0;
            for (int k = 0; k < 6; ++k) JtE[k] = 0;
            double e2 = 0;
            for (int k = 0; k < npts; ++k) {
                if (P[2 * k] < -5000.f || P[2 * k + 1] < -5000.f) continue;
                const double* M = V3[k / per];
                double u, v, x, y, z;
                project1(R, param + 3, cam, M, u, v, x, y, z);
                const double eu = u - (double)P[2 * k], evv = v - (double)P[2 * k + 1];
                e2 += eu * eu + evv * evv;
                double ju[6], jv[6];
                for (int j = 0; j < 3; ++j) {
                    const double dx0 = M[0] * dR[j * 9 + 0] + M[1] * dR[j * 9 + 1] + M[2] * dR[j * 9 + 2];
                    const double dy0 = M[0] * dR[j * 9 + 3] + M[1] * dR[j * 9 + 4] + M[2] * dR[j * 9 + 5];
                    const double dz0 = M[0] * dR[j * 9 + 6] + M[1] * dR[j * 9 + 7] + M[2] * dR[j * 9 + 8];
                    ju[j] = cam.fx * z * (dx0 - x * dz0);
                    jv[j] = cam.fy * z * (dy0 - y * dz0);
                }
                ju[3] = cam.fx * z; ju[4] = 0; ju[5] = -cam.fx * x * z;
                jv[3] = 0; jv[4] = cam.fy * z; jv[5] = -cam.fy * y * z;
                for (int a = 0; a < 6; ++a) {
                    JtE[a] += ju[a] * eu + jv[a] * evv;
                    for (int b = 0; b < 6; ++b) JtJ[a * 6 + b] += ju[a] * ju[b] + jv[a] * jv[b];
                }
            }
            for (int k = 0; k < 6; ++k) prev[k] = param[k];
            if (iters == 0) prev_err = sqrt(e2);
            calc_j = false;
        } else {
            rodrigues(param, R, nullptr);
            double e2 = 0;
            for (int k = 0; k < npts; ++k) {
                if (P[2 * k] < -5000.f || P[2 * k + 1] < -5000.f) continue;
                double u, v, x, y, z;
                project1(R, param + 3, cam, V3[k / per], u, v, x, y, z);
                const double eu = u - (double)P[2 * k], evv = v - (double)P[2 * k + 1];
                e2 += eu * eu + evv * evv;
            }
            const double err = sqrt(e2);
            bool retry = false;
            if (err > prev_err) {
                ++lambda_lg10;
                if (lambda_lg10 <= 16) retry = true;
            }
            if (!retry) {
                lambda_lg10 = lambda_lg10 - 1 < -16 ? -16 : lambda_lg10 - 1;
                ++iters;
                double dn = 0, pn = 0;
                for (int k = 0; k < 6; ++k) { dn += (param[k] - prev[k]) * (param[k] - prev[k]); pn += prev[k] * prev[k]; }
                const double rel = sqrt(dn) / (pn > 0 ? sqrt(pn) : 1.0);
                if (iters >= 20 || rel < FLT_EPS) break;
                prev_err = err;
                calc_j = true;
                continue;
            }
        }
        // step(): param = prev - solve(JtJ with diag *= 1 + lambda, JtErr)
        double Aq[36], bq[6], dx[6];
        const double lam = exp(lambda_lg10 * log(10.0));
        for (int k = 0; k < 36; ++k) Aq[k] = JtJ[k];
        for (int k = 0; k < 6; ++k) { Aq[k * 7] *= 1.0 + lam; bq[k] = JtE[k]; }
        solve6(Aq, bq, dx);
        for (int k = 0; k < 6; ++k) param[k] = prev[k] - dx[k];
    }
    // ---- outputs ----
    double R[9];
    rodrigues(param, R, nullptr);
    double e2 = 0;
    for (int k = 0; k < npts; ++k) {
        if (P[2 * k] < -5000.f || P[2 * k + 1] < -5000.f) continue;
        double u, v, x, y, z;
        project1(R, param + 3, cam, V3[k / per], u, v, x, y, z);
        e2 += (u - P[2 * k]) * (u - P[2 * k]) + (v - P[2 * k + 1]) * (v - P[2 * k + 1]);
    }
    for (int k = 0; k < 6; ++k) o[1 + k] = param[k];
    o[7] = sqrt(e2) / sqrt(2.0 * nv);
    for (int v = 0; v < 8; ++v) {
        double u, vv, x, y, z;
        project1(R, param + 3, cam, V3[v], u, vv, x, y, z);
        o[8 + 2 * v] = u;
        o[9 + 2 * v] = vv;
    }
    axis_angle_quat(param, o + 24);
    // OpenGL convention: M = [[0,1,0],[1,0,0],[0,0,-1]] applied on the left (cuboid_pnp_solver.py:179-196)
    const double Rg[9] = {R[3], R[4], R[5], R[0], R[1], R[2], -R[6], -R[7], -R[8]};
    o[28] = param[4]; o[29] = param[3]; o[30] = -param[5];
    double rg[3];
    rot_to_rvec(Rg, rg);
    axis_angle_quat(rg, o + 31);
    o[36] = iters;
    o[0] = (param[5] < 0) ? 2.0 : 1.0;  // 2: solved but behind the camera -> the reference drops it (:207-220)
}

}  // namespace

size_t cp_pnp_ws_bytes(int N) { return (size_t)N * 288 * sizeof(double) + 256; }

int cp_launch_pnp(hipStream_t s, const float* pts, const float* scale, const double* cam, int N, int npts, double* out,
                  void* ws) {
    if (N < 1) return CP_OK;
    if (npts != 8 && npts != 16) return CP_ERR_INVALID;
    hipLaunchKernelGGL(pnp_kernel, dim3((N + 63) / 64), dim3(64), 0, s, pts, scale, cam, N, npts, out, (double*)ws);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}
