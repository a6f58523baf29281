// The scalar numerical pieces of the PnP solve (pnp.hip), written so that the host compiler can build them too
// (tests/native/pnp_linalg_host.cpp, tests/test_pnp_linalg_cpu.py): Rodrigues' formula with its derivatives, the polar
// factor and matrix -> rotation vector of the DLT initialisation, the smallest eigenvector of the DLT / homography normal
// matrix and the 6 x 6 solve of a Levenberg-Marquardt step.  Included by pnp.hip inside its anonymous namespace.
#pragma once
#include <cmath>

#ifdef __HIPCC__
#define PNP_HD __host__ __device__
#define PNP_HD_INLINE __host__ __device__ __forceinline__
#else
#define PNP_HD static
#define PNP_HD_INLINE static inline
#endif

constexpr double DBL_EPS = 2.220446049250313e-16;
constexpr double FLT_EPS = 1.1920928955078125e-07;

// cv::Rodrigues vector -> matrix (+ the 27 derivatives dR/dr when WJ).  Every loop is unrolled and the outputs are
// references to fixed-size arrays: with a nullable pointer for J the caller's dR[27] stayed in scratch memory (a store ->
// load round trip per Jacobian of the Levenberg-Marquardt walk).
template <bool WJ>
PNP_HD_INLINE void rodrigues_t(const double* r, double (&R)[9], double (&J)[27]) {
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPS) {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        if (WJ) {
            const double j0[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 27; ++i) J[i] = j0[i];
        }
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1.0 - c, it = 1.0 / theta;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    const double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * rx[k];
    if (!WJ) return;
    const double drrt[27] = {x + x, y, z, y, 0, 0, z, 0, 0, 0, x, 0, x, y + y, z, 0, z, 0, 0, 0, x, 0, 0, y, x, y, z + z};
    const double drx[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
    const double a[3] = {x, y, z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double ri = a[i];
        const double a0 = -s * ri, a1 = (s - 2 * c1 * it) * ri, a2 = c1 * it, a3 = (c - s * it) * ri, a4 = s * it;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            J[i * 9 + k] = a0 * ((k % 4 == 0) ? 1.0 : 0.0) + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * rx[k] +
                           a4 * drx[i * 9 + k];
    }
}
PNP_HD_INLINE void rodrigues(const double* r, double (&R)[9], double (&J)[27]) { rodrigues_t<true>(r, R, J); }
PNP_HD_INLINE void rodrigues(const double* r, double (&R)[9], decltype(nullptr)) {
    double none[27];
    rodrigues_t<false>(r, R, none);
}

// polar factor U V^T of a 3x3 matrix with positive determinant (Newton iteration X <- (X + X^-T)/2)
PNP_HD void polar3(const double A[9], double R[9]) {
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
PNP_HD void rot_to_rvec(const double R[9], double r[3]) {
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

// Smallest eigenvector of a symmetric positive semi-definite n x n matrix (n = 12: DLT, n = 9: homography) (lower triangle, packed: element (i, j),
// j <= i, at i*(i+1)/2 + j), by shifted inverse iteration on a Cholesky factor held entirely in registers:
//   A + mu*I = L L^T (mu = 1e-13 * trace keeps the factorisation positive when the smallest eigenvalue is ~0, as it is for
//   exact correspondences, without moving the eigenvectors), then x <- normalise(L^-T L^-1 x) until the direction stops
//   changing.  ~300 FMAs for the factor + 160 per iteration, against ~10^5 strided global loads / stores for the cyclic
//   Jacobi sweep this replaces (which was 90 % of the kernel's time: 4.1 ms per 6400 detections).  The convergence
//   ratio is lambda_1 / lambda_2; slow walks end with a Rayleigh-Ritz step over the last two iterates (below).
#define TRI(i, j) ((i) * ((i) + 1) / 2 + (j))
template <int n>
PNP_HD void smallest_eigvec(double* A /*n (n + 1) / 2, destroyed*/, double* out /*n*/) {
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
    // one application of (A + mu I)^-1: y = L^-T L^-1 x
    auto apply_inv = [&](const double (&xin)[n], double (&y)[n]) {
        // forward: L z = x
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double v = xin[i];
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
    };
    // Plain inverse iteration converges like (lambda_1 / lambda_2)^k: a handful of steps on most point sets, hundreds when
    // the two smallest eigenvalues are close -- and a batch waits for its slowest detection (64 well-posed detections:
    // 314 us against 60 for one, because one of them walked to the old cap of 400 steps).  After SLOW_AFTER steps the last
    // two iterates span, to (lambda_1 / lambda_3)^k, the plane of the two slowest eigenvectors; a Rayleigh-Ritz step in
    // that plane (2 x 2 symmetric eigenproblem of the inverse operator) separates them exactly and ends the walk.  Point
    // sets that converge earlier leave through the same test as before, with the same result.
    constexpr int SLOW_AFTER = 48;
    double xp[n];  // the iterate before x
#pragma unroll
    for (int i = 0; i < n; ++i) xp[i] = x[i];
    bool converged = false;
    for (int it = 0; it < SLOW_AFTER; ++it) {
        double y[n];
        apply_inv(x, y);
        double nn = 0, dot = 0;
#pragma unroll
        for (int i = 0; i < n; ++i) nn += y[i] * y[i];
        const double inv = 1.0 / sqrt(nn);
#pragma unroll
        for (int i = 0; i < n; ++i) { y[i] *= inv; dot += y[i] * x[i]; }
#pragma unroll
        for (int i = 0; i < n; ++i) { xp[i] = x[i]; x[i] = y[i]; }
        if (it >= 2 && 1.0 - fabs(dot) < 1e-16) { converged = true; break; }
    }
    if (!converged) {
        // orthonormal basis {q1 = x, q2 = xp - (xp . x) x normalised} of span{x, xp}; H = Q^T B Q with B = (A + mu I)^-1
        double q2[n], b1[n], b2[n];
        double d = 0;
#pragma unroll
        for (int i = 0; i < n; ++i) d += xp[i] * x[i];
        double nn = 0;
#pragma unroll
        for (int i = 0; i < n; ++i) { q2[i] = xp[i] - d * x[i]; nn += q2[i] * q2[i]; }
        if (nn > 1e-28) {  // (else the two iterates coincide to rounding: x is the answer)
            const double inv = 1.0 / sqrt(nn);
#pragma unroll
            for (int i = 0; i < n; ++i) q2[i] *= inv;
            apply_inv(x, b1);
            apply_inv(q2, b2);
            double h11 = 0, h12 = 0, h22 = 0;
#pragma unroll
            for (int i = 0; i < n; ++i) { h11 += x[i] * b1[i]; h12 += x[i] * b2[i]; h22 += q2[i] * b2[i]; }
            // dominant eigenvector (c, s) of [[h11, h12], [h12, h22]] (largest eigenvalue of B = smallest of A)
            const double half = 0.5 * (h11 - h22), rad = sqrt(half * half + h12 * h12);
            double c, sn;
            if (half >= 0) { c = half + rad; sn = h12; } else { c = h12; sn = rad - half; }
            const double nrm = sqrt(c * c + sn * sn);
            if (nrm > 0) {
                c /= nrm; sn /= nrm;
                double m2 = 0;
#pragma unroll
                for (int i = 0; i < n; ++i) { x[i] = c * x[i] + sn * q2[i]; m2 += x[i] * x[i]; }
                const double im = 1.0 / sqrt(m2);
#pragma unroll
                for (int i = 0; i < n; ++i) x[i] *= im;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < n; ++i) out[i] = x[i];
}

// solve 6x6 A x = b (Gaussian elimination, partial pivoting); A, b destroyed.
// Every index is a compile-time constant after unrolling: the row exchange is a predicated select over the candidate rows,
// not an access through the run-time pivot index -- that form put A and b into scratch memory, and the dependent memory
// round trips of one elimination were most of a Levenberg-Marquardt step (pnp_kernel on 12 ill-posed detections: 701 us
// of a 2.45 ms batch-1 frame).  Same comparisons, same operations in the same order: the results are bit-identical.
PNP_HD_INLINE void solve6(double (&A)[36], double (&b)[6], double (&x)[6]) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        double best = fabs(A[c * 6 + c]);
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            const double a = fabs(A[r * 6 + c]);
            if (a > best) { best = a; piv = r; }
        }
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            const bool sw = piv == r;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double t = A[c * 6 + k], u = A[r * 6 + k];
                A[c * 6 + k] = sw ? u : t;
                A[r * 6 + k] = sw ? t : u;
            }
            const double t = b[c], u = b[r];
            b[c] = sw ? u : t;
            b[r] = sw ? t : u;
        }
        const double d = A[c * 6 + c];
        if (d != 0.0) {
#pragma unroll
            for (int r = c + 1; r < 6; ++r) {
                const double f = A[r * 6 + c] / d;
                if (f != 0.0) {
#pragma unroll
                    for (int k = c; k < 6; ++k) A[r * 6 + k] -= f * A[c * 6 + k];
                    b[r] -= f * b[c];
                }
            }
        }
    }
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        double s = b[r];
#pragma unroll
        for (int k = r + 1; k < 6; ++k) s -= A[r * 6 + k] * x[k];
        x[r] = (A[r * 6 + r] != 0.0) ? s / A[r * 6 + r] : 0.0;
    }
}

// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (row-major, destroyed: eigenvalues end on the diagonal);
// V (n x n, row-major) receives the eigenvectors as COLUMNS.  FLOOR_EXIT (the product's form): besides the plain test
// (off-diagonal mass <= 1e-34 of the diagonal's) the sweeps stop at the ROUNDING FLOOR -- once the mass is below 1e-24 of the
// diagonal's and a whole sweep no longer quarters it, the remaining rotations turn by angles below one ulp (rank-deficient
// matrices such as EPnP's 12 x 12 never reach 1e-34 and used to run all 60 sweeps; profiles/NOTES.md round 4).  On
// well-conditioned matrices both forms give the same eigen-pairs (tests/test_pnp_linalg_cpu.py compares them and numpy).
template <bool FLOOR_EXIT = true>
PNP_HD void jacobi_eig(double* A, int n, double* V) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
    double prev_off = 0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, dia = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                if (i == j) dia += A[i * n + j] * A[i * n + j];
                else off += A[i * n + j] * A[i * n + j];
            }
        if (off <= 1e-34 * dia || off == 0.0) break;
        if (FLOOR_EXIT && sweep > 0 && off <= 1e-24 * dia && off >= 0.25 * prev_off) break;  // the rounding floor
        prev_off = off;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                const double tau = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
                const double c = 1 / sqrt(1 + t * t), sn = t * c;
                for (int k = 0; k < n; ++k) {
                    const double a = A[k * n + p], b = A[k * n + q];
                    A[k * n + p] = c * a - sn * b;
                    A[k * n + q] = sn * a + c * b;
                }
                for (int k = 0; k < n; ++k) {
                    const double a = A[p * n + k], b = A[q * n + k];
                    A[p * n + k] = c * a - sn * b;
                    A[q * n + k] = sn * a + c * b;
                }
                for (int k = 0; k < n; ++k) {
                    const double a = V[k * n + p], b = V[k * n + q];
                    V[k * n + p] = c * a - sn * b;
                    V[k * n + q] = sn * a + c * b;
                }
            }
    }
}
