// Batched cuboid PnP on device, float64, sixteen lanes per detection (one per image point) in the common case.
//
// Replaces the per-detection host loop `pnp_shell` -> `CuboidPNPSolver.solve_pnp` ->
// `cv2.solvePnPGeneric(flags=SOLVEPNP_ITERATIVE)` + `cv2.projectPoints`
// (/root/reference/src/lib/utils/pnp/cuboid_pnp_shell.py:11-24, cuboid_pnp_solver.py:141-239,
// base_detector.py:547-654).  OpenCV's arithmetic is not in the reference tree (un-vendored
// opencv-python>=4.5.3.56); pnp_kernel restates calib3d's published SOLVEPNP_ITERATIVE for >= 6
// non-planar points exactly as oracle/pnp.py does (the common case: one lane per detection, everything in registers);
// pnp_rare_kernel then handles the two branches that only rep_mode 4 / heavily filtered detections reach -- 4-5 valid
// points (the reference switches to SOLVEPNP_EPNP, cuboid_pnp_solver.py:162-163: EPnP as published, no refinement)
// and coplanar model points (homography initialisation of SOLVEPNP_ITERATIVE, then the same LM).  Common case:
//   normalise by K -> DLT: smallest eigenvector of L^T L (12x12; shifted inverse iteration in registers) -> det sign fix ->
//   polar factor R = U V^T, t *= |R| / |R_raw| -> Rodrigues -> Levenberg-Marquardt (<= 20 iterations,
//   eps = FLT_EPSILON, lambda = 10^k from k = -3, diag(JtJ) *= 1 + lambda) on pixel reprojection error.
// The cuboid model, point filtering (x or y < -5000 dropped; point i belongs to vertex i / (n/8)),
// z < 0 rejection, OpenGL conversion and axis-angle quaternion follow the reference's Python.
//
// The work is ~10^5 float64 operations per detection and a few hundred detections per batch: it is
// latency-bound, not bandwidth- or MFMA-bound, and the whole batch is one launch instead of a Python loop.
// pnp_kernel gives a detection the 16 lanes of one DPP row: lane k owns image point k for everything that is a sum
// over points (the DLT normal matrix, LM's J^T J / J^T e / reprojection error), reduced with an xor butterfly whose
// result is bit-identical on all 16 lanes, so the small dense algebra in between (eigenvector, polar factor, Rodrigues,
// the 6x6 solve) runs redundantly and every data-dependent branch of the Levenberg-Marquardt walk stays uniform
// inside the group.  Per iteration that is ~1500 instead of ~5300 dependent instructions, and a wavefront now holds 4
// detections instead of 64, so one slow walk (20 iterations on a poorly conditioned point set) no longer holds 63
// finished ones.  Results are deterministic; they differ from the one-lane order only in the summation order over points.
#include <cmath>  // (the host-testable headers below are included inside the namespace: their std includes come first)
#include <cfloat>

#include "cp_common.h"

namespace {

#include "pnp_linalg.h"  // DBL_EPS / FLT_EPS, rodrigues, polar3, rot_to_rvec, smallest_eigvec<n>, solve6 (host-testable)

struct Cam { double fx, fy, cx, cy; };

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


struct Problem {
    const float* P;  // [npts][2] image points of this detection
    double V3[8][3];
    int npts, per, nv;
    Cam cam;
};

__device__ __forceinline__ bool pt_valid(const Problem& q, int k) { return !(q.P[2 * k] < -5000.f || q.P[2 * k + 1] < -5000.f); }

// CvLevMarq on the 6 pose parameters (<= 20 iterations, eps = FLT_EPSILON, lambda = 10^k from k = -3); returns iterations
__device__ __forceinline__ int lm_refine(const Problem& q, double param[6]) {
    const float* P = q.P;
    const Cam cam = q.cam;
    double prev[6];
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
            for (int k = 0; k < q.npts; ++k) {
                if (!pt_valid(q, k)) continue;
                const double* M = q.V3[k / q.per];
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
            for (int k = 0; k < q.npts; ++k) {
                if (!pt_valid(q, k)) continue;
                double u, v, x, y, z;
                project1(R, param + 3, cam, q.V3[k / q.per], u, v, x, y, z);
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
    return iters;
}

// result row from the final (rvec, tvec): RMS reprojection error, projected cuboid, quaternions, OpenGL-frame pose
__device__ __forceinline__ void write_pose(const Problem& q, const double param[6], int iters, double* o) {
    const float* P = q.P;
    double R[9];
    rodrigues(param, R, nullptr);
    double e2 = 0;
    for (int k = 0; k < q.npts; ++k) {
        if (!pt_valid(q, k)) continue;
        double u, v, x, y, z;
        project1(R, param + 3, q.cam, q.V3[k / q.per], u, v, x, y, z);
        e2 += (u - P[2 * k]) * (u - P[2 * k]) + (v - P[2 * k + 1]) * (v - P[2 * k + 1]);
    }
    for (int k = 0; k < 6; ++k) o[1 + k] = param[k];
    o[7] = sqrt(e2) / sqrt(2.0 * q.nv);
    for (int v = 0; v < 8; ++v) {
        double u, vv, x, y, z;
        project1(R, param + 3, q.cam, q.V3[v], u, vv, x, y, z);
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

// ---- sixteen lanes per detection (pnp_kernel) ----
// sum over the 16 lanes of the group: xor butterfly; a + b == b + a bit for bit, so every lane ends with the same value
__device__ __forceinline__ double gsum16(double v) {
    v += __shfl_xor(v, 1, 16);
    v += __shfl_xor(v, 2, 16);
    v += __shfl_xor(v, 4, 16);
    v += __shfl_xor(v, 8, 16);
    return v;
}

// lm_refine with lane `sub` owning image point `sub`: the sums over points are group reductions, everything else (and
// every branch) is computed identically by the 16 lanes
__device__ __forceinline__ int lm_refine16(const Problem& q, double param[6], int sub) {
    const Cam cam = q.cam;
    const bool pv = sub < q.npts && pt_valid(q, sub);
    const double* Mq = q.V3[(sub < q.npts ? sub : 0) / q.per];
    const double M[3] = {Mq[0], Mq[1], Mq[2]};  // this lane's model point, read once (the only run-time index of the walk)
    const double pu = pv ? (double)q.P[2 * sub] : 0.0, pw = pv ? (double)q.P[2 * sub + 1] : 0.0;
    double prev[6];
    int lambda_lg10 = -3, iters = 0;
    double prev_err = 0, JtJ[36], JtE[6];
    bool calc_j = true;
    for (int guard = 0; guard < 2000; ++guard) {
        double R[9], dR[27];
        if (calc_j) {
            rodrigues(param, R, dR);
            double u, v, x, y, z;
            project1(R, param + 3, cam, M, u, v, x, y, z);
            const double eu = pv ? u - pu : 0.0, evv = pv ? v - pw : 0.0;
            double ju[6], jv[6];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double dx0 = M[0] * dR[j * 9 + 0] + M[1] * dR[j * 9 + 1] + M[2] * dR[j * 9 + 2];
                const double dy0 = M[0] * dR[j * 9 + 3] + M[1] * dR[j * 9 + 4] + M[2] * dR[j * 9 + 5];
                const double dz0 = M[0] * dR[j * 9 + 6] + M[1] * dR[j * 9 + 7] + M[2] * dR[j * 9 + 8];
                ju[j] = pv ? cam.fx * z * (dx0 - x * dz0) : 0.0;
                jv[j] = pv ? cam.fy * z * (dy0 - y * dz0) : 0.0;
            }
            ju[3] = pv ? cam.fx * z : 0.0; ju[4] = 0; ju[5] = pv ? -cam.fx * x * z : 0.0;
            jv[3] = 0; jv[4] = pv ? cam.fy * z : 0.0; jv[5] = pv ? -cam.fy * y * z : 0.0;
            const double e2 = gsum16(eu * eu + evv * evv);
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                JtE[a] = gsum16(ju[a] * eu + jv[a] * evv);
#pragma unroll
                for (int b = a; b < 6; ++b) {
                    const double t = gsum16(ju[a] * ju[b] + jv[a] * jv[b]);
                    JtJ[a * 6 + b] = t;
                    JtJ[b * 6 + a] = t;
                }
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) prev[k] = param[k];
            if (iters == 0) prev_err = sqrt(e2);
            calc_j = false;
        } else {
            rodrigues(param, R, nullptr);
            double u, v, x, y, z;
            project1(R, param + 3, cam, M, u, v, x, y, z);
            const double eu = pv ? u - pu : 0.0, evv = pv ? v - pw : 0.0;
            const double err = sqrt(gsum16(eu * eu + evv * evv));
            bool retry = false;
            if (err > prev_err) {
                ++lambda_lg10;
                if (lambda_lg10 <= 16) retry = true;
            }
            if (!retry) {
                lambda_lg10 = lambda_lg10 - 1 < -16 ? -16 : lambda_lg10 - 1;
                ++iters;
                double dn = 0, pn = 0;
#pragma unroll
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
#pragma unroll
        for (int k = 0; k < 36; ++k) Aq[k] = JtJ[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) { Aq[k * 7] *= 1.0 + lam; bq[k] = JtE[k]; }
        solve6(Aq, bq, dx);
#pragma unroll
        for (int k = 0; k < 6; ++k) param[k] = prev[k] - dx[k];
    }
    return iters;
}

// write_pose with the reprojection error summed over the group; only lane 0 of the group writes
__device__ __forceinline__ void write_pose16(const Problem& q, const double param[6], int iters, double* o, int sub) {
    double R[9];
    rodrigues(param, R, nullptr);
    const bool pv = sub < q.npts && pt_valid(q, sub);
    double e2 = 0;
    {
        double u, v, x, y, z;
        project1(R, param + 3, q.cam, q.V3[(sub < q.npts ? sub : 0) / q.per], u, v, x, y, z);
        const double du = pv ? u - q.P[2 * sub] : 0.0, dv = pv ? v - q.P[2 * sub + 1] : 0.0;
        e2 = gsum16(du * du + dv * dv);
    }
    if (sub != 0) return;
    for (int k = 0; k < 6; ++k) o[1 + k] = param[k];
    o[7] = sqrt(e2) / sqrt(2.0 * q.nv);
    for (int v = 0; v < 8; ++v) {
        double u, vv, x, y, z;
        project1(R, param + 3, q.cam, q.V3[v], u, vv, x, y, z);
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

// cuboid model + valid-point count of detection i
__device__ __forceinline__ void load_problem(Problem& q, const float* pts, const float* scale, const double* camp, int i, int npts) {
    q.cam = {camp[i * 4 + 0], camp[i * 4 + 1], camp[i * 4 + 2], camp[i * 4 + 3]};
    // cuboid: size = scale / scale[1]  (cuboid_pnp_shell.py:12), vertices cuboid_objectron.py:97-109
    const double s1 = (double)scale[i * 3 + 1];
    const double hw = 0.5 * ((double)scale[i * 3 + 0] / s1), hh = 0.5 * ((double)scale[i * 3 + 1] / s1),
                 hd = 0.5 * ((double)scale[i * 3 + 2] / s1);
    for (int v = 0; v < 8; ++v) {
        q.V3[v][0] = (v & 4) ? hw : -hw;
        q.V3[v][1] = (v & 2) ? hh : -hh;
        q.V3[v][2] = (v & 1) ? hd : -hd;
    }
    q.npts = npts;
    q.per = npts / 8;
    q.P = pts + (size_t)i * npts * 2;
    q.nv = 0;
    for (int k = 0; k < npts; ++k) q.nv += pt_valid(q, k) ? 1 : 0;
}

// pts [N][npts][2] float (npts = 8 or 16), scale [N][3] float, cam [N][4] double (fx, fy, cx, cy)
// out [N][CP_PNP_STRIDE] double;  scratch: unused since the eigen-solver moved into registers (kept in the ABI).
// Sixteen consecutive lanes = one detection (4 per wavefront, or one 16-lane workgroup each: cp_launch_pnp); lane `sub`
// owns image point `sub`.
__global__ __launch_bounds__(64) void pnp_kernel(const float* __restrict__ pts, const float* __restrict__ scale,
                                                 const double* __restrict__ camp, int N, int npts,
                                                 double* __restrict__ out, double* __restrict__ scratch) {
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gi >> 4, sub = gi & 15;
    if (i >= N) return;  // whole groups leave together
    double* o = out + (size_t)i * CP_PNP_STRIDE;
    if (sub == 0)
        for (int k = 0; k < CP_PNP_STRIDE; ++k) o[k] = 0.0;
    Problem q;
    load_problem(q, pts, scale, camp, i, npts);
    const Cam cam = q.cam;
    const int per = q.per, nv = q.nv;
    const float* P = q.P;
    double Mc[3] = {0, 0, 0};
    for (int k = 0; k < npts; ++k) {
        if (!pt_valid(q, k)) continue;
        for (int d = 0; d < 3; ++d) Mc[d] += q.V3[k / per][d];
    }
    int* rare = reinterpret_cast<int*>(scratch);  // [0] = number of rare detections, [1 ..] their indices (pnp_rare_kernel)
    if (nv < 6) {  // < 4: no pose; 4-5: EPnP branch of the reference (cuboid_pnp_solver.py:162-163), pnp_rare_kernel
        if (sub == 0) {
            o[35] = nv;
            o[0] = nv < 4 ? -1 : -2;
            if (nv >= 4) rare[1 + atomicAdd(&rare[0], 1)] = i;
        }
        return;
    }
    for (int d = 0; d < 3; ++d) Mc[d] /= nv;
    // planarity test of cvFindExtrinsicCameraParams2: second/third singular value of the 3x3 scatter
    {
        double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < npts; ++k) {
            if (!pt_valid(q, k)) continue;
            double dd[3];
            for (int d = 0; d < 3; ++d) dd[d] = q.V3[k / per][d] - Mc[d];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) S[a * 3 + b] += dd[a] * dd[b];
        }
        // eigenvalues of symmetric 3x3 by Jacobi
        for (int sw = 0; sw < 20; ++sw)
            for (int pp = 0; pp < 2; ++pp)
                for (int qq = pp + 1; qq < 3; ++qq) {
                    const double apq = S[pp * 3 + qq];
                    if (fabs(apq) < 1e-300) continue;
                    const double tau = (S[qq * 3 + qq] - S[pp * 3 + pp]) / (2 * apq);
                    const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
                    const double c = 1 / sqrt(1 + t * t), sn = t * c;
                    for (int k = 0; k < 3; ++k) { const double a = S[k * 3 + pp], bb = S[k * 3 + qq]; S[k * 3 + pp] = c * a - sn * bb; S[k * 3 + qq] = sn * a + c * bb; }
                    for (int k = 0; k < 3; ++k) { const double a = S[pp * 3 + k], bb = S[qq * 3 + k]; S[pp * 3 + k] = c * a - sn * bb; S[qq * 3 + k] = sn * a + c * bb; }
                }
        double e0 = S[0], e1 = S[4], e2 = S[8], tmp;
        if (e0 < e1) { tmp = e0; e0 = e1; e1 = tmp; }
        if (e1 < e2) { tmp = e1; e1 = e2; e2 = tmp; }
        if (e0 < e1) { tmp = e0; e0 = e1; e1 = tmp; }
        if (e2 / e1 < 1e-3) {  // planar: homography initialisation, pnp_rare_kernel
            if (sub == 0) {
                o[35] = nv;
                o[0] = -3;
                rare[1 + atomicAdd(&rare[0], 1)] = i;
            }
            return;
        }
    }
    // ---- DLT ----  L^T L as a packed lower triangle in registers: this lane's point, then summed over the group
    double As[78];
    {
        const bool pv = sub < npts && pt_valid(q, sub);
        const double* M = q.V3[(sub < npts ? sub : 0) / per];
        const double x = pv ? -((double)P[2 * sub] - cam.cx) / cam.fx : 0.0, y = pv ? -((double)P[2 * sub + 1] - cam.cy) / cam.fy : 0.0;
        const double w = pv ? 1.0 : 0.0;  // an invalid / absent point contributes two zero rows
        const double r0[12] = {w * M[0], w * M[1], w * M[2], w, 0, 0, 0, 0, x * M[0], x * M[1], x * M[2], x};
        const double r1[12] = {0, 0, 0, 0, w * M[0], w * M[1], w * M[2], w, y * M[0], y * M[1], y * M[2], y};
#pragma unroll
        for (int a = 0; a < 12; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) As[TRI(a, b)] = gsum16(r0[a] * r0[b] + r1[a] * r1[b]);
    }
    double ev[12];
    smallest_eigvec<12>(As, ev);
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
    if (!(sc > DBL_EPS)) {
        if (sub == 0) { o[35] = nv; o[0] = 0; }
        return;
    }
    double R0[9];
    polar3(RR, R0);
    const double f = sqrt(3.0) / sc;  // |R|_F of an orthonormal matrix is sqrt(3)
    double param[6];
    rot_to_rvec(R0, param);
    param[3] = tt[0] * f; param[4] = tt[1] * f; param[5] = tt[2] * f;

    const int iters = lm_refine16(q, param, sub);
    if (sub == 0) o[35] = nv;
    write_pose16(q, param, iters, o, sub);
}


// ---------------------------------------------------------------------------------------------------------------
// Rare branches (one lane per detection whose status is -2 or -3; everything else returns at once).  Plain loops
// over small local arrays: these run for a handful of detections, if at all.
// ---------------------------------------------------------------------------------------------------------------
// (jacobi_eig: the sequential cyclic Jacobi eigen-decomposition lives in pnp_linalg.h, where the host tests build it too)

#ifdef CP_PNP_TIMING
__device__ double g_pnp_t_jacobi;  // tuning build: shader clocks of the last EPnP eigen-decomposition (one detection at a time)
#endif
// The same rotations by the 16 lanes of a detection's group (A, V in LDS; every lane runs the same (p, q) walk, lane k owns
// row / column k of each update): 3 LDS round trips per rotation instead of 3 n.  The convergence test sums in another order
// (group reduction) than jacobi_eig; the rotation sequence and arithmetic are the same.
__device__ void jacobi_eig16(double* A, int n, double* V, int k) {
    const bool mine = k < n;
    if (mine)
        for (int j = 0; j < n; ++j) V[k * n + j] = k == j ? 1.0 : 0.0;
    __syncthreads();
    double prev_off = 0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, dia = 0;
        if (mine)
            for (int j = 0; j < n; ++j) {
                const double a = A[k * n + j];
                if (k == j) dia += a * a;
                else off += a * a;
            }
        off = gsum16(off);
        dia = gsum16(dia);
        if (off <= 1e-34 * dia || off == 0.0) break;  // (the same value on every lane)
        // A rank-deficient matrix (EPnP's M^T M with 4 - 5 points has a 2 - 4 dimensional null space) never reaches 1e-34: the
        // off-diagonal mass stalls at the rounding floor (~1e-30 of the diagonal) and the walk used all 60 sweeps -- 4.7 M
        // clocks, all of pnp_rare_kernel's 2.1 ms.  Once the mass is tiny AND a whole sweep no longer quarters it, it is noise.
        if (sweep > 0 && off <= 1e-24 * dia && off >= 0.25 * prev_off) break;
        prev_off = off;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                const double tau = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
                const double c = 1 / sqrt(1 + t * t), sn = t * c;
                __syncthreads();  // everyone has read apq / app / aqq
                if (mine) {
                    const double a = A[k * n + p], b = A[k * n + q];
                    A[k * n + p] = c * a - sn * b;
                    A[k * n + q] = sn * a + c * b;
                    const double va = V[k * n + p], vb = V[k * n + q];
                    V[k * n + p] = c * va - sn * vb;
                    V[k * n + q] = sn * va + c * vb;
                }
                __syncthreads();
                if (mine) {
                    const double a = A[p * n + k], b = A[q * n + k];
                    A[p * n + k] = c * a - sn * b;
                    A[q * n + k] = sn * a + c * b;
                }
                __syncthreads();
            }
    }
    __syncthreads();
}

// order[k] = index of the k-th smallest diagonal entry of A (n <= 12)
__device__ void ascending(const double* A, int n, int* order) {
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 1; i < n; ++i) {
        const int o = order[i];
        int j = i - 1;
        while (j >= 0 && A[order[j] * n + order[j]] > A[o * n + o]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = o;
    }
}

// least squares A x = b (rows x m, m <= 5) through the normal equations; false when they are singular
// Least squares through the normal equations (ROWS x M, M <= 5), everything in registers: compile-time shapes, the pivot
// search unrolled and the row exchange as predicated selects (pnp_linalg.h: solve6).  The run-time-shaped form kept its
// matrices in scratch memory, and the 18 solves of an EPnP detection were most of its 2 ms (profiles/NOTES.md, round 4).
// Same comparisons and operations in the same order as the loop it replaces.
template <int ROWS, int M>
__device__ __forceinline__ bool lstsq_t(const double (&A)[ROWS * M], const double (&b)[ROWS], double (&x)[M]) {
    double Nn[M * M], r[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        r[i] = 0;
#pragma unroll
        for (int k = 0; k < ROWS; ++k) r[i] += A[k * M + i] * b[k];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            double v = 0;
#pragma unroll
            for (int k = 0; k < ROWS; ++k) v += A[k * M + i] * A[k * M + j];
            Nn[i * M + j] = v;
        }
    }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < M; ++c) {
        int piv = c;
        double best = fabs(Nn[c * M + c]);
#pragma unroll
        for (int k = c + 1; k < M; ++k) {
            const double a = fabs(Nn[k * M + c]);
            if (a > best) { best = a; piv = k; }
        }
        ok = ok && (best > 1e-300);
#pragma unroll
        for (int k2 = c + 1; k2 < M; ++k2) {
            const bool sw = piv == k2;
#pragma unroll
            for (int k = 0; k < M; ++k) {
                const double t = Nn[c * M + k], u = Nn[k2 * M + k];
                Nn[c * M + k] = sw ? u : t;
                Nn[k2 * M + k] = sw ? t : u;
            }
            const double t = r[c], u = r[k2];
            r[c] = sw ? u : t;
            r[k2] = sw ? t : u;
        }
        const double d = ok ? Nn[c * M + c] : 1.0;
#pragma unroll
        for (int k = c + 1; k < M; ++k) {
            const double f = Nn[k * M + c] / d;
#pragma unroll
            for (int j = c; j < M; ++j) Nn[k * M + j] -= f * Nn[c * M + j];
            r[k] -= f * r[c];
        }
    }
    if (!ok) return false;
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
        double v = r[i];
#pragma unroll
        for (int j = i + 1; j < M; ++j) v -= Nn[i * M + j] * x[j];
        x[i] = v / Nn[i * M + i];
    }
    return true;
}

__device__ double det3(const double* R) {
    return R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
}

// orthogonal polar factor U V^T of a non-singular 3x3 matrix (what the SVDs of cv::Rodrigues / epnp's absolute
// orientation return); false for a (numerically) singular input
__device__ bool polar_checked(const double* A, double* R) {
    const double d = det3(A);
    double n2 = 0;
    for (int k = 0; k < 9; ++k) n2 += A[k] * A[k];
    if (!(n2 > 0) || !(n2 < 1e300) || !(fabs(d) > 1e-13 * n2 * sqrt(n2))) return false;
    polar3(A, R);
    for (int k = 0; k < 9; ++k)
        if (!(fabs(R[k]) < 2.0)) return false;
    return true;
}

// Per-detection work arrays of the rare branches live in LDS (RARE_LANES detections per workgroup), not in private
// memory and not in global memory: ~6 KB of scratch per lane made the runtime re-allocate scratch on every launch
// (+3 ms per call), and a single lane walking Jacobi sweeps over a global-memory matrix pays a DRAM round trip per
// element (measured: 15.8 ms for a handful of planar detections).
constexpr int RARE_WS = 600;   // doubles per detection

struct Valid {  // the surviving correspondences of one detection
    int n;
    double (*X)[3];   // [16] model point
    double (*uv)[2];  // [16] pixel
};

__device__ void collect(const Problem& q, Valid& v, double* w) {
    v.X = reinterpret_cast<double(*)[3]>(w);        // 48
    v.uv = reinterpret_cast<double(*)[2]>(w + 48);  // 32
    v.n = 0;
    for (int k = 0; k < q.npts; ++k) {
        if (!pt_valid(q, k)) continue;
        for (int d = 0; d < 3; ++d) v.X[v.n][d] = q.V3[k / q.per][d];
        v.uv[v.n][0] = (double)q.P[2 * k];
        v.uv[v.n][1] = (double)q.P[2 * k + 1];
        ++v.n;
    }
}

// Planar initialisation of cvFindExtrinsicCameraParams2 (oracle/pnp.py planar_init / homography_dlt): rotate the model
// plane to z = 0, normalised-DLT homography to the K-normalised image points, [h1 h2 h1 x h2] -> nearest rotation,
// t from h3
__device__ bool planar_init(const Problem& q, const Valid& v, double param[6], double* w) {
    const int n = v.n;
    double Mc[3] = {0, 0, 0};
    for (int k = 0; k < n; ++k)
        for (int d = 0; d < 3; ++d) Mc[d] += v.X[k][d] / n;
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, E[9];
    for (int k = 0; k < n; ++k)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) S[a * 3 + b] += (v.X[k][a] - Mc[a]) * (v.X[k][b] - Mc[b]);
    jacobi_eig(S, 3, E);
    int ord[3];
    ascending(S, 3, ord);
    double Rt[9];  // rows = eigenvectors by DESCENDING eigenvalue (the V^T of the scatter's SVD)
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rt[r * 3 + c] = E[c * 3 + ord[2 - r]];
    if (Rt[2] * Rt[2] + Rt[5] * Rt[5] < 1e-10)
        for (int k = 0; k < 9; ++k) Rt[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (det3(Rt) < 0)
        for (int k = 0; k < 9; ++k) Rt[k] = -Rt[k];
    double Tt[3];
    for (int r = 0; r < 3; ++r) Tt[r] = -(Rt[r * 3] * Mc[0] + Rt[r * 3 + 1] * Mc[1] + Rt[r * 3 + 2] * Mc[2]);
    double(*src)[2] = reinterpret_cast<double(*)[2]>(w + 80);
    double(*dst)[2] = reinterpret_cast<double(*)[2]>(w + 112);
    double cm[2] = {0, 0}, cM[2] = {0, 0}, sm[2] = {0, 0}, sM[2] = {0, 0};
    for (int k = 0; k < n; ++k) {
        const double* M = v.X[k];
        src[k][0] = Rt[0] * M[0] + Rt[1] * M[1] + Rt[2] * M[2] + Tt[0];
        src[k][1] = Rt[3] * M[0] + Rt[4] * M[1] + Rt[5] * M[2] + Tt[1];
        dst[k][0] = (v.uv[k][0] - q.cam.cx) / q.cam.fx;
        dst[k][1] = (v.uv[k][1] - q.cam.cy) / q.cam.fy;
        for (int d = 0; d < 2; ++d) { cm[d] += dst[k][d] / n; cM[d] += src[k][d] / n; }
    }
    for (int k = 0; k < n; ++k)
        for (int d = 0; d < 2; ++d) { sm[d] += fabs(dst[k][d] - cm[d]) / n; sM[d] += fabs(src[k][d] - cM[d]) / n; }
    if (sm[0] < DBL_EPS || sm[1] < DBL_EPS || sM[0] < DBL_EPS || sM[1] < DBL_EPS) return false;
    for (int d = 0; d < 2; ++d) { sm[d] = 1.0 / sm[d]; sM[d] = 1.0 / sM[d]; }
    // smallest eigenvector of L^T L (9 x 9, packed lower triangle in registers, like the DLT's 12 x 12)
    double Lp[45], H0[9], H[9], T[9];
#pragma unroll
    for (int k = 0; k < 45; ++k) Lp[k] = 0;
    for (int k = 0; k < n; ++k) {
        const double x = (dst[k][0] - cm[0]) * sm[0], y = (dst[k][1] - cm[1]) * sm[1];
        const double X = (src[k][0] - cM[0]) * sM[0], Y = (src[k][1] - cM[1]) * sM[1];
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x}, Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
#pragma unroll
        for (int a = 0; a < 9; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) Lp[TRI(a, b)] += Lx[a] * Lx[b] + Ly[a] * Ly[b];
    }
    smallest_eigvec<9>(Lp, H0);
    // H = invHnorm * H0 * Hnorm2
    const double invH[9] = {1.0 / sm[0], 0, cm[0], 0, 1.0 / sm[1], cm[1], 0, 0, 1};
    const double Hn2[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T[r * 3 + c] = H0[r * 3] * Hn2[c] + H0[r * 3 + 1] * Hn2[3 + c] + H0[r * 3 + 2] * Hn2[6 + c];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H[r * 3 + c] = invH[r * 3] * T[c] + invH[r * 3 + 1] * T[3 + c] + invH[r * 3 + 2] * T[6 + c];
    if (!(fabs(H[8]) > 0)) return false;
    for (int k = 0; k < 9; ++k) H[k] /= (k == 8 ? 1.0 : H[8]);
    H[8] = 1.0;
    double h1[3] = {H[0], H[3], H[6]}, h2[3] = {H[1], H[4], H[7]}, h3[3] = {H[2], H[5], H[8]};
    const double n1 = sqrt(h1[0] * h1[0] + h1[1] * h1[1] + h1[2] * h1[2]), n2 = sqrt(h2[0] * h2[0] + h2[1] * h2[1] + h2[2] * h2[2]);
    if (!(n1 > 0) || !(n2 > 0) || !(n1 < 1e300) || !(n2 < 1e300)) return false;
    for (int d = 0; d < 3; ++d) { h1[d] /= n1; h2[d] /= n2; }
    double t[3];
    for (int d = 0; d < 3; ++d) t[d] = h3[d] * (2.0 / (n1 + n2));
    const double hx[3] = {h1[1] * h2[2] - h1[2] * h2[1], h1[2] * h2[0] - h1[0] * h2[2], h1[0] * h2[1] - h1[1] * h2[0]};
    const double Rraw[9] = {h1[0], h2[0], hx[0], h1[1], h2[1], hx[1], h1[2], h2[2], hx[2]};
    double R1[9], R[9];
    if (!polar_checked(Rraw, R1) || det3(R1) < 0) return false;
    double tf[3];
    for (int r = 0; r < 3; ++r) tf[r] = R1[r * 3] * Tt[0] + R1[r * 3 + 1] * Tt[1] + R1[r * 3 + 2] * Tt[2] + t[r];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = R1[r * 3] * Rt[c] + R1[r * 3 + 1] * Rt[3 + c] + R1[r * 3 + 2] * Rt[6 + c];
    rot_to_rvec(R, param);
    param[3] = tf[0]; param[4] = tf[1]; param[5] = tf[2];
    return true;
}

// EPnP (Lepetit, Moreno-Noguer, Fua 2009) as cv::epnp runs it (oracle/pnp.py solve_pnp_epnp): control points, barycentric
// coordinates, null space of M^T M, three beta linearisations + 5 Gauss-Newton steps each, absolute orientation, best
// reprojection error.  No LM refinement follows for this flag.
// Called by all 16 lanes of the detection's group with identical arguments: everything is computed redundantly (the LDS work
// space receives the same values from every lane) except the 12 x 12 eigen-decomposition, which the lanes share.
__device__ bool epnp(const Problem& q, const Valid& v, double param[6], double* w, int sub) {
    const int n = v.n;
    const Cam cam = q.cam;
    double cws[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int k = 0; k < n; ++k)
        for (int d = 0; d < 3; ++d) cws[0][d] += v.X[k][d] / n;
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, E[9];
    for (int k = 0; k < n; ++k)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) S[a * 3 + b] += (v.X[k][a] - cws[0][a]) * (v.X[k][b] - cws[0][b]);
    jacobi_eig(S, 3, E);
    int o3[3];
    ascending(S, 3, o3);
    for (int i = 0; i < 3; ++i) {
        const int e = o3[2 - i];
        const double lam = S[e * 3 + e] > 0 ? S[e * 3 + e] : 0.0, kf = sqrt(lam / n);
        for (int d = 0; d < 3; ++d) cws[i + 1][d] = cws[0][d] + kf * E[d * 3 + e];
    }
    // barycentric coordinates: CC [columns cws[j] - cws[0]] al = pw - cws[0]
    double CC[9], CCi[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) CC[r * 3 + c] = cws[c + 1][r] - cws[0][r];
    const double dC = det3(CC);
    if (!(fabs(dC) > 1e-300)) return false;  // coplanar control points: alphas undefined (cv::epnp returns garbage here)
    CCi[0] = (CC[4] * CC[8] - CC[5] * CC[7]) / dC; CCi[1] = (CC[2] * CC[7] - CC[1] * CC[8]) / dC; CCi[2] = (CC[1] * CC[5] - CC[2] * CC[4]) / dC;
    CCi[3] = (CC[5] * CC[6] - CC[3] * CC[8]) / dC; CCi[4] = (CC[0] * CC[8] - CC[2] * CC[6]) / dC; CCi[5] = (CC[2] * CC[3] - CC[0] * CC[5]) / dC;
    CCi[6] = (CC[3] * CC[7] - CC[4] * CC[6]) / dC; CCi[7] = (CC[1] * CC[6] - CC[0] * CC[7]) / dC; CCi[8] = (CC[0] * CC[4] - CC[1] * CC[3]) / dC;
    double(*al)[4] = reinterpret_cast<double(*)[4]>(w + 80);  // 64
    for (int k = 0; k < n; ++k) {
        const double d0 = v.X[k][0] - cws[0][0], d1 = v.X[k][1] - cws[0][1], d2 = v.X[k][2] - cws[0][2];
        for (int j = 0; j < 3; ++j) al[k][j + 1] = CCi[j * 3] * d0 + CCi[j * 3 + 1] * d1 + CCi[j * 3 + 2] * d2;
        al[k][0] = 1.0 - al[k][1] - al[k][2] - al[k][3];
    }
    double* MtM = w + 144;
    double* EV = w + 288;
    for (int k = 0; k < 144; ++k) MtM[k] = 0;
    for (int k = 0; k < n; ++k) {
        double r0[12], r1[12];
        for (int j = 0; j < 4; ++j) {
            r0[3 * j] = al[k][j] * cam.fx; r0[3 * j + 1] = 0; r0[3 * j + 2] = al[k][j] * (cam.cx - v.uv[k][0]);
            r1[3 * j] = 0; r1[3 * j + 1] = al[k][j] * cam.fy; r1[3 * j + 2] = al[k][j] * (cam.cy - v.uv[k][1]);
        }
        for (int a = 0; a < 12; ++a)
            for (int b = 0; b < 12; ++b) MtM[a * 12 + b] += r0[a] * r0[b] + r1[a] * r1[b];
    }
#ifdef CP_PNP_TIMING
    const long long tj0 = clock64();
#endif
    __syncthreads();
    jacobi_eig16(MtM, 12, EV, sub);
#ifdef CP_PNP_TIMING
    g_pnp_t_jacobi = (double)(clock64() - tj0);
#endif
    int o12[12];
    ascending(MtM, 12, o12);
    double(*vv)[12] = reinterpret_cast<double(*)[12]>(w + 432);  // 48: the four null-space vectors
    for (int k = 0; k < 4; ++k)
        for (int a = 0; a < 12; ++a) vv[k][a] = EV[a * 12 + o12[k]];
    constexpr int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    double(*L)[10] = reinterpret_cast<double(*)[10]>(w + 480);  // 60
    double rho[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double d[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) d[k][c] = vv[k][3 * pa[i] + c] - vv[k][3 * pb[i] + c];
        auto dot = [&](int a, int b) { return d[a][0] * d[b][0] + d[a][1] * d[b][1] + d[a][2] * d[b][2]; };
        L[i][0] = dot(0, 0); L[i][1] = 2 * dot(0, 1); L[i][2] = dot(1, 1); L[i][3] = 2 * dot(0, 2); L[i][4] = 2 * dot(1, 2);
        L[i][5] = dot(2, 2); L[i][6] = 2 * dot(0, 3); L[i][7] = 2 * dot(1, 3); L[i][8] = 2 * dot(2, 3); L[i][9] = dot(3, 3);
        rho[i] = 0;
        for (int c = 0; c < 3; ++c) rho[i] += (cws[pa[i]][c] - cws[pb[i]][c]) * (cws[pa[i]][c] - cws[pb[i]][c]);
    }
    double cand[3][4];
    bool cok[3] = {false, false, false};
    {   // N = 1: betas 11, 12, 13, 14
        double A[24], x[4];
        constexpr int cols[4] = {0, 1, 3, 6};
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) A[i * 4 + j] = L[i][cols[j]];
        if (lstsq_t<6, 4>(A, rho, x)) {
            const double sgn = x[0] < 0 ? -1.0 : 1.0, b0 = sqrt(fabs(x[0]));
            cand[0][0] = b0; cand[0][1] = sgn * x[1] / b0; cand[0][2] = sgn * x[2] / b0; cand[0][3] = sgn * x[3] / b0;
            cok[0] = b0 > 0;
        }
    }
#pragma unroll
    for (int variant = 0; variant < 2; ++variant) {  // N = 2 (betas 11, 12, 22) and N = 3 (+ 13, 23)
        double x[5] = {0, 0, 0, 0, 0};
        bool solved;
        if (variant == 0) {
            double A[18], x3[3];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) A[i * 3 + j] = L[i][j];
            solved = lstsq_t<6, 3>(A, rho, x3);
            x[0] = x3[0]; x[1] = x3[1]; x[2] = x3[2];
        } else {
            double A[30];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) A[i * 5 + j] = L[i][j];
            solved = lstsq_t<6, 5>(A, rho, x);
        }
        if (!solved) continue;
        double b0, b1;
        if (x[0] < 0) { b0 = sqrt(-x[0]); b1 = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
        else { b0 = sqrt(x[0]); b1 = x[2] > 0 ? sqrt(x[2]) : 0.0; }
        if (x[1] < 0) b0 = -b0;
        cand[1 + variant][0] = b0; cand[1 + variant][1] = b1;
        cand[1 + variant][2] = (variant == 1 && b0 != 0) ? x[3] / b0 : 0.0;
        cand[1 + variant][3] = 0.0;
        cok[1 + variant] = true;
    }
    double best = 1e300, bestR[9], bestT[3];
    bool have = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (!cok[c]) continue;
        double b[4] = {cand[c][0], cand[c][1], cand[c][2], cand[c][3]};
        bool fin = true;
        for (int it = 0; it < 5 && fin; ++it) {  // Gauss-Newton on the six distance constraints
            double A[24], r[6], dx[4];
            for (int i = 0; i < 6; ++i) {
                const double* l = L[i];
                A[i * 4 + 0] = 2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3];
                A[i * 4 + 1] = l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3];
                A[i * 4 + 2] = l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3];
                A[i * 4 + 3] = l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3];
                r[i] = rho[i] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] + l[4] * b[1] * b[2] +
                                 l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] + l[8] * b[2] * b[3] + l[9] * b[3] * b[3]);
            }
            if (!lstsq_t<6, 4>(A, r, dx)) { fin = false; break; }
            for (int k = 0; k < 4; ++k) b[k] += dx[k];
        }
        if (!fin) continue;
        double ccs[4][3], cc[3] = {0, 0, 0}, cw[3] = {0, 0, 0};
        double(*pcs)[3] = reinterpret_cast<double(*)[3]>(w + 540);  // 48
        for (int j = 0; j < 4; ++j)
            for (int d = 0; d < 3; ++d) ccs[j][d] = b[0] * vv[0][3 * j + d] + b[1] * vv[1][3 * j + d] + b[2] * vv[2][3 * j + d] + b[3] * vv[3][3 * j + d];
        for (int k = 0; k < n; ++k)
            for (int d = 0; d < 3; ++d) pcs[k][d] = al[k][0] * ccs[0][d] + al[k][1] * ccs[1][d] + al[k][2] * ccs[2][d] + al[k][3] * ccs[3][d];
        if (pcs[0][2] < 0)
            for (int k = 0; k < n; ++k)
                for (int d = 0; d < 3; ++d) pcs[k][d] = -pcs[k][d];
        for (int k = 0; k < n; ++k)
            for (int d = 0; d < 3; ++d) { cc[d] += pcs[k][d] / n; cw[d] += v.X[k][d] / n; }
        double ABt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, R[9];
        for (int k = 0; k < n; ++k)
            for (int a = 0; a < 3; ++a)
                for (int d = 0; d < 3; ++d) ABt[a * 3 + d] += (pcs[k][a] - cc[a]) * (v.X[k][d] - cw[d]);
        if (!polar_checked(ABt, R)) continue;
        if (det3(R) < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
        double t[3], err = 0;
        for (int r = 0; r < 3; ++r) t[r] = cc[r] - (R[r * 3] * cw[0] + R[r * 3 + 1] * cw[1] + R[r * 3 + 2] * cw[2]);
        for (int k = 0; k < n; ++k) {
            double u, w, x, y, z;
            project1(R, t, cam, v.X[k], u, w, x, y, z);
            err += sqrt((u - v.uv[k][0]) * (u - v.uv[k][0]) + (w - v.uv[k][1]) * (w - v.uv[k][1]));
        }
        if (err == err && err < best) {
            best = err;
            have = true;
            for (int k = 0; k < 9; ++k) bestR[k] = R[k];
            for (int k = 0; k < 3; ++k) bestT[k] = t[k];
        }
    }
    if (!have) return false;
    rot_to_rvec(bestR, param);
    param[3] = bestT[0]; param[4] = bestT[1]; param[5] = bestT[2];
    return true;
}

// One 16-lane workgroup per RARE detection, taken from the list pnp_kernel compacted (round 3 walked all N slots with one
// lane per detection, eight detections of different branches sharing a wavefront: 8 ms for 650 of them).  Lane 0 runs the
// branch's initialisation (EPnP: the whole solve; planar: homography -> pose) out of the workgroup's LDS work space; the
// Levenberg-Marquardt refinement of the planar branch then runs on all 16 lanes like pnp_kernel's (lane = image point).
constexpr int RARE_LANES = 16;  // workgroup size of pnp_rare_kernel: the launcher below and the kernel's guard use this one constant
__global__ __launch_bounds__(RARE_LANES) void pnp_rare_kernel(const float* __restrict__ pts, const float* __restrict__ scale,
                                                      const double* __restrict__ camp, int N, int npts,
                                                      double* __restrict__ out, const int* __restrict__ rare) {
    __shared__ double w[RARE_WS];
    // INVARIANT: the 16 lanes of this workgroup are one (partial) wavefront and run collect / epnp / planar_init in lock-step on
    // the ONE shared work space w[] -- every lane computes the same values and stores them to the same words (including the
    // read-modify-write accumulations), which is only sound inside a single wavefront.  Launch it with 16 threads, never more
    // than 64 (__launch_bounds__ above; any other block size is refused here rather than computing garbage, and LOUDLY: the
    // detections of the list get status 0 = "failure" instead of keeping pnp_kernel's internal -2 / -3 marks).
    if ((int)blockIdx.x >= rare[0]) return;
    if (blockDim.x != RARE_LANES) {
        if (threadIdx.x == 0) out[(size_t)rare[1 + blockIdx.x] * CP_PNP_STRIDE] = 0;
        return;
    }
    const int i = rare[1 + blockIdx.x], sub = threadIdx.x;
    double* o = out + (size_t)i * CP_PNP_STRIDE;
    const int status = (int)o[0];  // -2 (4-5 valid points) / -3 (planar model), written by pnp_kernel
#ifdef CP_PNP_TIMING
    const long long t0 = clock64();
#endif
    Problem q;
    load_problem(q, pts, scale, camp, i, npts);
    // every lane runs the branch's initialisation with the same data (same values into the shared work space)
    Valid v;
    collect(q, v, w);
    double param[6];
    bool ok;
    if (status == -2) {
        ok = epnp(q, v, param, w, sub);  // SOLVEPNP_EPNP: the pose is returned as is (no iterative refinement in OpenCV)
    } else {
        ok = planar_init(q, v, param, w);
        if (!ok)
            for (int k = 0; k < 6; ++k) param[k] = 0.0;  // cvFindExtrinsicCameraParams2 falls back to r = t = 0
        ok = true;
    }
#ifdef CP_PNP_TIMING
    const long long t1 = clock64();
#endif
    int iters = 0;
    if (status == -3) iters = lm_refine16(q, param, sub);
    for (int k = 0; k < 6; ++k) ok = ok && (param[k] == param[k]) && fabs(param[k]) < 1e300;
    if (!ok) {
        if (sub == 0) o[0] = 0;
        return;
    }
    write_pose16(q, param, iters, o, sub);
#ifdef CP_PNP_TIMING
    if (sub == 0) { o[37] = (double)(t1 - t0); o[38] = (double)(clock64() - t1); o[39] = status == -2 ? g_pnp_t_jacobi : 0.0; }
#endif
}

}  // namespace

size_t cp_pnp_ws_bytes(int N) { return (size_t)N * 288 * sizeof(double) + 256; }  // used: (1 + N) ints, the rare-detection list

int cp_launch_pnp(hipStream_t s, const float* pts, const float* scale, const double* cam, int N, int npts, double* out,
                  void* ws) {
    if (N < 1) return CP_OK;
    if (npts != 8 && npts != 16) return CP_ERR_INVALID;
    // Up to 256 problems (a batch-1 frame: K = 100 slots): one detection per wavefront (a 16-lane workgroup each) -- the
    // Levenberg-Marquardt walks of different detections branch differently (lambda retries, iteration counts), and four
    // ill-posed ones sharing a wavefront run one after the other (batch-1 frame of the random-weight network: 2.09 -> 2.02 ms).
    // Larger batches keep four detections per wavefront: with well-posed detections the walks agree and a quarter of the
    // wavefronts is the better trade (640 / 6400 detections: 314 / 583 us against 424 / 741).
    if (!ws || hipMemsetAsync(ws, 0, sizeof(int), s) != hipSuccess) return CP_ERR_INVALID;  // the rare list's counter
    if (N <= 256)
        hipLaunchKernelGGL(pnp_kernel, dim3(N), dim3(16), 0, s, pts, scale, cam, N, npts, out, (double*)ws);
    else
        hipLaunchKernelGGL(pnp_kernel, dim3((N * 16 + 63) / 64), dim3(64), 0, s, pts, scale, cam, N, npts, out, (double*)ws);
    // detections the common-case kernel marked -2 (4-5 valid points) / -3 (planar model) and appended to the list at `ws`:
    // one 16-lane workgroup per list entry (the grid covers the worst case; workgroups past the count leave at once)
    hipLaunchKernelGGL(pnp_rare_kernel, dim3(N), dim3(RARE_LANES), 0, s, pts, scale, cam, N, npts, out, (const int*)ws);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}
