"""ORACLE (test infrastructure): float64 numpy restatement of the cuboid PnP stage.

PARITY UNPINNED.  The arithmetic of this stage is not in the reference tree: the reference calls
``cv2.solvePnPGeneric(..., flags=SOLVEPNP_ITERATIVE)`` and ``cv2.projectPoints`` from the un-vendored,
un-pinned dependency ``opencv-python>=4.5.3.56`` (/root/reference/requirements.txt:11; call sites
/root/reference/src/lib/utils/pnp/cuboid_pnp_solver.py:159-171, :203-205) and no reference test pins
its results; OpenCV is not installed here.  This file restates OpenCV calib3d's published algorithms:
SOLVEPNP_EPNP for 4-5 correspondences (cuboid_pnp_solver.py:162-163; Lepetit et al. as cv::epnp runs it), the planar
(homography) initialisation of SOLVEPNP_ITERATIVE for coplanar model points, and
SOLVEPNP_ITERATIVE on >= 6 non-planar points (cvFindExtrinsicCameraParams2: normalise by K,
DLT on the 2N x 12 system via the smallest eigenvector of L^T L, det sign fix, SVD orthogonalisation
R = U V^T with t rescaled by |R|/|R_raw|, Rodrigues, then CvLevMarq: <= 20 iterations, eps =
FLT_EPSILON, lambda = 10^k with k starting at -3, JtJ diagonal scaled by (1 + lambda)), and anchors
correctness on construction: noise-free known poses are recovered to 1e-8, noisy ones agree with an
independent scipy least-squares minimiser (tests/test_pnp_oracle.py).

Everything around the solver follows the reference's own Python:
  cuboid vertices        cuboid_objectron.py:80-110
  point filtering/order  cuboid_pnp_solver.py:141-157
  z < 0 rejection        cuboid_pnp_solver.py:207-220
  OpenGL conversion      cuboid_pnp_solver.py:179-196
  quaternion             cuboid_pnp_solver.py:241-247 (pyrr axis-angle, xyzw)
  pnp_shell packaging    cuboid_pnp_shell.py:11-93
"""
import numpy as np

FLT_EPSILON = 1.1920928955078125e-07
DBL_EPSILON = 2.220446049250313e-16


def cuboid_vertices(size3d):
    """cuboid_objectron.py:80-110: x = width, y = height, z = depth, centred at the origin."""
    w, h, d = [float(v) for v in size3d]
    r, l = w / 2.0, -w / 2.0
    t, b = h / 2.0, -h / 2.0
    f, re = d / 2.0, -d / 2.0
    return np.array([[l, b, re], [l, b, f], [l, t, re], [l, t, f],
                     [r, b, re], [r, b, f], [r, t, re], [r, t, f]], dtype=np.float64)


def rodrigues_to_matrix(r, jac=False):
    """cv::Rodrigues vector -> matrix (+ 3x9 jacobian dR/dr, R row-major)."""
    r = np.asarray(r, np.float64).reshape(3)
    theta = np.linalg.norm(r)
    if theta < DBL_EPSILON:
        R = np.eye(3)
        J = np.array([[0, 0, 0, 0, 0, -1, 0, 1, 0], [0, 0, 1, 0, 0, 0, -1, 0, 0], [0, -1, 0, 1, 0, 0, 0, 0, 0]],
                     np.float64)
        return (R, J) if jac else R
    c, s = np.cos(theta), np.sin(theta)
    c1 = 1.0 - c
    it = 1.0 / theta
    a = r * it
    rrt = np.outer(a, a)
    rx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = c * np.eye(3) + c1 * rrt + s * rx
    if not jac:
        return R
    I = np.eye(3).reshape(9)
    x, y, z = a
    drrt = np.array([[x + x, y, z, y, 0, 0, z, 0, 0], [0, x, 0, x, y + y, z, 0, z, 0], [0, 0, x, 0, 0, y, x, y, z + z]])
    drx = np.array([[0, 0, 0, 0, 0, -1, 0, 1, 0], [0, 0, 1, 0, 0, 0, -1, 0, 0], [0, -1, 0, 1, 0, 0, 0, 0, 0]], np.float64)
    J = np.zeros((3, 9))
    for i in range(3):
        ri = a[i]
        a0, a1, a2, a3, a4 = -s * ri, (s - 2 * c1 * it) * ri, c1 * it, (c - s * it) * ri, s * it
        J[i] = a0 * I + a1 * rrt.reshape(9) + a2 * drrt[i] + a3 * rx.reshape(9) + a4 * drx[i]
    return R, J


def matrix_to_rodrigues(R):
    """cv::Rodrigues matrix -> vector (re-orthogonalises by SVD first)."""
    U, _, Vt = np.linalg.svd(np.asarray(R, np.float64))
    R = U @ Vt
    rx, ry, rz = R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]
    s = np.sqrt((rx * rx + ry * ry + rz * rz) * 0.25)
    c = min(max((R[0, 0] + R[1, 1] + R[2, 2] - 1) * 0.5, -1.0), 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1) * 0.5
        rx = np.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1) * 0.5
        ry = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1) * 0.5
        rz = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((R[1, 2] > 0) != (ry * rz > 0)):
            rz = -rz
        v = np.array([rx, ry, rz])
        return v * (theta / np.linalg.norm(v))
    return np.array([rx, ry, rz]) * (theta / (2 * s))


def project_points(obj, rvec, tvec, Kmat, jac=False):
    """cv::projectPoints without distortion (+ d(u,v)/d(rvec,tvec) as a 2N x 6 matrix)."""
    fx, fy, cx, cy = Kmat[0, 0], Kmat[1, 1], Kmat[0, 2], Kmat[1, 2]
    if jac:
        R, dRdr = rodrigues_to_matrix(rvec, True)
    else:
        R = rodrigues_to_matrix(rvec)
    P = obj @ R.T + np.asarray(tvec, np.float64).reshape(1, 3)
    z = np.where(P[:, 2] != 0, 1.0 / np.where(P[:, 2] != 0, P[:, 2], 1.0), 1.0)
    x, y = P[:, 0] * z, P[:, 1] * z
    uv = np.stack([fx * x + cx, fy * y + cy], axis=1)
    if not jac:
        return uv
    N = obj.shape[0]
    J = np.zeros((2 * N, 6))
    for j in range(3):
        dx0 = obj @ dRdr[j, 0:3]
        dy0 = obj @ dRdr[j, 3:6]
        dz0 = obj @ dRdr[j, 6:9]
        J[0::2, j] = fx * z * (dx0 - x * dz0)
        J[1::2, j] = fy * z * (dy0 - y * dz0)
    J[0::2, 3] = fx * z
    J[0::2, 5] = -fx * x * z
    J[1::2, 4] = fy * z
    J[1::2, 5] = -fy * y * z
    return uv, J


def dlt_init(obj, mn):
    """Non-planar initialisation of cvFindExtrinsicCameraParams2 (needs >= 6 points)."""
    N = obj.shape[0]
    L = np.zeros((2 * N, 12))
    for i in range(N):
        X, Y, Z = obj[i]
        x, y = -mn[i, 0], -mn[i, 1]
        L[2 * i] = [X, Y, Z, 1, 0, 0, 0, 0, x * X, x * Y, x * Z, x]
        L[2 * i + 1] = [0, 0, 0, 0, X, Y, Z, 1, y * X, y * Y, y * Z, y]
    LL = L.T @ L
    w, V = np.linalg.eigh(LL)  # ascending eigenvalues
    RRt = V[:, 0].reshape(3, 4)
    RR = RRt[:, :3]
    tt = RRt[:, 3]
    if np.linalg.det(RR) < 0:
        RR, tt = -RR, -tt
    sc = np.linalg.norm(RR)
    U, _, Vt = np.linalg.svd(RR)
    R = U @ Vt
    t = tt * (np.linalg.norm(R) / sc)
    return matrix_to_rodrigues(R), t


def homography_dlt(src, dst):
    """Normalised DLT (cv::findHomography method 0, runKernel of HomographyEstimatorCallback): both point sets are
    shifted to their centroid and scaled to unit mean |coordinate|, the 9-vector is the eigenvector of L^T L with the
    smallest eigenvalue, de-normalised and divided by h22.  OpenCV then polishes H with <= 10 LM iterations when there
    are more than 4 points; that polish is not restated -- the pose is refined by the full LM below anyway."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    n = len(src)
    cm, cM = dst.mean(0), src.mean(0)
    sm = np.abs(dst - cm).sum(0) / n
    sM = np.abs(src - cM).sum(0) / n
    if min(sm.min(), sM.min()) < 2.220446049250313e-16:
        return None
    sm, sM = 1.0 / sm, 1.0 / sM
    invHnorm = np.array([[1 / sm[0], 0, cm[0]], [0, 1 / sm[1], cm[1]], [0, 0, 1]])
    Hnorm2 = np.array([[sM[0], 0, -cM[0] * sM[0]], [0, sM[1], -cM[1] * sM[1]], [0, 0, 1]])
    LtL = np.zeros((9, 9))
    for i in range(n):
        x, y = (dst[i] - cm) * sm
        X, Y = (src[i] - cM) * sM
        Lx = np.array([X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x])
        Ly = np.array([0, 0, 0, X, Y, 1, -y * X, -y * Y, -y])
        LtL += np.outer(Lx, Lx) + np.outer(Ly, Ly)
    w, V = np.linalg.eigh(LtL)
    H0 = V[:, 0].reshape(3, 3)
    H = invHnorm @ H0 @ Hnorm2
    return H / H[2, 2]


def planar_init(obj, mn):
    """Planar initialisation of cvFindExtrinsicCameraParams2 (object points in one plane: rotate the plane to z = 0,
    homography to the normalised image points, [h1 h2 h1xh2] -> nearest rotation, t from h3)."""
    Mc = obj.mean(axis=0)
    MM = (obj - Mc).T @ (obj - Mc)
    _, _, Vt = np.linalg.svd(MM)
    Rt = Vt.copy()
    if Rt[0, 2] ** 2 + Rt[1, 2] ** 2 < 1e-10:
        Rt = np.eye(3)
    if np.linalg.det(Rt) < 0:
        Rt = -Rt
    Tt = -Rt @ Mc
    Mxy = (obj @ Rt.T + Tt)[:, :2]
    H = homography_dlt(Mxy, mn)
    if H is None or not np.all(np.isfinite(H)):
        return np.zeros(3), np.zeros(3)
    h1, h2, h3 = H[:, 0].copy(), H[:, 1].copy(), H[:, 2].copy()
    n1, n2 = np.linalg.norm(h1), np.linalg.norm(h2)
    h1 /= n1
    h2 /= n2
    t = h3 * (2.0 / (n1 + n2))
    R = np.stack([h1, h2, np.cross(h1, h2)], axis=1)
    R = rodrigues_to_matrix(matrix_to_rodrigues(R))
    t = R @ Tt + t
    R = R @ Rt
    return matrix_to_rodrigues(R), t


# ---------------------------------------------------------------------------------------------------------------
# EPnP (Lepetit, Moreno-Noguer, Fua, IJCV 2009) as cv::epnp runs it for SOLVEPNP_EPNP: 4 control points (centroid +
# principal directions), barycentric coordinates, the 12 x 12 null space of M^T M, betas from the three linearisations
# (N = 1, 2, 3 as find_betas_approx_1/2/3) polished by 5 Gauss-Newton steps each, absolute orientation per candidate,
# the candidate with the smallest reprojection error wins.  No LM refinement follows in OpenCV for this flag.
# ---------------------------------------------------------------------------------------------------------------
_EPNP_PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))


def _epnp_betas_gauss_newton(L, rho, betas, iters=5):
    b = np.array(betas, np.float64)
    for _ in range(iters):
        A = np.zeros((6, 4))
        r = np.zeros(6)
        for i in range(6):
            l = L[i]
            A[i] = [2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3],
                    l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3],
                    l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3],
                    l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3]]
            r[i] = rho[i] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] +
                             l[4] * b[1] * b[2] + l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] +
                             l[8] * b[2] * b[3] + l[9] * b[3] * b[3])
        b = b + np.linalg.lstsq(A, r, rcond=None)[0]
    return b


def _absolute_orientation(pw, pc):
    """R, t with pc ~ R pw + t (Horn / Arun by SVD, as epnp::estimate_R_and_t)."""
    cw, cc = pw.mean(0), pc.mean(0)
    ABt = (pc - cc).T @ (pw - cw)
    U, _, Vt = np.linalg.svd(ABt)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        R[2] = -R[2]
    return R, cc - R @ cw


def solve_pnp_epnp(obj, img, Kmat):
    obj = np.asarray(obj, np.float64)
    img = np.asarray(img, np.float64)
    n = len(obj)
    fx, fy, cx, cy = Kmat[0, 0], Kmat[1, 1], Kmat[0, 2], Kmat[1, 2]
    # control points
    cws = np.zeros((4, 3))
    cws[0] = obj.mean(0)
    PW0 = obj - cws[0]
    dc, uc = np.linalg.eigh(PW0.T @ PW0)
    for i in range(3):
        cws[i + 1] = cws[0] + np.sqrt(max(dc[2 - i], 0.0) / n) * uc[:, 2 - i]
    CC = (cws[1:] - cws[0]).T
    if abs(np.linalg.det(CC)) < 1e-300:  # coplanar / collinear points: no barycentric coordinates (cv::epnp returns garbage)
        return False, np.zeros(3), np.zeros(3)
    al = np.linalg.solve(CC, (obj - cws[0]).T).T
    alphas = np.hstack([1 - al.sum(1, keepdims=True), al])
    M = np.zeros((2 * n, 12))
    for i in range(n):
        for j in range(4):
            M[2 * i, 3 * j:3 * j + 3] = [alphas[i, j] * fx, 0, alphas[i, j] * (cx - img[i, 0])]
            M[2 * i + 1, 3 * j:3 * j + 3] = [0, alphas[i, j] * fy, alphas[i, j] * (cy - img[i, 1])]
    _, V = np.linalg.eigh(M.T @ M)
    v = [V[:, k] for k in range(4)]  # the four smallest eigenvalues, ascending (cv: ut[11], ut[10], ut[9], ut[8])
    dv = [[v[k][3 * a:3 * a + 3] - v[k][3 * b:3 * b + 3] for (a, b) in _EPNP_PAIRS] for k in range(4)]
    L = np.zeros((6, 10))
    for i in range(6):
        d = [dv[k][i] for k in range(4)]
        L[i] = [d[0] @ d[0], 2 * d[0] @ d[1], d[1] @ d[1], 2 * d[0] @ d[2], 2 * d[1] @ d[2], d[2] @ d[2],
                2 * d[0] @ d[3], 2 * d[1] @ d[3], 2 * d[2] @ d[3], d[3] @ d[3]]
    rho = np.array([np.sum((cws[a] - cws[b]) ** 2) for (a, b) in _EPNP_PAIRS])
    cands = []
    # N = 1 linearisation (betas 11, 12, 13, 14)
    b4 = np.linalg.lstsq(L[:, [0, 1, 3, 6]], rho, rcond=None)[0]
    if b4[0] < 0:
        b0 = np.sqrt(-b4[0])
        cands.append([b0, -b4[1] / b0, -b4[2] / b0, -b4[3] / b0])
    else:
        b0 = np.sqrt(b4[0])
        cands.append([b0, b4[1] / b0, b4[2] / b0, b4[3] / b0])
    # N = 2 (betas 11, 12, 22)
    b3 = np.linalg.lstsq(L[:, [0, 1, 2]], rho, rcond=None)[0]
    if b3[0] < 0:
        b0, b1 = np.sqrt(-b3[0]), (np.sqrt(-b3[2]) if b3[2] < 0 else 0.0)
    else:
        b0, b1 = np.sqrt(b3[0]), (np.sqrt(b3[2]) if b3[2] > 0 else 0.0)
    if b3[1] < 0:
        b0 = -b0
    cands.append([b0, b1, 0.0, 0.0])
    # N = 3 (betas 11, 12, 22, 13, 23)
    b5 = np.linalg.lstsq(L[:, [0, 1, 2, 3, 4]], rho, rcond=None)[0]
    if b5[0] < 0:
        b0, b1 = np.sqrt(-b5[0]), (np.sqrt(-b5[2]) if b5[2] < 0 else 0.0)
    else:
        b0, b1 = np.sqrt(b5[0]), (np.sqrt(b5[2]) if b5[2] > 0 else 0.0)
    if b5[1] < 0:
        b0 = -b0
    cands.append([b0, b1, b5[3] / b0 if b0 != 0 else 0.0, 0.0])
    best = None
    for betas in cands:
        if not np.all(np.isfinite(betas)):
            continue
        b = _epnp_betas_gauss_newton(L, rho, betas)
        ccs = sum(b[k] * v[k] for k in range(4)).reshape(4, 3)
        pcs = alphas @ ccs
        if pcs[0, 2] < 0:
            ccs, pcs = -ccs, -pcs
        R, t = _absolute_orientation(obj, pcs)
        P = obj @ R.T + t
        uv = np.stack([cx + fx * P[:, 0] / P[:, 2], cy + fy * P[:, 1] / P[:, 2]], 1)
        err = np.sqrt(((uv - img) ** 2).sum(1)).sum() / n
        if best is None or err < best[0]:
            best = (err, R, t)
    if best is None:
        return False, np.zeros(3), np.zeros(3)
    return True, matrix_to_rodrigues(best[1]), best[2]


def is_planar(obj):
    Mc = obj.mean(axis=0)
    MM = (obj - Mc).T @ (obj - Mc)
    w = np.linalg.svd(MM, compute_uv=False)
    return w[2] / w[1] < 1e-3


def solve_pnp_iterative(obj, img, Kmat, max_iter=20, eps=FLT_EPSILON, return_iters=False):
    """SOLVEPNP_ITERATIVE for >= 6 non-planar points, zero distortion.  Returns (ok, rvec, tvec)."""
    obj = np.asarray(obj, np.float64)
    img = np.asarray(img, np.float64)
    Kmat = np.asarray(Kmat, np.float64)
    N = obj.shape[0]
    if N < 4:
        raise ValueError("solvePnP needs at least 4 points")
    fx, fy, cx, cy = Kmat[0, 0], Kmat[1, 1], Kmat[0, 2], Kmat[1, 2]
    mn = np.stack([(img[:, 0] - cx) / fx, (img[:, 1] - cy) / fy], axis=1)
    if is_planar(obj):
        r, t = planar_init(obj, mn)
    elif N < 6:
        raise ValueError("DLT algorithm needs at least 6 points (OpenCV asserts here; the reference switches to EPnP)")
    else:
        r, t = dlt_init(obj, mn)
    param = np.concatenate([r, t])
    # ---- CvLevMarq state machine (completeSymmFlag = true, DECOMP_SVD) ----
    lambda_lg10 = -3
    iters = 0
    prev_param = param.copy()
    prev_err_norm = None
    state = "CALC_J"
    JtJ = JtErr = None
    while True:
        if state == "CALC_J":
            uv, J = project_points(obj, param[:3], param[3:], Kmat, True)
            err = (uv - img).reshape(-1)
            JtJ = J.T @ J
            JtErr = J.T @ err
            prev_param = param.copy()
            if iters == 0:
                prev_err_norm = np.linalg.norm(err)
            param = _lm_step(JtJ, JtErr, prev_param, lambda_lg10)
            state = "CHECK_ERR"
            continue
        # CHECK_ERR
        uv = project_points(obj, param[:3], param[3:], Kmat)
        err_norm = np.linalg.norm((uv - img).reshape(-1))
        if err_norm > prev_err_norm:
            lambda_lg10 += 1
            if lambda_lg10 <= 16:
                param = _lm_step(JtJ, JtErr, prev_param, lambda_lg10)
                continue
        lambda_lg10 = max(lambda_lg10 - 1, -16)
        iters += 1
        denom = np.linalg.norm(prev_param)
        rel = np.linalg.norm(param - prev_param) / (denom if denom > 0 else 1.0)
        if iters >= max_iter or rel < eps:
            break
        prev_err_norm = err_norm
        state = "CALC_J"
    if return_iters:
        return True, param[:3].copy(), param[3:].copy(), iters
    return True, param[:3].copy(), param[3:].copy()


def _lm_step(JtJ, JtErr, prev_param, lambda_lg10):
    lam = np.exp(lambda_lg10 * np.log(10.0))
    A = JtJ.copy()
    A[np.diag_indices_from(A)] *= 1.0 + lam
    delta = np.linalg.lstsq(A, JtErr, rcond=None)[0]
    return prev_param - delta


def solve_pnp_any(obj, img, Kmat, epnp=False):
    """cv2.solvePnPGeneric as the reference calls it (cuboid_pnp_solver.py:162-171): SOLVEPNP_EPNP when fewer than 6
    correspondences survive, SOLVEPNP_ITERATIVE (planar or DLT initialisation + LM) otherwise."""
    if epnp:
        return solve_pnp_epnp(obj, img, Kmat)
    return solve_pnp_iterative(obj, img, Kmat)


def axis_angle_quat_xyzw(rvec):
    """cuboid_pnp_solver.py:241-247 + pyrr.Quaternion.from_axis_rotation (normalised axis)."""
    rvec = np.asarray(rvec, np.float64).reshape(3)
    theta = np.sqrt(rvec[0] * rvec[0] + rvec[1] * rvec[1] + rvec[2] * rvec[2])
    axis = rvec / theta
    axis = axis / np.linalg.norm(axis)
    h = theta * 0.5
    return np.array([np.sin(h) * axis[0], np.sin(h) * axis[1], np.sin(h) * axis[2], np.cos(h)])


def quat_xyzw_to_matrix(q):
    """scipy Rotation.from_quat(q).as_matrix() (normalises q)."""
    x, y, z, w = np.asarray(q, np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def solve_cuboid_pnp(points2d, scale, Kmat, opencv_return=False):
    """CuboidPNPSolver.solve_pnp (cuboid_pnp_solver.py:91-239) for one detection.
    points2d: (8*n, 2), entries with x or y < -5000 are invalid.  scale: relative cuboid size.
    Returns dict(location, quaternion_xyzw, projected_points (8x2), rvec, tvec, reproj_err) or None."""
    pts = np.asarray(points2d, np.float64).reshape(-1, 2)
    verts = cuboid_vertices(np.asarray(scale, np.float64) / scale[1])  # cuboid_pnp_shell.py:12
    n_per = len(pts) / 8
    o2, o3 = [], []
    for i in range(len(pts)):
        if pts[i, 0] < -5000 or pts[i, 1] < -5000:
            continue
        o2.append(pts[i])
        o3.append(verts[int(i // n_per)])
    o2 = np.array(o2, dtype=float)
    o3 = np.array(o3, dtype=float)
    if len(o2) < 4:
        return None
    ok, rvec, tvec = solve_pnp_any(o3, o2, Kmat, epnp=len(o2) < 6)
    if not ok:
        return None
    proj_v = project_points(o3, rvec, tvec, Kmat)
    reproj = float(np.linalg.norm(proj_v - o2) / np.sqrt(2 * len(o2)))
    R = rodrigues_to_matrix(rvec)
    M3 = np.array([[0, 1, 0], [1, 0, 0], [0, 0, -1]], np.float64)
    R_gl = M3 @ R
    t_gl = M3 @ tvec
    rvec_gl = matrix_to_rodrigues(R_gl)
    projected = project_points(verts, rvec, tvec, Kmat)
    if tvec[2] < 0:
        return None
    if opencv_return:
        loc, quat = tvec.copy(), axis_angle_quat_xyzw(rvec)
    else:
        loc, quat = t_gl, axis_angle_quat_xyzw(rvec_gl)
    return {"location": loc, "quaternion_xyzw": quat, "projected_points": projected, "rvec": rvec, "tvec": tvec,
            "reproj_err": reproj}


def pnp_shell(points2d, scale, Kmat, width, height, category="cup", kps=None, opencv_return=False):
    """cuboid_pnp_shell.py:11-93.  Returns None when the detection is dropped, else a dict with
    kps_pnp (9x2 normalised), kps_3d_cam (9x3), location, quaternion_xyzw, projected_cuboid."""
    sol = solve_cuboid_pnp(points2d, scale, Kmat, opencv_return)
    if sol is None:
        return None
    verts = cuboid_vertices(np.asarray(scale, np.float64) / scale[1])
    ori = quat_xyzw_to_matrix(sol["quaternion_xyzw"])
    cam = verts @ ori.T + np.asarray(sol["location"]).reshape(1, 3)
    cam = np.vstack([cam.mean(axis=0, keepdims=True), cam])
    proj = sol["projected_points"]
    proj = np.vstack([proj.mean(axis=0, keepdims=True), proj]).copy()
    proj[:, 0] /= width
    proj[:, 1] /= height
    if category not in ("bike", "laptop", "shoe"):
        thr = 6 if category in ("book", "chair", "cereal_box") else 3
        nv = sum(1 for p in proj if p[0] < 0 or p[0] > 1 or p[1] < 0 or p[1] > 1)
        if nv >= thr:
            return None
    if not (0 < proj[0][0] < 1 and 0 < proj[0][1] < 1):
        return None
    out = dict(sol)
    out["kps_pnp"] = proj
    out["kps_3d_cam"] = cam
    return out
