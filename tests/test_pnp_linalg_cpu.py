"""The scalar numerics of the device PnP (centerpose_amd/csrc/pnp_linalg.h: Rodrigues' formula with derivatives, polar
factor, matrix -> rotation vector, smallest eigenvector by inverse iteration with a Rayleigh-Ritz finish, 6 x 6
elimination with register-resident pivoting) compiled for the host by
tests/native/pnp_linalg_host.cpp and compared with numpy on the matrices the solve meets: DLT normal matrices of posed
cuboids (well separated, and with the two smallest eigenvalues close), rank-deficient ones, damped J^T J systems."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import pnp as opnp

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def host():
    out = os.path.join(REPO, "tests", "_build", "libcp_pnp_linalg_host.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = os.path.join(REPO, "tests", "native", "pnp_linalg_host.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", out])
    lib = ctypes.CDLL(out)
    for f in (lib.pnp_host_smallest_eigvec12, lib.pnp_host_smallest_eigvec9, lib.pnp_host_solve6, lib.pnp_host_rodrigues,
              lib.pnp_host_rodrigues_nojac, lib.pnp_host_polar3, lib.pnp_host_rot_to_rvec, lib.pnp_host_jacobi_eig):
        f.restype = None
    return lib


def _packed(A):
    n = A.shape[0]
    return np.array([A[i, j] for i in range(n) for j in range(i + 1)], np.float64)


def _eig(host, A):
    n = A.shape[0]
    p, out = _packed(A), np.zeros(n)
    fn = host.pnp_host_smallest_eigvec12 if n == 12 else host.pnp_host_smallest_eigvec9
    fn(p.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


def _dlt_matrix(rng, z_range, noise):
    K = np.array([[663.0287679036459, 0, 300.2775065104167], [0, 663.0287679036459, 395.00066121419275], [0, 0, 1]])
    sc = np.array([rng.uniform(0.3, 3), rng.uniform(0.5, 2.0), rng.uniform(0.3, 3)])
    q = rng.randn(4)
    R = opnp.quat_xyzw_to_matrix(q / np.linalg.norm(q))
    t = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(*z_range)])
    obj = np.repeat(opnp.cuboid_vertices(sc / sc[1]), 2, axis=0)
    uv = opnp.project_points(obj, opnp.matrix_to_rodrigues(R), t, K) + rng.randn(16, 2) * noise
    mn = np.stack([(uv[:, 0] - K[0, 2]) / K[0, 0], (uv[:, 1] - K[1, 2]) / K[1, 1]], 1)
    L = np.zeros((32, 12))
    for i in range(16):
        X, Y, Z = obj[i]
        x, y = -mn[i, 0], -mn[i, 1]
        L[2 * i] = [X, Y, Z, 1, 0, 0, 0, 0, x * X, x * Y, x * Z, x]
        L[2 * i + 1] = [0, 0, 0, 0, X, Y, Z, 1, y * X, y * Y, y * Z, y]
    return L.T @ L


def _angle(a, b):
    return np.arccos(min(1.0, abs(float(a @ b))))


def test_smallest_eigvec_on_dlt_matrices(host):
    """Well-posed point sets (eigen-gap ratio << 1): the iteration converges and leaves through its own test."""
    rng = np.random.RandomState(5)
    for _ in range(200):
        A = _dlt_matrix(rng, (3.0, 12.0), rng.choice([0.0, 0.3, 1.0]))
        w, V = np.linalg.eigh(A)
        v = _eig(host, A)
        assert abs(np.linalg.norm(v) - 1) < 1e-12
        if w[0] / w[1] < 0.5:
            assert _angle(v, V[:, 0]) < 1e-7, (w[:3], _angle(v, V[:, 0]))


def test_smallest_eigvec_with_close_eigenvalues(host):
    """lambda_1 / lambda_2 in 0.75 .. 0.97 (far objects under pixel noise: plain inverse iteration would need 100 .. 600
    steps): the Rayleigh-Ritz finish after 48 steps must return the smallest eigenvector, not a mixture."""
    rng = np.random.RandomState(11)
    n_slow, worst = 0, 0.0
    for _ in range(3000):
        A = _dlt_matrix(rng, (20.0, 60.0), 2.0)
        w, V = np.linalg.eigh(A)
        r12, r13 = w[0] / w[1], w[0] / w[2]
        if not (0.75 < r12 < 0.97) or r13 > 0.5:
            continue
        n_slow += 1
        ang = _angle(_eig(host, A), V[:, 0])
        worst = max(worst, ang)
        # what is left after the Ritz step is the third eigenvector's share: (lambda_1 / lambda_3)^48, and the conditioning of
        # the 2 x 2 problem (1 / (1 - r12)) on float64 round-off
        assert ang < 1e-6 + 10 * r13 ** 48, (r12, r13, ang)
    assert n_slow >= 20, n_slow


def test_smallest_eigvec_synthetic_spectra(host):
    """Prescribed spectra (random orthogonal basis): separated, two close, exactly singular (exact correspondences)."""
    rng = np.random.RandomState(3)
    for n in (12, 9):
        for spec in ([1e-9] + list(np.linspace(1, 5, n - 1)), [1.0, 1.05] + list(np.linspace(3, 9, n - 2)),
                     [0.0] + list(np.linspace(0.5, 2, n - 1)), [2.0, 2.0002] + list(np.linspace(8, 20, n - 2))):
            Q, _ = np.linalg.qr(rng.randn(n, n))
            A = (Q * np.array(spec)) @ Q.T
            A = 0.5 * (A + A.T)
            v = _eig(host, A)
            gap = spec[0] / spec[1] if spec[1] else 0.0
            if gap < 0.999:
                assert _angle(v, Q[:, 0]) < (1e-7 if gap < 0.9 else 1e-4), (n, spec[:2], _angle(v, Q[:, 0]))
            # in every case: a unit vector inside the span of the two smallest eigenvectors
            resid = v - Q[:, :2] @ (Q[:, :2].T @ v)
            assert np.linalg.norm(resid) < 1e-6 and abs(np.linalg.norm(v) - 1) < 1e-12


def test_solve6_matches_numpy(host):
    """Damped normal equations of a Levenberg-Marquardt step (J^T J with its diagonal times 1 + lambda), matrices that need
    row exchanges, and a singular one (zero pivot -> that unknown is 0, as the device code defines it)."""
    rng = np.random.RandomState(9)

    def solve(A, b):
        x = np.zeros(6)
        Ac, bc = np.ascontiguousarray(A, np.float64).copy(), np.ascontiguousarray(b, np.float64).copy()
        host.pnp_host_solve6(Ac.ctypes.data_as(ctypes.c_void_p), bc.ctypes.data_as(ctypes.c_void_p),
                             x.ctypes.data_as(ctypes.c_void_p))
        return x

    for _ in range(300):
        J = rng.randn(32, 6) * rng.uniform(0.1, 100, 6)
        A = J.T @ J
        A[np.diag_indices(6)] *= 1 + 10.0 ** rng.randint(-6, 4)
        b = rng.randn(6)
        np.testing.assert_allclose(solve(A, b), np.linalg.solve(A, b), rtol=1e-8, atol=1e-12)
    for _ in range(100):  # general matrices: pivoting matters
        A = rng.randn(6, 6)
        A[rng.randint(6), :] *= 1e-6
        A[0, 0] = 0.0
        b = rng.randn(6)
        np.testing.assert_allclose(solve(A, b), np.linalg.solve(A, b), rtol=1e-6, atol=1e-9)
    A = np.diag([1.0, 2.0, 0.0, 4.0, 5.0, 6.0])
    x = solve(A, np.arange(1.0, 7.0))
    np.testing.assert_allclose(x, [1.0, 1.0, 0.0, 1.0, 1.0, 1.0])


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_rodrigues_and_derivatives_match_oracle(host):
    """cv::Rodrigues vector -> matrix and dR/dr (the Jacobian block of every Levenberg-Marquardt step): against the float64
    restatement, against finite differences, at theta ~ 0 (the identity branch) and near pi."""
    rng = np.random.RandomState(2)
    vecs = [rng.randn(3) * s for s in (1e-3, 0.1, 1.0, 2.0) for _ in range(25)]
    vecs += [np.zeros(3), np.array([1e-17, 0, 0]), np.array([np.pi - 1e-6, 0, 0]), np.array([0, 2.2, 2.2])]
    for r in vecs:
        r = np.ascontiguousarray(r, np.float64)
        R, J, R2 = np.zeros(9), np.zeros(27), np.zeros(9)
        host.pnp_host_rodrigues(_ptr(r), _ptr(R), _ptr(J))
        host.pnp_host_rodrigues_nojac(_ptr(r), _ptr(R2))
        Ro, Jo = opnp.rodrigues_to_matrix(r, jac=True)
        np.testing.assert_allclose(R.reshape(3, 3), Ro, rtol=0, atol=1e-14)
        np.testing.assert_array_equal(R, R2)
        np.testing.assert_allclose(J.reshape(3, 9), Jo, rtol=1e-12, atol=1e-13)
        if np.linalg.norm(r) > 1e-2:  # central differences of the oracle's matrix
            h = 1e-6
            for i in range(3):
                e = np.zeros(3)
                e[i] = h
                fd = (opnp.rodrigues_to_matrix(r + e) - opnp.rodrigues_to_matrix(r - e)).reshape(9) / (2 * h)
                np.testing.assert_allclose(J.reshape(3, 9)[i], fd, atol=1e-8)


def test_polar_factor_and_rotation_vector(host):
    """polar3 = U V^T of a matrix with positive determinant (what the DLT's 3 x 3 block is turned into); rot_to_rvec is the
    inverse of Rodrigues' formula, including rotations by (almost) pi."""
    rng = np.random.RandomState(4)
    for _ in range(200):
        q = rng.randn(4)
        R0 = opnp.quat_xyzw_to_matrix(q / np.linalg.norm(q))
        A = R0 @ (np.eye(3) + 0.2 * rng.randn(3, 3)) * rng.uniform(0.01, 50)
        if np.linalg.det(A) <= 0:
            continue
        A = np.ascontiguousarray(A.reshape(9))
        P = np.zeros(9)
        host.pnp_host_polar3(_ptr(A), _ptr(P))
        U, _, Vt = np.linalg.svd(A.reshape(3, 3))
        np.testing.assert_allclose(P.reshape(3, 3), U @ Vt, atol=1e-10)
    angles = list(rng.uniform(0.01, 3.1, 100)) + [np.pi - 1e-7, np.pi, 1e-9, 0.0]
    for th in angles:
        ax = rng.randn(3)
        ax /= np.linalg.norm(ax)
        R = np.ascontiguousarray(opnp.rodrigues_to_matrix(ax * th).reshape(9))
        r = np.zeros(3)
        host.pnp_host_rot_to_rvec(_ptr(R), _ptr(r))
        np.testing.assert_allclose(opnp.rodrigues_to_matrix(r), R.reshape(3, 3), atol=2e-7 if th > 3.1 else 1e-9)


@pytest.mark.parametrize("n", [3, 12])
def test_jacobi_eig_floor_exit_changes_nothing_on_well_conditioned_matrices(host, n):
    """jacobi_eig (pnp_linalg.h; the 3 x 3 decompositions of the planar / EPnP branches run it on the device): the round-4 exit
    at the rounding floor must be invisible where the plain 1e-34 test is reachable -- same eigenvalues and eigenvectors with
    and without it, both equal to numpy's -- and must still return a valid decomposition of a rank-deficient matrix (EPnP's
    12 x 12 of rank <= 11-ish, the case it was added for), where the plain form needs all 60 sweeps."""
    rng = np.random.RandomState(n)

    def run(A, floor):
        a, V = np.array(A, np.float64).copy(), np.zeros((n, n))
        host.pnp_host_jacobi_eig(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n), V.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(floor))
        return np.diag(a).copy(), V

    for trial in range(20):
        M = rng.randn(n + 4, n)
        A = M.T @ M + 0.1 * np.eye(n)          # well conditioned, distinct eigenvalues
        (w1, V1), (w0, V0) = run(A, 1), run(A, 0)
        np.testing.assert_allclose(w1, w0, rtol=1e-13, atol=1e-13 * np.abs(w0).max())
        ref = np.linalg.eigvalsh(A)
        np.testing.assert_allclose(np.sort(w1), ref, rtol=1e-11, atol=1e-12 * ref.max())
        for w, V in ((w1, V1), (w0, V0)):
            np.testing.assert_allclose(V.T @ V, np.eye(n), atol=1e-12)
            np.testing.assert_allclose(A @ V, V * w[None, :], atol=1e-10 * ref.max())
        # the eigenvectors agree up to sign
        np.testing.assert_allclose(np.abs(np.sum(V1 * V0, axis=0)), np.ones(n), atol=1e-10)
    if n == 12:  # rank-deficient: the floor exit's own case
        M = rng.randn(7, 12)
        A = M.T @ M
        w1, V1 = run(A, 1)
        np.testing.assert_allclose(A @ V1, V1 * w1[None, :], atol=1e-9 * np.abs(w1).max())
        assert np.sum(np.abs(w1) < 1e-9 * np.abs(w1).max()) == 5
