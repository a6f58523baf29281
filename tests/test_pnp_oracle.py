"""The PnP oracle (oracle/pnp.py) is PARITY UNPINNED -- the reference's solver is cv2.solvePnPGeneric, un-vendored and
absent here -- so it is validated by construction: noise-free known poses are recovered by every branch (DLT + LM for
>= 6 non-planar points, homography + LM for coplanar points, EPnP for 4-5 points), and under pixel noise the LM result
is the least-squares optimum an independent scipy minimiser finds.  What CAN be pinned to the reference is pinned:
pnp_shell's packaging and visibility rejects (cuboid_pnp_shell.py:26-91), with the solver's answer injected
(tests/golden/pnp_shell_ref.json, from the reference's own function via oracle/tools/track_golden.py)."""
import json
import os

import numpy as np
import pytest
from scipy.optimize import least_squares

from oracle import pnp as opnp
from oracle.tools import track_golden as tg

K = tg.K_DEMO
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _geo(Ra, Rb):
    return np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


def _pose(rng):
    sc = np.array([rng.uniform(0.5, 2), 1.0, rng.uniform(0.5, 2)])
    V = opnp.cuboid_vertices(sc)
    q = rng.randn(4)
    R = opnp.quat_xyzw_to_matrix(q / np.linalg.norm(q))
    t = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(5, 9)])
    return V, R, t, opnp.project_points(V, opnp.matrix_to_rodrigues(R), t, K)


@pytest.mark.parametrize("idx,epnp,tol_deg", [
    (list(range(8)) * 2, False, 1e-5),          # rep_mode 1: every vertex twice -> DLT + LM
    ([0, 1, 2, 3, 0, 1, 2, 3], False, 1e-5),    # one face, twice: planar -> homography + LM
    ([4, 5, 6, 7], False, 1e-5),                # one face once: planar with 4 points
    ([0, 1, 2, 4, 7], True, 1e-5),              # 5 points: EPnP, one null vector
])
def test_known_pose_recovery_every_branch(idx, epnp, tol_deg):
    rng = np.random.RandomState(len(idx))
    for _ in range(6):
        V, R, t, uv = _pose(rng)
        ok, r, tt = opnp.solve_pnp_any(V[idx], uv[idx], K, epnp=epnp)
        assert ok
        assert _geo(opnp.rodrigues_to_matrix(r), R) < tol_deg
        assert np.linalg.norm(tt - t) / np.linalg.norm(t) < 1e-7


def test_epnp_four_points_is_only_approximate():
    """Four non-coplanar points leave a 4-dimensional null space of M^T M; EPnP as published (and as cv::epnp implements
    it: linearisations for N = 1, 2, 3 only, then 5 Gauss-Newton steps) then only approximates the pose even on exact
    data -- a property of the algorithm the reference selects for < 6 points (cuboid_pnp_solver.py:162-163), restated
    as is.  Typical reprojection error: a few pixels; every result is finite and in front of the camera."""
    rng = np.random.RandomState(4)
    errs = []
    for _ in range(60):
        V, R, t, uv = _pose(rng)
        idx = [0, 3, 5, 6]
        ok, r, tt = opnp.solve_pnp_epnp(V[idx], uv[idx], K)
        assert ok and np.all(np.isfinite(r)) and tt[2] > 0
        errs.append(np.abs(opnp.project_points(V[idx], r, tt, K) - uv[idx]).max())
    assert np.median(errs) < 6.0, np.median(errs)


@pytest.mark.parametrize("planar", [False, True])
def test_lm_result_is_the_least_squares_optimum(planar):
    """Noisy correspondences: CvLevMarq's answer vs scipy's trust-region minimiser of the same pixel residual, started
    from the true pose (an independent route to the same optimum)."""
    rng = np.random.RandomState(11 + planar)
    idx = [0, 1, 2, 3, 0, 1, 2, 3] if planar else list(range(8)) * 2
    for _ in range(5):
        V, R, t, uv = _pose(rng)
        obs = uv[idx] + rng.randn(len(idx), 2) * 1.0
        ok, r, tt = opnp.solve_pnp_iterative(V[idx], obs, K)
        assert ok

        def res(p):
            return (opnp.project_points(V[idx], p[:3], p[3:], K) - obs).reshape(-1)

        sol = least_squares(res, np.concatenate([opnp.matrix_to_rodrigues(R), t]), method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-14)
        mine = np.concatenate([r, tt])
        assert np.sum(res(mine) ** 2) <= np.sum(sol.fun ** 2) * (1 + 1e-6)
        assert _geo(opnp.rodrigues_to_matrix(r), opnp.rodrigues_to_matrix(sol.x[:3])) < (0.5 if planar else 1e-2)


def test_pnp_shell_packaging_and_visibility_rejects_match_reference():
    """The mirror's pnp_shell / finish_detection against the REFERENCE's pnp_shell on injected solver answers:
    which detections are dropped (>= 3 / >= 6 projected points outside the image by category, none for shoe; centroid
    outside), and the normalised kps_pnp / kps_3d_cam / kps_ori of the kept ones."""
    from centerpose_amd.lib.utils.pnp import cuboid_pnp_shell as shell
    from centerpose_amd.lib.utils.pnp.cuboid_pnp_solver import CuboidPNPSolver

    with open(os.path.join(GOLD, "pnp_shell_ref.json")) as f:
        gold = json.load(f)
    mine = tg.run_shell_cases(shell.pnp_shell, CuboidPNPSolver)
    assert len(mine) == len(gold) == 24
    assert [m is None for m in mine] == [g is None for g in gold]
    assert 5 < sum(g is None for g in gold) < 20
    for m, g in zip(mine, gold):
        if g is None:
            continue
        assert m["bbox_keys"] == g["bbox_keys"]
        for k in ("kps_pnp", "kps_3d_cam", "obj_scale", "kps_ori"):
            np.testing.assert_allclose(m[k], g[k], rtol=1e-12, atol=1e-12, err_msg=k)
