"""The device tracker's logic on the CPU: centerpose_amd/csrc/track_common.h (the scalar functions track.hip's kernels
run) compiled for the host by tests/native/track_host.cpp and compared, frame by frame, with

* tests/golden/tracker_ref.json -- the REFERENCE's own Tracker.step on a seeded video (greedy association, Kalman
  read-out, scale pool, coasting, new ids);
* the reference-pinned Python loop of this repo (lib/detectors/base_detector.py + lib/utils/tracker.py, themselves
  pinned to the reference's run() by tests/golden/track_run.json) on the CenterPoseTrack video, with PnP: every track
  field, the `boxes` selection and the Gaussians drawn into the next frame's pre_hm / pre_hm_hp.
"""
import contextlib
import copy
import ctypes
import io
import json
import os
import subprocess

import numpy as np
import pytest

from centerpose_amd import hip
from centerpose_amd.lib.utils.image import get_affine_transform
from oracle.tools import make_goldens as mg
from oracle.tools import track_golden as tg

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLD = os.path.join(REPO, "tests", "golden")
STRIDE = 520
TR = dict(ID=0, AGE=1, ACTIVE=2, FLAGS=3, POST=4, FUS_MEAN=124, FUS_STD=140, LOC=156, QUAT=159, PROJ=163, KPS_PNP=179,
          KPS_3D=197, KPS_ORI=224, KF_X=242, KF_P=274, MEAN_KF=409, STD_KF=425, SCALE_KF=441, SCALE_UNC_KF=444, CONF=447,
          KPS_PNP_KF=455, KPS_3D_KF=473, KPS_ORI_KF=500)


Params = hip.TrackParams   # cp_track_params of include/centerpose_hip.h (same layout as track_common.h's TrackParams)


@pytest.fixture(scope="module")
def host():
    out = os.path.join(REPO, "tests", "_build", "libcp_track_host.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-ffp-contract=off", os.path.join(REPO, "tests", "native", "track_host.cpp"),
                    "-o", out], check=True)
    L = ctypes.CDLL(out)
    assert L.cp_track_host_stride() == STRIDE
    assert L.cp_track_host_params_bytes() == ctypes.sizeof(Params)
    L.cp_track_host_update.restype = ctypes.c_int
    return L


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class HostTracker(object):
    """One video's track table driven through the harness, the PnP in between supplied by `solve(pts, scale) -> rows`."""

    def __init__(self, L, params, vm, solve=None, cap=128):
        self.L, self.P, self.vm, self.solve, self.cap = L, params, np.ascontiguousarray(vm, np.float64), solve, cap
        self.P.cap = cap
        self.prev = np.zeros((cap, STRIDE))
        self.np_, self.id_count = 0, ctypes.c_int(0)

    def step(self, post, det_rows=None):
        post = np.ascontiguousarray(post, np.float64).reshape(-1, 120)
        nxt = np.zeros((self.cap, STRIDE))
        pts = np.zeros((self.cap, 16), np.float32)
        sc = np.zeros((self.cap, 3), np.float32)
        rows = None if det_rows is None else np.ascontiguousarray(det_rows, np.float64)
        n = self.L.cp_track_host_update(ctypes.byref(self.P), _ptr(self.vm), _ptr(post), len(post), _ptr(rows),
                                        _ptr(self.prev), self.np_, ctypes.byref(self.id_count), _ptr(nxt), _ptr(pts), _ptr(sc))
        assert n >= 0
        kf_rows = None
        if self.solve is not None and n:
            kf_rows = np.ascontiguousarray(self.solve(pts[:n].reshape(n, 8, 2), sc[:n]), np.float64)
        recs = np.zeros((max(n, 1), 9, 5))
        self.L.cp_track_host_finish(ctypes.byref(self.P), _ptr(self.vm), _ptr(nxt), n, _ptr(kf_rows), 0, 1, _ptr(recs))
        self.prev, self.np_ = nxt, n
        return nxt[:n], recs[:n]


def _post_from_dict(d, fusion_as_displacement=False):
    r = np.zeros(120)
    for k, (off, w) in hip.POST_FIELDS.items():
        if k in d:
            r[off:off + w] = np.asarray(d[k], np.float64).reshape(-1)
    if fusion_as_displacement:  # hand-made detections that only carry the fused estimate: heat-map side "missing"
        r[64:80] = np.asarray(d["kps_fusion_mean"], np.float64)
        r[8:24] = np.asarray(d["kps_fusion_std"], np.float64)
        r[80:96] = -1.0
    return r


@pytest.mark.parametrize("mode", mg.TRACKER_MODES)
def test_tracker_logic_matches_reference_golden(host, mode):
    """The device tracker's logic (host build of track_common.h) against the REFERENCE's own runs on the seeded video:
    Tracker.step and Tracker_baseline.step (--refined_Kalman: utils/tracker_baseline.py -- position-only filter with its
    broadcast initial covariance, plain scale average, raw centres against velocity-advanced track centres, the P[2v]
    read-out); greedy, and the optimal assignment of tracker.py:154-174 behind either solver: "*hungarian" = scikit-learn
    0.22.2's Munkres, the reference's pinned dependency (cp_track_params.hungarian = 1, trk_munkres), "*_scipy" = scipy's
    rectangular LSAP (2, trk_lsap); "ties_*" = the degenerate video on which the two optima hand out coasting slots and ids in
    different orders (make_goldens.tracker_frames_ties)."""
    with open(os.path.join(GOLD, "tracker_ref.json")) as f:
        gold = json.load(f)[mode]
    frames, hung, baseline, _ = mg.tracker_mode(mode)
    o = mg.TrackOpt(bool(hung))
    P = Params(new_thresh=o.new_thresh, pre_thresh=0.3, R=o.R, conf_lo=3, conf_hi=9, max_age=o.max_age, kalman=1,
               scale_pool=1, use_pnp=0, hps_uncertainty=1, show_axes=0, cat_rule=0, render_hm_mode=1, render_hmhp_mode=2,
               pre_hm=1, pre_hm_hp=1, K=100, hungarian=hung, baseline=int(baseline))
    vm = np.zeros(16)
    vm[[0, 4]] = 1.0
    vm[6:10] = 512
    ht = HostTracker(host, P, vm)
    for f, dets in enumerate(frames):
        post = np.stack([_post_from_dict(d, True) for d in dets])
        tracks, _ = ht.step(post)
        assert len(tracks) == len(gold[f]), f
        for t, g in zip(tracks, gold[f]):
            assert (int(t[0]), int(t[1]), int(t[2])) == (g["tracking_id"], g["age"], g["active"]), f
            np.testing.assert_allclose(t[4 + 28:4 + 30], g["ct"], rtol=1e-12)
            np.testing.assert_allclose(t[TR["MEAN_KF"]:TR["MEAN_KF"] + 16], g["kps_mean_kf"], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(t[TR["STD_KF"]:TR["STD_KF"] + 16], g["kps_std_kf"], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(t[TR["SCALE_KF"]:TR["SCALE_KF"] + 3], g["obj_scale_kf"], rtol=1e-6)
            np.testing.assert_allclose(t[TR["SCALE_UNC_KF"]:TR["SCALE_UNC_KF"] + 3], g["obj_scale_uncertainty_kf"], rtol=1e-6,
                                       atol=1e-12)
    ids = [t[0] for t in tracks]
    assert len(set(ids)) == len(ids)


def _vm(meta):
    v = np.zeros(16)
    v[0:6] = np.asarray(meta["trans_input"], np.float64).reshape(-1)
    v[6:10] = [meta["width"], meta["height"], meta["inp_width"], meta["inp_height"]]
    K = np.asarray(meta["camera_matrix"], np.float64)
    v[10:14] = [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]
    return v


def test_tracker_logic_matches_python_loop_with_pnp(host, monkeypatch):
    """The CenterPoseTrack video of tests/golden/track_run.json through the Python loop (pinned to the reference's
    run()) and, side by side, through the harness: every frame's detections (captured at `Tracker.step`) -> stages 1-5."""
    from tests import test_tracking_loop as ttl
    from centerpose_amd.lib.detectors import base_detector as bd
    from centerpose_amd.lib.utils.pnp import cuboid_pnp_solver as cps

    det = ttl._detector(monkeypatch, "-1")
    det.process = ttl._cpu_process(det)
    monkeypatch.setattr(cps, "solve_pnp_batch", ttl._oracle_pnp_rows)
    monkeypatch.setattr(bd, "solve_pnp_batch", ttl._oracle_pnp_rows)
    o = det.opt
    frames_in = tg.frame_inputs(get_affine_transform)
    meta0 = frames_in[0][1]
    Kmat = np.asarray(meta0["camera_matrix"], np.float64)

    def solve(pts, scales):
        return ttl._oracle_pnp_rows([p for p in pts], [s for s in scales], Kmat)

    P = Params(new_thresh=o.new_thresh, pre_thresh=o.pre_thresh, R=o.R, conf_lo=o.conf_border[o.c][0],
               conf_hi=o.conf_border[o.c][1], max_age=o.max_age, kalman=int(o.kalman), scale_pool=int(o.scale_pool),
               use_pnp=int(o.use_pnp), hps_uncertainty=int(o.hps_uncertainty), show_axes=int(o.show_axes), cat_rule=0,
               render_hm_mode=o.render_hm_mode, render_hmhp_mode=o.render_hmhp_mode, pre_hm=int(o.pre_hm),
               pre_hm_hp=int(o.pre_hm_hp), K=100)
    ht = HostTracker(host, P, _vm(meta0), solve)
    captured = {}
    real_step = det.tracker.step

    def spy(results, boxes=[]):
        captured["results"] = copy.deepcopy(list(results))
        return real_step(results, boxes)

    det.tracker.step = spy
    n_checked = 0
    prev_recs = None
    with contextlib.redirect_stdout(io.StringIO()):
        for f, (img, meta) in enumerate(frames_in):
            if prev_recs is not None:
                # the Gaussians the Python loop is about to draw from its tracks == the harness's records of last frame
                hm, hp, _ = det._track_records(det.tracker.tracks, dict(meta, id=f), True, True)
                mine_hm = [tuple(r[0]) for r in prev_recs if r[0, 0] >= 0]
                mine_hp = [tuple(q) for r in prev_recs for q in r[1:] if q[0] >= 0]
                assert len(hm) == len(mine_hm) and len(hp) == len(mine_hp), (f, len(hp), len(mine_hp))
                for a, b in zip(hm, mine_hm):
                    np.testing.assert_allclose(np.array(a, float), np.array(b), rtol=1e-9)
                for a, b in zip(hp, mine_hp):
                    assert int(a[0]) == int(b[0]) - 1 and (a[1], a[2], a[3]) == (b[1], b[2], b[3])
                    np.testing.assert_allclose(float(a[4]), b[4], rtol=1e-6)
            ret = det.run(img, meta_inp=copy.deepcopy(meta), preprocessed_flag=True)
            res = captured["results"]
            post = np.stack([_post_from_dict(d) for d in res]) if len(res) else np.zeros((0, 120))
            pts = [det._pnp_points(d) for d in res]
            scales = [np.asarray(d["obj_scale"], np.float64) / d["obj_scale"][1] for d in res]
            rows = ttl._oracle_pnp_rows(pts, scales, Kmat) if len(res) else None
            tracks, recs = ht.step(post, rows)
            prev_recs = recs
            assert len(tracks) == len(ret["results"]), f
            n_box = 0
            for t, g in zip(tracks, ret["results"]):
                assert (int(t[0]), int(t[1]), int(t[2])) == (g["tracking_id"], g["age"], g["active"]), f
                flags = int(t[3])
                n_box += (flags >> 2) & 1
                chk = [("MEAN_KF", "kps_mean_kf", 16), ("STD_KF", "kps_std_kf", 16), ("SCALE_KF", "obj_scale_kf", 3),
                       ("SCALE_UNC_KF", "obj_scale_uncertainty_kf", 3), ("FUS_MEAN", "kps_fusion_mean", 16),
                       ("FUS_STD", "kps_fusion_std", 16)]
                if "kps_pnp_kf" in g:
                    assert flags & 2
                    chk += [("KPS_PNP_KF", "kps_pnp_kf", 18), ("KPS_3D_KF", "kps_3d_cam_kf", 27), ("KPS_ORI_KF", "kps_ori_kf", 18)]
                else:
                    assert not flags & 2
                if "location" in g:
                    chk += [("LOC", "location", 3), ("QUAT", "quaternion_xyzw", 4), ("KPS_PNP", "kps_pnp", 18),
                            ("KPS_3D", "kps_3d_cam", 27)]
                for name, key, n in chk:
                    tol = 2e-7 if name in ("FUS_MEAN", "FUS_STD") else 1e-6  # the fusion restates the reference's float32 steps
                    np.testing.assert_allclose(t[TR[name]:TR[name] + n], np.asarray(g[key], np.float64).reshape(-1),
                                               rtol=tol, atol=tol, err_msg="frame %d %s" % (f, key))
                n_checked += 1
            assert n_box == len(ret["boxes"]), f
    assert n_checked >= 8


@pytest.mark.parametrize("hungarian,baseline", [(False, False), (True, False), (False, True), (True, True)])
def test_tracker_logic_random_scenarios_vs_python_tracker(host, hungarian, baseline):
    """(greedy, and the Hungarian association of tracker.py:154-174: both sides on scikit-learn 0.22.2's Munkres, restated.)  Crowded random videos (objects crossing, leaving, re-entering, weak detections, same-frame births and deaths) through
    the harness and through the reference-pinned Python ``Tracker`` (greedy, Kalman + scale pool, no PnP): identical ids,
    ages, activity and filter read-outs in every frame -- association order, coasting up to max_age, the new-track
    threshold and the float32 cost arithmetic included."""
    from centerpose_amd.lib.utils.tracker import Tracker, Tracker_baseline

    # (baseline: the reference-pinned Python Tracker_baseline, --refined_Kalman)
    class Opt(mg.TrackOpt):
        max_age = 3
        new_thresh = 0.35

    for seed in (1, 2, 3):
        rng = np.random.RandomState(seed)
        n_obj, n_frames = 14, 25
        base = rng.uniform(60, 450, (n_obj, 2))
        vel = rng.uniform(-9, 9, (n_obj, 2))
        size = rng.uniform(25, 80, n_obj)
        o = Opt(hungarian)
        py = (Tracker_baseline if baseline else Tracker)(o)
        py.init_track({"id": 0})
        P = Params(new_thresh=o.new_thresh, pre_thresh=0.3, R=o.R, conf_lo=3, conf_hi=9, max_age=o.max_age, kalman=1,
                   scale_pool=1, use_pnp=0, hps_uncertainty=1, show_axes=0, cat_rule=0, render_hm_mode=1, render_hmhp_mode=2,
                   pre_hm=1, pre_hm_hp=1, K=100, hungarian=int(hungarian), baseline=int(baseline))
        vm = np.zeros(16)
        vm[[0, 4]] = 1.0
        vm[6:10] = 512
        ht = HostTracker(host, P, vm)
        for f in range(n_frames):
            dets = []
            for i in rng.permutation(n_obj):
                if rng.rand() < 0.2:      # missed this frame
                    continue
                ct = base[i] + vel[i] * f + rng.randn(2) * 1.5
                kps = ct[None, :] + rng.uniform(-0.5, 0.5, (8, 2)) * size[i]
                dets.append({
                    "score": float(rng.uniform(0.2, 0.99)), "cls": 0,
                    "bbox": [ct[0] - size[i] / 2, ct[1] - size[i] / 2, ct[0] + size[i] / 2, ct[1] + size[i] / 2],
                    "ct": [float(ct[0]), float(ct[1])],
                    "tracking": (-vel[i] + rng.randn(2) * 2.0).astype(np.float32),
                    "tracking_hp": (np.tile(-vel[i], 8) + rng.randn(16) * 0.5).astype(np.float32),
                    "kps": kps.reshape(-1).astype(np.float32),
                    "kps_fusion_mean": kps.reshape(-1) + rng.randn(16) * 0.3,
                    "kps_fusion_std": rng.uniform(0.4, 6.0, 16),
                    "obj_scale": rng.uniform(0.5, 1.5, 3).astype(np.float32),
                    "obj_scale_uncertainty": rng.uniform(0.05, 0.3, 3).astype(np.float32)})
            post = np.stack([_post_from_dict(d, True) for d in dets]) if dets else np.zeros((0, 120))
            mine, _ = ht.step(post)
            theirs, _ = py.step(copy.deepcopy(dets))
            assert len(mine) == len(theirs), (seed, f)
            for t, g in zip(mine, theirs):
                assert (int(t[0]), int(t[1]), int(t[2])) == (g["tracking_id"], g["age"], g["active"]), (seed, f)
                np.testing.assert_allclose(t[4 + 28:4 + 30], g["ct"], rtol=1e-12)
                np.testing.assert_allclose(t[TR["MEAN_KF"]:TR["MEAN_KF"] + 16], np.asarray(g["kps_mean_kf"]).reshape(-1),
                                           rtol=1e-8, atol=1e-8)
                np.testing.assert_allclose(t[TR["STD_KF"]:TR["STD_KF"] + 16], g["kps_std_kf"], rtol=1e-8, atol=1e-8)
                np.testing.assert_allclose(t[TR["SCALE_KF"]:TR["SCALE_KF"] + 3], g["obj_scale_kf"], rtol=1e-6)
        assert py.id_count >= n_obj  # tracks were lost and re-born along the way


def test_tracker_logic_truncates_and_reports_overflow(host):
    """More live tracks than the table holds: the association keeps the first `cap` entries in the reference's order (here:
    new tracks by score), spends no id on the detections it drops and reports how many it dropped -- the device adds that to
    the video's sticky counter (cp_track_status / `DeviceTracker.check`); round 3 emptied the whole list instead."""
    o = mg.TrackOpt(False)
    P = Params(new_thresh=0.3, pre_thresh=0.3, R=o.R, conf_lo=3, conf_hi=9, max_age=5, kalman=1, scale_pool=1, use_pnp=0,
               hps_uncertainty=1, show_axes=0, cat_rule=0, render_hm_mode=1, render_hmhp_mode=2, pre_hm=1, pre_hm_hp=1, K=100)
    vm = np.zeros(16)
    vm[[0, 4]] = 1.0
    vm[6:10] = 512
    dets = mg.tracker_frames()[0]
    strong = [d for d in dets if d["score"] > 0.3]
    cap = len(strong) - 1
    ht = HostTracker(host, P, vm, cap=cap)
    post = np.stack([_post_from_dict(d, True) for d in dets])
    nxt = np.zeros((cap, STRIDE))
    pts = np.zeros((cap, 16), np.float32)
    sc = np.zeros((cap, 3), np.float32)
    host.cp_track_host_last_dropped.restype = ctypes.c_int
    n = host.cp_track_host_update(ctypes.byref(ht.P), _ptr(ht.vm), _ptr(post), len(post), None, _ptr(ht.prev), 0,
                                  ctypes.byref(ht.id_count), _ptr(nxt), _ptr(pts), _ptr(sc))
    assert n == cap and host.cp_track_host_last_dropped() == 1
    assert ht.id_count.value == cap                               # no id for the dropped detection
    assert sorted(nxt[:, TR["ID"]].tolist()) == list(range(1, cap + 1))
    kept = sorted(float(v) for v in nxt[:, TR["POST"] + 0])       # field 0 of the post record = score
    want = sorted(float(d["score"]) for d in strong)[1:]          # the weakest of the strong detections is the one dropped
    np.testing.assert_allclose(kept, want, rtol=1e-6)


def test_assignment_equals_scipy_linear_sum_assignment(host):
    """trk_lsap (track_common.h, compiled for the host) against scipy.optimize.linear_sum_assignment -- the solver the
    Hungarian goldens come from (sklearn 0.22's linear_assignment, which the reference imports, is absent; both return the
    optimum).  Same pairs, not just the same cost: continuous costs, tie-heavy small integers, constant matrices, tall and wide
    shapes, and the tracker's own pattern (float32 distances + 1e18 for forbidden pairs, clamped), where the duals swallow the
    low bits of the real costs and only the same operations in the same order reproduce scipy's choice."""
    from scipy.optimize import linear_sum_assignment

    host.cp_track_host_lsap.restype = None
    rng = np.random.RandomState(11)
    n_cases = 0
    for trial in range(1500):
        nd, nt = int(rng.randint(1, 14)), int(rng.randint(1, 14))
        kind = trial % 5
        if kind == 0:
            c = rng.rand(nd, nt) * 100.0
        elif kind == 1:
            c = rng.randint(0, 4, (nd, nt)).astype(np.float64)            # many ties
        elif kind == 2:
            c = np.full((nd, nt), float(rng.randint(0, 3)))                  # constant: scipy returns the identity pattern
        else:  # the tracker's matrix: squared float32 distances, forbidden pairs at exactly 1e18
            d32 = (rng.rand(nd, nt) * (2000.0 if kind == 3 else 90.0)).astype(np.float32)
            bad = rng.rand(nd, nt) < (0.6 if kind == 3 else 0.3)
            c = d32 + bad * 1e18
            c[c > 1e18] = 1e18
        c = np.ascontiguousarray(c, np.float64)
        r, col = linear_sum_assignment(c)
        want = -np.ones(nd, np.int32)
        want[r] = col
        got = np.zeros(nd, np.int32)
        host.cp_track_host_lsap(_ptr(c), nd, nt, _ptr(got))
        assert np.array_equal(got, want), (trial, kind, nd, nt, got.tolist(), want.tolist())
        n_cases += 1
    # a few large ones (the device's upper sizes: 100 detections x 128 tracks)
    for nd, nt in ((100, 128), (128, 100), (128, 128)):
        d32 = (rng.rand(nd, nt) * 5000.0).astype(np.float32)
        c = np.ascontiguousarray(d32 + (rng.rand(nd, nt) < 0.9) * 1e18)
        c[c > 1e18] = 1e18
        r, col = linear_sum_assignment(c)
        want = -np.ones(nd, np.int32)
        want[r] = col
        got = np.zeros(nd, np.int32)
        host.cp_track_host_lsap(_ptr(c), nd, nt, _ptr(got))
        assert np.array_equal(got, want), (nd, nt)
    assert n_cases == 1500


def test_munkres_equals_the_restatement_of_sklearn_022_and_is_optimal(host):
    """trk_munkres (track_common.h, host build; cp_track_params.hungarian = 1) against oracle/munkres.py -- the numpy restatement
    of scikit-learn 0.22.2's `linear_assignment` (the reference's pinned, no longer installable dependency: PARITY UNPINNED
    against the real module).  Same PAIRS on every matrix (continuous, tie-heavy, constant, the tracker's 1e18 pattern, all
    forbidden; tall, wide, square, empty sides), through both entry points (the test harness and the product's
    cp_linear_assignment); and the structural facts that hold for the real module whatever its tie-breaking: every row of the
    shorter side is assigned once, no column twice, and the total cost is scipy's optimum."""
    from scipy.optimize import linear_sum_assignment

    from centerpose_amd import hip as _hip
    from oracle.munkres import linear_assignment as munkres_ref

    host.cp_track_host_munkres.restype = None
    rng = np.random.RandomState(5)
    differs_from_scipy = 0
    for trial in range(2000):
        nd, nt = int(rng.randint(1, 15)), int(rng.randint(1, 15))
        kind = trial % 6
        if kind == 0:
            c = rng.rand(nd, nt) * 100.0
        elif kind == 1:
            c = rng.randint(0, 4, (nd, nt)).astype(np.float64)
        elif kind == 2:
            c = np.full((nd, nt), float(rng.randint(0, 3)))
        elif kind == 5:
            c = np.full((nd, nt), 1e18)
        else:
            d32 = (rng.rand(nd, nt) * (2000.0 if kind == 3 else 90.0)).astype(np.float32)
            c = d32 + (rng.rand(nd, nt) < (0.6 if kind == 3 else 0.9)) * 1e18
            c[c > 1e18] = 1e18
        c = np.ascontiguousarray(c, np.float64)
        want = munkres_ref(c)
        got = np.zeros(nd, np.int32)
        host.cp_track_host_munkres(_ptr(c), nd, nt, _ptr(got))
        pairs = [[i, int(got[i])] for i in range(nd) if got[i] >= 0]
        assert pairs == want.tolist(), (trial, kind, nd, nt)
        assert _hip.linear_assignment(c, 1).tolist() == pairs
        assert len(pairs) == min(nd, nt) and len({p[1] for p in pairs}) == len(pairs)
        r, col = linear_sum_assignment(c)
        assert np.isclose(sum(c[i, j] for i, j in pairs), c[r, col].sum(), rtol=1e-12, atol=1e-9)
        differs_from_scipy += int(sorted(map(tuple, pairs)) != sorted(zip(r.tolist(), col.tolist())))
    assert differs_from_scipy > 50   # the two optima really are different objects: that is why both are restated
    for nd, nt in ((100, 128), (128, 100), (0, 5), (5, 0)):   # the device's upper sizes (seconds, not minutes), empty sides
        d32 = (rng.rand(nd, nt) * 5000.0).astype(np.float32)
        c = np.ascontiguousarray(d32 + (rng.rand(nd, nt) < 0.9) * 1e18)
        c[c > 1e18] = 1e18
        assert _hip.linear_assignment(c, 1).tolist() == munkres_ref(c).tolist(), (nd, nt)
    with pytest.raises(RuntimeError):
        _hip.linear_assignment(np.zeros((2, 2)), 3)
