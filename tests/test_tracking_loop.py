"""CenterPoseTrack's per-frame loop: the detector mirror's ``run()`` with ``tracking_task`` against
tests/golden/track_run.json -- the output of the REFERENCE's own ``ObjectPoseDetector.run`` (base_detector.py:390-772:
previous-frame heat-maps rendered from the tracks, Gaussian fusion, ``Tracker.step``, filtered PnP) on the seeded
synthetic video of tests/scene.render_video, produced by oracle/tools/track_golden.py (which lists what was real and
what was substituted when the reference ran).

CPU test: the loop, the record builder for the previous-frame render, the fusion, the tracker hand-over and the output
schema, with the oracle's decode and PnP injected in place of the two device stages.
GPU test: the same video through cp_decode, cp_postprocess-free host merge, cp_pnp_solve and cp_render_gaussians."""
import contextlib
import copy
import io
import json
import os

import numpy as np
import pytest
import torch

from centerpose_amd import hip
from centerpose_amd.lib.opts import opts
from centerpose_amd.lib.utils.image import get_affine_transform
from oracle import decode as odec
from oracle import pnp as opnp
from oracle.tools import track_golden as tg

GOLD = os.path.join(os.path.dirname(__file__), "golden", "track_run.json")
# the same video through the reference's run() with --refined_Kalman on top (Tracker_baseline inside the two-frame loop)
GOLD_BASELINE = os.path.join(os.path.dirname(__file__), "golden", "track_run_baseline.json")


def _gold(baseline):
    with open(GOLD_BASELINE if baseline else GOLD) as fh:
        return json.load(fh)["frames"]


def _opt(gpus, baseline=False):
    argv = [a if a != "-1" else gpus for a in (tg.BASELINE_ARGV if baseline else tg.TRACK_ARGV)]
    with contextlib.redirect_stdout(io.StringIO()):
        o = opts().parser.parse_args(argv)
        o = tg.demo_flags(o)
        o = opts().init(opts().parse(o))
    return o


def _oracle_pnp_rows(points_list, scales, camera_matrices, device=None):
    """[N, 40] result rows of cp_pnp_solve (include/centerpose_hip.h) computed by the float64 oracle."""
    rows = np.zeros((len(points_list), hip.PNP_STRIDE))
    cams = np.asarray(camera_matrices, np.float64)
    for i, (pts, sc) in enumerate(zip(points_list, scales)):
        K = cams if cams.ndim == 2 else cams[i]
        pts = np.asarray(pts, np.float64).reshape(-1, 2)
        cv_ = opnp.solve_cuboid_pnp(pts, np.asarray(sc, np.float64), K, opencv_return=True)
        gl = opnp.solve_cuboid_pnp(pts, np.asarray(sc, np.float64), K, opencv_return=False)
        if cv_ is None:
            rows[i, 0] = 2 if int(((pts[:, 0] > -5000) & (pts[:, 1] > -5000)).sum()) >= 4 else -1
            continue
        rows[i, 0] = 1
        rows[i, 1:4], rows[i, 4:7], rows[i, 7] = cv_["rvec"], cv_["tvec"], cv_["reproj_err"]
        rows[i, 8:24] = cv_["projected_points"].reshape(-1)
        rows[i, 24:28] = cv_["quaternion_xyzw"]
        rows[i, 28:31], rows[i, 31:35] = gl["location"], gl["quaternion_xyzw"]
    return rows


class _Engine(object):
    """Stands in for HipPoseNet._engine(): the stub network's heads, sigmoid applied as the engine does."""

    def __init__(self, stub, device):
        self.stub, self.device = stub, device

    def forward(self, images, pre_images=None, pre_hms=None, pre_hm_hp=None, sigmoid_hm=True):
        z = self.stub(images, pre_images, pre_hms, pre_hm_hp)[-1]
        out = {}
        for k, v in z.items():
            v = torch.sigmoid(v) if (sigmoid_hm and k in ("hm", "hm_hp")) else v
            out[k] = v.to(self.device)
        return out


def _detector(monkeypatch, gpus, baseline=False):
    from centerpose_amd.lib.detectors import base_detector as bd
    from centerpose_amd.lib.detectors.object_pose import ObjectPoseDetector

    stub = tg.StubNetwork(tg.video_heads())
    monkeypatch.setattr(bd, "create_model", lambda *a, **k: stub)
    monkeypatch.setattr(bd, "load_model", lambda m, *a, **k: m)
    with contextlib.redirect_stdout(io.StringIO()):
        det = ObjectPoseDetector(_opt(gpus, baseline))
    stub._engine = lambda: _Engine(stub, det.opt.device)
    return det


def _cpu_process(det):
    """``process`` with the oracle's numpy decode in place of cp_decode (the rest of run() is the product code)."""
    def process(images, pre_images=None, pre_hms=None, pre_hm_hp=None, pre_inds=None, return_time=False):
        import time

        z = det.model._engine().forward(images, pre_images, pre_hms, pre_hm_hp, sigmoid_hm=True)
        n = {k: v.numpy() for k, v in z.items()}
        o = det.opt
        dets = odec.object_pose_decode(n["hm"], n["hps"], wh=n["wh"], kps_displacement_std=n["hps_uncertainty"],
                                       obj_scale=n["scale"], obj_scale_uncertainty=n["scale_uncertainty"], reg=n["reg"],
                                       hm_hp=n["hm_hp"], hp_offset=n["hp_offset"], tracking=n["tracking"],
                                       tracking_hp=n["tracking_hp"], K=o.K, rep_mode=o.rep_mode, tracking_task=True,
                                       balance_coefficient=o.balance_coefficient[o.c])
        return (z, dets, time.time()) if return_time else (z, dets)
    return process


def _compare(frames, gold):
    assert len(frames) == len(gold)
    for f, (a, g) in enumerate(zip(frames, gold)):
        assert a["keys"] == g["keys"], f                                  # P6 return schema
        assert a["n_boxes"] == g["n_boxes"], f
        assert len(a["tracks"]) == len(g["tracks"]), f
        # previous-frame inputs rendered from the tracks (zero on the first frame)
        assert a["pre_hm_hp_nonzero"] == g["pre_hm_hp_nonzero"], f
        np.testing.assert_allclose(a["pre_hm_sum"], g["pre_hm_sum"], rtol=1e-5, atol=1e-4, err_msg="frame %d" % f)
        np.testing.assert_allclose(a["pre_hm_max"], g["pre_hm_max"], rtol=1e-6, err_msg="frame %d" % f)
        np.testing.assert_allclose(a["pre_hm_hp_sum"], g["pre_hm_hp_sum"], rtol=1e-5, atol=1e-4, err_msg="frame %d" % f)
        for ta, tg_ in zip(a["tracks"], g["tracks"]):
            assert (ta["tracking_id"], ta["age"], ta["active"]) == (tg_["tracking_id"], tg_["age"], tg_["active"]), f
            assert set(ta) == set(tg_), (f, set(ta) ^ set(tg_))
            for k in tg_:
                if k in ("tracking_id", "age", "active"):
                    continue
                tol = dict(rtol=1e-5, atol=2e-3) if k in ("kps_3d_cam_kf",) else dict(rtol=1e-5, atol=1e-3)
                np.testing.assert_allclose(ta[k], tg_[k], err_msg="frame %d %s" % (f, k), **tol)
        for ba, bg in zip(a["boxes_kps_pnp"], g["boxes_kps_pnp"]):
            np.testing.assert_allclose(ba, bg, rtol=1e-5, atol=1e-5, err_msg="frame %d box" % f)


@pytest.mark.parametrize("baseline", [False, True])
def test_tracking_run_matches_reference_cpu(monkeypatch, baseline):
    from centerpose_amd.lib.utils.pnp import cuboid_pnp_solver as cps

    gold = _gold(baseline)   # baseline: the reference's run() with --refined_Kalman on top (Tracker_baseline in the loop)
    det = _detector(monkeypatch, "-1", baseline)
    assert type(det.tracker).__name__ == ("Tracker_baseline" if baseline else "Tracker")
    det.process = _cpu_process(det)
    monkeypatch.setattr(cps, "solve_pnp_batch", _oracle_pnp_rows)
    from centerpose_amd.lib.detectors import base_detector as bd
    monkeypatch.setattr(bd, "solve_pnp_batch", _oracle_pnp_rows)
    with contextlib.redirect_stdout(io.StringIO()):
        frames = tg.run_video(det, get_affine_transform)
    _compare(frames, gold)
    assert len(gold[-1]["tracks"]) == 2 and gold[1]["pre_hm_hp_nonzero"] > 0   # the golden itself is not degenerate
    det.reset_tracking()
    assert det.tracker.tracks == [] and det.pre_images is None


def test_refined_kalman_baseline_runs_and_differs(monkeypatch):
    """``--refined_Kalman`` (Tracker_baseline): position-only filter, plain scale average; schema as the tracking run."""
    from centerpose_amd.lib.detectors import base_detector as bd
    from centerpose_amd.lib.utils.pnp import cuboid_pnp_solver as cps
    from centerpose_amd.lib.utils.tracker import Tracker_baseline

    det = _detector(monkeypatch, "-1")
    det.opt.tracking_task = False
    det.opt.refined_Kalman = True
    det.tracker = Tracker_baseline(det.opt)
    det.process = _cpu_process(det)
    monkeypatch.setattr(cps, "solve_pnp_batch", _oracle_pnp_rows)
    monkeypatch.setattr(bd, "solve_pnp_batch", _oracle_pnp_rows)
    with contextlib.redirect_stdout(io.StringIO()):
        frames = tg.run_video(det, get_affine_transform)
    assert [len(f["tracks"]) for f in frames] == [2] * 5
    assert frames[-1]["tracks"][0]["obj_scale_uncertainty_kf"] == [0.0, 0.0, 0.0]
    assert frames[2]["pre_hm_sum"] is None   # no previous-frame inputs without tracking_task


@pytest.mark.gpu
@pytest.mark.parametrize("baseline", [False, True])
def test_tracking_run_matches_reference_gpu(device, monkeypatch, baseline):
    gold = _gold(baseline)
    det = _detector(monkeypatch, "0", baseline)
    assert det.opt.device.type == "cuda"
    with contextlib.redirect_stdout(io.StringIO()):
        frames = tg.run_video(det, get_affine_transform)
    _compare(frames, gold)


@pytest.mark.gpu
@pytest.mark.parametrize("device_tracker,baseline", [(False, False), (True, False), (False, True), (True, True)])
def test_batched_tracking_matches_reference_per_video(device, monkeypatch, device_tracker, baseline):
    """BatchedTracking (B concurrent videos: batched render / network / decode / post-process / PnP on the device; the
    per-video track bookkeeping either by the reference-shaped Python Tracker on the host or by cp_track_step on the
    device) must give every video exactly what the reference's ``run()`` gives the single video of
    tests/golden/track_run.json: tracks, filter read-outs, filtered poses, `boxes`, and the previous-frame heat-maps the
    network is fed.  The same for the reference's run() with --refined_Kalman (tests/golden/track_run_baseline.json)."""
    import types

    from centerpose_amd.lib.detectors.batch_tracking import BatchedTracking

    gold = _gold(baseline)   # baseline: --refined_Kalman on top, i.e. Tracker_baseline (cp_track_params.baseline on the device)
    det = _detector(monkeypatch, "0", baseline)
    stub, B = det.model, 3

    class BatchEngine(object):
        calls = 0
        last_pre_hm = last_pre_hm_hp = None

        def forward(self, images, pre_images=None, pre_hms=None, pre_hm_hp=None, sigmoid_hm=True):
            assert images.shape[0] == B and pre_images is not None and pre_images.shape == images.shape
            self.last_pre_hm, self.last_pre_hm_hp = pre_hms.detach().cpu().clone(), pre_hm_hp.detach().cpu().clone()
            h = stub.frames[self.calls]
            self.calls += 1
            out = {}
            for k, v in h.items():
                v = torch.sigmoid(v) if (sigmoid_hm and k in ("hm", "hm_hp")) else v
                out[k] = v.repeat(B, 1, 1, 1).contiguous().to(device)
            return out

    eng = BatchEngine()
    stub._engine = lambda: eng
    bt = BatchedTracking(det, B, device_tracker=device_tracker)
    frames = [[] for _ in range(B)]
    with contextlib.redirect_stdout(io.StringIO()):
        for img, meta in tg.frame_inputs(get_affine_transform):
            images = torch.from_numpy(img)[None].repeat(B, 1, 1, 1)
            outs = bt.step(images, [copy.deepcopy(meta) for _ in range(B)])
            for b in range(B):
                view = types.SimpleNamespace(model=types.SimpleNamespace(
                    last_pre_hm=eng.last_pre_hm[b:b + 1], last_pre_hm_hp=eng.last_pre_hm_hp[b:b + 1]))
                s = tg.summarise({"results": outs[b]["results"], "boxes": outs[b]["boxes"]}, view)
                s["keys"] = gold[len(frames[b])]["keys"]   # run()'s timing keys are not part of a batched step
                frames[b].append(s)
    for b in range(B):
        _compare(frames[b], gold)
    assert bt.times["steps"] == len(gold) and (device_tracker or bt.times["host_tracks"] > 0)
    bt.reset()
    assert all(t.tracks == [] for t in bt.trackers) and bt.pre_images is None
    if device_tracker:
        assert all(len(a) == 0 for a in bt.dev.read())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["greedy", "hungarian", "hungarian_scipy", "baseline", "baseline_hungarian", "ties_greedy",
                                  "ties_hungarian", "ties_hungarian_scipy", "ties_baseline_hungarian"])
def test_device_tracker_matches_reference_tracker_golden(device, mode):
    """cp_track_step alone (no PnP) on the seeded detections of tests/golden/tracker_ref.json -- the REFERENCE's own
    Tracker.step: ids, ages, coasting, Kalman read-out and scale pool, for three videos that start one frame apart; greedy
    association and the Hungarian mode behind either solver ("*hungarian": scikit-learn 0.22.2's Munkres, the reference's pinned
    dependency, restated as trk_munkres; "*_scipy": scipy's rectangular assignment, trk_lsap); "baseline*" = the reference's
    Tracker_baseline.step (--refined_Kalman) on the same detections; "ties_*" = the degenerate video on which the two optima
    order coasting tracks and new ids differently (make_goldens.tracker_frames_ties: up to 21 tracks, two classes)."""
    from oracle.tools import make_goldens as mg

    with open(os.path.join(os.path.dirname(GOLD), "tracker_ref.json")) as fh:
        gold = json.load(fh)[mode]
    frames, hung, baseline, _ = mg.tracker_mode(mode)
    o = mg.TrackOpt(bool(hung))
    B, K = 3, 100
    P = hip.TrackParams(new_thresh=o.new_thresh, pre_thresh=0.3, R=o.R, conf_lo=3, conf_hi=9, max_age=o.max_age, kalman=1,
                        scale_pool=1, use_pnp=0, hps_uncertainty=1, show_axes=0, cat_rule=0, render_hm_mode=1,
                        render_hmhp_mode=2, pre_hm=1, pre_hm_hp=1, K=K, cap=hip.TRACK_CAP, hungarian=hung,
                        baseline=int(baseline))
    vm = np.zeros((B, 16))
    vm[:, [0, 4]] = 1.0
    vm[:, 6:10] = 512
    dt = hip.DeviceTracker(B, P, vm, device, 512, 512)

    def post_of(dets):
        post = np.zeros((K, hip.POST_STRIDE))
        for i, d in enumerate(dets):
            for k, (off, w) in hip.POST_FIELDS.items():
                if k in d:
                    post[i, off:off + w] = np.asarray(d[k], np.float64).reshape(-1)
            post[i, 64:80] = np.asarray(d["kps_fusion_mean"], np.float64)   # only the fused estimate is given:
            post[i, 8:24] = np.asarray(d["kps_fusion_std"], np.float64)     # heat-map side "missing"
            post[i, 80:96] = -1.0
        return post, len(dets)

    for f in range(len(frames) + B - 1):
        posts, cnts = [], []
        for b in range(B):   # video b lags b frames behind and idles (no detections) outside the sequence
            p, n = post_of(frames[f - b]) if 0 <= f - b < len(frames) else (np.zeros((K, hip.POST_STRIDE)), 0)
            posts.append(p)
            cnts.append(n)
        dt.step(torch.from_numpy(np.stack(posts)).to(device), torch.tensor(cnts, dtype=torch.int32, device=device))
        lists = dt.read()
        for b in range(B):
            if not 0 <= f - b < len(frames):
                continue
            g = gold[f - b]
            assert len(lists[b]) == len(g), (f, b)
            for t, gt in zip(lists[b], g):
                assert (int(t[0]), int(t[1]), int(t[2])) == (gt["tracking_id"], gt["age"], gt["active"]), (f, b)
                np.testing.assert_allclose(t[409:425], gt["kps_mean_kf"], rtol=1e-9, atol=1e-9)
                np.testing.assert_allclose(t[425:441], gt["kps_std_kf"], rtol=1e-9, atol=1e-9)
                np.testing.assert_allclose(t[441:444], gt["obj_scale_kf"], rtol=1e-6)
                np.testing.assert_allclose(t[444:447], gt["obj_scale_uncertainty_kf"], rtol=1e-6, atol=1e-12)
