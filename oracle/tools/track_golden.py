"""ORACLE tooling (build container only): CenterPoseTrack's per-frame loop, run by the REFERENCE's own
``ObjectPoseDetector.run`` (detectors/base_detector.py:390-772) on a seeded synthetic video -> tests/golden/track_run.json.

What is real and what is substituted when the reference runs here:
  * real: ``BaseDetector.run`` / ``_get_additional_inputs`` / ``gaussian_fusion``, ``ObjectPoseDetector.process`` /
    ``post_process`` / ``merge_outputs`` (soft-NMS), ``object_pose_decode`` (torch <= 1.1 comparison semantics emulated
    at run time, as for the decode goldens), ``object_pose_post_process``, ``Tracker.step``, ``pnp_shell`` /
    ``CuboidPNPSolver.solve_pnp`` / ``Cuboid3d``, ``opts``;
  * substituted: the network (a stub returning the seeded head tensors of tests/scene.render_video -- the backbone has
    its own goldens), ``cv2.solvePnPGeneric`` / ``projectPoints`` (absent third party: the float64 restatement of
    oracle/pnp.py, PARITY UNPINNED as stated there), ``pyrr.Quaternion``, ``filterpy.KalmanFilter`` (this repo's
    restatement, checked against the textbook equations), the Debugger, ``torch.cuda.synchronize``.
The same ``run_video`` / ``summarise`` drive this repo's detector mirror in tests/ (CPU: oracle decode + oracle PnP
injected; GPU: cp_decode + cp_pnp_solve + cp_render_gaussians), so the golden pins the loop, the previous-frame
rendering, the fusion, the tracker hand-over and the output schema.
"""
import copy
import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLD = os.path.join(REPO, "tests", "golden")
N_FRAMES, N_OBJ, SEED = 5, 2, 21
K_DEMO = np.array([[663.0287679036459, 0, 300.2775065104167], [0, 663.0287679036459, 395.00066121419275], [0, 0, 1]])
TRACK_ARGV = ["--tracking_task", "--arch", "dla_34", "--c", "cup", "--gpus", "-1", "--debug", "0"]
BASELINE_ARGV = TRACK_ARGV + ["--refined_Kalman"]


def demo_flags(opt):
    """What src/demo.py:111-149 sets before opts().parse / init."""
    opt.nms = True
    opt.obj_scale = True
    if opt.tracking_task:
        opt.pre_img = True
        opt.pre_hm = True
        opt.tracking = True
        opt.pre_hm_hp = True
        opt.tracking_hp = True
        opt.track_thresh = 0.1
        opt.obj_scale_uncertainty = True
        opt.hps_uncertainty = True
        opt.kalman = True
        opt.scale_pool = True
        opt.vis_thresh = max(opt.track_thresh, opt.vis_thresh)
        opt.pre_thresh = max(opt.track_thresh, opt.pre_thresh)
        opt.new_thresh = max(opt.track_thresh, opt.new_thresh)
    opt.cam_intrinsic = K_DEMO
    opt.use_pnp = True
    return opt


def video_heads():
    """Per-frame head tensors (logits for hm / hm_hp, as the network emits them)."""
    from tests import scene

    frames = scene.render_video(N_FRAMES, N_OBJ, SEED)
    out = []
    for h in frames:
        t = {k: torch.from_numpy(v.copy()) for k, v in h.items()}
        for k in ("hm", "hm_hp"):
            p = t[k].clamp(1e-6, 1 - 1e-6)
            t[k] = torch.log(p / (1 - p))
        out.append(t)
    return out


def frame_inputs(get_affine_transform):
    """(preprocessed image [3,512,512], meta) per frame: 512x512 frames, identity crop."""
    g = torch.Generator().manual_seed(SEED)
    c = np.array([256.0, 256.0], dtype=np.float32)
    s = 512.0
    res = []
    for f in range(N_FRAMES):
        img = torch.randn(3, 512, 512, generator=g).numpy()
        meta = {"c": c, "s": s, "height": 512, "width": 512, "out_height": 128, "out_width": 128, "inp_height": 512,
                "inp_width": 512, "trans_input": get_affine_transform(c, s, 0, [512, 512]),
                "trans_output": get_affine_transform(c, s, 0, [128, 128]), "camera_matrix": K_DEMO, "id": f}
        res.append((img, meta))
    return res


def run_video(detector, get_affine_transform):
    """Feed the frames through ``detector.run`` (pre-processed path) and summarise every frame."""
    out = []
    for img, meta in frame_inputs(get_affine_transform):
        ret = detector.run(img, meta_inp=copy.deepcopy(meta), preprocessed_flag=True)
        out.append(summarise(ret, detector))
    return out


def _f(v):
    return np.asarray(v, np.float64).reshape(-1).tolist()


def summarise(ret, detector):
    tracks = []
    for t in ret["results"]:
        d = {"tracking_id": int(t["tracking_id"]), "age": int(t["age"]), "active": int(t["active"]),
             "score": float(t["score"]), "bbox": _f(t["bbox"]), "ct": _f(t["ct"]), "kps": _f(t["kps"]),
             "kps_fusion_mean": _f(t["kps_fusion_mean"]), "kps_fusion_std": _f(t["kps_fusion_std"]),
             "kps_mean_kf": _f(t["kps_mean_kf"]), "kps_std_kf": _f(t["kps_std_kf"]),
             "obj_scale_kf": _f(t["obj_scale_kf"]), "obj_scale_uncertainty_kf": _f(t["obj_scale_uncertainty_kf"]),
             "tracking": _f(t["tracking"])}
        for k in ("kps_pnp", "kps_pnp_kf", "kps_3d_cam_kf", "location", "quaternion_xyzw"):
            if k in t:
                d[k] = _f(t[k])
        tracks.append(d)
    model = detector.model
    return {"tracks": tracks, "n_boxes": len(ret["boxes"]),
            "boxes_kps_pnp": [_f(b[0]) for b in ret["boxes"]],
            "keys": sorted(k for k in ret if k != "output"),
            # what the network was fed as previous-frame heat-maps (rendered from the tracks)
            "pre_hm_sum": float(model.last_pre_hm.double().sum()) if model.last_pre_hm is not None else None,
            "pre_hm_max": float(model.last_pre_hm.max()) if model.last_pre_hm is not None else None,
            "pre_hm_hp_sum": _f(model.last_pre_hm_hp.double().sum(dim=(0, 2, 3))) if model.last_pre_hm_hp is not None else None,
            "pre_hm_hp_nonzero": int((model.last_pre_hm_hp > 0).sum()) if model.last_pre_hm_hp is not None else None}


class StubNetwork(object):
    """``model(images, pre_images, pre_hms, pre_hm_hp) -> [heads]`` returning the seeded heads frame by frame."""

    def __init__(self, heads_per_frame):
        self.frames = heads_per_frame
        self.calls = 0
        self.last_pre_hm = None
        self.last_pre_hm_hp = None
        self.pre_image_was_current = []

    def to(self, device):
        return self

    def eval(self):
        return self

    def __call__(self, images, pre_images=None, pre_hms=None, pre_hm_hp=None):
        self.last_pre_hm = None if pre_hms is None else pre_hms.detach().cpu().clone()
        self.last_pre_hm_hp = None if pre_hm_hp is None else pre_hm_hp.detach().cpu().clone()
        h = self.frames[self.calls]
        self.calls += 1
        return [{k: v.clone() for k, v in h.items()}]


def _install_reference_shims():
    """cv2 PnP entry points, pyrr and the Debugger, on top of ref_harness.install_host_shims()."""
    from oracle import pnp as opnp
    from oracle.tools import ref_harness as rh

    rh.setup()
    rh.install_host_shims()
    cv2 = sys.modules["cv2"]

    def solvePnPGeneric(obj, img, K, dist, flags=0):
        obj = np.asarray(obj, np.float64).reshape(-1, 3)
        img = np.asarray(img, np.float64).reshape(-1, 2)  # the tracker passes (8, 2, 1) columns of the filter state
        ok, rvec, tvec = opnp.solve_pnp_any(obj, img, np.asarray(K, np.float64), epnp=(flags == cv2.SOLVEPNP_EPNP))
        proj = opnp.project_points(obj, rvec, tvec, np.asarray(K, np.float64))
        rms = np.sqrt(np.mean(np.sum((proj - img) ** 2, axis=1)))
        return ok, [rvec.reshape(3, 1)], [tvec.reshape(3, 1)], np.array([[rms]])

    def projectPoints(pts, rvec, tvec, K, dist):
        uv = opnp.project_points(np.asarray(pts, np.float64), np.asarray(rvec, np.float64).reshape(3),
                                 np.asarray(tvec, np.float64).reshape(3), np.asarray(K, np.float64))
        return uv.reshape(-1, 1, 2), None

    cv2.SOLVEPNP_ITERATIVE, cv2.SOLVEPNP_EPNP = 0, 1
    cv2.solvePnPGeneric = solvePnPGeneric
    cv2.projectPoints = projectPoints

    class Quaternion(object):
        @staticmethod
        def from_axis_rotation(axis, theta):
            axis = np.asarray(axis, np.float64)
            axis = axis / np.linalg.norm(axis)
            h = theta * 0.5
            return np.array([np.sin(h) * axis[0], np.sin(h) * axis[1], np.sin(h) * axis[2], np.cos(h)])

    sys.modules["pyrr"].Quaternion = Quaternion
    dbg = types.ModuleType("lib.utils.debugger")

    class Debugger(object):
        def __init__(self, *a, **k):
            pass

    dbg.Debugger = Debugger
    sys.modules["lib.utils.debugger"] = dbg
    torch.cuda.synchronize = lambda *a, **k: None


def reference_detector(argv=TRACK_ARGV):
    """The reference's ObjectPoseDetector built with demo.py's flag set, a stub network and uint8 decode semantics."""
    import contextlib
    import io
    import warnings

    from oracle.tools.make_goldens import U8Cmp

    _install_reference_shims()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib.detectors import base_detector as rbd
        from lib.detectors import object_pose as rop
        from lib.opts import opts as ref_opts
        from lib.utils import image as rimage
    with contextlib.redirect_stdout(io.StringIO()):
        opt = ref_opts().parser.parse_args(argv)
        opt = demo_flags(opt)
        opt = ref_opts().parse(opt)
        opt = ref_opts().init(opt)
    stub = StubNetwork(video_heads())
    rbd.create_model = lambda *a, **k: stub
    rbd.load_model = lambda m, *a, **k: m
    real_decode = rop.object_pose_decode
    if not getattr(real_decode, "_u8", False):
        def decode_u8(*a, **k):
            a = [x.as_subclass(U8Cmp) if isinstance(x, torch.Tensor) else x for x in a]
            k = {n: (x.as_subclass(U8Cmp) if isinstance(x, torch.Tensor) else x) for n, x in k.items()}
            out = real_decode(*a, **k)
            return {n: (v.as_subclass(torch.Tensor) if isinstance(v, torch.Tensor) else v) for n, v in out.items()}

        decode_u8._u8 = True
        rop.object_pose_decode = decode_u8
    with contextlib.redirect_stdout(io.StringIO()):
        det = rop.ObjectPoseDetector(opt)
    return det, rimage.get_affine_transform


def shell_cases():
    """Injected poses for the packaging / visibility logic of pnp_shell (cuboid_pnp_shell.py:26-91): a centred object,
    objects sliding out of the frame to the right / bottom until 3 resp. 6 projected points leave the unit square, a
    centroid outside, per category family (thresholds 3, 6, none)."""
    from oracle import pnp as opnp

    cases = []
    scale = np.array([0.8, 1.0, 1.3])
    V = opnp.cuboid_vertices(scale / scale[1])
    q = np.array([0.2, -0.4, 0.1, 0.9])
    R = opnp.quat_xyzw_to_matrix(q / np.linalg.norm(q))
    rvec = opnp.matrix_to_rodrigues(R)
    for cat in ("cup", "chair", "shoe"):
        for tx, ty in ((0.0, 0.0), (0.9, 0.0), (1.15, 0.0), (1.35, 0.0), (1.8, 0.2), (0.0, 0.75), (0.0, 1.2), (-1.4, -1.6)):
            t = np.array([tx, ty, 3.0])
            proj = opnp.project_points(V, rvec, t, K_DEMO)
            cases.append({"c": cat, "location": t.tolist(), "quaternion": opnp.axis_angle_quat_xyzw(rvec).tolist(),
                          "projected": proj.tolist(), "scale": scale.tolist(),
                          "kps": (proj + 1.5).reshape(-1).tolist()})
    return cases


def run_shell_cases(pnp_shell_fn, solver_cls):
    """pnp_shell with the solver's answer injected (so only the reference's own packaging logic runs)."""
    out = []
    for cs in shell_cases():
        opt = types.SimpleNamespace(c=cs["c"])
        meta = {"camera_matrix": K_DEMO, "width": 600, "height": 800}
        bbox = {"kps": list(cs["kps"]), "obj_scale": np.array(cs["scale"])}
        orig = solver_cls.solve_pnp
        solver_cls.solve_pnp = lambda self, pts, OPENCV_RETURN=False, **k: (
            list(cs["location"]), np.array(cs["quaternion"]), np.array(cs["projected"]), 0.5)
        try:
            ret = pnp_shell_fn(opt, meta, bbox, np.zeros((16, 2)), cs["scale"], OPENCV_RETURN=True)
        finally:
            solver_cls.solve_pnp = orig
        if ret is None:
            out.append(None)
        else:
            out.append({"kps_pnp": _f(ret[0]), "kps_3d_cam": _f(ret[1]), "obj_scale": _f(ret[2]), "kps_ori": _f(ret[3]),
                        "bbox_keys": sorted(ret[4].keys())})
    return out


def shell_golden():
    _install_reference_shims()
    from lib.utils.pnp import cuboid_pnp_shell as rshell
    from lib.utils.pnp.cuboid_pnp_solver import CuboidPNPSolver as RSolver

    out = run_shell_cases(rshell.pnp_shell, RSolver)
    with open(os.path.join(GOLD, "pnp_shell_ref.json"), "w") as f:
        json.dump(out, f)
    print("pnp_shell_ref.json: kept / dropped", sum(o is not None for o in out), sum(o is None for o in out))


def main():
    import contextlib
    import io

    shell_golden()

    det, gat = reference_detector()
    with contextlib.redirect_stdout(io.StringIO()):
        frames = run_video(det, gat)
    with open(os.path.join(GOLD, "track_run.json"), "w") as f:
        json.dump({"argv": TRACK_ARGV, "frames": frames}, f)
    print("track_run.json: tracks per frame", [len(x["tracks"]) for x in frames], "boxes", [x["n_boxes"] for x in frames],
          "pre_hm_hp pixels", [x["pre_hm_hp_nonzero"] for x in frames])
    # the same video with --refined_Kalman on top: base_detector.py:53-57 then installs Tracker_baseline (position-only
    # filter, plain scale average) inside the same two-frame loop
    det, gat = reference_detector(BASELINE_ARGV)
    with contextlib.redirect_stdout(io.StringIO()):
        frames = run_video(det, gat)
    with open(os.path.join(GOLD, "track_run_baseline.json"), "w") as f:
        json.dump({"argv": BASELINE_ARGV, "frames": frames}, f)
    print("track_run_baseline.json: tracks per frame", [len(x["tracks"]) for x in frames])


if __name__ == "__main__":
    main()
