"""Run in a fresh interpreter by tests/test_demo_dropin.py: the REFERENCE's src/demo.py, unmodified, with this repo's
package in place of the reference's ``lib`` (INTEGRATION.md level 0: put centerpose_amd/ ahead of src/ on sys.path).

usage: python demo_dropin_script.py <reference src dir> <image> <checkpoint>
Only what cannot exist on this GPU-less build box is substituted: cv2 (absent: imread through PIL, no windows) and the
two device stages of ``process`` / PnP (the oracle's CPU forward, decode and PnP stand in for libcenterpose_hip.so)."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ref_src, image, ckpt = sys.argv[1], sys.argv[2], sys.argv[3]
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "centerpose_amd"))   # `lib` now resolves to centerpose_amd/lib

try:
    import cv2  # noqa: F401
except ImportError:
    from PIL import Image

    cv2 = types.ModuleType("cv2")
    cv2.imread = lambda p: np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1].copy()
    cv2.imshow = lambda *a, **k: None
    cv2.waitKey = lambda *a, **k: -1
    cv2.VideoCapture = None
    sys.modules["cv2"] = cv2

spec = importlib.util.spec_from_file_location("reference_demo", os.path.join(ref_src, "demo.py"))
demo = importlib.util.module_from_spec(spec)
spec.loader.exec_module(demo)                    # `from lib.opts import opts`, `from lib.detectors...` -> this repo
import lib  # noqa: E402

assert os.path.realpath(os.path.dirname(lib.__file__)).startswith(os.path.realpath(REPO)), lib.__file__
from lib.opts import opts  # noqa: E402

# ---- what demo.py's __main__ block does (demo.py:88-155) ----
opt = opts().parser.parse_args(["--demo", image, "--arch", "dla_34", "--load_model", ckpt, "--gpus", "-1", "--c", "chair"])
opt.nms = True
opt.obj_scale = True
meta = {"camera_matrix": np.array([[663.0287679036459, 0, 300.2775065104167], [0, 663.0287679036459, 395.00066121419275],
                                   [0, 0, 1]])}
opt.cam_intrinsic = meta["camera_matrix"]
opt.use_pnp = True
opt = opts().parse(opt)
opt = opts().init(opt)

# ---- CPU stand-ins for the device stages (this box has no GPU) ----
from lib.detectors import base_detector as bd  # noqa: E402
from lib.detectors.object_pose import ObjectPoseDetector  # noqa: E402
from lib.utils.pnp import cuboid_pnp_solver as cps  # noqa: E402
from oracle import backbone as ob  # noqa: E402
from oracle import decode as odec  # noqa: E402
from tests.test_tracking_loop import _oracle_pnp_rows  # noqa: E402

sd = torch.load(ckpt, map_location="cpu")["state_dict"]


def process(self, images, pre_images=None, pre_hms=None, pre_hm_hp=None, pre_inds=None, return_time=False):
    import time

    z = ob.dlaseg_forward(sd, images.float(), self.opt.heads, arch="dla")
    z = {k: v for k, v in z.items()}
    z["hm"], z["hm_hp"] = torch.sigmoid(z["hm"]), torch.sigmoid(z["hm_hp"])
    n = {k: v.numpy() for k, v in z.items()}
    dets = odec.object_pose_decode(n["hm"], n["hps"], wh=n["wh"], obj_scale=n["scale"], reg=n["reg"], hm_hp=n["hm_hp"],
                                   hp_offset=n["hp_offset"], K=self.opt.K, rep_mode=self.opt.rep_mode)
    return (z, dets, time.time()) if return_time else (z, dets)


ObjectPoseDetector.process = process
cps.solve_pnp_batch = _oracle_pnp_rows
bd.solve_pnp_batch = _oracle_pnp_rows
demo.demo(opt, meta)
print("DEMO_DROPIN_OK")
