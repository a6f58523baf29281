"""ORACLE tooling (build container only): import the REFERENCE's own Python modules from
/root/reference so the oracle restatements can be pinned against them and golden vectors can
be generated.  Nothing here is used at GPU-box run time (/root/reference does not exist
there); only the fixtures it writes under tests/golden/ travel.

What is patched, and why:
  * ``_ext``  — the reference's pybind DCNv2 module cannot be built against torch 2.10 (TH/THC
                headers are gone).  A shim module with the identical 14-argument
                ``dcn_v2_forward`` is registered; it runs the reference's own, unmodified CPU
                im2col (compiled into oracle/_ref/libdcn_im2col_ref.so) + torch.addmm, exactly
                as dcn_v2_cpu.cpp:68-105 orchestrates it.
  * ``DLA.load_pretrained_model`` — no-op (it downloads ImageNet weights over HTTP).
"""
import os
import sys
import types

import torch

REF = os.environ.get("CENTERPOSE_REFERENCE", "/root/reference")
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def available():
    return os.path.isdir(os.path.join(REF, "src", "lib"))


def _install_ext_shim():
    from oracle import dcn as odcn

    if "_ext" in sys.modules:
        return
    ext = types.ModuleType("_ext")

    def dcn_v2_forward(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, dg):
        return odcn.dcn_v2_forward(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw,
                                   dh, dw, dg, kind="reference")

    def _unsupported(*a, **k):
        raise RuntimeError("oracle _ext shim: forward only")

    ext.dcn_v2_forward = dcn_v2_forward
    ext.dcn_v2_backward = _unsupported
    ext.dcn_v2_psroi_pooling_forward = _unsupported
    ext.dcn_v2_psroi_pooling_backward = _unsupported
    sys.modules["_ext"] = ext


def setup():
    if not available():
        raise RuntimeError("reference tree not present at " + REF)
    _install_ext_shim()
    for p in (os.path.join(REF, "src"), os.path.join(REF, "src", "lib")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from lib.models.networks import pose_dla_dcn

    pose_dla_dcn.DLA.load_pretrained_model = lambda self, *a, **k: None
    return pose_dla_dcn


class Opt:
    """Minimal stand-in for the argparse namespace the model constructors read
    (pose_dla_dcn.py:253-271, 473)."""

    def __init__(self, tracking=False):
        self.pre_img = tracking
        self.pre_hm = tracking
        self.pre_hm_hp = tracking
        self.tracking_task = tracking


def create_reference_model(arch, heads, tracking=False, head_conv=256):
    """The reference's own ``create_model`` (lib/models/model.py:26-31), eval mode, CPU."""
    setup()
    from lib.models.model import create_model

    model = create_model(arch, dict(heads), head_conv, Opt(tracking))
    model.eval()
    return model


def reference_decode():
    """The reference's ``object_pose_decode`` as imported under modern torch."""
    setup()
    from lib.models.decode import object_pose_decode

    return object_pose_decode


def install_host_shims():
    """Fake third-party modules so the reference's HOST-side detector code (post-process, soft-NMS,
    merge) can be imported for pinning.  cv2.getAffineTransform is the documented 3-point linear solve;
    everything else is an import-time placeholder that is never called by the pinned functions."""
    import json as _json

    import numpy as np

    def fake(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def get_affine(src, dst):
        src = np.asarray(src, np.float64)
        dst = np.asarray(dst, np.float64)
        A = np.zeros((6, 6))
        b = np.zeros(6)
        for i in range(3):
            A[2 * i, 0:3] = [src[i, 0], src[i, 1], 1]
            A[2 * i + 1, 3:6] = [src[i, 0], src[i, 1], 1]
            b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
        return np.linalg.solve(A, b).reshape(2, 3)

    try:
        import cv2  # noqa: F401
    except ImportError:
        fake("cv2", getAffineTransform=get_affine, __version__="4.5.3", INTER_LINEAR=1)
    fake("progress")
    fake("progress.bar", Bar=object)
    fake("simplejson", dump=_json.dump, dumps=_json.dumps, load=_json.load)
    fake("pyrr", Quaternion=object)
    # filterpy>=1.4.5 (requirements.txt:14) is absent: the reference tracker runs on this repo's restatement of its
    # published KalmanFilter (centerpose_amd/lib/utils/kalman.py), so tracker goldens pin the TRACKER logic; the filter
    # arithmetic itself is checked against the textbook equations in tests/test_host_mirror.py
    from centerpose_amd.lib.utils.kalman import KalmanFilter as _KF

    fake("filterpy")
    fake("filterpy.kalman", KalmanFilter=_KF)
    fake("filterpy.common", Q_discrete_white_noise=None)
    fake("numba", jit=lambda *a, **k: (lambda f: f))
    import sklearn.utils  # noqa: F401

    # scikit-learn 0.22.2's sklearn.utils.linear_assignment_ (requirements.txt:13; removed in 0.23, absent here): the reference
    # tracker runs on oracle/munkres.py's operation-for-operation restatement of that module's Munkres state machine, or -- for
    # the "*_scipy" goldens, what a host with a current scipy would compute -- on scipy's rectangular LSAP ($CP_REF_LSAP=scipy)
    def linear_assignment(cost):
        if os.environ.get("CP_REF_LSAP") == "scipy":
            from scipy.optimize import linear_sum_assignment

            r, c = linear_sum_assignment(cost)
            return np.stack([r, c], 1)
        from oracle.munkres import linear_assignment as munkres

        return munkres(cost)

    fake("sklearn.utils.linear_assignment_", linear_assignment=linear_assignment)


def reference_tracker():
    """The reference's ``Tracker`` class (utils/tracker.py) under the host shims."""
    setup()
    install_host_shims()
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib.utils.tracker import Tracker
    return Tracker


def reference_tracker_baseline():
    """The reference's ``Tracker_baseline`` class (utils/tracker_baseline.py: ``--refined_Kalman``) under the host shims."""
    setup()
    install_host_shims()
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib.utils.tracker_baseline import Tracker_baseline
    return Tracker_baseline


def reference_host_modules():
    """(lib.utils.image, lib.utils.post_process, lib.detectors.object_pose) of the reference."""
    setup()
    install_host_shims()
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib.detectors import object_pose as rop
        from lib.utils import image as rimage
        from lib.utils import post_process as rpost
    return rimage, rpost, rop
