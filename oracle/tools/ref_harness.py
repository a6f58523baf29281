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
