"""The device post-process (centerpose_amd/csrc/post_common.h: what postprocess_kernel runs per record and per image --
inverse-affine transform in numpy's float32 / float64 mix, threshold filter, the reference's selection-sort Gaussian
soft-NMS) compiled for the host by tests/native/post_host.cpp and pinned to the REFERENCE's own post_process +
merge_outputs output on the same seeded detections (tests/golden/host_post.json, oracle/tools/make_goldens.py)."""
import ctypes
import json
import os
import subprocess

import numpy as np
import pytest

from centerpose_amd import hip
from centerpose_amd.lib.utils.image import get_affine_transform
from oracle.tools import make_goldens as mg

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def host():
    out = os.path.join(REPO, "tests", "_build", "libcp_post_host.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = os.path.join(REPO, "tests", "native", "post_host.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", out])
    lib = ctypes.CDLL(out)
    lib.cp_post_host_image.restype = ctypes.c_int
    lib.cp_post_host_image.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_int,
                                       ctypes.c_float, ctypes.c_void_p]
    return lib


def _inputs():
    dets, metas = mg.host_cases()
    B, K = dets["scores"].shape[0], dets["scores"].shape[1]
    raw = np.zeros((B, K, hip.DET_STRIDE), np.float32)
    for k, (off, w) in hip.DET_FIELDS.items():
        raw[..., off:off + w] = dets[k].reshape(B, K, w)
    meta = np.zeros((B, 8))
    for b, m in enumerate(metas):
        meta[b, :6] = get_affine_transform(m["c"], m["s"], 0, (m["out_width"], m["out_height"]), inv=1).reshape(-1)
        meta[b, 6] = m["s"] / max(m["out_width"], m["out_height"])
    return raw, meta


def _run(host, raw_b, meta_b, thresh, nms, div_scale=1.0):
    K = raw_b.shape[0]
    out = np.zeros((K, hip.POST_STRIDE))
    raw_b, meta_b = np.ascontiguousarray(raw_b), np.ascontiguousarray(meta_b)
    n = host.cp_post_host_image(raw_b.ctypes.data_as(ctypes.c_void_p), K, meta_b.ctypes.data_as(ctypes.c_void_p),
                                float(thresh), int(nms), float(div_scale), out.ctypes.data_as(ctypes.c_void_p))
    return out[:n]


def test_post_process_and_soft_nms_match_reference_golden(host):
    with open(os.path.join(REPO, "tests", "golden", "host_post.json")) as f:
        g = json.load(f)
    raw, meta = _inputs()
    for b in range(raw.shape[0]):
        rec = _run(host, raw[b], meta[b], mg.HostOpt.vis_thresh, True)
        ref = g["cases"][b]["merged"]
        assert len(rec) == len(ref)
        for r, theirs in zip(rec, ref):
            for k, (off, w) in hip.POST_FIELDS.items():
                np.testing.assert_allclose(r[off:off + w], np.asarray(theirs[k], np.float64).reshape(-1), rtol=1e-9,
                                           atol=1e-9, err_msg=k)


def test_threshold_filter_keeps_decode_order(host):
    raw, meta = _inputs()
    for b in range(raw.shape[0]):
        rec = _run(host, raw[b], meta[b], mg.HostOpt.vis_thresh, False)
        keep = raw[b, :, 4] > mg.HostOpt.vis_thresh
        assert len(rec) == int(keep.sum())
        np.testing.assert_array_equal(rec[:, 0], raw[b, keep, 4].astype(np.float64))


def test_empty_and_all_suppressed(host):
    raw, meta = _inputs()
    assert len(_run(host, raw[0], meta[0], 2.0, True)) == 0          # nothing above the threshold
    same = np.repeat(raw[0, :1], 8, axis=0)                          # eight copies of one box: soft-NMS keeps decaying them
    same[:, 4] = np.linspace(0.9, 0.5, 8)
    rec = _run(host, same, meta[0], 0.3, True)
    assert 1 <= len(rec) < 8 and rec[0, 0] == pytest.approx(0.9)
    assert np.all(np.diff(rec[:, 0]) <= 0) or len(rec) <= 2
