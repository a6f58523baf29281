"""ctypes binding of libcenterpose_hip.so (C ABI: include/centerpose_hip.h).

PyTorch-ROCm is only the tensor container / allocator / stream provider here: every call hands raw
``data_ptr()`` device pointers and the current HIP stream to the library.  There is NO fallback:
if the shared library is missing or a call fails, a ``RuntimeError`` is raised.
"""
import ctypes
import os
from collections import OrderedDict

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# $CENTERPOSE_HIP_LIB selects another build of the same ABI (tuning variants: make -C csrc EXP=n)
LIB_PATH = os.environ.get("CENTERPOSE_HIP_LIB") or os.path.join(_HERE, "libcenterpose_hip.so")

_lib = None
ABI_VERSION = 6  # CP_ABI_VERSION of include/centerpose_hip.h this binding was written against

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_size_t = ctypes.c_size_t
c_char_p = ctypes.c_char_p


class TrackParams(ctypes.Structure):
    """cp_track_params of include/centerpose_hip.h (field for field)."""
    _fields_ = [(n, ctypes.c_double) for n in ("new_thresh", "pre_thresh", "R", "conf_lo", "conf_hi")] + \
               [(n, ctypes.c_int) for n in ("max_age", "kalman", "scale_pool", "use_pnp", "hps_uncertainty", "show_axes",
                                            "cat_rule", "render_hm_mode", "render_hmhp_mode", "pre_hm", "pre_hm_hp", "K",
                                            "cap", "hungarian", "baseline")]


def _sig(fn, restype, *argtypes):
    fn.restype = restype
    fn.argtypes = list(argtypes)


def lib():
    """Load (once) and return the C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "centerpose_amd: %s not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C centerpose_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    # the ABI guard comes before any other symbol is bound: a stale library is exactly the case it exists for, and it must
    # fail with "rebuild", not with ctypes' "undefined symbol"
    if not hasattr(L, "cp_abi_version"):
        raise RuntimeError("centerpose_amd: %s predates the ABI guard (no cp_abi_version): rebuild the library" % LIB_PATH)
    _sig(L.cp_abi_version, c_int)
    if L.cp_abi_version() != ABI_VERSION:
        raise RuntimeError("centerpose_amd: %s has ABI version %d, this binding was written for %d (rebuild the library)"
                           % (LIB_PATH, L.cp_abi_version(), ABI_VERSION))
    _sig(L.cp_version, c_char_p)
    _sig(L.cp_last_error, c_char_p)
    _sig(L.cp_num_kernel_variants, c_int)
    _sig(L.cp_num_roles, c_int)
    _sig(L.cp_dcnv2_workspace_bytes, c_size_t, c_int, c_int, c_int, c_int, c_int)
    _sig(L.cp_dcnv2_forward, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
         *([c_int] * 14), c_void_p, c_size_t)
    _sig(L.cp_model_create, c_int, c_char_p, c_int, c_int, ctypes.POINTER(c_char_p), ctypes.POINTER(c_int), c_int,
         ctypes.POINTER(c_void_p))
    _sig(L.cp_model_set_param, c_int, c_void_p, c_char_p, c_void_p, ctypes.c_int64)
    _sig(L.cp_model_finalize, c_int, c_void_p)
    _sig(L.cp_model_destroy, None, c_void_p)
    _sig(L.cp_model_workspace_bytes, c_size_t, c_void_p, c_int, c_int, c_int)
    _sig(L.cp_model_forward, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
         ctypes.POINTER(c_void_p), c_int, c_void_p, c_size_t)
    _sig(L.cp_model_forward_tap, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
         c_void_p, ctypes.POINTER(c_void_p), c_int, c_void_p, c_size_t, c_char_p, c_void_p, ctypes.POINTER(c_int))
    _sig(L.cp_conv2d_workspace_bytes, c_size_t, c_int, c_int, c_int, c_int)
    _sig(L.cp_conv2d_nhwc, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
         *([c_int] * 10), c_void_p, c_size_t)
    _sig(L.cp_decode_workspace_bytes, c_size_t, c_int, c_int)
    _sig(L.cp_decode, c_int, c_void_p, c_int, c_int, c_int, *([c_void_p] * 11), c_int, c_int, c_int, ctypes.c_float,
         c_int, c_int, c_void_p, c_void_p, c_size_t)
    _sig(L.cp_model_detect_workspace_bytes, c_size_t, c_void_p, c_int, c_int, c_int, c_int)
    _sig(L.cp_model_detect, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
         ctypes.POINTER(c_void_p), c_int, c_int, c_int, ctypes.c_float, c_int, c_void_p, c_void_p, c_size_t, c_int)
    _sig(L.cp_set_default_precision, c_int, c_int)
    _sig(L.cp_set_debug, c_int, c_int)
    _sig(L.cp_render_gaussians, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int)
    _sig(L.cp_postprocess_workspace_bytes, c_size_t, c_int, c_int)
    _sig(L.cp_postprocess, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, ctypes.c_double, c_int, ctypes.c_float,
         c_void_p, c_void_p, c_void_p, c_size_t)
    _sig(L.cp_preprocess, c_int, c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(ctypes.c_double),
         ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), c_void_p, c_int, c_int)
    _sig(L.cp_preprocess_batch, c_int, c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_double),
         ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), c_void_p, c_int, c_int)
    _sig(L.cp_resize_u8, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int)
    _sig(L.cp_model_set_precision, c_int, c_void_p, c_int)
    _sig(L.cp_model_profile, c_int, c_void_p, c_int)
    _sig(L.cp_model_profile_read, c_int, c_void_p, ctypes.POINTER(ctypes.c_double), c_int)
    _sig(L.cp_kernel_variant_name, c_char_p, c_int)
    _sig(L.cp_model_profile_roles, c_int, c_void_p, ctypes.POINTER(ctypes.c_double), c_int)
    _sig(L.cp_role_name, c_char_p, c_int)
    _sig(L.cp_pnp_workspace_bytes, c_size_t, c_int)
    _sig(L.cp_pnp_from_post_workspace_bytes, c_size_t, c_int, c_int)
    _sig(L.cp_pnp_from_post, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t)
    _sig(L.cp_pnp_solve, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t)
    _sig(L.cp_track_state_bytes, c_size_t, c_int, c_int)
    _sig(L.cp_track_workspace_bytes, c_size_t, c_int, c_int, c_int)
    _sig(L.cp_track_reset, c_int, c_void_p, c_void_p, c_int, c_int)
    _sig(L.cp_track_status, c_int, c_void_p, c_void_p, c_int, ctypes.POINTER(c_int))
    _sig(L.cp_track_step, c_int, c_void_p, ctypes.POINTER(TrackParams), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
         c_void_p, c_void_p, c_size_t)
    _sig(L.cp_linear_assignment, c_int, c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_int))
    _lib = L
    return L


def linear_assignment(cost, solver=1):
    """The tracker's optimal assignment on the host (cp_linear_assignment; no device involved): ``cost`` [n, m] float64 ->
    int64 array [min(n, m), 2] of (row, column) pairs sorted by row, the return value of scikit-learn 0.22's
    ``linear_assignment`` (tracker.py:157).  solver 1 = that module's Munkres, 2 = scipy's rectangular LSAP."""
    import numpy as np

    cost = np.ascontiguousarray(cost, dtype=np.float64)
    if cost.ndim != 2:
        raise ValueError("linear_assignment: a 2-D cost matrix is expected")
    n, m = cost.shape
    match = (c_int * max(n, 1))()
    _check(lib().cp_linear_assignment(cost.ctypes.data_as(c_void_p), n, m, int(solver), match), "cp_linear_assignment")
    return np.array([(i, match[i]) for i in range(n) if match[i] >= 0], dtype=np.int64).reshape(-1, 2)


def exported_symbols():
    """Names every declaration in include/centerpose_hip.h (+ the test hook of centerpose_hip_testing.h) must resolve to (used by CPU tests)."""
    return ["cp_version", "cp_last_error", "cp_dcnv2_workspace_bytes", "cp_dcnv2_forward", "cp_model_create",
            "cp_model_set_param", "cp_model_finalize", "cp_model_destroy", "cp_model_workspace_bytes",
            "cp_model_forward", "cp_model_forward_tap", "cp_conv2d_workspace_bytes", "cp_conv2d_nhwc",
            "cp_decode_workspace_bytes", "cp_decode", "cp_pnp_workspace_bytes", "cp_pnp_solve", "cp_model_profile", "cp_model_profile_read",
            "cp_kernel_variant_name", "cp_set_default_precision", "cp_model_set_precision", "cp_model_detect_workspace_bytes", "cp_model_detect", "cp_set_debug", "cp_preprocess", "cp_preprocess_batch", "cp_postprocess_workspace_bytes", "cp_postprocess", "cp_render_gaussians",
            "cp_model_profile_roles", "cp_role_name", "cp_pnp_from_post_workspace_bytes", "cp_pnp_from_post", "cp_resize_u8",
            "cp_abi_version", "cp_num_kernel_variants", "cp_num_roles", "cp_track_state_bytes", "cp_track_workspace_bytes",
            "cp_track_reset", "cp_track_step", "cp_track_status", "cp_linear_assignment"]


def _check(rc, what):
    if rc != 0:
        msg = lib().cp_last_error().decode() if _lib is not None else ""
        raise RuntimeError("centerpose_hip: %s failed with code %d (%s)" % (what, rc, msg))


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _dev(t):
    if not t.is_cuda:
        raise RuntimeError("centerpose_hip: tensors must live on the HIP device (no CPU path)")
    return t.contiguous().float()


def dcn_v2_forward(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, deformable_group):
    """Same 14-argument signature as the reference's ``_ext.dcn_v2_forward`` (DCNv2/src/vision.cpp:5)."""
    L = lib()
    input, weight, bias, offset, mask = map(_dev, (input, weight, bias, offset, mask))
    B, C, H, W = input.shape
    Co = weight.shape[0]
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1  # dcn_v2_cuda.cu:75-76
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    if Ho < 1 or Wo < 1:
        raise RuntimeError("dcn_v2_forward: empty output")
    for name, t, shape in (("weight", weight, (Co, C, kh, kw)), ("bias", bias, (Co,)),
                           ("offset", offset, (B, deformable_group * 2 * kh * kw, Ho, Wo)),
                           ("mask", mask, (B, deformable_group * kh * kw, Ho, Wo))):
        if tuple(t.shape) != shape:  # the reference's AT_ASSERTM shape checks (dcn_v2_cuda.cu:60-66)
            raise RuntimeError("dcn_v2_forward: %s has shape %s, expected %s" % (name, tuple(t.shape), shape))
    out = torch.empty(B, Co, Ho, Wo, device=input.device, dtype=torch.float32)
    nbytes = L.cp_dcnv2_workspace_bytes(B, C, H, W, Co)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=input.device)
    rc = L.cp_dcnv2_forward(_stream(), _ptr(input), _ptr(weight), _ptr(bias), _ptr(offset), _ptr(mask), _ptr(out),
                            B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, deformable_group, _ptr(ws), nbytes)
    _check(rc, "cp_dcnv2_forward")
    return out


def conv2d_nhwc(x, w, scale=None, shift=None, residual=None, stride=1, pad=0, act=0):
    """x [B,H,W,Cin] NHWC, w [Cout,Cin,KH,KW] -> [B,Ho,Wo,Cout] NHWC (unit-test entry of the igemm kernel)."""
    L = lib()
    x, w = _dev(x), _dev(w)
    B, H, W, Cin = x.shape
    Cout, _, KH, KW = w.shape
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    out = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=torch.float32)
    nbytes = L.cp_conv2d_workspace_bytes(Cin, Cout, KH, KW)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    scale = _dev(scale) if scale is not None else None
    shift = _dev(shift) if shift is not None else None
    residual = _dev(residual) if residual is not None else None
    rc = L.cp_conv2d_nhwc(_stream(), _ptr(x), _ptr(w), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(out),
                          B, H, W, Cin, Cout, KH, KW, stride, pad, act, _ptr(ws), nbytes)
    _check(rc, "cp_conv2d_nhwc")
    return out


PRECISIONS = {"f32": 0, "f16x3": 1}


def set_default_precision(name):
    """'f32' (exact float32 MFMA) or 'f16x3' (split-binary16 MFMA, float32-class accuracy)."""
    _check(lib().cp_set_default_precision(PRECISIONS[name]), "cp_set_default_precision")


DET_FIELDS = OrderedDict([  # field -> (offset, width) inside a 118-float detection record (decode.py:347-361)
    ("bboxes", (0, 4)), ("scores", (4, 1)), ("kps", (5, 16)), ("clses", (21, 1)), ("obj_scale", (22, 3)),
    ("obj_scale_uncertainty", (25, 3)), ("tracking", (28, 2)), ("tracking_hp", (30, 16)),
    ("kps_displacement_mean", (46, 16)), ("kps_displacement_std", (62, 16)), ("kps_heatmap_mean", (78, 16)),
    ("kps_heatmap_std", (94, 16)), ("kps_heatmap_height", (110, 8))])
DET_STRIDE = 118


def decode_raw(hm, hps, wh, hm_hp, hps_uncertainty=None, scale=None, scale_uncertainty=None, reg=None,
               hp_offset=None, tracking=None, tracking_hp=None, K=100, rep_mode=1, fit_gaussian=False,
               balance=2.0, legacy_bool_mask=False, apply_sigmoid=False):
    """Device decode -> det [B,K,118] (device tensor).  hm / hm_hp are modified in place when
    apply_sigmoid is set.  Tensors must be contiguous float32 NCHW on the HIP device."""
    L = lib()
    for t in (hm, hps, wh, hm_hp):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise RuntimeError("decode: required heads must be contiguous float32 device tensors")
    opt = [None if t is None else _dev(t) for t in (hps_uncertainty, scale, scale_uncertainty, reg, hp_offset,
                                                    tracking, tracking_hp)]
    B, _, H, W = hm.shape
    det = torch.empty(B, K, DET_STRIDE, device=hm.device, dtype=torch.float32)
    n = L.cp_decode_workspace_bytes(B, K)
    ws = torch.empty(n, dtype=torch.uint8, device=hm.device)
    rc = L.cp_decode(_stream(), B, H, W, _ptr(hm), _ptr(hps), _ptr(wh), _ptr(opt[0]), _ptr(opt[1]), _ptr(opt[2]),
                     _ptr(opt[3]), _ptr(hm_hp), _ptr(opt[4]), _ptr(opt[5]), _ptr(opt[6]), int(K), int(rep_mode),
                     int(bool(fit_gaussian)), float(balance), int(bool(legacy_bool_mask)), int(bool(apply_sigmoid)),
                     _ptr(det), _ptr(ws), n)
    _check(rc, "cp_decode")
    return det


def split_detections(det):
    """[B,K,118] -> dict of the 13 reference keys (views)."""
    return OrderedDict((k, det[..., o:o + w]) for k, (o, w) in DET_FIELDS.items())


def preprocess(image_u8_hwc, trans_input, mean, std, out_h, out_w):
    """Device warp + normalise of one BGR uint8 frame [H,W,3] -> float32 [1,3,out_h,out_w]
    (BaseDetector.pre_process, base_detector.py:127-134).  ``trans_input`` is the 2x3 source->input affine."""
    import numpy as np

    L = lib()
    if not (image_u8_hwc.is_cuda and image_u8_hwc.dtype == torch.uint8 and image_u8_hwc.is_contiguous()):
        raise RuntimeError("preprocess: image must be a contiguous uint8 device tensor [H,W,3]")
    H, W = int(image_u8_hwc.shape[0]), int(image_u8_hwc.shape[1])
    fwd = np.asarray(trans_input, np.float64).reshape(-1)
    f3 = ctypes.c_float * 3
    out = torch.empty(1, 3, out_h, out_w, device=image_u8_hwc.device, dtype=torch.float32)
    rc = L.cp_preprocess(_stream(), _ptr(image_u8_hwc), H, W, (ctypes.c_double * 6)(*fwd.tolist()),
                         f3(*[float(v) for v in np.asarray(mean).reshape(-1)]),
                         f3(*[float(v) for v in np.asarray(std).reshape(-1)]), _ptr(out), out_h, out_w)
    _check(rc, "cp_preprocess")
    return out


def preprocess_batch(images_u8_bhwc, trans_input, mean, std, out_h, out_w, out=None):
    """`preprocess` for B frames of one size sharing the transform: uint8 [B,H,W,3] -> float32 [B,3,out_h,out_w], one launch."""
    import numpy as np

    L = lib()
    t = images_u8_bhwc
    if not (t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and t.dim() == 4 and t.shape[3] == 3):
        raise RuntimeError("preprocess_batch: images must be a contiguous uint8 device tensor [B,H,W,3]")
    B, H, W = int(t.shape[0]), int(t.shape[1]), int(t.shape[2])
    fwd = np.asarray(trans_input, np.float64).reshape(-1)
    f3 = ctypes.c_float * 3
    if out is None:
        out = torch.empty(B, 3, out_h, out_w, device=t.device, dtype=torch.float32)
    rc = L.cp_preprocess_batch(_stream(), _ptr(t), B, H, W, (ctypes.c_double * 6)(*fwd.tolist()),
                               f3(*[float(v) for v in np.asarray(mean).reshape(-1)]),
                               f3(*[float(v) for v in np.asarray(std).reshape(-1)]), _ptr(out), out_h, out_w)
    _check(rc, "cp_preprocess_batch")
    return out


def resize_u8(image_u8_hwc, out_h, out_w):
    """cv2.resize(img, (out_w, out_h)) (INTER_LINEAR, OpenCV's fixed-point form) of a uint8 [H,W,C] device frame."""
    L = lib()
    if not (image_u8_hwc.is_cuda and image_u8_hwc.dtype == torch.uint8 and image_u8_hwc.is_contiguous()
            and image_u8_hwc.dim() == 3):
        raise RuntimeError("resize_u8: image must be a contiguous uint8 device tensor [H,W,C]")
    H, W, C = (int(v) for v in image_u8_hwc.shape)
    out = torch.empty(out_h, out_w, C, device=image_u8_hwc.device, dtype=torch.uint8)
    _check(L.cp_resize_u8(_stream(), _ptr(image_u8_hwc), H, W, C, _ptr(out), out_h, out_w), "cp_resize_u8")
    return out


def render_gaussians(records, C, H, W, device, out=None):
    """Draw (channel, x, y, radius, k) Gaussians into a float32 [C,H,W] device map with max() merging
    (draw_umich_gaussian, utils/image.py:135-150): the pre_hm / pre_hm_hp render of CenterPoseTrack."""
    L = lib()
    rec = torch.as_tensor(records, dtype=torch.float64).reshape(-1, 5).contiguous().to(device)
    if out is None:
        out = torch.empty(C, H, W, dtype=torch.float32, device=device)
    elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (C, H, W)):
        raise RuntimeError("render_gaussians: out must be a contiguous float32 device tensor [C,H,W]")
    _check(L.cp_render_gaussians(_stream(), _ptr(rec) if rec.numel() else None, int(rec.shape[0]), _ptr(out), C, H, W,
                                 1), "cp_render_gaussians")
    return out


POST_STRIDE = 120
POST_FIELDS = OrderedDict([  # field -> (offset, width) inside a post-processed record (post_process.py:21-58)
    ("score", (0, 1)), ("cls", (1, 1)), ("obj_scale", (2, 3)), ("obj_scale_uncertainty", (5, 3)),
    ("kps_displacement_std", (8, 16)), ("bbox", (24, 4)), ("ct", (28, 2)), ("kps", (30, 16)), ("tracking", (46, 2)),
    ("tracking_hp", (48, 16)), ("kps_displacement_mean", (64, 16)), ("kps_heatmap_mean", (80, 16)),
    ("kps_heatmap_std", (96, 16)), ("kps_heatmap_height", (112, 8))])


def postprocess(det, meta, vis_thresh, nms=True, div_scale=1.0, out=None, cnt=None, ws=None):
    """Device post-process + Gaussian soft-NMS of a whole batch (object_pose.py:167-197).
    det [B,K,118] float32 device; meta [B,8] float64 (inverse affine 6, ratio, pad) -> (records [B,K,120] float64
    device, counts [B] int32 device); image b keeps records[b, :counts[b]] in the reference's final order.
    ``out`` / ``cnt`` / ``ws``: caller-owned result and workspace tensors (PoseStage rotates its own)."""
    L = lib()
    if not (det.is_cuda and det.dtype == torch.float32 and det.is_contiguous() and det.dim() == 3
            and det.shape[2] == DET_STRIDE):
        raise RuntimeError("postprocess: det must be a contiguous float32 device tensor [B,K,118]")
    B, K = int(det.shape[0]), int(det.shape[1])
    meta = torch.as_tensor(meta, dtype=torch.float64).reshape(B, 8).contiguous().to(det.device)
    if out is None:
        out = torch.empty(B, K, POST_STRIDE, dtype=torch.float64, device=det.device)
    if cnt is None:
        cnt = torch.empty(B, dtype=torch.int32, device=det.device)
    n = L.cp_postprocess_workspace_bytes(B, K)
    if ws is None:
        ws = torch.empty(n, dtype=torch.uint8, device=det.device)
    if tuple(out.shape) != (B, K, POST_STRIDE) or out.dtype != torch.float64 or cnt.numel() != B or ws.numel() < n:
        raise RuntimeError("postprocess: out / cnt / ws do not fit this batch")
    rc = L.cp_postprocess(_stream(), _ptr(det), B, K, _ptr(meta), float(vis_thresh), int(bool(nms)), float(div_scale),
                          _ptr(out), _ptr(cnt), _ptr(ws), n)
    _check(rc, "cp_postprocess")
    return out, cnt


PNP_STRIDE = 40


def pnp_solve(pts, scale, cam):
    """Batched cuboid PnP.  pts [N,npts,2] float32 (npts 8 or 16), scale [N,3] float32, cam [N,4] float64
    (fx, fy, cx, cy); all on the HIP device.  Returns out [N,40] float64 (layout: centerpose_hip.h)."""
    L = lib()
    if not (pts.is_cuda and scale.is_cuda and cam.is_cuda):
        raise RuntimeError("pnp_solve: tensors must live on the HIP device (no CPU path)")
    pts = pts.contiguous().float()
    scale = scale.contiguous().float()
    cam = cam.contiguous().double()
    N, npts = pts.shape[0], pts.shape[1]
    out = torch.zeros(N, PNP_STRIDE, device=pts.device, dtype=torch.float64)
    if N == 0:
        return out
    n = L.cp_pnp_workspace_bytes(N)
    ws = torch.empty(n, dtype=torch.uint8, device=pts.device)
    _check(L.cp_pnp_solve(_stream(), _ptr(pts), _ptr(scale), _ptr(cam), N, npts, _ptr(out), _ptr(ws), n),
           "cp_pnp_solve")
    return out


_pnp_ws_cache = {}


def pnp_from_post(post, count, cam, rep_mode=1, out=None, ws=None):
    """PnP of every post-processed slot on the device (cp_pnp_from_post): post [B,K,120] float64 + count [B] int32 from
    ``postprocess``, cam [B,4] float64 (fx, fy, cx, cy).  Returns [B,K,40] float64; rows k >= count[b] carry status -1.
    No host synchronisation.  ``out`` / ``ws``: caller-owned result and workspace (default: a fresh result and one
    cached workspace per (shape, device, stream) -- launches on the same stream are ordered, so they may share it)."""
    L = lib()
    B, K = int(post.shape[0]), int(post.shape[1])
    if not (post.is_cuda and post.dtype == torch.float64 and post.is_contiguous() and count.is_cuda and cam.is_cuda):
        raise RuntimeError("pnp_from_post: contiguous device tensors expected (no CPU path)")
    cam = cam.contiguous().double().reshape(B, 4)
    if out is None:
        out = torch.empty(B, K, PNP_STRIDE, dtype=torch.float64, device=post.device)
    n = L.cp_pnp_from_post_workspace_bytes(B, K)
    if ws is None:
        key = (B, K, post.device, torch.cuda.current_stream(post.device).cuda_stream)
        ws = _pnp_ws_cache.get(key)
        if ws is None:
            if len(_pnp_ws_cache) >= 16:
                _pnp_ws_cache.clear()
            ws = _pnp_ws_cache[key] = torch.empty(n, dtype=torch.uint8, device=post.device)
    if tuple(out.shape) != (B, K, PNP_STRIDE) or out.dtype != torch.float64 or ws.numel() < n:
        raise RuntimeError("pnp_from_post: out / ws do not fit this batch")
    _check(L.cp_pnp_from_post(_stream(), _ptr(post), _ptr(count), B, K, int(rep_mode), _ptr(cam), _ptr(out), _ptr(ws), n),
           "cp_pnp_from_post")
    return out


_masked_streams = {}   # (device index, n_cus) -> (handle, torch stream): ONE stream per mask for the life of the process (every
                       # hipExtStreamCreateWithCUMask is a hardware queue of its own; a process that kept creating them would
                       # push its other streams onto shared queues)


def masked_stream(device, n_cus):
    """A HIP stream whose kernels may only run on the first ``n_cus`` compute units (hipExtStreamCreateWithCUMask; bits are spread
    over the XCDs by the runtime), wrapped as a torch stream.  For latency-bound side work that must not take registers away from
    the kernels of the main stream everywhere on the chip: the batched PnP solve holds 286 - 330 vector registers per wavefront
    -- one such wave takes more than half of a SIMD's register file for the ~0.5 ms of its Levenberg-Marquardt walk, and a few
    hundred of them spread over all 1024 SIMDs cost the network kernels they overlap ~0.3 ms per step (profiles/NOTES.md round 6)."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(n_cus))
    if key in _masked_streams:
        return _masked_streams[key][1]
    rt = ctypes.CDLL("libamdhip64.so")
    words = (int(n_cus) + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(int(n_cus)):
        mask[i // 32] |= 1 << (i % 32)
    h = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), mask)
    if rc != 0 or not h.value:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed with code %d" % rc)
    st = torch.cuda.ExternalStream(h.value, device=dev)
    _masked_streams[key] = (h, st)
    return st


class PoseStage(object):
    """Post-process + soft-NMS + batched PnP of decoded batches (base_detector.py:547-654 for a whole batch), with the
    PnP -- and, from 8 images per batch, the post-process -- on a side stream.  The solve is at most B*K independent float64 problems of ~1e5 operations each: a few dozen
    wavefronts whose run time is the slowest lane's Levenberg-Marquardt walk (0.5 - 2.5 ms), during which the rest of
    the chip would idle.  ``submit`` therefore queues the solve (and the post-process of a batch of 8 or more, behind a copy of the
    decoded records) on the stage's own stream, so that the solve of batch i runs under the network of batch i+1; ``depth`` sets of result buffers
    rotate, and the caller's stream waits for the solve that last used a set before post-process overwrites it.

    submit() -> (post [B,K,120], count [B], poses [B,K,40], done): the tensors are valid once ``done`` (a
    torch.cuda.Event) has completed -- ``done.synchronize()`` on the host or ``stream.wait_event(done)``.
    Caller-owned inputs (``det``, ``meta``, ``cam``) are read on the side stream as well: the stage tells the caching
    allocator (``record_stream``), so the caller may drop them right after ``submit``."""

    # batches of at least this many images post-process on the side stream too (below: on the caller's, one hop less per frame);
    # $CP_POST_SIDE_FROM overrides (A/B runs)
    SIDE_POST_FROM = int(os.environ.get("CP_POST_SIDE_FROM", "8"))

    def __init__(self, B, K, device, depth=2, cus=None):
        """``cus``: run the solve on a stream restricted to that many compute units (``masked_stream``); None = $CP_PNP_CUS or,
        unset, 32 (B = 64, 1650 mostly ill-posed detections per batch, profiles/r06_pnp_cu_mask_ab.txt: the first layers of the
        next batch, which the solve overlaps, 1.60 -> 1.45 ms, step 18.74 -> 18.63 ms with the longer solve's tail included; 8 / 16
        CUs stretch the solve to 5.4 / 3.4 ms and lose); 0 = an ordinary stream (the solve's waves land on every CU)."""
        L = lib()
        self.B, self.K, self.depth, self.i = int(B), int(K), int(depth), 0
        if cus is None:
            cus = int(os.environ.get("CP_PNP_CUS", "32"))
        self.cus = int(cus)
        self.side = None
        if self.cus > 0:
            try:
                self.side = masked_stream(device, self.cus)
            except (RuntimeError, OSError, AttributeError):   # runtime without the extension: an ordinary stream does the same work
                self.cus = 0
        if self.side is None:
            self.side = torch.cuda.Stream(device=device)
        n_post, n_pnp = L.cp_postprocess_workspace_bytes(B, K), L.cp_pnp_from_post_workspace_bytes(B, K)
        self.sets = []
        for _ in range(self.depth):
            self.sets.append(dict(
                post=torch.empty(B, K, POST_STRIDE, dtype=torch.float64, device=device),
                cnt=torch.empty(B, dtype=torch.int32, device=device),
                poses=torch.empty(B, K, PNP_STRIDE, dtype=torch.float64, device=device),
                det=torch.empty(B, K, DET_STRIDE, dtype=torch.float32, device=device),
                meta=torch.empty(B, 8, dtype=torch.float64, device=device),
                ws_post=torch.empty(n_post, dtype=torch.uint8, device=device),
                ws_pnp=torch.empty(n_pnp, dtype=torch.uint8, device=device),
                ready=torch.cuda.Event(), done=torch.cuda.Event(enable_timing=True),
                begin=torch.cuda.Event(enable_timing=True)))
        self._timed = []

    def submit(self, det, meta, cam, vis_thresh, nms=True, rep_mode=1):
        s = self.sets[self.i % self.depth]
        first_use = self.i < self.depth
        self.i += 1
        main = torch.cuda.current_stream()
        if not first_use:
            main.wait_event(s["done"])  # the post-process / solve that read this set `depth` batches ago
        B = int(det.shape[0])
        side_post = B >= self.SIDE_POST_FROM
        if side_post:
            # the post-process (one latency-bound launch of ~0.15 ms at B = 64: a serial soft-NMS walk per image) joins the solve on the
            # side stream; the caller's stream only copies the decoded records (3 MB) and the per-image affine into the set, so the
            # caller may overwrite `det` / `meta` with the next batch at once
            s["det"][:B].copy_(det, non_blocking=True)
            s["meta"][:B].copy_(torch.as_tensor(meta, dtype=torch.float64).reshape(B, 8).to(det.device), non_blocking=True)
        else:
            postprocess(det, meta, vis_thresh, nms=nms, out=s["post"], cnt=s["cnt"], ws=s["ws_post"])
        s["ready"].record(main)
        if torch.is_tensor(cam) and cam.is_cuda:
            cam.record_stream(self.side)  # read by the solve after this call returns
        with torch.cuda.stream(self.side):
            self.side.wait_event(s["ready"])
            s["begin"].record(self.side)
            if side_post:
                postprocess(s["det"][:B], s["meta"][:B], vis_thresh, nms=nms, out=s["post"], cnt=s["cnt"], ws=s["ws_post"])
            pnp_from_post(s["post"], s["cnt"], cam, rep_mode=rep_mode, out=s["poses"], ws=s["ws_pnp"])
            s["done"].record(self.side)
        self._timed = [(s["begin"], s["done"])]
        return s["post"], s["cnt"], s["poses"], s["done"]

    def take_solve_ms(self):
        """Milliseconds the most recent assembly + solve took on the side stream (synchronises on it); None before the
        first submit."""
        if not self._timed:
            return None
        b, d = self._timed[-1]
        d.synchronize()
        return b.elapsed_time(d)


TRACK_STRIDE = 520
TRACK_CAP = 128
TRACK_FIELDS = OrderedDict([  # field -> (offset, width) inside a device track record (include/centerpose_hip.h)
    ("tracking_id", (0, 1)), ("age", (1, 1)), ("active", (2, 1)), ("flags", (3, 1)), ("post", (4, 120)),
    ("kps_fusion_mean", (124, 16)), ("kps_fusion_std", (140, 16)), ("location", (156, 3)), ("quaternion_xyzw", (159, 4)),
    ("projected_cuboid", (163, 16)), ("kps_pnp", (179, 18)), ("kps_3d_cam", (197, 27)), ("kps_ori", (224, 18)),
    ("kf_x", (242, 32)), ("kf_P", (274, 128)), ("kps_mean_kf", (409, 16)), ("kps_std_kf", (425, 16)),
    ("obj_scale_kf", (441, 3)), ("obj_scale_uncertainty_kf", (444, 3)), ("conf", (447, 8)), ("kps_pnp_kf", (455, 18)),
    ("kps_3d_cam_kf", (473, 27)), ("kps_ori_kf", (500, 18))])


def track_params_from_opt(opt, K=100, cap=TRACK_CAP):
    """cp_track_params for a reference ``opt`` (opts.py:242-300); raises for what only the host tracker does."""
    baseline = bool(getattr(opt, "refined_Kalman", False))  # Tracker_baseline wins when both flags are set (base_detector.py:53-57)
    if not (getattr(opt, "tracking_task", False) or baseline) or not (opt.kalman or opt.scale_pool):
        raise RuntimeError("device tracker: tracking_task or refined_Kalman, with kalman and / or scale_pool (demo.py:117-129)")
    if getattr(opt, "gt_pre_hm_hmhp", False) or getattr(opt, "gt_pre_hm_hmhp_first", False) or getattr(opt, "empty_pre_hm", False):
        raise RuntimeError("device tracker: ground-truth / empty previous heat-maps are host-only modes")
    cat = {"camera": 0, "bottle": 0, "cup": 0, "book": 1, "chair": 1, "cereal_box": 1, "bike": 2, "laptop": 2, "shoe": 2}
    lo, hi = opt.conf_border[opt.c][0], opt.conf_border[opt.c][1]
    return TrackParams(new_thresh=opt.new_thresh, pre_thresh=opt.pre_thresh, R=opt.R, conf_lo=lo, conf_hi=hi,
                       max_age=int(opt.max_age), kalman=int(bool(opt.kalman)), scale_pool=int(bool(opt.scale_pool)),
                       use_pnp=int(bool(opt.use_pnp)), hps_uncertainty=int(bool(opt.hps_uncertainty)),
                       show_axes=int(bool(opt.show_axes)), cat_rule=cat[opt.c], render_hm_mode=int(opt.render_hm_mode),
                       render_hmhp_mode=int(opt.render_hmhp_mode), pre_hm=int(bool(opt.pre_hm)),
                       pre_hm_hp=int(bool(opt.pre_hm_hp)), K=int(K), cap=int(cap),
                       hungarian=(2 if getattr(opt, "hungarian_solver", "munkres") == "scipy" else 1) if getattr(opt, "hungarian", False) else 0,
                       baseline=int(baseline))


def track_vmeta(metas):
    """[B,16] float64 rows of cp_track_step from the ``meta`` dicts of ``pre_process`` (+ 'camera_matrix')."""
    import numpy as np

    v = np.zeros((len(metas), 16))
    for b, m in enumerate(metas):
        v[b, 0:6] = np.asarray(m["trans_input"], np.float64).reshape(-1)
        v[b, 6:10] = [m["width"], m["height"], m["inp_width"], m["inp_height"]]
        if "camera_matrix" in m:
            K = np.asarray(m["camera_matrix"], np.float64)
            v[b, 10:14] = [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]
    return v


class DeviceTracker(object):
    """CenterPoseTrack's per-video track tables on the device (cp_track_*): ``step`` consumes the outputs of
    ``postprocess`` / ``pnp_from_post`` of a frame of B videos, ``render`` draws the next frame's pre_hm / pre_hm_hp from
    the tracks, ``read`` copies the current lists to the host.  ``read`` synchronises; so does ``step`` once every
    ``STATUS_EVERY`` frames, when it looks at the overflow counters (``check``) and raises if a frame needed more than ``cap``
    tracks -- AFTER the device state has advanced by that frame.  Set ``STATUS_EVERY = 0`` (class or instance attribute) for a
    loop that must never synchronise or that is being captured into a hipGraph, and poll ``dropped()`` / ``check()`` yourself."""

    STATUS_EVERY = 64  # frames between two looks at the overflow counters in step() (each look synchronises the stream); 0 = never

    def __init__(self, B, params, vmeta, device, inp_h, inp_w):
        L = lib()
        self.frames = 0
        self.B, self.P, self.device = int(B), params, device
        self.K, self.cap = int(params.K), int(params.cap)
        self.inp_h, self.inp_w = int(inp_h), int(inp_w)
        self.vmeta = torch.as_tensor(vmeta, dtype=torch.float64).reshape(self.B, 16).contiguous().to(device)
        n_state, n_ws = L.cp_track_state_bytes(self.B, self.cap), L.cp_track_workspace_bytes(self.B, self.K, self.cap)
        if n_state == 0 or n_ws == 0:
            raise RuntimeError("DeviceTracker: unsupported B / K / cap")
        self.state = torch.empty(n_state, dtype=torch.uint8, device=device)
        self.ws = torch.empty(n_ws, dtype=torch.uint8, device=device)
        self.recs = torch.empty(self.B, self.cap, 9, 5, dtype=torch.float64, device=device)
        self.planes = torch.empty(9 * self.B, self.inp_h, self.inp_w, dtype=torch.float32, device=device)
        self.hdr_bytes = ((4 + 4 * self.B) * 4 + 255) // 256 * 256
        self.reset()

    def reset(self):
        _check(lib().cp_track_reset(_stream(), _ptr(self.state), self.B, self.cap), "cp_track_reset")
        self.recs[..., 0] = -1.0  # nothing to draw before the first frame
        self.recs[..., 1:] = 0.0

    def step(self, post, count, det_pnp=None, check=True):
        """One frame of every video.  Raises BEFORE the device state has moved if the arguments or the launch are refused;
        with ``check`` (default) the periodic look at the overflow counters follows and may raise AFTER it has moved -- a caller
        that keeps per-frame state of its own passes ``check=False``, updates that state, then calls ``check_due()``."""
        if not (post.is_cuda and post.dtype == torch.float64 and post.is_contiguous() and tuple(post.shape) ==
                (self.B, self.K, POST_STRIDE) and count.is_cuda and count.dtype == torch.int32):
            raise RuntimeError("DeviceTracker.step: post [B,K,120] float64 / count [B] int32 device tensors expected")
        if det_pnp is not None and not (det_pnp.is_cuda and det_pnp.dtype == torch.float64 and det_pnp.is_contiguous() and
                                        tuple(det_pnp.shape) == (self.B, self.K, PNP_STRIDE)):
            raise RuntimeError("DeviceTracker.step: det_pnp must be the [B,K,40] float64 output of pnp_from_post")
        _check(lib().cp_track_step(_stream(), ctypes.byref(self.P), _ptr(self.vmeta), _ptr(post), _ptr(count), _ptr(det_pnp),
                                   self.B, _ptr(self.state), _ptr(self.recs), _ptr(self.ws), self.ws.numel()), "cp_track_step")
        self.frames += 1
        if check:
            self.check_due()

    def check_due(self):
        """The periodic overflow check of ``step`` (every STATUS_EVERY frames; synchronises when it runs)."""
        if self.STATUS_EVERY and self.frames % self.STATUS_EVERY == 0:  # the device-resident loop never calls read(): surface overflows here
            self.check()

    def dropped(self):
        """Per video: list entries dropped so far because a frame needed more than `cap` tracks (cp_track_status; the
        tracker then keeps the first `cap` entries in the reference's order -- matched, new by score, coasting)."""
        out = (ctypes.c_int * self.B)()
        _check(lib().cp_track_status(_stream(), _ptr(self.state), self.B, out), "cp_track_status")
        return list(out)

    def check(self):
        d = self.dropped()
        if any(d):
            raise RuntimeError("DeviceTracker: more than cap = %d tracks in a frame of video(s) %s (entries dropped: %s); "
                               "raise cap or the thresholds" % (self.cap, [b for b, v in enumerate(d) if v], [v for v in d if v]))

    def render(self):
        """-> (pre_hm [B,1,H,W], pre_hm_hp [B,8,H,W]) drawn from the current tracks (views of one plane buffer)."""
        B, H, W = self.B, self.inp_h, self.inp_w
        _check(lib().cp_render_gaussians(_stream(), _ptr(self.recs), B * self.cap * 9, _ptr(self.planes), 9 * B, H, W, 1),
               "cp_render_gaussians")
        return self.planes[:B].view(B, 1, H, W), self.planes[B:].view(B, 8, H, W)

    def read(self):
        """Host copy of the current lists: a list of B float64 arrays [n_b, 520] (layout: TRACK_FIELDS)."""
        import numpy as np

        raw = self.state.cpu().numpy()
        hdr = raw[: (4 + 4 * self.B) * 4].view(np.int32)
        tr = raw[self.hdr_bytes:].view(np.float64).reshape(2, self.B, self.cap, TRACK_STRIDE)
        out = []
        for b in range(self.B):
            n, overflow = int(hdr[4 + 4 * b]), int(hdr[4 + 4 * b + 2])
            if overflow:
                raise RuntimeError("DeviceTracker: video %d needed more than %d tracks (%d list entries dropped)" % (b, self.cap, overflow))
            out.append(tr[int(hdr[0]), b, :n].copy())
        return out


def track_record_to_dict(r, opt=None):
    """One device track record -> the reference's per-track dict (the keys `Tracker.step` leaves on a track)."""
    import numpy as np

    d = {}
    post = r[4:124]
    for k, (off, w) in POST_FIELDS.items():
        v = post[off:off + w]
        d[k] = float(v[0]) if k == "score" else int(v[0]) if k == "cls" else [v[0], v[1]] if k == "ct" else v.copy()
    flags = int(r[3])
    d.update(tracking_id=int(r[0]), age=int(r[1]), active=int(r[2]))
    f = lambda k: r[TRACK_FIELDS[k][0]:TRACK_FIELDS[k][0] + TRACK_FIELDS[k][1]].copy()
    d["kps_fusion_mean"], d["kps_fusion_std"] = f("kps_fusion_mean"), f("kps_fusion_std")
    d["kps_mean_kf"], d["kps_std_kf"] = f("kps_mean_kf").reshape(8, 2), list(f("kps_std_kf"))
    d["obj_scale_kf"], d["obj_scale_uncertainty_kf"] = f("obj_scale_kf"), f("obj_scale_uncertainty_kf")
    if flags & 1:
        d["location"], d["quaternion_xyzw"] = list(f("location")), f("quaternion_xyzw")
        d["projected_cuboid"] = f("projected_cuboid").reshape(8, 2)
        d["kps_pnp"], d["kps_3d_cam"] = f("kps_pnp").reshape(9, 2), f("kps_3d_cam").reshape(9, 3)
    if flags & 8:
        d["kps_ori"] = f("kps_ori").reshape(9, 2)
    if flags & 2:
        d["kps_pnp_kf"], d["kps_3d_cam_kf"] = f("kps_pnp_kf").reshape(9, 2), f("kps_3d_cam_kf").reshape(9, 3)
        d["kps_ori_kf"] = f("kps_ori_kf").reshape(9, 2)
    d["in_boxes"] = bool(flags & 4)
    return d


class HipModel(object):
    """Device-resident DLA-34 / DLA-34+ConvGRU network built from a reference-format state dict."""

    def __init__(self, arch, heads, state_dict, tracking_task=False, head_conv=256, precision=None):
        L = lib()
        self.arch = arch
        self.heads = OrderedDict(heads)
        self.tracking_task = bool(tracking_task)
        names = (c_char_p * len(self.heads))(*[k.encode() for k in self.heads])
        classes = (c_int * len(self.heads))(*[int(v) for v in self.heads.values()])
        h = c_void_p()
        _check(L.cp_model_create(arch.encode(), int(self.tracking_task), len(self.heads), names, classes,
                                 int(head_conv), ctypes.byref(h)), "cp_model_create")
        self._h = h
        for k, v in state_dict.items():
            if k.startswith("module.") and not k.startswith("module_list"):
                k = k[7:]  # lib/models/model.py:43-48
            if not torch.is_floating_point(v):
                continue  # num_batches_tracked
            t = v.detach().cpu().contiguous().float()
            _check(L.cp_model_set_param(h, k.encode(), c_void_p(t.data_ptr()), t.numel()), "cp_model_set_param")
        _check(L.cp_model_finalize(h), "cp_model_finalize")
        if precision is not None:
            self.set_precision(precision)
        self._ws = None
        self._ws_key = None

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and _lib is not None:
                _lib.cp_model_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_precision(self, name):
        _check(lib().cp_model_set_precision(self._h, PRECISIONS[name]), "cp_model_set_precision")
        self.precision = name

    def profile(self, enable=True):
        """Arm / disarm per-launch HIP-event timing of the implicit-GEMM kernels."""
        _check(lib().cp_model_profile(self._h, int(bool(enable))), "cp_model_profile")

    def profile_read(self):
        """-> {kernel name: dict(launches, ms, flops, bytes)} accumulated since the last read."""
        nv = lib().cp_num_kernel_variants()  # sized by the library, not by a copy of CP_NUM_KERNEL_VARIANTS
        buf = (ctypes.c_double * (nv * 4))()
        _check(lib().cp_model_profile_read(self._h, buf, nv), "cp_model_profile_read")
        out = OrderedDict()
        for v in range(nv):
            if buf[v * 4] > 0:
                out[lib().cp_kernel_variant_name(v).decode()] = dict(
                    launches=int(buf[v * 4]), ms=buf[v * 4 + 1], flops=buf[v * 4 + 2], bytes=buf[v * 4 + 3])
        return out

    def profile_roles(self):
        """-> {role: dict(launches, ms, flops, bytes)} of the launches drained by the last profile_read()."""
        nr = lib().cp_num_roles()
        buf = (ctypes.c_double * (nr * 4))()
        _check(lib().cp_model_profile_roles(self._h, buf, nr), "cp_model_profile_roles")
        out = OrderedDict()
        for r in range(nr):
            if buf[r * 4] > 0:
                out[lib().cp_role_name(r).decode()] = dict(
                    launches=int(buf[r * 4]), ms=buf[r * 4 + 1], flops=buf[r * 4 + 2], bytes=buf[r * 4 + 3])
        return out

    def workspace_bytes(self, B, H, W):
        n = lib().cp_model_workspace_bytes(self._h, B, H, W)
        if n == 0:
            raise RuntimeError("cp_model_workspace_bytes failed: " + lib().cp_last_error().decode())
        return n

    def _workspace(self, B, H, W, device):
        key = (B, H, W, str(device))
        if self._ws_key != key:
            self._ws = None
            n = self.workspace_bytes(B, H, W)
            self._ws = torch.empty(n, dtype=torch.uint8, device=device)
            self._ws_key = key
        return self._ws

    def forward(self, images, pre_img=None, pre_hm=None, pre_hm_hp=None, sigmoid_hm=False, tap=None):
        """images [B,3,H,W] on the HIP device -> OrderedDict head -> [B,classes,H/4,W/4].
        With ``tap`` also returns the named intermediate activation as NCHW."""
        L = lib()
        images = _dev(images)
        B, _, H, W = images.shape
        pre_img = _dev(pre_img) if pre_img is not None else None
        pre_hm = _dev(pre_hm) if pre_hm is not None else None
        pre_hm_hp = _dev(pre_hm_hp) if pre_hm_hp is not None else None
        outs = OrderedDict()
        for k, c in self.heads.items():
            outs[k] = torch.empty(B, c, H // 4, W // 4, device=images.device, dtype=torch.float32)
        ptrs = (c_void_p * len(outs))(*[t.data_ptr() for t in outs.values()])
        ws = self._workspace(B, H, W, images.device)
        if tap is None:
            rc = L.cp_model_forward(self._h, _stream(), B, H, W, _ptr(images), _ptr(pre_img), _ptr(pre_hm),
                                    _ptr(pre_hm_hp), ptrs, int(bool(sigmoid_hm)), _ptr(ws), ws.numel())
            _check(rc, "cp_model_forward")
            return outs
        tap_buf = torch.zeros(B * 512 * (H // 4) * (W // 4) if False else B * 16 * H * W, device=images.device,
                              dtype=torch.float32)
        dims = (c_int * 3)(0, 0, 0)
        rc = L.cp_model_forward_tap(self._h, _stream(), B, H, W, _ptr(images), _ptr(pre_img), _ptr(pre_hm),
                                    _ptr(pre_hm_hp), ptrs, int(bool(sigmoid_hm)), _ptr(ws), ws.numel(),
                                    tap.encode(), _ptr(tap_buf), dims)
        _check(rc, "cp_model_forward_tap")
        C, h, w = dims[0], dims[1], dims[2]
        if C == 0:
            raise RuntimeError("unknown tap %r" % tap)
        return outs, tap_buf[: B * C * h * w].view(B, C, h, w)

    def detect(self, images, pre_img=None, pre_hm=None, pre_hm_hp=None, K=100, rep_mode=1, fit_gaussian=False,
               balance=2.0, legacy_bool_mask=False, graph=True):
        """backbone + heads + sigmoid + decode in one library call -> (heads dict, det [B,K,118]).
        Output tensors are owned by the model and REUSED by the next call with the same batch shape (that is what lets
        the launch sequence be replayed from a hipGraph).  With ``graph`` the caller must run on a non-default stream
        and pass the same input tensors (copy new frames into them)."""
        L = lib()
        B, _, H, W = images.shape
        key = ("det", B, H, W, str(images.device), K)
        st = getattr(self, "_det_state", None)
        if st is None or st[0] != key:
            outs = OrderedDict((k, torch.empty(B, c, H // 4, W // 4, device=images.device, dtype=torch.float32))
                               for k, c in self.heads.items())
            det = torch.empty(B, K, DET_STRIDE, device=images.device, dtype=torch.float32)
            n = L.cp_model_detect_workspace_bytes(self._h, B, H, W, K)
            if n == 0:
                raise RuntimeError("cp_model_detect_workspace_bytes failed: " + L.cp_last_error().decode())
            ws = torch.empty(n, dtype=torch.uint8, device=images.device)
            ptrs = (c_void_p * len(outs))(*[t.data_ptr() for t in outs.values()])
            st = (key, outs, det, ws, ptrs)
            self._det_state = st
        _, outs, det, ws, ptrs = st
        for t in (images, pre_img, pre_hm, pre_hm_hp):
            if t is not None and not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                raise RuntimeError("detect: inputs must be contiguous float32 device tensors")
        rc = L.cp_model_detect(self._h, _stream(), B, H, W, _ptr(images), _ptr(pre_img), _ptr(pre_hm), _ptr(pre_hm_hp),
                               ptrs, int(K), int(rep_mode), int(bool(fit_gaussian)), float(balance),
                               int(bool(legacy_bool_mask)), _ptr(det), _ptr(ws), ws.numel(), int(bool(graph)))
        _check(rc, "cp_model_detect")
        return outs, det

    __call__ = forward
