"""ORACLE (test infrastructure): DCNv2 forward on the CPU.

``dcn_v2_forward`` restates the reference's CPU orchestration
(/root/reference/src/lib/models/networks/DCNv2/src/cpu/dcn_v2_cpu.cpp:40-105):
per sample, ``out = bias (broadcast)``, ``columns = im2col(...)``,
``out += W.view(Co, -1) @ columns``.  The im2col step is either this repo's C
restatement (``oracle/dcn_im2col.c`` -> ``libcp_oracle.so``, kind="port") or the
reference's own unmodified source compiled into ``oracle/_ref/libdcn_im2col_ref.so``
(kind="reference").  ``torch.addmm`` stands in for ``THFloatBlas_gemm``.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT = os.path.join(_HERE, "libcp_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libdcn_im2col_ref.so")

_IM2COL_ARGS = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 15 + [ctypes.c_void_p]


def build(verbose=False):
    """(Re)build the oracle shared objects with oracle/Makefile (gcc only)."""
    out = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)


_libs = {}


def _lib(kind):
    if kind in _libs:
        return _libs[kind]
    if kind == "port":
        if not os.path.exists(_PORT):
            build()
        lib = ctypes.CDLL(_PORT)
        fn = lib.cp_oracle_dcn_im2col
    elif kind == "reference":
        if not os.path.exists(_REF):
            raise FileNotFoundError(_REF + " (built only where /root/reference exists)")
        lib = ctypes.CDLL(_REF)
        fn = lib.modulated_deformable_im2col_cpu  # dcn_v2_im2col_cpu.cpp:331
    else:
        raise ValueError(kind)
    fn.argtypes = _IM2COL_ARGS
    fn.restype = None
    _libs[kind] = fn
    return fn


def have_reference():
    return os.path.exists(_REF)


def im2col(x, offset, mask, kh, kw, ph, pw, sh, sw, dh, dw, dg, kind="port"):
    """x [B,C,H,W], offset [B,dg*2*kh*kw,Ho,Wo], mask [B,dg*kh*kw,Ho,Wo] -> [B, C*kh*kw, Ho*Wo]."""
    x = x.contiguous().float()
    offset = offset.contiguous().float()
    mask = mask.contiguous().float()
    B, C, H, W = x.shape
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    col = torch.empty(B, C * kh * kw, Ho * Wo, dtype=torch.float32)
    fn = _lib(kind)
    # argument order of dcn_v2_im2col_cpu.h:68-80
    fn(x.data_ptr(), offset.data_ptr(), mask.data_ptr(), B, C, H, W, Ho, Wo,
       kh, kw, ph, pw, sh, sw, dh, dw, dg, col.data_ptr())
    return col, Ho, Wo


def dcn_v2_forward(x, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, dg,
                   kind="port"):
    """Same 14-argument signature as the reference's ``_ext.dcn_v2_forward`` (vision.cpp:5)."""
    B = x.shape[0]
    Co = weight.shape[0]
    assert weight.shape[2] == kh and weight.shape[3] == kw
    assert weight.shape[1] == x.shape[1]
    col, Ho, Wo = im2col(x, offset, mask, kh, kw, ph, pw, sh, sw, dh, dw, dg, kind=kind)
    w2 = weight.reshape(Co, -1).float()
    out = torch.empty(B, Co, Ho * Wo, dtype=torch.float32)
    for b in range(B):  # per-sample loop as dcn_v2_cpu.cpp:68
        out[b] = torch.addmm(bias.float().view(Co, 1), w2, col[b])
    return out.view(B, Co, Ho, Wo)


def dcn_v2_forward_f64(x, weight, bias, offset, mask, pad=1):
    """Independent float64 numpy restatement (3x3, stride 1, dil 1, dg 1) used to bound the
    float32 oracle's own rounding error in tests.  Vectorised gather, no C."""
    x = x.double().numpy()
    offset = offset.double().numpy()
    mask = mask.double().numpy()
    w = weight.double().numpy()
    B, C, H, W = x.shape
    Co = w.shape[0]
    out = np.zeros((B, Co, H, W))
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for b in range(B):
        acc = np.zeros((Co, H, W))
        for t in range(9):
            i, j = divmod(t, 3)
            h_im = ys - pad + i + offset[b, 2 * t]
            w_im = xs - pad + j + offset[b, 2 * t + 1]
            valid = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
            h0 = np.floor(h_im).astype(np.int64)
            w0 = np.floor(w_im).astype(np.int64)
            lh = h_im - h0
            lw = w_im - w0
            val = np.zeros((C, H, W))
            for (hh, ww, wt) in ((h0, w0, (1 - lh) * (1 - lw)), (h0, w0 + 1, (1 - lh) * lw),
                                 (h0 + 1, w0, lh * (1 - lw)), (h0 + 1, w0 + 1, lh * lw)):
                ok = valid & (hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)
                hc = np.clip(hh, 0, H - 1)
                wc = np.clip(ww, 0, W - 1)
                val += x[b][:, hc, wc] * (wt * ok)[None]
            val *= mask[b, t][None]
            acc += np.einsum("oc,chw->ohw", w[:, :, i, j], val)
        out[b] = acc + bias.double().numpy()[:, None, None]
    return torch.from_numpy(out)
