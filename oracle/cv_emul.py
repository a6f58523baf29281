"""ORACLE (test infrastructure): integer emulation of the two OpenCV image operations on the pre-process path,
``cv2.warpAffine(img, M, dsize, flags=INTER_LINEAR)`` and ``cv2.resize(img, dsize)`` on 8-bit images
(/root/reference/src/lib/detectors/base_detector.py:127-131).

PARITY UNPINNED: OpenCV (opencv-python>=4.5.3.56, requirements.txt:11) is an un-vendored dependency that is absent here
and the reference holds no fixture for this stage.  Restated from OpenCV imgproc's published fixed-point algorithms:

warpAffine (imgwarp.cpp, WarpAffineInvoker + remapBilinear<FixedPtCast<int, uchar, 15>>):
  * the forward matrix is inverted in float64 (invertAffineTransform: D = 1 / (m00 m11 - m01 m10), ...);
  * source coordinates are computed in fixed point with AB_BITS = 10: adelta[x] = round(m00 * x * 1024),
    bdelta[x] = round(m10 * x * 1024), X0 = round((m01 * y + m02) * 1024) + 16, Y0 likewise (16 = half a 1/32 step),
    X = (X0 + adelta[x]) >> 5 carries INTER_BITS = 5 fractional bits (round = lrint: half to even);
  * the four taps are weighted with integer weights (32 - fx)(32 - fy) * 32 ... (they sum to 2^15 exactly for the
    bilinear table, so the table's sum correction never fires), result = (sum + 2^14) >> 15; taps outside the image
    read the constant border 0.
resize (resize.cpp, HResizeLinear + VResizeLinear for uchar, INTER_RESIZE_COEF_BITS = 11):
  * fx = (dx + 0.5) * (src / dst) - 0.5, sx = floor(fx), clamped at both ends; coefficients round(w * 2048) as int16;
  * horizontal pass in int32, vertical pass dst = ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
Self-checks (tests/test_cv_emul.py): the identity and integer translations reproduce the image, a 2x up-sample of a
ramp is the exact ramp, and both agree with float bilinear interpolation to the quantisation bound stated there.
"""
import numpy as np


def invert_affine(M):
    """cv::invertAffineTransform in float64."""
    M = np.asarray(M, np.float64).reshape(2, 3)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    A12, A21 = -M[0, 1] * D, -M[1, 0] * D
    b1 = -A11 * M[0, 2] - A12 * M[1, 2]
    b2 = -A21 * M[0, 2] - A22 * M[1, 2]
    return np.array([[A11, A12, b1], [A21, A22, b2]])


def warp_affine_u8(img, M, dsize):
    """img [H,W] or [H,W,C] uint8, M 2x3 forward (source -> destination), dsize = (width, height) -> uint8."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    squeeze = img.ndim == 2
    if squeeze:
        img = img[..., None]
    H, W, C = img.shape
    ow, oh = int(dsize[0]), int(dsize[1])
    Mi = invert_affine(M)
    xs = np.arange(ow, dtype=np.float64)
    ys = np.arange(oh, dtype=np.float64)
    adelta = np.rint(Mi[0, 0] * xs * 1024).astype(np.int64)
    bdelta = np.rint(Mi[1, 0] * xs * 1024).astype(np.int64)
    X0 = np.rint((Mi[0, 1] * ys + Mi[0, 2]) * 1024).astype(np.int64) + 16
    Y0 = np.rint((Mi[1, 1] * ys + Mi[1, 2]) * 1024).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = X >> 5, Y >> 5
    fx, fy = X & 31, Y & 31
    acc = np.zeros((oh, ow, C), np.int64)
    for dy, wy in ((0, 32 - fy), (1, fy)):
        for dx, wx in ((0, 32 - fx), (1, fx)):
            yy, xx = sy + dy, sx + dx
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64) * ok[..., None]
            acc += v * (wy * wx * 32)[..., None]
    out = ((acc + (1 << 14)) >> 15).astype(np.uint8)
    return out[..., 0] if squeeze else out


def _resize_axis(dst, src):
    """Coefficients of one axis in OpenCV's order of operations (resize.cpp, INTER_LINEAR): scale = 1. / (dst / src) in
    double, fx = (float)((d + 0.5) * scale - 0.5) rounded to float BEFORE cvFloor, fx -= sx in float, border clamps,
    then cvRound(coef * 2048).  PARITY UNPINNED (cv2 absent): restated from the published source, not run against it."""
    scale = 1.0 / (float(dst) / float(src))
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0, src - 1
    a0 = np.rint((np.float32(1.0) - f).astype(np.float32) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, np.minimum(s + 1, src - 1), a0, a1


def resize_linear_u8(img, dsize):
    """cv2.resize(img, dsize) (INTER_LINEAR) for uint8; dsize = (width, height)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    squeeze = img.ndim == 2
    if squeeze:
        img = img[..., None]
    H, W, C = img.shape
    ow, oh = int(dsize[0]), int(dsize[1])
    if (ow, oh) == (W, H):
        return img[..., 0].copy() if squeeze else img.copy()
    x0, x1, ax0, ax1 = _resize_axis(ow, W)
    y0, y1, ay0, ay1 = _resize_axis(oh, H)
    src = img.astype(np.int64)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]      # [H, ow, C] horizontal pass
    S0, S1 = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * (S0 >> 4)) >> 16) + ((ay1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[..., 0] if squeeze else out
