"""Affine helpers of the reference (utils/image.py:23-74) without OpenCV: ``cv2.getAffineTransform`` is a
3-point, 6-unknown linear solve, done here with numpy in float64 (the reference's result type)."""
import numpy as np


def _affine_from_3pts(src, dst):
    """2x3 matrix M with M @ [x, y, 1] = (u, v) for the three correspondences (cv2.getAffineTransform)."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        A[2 * i, 0:3] = [src[i, 0], src[i, 1], 1.0]
        A[2 * i + 1, 3:6] = [src[i, 0], src[i, 1], 1.0]
        b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(A, b).reshape(2, 3)


def get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def get_3rd_point(a, b):
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    """utils/image.py:35-68"""
    if not isinstance(scale, np.ndarray) and not isinstance(scale, list):
        scale = np.array([scale, scale], dtype=np.float32)
    src_w = scale[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = get_dir([0, src_w * -0.5], rot_rad)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale * shift
    src[1, :] = center + src_dir + scale * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    src[2:, :] = get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = get_3rd_point(dst[0, :], dst[1, :])
    if inv:
        return _affine_from_3pts(np.float32(dst), np.float32(src))
    return _affine_from_3pts(np.float32(src), np.float32(dst))


def affine_transform(pt, t):
    new_pt = np.array([pt[0], pt[1], 1.], dtype=np.float32).T
    return np.dot(t, new_pt)[:2]


def transform_preds(coords, center, scale, output_size):
    """utils/image.py:23-32: (-10000, -10000) marks a missing point and passes through."""
    target = np.zeros(coords.shape)
    trans = get_affine_transform(center, scale, 0, output_size, inv=1)
    for p in range(coords.shape[0]):
        if coords[p, 0] == -10000 and coords[p, 1] == -10000:
            target[p, 0:2] = [-10000, -10000]
        else:
            target[p, 0:2] = affine_transform(coords[p, 0:2], trans)
    return target


def warp_affine_bilinear(img, trans, out_w, out_h):
    """Host stand-in for ``cv2.warpAffine(img, trans, (out_w, out_h), flags=INTER_LINEAR)`` (constant-0
    border).  cv2 uses fixed-point 1/32 interpolation weights; this is the float32 bilinear form, so
    pixel values can differ from cv2 by a few 1/255 steps (pre-processing is a "next" row)."""
    M = np.vstack([np.asarray(trans, np.float64), [0, 0, 1]])
    Minv = np.linalg.inv(M)
    xs, ys = np.meshgrid(np.arange(out_w, dtype=np.float64), np.arange(out_h, dtype=np.float64))
    sx = Minv[0, 0] * xs + Minv[0, 1] * ys + Minv[0, 2]
    sy = Minv[1, 0] * xs + Minv[1, 1] * ys + Minv[1, 2]
    x0 = np.floor(sx).astype(np.int64)
    y0 = np.floor(sy).astype(np.int64)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
    H, W = img.shape[:2]
    img = img.astype(np.float32)
    if img.ndim == 2:
        img = img[..., None]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return v * ok[..., None]

    out = (tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy) +
           tap(y0 + 1, x0) * (1 - fx) * fy + tap(y0 + 1, x0 + 1) * fx * fy)
    return out


def gaussian_radius(det_size, min_overlap=0.7):
    """CornerNet radius rule (utils/image.py:103-123): the smallest of the three quadratic roots."""
    h, w = det_size
    o = min_overlap
    roots = []
    for a, b, c in ((1, h + w, w * h * (1 - o) / (1 + o)), (4, 2 * (h + w), (1 - o) * w * h),
                    (4 * o, -2 * o * (h + w), (o - 1) * w * h)):
        roots.append((b + np.sqrt(b ** 2 - 4 * a * c)) / 2)
    return min(roots)


def draw_gaussian_records(records, C, H, W):
    """Host renderer with the semantics of cp_render_gaussians / draw_umich_gaussian (utils/image.py:126-150):
    records (channel, x, y, radius, k) -> float32 [C,H,W], merged with max(), clipped to the map."""
    out = np.zeros((C, H, W), np.float32)
    for ch, x, y, radius, k in records:
        ch, x, y, radius = int(ch), int(x), int(y), int(radius)
        d = 2 * radius + 1
        yy, xx = np.ogrid[-radius:radius + 1, -radius:radius + 1]
        sigma = d / 6
        g = np.exp(-(xx * xx + yy * yy) / (2 * sigma * sigma))
        g[g < np.finfo(g.dtype).eps * g.max()] = 0
        left, right = min(x, radius), min(W - x, radius + 1)
        top, bottom = min(y, radius), min(H - y, radius + 1)
        dst = out[ch, y - top:y + bottom, x - left:x + right]
        src = g[radius - top:radius + bottom, radius - left:radius + right]
        if min(src.shape) > 0 and min(dst.shape) > 0:
            np.maximum(dst, src * k, out=dst)
    return out
