"""ORACLE (test infrastructure): numpy restatement of the reference's heat-map decode.

Follows /root/reference/src/lib/models/decode.py (`_nms` :17-23, `_topk_channel` :40-49, `_topk`
:52-68, `object_pose_decode` :72-375), /root/reference/src/lib/models/utils.py
(`_transpose_and_gather_feat` :43-47) and /root/reference/src/lib/utils/gpfit.py (:13-41).
All tensor arithmetic is float32 in the reference's operation order; the per-point loop
(decode.py:209-252) is numpy/python in the reference too and is restated literally.

Two deliberate knobs:
  * ``mask_semantics``: "uint8" = torch<=1.1 behaviour the published weights were used with
    (comparison results add, ``== 7`` is the AND of 7 conditions); "bool" = what the unmodified
    reference computes under torch>=1.2 (``bool+bool`` stays bool, ``== 7`` is always False).
  * top-K order among exactly equal scores is implementation-defined in torch; this restatement
    uses (value desc, index asc).  Parity is asserted on entries with score > 0 and distinct values.

Pinned by tests/test_oracle_pins.py against the reference function imported from /root/reference
(build container) and by the committed fixtures tests/golden/decode_*.npz.
"""
import numpy as np

NEG = -10000.0


def nms(heat):
    """decode.py:17-23: 3x3 stride-1 max-pool with -inf padding; keep where equal."""
    B, C, H, W = heat.shape
    pad = np.full((B, C, H + 2, W + 2), -np.inf, dtype=heat.dtype)
    pad[:, :, 1:-1, 1:-1] = heat
    hmax = pad[:, :, 0:H, 0:W].copy()
    for dy in range(3):
        for dx in range(3):
            np.maximum(hmax, pad[:, :, dy:dy + H, dx:dx + W], out=hmax)
    keep = (hmax == heat).astype(heat.dtype)
    return heat * keep


def _topk_rows(scores, K):
    """top-K of every row of a 2-D array: (value desc, index asc)."""
    order = np.argsort(-scores, axis=1, kind="stable")[:, :K]
    vals = np.take_along_axis(scores, order, axis=1)
    return vals, order


def topk_channel(scores, K):
    """decode.py:40-49"""
    B, C, H, W = scores.shape
    vals, inds = _topk_rows(scores.reshape(B * C, H * W), K)
    vals = vals.reshape(B, C, K)
    inds = inds.reshape(B, C, K) % (H * W)
    ys = (inds // W).astype(np.float32)
    xs = (inds % W).astype(np.float32)
    return vals, inds, ys, xs


def topk(scores, K):
    """decode.py:52-68"""
    B, C, H, W = scores.shape
    vals, inds, ys, xs = topk_channel(scores, K)
    score, ind = _topk_rows(vals.reshape(B, C * K), K)
    clses = (ind // K).astype(np.int32)
    inds = np.take_along_axis(inds.reshape(B, C * K), ind, axis=1)
    ys = np.take_along_axis(ys.reshape(B, C * K), ind, axis=1)
    xs = np.take_along_axis(xs.reshape(B, C * K), ind, axis=1)
    return score, inds, clses, ys, xs


def gather_feat(feat, ind):
    """_transpose_and_gather_feat (utils.py:43-47): feat [B,C,H,W], ind [B,N] -> [B,N,C]."""
    B, C, H, W = feat.shape
    f = feat.reshape(B, C, H * W).transpose(0, 2, 1)
    return np.take_along_axis(f, ind[:, :, None].astype(np.int64), axis=1)


def moments(data):
    """gpfit.py:13-26 (restated literally, including the x/y naming quirk)."""
    total = data.sum()
    X, Y = np.indices(data.shape)
    x = (X * data).sum() / total
    y = (Y * data).sum() / total
    col = data[:, int(y)]
    width_x = np.sqrt(np.abs((np.arange(col.size) - y) ** 2 * col).sum() / col.sum())
    row = data[int(x), :]
    width_y = np.sqrt(np.abs((np.arange(row.size) - x) ** 2 * row).sum() / row.sum())
    height = data.max()
    return height, x, y, width_x, width_y


def fitgaussian(data):
    """gpfit.py:29-41: scipy least_squares with max_nfev=1 (returns the strictly-feasible start)."""
    from scipy import optimize

    params = moments(data)

    def gaussian(height, center_x, center_y, width_x, width_y):
        width_x = float(width_x)
        width_y = float(width_y)
        return lambda x, y: height * np.exp(-(((center_x - x) / width_x) ** 2 + ((center_y - y) / width_y) ** 2) / 2)

    def errorfunction(p):
        return np.ravel(gaussian(*p)(*np.indices(data.shape)) - data)

    res = optimize.least_squares(errorfunction, params,
                                 bounds=(0, [np.inf, data.shape[0], data.shape[1], np.inf, np.inf]), max_nfev=1)
    return res.x


def object_pose_decode(heat, kps, wh=None, kps_displacement_std=None, obj_scale=None,
                       obj_scale_uncertainty=None, reg=None, hm_hp=None, hp_offset=None, tracking=None,
                       tracking_hp=None, K=100, rep_mode=1, tracking_task=False, refined_Kalman=False,
                       balance_coefficient=2.0, mask_semantics="uint8"):
    """decode.py:72-375 with Inference=True and wh / hm_hp given (the detector's configuration,
    object_pose.py:140-157).  Inputs are numpy float32 NCHW, heat / hm_hp already sigmoided."""
    f32 = np.float32
    heat = np.asarray(heat, f32)
    kps = np.asarray(kps, f32)
    B, cat, H, W = heat.shape
    J = kps.shape[1] // 2
    assert wh is not None and hm_hp is not None, "oracle covers the detector's Inference configuration"

    heat = nms(heat)
    scores, inds, clses, ys, xs = topk(heat, K)

    kps = gather_feat(kps, inds).reshape(B, K, J * 2).copy()
    kps[..., ::2] += xs.reshape(B, K, 1)
    kps[..., 1::2] += ys.reshape(B, K, 1)
    if reg is not None:
        r = gather_feat(np.asarray(reg, f32), inds).reshape(B, K, 2)
        xs = xs.reshape(B, K, 1) + r[:, :, 0:1]
        ys = ys.reshape(B, K, 1) + r[:, :, 1:2]
    else:
        xs = xs.reshape(B, K, 1) + f32(0.5)
        ys = ys.reshape(B, K, 1) + f32(0.5)
    clses = clses.reshape(B, K, 1).astype(f32)
    scores = scores.reshape(B, K, 1)

    whg = gather_feat(np.asarray(wh, f32), inds).reshape(B, K, 2)
    bboxes = np.concatenate([xs - whg[..., 0:1] / 2, ys - whg[..., 1:2] / 2,
                             xs + whg[..., 0:1] / 2, ys + whg[..., 1:2] / 2], axis=2).astype(f32)

    hm_hp = np.asarray(hm_hp, f32)
    hm_hp_copy = hm_hp.copy()
    hm_hp = nms(hm_hp)
    thresh = f32(0.1)
    kps = kps.reshape(B, K, J, 2).transpose(0, 2, 1, 3).copy()  # b x J x K x 2
    kps_displacement_mean = kps.transpose(0, 2, 1, 3).reshape(B, K, J * 2).copy()

    hm_score, hm_inds, hm_ys, hm_xs = topk_channel(hm_hp, K)  # b x J x K
    if hp_offset is not None:
        off = gather_feat(np.asarray(hp_offset, f32), hm_inds.reshape(B, -1)).reshape(B, J, K, 2)
        hm_xs = hm_xs + off[:, :, :, 0]
        hm_ys = hm_ys + off[:, :, :, 1]
    else:
        hm_xs = hm_xs + f32(0.5)
        hm_ys = hm_ys + f32(0.5)

    mask = (hm_score > thresh).astype(f32)
    hm_score = (1 - mask) * -1 + mask * hm_score
    hm_ys = (1 - mask) * f32(NEG) + mask * hm_ys
    hm_xs = (1 - mask) * f32(NEG) + mask * hm_xs

    hm_kps_all = np.stack([hm_xs, hm_ys], axis=-1).astype(f32)  # b x J x K x 2
    d = kps[:, :, :, None, :] - hm_kps_all[:, :, None, :, :]     # b x J x K x K x 2
    d2 = d * d
    dist = np.sqrt(d2[..., 0] + d2[..., 1]).astype(f32)
    min_ind = dist.argmin(axis=3)
    min_dist = np.take_along_axis(dist, min_ind[..., None], axis=3)  # b x J x K x 1
    hm_score = np.take_along_axis(hm_score, min_ind, axis=2)[..., None]
    hm_kps = np.take_along_axis(hm_kps_all, min_ind[..., None].repeat(2, axis=-1), axis=2)  # b x J x K x 2

    l = bboxes[:, :, 0].reshape(B, 1, K, 1)
    t = bboxes[:, :, 1].reshape(B, 1, K, 1)
    r_ = bboxes[:, :, 2].reshape(B, 1, K, 1)
    b_ = bboxes[:, :, 3].reshape(B, 1, K, 1)
    size = np.maximum(b_ - t, r_ - l)
    m = (hm_kps[..., 0:1] < l) | (hm_kps[..., 0:1] > r_) | (hm_kps[..., 1:2] < t) | (hm_kps[..., 1:2] > b_) | \
        (hm_score < thresh) | (min_dist > size * f32(0.3))
    m = np.broadcast_to(m.astype(f32), (B, J, K, 2))
    if rep_mode == 3:
        pass
    elif rep_mode == 4:
        kps = hm_kps
    else:
        kps = (1 - m) * hm_kps + m * kps
    kps = kps.transpose(0, 2, 1, 3).reshape(B, K, J * 2).astype(f32)

    sc = scores[:, None, :, :]  # b x 1 x K x 1
    conds = [hm_kps[..., 0:1] > f32(0.8) * l, hm_kps[..., 0:1] < f32(1.2) * r_,
             hm_kps[..., 1:2] > f32(0.8) * t, hm_kps[..., 1:2] < f32(1.2) * b_,
             hm_score > thresh, min_dist < size * f32(0.5), np.broadcast_to(sc > thresh, hm_score.shape)]
    if mask_semantics == "uint8":
        m2 = np.logical_and.reduce(conds)
    elif mask_semantics == "bool":
        m2 = np.zeros_like(conds[0])  # (bool + bool + ...) == 7 is never true
    else:
        raise ValueError(mask_semantics)
    m2 = np.broadcast_to(m2.astype(f32), (B, J, K, 2))
    hm_kps_filtered = (m2 * hm_kps + (1 - m2) * f32(NEG)).astype(f32)
    hm_xs_f = hm_kps_filtered[:, :, :, 0]
    hm_ys_f = hm_kps_filtered[:, :, :, 1]

    kps_heatmap_mean = np.full((B, K, J * 2), NEG, f32)
    kps_heatmap_std = np.full((B, K, J * 2), NEG, f32)
    kps_heatmap_height = np.full((B, K, J), NEG, f32)
    if rep_mode in (1, 2):
        win = 11
        ran = win // 2
        for ib in range(B):
            for ij in range(J):
                data = hm_hp_copy[ib][ij]
                for ik in range(K):
                    x_ = hm_xs_f[ib][ij][ik]
                    y_ = hm_ys_f[ib][ij][ik]
                    if x_ == NEG or y_ == NEG:
                        continue
                    if tracking_task or refined_Kalman or rep_mode == 2:
                        big = np.zeros((data.shape[0] + 2 * ran, data.shape[1] + 2 * ran))
                        big[ran:data.shape[0] + ran, ran:data.shape[1] + ran] = data
                        weights = big[int(y_):int(y_ + 2 * ran + 1), int(x_):int(x_ + 2 * ran + 1)]
                        height, mu_x, mu_y, std_x, std_y = fitgaussian(weights)
                    else:
                        mu_x = ran
                        mu_y = ran
                        height = data[int(y_), int(x_)]
                        std_x = 1
                        std_y = 1
                    # numpy scalar arithmetic exactly as the reference writes it (decode.py:248-249)
                    kps_heatmap_mean[ib, ik, ij * 2] = f32(x_ + mu_x - ran)
                    kps_heatmap_mean[ib, ik, ij * 2 + 1] = f32(y_ + mu_y - ran)
                    kps_heatmap_std[ib, ik, ij * 2] = f32(std_x)
                    kps_heatmap_std[ib, ik, ij * 2 + 1] = f32(std_y)
                    kps_heatmap_height[ib, ik, ij] = f32(height)

    def opt_head(t, c, fn=None):
        if t is None:
            return np.zeros((B, K, c), f32)
        g = gather_feat(np.asarray(t, f32), inds)
        if fn is not None:
            g = fn(g)
        return g.reshape(B, K, c).astype(f32)

    kds = opt_head(kps_displacement_std, J * 2, lambda g: np.sqrt(np.exp(g)) * f32(balance_coefficient))
    osc = opt_head(obj_scale, 3)
    oscu = opt_head(obj_scale_uncertainty, 3, lambda g: np.sqrt(np.exp(g)))
    trk = opt_head(tracking, 2)
    trkhp = opt_head(tracking_hp, J * 2)

    return {"bboxes": bboxes, "scores": scores.astype(f32), "kps": kps, "clses": clses,
            "obj_scale": osc, "obj_scale_uncertainty": oscu, "tracking": trk, "tracking_hp": trkhp,
            "kps_displacement_mean": kps_displacement_mean.astype(f32), "kps_displacement_std": kds,
            "kps_heatmap_mean": kps_heatmap_mean, "kps_heatmap_std": kps_heatmap_std,
            "kps_heatmap_height": kps_heatmap_height}


def synth_heads(B, seed=317, H=128, W=128, tracking=False):
    """Directly-drawn decode inputs of SURVEY.md section 8(d): sparse-peak heat-maps
    (``rand()**8``), hps ~ N(0,5^2), wh ~ U(5,35), reg/hp_offset ~ U(0,1), scale ~ U(0.5,1.5)."""
    rng = np.random.RandomState(seed)
    f32 = np.float32
    d = {
        "hm": (rng.rand(B, 1, H, W) ** 8).astype(f32),
        "hm_hp": (rng.rand(B, 8, H, W) ** 8).astype(f32),
        "hps": (rng.randn(B, 16, H, W) * 5).astype(f32),
        "wh": rng.uniform(5, 35, (B, 2, H, W)).astype(f32),
        "reg": rng.rand(B, 2, H, W).astype(f32),
        "hp_offset": rng.rand(B, 2, H, W).astype(f32),
        "scale": rng.uniform(0.5, 1.5, (B, 3, H, W)).astype(f32),
    }
    if tracking:
        d["hps_uncertainty"] = rng.randn(B, 16, H, W).astype(f32)
        d["scale_uncertainty"] = rng.randn(B, 3, H, W).astype(f32)
        d["tracking"] = rng.randn(B, 2, H, W).astype(f32)
        d["tracking_hp"] = rng.randn(B, 16, H, W).astype(f32)
    return d
