"""``object_pose_post_process`` of the reference (utils/post_process.py:12-68): output-grid detections ->
original-image coordinates, regrouped into per-detection dicts with the reference's keys.

Table-driven: every output key is one of four kinds of the decode output of the same object
  xy     points through the image's inverse affine (``transform_preds``; (-10000, -10000) passes through)
  ratio  lengths scaled by s / max(w, h)            ratio_c  the same times the reference's 0.32 coefficient
  raw    copied
(the device version of the same table is centerpose_amd/csrc/post.hip).
"""
import numpy as np

from .image import transform_preds

_COEFFICIENT = 0.32
# output key -> (decode key, kind, flattened width); order = the reference's insertion order
_ALWAYS = (('obj_scale', 'obj_scale', 'raw', 0), ('obj_scale_uncertainty', 'obj_scale_uncertainty', 'raw', 0),
           ('kps_displacement_std', 'kps_displacement_std', 'ratio_c', 16), ('bbox', 'bboxes', 'xy', 4),
           ('ct', None, 'centre', 0), ('kps', 'kps', 'xy', 16), ('tracking', 'tracking', 'ratio', 2),
           ('tracking_hp', 'tracking_hp', 'ratio', 16))
_INFERENCE = (('kps_displacement_mean', 'kps_displacement_mean', 'xy', 16), ('kps_heatmap_mean', 'kps_heatmap_mean', 'xy', 16),
              ('kps_heatmap_std', 'kps_heatmap_std', 'ratio_c', 16), ('kps_heatmap_height', 'kps_heatmap_height', 'raw', 0))


def _field(item, dets, i, j, spec, c, s, wh, ratio):
    key, src, kind, width = spec
    if kind == 'centre':
        b = item['bbox']
        item[key] = [(b[0] + b[2]) / 2, (b[1] + b[3]) / 2]
    elif kind == 'raw':
        item[key] = dets[src][i][j]
    elif kind == 'xy':
        item[key] = transform_preds(dets[src][i, j].reshape(-1, 2), c, s, wh).reshape(-1, width).flatten()
    else:
        v = dets[src][i, j] * ratio
        if kind == 'ratio_c':
            v = v * _COEFFICIENT
        item[key] = v.reshape(-1, width).flatten()


def object_pose_post_process(dets, c, s, h, w, opt, Inference=False):
    if 'scores' not in dets:
        return [[{}]]
    specs = _ALWAYS + (_INFERENCE if Inference == True else ())  # noqa: E712
    ret = []
    for i in range(dets['scores'].shape[0]):
        ratio = s[i] / max(w, h)
        preds = []
        for j in range(len(dets['scores'][i])):
            item = {'score': float(np.asarray(dets['scores'][i][j]).reshape(-1)[0]),
                    'cls': int(np.asarray(dets['clses'][i][j]).reshape(-1)[0])}
            for spec in specs:
                _field(item, dets, i, j, spec, c[i], s[i], (w, h), ratio)
            preds.append(item)
        ret.append(preds)
    return ret
