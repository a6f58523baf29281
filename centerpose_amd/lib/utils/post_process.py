"""``object_pose_post_process`` of the reference (utils/post_process.py:12-68): output-grid detections ->
original-image coordinates, regrouped into per-detection dicts with the reference's keys."""
import numpy as np

from .image import transform_preds


def object_pose_post_process(dets, c, s, h, w, opt, Inference=False):
    coefficient = 0.32
    if 'scores' not in dets:
        return [[{}]]
    ret = []
    for i in range(dets['scores'].shape[0]):
        preds = []
        ratio = s[i] / max(w, h)
        for j in range(len(dets['scores'][i])):
            item = {}
            item['score'] = float(np.asarray(dets['scores'][i][j]).reshape(-1)[0])
            item['cls'] = int(np.asarray(dets['clses'][i][j]).reshape(-1)[0])
            item['obj_scale'] = dets['obj_scale'][i][j]
            item['obj_scale_uncertainty'] = dets['obj_scale_uncertainty'][i][j]
            item['kps_displacement_std'] = (dets['kps_displacement_std'][i, j] * ratio * coefficient
                                            ).reshape(-1, 16).flatten()
            bbox = transform_preds(dets['bboxes'][i, j].reshape(-1, 2), c[i], s[i], (w, h))
            item['bbox'] = bbox.reshape(-1, 4).flatten()
            item['ct'] = [(item['bbox'][0] + item['bbox'][2]) / 2, (item['bbox'][1] + item['bbox'][3]) / 2]
            kps = transform_preds(dets['kps'][i, j].reshape(-1, 2), c[i], s[i], (w, h))
            item['kps'] = kps.reshape(-1, 16).flatten()
            item['tracking'] = (dets['tracking'][i, j] * ratio).reshape(-1, 2).flatten()
            item['tracking_hp'] = (dets['tracking_hp'][i, j] * ratio).reshape(-1, 16).flatten()
            if Inference == True:  # noqa: E712
                m = transform_preds(dets['kps_displacement_mean'][i, j].reshape(-1, 2), c[i], s[i], (w, h))
                item['kps_displacement_mean'] = m.reshape(-1, 16).flatten()
                m = transform_preds(dets['kps_heatmap_mean'][i, j].reshape(-1, 2), c[i], s[i], (w, h))
                item['kps_heatmap_mean'] = m.reshape(-1, 16).flatten()
                item['kps_heatmap_std'] = (dets['kps_heatmap_std'][i, j] * ratio * coefficient
                                           ).reshape(-1, 16).flatten()
                item['kps_heatmap_height'] = dets['kps_heatmap_height'][i, j]
            preds.append(item)
        ret.append(preds)
    return ret
