"""``ObjectPoseDetector`` (detectors/object_pose.py:126-197) and ``soft_nms_nvidia`` (:27-124)."""
import time

import numpy as np
import torch

from ..models.decode import object_pose_decode_raw
from ..utils.post_process import object_pose_post_process
from .base_detector import BaseDetector


def soft_nms_nvidia(src_boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """Selection-sort style soft-NMS over an array of detection dicts, in place, returning the kept
    index range — restated from the reference (:27-124) including its whole-dict swaps.
    method 1 linear, 2 gaussian (weight = exp(-ov^2 / sigma)), else hard NMS at Nt."""
    N = src_boxes.shape[0]

    def swap_rest(a, b):
        for key in src_boxes[0]:
            if key != 'bbox' and key != 'score':
                src_boxes[a][key], src_boxes[b][key] = src_boxes[b][key], src_boxes[a][key]

    for i in range(N):
        maxscore, maxpos = src_boxes[i]['score'], i
        tx1, ty1, tx2, ty2 = src_boxes[i]['bbox'][0:4]
        ts = src_boxes[i]['score']
        for pos in range(i + 1, N):
            if maxscore < src_boxes[pos]['score']:
                maxscore, maxpos = src_boxes[pos]['score'], pos
        src_boxes[i]['bbox'] = src_boxes[maxpos]['bbox']
        src_boxes[i]['score'] = src_boxes[maxpos]['score']
        src_boxes[maxpos]['bbox'] = [tx1, ty1, tx2, ty2]
        src_boxes[maxpos]['score'] = ts
        swap_rest(i, maxpos)
        tx1, ty1, tx2, ty2 = src_boxes[i]['bbox'][0:4]
        pos = i + 1
        while pos < N:
            x1, y1, x2, y2 = src_boxes[pos]['bbox'][0:4]
            area = (x2 - x1 + 1) * (y2 - y1 + 1)
            iw = (min(tx2, x2) - max(tx1, x1) + 1)
            if iw > 0:
                ih = (min(ty2, y2) - max(ty1, y1) + 1)
                if ih > 0:
                    ua = float((tx2 - tx1 + 1) * (ty2 - ty1 + 1) + area - iw * ih)
                    ov = iw * ih / ua
                    if method == 1:
                        weight = 1 - ov if ov > Nt else 1
                    elif method == 2:
                        weight = np.exp(-(ov * ov) / sigma)
                    else:
                        weight = 0 if ov > Nt else 1
                    src_boxes[pos]['score'] = weight * src_boxes[pos]['score']
                    if src_boxes[pos]['score'] < threshold:
                        src_boxes[pos]['bbox'] = src_boxes[N - 1]['bbox']
                        src_boxes[pos]['score'] = src_boxes[N - 1]['score']
                        swap_rest(pos, N - 1)
                        N = N - 1
                        pos = pos - 1
            pos = pos + 1
    return [i for i in range(N)]


class ObjectPoseDetector(BaseDetector):
    def __init__(self, opt):
        super(ObjectPoseDetector, self).__init__(opt)
        self.flip_idx = opt.flip_idx

    def process(self, images, pre_images=None, pre_hms=None, pre_hm_hp=None, pre_inds=None, return_time=False):
        """object_pose.py:131-165.  The sigmoid of hm / hm_hp (:136-138) is fused into the head epilogue."""
        with torch.no_grad():
            output = self.model._engine().forward(images, pre_images, pre_hms, pre_hm_hp,
                                                  sigmoid_hm=not self.opt.mse_loss)
            output = dict(output)
            if self.opt.mse_loss:  # :136-138: hm is always sigmoided, hm_hp only without mse_loss
                output['hm'] = output['hm'].sigmoid_()
            output.update({'pre_inds': pre_inds})
            o = self.opt
            wh = output['wh'] if o.reg_bbox else None
            reg = output['reg'] if o.reg_offset else None
            hps_unc = output['hps_uncertainty'] if o.hps_uncertainty else None
            hm_hp = output['hm_hp'] if o.hm_hp else None
            hp_offset = output['hp_offset'] if o.reg_hp_offset else None
            obj_scale = output['scale'] if o.obj_scale else None
            obj_scale_unc = output['scale_uncertainty'] if o.obj_scale_uncertainty else None
            tracking = output['tracking'] if 'tracking' in o.heads else None
            tracking_hp = output['tracking_hp'] if 'tracking_hp' in o.heads else None
            if images.is_cuda:
                torch.cuda.synchronize()
            forward_time = time.time()
            raw = object_pose_decode_raw(output['hm'], output['hps'], wh=wh, kps_displacement_std=hps_unc,
                                         obj_scale=obj_scale, obj_scale_uncertainty=obj_scale_unc, reg=reg,
                                         hm_hp=hm_hp, hp_offset=hp_offset, tracking=tracking, tracking_hp=tracking_hp,
                                         opt=o, Inference=True)
            self.raw_dets = raw  # packed [B,K,118] on the device, for the device post-process (run_batch)
            if getattr(self, '_skip_host_dets', False):
                dets = None
            else:  # one device->host copy, then the reference's 13-key dict as numpy views
                host = raw.detach().cpu().numpy()
                from centerpose_amd import hip as _hip
                dets = {k: host[..., off:off + w] for k, (off, w) in _hip.DET_FIELDS.items()}
        if return_time:
            return output, dets, forward_time
        return output, dets

    def post_process(self, dets, meta, scale=1):
        dets = object_pose_post_process(dets.copy(), [meta['c']], [meta['s']], meta['out_height'], meta['out_width'],
                                        self.opt, Inference=True)
        if scale != 1:
            for i in range(len(dets[0])):
                for k in ['bbox', 'kps', 'kps_displacement_std', 'tracking', 'tracking_hp', 'kps_displacement_mean',
                          'kps_heatmap_mean']:
                    if k in dets[0][i]:
                        dets[0][i][k] = (np.array(dets[0][i][k], np.float32) / scale).tolist()
        return dets[0]

    def merge_outputs(self, detections):
        results = np.array([det for det in detections[0] if det['score'] > self.opt.vis_thresh])
        if self.opt.nms or len(self.opt.test_scales) > 1:
            keep = soft_nms_nvidia(results, Nt=0.5, method=2, threshold=self.opt.vis_thresh)
            results = results[keep]
        return results

    def post_pnp_device(self, metas):
        """cp_postprocess (+ cp_pnp_from_post) of the last ``process`` call, results left on the device:
        (records [B,K,120] float64, counts [B] int32, poses [B,K,40] float64 or None)."""
        from centerpose_amd import hip as _hip
        from ..utils.image import get_affine_transform

        B = len(metas)
        arr = np.zeros((B, 8), np.float64)
        for b, meta in enumerate(metas):
            w, h = meta['out_width'], meta['out_height']
            arr[b, :6] = get_affine_transform(meta['c'], meta['s'], 0, (w, h), inv=1).reshape(-1)
            arr[b, 6] = meta['s'] / max(w, h)
        use_nms = bool(self.opt.nms or len(self.opt.test_scales) > 1)
        rec, cnt = _hip.postprocess(self.raw_dets, arr, self.opt.vis_thresh, use_nms)
        self.post_dev = (rec, cnt)  # stays on the device for cp_pnp_from_post (run_batch)
        self.pnp_dev = None
        if self.opt.use_pnp == True and self.opt.rep_mode != 2 and all('camera_matrix' in m for m in metas):  # noqa: E712
            cams = np.array([[np.asarray(m['camera_matrix'], np.float64)[0, 0], np.asarray(m['camera_matrix'], np.float64)[1, 1],
                              np.asarray(m['camera_matrix'], np.float64)[0, 2], np.asarray(m['camera_matrix'], np.float64)[1, 2]]
                             for m in metas])
            # enqueued behind the post-process: a single device->host copy can then carry both results
            self.pnp_dev = _hip.pnp_from_post(rec, cnt, torch.from_numpy(cams).to(rec.device), self.opt.rep_mode)
        return rec, cnt, self.pnp_dev

    def post_process_merge_device(self, metas):
        """post_process + merge_outputs of every image of the last ``process`` call in one device launch
        (cp_postprocess); returns a list of per-image numpy arrays of detection dicts like ``merge_outputs``."""
        from centerpose_amd import hip as _hip

        B = len(metas)
        rec, cnt, _ = self.post_pnp_device(metas)
        rec = rec.cpu().numpy()
        cnt = cnt.cpu().numpy()
        f32_fields = ('obj_scale', 'obj_scale_uncertainty', 'kps_displacement_std', 'tracking', 'tracking_hp',
                      'kps_heatmap_std', 'kps_heatmap_height')
        out = []
        for b in range(B):
            items = []
            for r in rec[b, :int(cnt[b])]:
                item = {}
                for k, (off, w) in _hip.POST_FIELDS.items():
                    v = r[off:off + w]
                    if k == 'score':
                        item[k] = float(v[0])
                    elif k == 'cls':
                        item[k] = int(v[0])
                    elif k == 'ct':
                        item[k] = [v[0], v[1]]
                    elif k in f32_fields:
                        item[k] = v.astype(np.float32)
                    else:
                        item[k] = v.copy()
                items.append(item)
            out.append(np.array(items))
        return out

    def show_results(self, debugger, image, results):
        print('[centerpose_hip] %d detection(s) (drawing is not part of this library)' % len(results))
