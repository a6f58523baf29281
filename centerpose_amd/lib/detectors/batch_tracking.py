"""CenterPoseTrack for B concurrent videos (added; the reference runs one video, one frame per ``run()`` call,
detectors/base_detector.py:390-772).  One ``step`` is the reference's per-frame loop for every video at once:

    tracks of video b --(host: `_track_records`, :150-388)--> Gaussian records --(device: one cp_render_gaussians per
    input kind for the whole batch)--> pre_hm / pre_hm_hp --(device: two-frame network + decode, cp_postprocess,
    cp_pnp_from_post)--> detections + poses --(host per video: Gaussian fusion :501-544, `pnp_shell` packaging,
    ``Tracker.step`` :660-665 incl. its Kalman update, scale pool and the PnP of the filtered vertices)--> tracks.

Per-video state (``Tracker``, previous frame) is exactly the reference's; only the device stages are batched.  The
wall-clock split between the host stages and the device stages is kept in ``times`` (bench.py --workload track_e2e
reports the host fraction of a step: SURVEY 8(f) N2's open question).

``device_tracker=True`` moves the host stages to the device as well (cp_track_step, centerpose_amd/csrc/track.hip):
the track tables of all videos live in HBM, a frame is a fixed launch sequence render -> network -> decode ->
post-process -> PnP -> track update -> filtered PnP -> records of the next render, and the host only reads the tracks
when asked to (``step(..., read=True)``).
"""
import time

import numpy as np
import torch

from ..utils.pnp.cuboid_pnp_shell import finish_detection
from .base_detector import _build_tracker


class BatchedTracking(object):
    def __init__(self, detector, n_videos, device_tracker=False):
        opt = detector.opt
        self.device_tracker = bool(device_tracker)
        self.dev = None
        if not (opt.tracking_task or opt.refined_Kalman):
            raise ValueError("BatchedTracking needs opt.tracking_task or opt.refined_Kalman")
        if opt.device.type != 'cuda':
            raise RuntimeError("BatchedTracking runs on the HIP device only")
        self.det = detector
        self.n = int(n_videos)
        # the same choice as run() makes (base_detector.py:53-57: refined_Kalman wins when both flags are set)
        self.trackers = [_build_tracker(opt) for _ in range(self.n)]
        self._vmeta = None
        self.pre_images = None
        self.frames = 0
        self.times = {'host_records': 0.0, 'device': 0.0, 'host_tracks': 0.0, 'steps': 0}

    def reset(self):
        for t in self.trackers:
            t.reset()
        if self.dev is not None:
            self.dev.reset()
            self._vmeta = None   # the next frame's meta is taken as the videos' meta again
        self.pre_images = None
        self.frames = 0

    def _render(self, recs, C, ih, iw):
        from centerpose_amd import hip as _hip

        return _hip.render_gaussians(np.array(recs, np.float64).reshape(-1, 5), C, ih, iw, self.det.opt.device)

    def _step_device(self, images, metas, read):
        """The whole frame on the device (cp_track_step); the host builds nothing unless ``read``."""
        from centerpose_amd import hip as _hip

        det, opt, B = self.det, self.det.opt, self.n
        t0 = time.time()
        images = images.to(opt.device)
        if self.dev is None:
            if any('pre_dets' in m for m in metas):
                raise RuntimeError("device tracker: seeding from meta['pre_dets'] is a host-tracker feature")
            self._vmeta = np.asarray(_hip.track_vmeta(metas), np.float64)
            self.dev = _hip.DeviceTracker(B, _hip.track_params_from_opt(opt, K=opt.K), self._vmeta, opt.device,
                                          metas[0]['inp_height'], metas[0]['inp_width'])
        elif self._vmeta is None:  # first frame after reset()
            self._vmeta = np.asarray(_hip.track_vmeta(metas), np.float64)
            self.dev.vmeta.copy_(torch.from_numpy(self._vmeta).reshape(B, 16))
        else:
            # the device keeps the per-video meta (trans_input, frame size, intrinsics).  The reference hands the tracker the
            # CURRENT frame's meta (base_detector.py:441 with --refined_Kalman, and pnp_shell uses the frame's meta), so a video
            # whose meta changes from frame to frame is followed by re-uploading the rows (one small copy, no synchronisation).
            vm = np.asarray(_hip.track_vmeta(metas), np.float64)
            if not np.array_equal(vm, self._vmeta):
                self._vmeta = vm
                self.dev.vmeta.copy_(torch.from_numpy(vm).reshape(B, 16))
        if self.pre_images is None:
            self.pre_images = images
        pre_hm = pre_hm_hp = None
        if opt.pre_hm or opt.pre_hm_hp:
            hm, hp = self.dev.render()
            pre_hm = hm if opt.pre_hm else None
            pre_hm_hp = hp if opt.pre_hm_hp else None
        det._skip_host_dets = True
        try:
            det.process(images, self.pre_images if opt.tracking_task else None, pre_hm, pre_hm_hp, None)
        finally:
            det._skip_host_dets = False
        rec, cnt, poses = det.post_pnp_device(metas)
        # a refused cp_track_step raises here, BEFORE the device state has moved: the host's view (previous frame, frame count)
        # must not move either.  The periodic overflow check raises AFTER the device has advanced, so it runs once the host's
        # view has followed.
        self.dev.step(rec, cnt, poses, check=False)
        self.pre_images = images
        self.frames += 1
        self.dev.check_due()
        outs = None
        if read:
            outs = []
            for arr in self.dev.read():
                tracks = [_hip.track_record_to_dict(r) for r in arr]
                boxes = [(t['kps_pnp_kf'], t['kps_3d_cam_kf'], t['obj_scale'], t['kps_ori_kf'], t) for t in tracks
                         if t['in_boxes']]
                outs.append({'results': tracks, 'boxes': boxes})
        elif opt.device.type == 'cuda':
            torch.cuda.synchronize()
        self.times['device'] += time.time() - t0
        self.times['steps'] += 1
        return outs

    def step(self, images, metas, read=True):
        """images [B,3,H,W] pre-processed frames (frame t of each video), metas: B meta dicts as ``pre_process`` builds
        them (+ 'camera_matrix', 'id').  Returns a list of B dicts {'results': tracks, 'boxes': boxes} (``read=False``
        with the device tracker: None, the tracks stay in HBM)."""
        det, opt = self.det, self.det.opt
        B = self.n
        if images.shape[0] != B or len(metas) != B:
            raise ValueError("expected one frame and one meta per video")
        if self.device_tracker:
            return self._step_device(images, metas, read)
        t0 = time.time()
        images = images.to(opt.device)
        first = self.pre_images is None
        if first:
            self.pre_images = images  # the first frame of a video is its own predecessor (:447-449)
        for b, tr in enumerate(self.trackers):
            if first or opt.refined_Kalman or (opt.tracking_task and (opt.gt_pre_hm_hmhp or (
                    opt.gt_pre_hm_hmhp_first and metas[b].get('id') == 0))):
                tr.init_track(metas[b])
        pre_hm = pre_hm_hp = None
        ih, iw = metas[0]['inp_height'], metas[0]['inp_width']
        if opt.tracking_task and (opt.pre_hm or opt.pre_hm_hp):
            hm_all, hp_all = [], []
            for b, tr in enumerate(self.trackers):
                hm, hp, _ = det._track_records(tr.tracks, metas[b], opt.pre_hm, opt.pre_hm_hp)
                hm_all += [(b + c, x, y, r, k) for c, x, y, r, k in hm]          # one plane per video
                hp_all += [(8 * b + c, x, y, r, k) for c, x, y, r, k in hp]      # eight planes per video
            t1 = time.time()
            if opt.pre_hm:
                pre_hm = self._render(hm_all, B, ih, iw).view(B, 1, ih, iw)
            if opt.pre_hm_hp:
                pre_hm_hp = self._render(hp_all, 8 * B, ih, iw).view(B, 8, ih, iw)
        else:
            t1 = time.time()
        det._skip_host_dets = True
        try:
            det.process(images, self.pre_images if opt.tracking_task else None, pre_hm, pre_hm_hp, None)
        finally:
            det._skip_host_dets = False
        all_results = det.post_process_merge_device(metas)  # cp_postprocess (+ cp_pnp_from_post), one copy to the host
        raw = det.pnp_dev.cpu().numpy() if getattr(det, 'pnp_dev', None) is not None else None
        t2 = time.time()
        outs = []
        for b, tr in enumerate(self.trackers):
            results = all_results[b]
            for d in results:
                m_, s_ = det._fuse_keypoints(d)
                d['kps_fusion_mean'] = np.array(m_)
                d['kps_fusion_std'] = np.array(s_)
            boxes = []
            if opt.use_pnp == True and raw is not None:  # noqa: E712
                for k, d in enumerate(results):
                    r = raw[b, k]
                    if int(r[0]) != 1:
                        continue
                    proj = r[8:24].reshape(8, 2).copy()
                    loc, quat = (list(r[4:7]), r[24:28].copy()) if opt.show_axes else (list(r[28:31]), r[31:35].copy())
                    ret = finish_detection(opt, metas[b], d, d['obj_scale'], loc, quat, proj)
                    if ret is not None:
                        boxes.append(ret)
            results, boxes = tr.step(results, boxes)
            outs.append({'results': results, 'boxes': boxes})
        t3 = time.time()
        if opt.tracking_task:
            self.pre_images = images
        self.frames += 1
        self.times['host_records'] += t1 - t0
        self.times['device'] += t2 - t1
        self.times['host_tracks'] += t3 - t2
        self.times['steps'] += 1
        return outs
