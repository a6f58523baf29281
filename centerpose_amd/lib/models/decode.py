"""``object_pose_decode`` with the reference's signature and return schema (models/decode.py:72-375),
executed by the device decode kernels (centerpose_hip.h: cp_decode).

Differences that are deliberate and documented:
  * ``mask_2 == 7`` uses the torch<=1.1 semantics (AND of its 7 conditions) by default; set
    ``opt.legacy_bool_mask = True`` to reproduce what the unmodified reference computes on torch>=1.2.
  * order among exactly equal scores is (score desc, pixel index asc).
Only the detector's configuration is supported (``Inference=True``, ``wh`` and ``hm_hp`` given, one
category, 8 joints); anything else raises.
"""
import torch

from centerpose_amd import hip as _hip


def _nms(heat, kernel=3):
    """models/decode.py:17-23 (plumbing helper; the decode kernel fuses this)."""
    pad = (kernel - 1) // 2
    hmax = torch.nn.functional.max_pool2d(heat, (kernel, kernel), stride=1, padding=pad)
    return heat * (hmax == heat).float()


def _balance(opt):
    bc = getattr(opt, 'balance_coefficient', 2.0)
    if isinstance(bc, dict):
        bc = bc[opt.c]
    return float(bc)


def object_pose_decode_raw(heat, kps, wh=None, kps_displacement_std=None, obj_scale=None, obj_scale_uncertainty=None,
                           reg=None, hm_hp=None, hp_offset=None, tracking=None, tracking_hp=None, opt=None,
                           Inference=False):
    """Same contract as ``object_pose_decode`` but returns the packed device tensor [B,K,118] (hip.DET_FIELDS)."""
    if not Inference or wh is None or hm_hp is None:
        raise NotImplementedError("centerpose_hip decodes the detector's configuration: Inference=True with wh and "
                                  "hm_hp (decode.py:110-252)")
    if heat.size(1) != 1 or kps.size(1) != 16:
        raise NotImplementedError("one category / 8 joints (opts.py:435-440)")
    fit = bool(getattr(opt, 'tracking_task', False) or getattr(opt, 'refined_Kalman', False) or opt.rep_mode == 2)
    return _hip.decode_raw(heat.contiguous(), kps.contiguous(), wh.contiguous(), hm_hp.contiguous(),
                           hps_uncertainty=kps_displacement_std, scale=obj_scale,
                           scale_uncertainty=obj_scale_uncertainty, reg=reg, hp_offset=hp_offset, tracking=tracking,
                           tracking_hp=tracking_hp, K=opt.K, rep_mode=opt.rep_mode, fit_gaussian=fit,
                           balance=_balance(opt), legacy_bool_mask=bool(getattr(opt, 'legacy_bool_mask', False)))


def object_pose_decode(heat, kps, wh=None, kps_displacement_std=None, obj_scale=None, obj_scale_uncertainty=None,
                       reg=None, hm_hp=None, hp_offset=None, tracking=None, tracking_hp=None, opt=None,
                       Inference=False):
    det = object_pose_decode_raw(heat, kps, wh=wh, kps_displacement_std=kps_displacement_std, obj_scale=obj_scale,
                                 obj_scale_uncertainty=obj_scale_uncertainty, reg=reg, hm_hp=hm_hp,
                                 hp_offset=hp_offset, tracking=tracking, tracking_hp=tracking_hp, opt=opt,
                                 Inference=Inference)
    return dict(_hip.split_detections(det))
