"""CenterPoseTrack's host-side track bookkeeping with the interface of the reference ``Tracker``
(utils/tracker.py:14-302): ``Tracker(opt)``, ``.init_track(meta)``, ``.reset()``, ``.tracks``,
``.step(dets, boxes=[]) -> (tracks, boxes)``.  Detections and tracks are the reference's per-object dicts.

What a step does (reference line numbers in brackets):
  * association [125-176]: squared distance between each track centre and each detection centre displaced by its
    ``tracking`` offset; a pair is forbidden when the distance exceeds either box area or the classes differ; greedy
    (detection order, nearest free track) or Hungarian assignment;
  * matched detections inherit id / activity, age 1; their 32-state keypoint Kalman filter (x, y, vx, vy per vertex,
    everything observed, R from the fused keypoint std and ``opt.R``) is predicted + updated, their scale sample
    joins the track's pool [178-199];  unmatched detections above ``new_thresh`` start tracks [201-218];
    unmatched tracks coast (unchanged) until ``max_age`` [220-236];
  * filter read-out [238-294]: filtered vertex means / stds, a confidence from the combined std (vertices under
    0.15 are dropped), precision-weighted scale fusion, and -- with ``use_pnp`` -- a fresh PnP on the filtered vertices.
"""
import numpy as np

from .kalman import KalmanFilter

_FORBIDDEN = 1e18


def _greedy(cost):
    """Detections in order, each takes its nearest still-free admissible track [305-314]."""
    cost = cost.copy()
    pairs = []
    if cost.shape[1]:
        for d in range(cost.shape[0]):
            t = int(cost[d].argmin())
            if cost[d, t] < 1e16:
                cost[:, t] = _FORBIDDEN
                pairs.append((d, t))
    return np.array(pairs, np.int32).reshape(-1, 2)


def _hungarian(cost, solver=1):
    """``sklearn.utils.linear_assignment_.linear_assignment`` (the reference's import, tracker.py:6,157; scikit-learn 0.22.2,
    a module that no longer exists in scikit-learn >= 0.23): the (row, column) pairs of the optimal assignment as an [n, 2]
    array sorted by row.  Solved by the library's restatement of that module's Munkres state machine (``solver`` 1,
    cp_linear_assignment -- the routine the device tracker runs) or of scipy's rectangular LSAP (``solver`` 2): the same
    optimum value, possibly different pairs among tied / forbidden entries (include/centerpose_hip.h: cp_track_params)."""
    from centerpose_amd import hip

    return hip.linear_assignment(np.asarray(cost, np.float64), solver)


def _area(box):
    return (box[2] - box[0]) * (box[3] - box[1])


class Tracker(object):
    def __init__(self, opt):
        self.opt = opt
        self.meta = None
        self.reset()

    def reset(self):
        self.id_count = 0
        self.tracks = []

    # ---- per-track filter state -------------------------------------------------------------------------------
    def _noise(self, det):
        """Observation covariance of one detection: (x, y) variances from the fused keypoint std, opt.R for (vx, vy)."""
        std = np.asarray(det['kps_fusion_std'], float).reshape(8, 2)
        diag = np.concatenate([std ** 2, np.full((8, 2), float(self.opt.R))], 1).reshape(32)
        return np.diag(diag)

    @staticmethod
    def _observation(det):
        """(x, y, vx, vy) per vertex; the velocity is minus the predicted displacement to the previous frame."""
        pos = np.asarray(det['kps_fusion_mean'], float).reshape(8, 2)
        vel = -np.asarray(det['tracking_hp'], float).reshape(8, 2)
        return np.concatenate([pos, vel], 1).reshape(32)

    def init_kf(self, det):
        kf = KalmanFilter(dim_x=32, dim_z=32)
        kf.H = np.eye(32)
        for v in range(8):  # constant-velocity model, unit time step
            kf.F[4 * v, 4 * v + 2] = 1
            kf.F[4 * v + 1, 4 * v + 3] = 1
        kf.R = self._noise(det)
        kf.P = kf.R  # the reference aliases the two matrices; predict() replaces P before either changes
        kf.x = self._observation(det).reshape(32, 1)
        return kf

    def update_kf(self, det):
        det['kf'].update(self._observation(det), R=self._noise(det))

    @staticmethod
    def update_scale_pool(det):
        """Precision-weighted (Bayesian) fusion of every (scale, uncertainty) sample of the track [98-110]."""
        prec = np.zeros(3)
        acc = np.zeros(3)
        for mean, unc in det['scale_pool']:
            w = np.array(unc) ** -2
            prec += w
            acc += w * np.array(mean)
        std = prec ** -0.5
        return acc * std ** 2, std

    def _start(self, item):
        self.id_count += 1
        item['tracking_id'] = self.id_count
        item['age'] = 1
        item['active'] = 1
        if self.opt.kalman == True:  # noqa: E712
            item['kf'] = self.init_kf(item)
        if self.opt.scale_pool == True:  # noqa: E712
            item['scale_pool'] = [(item['obj_scale'], item['obj_scale_uncertainty'])]
        return item

    def init_track(self, meta):
        """[21-49] seed the tracks from ``meta['pre_dets']`` (external first-frame annotations), if present."""
        self.meta = meta
        seeds = []
        if 'pre_dets' in meta:
            seeds = meta['pre_dets']
            self.reset()
        for item in seeds:
            if item['score'] > self.opt.new_thresh:
                if 'ct' not in item:
                    b = item['bbox']
                    item['ct'] = [(b[0] + b[2]) / 2, (b[1] + b[3]) / 2]
                self.tracks.append(self._start(item))

    # ---- one frame --------------------------------------------------------------------------------------------
    P_STRIDE = 4            # diagonal entries of P read as vertex v's (x, y) variance: P[s*v], P[s*v + 1]
    PNP_OPENCV_RETURN = True   # the filtered PnP follows opt.show_axes (Tracker) or the default frame (baseline)

    def _det_centres(self, dets):
        """Detection centres moved back to the previous frame by the predicted ``tracking`` offset [130]."""
        return [np.asarray(d['ct']) + np.asarray(d['tracking']) for d in dets]

    def _track_centres(self):
        return [t['ct'] for t in self.tracks]

    def _associate(self, dets):
        n, m = len(dets), len(self.tracks)
        det_ct = np.array(self._det_centres(dets), np.float32).reshape(n, 2)
        trk_ct = np.array(self._track_centres(), np.float32).reshape(m, 2)
        trk_area = np.array([_area(t['bbox']) for t in self.tracks], np.float32)
        det_area = np.array([_area(d['bbox']) for d in dets], np.float32)
        trk_cls = np.array([t['cls'] for t in self.tracks], np.int32)
        det_cls = np.array([d['cls'] for d in dets], np.int32)
        if n and m:
            cost = ((trk_ct[None, :, :] - det_ct[:, None, :]) ** 2).sum(2)
        else:
            cost = np.zeros((n, m), np.float32)
        bad = (cost > trk_area[None, :]) | (cost > det_area[:, None]) | (det_cls[:, None] != trk_cls[None, :])
        cost = cost + bad * _FORBIDDEN
        if self.opt.hungarian:
            cost[cost > _FORBIDDEN] = _FORBIDDEN
            # opt.hungarian is the reference's flag (opts.py); `hungarian_solver` = 'scipy' selects the other optimum finder
            pairs = _hungarian(cost, 2 if getattr(self.opt, 'hungarian_solver', 'munkres') == 'scipy' else 1)
        else:
            pairs = _greedy(cost)
        free_d = [d for d in range(n) if d not in pairs[:, 0]]
        free_t = [t for t in range(m) if t not in pairs[:, 1]]
        if self.opt.hungarian:  # the optimal assignment may still use a forbidden pair: undo those
            ok = []
            for d, t in pairs:
                if cost[d, t] > 1e16:
                    free_d.append(d)
                    free_t.append(t)
                else:
                    ok.append((d, t))
            pairs = np.array(ok).reshape(-1, 2)
        return pairs, free_d, free_t

    def step(self, dets, boxes=[]):
        opt = self.opt
        if opt.use_pnp == True and boxes:  # noqa: E712  PnP results carry their detection [116-123]
            dets = []
            for b in boxes:
                d = b[4]
                d['kps_pnp'], d['kps_3d_cam'], d['kps_ori'] = b[0], b[1], b[3]
                dets.append(d)

        pairs, free_d, free_t = self._associate(dets)
        out = []
        for d, t in pairs:
            det, old = dets[d], self.tracks[t]
            det['tracking_id'] = old['tracking_id']
            det['age'] = 1
            det['active'] = old['active'] + 1
            if opt.kalman == True:  # noqa: E712
                det['kf'] = old['kf']
                det['kf'].predict()
                self.update_kf(det)
            if opt.scale_pool == True:  # noqa: E712
                det['scale_pool'] = old['scale_pool']
                det['scale_pool'].append((det['obj_scale'], det['obj_scale_uncertainty']))
            out.append(det)
        for d in free_d:
            if dets[d]['score'] > opt.new_thresh:
                out.append(self._start(dets[d]))
        for t in free_t:
            trk = self.tracks[t]
            if trk['age'] < opt.max_age:  # coast: assumed not to move
                trk['age'] += 1
                trk['active'] = 0
                trk['bbox'] = [trk['bbox'][0], trk['bbox'][1], trk['bbox'][2], trk['bbox'][3]]
                trk['ct'] = [trk['ct'][0], trk['ct'][1]]
                out.append(trk)

        if opt.kalman == True or opt.scale_pool == True:  # noqa: E712
            if opt.use_pnp == True:  # noqa: E712
                boxes = []
            lo, hi = opt.conf_border[opt.c][0], opt.conf_border[opt.c][1]
            for trk in out:
                kps = trk['kps']
                conf = []
                if opt.kalman == True:  # noqa: E712
                    kf = trk['kf']
                    trk['kps_mean_kf'] = np.array([kf.x[4 * v:4 * v + 2] for v in range(8)])
                    kps = trk['kps_mean_kf']
                    dg, st = np.diag(kf.P), self.P_STRIDE
                    var = np.array([[dg[st * v], dg[st * v + 1]] for v in range(8)])
                    trk['kps_std_kf'] = list(np.sqrt(var).reshape(-1))
                    # confidence decays exponentially with the combined std, 0.15 at conf_border[0] [254-262]
                    comb = np.sqrt(var.sum(1))
                    conf = list(np.maximum(1 - np.exp(np.log(0.15) / (lo - hi)) ** (comb - hi), 0))
                    for v in range(8):
                        if conf[v] < 0.15:
                            kps[v][0] = -10000
                            kps[v][1] = -10000
                scale = trk['obj_scale']
                if opt.scale_pool == True:  # noqa: E712
                    trk['obj_scale_kf'], trk['obj_scale_uncertainty_kf'] = self.update_scale_pool(trk)
                    scale = trk['obj_scale_kf']
                if opt.use_pnp == True:  # noqa: E712
                    from .pnp.cuboid_pnp_shell import pnp_shell

                    res = pnp_shell(opt, self.meta, trk, kps, scale,
                                    OPENCV_RETURN=opt.show_axes if self.PNP_OPENCV_RETURN else False)
                    if res is not None:
                        if np.sum(conf) / 8 > 0.25:
                            boxes.append(res)
                        trk['kps_pnp_kf'], trk['kps_3d_cam_kf'], trk['kps_ori_kf'] = res[0], res[1], res[3]
        self.tracks = out
        return out, boxes


class Tracker_baseline(Tracker):
    """``opt.refined_Kalman``: CenterPose + a position-only Kalman filter (utils/tracker_baseline.py:14-310).  Differences
    from ``Tracker`` (reference lines of tracker_baseline.py in brackets): only (x, y) of each vertex is observed,
    H is 16 x 32 [56-63]; the initial covariance block of a vertex is the 2 x 2 broadcast of its two variances [70];
    the scale pool is a plain average with a placeholder uncertainty [94-101, 267-269]; association uses the raw
    detection centres against track centres advanced by the mean filtered velocity [121, 134-140]; the read-out
    takes P[2v], P[2v+1] [251-254]; the filtered PnP always returns the default (OpenGL) frame [276]."""
    P_STRIDE = 2
    PNP_OPENCV_RETURN = False

    def init_kf(self, det):
        kf = KalmanFilter(dim_x=32, dim_z=16)
        kf.H = np.zeros((16, 32))
        std = np.asarray(det['kps_fusion_std'], float)
        for v in range(8):
            kf.H[2 * v, 4 * v] = 1
            kf.H[2 * v + 1, 4 * v + 1] = 1
            kf.F[4 * v, 4 * v + 2] = 1
            kf.F[4 * v + 1, 4 * v + 3] = 1
            kf.R[2 * v, 2 * v] *= std[2 * v] ** 2
            kf.R[2 * v + 1, 2 * v + 1] *= std[2 * v + 1] ** 2
            # a 2-vector assigned to a 2 x 2 block broadcasts over its rows: [[vx, vy], [vx, vy]]
            kf.P[4 * v:4 * v + 2, 4 * v:4 * v + 2] = [kf.R[2 * v, 2 * v], kf.R[2 * v + 1, 2 * v + 1]]
        mean = np.asarray(det['kps_fusion_mean'], float)
        for v in range(8):
            kf.x[4 * v:4 * v + 2] = mean[2 * v:2 * v + 2].reshape(-1, 1)
        return kf

    def update_kf(self, det):
        z = np.asarray(det['kps_fusion_mean'], float).reshape(16).copy()
        R = np.diag(np.asarray(det['kps_fusion_std'], float).reshape(16) ** 2)
        det['kf'].update(z, R=R)

    @staticmethod
    def update_scale_pool(det):
        mean = np.zeros(3)
        for sample, _ in det['scale_pool']:
            mean += np.array(sample)
        return mean / len(det['scale_pool']), np.array([0, 0, 0])  # the reference stores a placeholder uncertainty

    def _det_centres(self, dets):
        return [d['ct'] for d in dets]

    def _track_centres(self):
        out = []
        for t in self.tracks:
            v = np.array([0, 0]).astype('float64')
            for i in range(8):
                v += np.array(t['kf'].x[4 * i + 2:4 * i + 4]).flatten()
            out.append(t['ct'] + v / 8)
        return out
